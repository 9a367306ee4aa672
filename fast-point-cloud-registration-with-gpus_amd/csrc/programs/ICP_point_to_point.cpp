// ICP_point_to_point -- src/ICP_point_to_point.cu (synthetic 128x128, MAX_ITER 40, tol 1e-6), and with
// --bunny / --hall the dataset variants src/CUDA/GPU_point_to_point_bunny.cu and
// src/CUDA/GPU_point_to_point_real.cu (MAX_ITER 100 there, :18).  stdout follows :290,:428-433.
#include "common.h"

int main(int argc, char** argv)
{
    Args a;
    if (!parse_args(argc, argv, a, "ICP_point_to_point")) return 2;
    icp_ctx* ctx = nullptr;
    ICP_CHECK(icp_create(0, &ctx));
    std::vector<float> D, M;
    double conv_ms = 0.0;
    const int n = build_clouds_f32(ctx, a, 128, D, M, &conv_ms);
    if (n < 0) { std::fprintf(stderr, "input: %s (%s)\n", icp_strerror(n), icp_last_error()); return -1; }
    const bool dataset = !a.hall_packets.empty() || !a.bunny.empty();
    const int max_iter = a.max_iter > 0 ? a.max_iter : (dataset ? 100 : 40);
    if (!a.hall_packets.empty()) std::printf("Conversion kernel's elapsed time: %.3f ms\n", conv_ms);

    ICP_CHECK(icp_set_model(ctx, M.data(), n, ICP_F32));
    ICP_CHECK(icp_set_moving(ctx, D.data(), n, ICP_F32));
    ICP_CHECK(icp_nn_match_resident(ctx, nullptr));
    int blocks = 0, threads = 0;
    ICP_CHECK(icp_nn_launch_info(ctx, nullptr, &blocks, &threads, nullptr, nullptr));
    std::printf("Grid Size: %d, Block Size: %d\n", blocks, threads);

    icp_params prm{max_iter, 0.000001, 0, ICP_F32, ICP_POINT_TO_POINT};
    std::vector<double> err((size_t)max_iter + 1, 0.0);
    icp_result res{};
    res.err = err.data();
    ICP_CHECK(icp_set_profiling(ctx, 1));
    if (a.trace.empty()) {
        ICP_CHECK(icp_point_to_point(ctx, D.data(), n, M.data(), n, &prm, &res));
    } else {
        // step-wise so that the cloud of every iteration can be captured (pt_total of src/ICP_CPU.c:197-201,254)
        std::vector<std::vector<float>> pt_total;
        ICP_CHECK(icp_set_moving(ctx, D.data(), n, ICP_F32));
        ICP_CHECK(icp_loop_begin(ctx, &prm));
        const auto l0 = std::chrono::steady_clock::now();
        int done = 0, passes = 0, seen = 0;
        while (!done) {
            ICP_CHECK(icp_loop_enqueue(ctx));
            ICP_CHECK(icp_loop_complete(ctx, &done));
            ICP_CHECK(icp_loop_state(ctx, &res.iterations, &passes, err.data(), (int)err.size(), res.T));
            if (passes > seen) {
                pt_total.emplace_back(3 * (size_t)n);
                ICP_CHECK(icp_get_moving(ctx, pt_total.back().data()));
                seen = passes;
            }
        }
        res.passes = passes;
        res.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - l0).count();
        ICP_CHECK(icp_loop_phase_seconds(ctx, &res.seconds_nn, &res.seconds_host));
        if (!write_trace(a.trace, D, M, pt_total, err.data(), n)) { std::perror("trace file"); return -1; }
    }

    std::printf("Error:\n");
    print_sarray(err.data(), res.iterations + 1);
    if (!a.hall_packets.empty()) {
        // src/CUDA/GPU_point_to_point_real.cu:386-403: `iteration + 1` and four phase lines.  The reference times four host
        // phases between synchronisations; here the transformation and the error estimation of pass k are the front end of
        // matching kernel k + 1 -- the same launch -- so their time is inside the matching line and their own lines read 0;
        // the minimisation line is the host's share of the iterations (error, stop rule, 3x3 SVD).
        const double seconds = res.seconds_total;
        std::printf("\nThe ICP algorithm was computed in %.4f ms with %d iterations\n\n", 1000.0 * seconds, res.iterations + 1);
        std::printf("The matching step represents the %.4f%% of the total time with %.4f ms\n\n", res.seconds_nn * 100.0 / seconds,
                    1000.0 * res.seconds_nn);
        std::printf("The minimization step represents the %.4f%% of the total time with %.4f ms\n\n", res.seconds_host * 100.0 / seconds,
                    1000.0 * res.seconds_host);
        std::printf("The transformation step represents the %.4f%% of the total time with %.4f ms\n\n", 0.0, 0.0);
        std::printf("The error estimation step represents the %.4f%% of the total time with %.4f ms\n\n", 0.0, 0.0);
    } else {
        std::printf("ICP converged successfully!\n\n");
        std::printf("Elapsed time: %f ms\n", (float)(1000.0 * res.seconds_total));
    }
    if (a.dump_T) print_transform(res.T);
    icp_destroy(ctx);
    return 0;
}
