// ICP_point_to_plane -- src/ICP_point_to_plane.cu (synthetic 128x128, k = 4 PCA normals, 6x6 normal
// equations, MAX_ITER 50, tol 1e-6) and with --bunny / --hall the dataset variants
// src/CUDA/GPU_point_to_plane_bunny.cu / GPU_point_to_plane_real.cu.  stdout follows :381,:427,:513,:623,:636-641.
#include "common.h"

int main(int argc, char** argv)
{
    Args a;
    if (!parse_args(argc, argv, a, "ICP_point_to_plane")) return 2;
    icp_ctx* ctx = nullptr;
    ICP_CHECK(icp_create(0, &ctx));
    std::vector<float> D, M;
    double conv_ms = 0.0;
    const int n = build_clouds_f32(ctx, a, 128, D, M, &conv_ms);
    if (n < 0) { std::fprintf(stderr, "input: %s (%s)\n", icp_strerror(n), icp_last_error()); return -1; }
    const bool dataset = !a.hall_packets.empty() || !a.bunny.empty();
    const int max_iter = a.max_iter > 0 ? a.max_iter : (dataset ? 100 : 50);
    if (!a.hall_packets.empty()) std::printf("Conversion kernel's elapsed time: %.3f ms\n", conv_ms);

    ICP_CHECK(icp_set_model(ctx, M.data(), n, ICP_F32));
    std::printf("For normals:\nGrid Size: %d, Block Size: %d\n", (n + 255) / 256, 256);
    const auto t0 = std::chrono::steady_clock::now();
    ICP_CHECK(icp_estimate_normals(ctx, nullptr, nullptr));
    const double normals_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("\nNormals were calculated in %f ms\n\n", normals_ms);

    ICP_CHECK(icp_set_moving(ctx, D.data(), n, ICP_F32));
    ICP_CHECK(icp_nn_match_resident(ctx, nullptr));
    int blocks = 0, threads = 0;
    ICP_CHECK(icp_nn_launch_info(ctx, nullptr, &blocks, &threads, nullptr, nullptr));
    std::printf("For ICP loop:\nGrid Size: %d, Block Size: %d\n", blocks, threads);

    icp_params prm{max_iter, 0.000001, 0, ICP_F32, ICP_POINT_TO_PLANE};
    ICP_CHECK(icp_loop_begin(ctx, &prm));
    std::vector<double> err((size_t)max_iter + 1, 0.0);
    double T[16];
    int done = 0, iterations = 0, passes = 0, printed = 0;
    const auto l0 = std::chrono::steady_clock::now();
    while (!done) {
        ICP_CHECK(icp_loop_enqueue(ctx));
        ICP_CHECK(icp_loop_complete(ctx, &done));
        ICP_CHECK(icp_loop_state(ctx, &iterations, &passes, err.data(), (int)err.size(), T));
        for (; printed < passes; ++printed) std::printf("Current error (%d): %.4f\n", printed + 1, (float)err[printed + 1]);
    }
    const double loop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - l0).count();
    std::printf("Error:\n");
    print_sarray(err.data(), iterations + 1);
    std::printf("ICP converged successfully!\n\n");
    std::printf("Elapsed time: %f ms\n", (float)loop_ms);
    if (a.dump_T) print_transform(T);
    icp_destroy(ctx);
    return 0;
}
