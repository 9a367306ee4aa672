// icp_standard -- the README / CMake entry point of the reference (README.md:12, CMakeLists.txt:26-28,
// src/ICP_standard.cu): point-to-point ICP on the 32x32 synthetic grid against the model built with
// the hard-coded rotation of src/ICP_standard.cu:247-249, a FIXED 40 passes (no early exit, :369),
// stdout in the reference's format (:358, :472-475).
#include "common.h"

int main(int argc, char** argv)
{
    Args a;
    if (!parse_args(argc, argv, a, "icp_standard")) return 2;
    const int W = a.width > 0 ? a.width : 32;
    const int n = W * W;
    const int max_iter = a.max_iter > 0 ? a.max_iter : 40;
    std::vector<float> D(3 * (size_t)n), M(3 * (size_t)n);
    ICP_CHECK(icp_synthetic_grid_f32(W, -2.0f, 2.0f, D.data()));
    ICP_CHECK(icp_make_model_standard_f32(D.data(), n, M.data()));

    icp_ctx* ctx = nullptr;
    ICP_CHECK(icp_create(0, &ctx));
    icp_params prm{max_iter, 0.0, /*fixed_iterations=*/1, ICP_F32, ICP_POINT_TO_POINT};
    std::vector<double> err((size_t)max_iter + 1, 0.0);
    icp_result res{};
    res.err = err.data();
    // warm-up pass so that the geometry line below reports the launch actually used
    ICP_CHECK(icp_set_model(ctx, M.data(), n, ICP_F32));
    ICP_CHECK(icp_set_moving(ctx, D.data(), n, ICP_F32));
    ICP_CHECK(icp_nn_match_resident(ctx, nullptr));
    int blocks = 0, threads = 0;
    ICP_CHECK(icp_nn_launch_info(ctx, nullptr, &blocks, &threads, nullptr, nullptr));
    std::printf("Grid Size: %d, Block Size: %d\n", blocks, threads);

    ICP_CHECK(icp_point_to_point(ctx, D.data(), n, M.data(), n, &prm, &res));
    std::printf("Error:\n");
    print_sarray(err.data() + 1, max_iter);  // the reference stores the error of pass k at h_error[k]
    std::printf("Elapsed time: %f ms\n", (float)(1000.0 * res.seconds_total));
    if (a.dump_T) print_transform(res.T);
    icp_destroy(ctx);
    return 0;
}
