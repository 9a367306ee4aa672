// ICP_time_complexity -- regenerates the reference's sweep CSVs on MI355X:
//   default      GPU_ICP_point_to_point_TimeComp.csv   one full ICP iteration, WIDTH = 3..128   (src/CUDA/GPU_time_complexity_point.cu:108-110,451)
//   --plane      GPU_ICP_point_to_plane_TimeComp.csv   same, point-to-plane                     (src/CUDA/GPU_time_complexity_plane.cu)
//   --matching   Matching_mi355x.csv                   the matching kernel alone, min of 10     (src/CUDA/Matching_opt.cu:200-229)
// Same generator and constants as the reference sweeps (synthetic z = x^2 - y^2 grid, MAX_ITER 1), "NUM_POINTS,TIME"
// rows in milliseconds.  Each row is the minimum of 10 runs (the reference's CPU sweeps and Matching_*.csv use that).
#include "common.h"

int main(int argc, char** argv)
{
    bool plane = false, matching = false;
    std::string out;
    int wmax = 128;
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i];
        if (s == "--plane") plane = true;
        else if (s == "--matching") matching = true;
        else if (s == "--out" && i + 1 < argc) out = argv[++i];
        else if (s == "--max-width" && i + 1 < argc) wmax = std::atoi(argv[++i]);
        else { std::fprintf(stderr, "usage: ICP_time_complexity [--plane | --matching] [--out file.csv] [--max-width W]\n"); return 2; }
    }
    if (out.empty()) out = matching ? "Matching_mi355x.csv" : plane ? "GPU_ICP_point_to_plane_TimeComp.csv" : "GPU_ICP_point_to_point_TimeComp.csv";
    FILE* doc = std::fopen(out.c_str(), "w");
    if (!doc) { std::perror("File opening failed"); return -1; }
    std::fprintf(doc, matching ? "#POINTS,TIME\n" : "NUM_POINTS,TIME\n");

    icp_ctx* ctx = nullptr;
    ICP_CHECK(icp_create(0, &ctx));
    const float ang[3] = {0.2f, -0.2f, 0.05f}, t[3] = {0.8f, -0.3f, 0.2f};
    for (int W = 3; W <= wmax; ++W) {
        const int n = W * W;
        std::vector<float> D(3 * (size_t)n), M(3 * (size_t)n);
        ICP_CHECK(icp_synthetic_grid_f32(W, -2.0f, 2.0f, D.data()));
        ICP_CHECK(icp_make_model_f32(D.data(), n, ang, t, M.data()));
        double best_ms = 1e30;
        if (matching) {
            ICP_CHECK(icp_set_model(ctx, M.data(), n, ICP_F32));
            ICP_CHECK(icp_set_moving(ctx, D.data(), n, ICP_F32));
            for (int r = 0; r < 10; ++r) {
                float ms = 0.f;
                ICP_CHECK(icp_nn_match_resident(ctx, &ms));
                if (ms < best_ms) best_ms = ms;
            }
            std::fprintf(doc, "%d,%f\n", n, best_ms);
        } else {
            if (plane && n < 5) continue;  // k = 4 neighbours + self
            icp_params prm{1, 0.000001, 0, ICP_F32, plane ? ICP_POINT_TO_PLANE : ICP_POINT_TO_POINT};
            ICP_CHECK(icp_set_model(ctx, M.data(), n, ICP_F32));
            if (plane) ICP_CHECK(icp_estimate_normals(ctx, nullptr, nullptr));   // outside the timed loop, as in the reference
            for (int r = 0; r < 10; ++r) {
                ICP_CHECK(icp_set_moving(ctx, D.data(), n, ICP_F32));
                ICP_CHECK(icp_loop_begin(ctx, &prm));
                const auto t0 = std::chrono::steady_clock::now();
                int done = 0;
                while (!done) {
                    const int rc = icp_loop_run(ctx, 1 << 20, nullptr, &done);   // the library's own loop (resident kernel where it fits)
                    if (rc == ICP_ERR_SINGULAR) break;   // tiny planar grids: the reference's potrf would fail too
                    ICP_CHECK(rc);
                }
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (ms < best_ms) best_ms = ms;
            }
            std::fprintf(doc, "%d,%.4f\n", n, best_ms);
        }
        std::printf("%d\t%f\n", n, best_ms);
    }
    std::fclose(doc);
    icp_destroy(ctx);
    return 0;
}
