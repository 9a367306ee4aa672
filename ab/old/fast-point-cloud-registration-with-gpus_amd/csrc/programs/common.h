// common.h -- shared by the three reference-named executables.  They are thin: inputs are built or
// read through the C ABI, the loop runs in libicp_mi355x.so, and stdout keeps the reference's format.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "icp_mi355x.h"

#define ICP_CHECK(call)                                                                           \
    do {                                                                                          \
        const int rc_ = (call);                                                                   \
        if (rc_ < 0) {                                                                            \
            std::fprintf(stderr, "%s failed: %s (%s)\n", #call, icp_strerror(rc_), icp_last_error()); \
            return -1;                                                                            \
        }                                                                                         \
    } while (0)

// printSarray of the reference (src/ICP_point_to_point.cu:500-508): "%d: %.4f\n" per entry, blank line
inline void print_sarray(const double* a, int count)
{
    for (int i = 0; i < count; ++i) std::printf("%d: %.4f\n", i + 1, (float)a[i]);
    std::printf("\n");
}

struct Args {
    std::string bunny, hall_packets, hall_intrinsics;
    int width = 0;
    int max_iter = 0;
    bool f64 = false;
    bool dump_T = false;
    std::string trace;  // --trace file: per-iteration clouds in the layout of print_all (src/ICP_CPU.c:409-448)
};

inline bool parse_args(int argc, char** argv, Args& a, const char* prog)
{
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i];
        if (s == "--bunny" && i + 1 < argc) a.bunny = argv[++i];
        else if (s == "--hall" && i + 2 < argc) { a.hall_packets = argv[++i]; a.hall_intrinsics = argv[++i]; }
        else if (s == "--width" && i + 1 < argc) a.width = std::atoi(argv[++i]);
        else if (s == "--max-iter" && i + 1 < argc) a.max_iter = std::atoi(argv[++i]);
        else if (s == "--f64") a.f64 = true;
        else if (s == "--transform") a.dump_T = true;
        else if (s == "--trace" && i + 1 < argc) a.trace = argv[++i];
        else {
            std::fprintf(stderr,
                         "usage: %s [--width W] [--bunny file.csv] [--hall packets.csv beam_intrinsics.csv]\n"
                         "          [--max-iter K] [--transform] [--trace file]\n"
                         "  no arguments = the reference program's built-in synthetic input\n", prog);
            return false;
        }
    }
    return true;
}

inline void print_transform(const double* T)
{
    std::printf("Transform (row-major 4x4, moving -> model):\n");
    for (int r = 0; r < 4; ++r) std::printf("% .9f % .9f % .9f % .9f\n", T[4 * r], T[4 * r + 1], T[4 * r + 2], T[4 * r + 3]);
}

// print_all of the reference (src/ICP_CPU.c:409-448, never called there): one row per point, '|'-separated,
// data | model | the transformed data cloud after every iteration, then the error series.  Clouds are AoS here.
inline bool write_trace(const std::string& path, const std::vector<float>& D, const std::vector<float>& M,
                        const std::vector<std::vector<float>>& pt_total, const double* E, int num_points)
{
    FILE* document = std::fopen(path.c_str(), "w");
    if (!document) return false;
    const int num_iterations = (int)pt_total.size();
    std::fprintf(document, "x_data|y_data|z_data|x_model|y_model|z_model");
    for (int i = 0; i < num_iterations; i++) std::fprintf(document, "|TDx_%d|TDy_%d|TDz_%d", i + 1, i + 1, i + 1);
    std::fprintf(document, "\n");
    for (int i = 0; i < num_points; i++) {
        for (int k = 0; k < 3; k++) std::fprintf(document, "%- 7.3f| ", D[3 * (size_t)i + k]);
        for (int k = 0; k < 3; k++) std::fprintf(document, "%- 7.3f| ", M[3 * (size_t)i + k]);
        for (int j = 0; j < num_iterations; j++)
            for (int k = 0; k < 3; k++) std::fprintf(document, "%- 7.3f| ", pt_total[j][3 * (size_t)i + k]);
        std::fprintf(document, "\n");
    }
    std::fprintf(document, "\nError|");
    for (int i = 0; i < num_iterations; i++) std::fprintf(document, "%.3f|", E[i]);
    std::fprintf(document, "\n");
    std::fclose(document);
    return true;
}

// Build the (data, model) pair the way the reference programs do for each input kind (fp32 AoS).
// Returns the point count or a negative error code.
inline int build_clouds_f32(icp_ctx* ctx, const Args& a, int default_width, std::vector<float>& D, std::vector<float>& M,
                            double* conversion_ms)
{
    if (!a.hall_packets.empty()) {
        // src/CUDA/GPU_point_to_point_real.cu:432-623 + :169-171
        uint32_t enc = 0;
        int n = icp_read_os1_ranges(a.hall_packets.c_str(), nullptr, 0, &enc);
        if (n < 0) return n;
        std::vector<uint32_t> r((size_t)n);
        n = icp_read_os1_ranges(a.hall_packets.c_str(), r.data(), n, &enc);
        if (n < 0) return n;
        float alt[16], az[16];
        int rc = icp_read_os1_intrinsics(a.hall_intrinsics.c_str(), alt, az);
        if (rc < 0) return rc;
        D.resize(3 * (size_t)n);
        M.resize(3 * (size_t)n);
        const auto t0 = std::chrono::steady_clock::now();
        rc = icp_os1_to_cartesian(ctx, r.data(), n, enc, alt, az, D.data());
        if (rc < 0) return rc;
        if (conversion_ms) *conversion_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const float ang[3] = {0.01f, -0.003f, 0.05f}, t[3] = {0.001f, -0.0202f, 0.02f};
        rc = icp_make_model_f32(D.data(), n, ang, t, M.data());
        if (rc < 0) return rc;
        const float s = (float)(1.0 / 1000.0);  // millimetres -> metres
        for (auto& v : D) v *= s;
        for (auto& v : M) v *= s;
        return n;
    }
    if (!a.bunny.empty()) {
        // src/CUDA/GPU_point_to_point_bunny.cu:114-160
        int n = icp_read_xyz_text(a.bunny.c_str(), nullptr, 0);
        if (n < 0) return n;
        D.resize(3 * (size_t)n);
        M.resize(3 * (size_t)n);
        n = icp_read_xyz_text(a.bunny.c_str(), D.data(), n);
        if (n < 0) return n;
        const float ang[3] = {0.15f, -0.1f, 0.05f}, t[3] = {0.01f, -0.04f, 0.02f};
        const int rc = icp_make_model_f32(D.data(), n, ang, t, M.data());
        return rc < 0 ? rc : n;
    }
    const int W = a.width > 0 ? a.width : default_width;
    const int n = W * W;
    D.resize(3 * (size_t)n);
    M.resize(3 * (size_t)n);
    int rc = icp_synthetic_grid_f32(W, -2.0f, 2.0f, D.data());
    if (rc < 0) return rc;
    const float ang[3] = {0.2f, -0.2f, 0.05f}, t[3] = {0.8f, -0.3f, 0.2f};  // src/ICP_point_to_point.cu:157-165
    rc = icp_make_model_f32(D.data(), n, ang, t, M.data());
    return rc < 0 ? rc : n;
}
