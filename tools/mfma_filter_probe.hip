// mfma_filter_probe.hip -- bounded experiment: a half-precision matrix-core FILTER in front of the exact
// arithmetic of the hierarchical search (configs[4]).  NOT part of the library: a stand-alone probe.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o bin/mfma_filter_probe tools/mfma_filter_probe.hip && bin/mfma_filter_probe
//
// What is probed.  Three quarters of the listed chunks of a pass die at the per-point BOX test (icp_device.h, box_may_improve: 19
// vector instructions per (wave, chunk), a lower bound of every distance between the wave's 128 points and the chunk's box).  A
// matrix core can produce all 32 x 32 squared distances between 32 points and 4 chunks in ONE v_mfma_f32_32x32x16_f16 -- not the
// reference's roundings (no MFMA can: the subtraction is per pair), but a filter only has to be CONSERVATIVE: no pair that could win
// may be rejected; whatever passes goes through the exact arithmetic as before (src/CUDA/GPU_point_to_point_real.cu:60-75 decides).
//
// The arithmetic.  Per row: centre c, scale s (the extent of the row's box + the batch's chunk boxes), p' = (p - c) / s,
// q' = (q - c) / s in [-1, 1].  d^2 < bound  <=>  F = |q'|^2 - 2 p'.q' - tau_p < 0,  tau_p = bound_p / s^2 - |p'|^2.  Every fp32 value v
// goes in as TWO halves, vh = fp16(v), vl = fp16(v - vh) (|v - vh - vl| <= 2^-22 |v| + 2^-25: the second term is the subnormal floor),
// over the 16 columns of the K dimension:
//     A (point m):   ph.x ph.y ph.z | ph.x ph.y ph.z | pl.x pl.y pl.z |  1   1  | tau_h tau_l | 0 0 0
//     B (model n): -2qh.x ..        | -2ql.x ..      | -2qh.x ..      |  Qh  Ql |  -1    -1   | 0 0 0        (Q = |q'|^2 in fp32)
// so that C[m][n] = F up to: the dropped pl.ql term (|pl| <= 2^-12, |2 ql| <= 2^-11: <= 3 x 2^-23), the halves' truncation (a half of
// a half is 2^-12 of 2^-12 of its value, plus the subnormal floor 2^-25: <= 6 x 2^-23 for the products, 2^-22 each for Q <= 3 and
// |tau| <= 4), and the fp32 accumulation of 13 products whose partial sums stay below 16 (<= 13 x 2^-20 = 1.24e-5 if every addition
// rounds; the matrix core may well do better).  All of it: |C - F| < 1.4e-5; EPS = 2^-15 (3.05e-5) leaves a factor two.
// The filter passes a pair when C < EPS: a true winner (F < 0) always passes.  In row units EPS x s^2 -- with s ~ 0.06 in a late pass
// (a 0.03 row and its neighbourhood) 1.1e-7 against bounds of 2.5e-5: the filter is tight to 0.4 % of the bound.
//
// What the probe measures (fuzz + counts + time), on synthetic late-pass geometry (a row of 128 points on z = x^2 - y^2 at the model's
// point spacing, chunks of 8 consecutive model points along a Hilbert-like raster around it, bounds = (nearest distance)^2 x 1.1):
//   1. conservativeness: over all (row, batch) pairs, no (point, model point) with exact d^2 < bound is rejected;
//   2. selectivity: chunks that pass (a) the group-box test only, (b) the per-point box test, (c) the MFMA pair filter, (d) truly hold a
//      distance below some point's bound;
//   3. cost: the filter loop (B operand from staged fp32 coordinates: centre, scale, split, pack; 4 MFMAs; the sign reduction) against
//      the box-test loop over the same batches, in wave-instructions (from the ISA) and in time.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr float EPS = 3.0517578125e-5f;   // 2^-15, see above

__device__ __forceinline__ void split(float v, _Float16& h, _Float16& l) { h = (_Float16)v; l = (_Float16)(v - (float)h); }

// one wave = one row of 128 points (two per lane, as nn_match_sparse holds them) against NB batches of 4 chunks (32 model points:
// x[32] y[32] z[32] per batch + the four chunk boxes), staged in LDS first as the kernel's hits are: the timed loops read LDS only.
// Blocks of 4 waves (4 rows), 39 KB of LDS: four blocks to a CU, the occupancy of the kernel's 4-wave form.
// out[batch] = 4-bit mask of the chunks that pass.  MODE 0: the MFMA pair filter; 1: the per-point box test of the library
constexpr int NB = 16;
template <int MODE>
__global__ __launch_bounds__(256, 4) void filter_kernel(const float* __restrict__ P /*[rows][3][128]*/, const float* __restrict__ bound /*[rows][128]*/,
                                                    const float* __restrict__ Qb /*[rows][nb][3][32]*/, const float* __restrict__ boxes /*[rows][nb][4][6]*/,
                                                    const float* __restrict__ cs /*[rows][4]: centre xyz, 1/s*/, int nb, int reps, unsigned int* __restrict__ out)
{
    const int w = threadIdx.x >> 6, row = blockIdx.x * 4 + w, lane = threadIdx.x & 63;
    __shared__ float stage_all[4][NB * 120];   // per wave and batch: x[32] y[32] z[32] | 4 x {lo.xyz hi.xyz}
    __shared__ float sp_all[4][4][128];
    float* stage = stage_all[w];
    for (int i = lane; i < nb * 96; i += 64) stage[(i / 96) * 120 + i % 96] = Qb[(size_t)row * nb * 96 + i];
    for (int i = lane; i < nb * 24; i += 64) stage[(i / 24) * 120 + 96 + i % 24] = boxes[(size_t)row * nb * 24 + i];
    const float* p = P + (size_t)row * 384;
    const float px[2] = {p[lane], p[lane + 64]}, py[2] = {p[128 + lane], p[128 + lane + 64]}, pz[2] = {p[256 + lane], p[256 + lane + 64]};
    const float bd[2] = {bound[(size_t)row * 128 + lane], bound[(size_t)row * 128 + lane + 64]};
    const float cx = cs[row * 4], cy = cs[row * 4 + 1], cz = cs[row * 4 + 2], is = cs[row * 4 + 3];
    float (*sp)[128] = sp_all[w];   // the row's points, centred and scaled, and tau: the A operand's source (points live two per lane, tiles want 32 rows)
    unsigned int acc = 0;
    __syncthreads();
    if constexpr (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float x = (px[t] - cx) * is, y = (py[t] - cy) * is, z = (pz[t] - cz) * is;
            sp[0][lane + 64 * t] = x; sp[1][lane + 64 * t] = y; sp[2][lane + 64 * t] = z;
            sp[3][lane + 64 * t] = bd[t] * is * is - ((x * x + y * y) + z * z);
        }
        __syncthreads();
        // A operand of the four 32-point tiles: lane (r = lane & 31, h = lane >> 5) holds A[row r][k = 8 h + j]
        h8 A[4];
        const int r = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            const int m = tile * 32 + r;
            _Float16 xh, xl, yh, yl, zh, zl, th, tl;
            split(sp[0][m], xh, xl); split(sp[1][m], yh, yl); split(sp[2][m], zh, zl); split(sp[3][m], th, tl);
            // k: 0-2 ph | 3-5 ph | 6-8 pl | 9 10: 1 1 | 11 12: tau_h tau_l | 13-15: 0
            if (hh == 0) A[tile] = h8{xh, yh, zh, xh, yh, zh, xl, yl};
            else A[tile] = h8{zl, (_Float16)1.f, (_Float16)1.f, th, tl, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        }
        for (int rep = 0; rep < reps; ++rep)
        for (int b = 0; b < nb; ++b) {
            const float* q = stage + b * 120;
            // B operand: lane (n = lane & 31, h) holds B[k = 8 h + j][col n] -- centre, scale, split, pack (per batch: the VALU cost of the filter)
            const float qx = (q[r] - cx) * is, qy = (q[32 + r] - cy) * is, qz = (q[64 + r] - cz) * is;
            const float Q = (qx * qx + qy * qy) + qz * qz;
            _Float16 xh, xl, yh, yl, zh, zl, Qh, Ql;
            split(-2.f * qx, xh, xl); split(-2.f * qy, yh, yl); split(-2.f * qz, zh, zl); split(Q, Qh, Ql);
            h8 B;
            if (hh == 0) B = h8{xh, yh, zh, xl, yl, zl, xh, yh};
            else B = h8{zh, Qh, Ql, (_Float16)-1.f, (_Float16)-1.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            unsigned int pass = 0;   // bit c: some point of the row may reach chunk c
#pragma unroll
            for (int tile = 0; tile < 4; ++tile) {
                f16v C = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[tile], B, C, 0, 0, 0);
                // C: col = lane & 31 (the model point), 16 rows (points) per lane: the minimum over them, then "below EPS" per lane
                float mn = fminf(fminf(C[0], C[1]), C[2]);
#pragma unroll
                for (int i = 3; i < 15; i += 2) mn = fminf(fminf(mn, C[i]), C[i + 1]);
                mn = fminf(mn, C[15]);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(mn < EPS);
                const unsigned int both = (unsigned int)m | (unsigned int)(m >> 32);   // rows 0-3.. of lanes 0-31 and rows 4-7.. of lanes 32-63: the same 32 columns
                pass |= ((both & 0xffu) ? 1u : 0u) | ((both & 0xff00u) ? 2u : 0u) | ((both & 0xff0000u) ? 4u : 0u) | ((both & 0xff000000u) ? 8u : 0u);
            }
            acc += pass;
            if (rep == 0 && lane == 0) out[(size_t)row * nb + b] = pass;
        }
    } else {
        for (int rep = 0; rep < reps; ++rep)
        for (int b = 0; b < nb; ++b) {
            unsigned int pass = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* bx = stage + b * 120 + 96 + c * 6;   // (wave-uniform address: an LDS broadcast read, as in the kernel)
                bool need = false;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float gx = px[t] - __builtin_amdgcn_fmed3f(px[t], bx[0], bx[3]);
                    const float gy = py[t] - __builtin_amdgcn_fmed3f(py[t], bx[1], bx[4]);
                    const float gz = pz[t] - __builtin_amdgcn_fmed3f(pz[t], bx[2], bx[5]);
                    const float L = ((gx * gx + gy * gy) + gz * gz) * 0.99999905f;
                    need |= L <= bd[t];
                }
                if (__builtin_amdgcn_ballot_w64(need) != 0ull) pass |= 1u << c;
            }
            acc += pass;
            if (rep == 0 && lane == 0) out[(size_t)row * nb + b] = pass;
        }
    }
    if (acc == 0xffffffffu) out[0] = acc;   // (keeps the repeats alive)
}

static float surf(float x, float y) { return x * x - y * y; }

int main()
{
    // a 4 x 4 surface at the spacing of the 10 M-point model (3163 x 3163): a row = 128 consecutive points of a 12 x 11 patch; the moving
    // cloud is the model shifted along the surface by `shift` (a late pass: 0.005) -- every point's bound = its exact nearest distance^2 x 1.1
    const int rows = 4096, nb = NB;
    const float h = 4.0f / 3162.0f;
    std::vector<float> P((size_t)rows * 384), B((size_t)rows * 128), Q((size_t)rows * nb * 96), BX((size_t)rows * nb * 24), CS((size_t)rows * 4);
    std::vector<unsigned char> truth((size_t)rows * nb);   // bit c: chunk c really holds a model point below some point's bound
    std::vector<unsigned char> group((size_t)rows * nb);   // ... passes the group-box test (what lists it)
    srand(5);
    auto rnd = []() { return (float)rand() / (float)RAND_MAX; };
    long long rejected_winners = 0;
    for (int r = 0; r < rows; ++r) {
        const float x0 = -1.8f + 3.6f * rnd(), y0 = -1.8f + 3.6f * rnd(), shift = 0.002f + 0.006f * rnd();
        // model patch: 40 x 24 points around the row, chunks = 8 consecutive points of a raster line (what a Hilbert range looks like locally)
        std::vector<float> mx, my, mz;
        for (int j = -6; j < 18; ++j)
            for (int i = -14; i < 26; ++i) { const float x = x0 + i * h, y = y0 + j * h; mx.push_back(x); my.push_back(y); mz.push_back(surf(x, y)); }
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        float* p = &P[(size_t)r * 384];
        for (int k = 0; k < 128; ++k) {
            const int i = k % 12, j = k / 12;
            const float x = x0 + i * h + shift * 0.8f + 0.3f * h * rnd(), y = y0 + j * h + shift * 0.6f + 0.3f * h * rnd(), z = surf(x, y) + 0.2f * h * (rnd() - 0.5f);
            p[k] = x; p[128 + k] = y; p[256 + k] = z;
            float best = 1e30f;
            for (size_t q = 0; q < mx.size(); ++q) { const float dx = mx[q] - x, dy = my[q] - y, dz = mz[q] - z; best = fminf(best, (dx * dx + dy * dy) + dz * dz); }
            B[(size_t)r * 128 + k] = best * 1.1f;
            const float c3[3] = {x, y, z};
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], c3[a]); hi[a] = fmaxf(hi[a], c3[a]); }
        }
        // the batches: the 4 x NB chunks nearest to the row's centre
        const int chunks = (int)mx.size() / 8;
        std::vector<std::pair<float, int>> order;
        for (int c = 0; c < chunks; ++c) {
            const float cx = 0.5f * (mx[c * 8] + mx[c * 8 + 7]) - 0.5f * (lo[0] + hi[0]), cy = my[c * 8] - 0.5f * (lo[1] + hi[1]);
            order.push_back({cx * cx + cy * cy, c});
        }
        std::sort(order.begin(), order.end());
        float ext = 0.f;
        for (int a = 0; a < 3; ++a) ext = fmaxf(ext, hi[a] - lo[a]);
        float reach = 0.f;
        for (int b = 0; b < nb; ++b)
            for (int c = 0; c < 4; ++c) {
                const int ch = order[b * 4 + c].second;
                float bl[3] = {1e30f, 1e30f, 1e30f}, bh[3] = {-1e30f, -1e30f, -1e30f};
                for (int k = 0; k < 8; ++k) {
                    const float q3[3] = {mx[ch * 8 + k], my[ch * 8 + k], mz[ch * 8 + k]};
                    Q[((size_t)r * nb + b) * 96 + c * 8 + k] = q3[0]; Q[((size_t)r * nb + b) * 96 + 32 + c * 8 + k] = q3[1]; Q[((size_t)r * nb + b) * 96 + 64 + c * 8 + k] = q3[2];
                    for (int a = 0; a < 3; ++a) { bl[a] = fminf(bl[a], q3[a]); bh[a] = fmaxf(bh[a], q3[a]); reach = fmaxf(reach, fmaxf(fabsf(q3[a] - lo[a]), fabsf(q3[a] - hi[a]))); }
                }
                for (int a = 0; a < 3; ++a) { BX[(((size_t)r * nb + b) * 4 + c) * 6 + a] = bl[a]; BX[(((size_t)r * nb + b) * 4 + c) * 6 + 3 + a] = bh[a]; }
                bool any = false, grp = false;
                float Bmax = 0.f;
                for (int k = 0; k < 128; ++k) {
                    Bmax = fmaxf(Bmax, B[(size_t)r * 128 + k]);
                    for (int kk = 0; kk < 8; ++kk) {
                        const float dx = mx[ch * 8 + kk] - p[k], dy = my[ch * 8 + kk] - p[128 + k], dz = mz[ch * 8 + kk] - p[256 + k];
                        any |= ((dx * dx + dy * dy) + dz * dz) < B[(size_t)r * 128 + k];
                    }
                }
                float g2 = 0.f;
                for (int a = 0; a < 3; ++a) { const float g = fmaxf(fmaxf(bl[a] - hi[a], lo[a] - bh[a]), 0.f); g2 += g * g; }
                grp = g2 < Bmax;
                if (any) truth[(size_t)r * nb + b] |= 1u << c;
                if (grp) group[(size_t)r * nb + b] |= 1u << c;
            }
        CS[r * 4] = 0.5f * (lo[0] + hi[0]); CS[r * 4 + 1] = 0.5f * (lo[1] + hi[1]); CS[r * 4 + 2] = 0.5f * (lo[2] + hi[2]);
        CS[r * 4 + 3] = 1.0f / fmaxf(reach, ext);   // everything the tiles see lies within [-1, 1] of the centre
    }
    float *dP, *dB, *dQ, *dBX, *dCS;
    unsigned int* dO;
    CHECK(hipMalloc(&dP, P.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&dQ, Q.size() * 4)); CHECK(hipMalloc(&dBX, BX.size() * 4));
    CHECK(hipMalloc(&dCS, CS.size() * 4)); CHECK(hipMalloc(&dO, (size_t)rows * nb * 4));
    CHECK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dBX, BX.data(), BX.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dCS, CS.data(), CS.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned int> out[2];
    float ms[2];
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 200;
    for (int mode = 0; mode < 2; ++mode) {
        for (int warm = 0; warm < 2; ++warm) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(filter_kernel<0>, dim3(rows / 4), dim3(256), 0, 0, dP, dB, dQ, dBX, dCS, nb, reps, dO);
            else hipLaunchKernelGGL(filter_kernel<1>, dim3(rows / 4), dim3(256), 0, 0, dP, dB, dQ, dBX, dCS, nb, reps, dO);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
        out[mode].resize((size_t)rows * nb);
        CHECK(hipMemcpy(out[mode].data(), dO, out[mode].size() * 4, hipMemcpyDeviceToHost));
    }
    long long n_list = 0, n_box = 0, n_mfma = 0, n_true = 0, n_all = 0, box_missed = 0;
    for (size_t k = 0; k < truth.size(); ++k)
        for (int c = 0; c < 4; ++c) {
            const bool t = truth[k] >> c & 1, g = group[k] >> c & 1, bx = out[1][k] >> c & 1, mf = out[0][k] >> c & 1;
            ++n_all;
            n_true += t; n_list += g; n_box += bx && g; n_mfma += mf && g;
            if (t && !mf) ++rejected_winners;
            if (t && !bx) ++box_missed;
        }
    std::printf("rows %d x %d batches of 4 chunks: %lld (row, chunk) pairs\n", rows, nb, n_all);
    std::printf("  listed by the group-box test            %8lld (%.1f %%)\n", n_list, 100.0 * n_list / n_all);
    std::printf("  ... through the per-point box test      %8lld (%.1f %% of the listed)\n", n_box, 100.0 * n_box / n_list);
    std::printf("  ... through the MFMA pair filter        %8lld (%.1f %% of the listed)\n", n_mfma, 100.0 * n_mfma / n_list);
    std::printf("  ... that hold a distance below a bound  %8lld (%.1f %% of the listed)\n", n_true, 100.0 * n_true / n_list);
    std::printf("  true winners rejected by the MFMA filter: %lld (must be 0);  by the box test: %lld (must be 0)\n", rejected_winners, box_missed);
    const double per_batch_ns[2] = {1e6 * ms[0] / ((double)reps * nb), 1e6 * ms[1] / ((double)reps * nb)};
    // (1024 blocks of 4 waves on 256 CUs: four blocks -- sixteen waves -- per CU, all of them in the loop at once; the time of the launch x 256 CUs x 4 SIMDs
    // over the (row, batch) pairs it worked through = SIMD time per pair)
    const double pairs = (double)reps * nb * rows;
    std::printf("SIMD time per (row, batch of 4 chunks) at 4 waves per SIMD, operands in LDS: MFMA filter %.1f ns, box tests %.1f ns (ratio %.2f)\n",
                1e6 * ms[0] * 1024.0 / pairs, 1e6 * ms[1] * 1024.0 / pairs, ms[0] / ms[1]);
    (void)per_batch_ns;
    return rejected_winners == 0 && box_missed == 0 ? 0 : 1;
}
