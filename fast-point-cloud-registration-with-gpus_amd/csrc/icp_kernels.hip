// icp_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ICP hot path.
//
// Kernel inventory (reference statement each one replaces; file:line relative to the reference):
//   nn_match_kernel        Matching<<<>>>  src/CUDA/GPU_point_to_point_real.cu:38-79 (fp32),
//                          MKL matching loop src/ICP_CPU.c:220-234 (fp64)
//   moments_kernel         Q_index + 2x cublasSgemv + deviation + cublasSgemm
//                          src/ICP_point_to_point.cu:308-357; Cxb + 2x Sgemv
//                          src/CUDA/GPU_point_to_plane_real.cu:246-288,532-549
//   transform_error_kernel RyT + Scopy + Scopy/Saxpy/Snrm2  src/ICP_point_to_point.cu:81-88,403-416
//   finalize_kernel        (the reductions hidden inside cuBLAS)
//   knn4_kernel/normals    knn + Normals + host ssyev loop  src/CUDA/GPU_point_to_plane_real.cu:54-188,413-423
//   os1_conversion_kernel  Conversion     src/CUDA/GPU_point_to_point_real.cu:20-36
//
// Numerics contract of the matching kernels: the squared distance is evaluated exactly as the CPU
// path does -- dx = q-p; dx*dx; (dx2+dy2)+dz2, every operation rounded separately -- so the file is
// compiled with FP contraction OFF and tests/test_build.py greps the ISA of nn_match_kernel for
// fma/mad/fmac.  Ties resolve to the lowest model index.
#include "icp_kernels.h"

#include <math.h>
#include <stdlib.h>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <rocprim/device/device_radix_sort.hpp>

#pragma clang fp contract(off)

namespace icp {

size_t elem_size(int precision) { return precision == ICP_F64 ? sizeof(double) : sizeof(float); }

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
template <typename F> struct Vec16;  // 16-byte vector of F
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };

__device__ __forceinline__ float vget(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ double vget(const double2& v, int i) { return i == 0 ? v.x : v.y; }

__device__ __forceinline__ float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double fmin_(double a, double b) { return __builtin_fmin(a, b); }

template <typename F> __device__ __forceinline__ F inf_();
template <> __device__ __forceinline__ float inf_<float>() { return __builtin_huge_valf(); }
template <> __device__ __forceinline__ double inf_<double>() { return __builtin_huge_val(); }

// (dx*dx + dy*dy) + dz*dz, each op rounded on its own (contraction is off for this TU)
template <typename F>
__device__ __forceinline__ F dist2(F px, F py, F pz, F qx, F qy, F qz)
{
    F dx = qx - px;
    F dy = qy - py;
    F dz = qz - pz;
    dx = dx * dx;
    dy = dy * dy;
    dz = dz * dz;
    F d = dx + dy;
    return d + dz;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-wide sum of NACC per-thread doubles -> out[0..NACC) (written by threads 0..NACC-1).
// Fixed combination order => bitwise reproducible for a fixed launch geometry.
template <int NACC, int BLOCK>
__device__ __forceinline__ void block_sum_store(const double (&acc)[NACC], double* out)
{
    constexpr int NW = BLOCK / 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if constexpr (NW == 1 && NACC > 2) {
        // single wave, many slots: transpose through LDS (rows padded to 65 doubles: lane k reads bank 2k)
        // and let lane k add its slot's 64 entries in lane order -- far fewer cross-lane ops than NACC butterflies
        __shared__ double tr[NACC][65];
#pragma unroll
        for (int k = 0; k < NACC; ++k) tr[k][lane] = acc[k];
        __syncthreads();
        if (lane < NACC) {
            double s = 0.0;
#pragma unroll 8
            for (int l = 0; l < 64; ++l) s += tr[lane][l];
            out[lane] = s;
        }
        return;
    }
    __shared__ double red[NW][NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) red[w][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        double s = red[0][threadIdx.x];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) s += red[ww][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// layout conversion
// ------------------------------------------------------------------------------------------------
// `nonfinite` (pinned host memory, or NULL): counts the points with a NaN or an infinite coordinate -- icp_set_* refuse such a
// cloud (include/icp_mi355x.h).  Written only when there is something to count.
template <typename F>
__global__ void aos_to_soa_kernel(const F* __restrict__ aos, int n, int n_pad, F* __restrict__ soa, unsigned int* nonfinite)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    const int s = i < n ? i : n - 1;  // padding replicates the last real point
    const F x = aos[3 * (size_t)s + 0], y = aos[3 * (size_t)s + 1], z = aos[3 * (size_t)s + 2];
    soa[i] = x;
    soa[(size_t)n_pad + i] = y;
    soa[2 * (size_t)n_pad + i] = z;
    // (x - x is 0 for every finite x, NaN for NaN and for +-inf)
    if (nonfinite != nullptr && i < n && !((x - x) == F(0) && (y - y) == F(0) && (z - z) == F(0)))
        __hip_atomic_fetch_add(nonfinite, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename F>
__global__ void soa_to_aos_kernel(const F* __restrict__ soa, int n, int n_pad, F* __restrict__ aos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    aos[3 * (size_t)i + 0] = soa[i];
    aos[3 * (size_t)i + 1] = soa[(size_t)n_pad + i];
    aos[3 * (size_t)i + 2] = soa[2 * (size_t)n_pad + i];
}

// ------------------------------------------------------------------------------------------------
// matching
//
// grid = (n_pad / (256*T), S).  A block owns 256*T moving points (T per lane, in registers) and one
// segment [q0, q1) of the model.  The segment streams through LDS in SoA tiles; every lane reads the
// SAME LDS address (broadcast, conflict-free), 16 bytes per ds_read.
//
// The inner loop keeps only the running MINIMUM per moving point (8 rounding-exact VALU ops per
// pair + a min), not the arg-min: per NN_CHUNK model points one compare records the id of the
// chunk that last lowered the minimum.  Because the compare is strict, that is the FIRST chunk
// holding the final minimum; the index is recovered afterwards by re-evaluating just that chunk
// (16 pairs per moving point) and taking the lowest j with d_j == min.  Same answer as the
// reference's ascending strict-< scan, ~25% fewer VALU ops per pair.
// ------------------------------------------------------------------------------------------------
template <typename F, int T, int TQ>
__global__ __launch_bounds__(NN_BLOCK) void nn_match_kernel(const F* __restrict__ P, int n_pad,
                                                            const F* __restrict__ Q, int m_pad, int seg_len,
                                                            F* __restrict__ part_d, int32_t* __restrict__ part_idx)
{
    using V = typename Vec16<F>::type;
    constexpr int VN = Vec16<F>::N;
    constexpr int C = NN_CHUNK;
    __shared__ __attribute__((aligned(16))) F sq[3 * TQ];

    const int q0 = blockIdx.y * seg_len;
    const int q1 = min(q0 + seg_len, m_pad);
    const int ibase = blockIdx.x * (NN_BLOCK * T) + threadIdx.x;

    F px[T], py[T], pz[T], best[T];
    int cst[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int i = ibase + t * NN_BLOCK;
        px[t] = P[i];
        py[t] = P[(size_t)n_pad + i];
        pz[t] = P[2 * (size_t)n_pad + i];
        best[t] = inf_<F>();
        cst[t] = q0 / C;
    }

    for (int tile = q0; tile < q1; tile += TQ) {
        const int len = min(TQ, q1 - tile);  // multiple of C
        __syncthreads();
        for (int e = threadIdx.x * VN; e < len; e += NN_BLOCK * VN) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                *reinterpret_cast<V*>(&sq[a * TQ + e]) =
                    *reinterpret_cast<const V*>(&Q[(size_t)a * m_pad + tile + e]);
        }
        __syncthreads();

        for (int c = 0; c < len; c += C) {
            F bo[T];
#pragma unroll
            for (int t = 0; t < T; ++t) bo[t] = best[t];
#pragma unroll
            for (int k = 0; k < C; k += VN) {
                const V qx = *reinterpret_cast<const V*>(&sq[c + k]);
                const V qy = *reinterpret_cast<const V*>(&sq[TQ + c + k]);
                const V qz = *reinterpret_cast<const V*>(&sq[2 * TQ + c + k]);
#pragma unroll
                for (int v = 0; v < VN; ++v) {
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const F d = dist2<F>(px[t], py[t], pz[t], vget(qx, v), vget(qy, v), vget(qz, v));
                        best[t] = fmin_(best[t], d);
                    }
                }
            }
            const int cid = (tile + c) / C;
#pragma unroll
            for (int t = 0; t < T; ++t) cst[t] = (best[t] < bo[t]) ? cid : cst[t];
        }
    }

    // index recovery inside the winning chunk (global memory, L2-resident)
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int base = cst[t] * C;
        int idx = base;
        const F b = best[t];
        for (int k = C - 1; k >= 0; --k) {
            const int j = base + k;
            const F d = dist2<F>(px[t], py[t], pz[t], Q[j], Q[(size_t)m_pad + j], Q[2 * (size_t)m_pad + j]);
            idx = (d == b) ? j : idx;
        }
        const size_t o = (size_t)blockIdx.y * n_pad + ibase + t * NN_BLOCK;
        part_d[o] = b;
        part_idx[o] = idx;
    }
}

// ------------------------------------------------------------------------------------------------
// matching, fp32, v2 -- the shipped fp32 kernel.
//
// What the gfx950 VALU probe (profiles/r1/valu_rate_gfx950.txt) says and how the kernel answers:
//   * one wave issues a VALU instruction only every ~6 cycles whatever its ILP; a SIMD saturates
//     at ~8 resident waves  -> <= 64 VGPRs (launch_bounds(256, 8)), 16 KB LDS per block;
//   * v_pk_add/mul_f32 retire 2 results per issue slot (70 T results/s vs 51 T for plain ops)
//     -> every sub/mul/add of the distance is a packed op over TWO MOVING POINTS of the lane; the
//     model coordinate is broadcast into both halves with op_sel straight from the LDS quad,
//     no v_mov.  Each half is an ordinary IEEE add/mul, so rounding is identical to the scalar form;
//   * v_cndmask (VCC read) costs ~9 issue slots -> the chunk-id update sits behind a wave-uniform
//     branch that is skipped while no lane's minimum moved.
// Small clouds cannot fill 8 waves x 1024 SIMDs along the moving axis, so the model range is split
// twice: grid.y segments (merged later from the partials) and, inside a block, one contiguous
// quarter of the segment per wave (merged through LDS in ascending order, strict <, so the lowest
// index still wins).  All four waves of a block own the SAME 64*T moving points.
// ------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

// (q.lo - p.lo, q.lo - p.hi) / (q.hi - p.lo, q.hi - p.hi): src0 half broadcast by op_sel, src1 negated
template <int HI>
__device__ __forceinline__ f2 pk_sub_bcast(f2 q, f2 p)
{
    f2 r;
    if constexpr (HI == 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(q), "v"(p));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(q), "v"(p));
    return r;
}

template <int HI>
__device__ __forceinline__ f2 pk_dist2(f2 qx, f2 qy, f2 qz, f2 px, f2 py, f2 pz)
{
    f2 dx = pk_sub_bcast<HI>(qx, px);
    f2 dy = pk_sub_bcast<HI>(qy, py);
    f2 dz = pk_sub_bcast<HI>(qz, pz);
    dx = dx * dx;
    dy = dy * dy;
    dz = dz * dz;
    f2 d = dx + dy;
    return d + dz;
}

// One wave passing data to itself through LDS: DS instructions of a wave execute in order, so all that is needed is
// that the COMPILER keeps the order (and does not cache the values in registers).  A workgroup-scope fence would also
// drain the wave's global stores (s_waitcnt vmcnt(0)) -- ~1 us of idle time in the matching kernel's tail.
__device__ __forceinline__ void lds_same_wave_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// a wave-uniform float, moved to a scalar register
__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// An index the compiler must treat as new: keeps it from hoisting the per-lane 64-bit addresses derived from a
// loop-invariant index out of the resident kernel's pass loop (a dozen register pairs held for nothing -- it spilled).
__device__ __forceinline__ int fresh(int i) { asm volatile("" : "+v"(i)); return i; }

// wave-wide min / max of a float by DPP (no LDS): row_shr 1,2,4,8 leave each row's result in its lane 15
// (min/max are idempotent, overlapping windows are harmless), row_bcast15/31 carry it to lane 63.
template <bool MAX>
__device__ __forceinline__ float wave_minmax(float v)
{
    // written as DPP-fused instructions (the compiler would spend seven per step); s_nop 1 covers the
    // VALU-write -> DPP-read hazard, lanes without a source keep their value
#define ICP_DPP_STEP(CTRL)                                                                       \
    if constexpr (MAX) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " CTRL : "+v"(v));   \
    else asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " CTRL : "+v"(v));
    ICP_DPP_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
    ICP_DPP_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
    ICP_DPP_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
    ICP_DPP_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
    ICP_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
    ICP_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef ICP_DPP_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// bounding box of the wave's values: three minima and three maxima reduced together, so that the six dependent DPP
// chains overlap (every DPP reads a register written six instructions earlier: no wait states except the first)
__device__ __forceinline__ void wave_box(float (&lo)[3], float (&hi)[3])
{
#define ICP_BOX_STEP(CTRL)                                                                                      \
    asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " CTRL "\n\tv_min_f32_dpp %1, %1, %1 " CTRL "\n\tv_min_f32_dpp %2, %2, %2 " CTRL \
                 "\n\tv_max_f32_dpp %3, %3, %3 " CTRL "\n\tv_max_f32_dpp %4, %4, %4 " CTRL "\n\tv_max_f32_dpp %5, %5, %5 " CTRL        \
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]));
    ICP_BOX_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
    ICP_BOX_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
    ICP_BOX_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
    ICP_BOX_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
    ICP_BOX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
    ICP_BOX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef ICP_BOX_STEP
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo[a]), 63));
        hi[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hi[a]), 63));
    }
}

constexpr int NN2_TQW = 256;  // model points per wave per LDS tile step

// One chunk of C model points against the lane's packed moving points, with the xy early-out:
// phase A forms pxy = dx*dx + dy*dy (the inner sum of the reference's association) for the whole chunk;
// d = fl(pxy + dz*dz) >= pxy, so a chunk whose smallest pxy is not below any lane's running minimum cannot
// lower it (nor win a tie: ascending order, strict <) and its z half is skipped; phase B finishes the chunk
// exactly as the un-culled kernel would have.
template <int TP, int C>
__device__ __forceinline__ void scan_chunk_xy_cull(const float* qxp, const float* qyp, const float* qzp, const f2 (&px)[TP],
                                                   const f2 (&py)[TP], const f2 (&pz)[TP], float (&best)[2 * TP])
{
    f2 pxy[TP][C];
    float mxy[2 * TP];
#pragma unroll
    for (int t = 0; t < 2 * TP; ++t) mxy[t] = inf_<float>();
#pragma unroll
    for (int kk = 0; kk < C; kk += 4) {
        const float4 qx4 = *reinterpret_cast<const float4*>(qxp + kk);
        const float4 qy4 = *reinterpret_cast<const float4*>(qyp + kk);
        const f2 qxa = f2{qx4.x, qx4.y}, qxb = f2{qx4.z, qx4.w};
        const f2 qya = f2{qy4.x, qy4.y}, qyb = f2{qy4.z, qy4.w};
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            f2 ax, ay;
            ax = pk_sub_bcast<0>(qxa, px[u]); ay = pk_sub_bcast<0>(qya, py[u]);
            pxy[u][kk + 0] = ax * ax + ay * ay;
            ax = pk_sub_bcast<1>(qxa, px[u]); ay = pk_sub_bcast<1>(qya, py[u]);
            pxy[u][kk + 1] = ax * ax + ay * ay;
            ax = pk_sub_bcast<0>(qxb, px[u]); ay = pk_sub_bcast<0>(qyb, py[u]);
            pxy[u][kk + 2] = ax * ax + ay * ay;
            ax = pk_sub_bcast<1>(qxb, px[u]); ay = pk_sub_bcast<1>(qyb, py[u]);
            pxy[u][kk + 3] = ax * ax + ay * ay;
            mxy[2 * u] = fmin_(fmin_(mxy[2 * u], pxy[u][kk].x), pxy[u][kk + 1].x);
            mxy[2 * u] = fmin_(fmin_(mxy[2 * u], pxy[u][kk + 2].x), pxy[u][kk + 3].x);
            mxy[2 * u + 1] = fmin_(fmin_(mxy[2 * u + 1], pxy[u][kk].y), pxy[u][kk + 1].y);
            mxy[2 * u + 1] = fmin_(fmin_(mxy[2 * u + 1], pxy[u][kk + 2].y), pxy[u][kk + 3].y);
        }
    }
    bool need = false;
#pragma unroll
    for (int t = 0; t < 2 * TP; ++t) need |= mxy[t] < best[t];
    if (__builtin_amdgcn_ballot_w64(need) == 0ull) return;  // wave-uniform early-out
#pragma unroll
    for (int kk = 0; kk < C; kk += 4) {
        const float4 qz4 = *reinterpret_cast<const float4*>(qzp + kk);
        const f2 qza = f2{qz4.x, qz4.y}, qzb = f2{qz4.z, qz4.w};
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            f2 az;
            az = pk_sub_bcast<0>(qza, pz[u]); const f2 d0 = pxy[u][kk + 0] + az * az;
            az = pk_sub_bcast<1>(qza, pz[u]); const f2 d1 = pxy[u][kk + 1] + az * az;
            az = pk_sub_bcast<0>(qzb, pz[u]); const f2 d2 = pxy[u][kk + 2] + az * az;
            az = pk_sub_bcast<1>(qzb, pz[u]); const f2 d3 = pxy[u][kk + 3] + az * az;
            best[2 * u] = fmin_(fmin_(best[2 * u], d0.x), d1.x);
            best[2 * u] = fmin_(fmin_(best[2 * u], d2.x), d3.x);
            best[2 * u + 1] = fmin_(fmin_(best[2 * u + 1], d0.y), d1.y);
            best[2 * u + 1] = fmin_(fmin_(best[2 * u + 1], d2.y), d3.y);
        }
    }
}

// lower bound of every reference distance between the lane's points and a box: per axis
// g = max(lo - p, p - hi, 0) <= |q - p| for every q inside, rounding is monotonic, and L uses the reference's own
// association (gx*gx + gy*gy) + gz*gz, so L <= d operation by operation; the 2^-20 shave is belt and braces.
template <int TP, bool LE = false /*ties count: the chunks are not visited in ascending order*/>
__device__ __forceinline__ bool box_may_improve(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                                const f2 (&px)[TP], const f2 (&py)[TP], const f2 (&pz)[TP],
                                                const float (&best)[2 * TP])
{
    bool needb = false;
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        // g = p - clamp(p, lo, hi): one v_med3 per coordinate and point, one packed subtraction per axis -- |g| is max(lo - p,
        // p - hi, 0) bit for bit (a difference and its negation round alike), in 9 instructions instead of 12
        const f2 gx = px[u] - f2{__builtin_amdgcn_fmed3f(px[u].x, lox, hix), __builtin_amdgcn_fmed3f(px[u].y, lox, hix)};
        const f2 gy = py[u] - f2{__builtin_amdgcn_fmed3f(py[u].x, loy, hiy), __builtin_amdgcn_fmed3f(py[u].y, loy, hiy)};
        const f2 gz = pz[u] - f2{__builtin_amdgcn_fmed3f(pz[u].x, loz, hiz), __builtin_amdgcn_fmed3f(pz[u].y, loz, hiz)};
        f2 L = (gx * gx + gy * gy) + gz * gz;
        L = L * f2{0.99999905f, 0.99999905f};  // 1 - 2^-20
        if constexpr (LE) needb |= (L.x <= best[2 * u]) | (L.y <= best[2 * u + 1]);
        else needb |= (L.x < best[2 * u]) | (L.y < best[2 * u + 1]);
    }
    return needb;
}

// Optional fused TAIL of the matching kernel (TAIL = 1 point-to-point, 2 point-to-plane): instead of leaving
// per-segment (d, idx) partials for a second kernel, every block folds its result into one 64-bit key per moving
// point with a device-scope atomic min -- key = (float bits of d) << 32 | idx, so the integer order IS the
// lexicographic (d, idx) order the tie rule needs -- then draws a ticket for its row of moving points.  The block
// that draws the last ticket of a row (all S segment blocks have contributed) reads the final keys, stores idx,
// gathers q (and the normal) and produces the row's moment sums: the work of moments_kernel without a second
// launch, a dependent dispatch or the partial arrays.  Protocol (agent scope, placement independent): the payload
// is written ONLY by agent-scope atomics; every wave drains them (s_waitcnt vmcnt(0)) and the block barriers
// before one lane adds the ticket; the last arriver reads the keys back with agent-scope atomic loads.
struct NNTail {
    unsigned long long* keys;  // [n_pad], all ones between launches (the last block of a row resets them)
    unsigned int* tickets;     // [gridDim.x], zero between launches (reset by the last block)
    double* err_tile;          // [gridDim.x] device: error of the fused transform, from the grid.y == 0 block
    int32_t* idx_out;          // [n_pad]
    int32_t* idx_out_odd;      // resident launch: the odd passes' correspondences (ping-pong with idx_out)
    const float* Nrm;          // model normals (SoA, m_pad) for TAIL == 2
    double* rows;              // [gridDim.x][ICP_NMOM]: pinned host (single GPU) or device (finalize follows)
    double tag;                // completion tag stored in slot ICP_NMOM-1 of the row
    // Compact rows (sparse kernels, point-to-point, rows read by the host): what the host has to pull out of memory the
    // GPU has just written is part of the floor of a short iteration (tools/rows_probe.hip: 256 rows of 4 cache lines
    // 5.8 us, of 2 lines 5.0 us, against 2.4 us for one row; a separate array for the error shares, eight blocks to a
    // line, costs 2.7 us MORE), so a row shrinks from four cache lines to exactly two --
    //   [gridDim.x][NN_CROW = 16]: {error share + tag, sum p (3), sum q (3), sum q p^T (9)}.
    // The point count is not sent (the host knows how many real points a row holds), sum |p|^2 and sum |q|^2 are not sent
    // (nothing reads them), and the tag rides in the low NN_CROW_TAG_BITS mantissa bits of the error share, a sum of
    // squares whose last 16 bits (2^-36 of its value) nothing can resolve: slot 0 is written last, after the others have
    // drained, exactly as the separate tag word was.
    int compact;
    int rows_on_device;        // the rows are read by a later kernel, not by a polling host: plain stores, nothing to drain
    unsigned int tag_lo;       // low 32 bits of `tag` as an integer (a double -> integer conversion on the device expands to f64 fma code)
    int row;                   // the row this block closes, or -1: blockIdx.x (shared rows: a block's row is not its index)
    int idx_through;           // the correspondences leave as agent-scope (write-through) stores: a resident launch whose rows are closed now by one
                               // block, now by another -- two XCDs' L2s holding dirty copies of one line would write them back in no order
};
__device__ __forceinline__ double crow_pack(double err, unsigned int tag_lo)
{
    const unsigned long long t = (unsigned long long)tag_lo & ((1ull << NN_CROW_TAG_BITS) - 1ull);
    return __longlong_as_double((long long)(((unsigned long long)__double_as_longlong(err) & ~((1ull << NN_CROW_TAG_BITS) - 1ull)) | t));
}

// rigid motion applied to the moving cloud; travels by value in the kernel-argument segment
template <typename F> struct RT { F r[9]; F t[3]; };

// ((r0*x + r1*y) + r2*z) + t with separately rounded products and sums -- the association of RyT
// (src/ICP_point_to_point.cu:85).  One definition for every kernel that moves points, so the fused
// and the stand-alone transform produce the same bits.
template <typename F>
__device__ __forceinline__ void apply_rt(const RT<F>& rt, F x, F y, F z, F& ox, F& oy, F& oz)
{
    F o[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        F c = rt.r[a * 3 + 0] * x;
        c = c + rt.r[a * 3 + 1] * y;
        c = c + rt.r[a * 3 + 2] * z;
        o[a] = c + rt.t[a];
    }
    ox = o[0]; oy = o[1]; oz = o[2];
}

// optional fused front end of the matching kernel: the transform of the PREVIOUS pass
struct NNFuse {
    int apply;               // 0: match P as it is
    int n;                   // real moving points (the error and the seeds skip the padding)
    int m;                   // real model points (seed validation)
    const int32_t* idx_prev; // correspondences the applied (R, t) came from
    float* P_out;            // transformed cloud (written by the grid.y == 0 blocks only)
    double* err_rows;        // [gridDim.x] sum |p_new - q[idx_prev]|^2 per block
    const int32_t* seed_idx; // CULL kernels: any valid model index per moving point (or NULL); it only
                             // tightens the starting bound, the result does not depend on it
    const float* Q_gather;   // the unmodified model (Q passed to a CULL kernel has its exact duplicates voided)
    const float* boxes;      // CULL kernels: per 8-point chunk of the scan copy {lo.xyz, hi.xyz, -, -} (or NULL)
    int sample_groups;       // sparse kernel: at most this many groups of 8 samples are used by the cold start (<= 256)
    float resample_bound;    // hierarchical search: a SEEDED pass whose largest bound exceeds this takes the sample round too (0: never)
    const int32_t* q_perm;   // sparse kernel: the scan copy is spatially sorted; q_perm[sorted j] = model index (NULL: identity)
    const int32_t* p_perm;   // sparse kernel: slot -> moving point handled there (spatially sorted groups; NULL: identity)
    int store_first;         // resident launch reading a pristine copy: pass 0 stores the cloud to P_out even without a transform
    int resident;            // resident launch: after a pass the block waits for the next message instead of ending
    NNMailbox* relay;        // ... relayed by block 0 to the other blocks through this device-memory copy
    const NNMailbox* mailbox; // armed launch (sparse kernel): (R, t) arrive here from the host AFTER the kernel was enqueued
    double want;             // ... under this sequence number (it also tags the rows the pass writes)
    unsigned int want_lo;    // its low 32 bits (the mailbox tag of pass p is mailbox_tag(want + p) = (want_lo + p) | top bit:
                             // integer arithmetic -- a double -> integer conversion on the device expands to f64 fma code)
    const float* samples;    // sparse kernel: one point per chunk of the scan copy (SoA, round_up(m_pad/8, 8) entries) or NULL
    float* slot_state;       // sparse kernel, one launch per pass: the moving points and the matched model points IN SLOT ORDER
                             // (6 arrays of n_pad floats: p.xyz, q.xyz), written by every such pass for the next one -- its front
                             // end is then one level of coalesced loads instead of slot -> point -> seed -> model point; or NULL
    int slot_valid;          // ... the previous pass wrote them: read them
    int slot_flip;           // ... the points have two planes (a row may be searched by several blocks, only one of which stores the
                             // moved points -- never over what the others still read): read plane slot_flip, write the other.
                             // Layout: [points, plane 0: 3 x n_pad][matches: 3 x n_pad][points, plane 1: 3 x n_pad]
    long long* tlog;         // diagnostic (ICP_NN_PHASES): per-wave s_memrealtime stamps, 10 slots per wave, or NULL
    long long tlog_cap;      // slots available
    int tlog_pass;           // resident launch: stamp this pass only (-1: every pass, the last one survives)
    int speculate;           // resident launch: prepare the next pass's hit list while the block waits for its message (see the end of the pass loop)
    float spec_gain, spec_floor; // ... the guess: next displacement <= spec_gain x this one + spec_floor x the group box's extent
    unsigned long long* work; // diagnostic (icp_set_work_counting): NN_WORK_SLOTS device counters of the work the sparse kernel EXECUTES, or NULL
    // shared rows (nn_match_sparse, one launch per pass, more blocks than rows): hits per row of the PREVIOUS launch decide how
    // many blocks work on each row of this one -- see the kernel; NULL: one block per row (per model segment)
    const unsigned int* share_prev;
    unsigned int* share_cur;   // ... this launch's hits per row (added up by its blocks; zero when it starts)
    unsigned int* share_next;  // ... zeroed by this launch for the next one
    unsigned int* share_cur2;  // ... a second copy of this launch's counts (the first pass of a registration: kept for the next registration's first pass) or NULL
    unsigned int* share_zero2; // ... a second array to zero (the one the next first pass will add to) or NULL
    int share_rows;            // rows of the launch (<= threads of a block)
    int share_min;             // a part is never made smaller than this many hits (of the previous launch)
    const int32_t* row_order;  // ordered rows: block b works on row row_order[b] (heaviest first) -- or NULL
    unsigned int* row_hits;    // ... and adds the hits of its lists to row_hits[row]
    int refine_min, refine_cnt; // hierarchical search: a pass that lists at least refine_min super boxes takes a refinement round over <= refine_cnt of their chunk samples (0: never)
    int round_supers;          // hierarchical search: super boxes per round of the chunk find (<= 64: the hit list holds their chunks)
    const float* records;      // hierarchical search: one 160-byte record per chunk (model_records_kernel) -- a hit is fetched from it -- or NULL
    float* seed_pub;           // resident launch with shared rows: [rows][3][128] -- whichever block closes a split row leaves the matches' coordinates
                               // here (they seed the next pass and are what its error is measured against) for the row's other blocks
};

// phase stamp of the diagnostic log: one scalar branch when the log is off
#define ICP_PHASE(PH)                                                                                              \
    if constexpr (phase_diag_) if (fuse.tlog != nullptr && lane == 0 && (fuse.tlog_pass < 0 || fuse.tlog_pass == phase_pass_)) {                                                                       \
        const long long slot_ = (((long long)blockIdx.y * gridDim.x + blockIdx.x) * phase_nw_ + w) * 10 + (PH);                     \
        if (slot_ < fuse.tlog_cap) fuse.tlog[slot_] = (long long)wall_clock64();                                   \
    }

template <int T /*2 or 4*/, int C /*chunk: 8 or 16*/, int CULL /*0: plain; 1: seeded bound + box/xy early-out over LDS tiles*/, int TAIL = 0>
__global__ __launch_bounds__(NN_BLOCK, (T == 2 ? 8 : 4)) void nn_match_f32_v2(const float* __restrict__ P, int n_pad,
                                                               const float* __restrict__ Q, int m_pad, int seg_len,
                                                               float* __restrict__ part_d,
                                                               int32_t* __restrict__ part_idx, RT<float> rt,
                                                               NNFuse fuse, NNTail tail)
{
    constexpr int TP = T / 2;  // packed pairs of moving points per lane
    // one raw LDS block, carved by hand: the tail's transpose buffer overlays the tile + merge scratch
    constexpr int SQ_BYTES = 4 * 3 * NN2_TQW * 4, MD_BYTES = 4 * 64 * T * 4;
    constexpr int TR_BYTES = TAIL ? (TAIL == 2 ? 28 : 18) * 65 * 8 : 0;
    constexpr int LDS_BYTES = (SQ_BYTES + 2 * MD_BYTES + 16) > TR_BYTES ? (SQ_BYTES + 2 * MD_BYTES + 16) : TR_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    float (*sq)[3][NN2_TQW] = reinterpret_cast<float (*)[3][NN2_TQW]>(lds_raw);
    float (*md)[64 * T] = reinterpret_cast<float (*)[64 * T]>(lds_raw + SQ_BYTES);
    int (*mi)[64 * T] = reinterpret_cast<int (*)[64 * T]>(lds_raw + SQ_BYTES + MD_BYTES);
    int* s_flag = reinterpret_cast<int*>(lds_raw + SQ_BYTES + 2 * MD_BYTES);

    const int lane = threadIdx.x & 63;
    // the wave id as a SCALAR: everything derived from it (ranges, loop bounds, the box addresses) then lives in
    // SGPRs, the loops are scalar loops and the per-chunk boxes arrive through the scalar cache
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int phase_pass_ = 0, phase_nw_ = 4;  // (phase log)
    constexpr bool phase_diag_ = true;
    const int wseg = seg_len >> 2;              // model points per wave (multiple of C)
    const int q0 = blockIdx.y * seg_len;
    const int my0 = q0 + w * wseg;
    const int my1 = min(my0 + wseg, m_pad);     // may be <= my0: this wave's range is empty
    const int ibase = blockIdx.x * (64 * T) + lane;

    f2 px[TP], py[TP], pz[TP];
    float best[T];
    int cst[T];
    ICP_PHASE(0)
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        const int i0 = ibase + (2 * u) * 64, i1 = i0 + 64;
        px[u] = f2{P[i0], P[i1]};
        py[u] = f2{P[(size_t)n_pad + i0], P[(size_t)n_pad + i1]};
        pz[u] = f2{P[2 * (size_t)n_pad + i0], P[2 * (size_t)n_pad + i1]};
    }
    if (fuse.apply) {
        // every block re-derives the moved points in registers (same instructions => same bits);
        // the grid.y == 0 row stores them and accounts the error of the pass that produced (R, t)
        double err = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int u = t >> 1;
            float x = (t & 1) ? px[u].y : px[u].x, y = (t & 1) ? py[u].y : py[u].x, z = (t & 1) ? pz[u].y : pz[u].x;
            apply_rt<float>(rt, x, y, z, x, y, z);
            if (t & 1) { px[u].y = x; py[u].y = y; pz[u].y = z; } else { px[u].x = x; py[u].x = y; pz[u].x = z; }
            if (blockIdx.y == 0 && w == 0) {
                const int i = ibase + t * 64;
                fuse.P_out[i] = x;
                fuse.P_out[(size_t)n_pad + i] = y;
                fuse.P_out[2 * (size_t)n_pad + i] = z;
                if (i < fuse.n) {
                    const int j = fuse.idx_prev[i];
                    const float* Qg = fuse.Q_gather;
                    const double ex = (double)Qg[j] - (double)x;
                    const double ey = (double)Qg[(size_t)m_pad + j] - (double)y;
                    const double ez = (double)Qg[2 * (size_t)m_pad + j] - (double)z;
                    err += ex * ex + ey * ey + ez * ez;
                }
            }
        }
        if (blockIdx.y == 0 && w == 0) {
            err = wave_sum(err);
            if (lane == 0) {
                if constexpr (TAIL != 0)  // read by whichever block closes this row: agent-scope store, drained before our ticket
                    __hip_atomic_store(&tail.err_tile[blockIdx.x], err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    fuse.err_rows[blockIdx.x] = err;
            }
        }
    }
    ICP_PHASE(1)
#pragma unroll
    for (int t = 0; t < T; ++t) { best[t] = inf_<float>(); cst[t] = -1; }
    if constexpr (CULL) {
        // Seeded bound: start from the distance to ANY model point (last pass's match) bumped by one ulp.
        // The true minimum is <= that distance < bound, so the ordinary ascending strict-< scan still ends
        // on the first index of the minimum -- the seed changes how much work is skipped, never the answer.
        if (fuse.seed_idx) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int u = t >> 1;
                const float x = (t & 1) ? px[u].y : px[u].x, y = (t & 1) ? py[u].y : py[u].x, z = (t & 1) ? pz[u].y : pz[u].x;
                // padding lanes (i >= n) have no previous match, and a seed is trusted only if it is a
                // real model index: anything else simply starts unbounded
                const int i = ibase + t * 64;
                int j = (i < fuse.n) ? fuse.seed_idx[i] : -1;
                const bool ok = (unsigned)j < (unsigned)fuse.m;
                j = ok ? j : 0;
                const float* Qg = fuse.Q_gather;
                const float d = dist2<float>(x, y, z, Qg[j], Qg[(size_t)m_pad + j], Qg[2 * (size_t)m_pad + j]);
                // next float above d (d >= 0, finite): bit pattern + 1; inf stays inf
                best[t] = (ok && d < inf_<float>()) ? __uint_as_float(__float_as_uint(d) + 1u) : inf_<float>();
                // padding lanes can never improve on a bound of zero: they cost no chunk visits (their result,
                // "nothing found", is never read)
                best[t] = (i < fuse.n) ? best[t] : 0.f;
            }
        }
    }

    ICP_PHASE(2)
    {
    const int ntile = (wseg + NN2_TQW - 1) / NN2_TQW;
    for (int k = 0; k < ntile; ++k) {
        __syncthreads();
        // cooperative fill of the four per-wave sub-tiles: 4 x 3 x 256 floats = 768 float4, 3 per thread
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int v = threadIdx.x + r * NN_BLOCK;      // 0..767
            const int ww = v / 192, rem = v % 192;         // 192 float4 per wave sub-tile
            const int a = rem / 64, e = (rem % 64) * 4;    // coordinate array, element offset
            const int off = k * NN2_TQW + e;               // offset inside the wave's range
            const int src = q0 + ww * wseg + off;
            if (off < wseg && src < m_pad)
                *reinterpret_cast<float4*>(&sq[ww][a][e]) = *reinterpret_cast<const float4*>(&Q[(size_t)a * m_pad + src]);
        }
        __syncthreads();

        const int tile0 = my0 + k * NN2_TQW;
        const int len = min(NN2_TQW, my1 - tile0);  // multiple of C, <= 0 when exhausted
        for (int c = 0; c < len; c += C) {
            float bo[T];
#pragma unroll
            for (int t = 0; t < T; ++t) bo[t] = best[t];
            if constexpr (CULL) {
                // level 0: the chunk's bounding box (precomputed once per model over the scan copy): ~20 VALU ops
                // per chunk and lane pair instead of ~50, wave-uniform skip; then the xy early-out
                if (fuse.boxes) {
                    const float* bx = fuse.boxes + (size_t)((tile0 + c) / C) * 8;  // scalar address -> s_load
                    if (__builtin_amdgcn_ballot_w64(box_may_improve<TP>(bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], px, py, pz, best)) == 0ull)
                        continue;
                }
                scan_chunk_xy_cull<TP, C>(&sq[w][0][c], &sq[w][1][c], &sq[w][2][c], px, py, pz, best);
            } else {
#pragma unroll
                for (int kk = 0; kk < C; kk += 4) {
                    const float4 qx4 = *reinterpret_cast<const float4*>(&sq[w][0][c + kk]);
                    const float4 qy4 = *reinterpret_cast<const float4*>(&sq[w][1][c + kk]);
                    const float4 qz4 = *reinterpret_cast<const float4*>(&sq[w][2][c + kk]);
                    const f2 qxa = f2{qx4.x, qx4.y}, qxb = f2{qx4.z, qx4.w};
                    const f2 qya = f2{qy4.x, qy4.y}, qyb = f2{qy4.z, qy4.w};
                    const f2 qza = f2{qz4.x, qz4.y}, qzb = f2{qz4.z, qz4.w};
#pragma unroll
                    for (int u = 0; u < TP; ++u) {
                        const f2 d0 = pk_dist2<0>(qxa, qya, qza, px[u], py[u], pz[u]);
                        const f2 d1 = pk_dist2<1>(qxa, qya, qza, px[u], py[u], pz[u]);
                        const f2 d2 = pk_dist2<0>(qxb, qyb, qzb, px[u], py[u], pz[u]);
                        const f2 d3 = pk_dist2<1>(qxb, qyb, qzb, px[u], py[u], pz[u]);
                        best[2 * u] = fmin_(fmin_(best[2 * u], d0.x), d1.x);
                        best[2 * u] = fmin_(fmin_(best[2 * u], d2.x), d3.x);
                        best[2 * u + 1] = fmin_(fmin_(best[2 * u + 1], d0.y), d1.y);
                        best[2 * u + 1] = fmin_(fmin_(best[2 * u + 1], d2.y), d3.y);
                    }
                }
            }
            bool any = false;
#pragma unroll
            for (int t = 0; t < T; ++t) any |= best[t] < bo[t];
            if (__builtin_amdgcn_ballot_w64(any) != 0ull) {  // wave-uniform: skipped while no minimum moved
                const int cid = (tile0 + c) / C;
#pragma unroll
                for (int t = 0; t < T; ++t) cst[t] = (best[t] < bo[t]) ? cid : cst[t];
            }
        }
    }
    }  // tile scan
    ICP_PHASE(3)

    // index recovery inside the winning chunk (lowest j with d_j == min), then the in-block merge
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float pxt = (t & 1) ? px[t >> 1].y : px[t >> 1].x;
        const float pyt = (t & 1) ? py[t >> 1].y : py[t >> 1].x;
        const float pzt = (t & 1) ? pz[t >> 1].y : pz[t >> 1].x;
        const bool found = cst[t] >= 0;  // this wave's range lowered the (possibly seeded) bound at least once
        const int base = found ? cst[t] * C : 0;
        int idx = 0x7fffffff;
        const float b = found ? best[t] : inf_<float>();
        if (found) {
            idx = base;
#pragma unroll 4
            for (int kk = C - 1; kk >= 0; --kk) {
                const int j = base + kk;
                const float d = dist2<float>(pxt, pyt, pzt, Q[j], Q[(size_t)m_pad + j], Q[2 * (size_t)m_pad + j]);
                idx = (d == b) ? j : idx;
            }
        }
        md[w][lane + t * 64] = b;
        mi[w][lane + t * 64] = idx;
    }
    ICP_PHASE(4)
    __syncthreads();
    ICP_PHASE(5)
    if (threadIdx.x < 64 * T) {
        float b = md[0][threadIdx.x];
        int bi = mi[0][threadIdx.x];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const float d = md[ww][threadIdx.x];
            const int j = mi[ww][threadIdx.x];
            if (d < b) { b = d; bi = j; }
        }
        if constexpr (TAIL == 0) {
            const size_t o = (size_t)blockIdx.y * n_pad + (size_t)blockIdx.x * (64 * T) + threadIdx.x;
            part_d[o] = b;
            part_idx[o] = bi;
        } else {
            const unsigned long long key = ((unsigned long long)__float_as_uint(b) << 32) | (unsigned int)bi;
            __hip_atomic_fetch_min(&tail.keys[(size_t)blockIdx.x * (64 * T) + threadIdx.x], key, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (TAIL != 0) {
        static_assert(TAIL == 0 || T == 2, "the fused tail is written for two moving points per lane");
        // every wave drains its atomics, the block meets, one lane draws the row's ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ICP_PHASE(6)
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int ticket = __hip_atomic_fetch_add(&tail.tickets[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_flag = (ticket == gridDim.y - 1) ? 1 : 0;
        }
        __syncthreads();
        ICP_PHASE(7)
        if (*s_flag == 0 || w != 0) return;  // only wave 0 of the row's last block goes on (the LDS is all its own now)

        constexpr int NACC = TAIL == 2 ? 28 : 18;
        double acc[NACC];
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
        const float* Qg = fuse.Q_gather;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = ibase + t * 64;
            const unsigned long long key = __hip_atomic_load(&tail.keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tail.keys[i] = ~0ull;  // ready for the next launch (nobody touches this row again in this one)
            int j = (int)(unsigned int)(key & 0xffffffffull);
            j = ((unsigned)j < (unsigned)fuse.m) ? j : fuse.m - 1;  // unreachable clamp, keeps idx in range by construction
            if (i < fuse.n) {
                tail.idx_out[i] = j;
                const double ppx = (double)(t ? px[0].y : px[0].x), ppy = (double)(t ? py[0].y : py[0].x),
                             ppz = (double)(t ? pz[0].y : pz[0].x);
                const double qx = (double)Qg[j], qy = (double)Qg[(size_t)m_pad + j], qz = (double)Qg[2 * (size_t)m_pad + j];
                acc[0] += 1.0;
                if constexpr (TAIL == 1) {
                    acc[1] += ppx; acc[2] += ppy; acc[3] += ppz;
                    acc[4] += qx; acc[5] += qy; acc[6] += qz;
                    acc[7] += qx * ppx; acc[8] += qx * ppy; acc[9] += qx * ppz;
                    acc[10] += qy * ppx; acc[11] += qy * ppy; acc[12] += qy * ppz;
                    acc[13] += qz * ppx; acc[14] += qz * ppy; acc[15] += qz * ppz;
                    acc[16] += ppx * ppx + ppy * ppy + ppz * ppz;
                    acc[17] += qx * qx + qy * qy + qz * qz;
                } else {
                    const double nx = (double)tail.Nrm[j], ny = (double)tail.Nrm[(size_t)m_pad + j],
                                 nz = (double)tail.Nrm[2 * (size_t)m_pad + j];
                    double cn[6];
                    cn[0] = ppy * nz - ppz * ny;
                    cn[1] = ppz * nx - ppx * nz;
                    cn[2] = ppx * ny - ppy * nx;
                    cn[3] = nx; cn[4] = ny; cn[5] = nz;
                    const double bb = (ppx - qx) * nx + (ppy - qy) * ny + (ppz - qz) * nz;
                    int o = 1;
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int c2 = a; c2 < 6; ++c2) acc[o++] += cn[a] * cn[c2];
#pragma unroll
                    for (int a = 0; a < 6; ++a) acc[22 + a] -= cn[a] * bb;
                }
            }
        }
        // one wave: transpose through LDS (rows padded to 65 doubles), lane k adds slot k in lane order
        double (*tr)[65] = reinterpret_cast<double (*)[65]>(lds_raw);
#pragma unroll
        for (int k = 0; k < NACC; ++k) tr[k][lane] = acc[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // same wave: DS ops are in order; this pins the compiler
        double* row = tail.rows + (size_t)blockIdx.x * ICP_NMOM;
        if (lane < NACC) {
            double sum = 0.0;
#pragma unroll 8
            for (int l = 0; l < 64; ++l) sum += tr[lane][l];
            row[1 + lane] = sum;
        }
        if (lane == 0) {
            row[ICP_MOM_ERR] = fuse.apply ? __hip_atomic_load(&tail.err_tile[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            tail.tickets[blockIdx.x] = 0u;
        }
        __threadfence_system();  // the row is visible to a polling host before its tag
        if (lane == 0) __hip_atomic_store(&row[ICP_NMOM - 1], tail.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ICP_PHASE(8)
    }
}

// ------------------------------------------------------------------------------------------------
// matching, fp32, sparse -- the shipped kernel whenever the model has chunk boxes.
//
// The phase log of the tiled early-out kernel (ICP_NN_PHASES) showed what was left once ~99 % of the chunks
// were being skipped: the chunks that do survive all sit in the range of ONE wave of the 64 that share a group
// of moving points, and that wave worked through them alone (median scan 1 us, slowest 17 us of a 27 us kernel).
// This kernel separates FINDING the surviving chunks from PROCESSING them:
//   * a block is SP_NW waves that all hold the same 128 moving points (two per lane, packed);
//   * find: the bounding box G of the 128 points and their largest running bound B are wave-uniform, so the
//     test "chunk box closer to G than B" runs lane-parallel -- lane l tests chunk l, 64 chunks per ~25 VALU ops;
//     survivors are appended to a hit list in LDS;
//   * process: the hits are dealt round-robin to the waves.  Each goes through the per-point box test and the
//     xy early-out as before; its coordinates arrive through the scalar cache (wave-uniform address).
// The list is unordered (atomic append), so the tie rule is explicit here instead of implied by scan order:
// a chunk takes a point's minimum if its own minimum is smaller, or equal with a lower chunk number; the
// pruning tests therefore let ties through (<=).  Between rounds of the find step the waves exchange their
// minima through LDS and restart from the best one bumped by an ulp (the seeded-bound argument again), which is
// what makes an unseeded (cold) pass converge quickly too.  Results are bit-identical to the plain scan.
// Round 2 added, on the same skeleton: 8-wave blocks (two to a CU) whose launches deal their spare blocks to the heavy rows
// (shared rows, clouds of 33-57 k points); for the hierarchical search the rows taken heaviest first (ordered rows) and 16
// hits per trip to memory; group boxes over a row's real points only; a 16-bit hit list for the flat search.
// ------------------------------------------------------------------------------------------------
constexpr int SP_NW = 16;                       // waves per block (the default; NWS = 8 is the other instantiation)
constexpr int SP_HCAP = 4096;                   // hit-list entries of the hierarchical search = chunks per round (SP_NW * 64 * passes <= this)
constexpr int SP_MAX_PASSES = SP_HCAP / (SP_NW * 64);
// flat search (models below 2^19 points = 65 536 chunks): the list holds 16-bit chunk numbers, twice as many in the same
// 16 KB -- a model of up to 65 536 points is one round of the find (Bunny.csv: 5040 chunks, two rounds with 4096 entries)
constexpr int SP_HCAP_FLAT = 2 * SP_HCAP;
template <bool HIER> struct SpHit { using type = unsigned short; };
template <> struct SpHit<true> { using type = int; };

// one hit chunk against the lane's packed pair; (best, bj) follow the lexicographic (distance, MODEL index) rule:
// the chunk takes a point's minimum if its own minimum is smaller, or equal with a lower model index.
// `sb` is the hit's LDS stage {box 8, x 8, y 8, z 8, model index 8}, read level by level: most hits end at the box
// test or at the xy early-out, and with many hits per wave the 16 waves share the LDS bandwidth (reading a hit in
// one go was measured: no gain on the hall scan, 13 % slower on the hit-heavy grid).
// PERM: the scan copy is a sorted view, element k of the chunk is model point qo[k] (looked at only on the rare path
// where the chunk's minimum reaches the running one); else it is point ch * 8 + k and "lowest model index" is
// simply "lowest k".
// Returns how far the hit got (wave-uniform; only the work-counting instantiation looks at it): 0 = rejected by the
// per-point box test, 1 = by the xy early-out, 2 = the eight distances were evaluated in full.
constexpr bool SP_XY_EARLY_OUT = false;
template <bool PERM>
__device__ __forceinline__ int scan_hit(const float* sb, int ch, const f2 px, const f2 py, const f2 pz, float (&best)[2], int (&bj)[2],
                                        float (&bq)[2][3])
{
    constexpr int C = 8;
    {
        // level 0: the chunk's bounding box against each of the lane's points (ties pass: the hits are unordered)
        const f2 pxa[1] = {px}, pya[1] = {py}, pza[1] = {pz};
        if (__builtin_amdgcn_ballot_w64(box_may_improve<1, true>(sb[0], sb[1], sb[2], sb[3], sb[4], sb[5], pxa, pya, pza, best)) == 0ull) return 0;
    }
    const float *qxp = sb + 8, *qyp = sb + 16, *qzp = sb + 24;
    f2 d[C];  // first dx*dx + dy*dy (the inner sum of the reference's association), then the distances
    float mxy0 = inf_<float>(), mxy1 = inf_<float>();
#pragma unroll
    for (int kk = 0; kk < C; kk += 4) {
        const float4 qx4 = *reinterpret_cast<const float4*>(qxp + kk);
        const float4 qy4 = *reinterpret_cast<const float4*>(qyp + kk);
        const f2 qxa = f2{qx4.x, qx4.y}, qxb = f2{qx4.z, qx4.w};
        const f2 qya = f2{qy4.x, qy4.y}, qyb = f2{qy4.z, qy4.w};
        f2 ax, ay;
        ax = pk_sub_bcast<0>(qxa, px); ay = pk_sub_bcast<0>(qya, py);
        d[kk + 0] = ax * ax + ay * ay;
        ax = pk_sub_bcast<1>(qxa, px); ay = pk_sub_bcast<1>(qya, py);
        d[kk + 1] = ax * ax + ay * ay;
        ax = pk_sub_bcast<0>(qxb, px); ay = pk_sub_bcast<0>(qyb, py);
        d[kk + 2] = ax * ax + ay * ay;
        ax = pk_sub_bcast<1>(qxb, px); ay = pk_sub_bcast<1>(qyb, py);
        d[kk + 3] = ax * ax + ay * ay;
        if constexpr (SP_XY_EARLY_OUT) {
            mxy0 = fmin_(fmin_(mxy0, d[kk].x), d[kk + 1].x);
            mxy0 = fmin_(fmin_(mxy0, d[kk + 2].x), d[kk + 3].x);
            mxy1 = fmin_(fmin_(mxy1, d[kk].y), d[kk + 1].y);
            mxy1 = fmin_(fmin_(mxy1, d[kk + 2].y), d[kk + 3].y);
        }
    }
    // d = fl(pxy + dz*dz) >= pxy: a chunk whose smallest pxy is above every lane's minimum cannot matter (ties pass).
    // (Round 3: compiled out.  Behind the per-point box test this early-out stops 6-7 % of the chunks that reach it -- 10 M x 10 M:
    // 328 M in, 304 M on; Bunny.csv 338 k / 318 k; the hall scan 56 565 / 56 565 -- and costs every one of them eight v_min, a
    // ballot and a branch between the two halves of the arithmetic; nn_match_row64 dropped it in round 2 for the same reason.)
    if constexpr (SP_XY_EARLY_OUT) {
        if (__builtin_amdgcn_ballot_w64((mxy0 <= best[0]) | (mxy1 <= best[1])) == 0ull) return 1;
    }
    float c0 = inf_<float>(), c1 = inf_<float>();  // the chunk's own minima
#pragma unroll
    for (int kk = 0; kk < C; kk += 4) {
        const float4 qz4 = *reinterpret_cast<const float4*>(qzp + kk);
        const f2 qza = f2{qz4.x, qz4.y}, qzb = f2{qz4.z, qz4.w};
        f2 az;
        az = pk_sub_bcast<0>(qza, pz); d[kk + 0] = d[kk + 0] + az * az;
        az = pk_sub_bcast<1>(qza, pz); d[kk + 1] = d[kk + 1] + az * az;
        az = pk_sub_bcast<0>(qzb, pz); d[kk + 2] = d[kk + 2] + az * az;
        az = pk_sub_bcast<1>(qzb, pz); d[kk + 3] = d[kk + 3] + az * az;
        c0 = fmin_(fmin_(c0, d[kk].x), d[kk + 1].x);
        c0 = fmin_(fmin_(c0, d[kk + 2].x), d[kk + 3].x);
        c1 = fmin_(fmin_(c1, d[kk].y), d[kk + 1].y);
        c1 = fmin_(fmin_(c1, d[kk + 2].y), d[kk + 3].y);
    }
    if constexpr (PERM) {
        const bool cand0 = c0 <= best[0], cand1 = c1 <= best[1];
        if (__builtin_amdgcn_ballot_w64(cand0 | cand1) != 0ull) {
            // lowest model index among the chunk elements at the chunk's minimum, and where it sits
            const int* qo = reinterpret_cast<const int*>(sb + 32);
            int o0 = 0x7fffffff, o1 = 0x7fffffff, k0 = 0, k1 = 0;
#pragma unroll
            for (int kk = C - 1; kk >= 0; --kk) {
                const int oj = qo[kk];  // wave-uniform address: one broadcast read
                const bool e0 = (d[kk].x == c0) & (oj < o0), e1 = (d[kk].y == c1) & (oj < o1);
                o0 = e0 ? oj : o0; k0 = e0 ? kk : k0;
                o1 = e1 ? oj : o1; k1 = e1 ? kk : k1;
            }
            const bool take0 = cand0 & ((c0 < best[0]) | (bj[0] < 0) | (o0 < bj[0]));  // bj < 0: nothing to tie with yet
            const bool take1 = cand1 & ((c1 < best[1]) | (bj[1] < 0) | (o1 < bj[1]));
            best[0] = take0 ? c0 : best[0];
            bj[0] = take0 ? o0 : bj[0];
            best[1] = take1 ? c1 : best[1];
            bj[1] = take1 ? o1 : bj[1];
            if (take0) { bq[0][0] = qxp[k0]; bq[0][1] = qyp[k0]; bq[0][2] = qzp[k0]; }
            if (take1) { bq[1][0] = qxp[k1]; bq[1][1] = qyp[k1]; bq[1][2] = qzp[k1]; }
        }
    } else {
        // identity order: chunks are disjoint index ranges, "lower model index" is "lower chunk, then lower k"
        const bool take0 = (c0 < best[0]) | ((c0 == best[0]) & (ch < (bj[0] >> 3)));  // bj = -1: nothing to tie with
        const bool take1 = (c1 < best[1]) | ((c1 == best[1]) & (ch < (bj[1] >> 3)));
        if (__builtin_amdgcn_ballot_w64(take0 | take1) != 0ull) {
            int k0 = C - 1, k1 = C - 1;
#pragma unroll
            for (int kk = C - 2; kk >= 0; --kk) {
                k0 = (d[kk].x == c0) ? kk : k0;
                k1 = (d[kk].y == c1) ? kk : k1;
            }
            best[0] = take0 ? c0 : best[0];
            bj[0] = take0 ? ch * C + k0 : bj[0];
            best[1] = take1 ? c1 : best[1];
            bj[1] = take1 ? ch * C + k1 : bj[1];
            // the coordinates of the new minimum are at hand (LDS stage): keeping them saves the closing wave a
            // dependent gather from global memory
            if (take0) { bq[0][0] = qxp[k0]; bq[0][1] = qyp[k0]; bq[0][2] = qzp[k0]; }
            if (take1) { bq[1][0] = qxp[k1]; bq[1][1] = qyp[k1]; bq[1][2] = qzp[k1]; }
        }
    }
    return 2;
}

// distances from the lane's packed pair to 8 model points, folded into running minima (no index)
__device__ __forceinline__ void scan8_min(const float4 qx0, const float4 qx1, const float4 qy0, const float4 qy1,
                                          const float4 qz0, const float4 qz1, const f2 px, const f2 py, const f2 pz,
                                          float (&best)[2])
{
    const f2 qx[4] = {f2{qx0.x, qx0.y}, f2{qx0.z, qx0.w}, f2{qx1.x, qx1.y}, f2{qx1.z, qx1.w}};
    const f2 qy[4] = {f2{qy0.x, qy0.y}, f2{qy0.z, qy0.w}, f2{qy1.x, qy1.y}, f2{qy1.z, qy1.w}};
    const f2 qz[4] = {f2{qz0.x, qz0.y}, f2{qz0.z, qz0.w}, f2{qz1.x, qz1.y}, f2{qz1.z, qz1.w}};
#pragma unroll
    for (int k = 0; k < 4; k += 2) {
        f2 ax, ay, az;
        ax = pk_sub_bcast<0>(qx[k], px); ay = pk_sub_bcast<0>(qy[k], py); az = pk_sub_bcast<0>(qz[k], pz);
        const f2 d0 = (ax * ax + ay * ay) + az * az;
        ax = pk_sub_bcast<1>(qx[k], px); ay = pk_sub_bcast<1>(qy[k], py); az = pk_sub_bcast<1>(qz[k], pz);
        const f2 d1 = (ax * ax + ay * ay) + az * az;
        ax = pk_sub_bcast<0>(qx[k + 1], px); ay = pk_sub_bcast<0>(qy[k + 1], py); az = pk_sub_bcast<0>(qz[k + 1], pz);
        const f2 d2 = (ax * ax + ay * ay) + az * az;
        ax = pk_sub_bcast<1>(qx[k + 1], px); ay = pk_sub_bcast<1>(qy[k + 1], py); az = pk_sub_bcast<1>(qz[k + 1], pz);
        const f2 d3 = (ax * ax + ay * ay) + az * az;
        best[0] = fmin_(fmin_(best[0], d0.x), d1.x);
        best[0] = fmin_(fmin_(best[0], d2.x), d3.x);
        best[1] = fmin_(fmin_(best[1], d0.y), d1.y);
        best[1] = fmin_(fmin_(best[1], d2.y), d3.y);
    }
}

template <int TAIL, bool phase_diag_, int NWP, bool ROWARG = false>
__device__ __forceinline__ void tail_reduce_store(double (*tr)[65], int lane, const NNFuse& fuse, const NNTail& tail, double err_row, int phase_pass_);

// moment row of one row of 128 moving points, by ONE wave holding them two per lane (px.x = point lane,
// px.y = point lane + 64) with their final correspondences j[]: stores idx, gathers q (and the normal),
// accumulates in fp64, reduces through LDS in lane order and writes the row + completion tag.
// (qio: the coordinates of the correspondences -- gathered here when `gather`, else supplied by the caller)
// (ONE: only the lane's first point exists -- rows of 64 points; its contributions go straight to the transpose buffer
// instead of through 18 register pairs)
// (ROWARG: the row and the kind of index store come with the tail arguments -- nn_match_sparse, whose blocks may share rows; the
// other kernels close row blockIdx.x with plain stores, and do not pay for the choice)
template <int TAIL, bool phase_diag_, int NWP = SP_NW, bool ONE = false, bool ROWARG = false>
__device__ __forceinline__ void tail_close_row(const f2 px, const f2 py, const f2 pz, const int (&j)[2], int lane, const int (&pi)[2],
                                               int m_pad, const NNFuse& fuse, const NNTail& tail, double err_row,
                                               unsigned char* lds_raw, float (&qio)[2][3], bool gather, int phase_pass_ = 0)
{
    constexpr int w = 0, phase_nw_ = NWP;  // (phase log) the closing wave of a sparse-kernel block
    constexpr int NACC = TAIL == 2 ? 28 : 18;
    // one wave: the lanes' contributions are transposed through LDS (rows padded to 65 doubles), slot k is then added
    // up in lane order
    double (*tr)[65] = reinterpret_cast<double (*)[65]>(lds_raw);
    const float* Qg = fuse.Q_gather;
    if constexpr (TAIL == 1 && ONE) {
        const int i = fresh(pi[0]);
        const bool live = i < fuse.n;
        double ppx = 0.0, ppy = 0.0, ppz = 0.0, qx = 0.0, qy = 0.0, qz = 0.0;
        if (live) {
            const int jj = j[0];
            if (ROWARG && tail.idx_through) __hip_atomic_store(&tail.idx_out[i], jj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else tail.idx_out[i] = jj;
            ppx = (double)px.x; ppy = (double)py.x; ppz = (double)pz.x;
            if (gather) { qio[0][0] = Qg[jj]; qio[0][1] = Qg[(size_t)m_pad + jj]; qio[0][2] = Qg[2 * (size_t)m_pad + jj]; }
            qx = (double)qio[0][0]; qy = (double)qio[0][1]; qz = (double)qio[0][2];
        }
        // (0.0 + v: the value the two-point routine's accumulator holds after its one addition)
        tr[0][lane] = 0.0 + (live ? 1.0 : 0.0);
        tr[1][lane] = 0.0 + ppx; tr[2][lane] = 0.0 + ppy; tr[3][lane] = 0.0 + ppz;
        tr[4][lane] = 0.0 + qx; tr[5][lane] = 0.0 + qy; tr[6][lane] = 0.0 + qz;
        tr[7][lane] = 0.0 + qx * ppx; tr[8][lane] = 0.0 + qx * ppy; tr[9][lane] = 0.0 + qx * ppz;
        tr[10][lane] = 0.0 + qy * ppx; tr[11][lane] = 0.0 + qy * ppy; tr[12][lane] = 0.0 + qy * ppz;
        tr[13][lane] = 0.0 + qz * ppx; tr[14][lane] = 0.0 + qz * ppy; tr[15][lane] = 0.0 + qz * ppz;
        tr[16][lane] = 0.0 + (ppx * ppx + ppy * ppy + ppz * ppz);
        tr[17][lane] = 0.0 + (qx * qx + qy * qy + qz * qz);
        ICP_PHASE(7)
    } else if constexpr (TAIL == 1) {
        double acc[NACC];
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = fresh(pi[t]);
            if (i < fuse.n) {
                const int jj = j[t];
                if (ROWARG && tail.idx_through) __hip_atomic_store(&tail.idx_out[i], jj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else tail.idx_out[i] = jj;
                const double ppx = (double)(t ? px.y : px.x), ppy = (double)(t ? py.y : py.x), ppz = (double)(t ? pz.y : pz.x);
                if (gather) { qio[t][0] = Qg[jj]; qio[t][1] = Qg[(size_t)m_pad + jj]; qio[t][2] = Qg[2 * (size_t)m_pad + jj]; }
                const double qx = (double)qio[t][0], qy = (double)qio[t][1], qz = (double)qio[t][2];
                acc[0] += 1.0;
                acc[1] += ppx; acc[2] += ppy; acc[3] += ppz;
                acc[4] += qx; acc[5] += qy; acc[6] += qz;
                acc[7] += qx * ppx; acc[8] += qx * ppy; acc[9] += qx * ppz;
                acc[10] += qy * ppx; acc[11] += qy * ppy; acc[12] += qy * ppz;
                acc[13] += qz * ppx; acc[14] += qz * ppy; acc[15] += qz * ppz;
                acc[16] += ppx * ppx + ppy * ppy + ppz * ppz;
                acc[17] += qx * qx + qy * qy + qz * qz;
            }
        }
        ICP_PHASE(7)
#pragma unroll
        for (int k = 0; k < NACC; ++k) tr[k][lane] = acc[k];
    } else {
        // point-to-plane: 28 sums.  The second point's terms are added to the first one's in LDS rather than in 28
        // register pairs (the sums are the same, 0 + x0 + x1; the kernel no longer spills)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = fresh(pi[t]);
            const bool live = i < fuse.n;
            double cn[6] = {0, 0, 0, 0, 0, 0}, bb = 0.0;
            if (live) {
                const int jj = j[t];
                if (ROWARG && tail.idx_through) __hip_atomic_store(&tail.idx_out[i], jj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else tail.idx_out[i] = jj;
                const double ppx = (double)(t ? px.y : px.x), ppy = (double)(t ? py.y : py.x), ppz = (double)(t ? pz.y : pz.x);
                if (gather) { qio[t][0] = Qg[jj]; qio[t][1] = Qg[(size_t)m_pad + jj]; qio[t][2] = Qg[2 * (size_t)m_pad + jj]; }
                const double qx = (double)qio[t][0], qy = (double)qio[t][1], qz = (double)qio[t][2];
                const double nx = (double)tail.Nrm[jj], ny = (double)tail.Nrm[(size_t)m_pad + jj],
                             nz = (double)tail.Nrm[2 * (size_t)m_pad + jj];
                cn[0] = ppy * nz - ppz * ny;
                cn[1] = ppz * nx - ppx * nz;
                cn[2] = ppx * ny - ppy * nx;
                cn[3] = nx; cn[4] = ny; cn[5] = nz;
                bb = (ppx - qx) * nx + (ppy - qy) * ny + (ppz - qz) * nz;
            }
            auto put = [&](int k, double v) {
                if (t == 0) tr[k][lane] = 0.0 + v; else tr[k][lane] = tr[k][lane] + v;
            };
            put(0, live ? 1.0 : 0.0);
            int o = 1;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c2 = a; c2 < 6; ++c2) put(o++, cn[a] * cn[c2]);
#pragma unroll
            for (int a = 0; a < 6; ++a) put(22 + a, -(cn[a] * bb));
        }
        ICP_PHASE(7)
    }
    tail_reduce_store<TAIL, phase_diag_, NWP, ROWARG>(tr, lane, fuse, tail, err_row, phase_pass_);
}

// second half of a row tail: the transpose buffer is summed slot by slot in a fixed order and the row goes out
template <int TAIL, bool phase_diag_, int NWP, bool ROWARG>
__device__ __forceinline__ void tail_reduce_store(double (*tr)[65], int lane, const NNFuse& fuse, const NNTail& tail, double err_row, int phase_pass_)
{
    constexpr int w = 0, phase_nw_ = NWP;  // (phase log) the closing wave of a sparse-kernel block
    constexpr int NACC = TAIL == 2 ? 28 : 18;
    lds_same_wave_order();
    const unsigned int rowi = (ROWARG && tail.row >= 0) ? (unsigned int)tail.row : blockIdx.x;
    double* row = tail.rows + (size_t)rowi * ICP_NMOM;
    // Slot k is the sum of its 64 lane entries in a FIXED order: PARTS lanes per slot add a contiguous share each
    // (loaded first, added after: the LDS latencies overlap), the shares are then added in part order.
    constexpr int PARTS = 64 / NACC, PER = (64 + PARTS - 1) / PARTS;
    double* tp = &tr[NACC][0];  // PARTS x NACC partial sums, behind the transpose rows
    {
        const int slot = lane % NACC, part = lane / NACC;
        if (part < PARTS) {
            constexpr int CH = PER > 22 ? 16 : PER;   // loads in flight: all of a share, or 16 at a time for the long ones
            double sum = 0.0;
#pragma unroll
            for (int l0 = 0; l0 < PER; l0 += CH) {
                double v[CH];   // loads first, then the adds: one LDS latency per CH entries instead of one per entry
#pragma unroll
                for (int l = 0; l < CH; ++l) v[l] = (l0 + l < PER && part * PER + l0 + l < 64) ? tr[slot][part * PER + l0 + l] : 0.0;
#pragma unroll
                for (int l = 0; l < CH; ++l) sum += v[l];
            }
            tp[part * NACC + slot] = sum;
        }
    }
    lds_same_wave_order();
    // The row goes out as system-scope (write-through) stores, drained before the tag is issued: the host may
    // read the row as soon as it sees the tag.  (No L2 write-back here -- it would flush the whole cache for the
    // sake of 19 doubles; the other outputs of the pass are for later kernels and become visible at kernel end.)
    const bool compact = TAIL == 1 && tail.compact != 0;
    if (compact) row = tail.rows + (size_t)rowi * NN_CROW;
    if (tail.rows_on_device != 0) {
        // (round 3: rows that a later kernel adds up -- finalize, clouds of many rows or a device communicator -- need neither
        // write-through stores nor the wait for them nor a tag: the kernel boundary orders them)
        if (lane < NACC) {
            double sum = tp[lane];
#pragma unroll
            for (int q = 1; q < PARTS; ++q) sum += tp[q * NACC + lane];
            row[1 + lane] = sum;
        }
        if (lane == 0) { row[ICP_MOM_ERR] = err_row; row[ICP_NMOM - 1] = tail.tag; }
        ICP_PHASE(8)
        return;
    }
    if (lane < NACC) {
        double sum = tp[lane];
#pragma unroll
        for (int q = 1; q < PARTS; ++q) sum += tp[q * NACC + lane];
        // (slot lane of the sums is the count for lane 0, then sum p, sum q, sum q p^T: lanes 1..15 are the compact row's slots 1..15)
        if (!compact) __hip_atomic_store(&row[1 + lane], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (lane >= 1 && lane < NN_CROW) __hip_atomic_store(&row[lane], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (lane == 0 && !compact) __hip_atomic_store(&row[ICP_MOM_ERR], err_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    ICP_PHASE(8)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(compact ? &row[0] : &row[ICP_NMOM - 1], compact ? crow_pack(err_row, tail.tag_lo) : tail.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// DIAG: the phase-stamp instrumentation (ICP_NN_PHASES) is compiled into its own instantiation -- its pointers and
// branches cost the production kernel scalar registers it does not have to spare
// PERM: the scan copy is a Morton-ordered view with a permutation (compiled apart as well: carrying both forms of the
// hit processing in one loop body cost the common, identity-order case 16 % on a hit-heavy cloud)
// HIER: two-level search (large models): boxes of 64 chunks are tested first, lane-parallel like the chunks, and only
// the chunks of the surviving ones after them -- compiled apart for the same reason (the extra level costs a small
// model more than it saves)
template <int TAIL, bool DIAG, bool PERM, bool HIER = false, int NWS = SP_NW>
__global__ __launch_bounds__(NWS * 64, NWS == 16 ? 1 : 4) void nn_match_sparse(const float* __restrict__ P, int n_pad,
                                                              const float* __restrict__ Q, int m_pad, int seg_len,
                                                              int round_passes, float* __restrict__ part_d,
                                                              int32_t* __restrict__ part_idx, RT<float> rt_arg, NNFuse fuse,
                                                              NNTail tail)
{
    constexpr int STG = PERM ? 40 : 32;  // floats per staged hit: box 8, x 8, y 8, z 8 (, model indices 8)
    // The head of the block's LDS is an overlay: the hit list, under it the cold start's sample stage (3 x SMAX floats, used
    // before there is a list) and the tail's transpose buffer (after it), and behind both the 128 merge keys.  16 waves:
    // 2048 samples, 32 KB in all.  8 waves (two blocks share a CU's 160 KB): 1024 samples, so that the keys follow the
    // hit list directly -- 17 KB.
    constexpr int SMAX = NWS != 16 ? 1024 : 2048;
    constexpr int HITS_BYTES = SP_HCAP * 4, MQ_BYTES = NWS * 128 * 4;
    constexpr int SUPER_BYTES = HIER ? 2 * NWS * 64 * 4 : 0;   // (hierarchical search: the super-box hit list lies behind the chunk hit list)
    constexpr int MKEY_OFF = 3 * SMAX * 4 > HITS_BYTES + SUPER_BYTES ? 3 * SMAX * 4 : HITS_BYTES + SUPER_BYTES;
    constexpr int OVL_BYTES = NWS != 16 ? MKEY_OFF + 128 * 8 : HITS_BYTES + 2 * SP_NW * 128 * 4;
    constexpr int TR_BYTES = TAIL ? ((TAIL == 2 ? 28 : 18) * 65 + 64) * 8 : 0;
    static_assert(TR_BYTES <= HITS_BYTES, "the tail's transpose buffer overlays the hit list");
    // hits a wave fetches per trip to memory (its stage holds them): 8 per gather instruction; the hierarchical search (no box
    // cache in LDS) has the room for two -- a block with 9..16 hits per wave makes one trip, not two, and the median block of a
    // late pass on the 10 M-point model has 8.3
    constexpr int HB = HIER ? 16 : 8;
    constexpr int STAGE_OFF = OVL_BYTES + 128 * 4 + 16, STAGE_BYTES = NWS * HB * STG * 4;
    constexpr int MSG_OFF = STAGE_OFF + STAGE_BYTES, SEED_OFF = MSG_OFF + 64;  // message: 12 floats + cmd; seeds: 3 x 128 floats
    constexpr int MQ_OFF = SEED_OFF + 3 * 128 * 4;                             // every wave's candidate coordinates: 3 x NWS x 128
    // flat search: the chunk boxes of every wave's first PRE find passes are cached in LDS (the model does not change during
    // a resident launch; in registers they cost 16 VGPRs the kernel does not have): [wave][pass][half][lane] float4
    constexpr int PRE = 2;
    constexpr int BOXC_OFF = MQ_OFF + 3 * MQ_BYTES, BOXC_BYTES = HIER ? 0 : NWS * PRE * 2 * 64 * 16;
    constexpr int SPST_OFF = BOXC_OFF + BOXC_BYTES, SPST_BYTES = HIER ? 0 : NWS * 8 * 4;   // per wave: what its speculative list was built for
    constexpr int ROLE_OFF = SPST_OFF + SPST_BYTES + (HIER ? NWS * 64 * 4 : 0);   // (+ the level-3 hit list) then: the block's role {row, part, parts, -} and 2 x NWS wave totals
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[ROLE_OFF + (4 + 2 * NWS) * 4];
    using hit_t = typename SpHit<HIER>::type;
    hit_t* hits = reinterpret_cast<hit_t*>(lds_raw);
    // merge scratch: one (distance, index, wave) key per moving point, folded with LDS atomic mins -- the 64-bit
    // integer order is the lexicographic order the tie rule needs (d >= 0; index < 2^28; the wave id rides in the
    // low 4 bits and tells the closing wave whose coordinates to pick up)
    // (placed behind the 24 KB the cold start stages its samples in)
    unsigned long long* mkey = reinterpret_cast<unsigned long long*>(lds_raw + MKEY_OFF);
    static_assert(MKEY_OFF + 128 * 8 <= OVL_BYTES, "merge keys fit behind the sample stage");
    unsigned int* smin = reinterpret_cast<unsigned int*>(lds_raw + OVL_BYTES);
    int* hcount = reinterpret_cast<int*>(lds_raw + OVL_BYTES + 128 * 4);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- the block's role: which row of 128 moving points, and which share of the model's chunks --------------------
    // Ordinarily block (x, y) searches model segment y for row x.  SHARED ROWS (one launch per pass, a grid of more blocks
    // than rows, one segment): a cloud of 33-65 k points has more rows than the machine has CUs but fewer than it has room
    // for 8-wave blocks, and the time of a pass is set by its heaviest rows (Bunny.csv: 83 hit chunks per row on average
    // late in a registration, 1100-4200 on the heaviest).  So the spare blocks go to the rows that need them: every block
    // reads the hits each row had in the PREVIOUS launch (which row is heavy changes slowly), computes -- all blocks the
    // same numbers -- parts(row) = ceil(hits / T) with T = max(ceil(total / spare), one batch per wave), and takes the
    // role its own index falls on in the running sum.  The parts of a row interleave the model's 64-chunk tiles
    // (part p searches tiles p, p + parts, ...), fold their results into the row's 64-bit (distance, index) keys and
    // draw a ticket; the last one closes the row (the protocol of the segment blocks).  Any assignment is exact; the
    // counts only decide how even the load is.  Blocks beyond the sum have no role and end at once.
    // (the role of a shared-row block lives in LDS and is read where it is needed: the kernel has no scalar registers to
    // carry it through the pass loop)
    int* role = reinterpret_cast<int*>(lds_raw + ROLE_OFF);
#define SP_SHARED (!HIER && TAIL != 0 && fuse.share_prev != nullptr)
#define SP_ORDERED (HIER && TAIL != 0 && fuse.row_order != nullptr)   // (compiled into the hierarchical search only: the flat kernels have no register to spare)
#define SP_ROW ((SP_SHARED || SP_ORDERED) ? __builtin_amdgcn_readfirstlane(role[0]) : (int)blockIdx.x)
#define SP_PART ((SP_SHARED || SP_ORDERED) ? __builtin_amdgcn_readfirstlane(role[1]) : (int)blockIdx.y)
#define SP_PARTS ((SP_SHARED || SP_ORDERED) ? __builtin_amdgcn_readfirstlane(role[2]) : (int)gridDim.y)
    if constexpr (!HIER && TAIL != 0) {
        if (fuse.share_prev != nullptr) {
            const int R = fuse.share_rows, t = threadIdx.x;
            // (the counts were made by agent-scope atomics of the previous launch, at the memory side: they are read the same
            // way -- a plain load may be served by a stale line of this XCD's L2, and blocks that disagree about the counts
            // disagree about the roles)
            // (clamped: the sums below stay within 32 bits whatever the counters hold, and with them the guarantee that the parts
            // fit the grid -- a wrapped total once dealt more roles than there were blocks: rows without all their parts never close)
            unsigned int h = t < R ? __hip_atomic_load(&fuse.share_prev[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            h = share_clamp(h);
            if (blockIdx.x == 0 && t < R) {
                __hip_atomic_store(&fuse.share_next[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fuse.share_zero2 != nullptr) __hip_atomic_store(&fuse.share_zero2[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned int hs = h;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) hs += (unsigned int)__shfl_xor((int)hs, off, 64);
            if (lane == 0) role[4 + w] = (int)hs;
            if (t == 0) role[0] = -1;
            __syncthreads();
            unsigned int total = 0;
#pragma unroll
            for (int k = 0; k < NWS; ++k) total += (unsigned int)role[4 + k];
            const unsigned int spare = gridDim.x > (unsigned)R ? gridDim.x - (unsigned)R : 0u;
            // (the arithmetic of the assignment is in icp_kernels.h, share_*: the host computes the same for the tests)
            const unsigned int T0 = share_first_target(total, spare);
            const unsigned int cap = share_cap(m_pad);
            auto parts_for = [&](unsigned int T) { return t < R ? share_parts(h, T, cap) : 0u; };
            const unsigned int Tmin = (unsigned int)fuse.share_min;
            unsigned int T = T0 < Tmin ? Tmin : T0;
            if (share_tries_candidates(T0, Tmin, spare)) {
                // four tighter targets at once: their block counts are added up in one reduction (two 16-bit counts a word:
                // <= 512 rows x 32 parts)
                unsigned int c01 = parts_for(share_candidate(T0, 0)) | (parts_for(share_candidate(T0, 1)) << 16);
                unsigned int c23 = parts_for(share_candidate(T0, 2)) | (parts_for(share_candidate(T0, 3)) << 16);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    c01 += (unsigned int)__shfl_xor((int)c01, off, 64);
                    c23 += (unsigned int)__shfl_xor((int)c23, off, 64);
                }
                __syncthreads();   // (the wave totals of the hits have been read)
                if (lane == 0) { role[4 + w] = (int)c01; role[4 + NWS + w] = (int)c23; }
                __syncthreads();
                unsigned int s01 = 0, s23 = 0;
#pragma unroll
                for (int k = 0; k < NWS; ++k) { s01 += (unsigned int)role[4 + k]; s23 += (unsigned int)role[4 + NWS + k]; }
                __syncthreads();   // (... and these: the running sums below use the same words)
                const unsigned int sums[4] = {s01 & 0xffffu, s01 >> 16, s23 & 0xffffu, s23 >> 16};
                T = share_pick(T0, Tmin, sums, gridDim.x);
            }
            unsigned int S = parts_for(T);
            int v = (int)S;   // inclusive running sum within the wave, then across the waves
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o, 64); v += lane >= o ? u : 0; }
            if (lane == 63) role[4 + NWS + w] = v;
            __syncthreads();
            int base = 0;
#pragma unroll
            for (int k = 0; k < NWS; ++k) base += k < w ? role[4 + NWS + k] : 0;
            int all = 0;   // every role of the launch; belt and braces: if they do not fit the grid, every row is one block
#pragma unroll
            for (int k = 0; k < NWS; ++k) all += role[4 + NWS + k];
            const bool fits = all <= (int)gridDim.x;
            const int excl = fits ? base + v - (int)S : t;
            if (!fits) S = t < R ? 1u : 0u;
            if (t < R && (int)blockIdx.x >= excl && (int)blockIdx.x < excl + (int)S) { role[0] = t; role[1] = (int)blockIdx.x - excl; role[2] = (int)S; if constexpr (DIAG) { role[3] = (int)h; role[4] = (int)T; } }
            __syncthreads();
            if (role[0] < 0) return;
            if (t == 0) role[5] = 0;   // (wave 0's note to itself, resident launches: "the row's last pass was closed elsewhere")
        }
    }
    if constexpr (HIER && TAIL != 0) {
        // ordered rows (many more rows than the machine holds blocks): the blocks take the rows heaviest first, and the
        // heaviest of all are split over several blocks (launch_row_order deals the roles; NN_ORDER_*, icp_kernels.h)
        if (fuse.row_order != nullptr) {
            const int ro = fuse.row_order[blockIdx.x];   // (one word for the whole block)
            if (ro < 0) return;                          // a spare block the split rows did not need
            if (threadIdx.x == 0) {
                role[0] = ro & ((1 << NN_ROLE_ROW_BITS) - 1);
                role[1] = (ro >> NN_ROLE_ROW_BITS) & ((1 << NN_ROLE_PART_BITS) - 1);
                role[2] = 1 << ((ro >> (NN_ROLE_ROW_BITS + NN_ROLE_PART_BITS)) & 7);
            }
            __syncthreads();
        }
    }
    unsigned int hsum = 0;   // (thread 0, ordered rows: the hits of this block's lists)
    const int ibase = SP_ROW * 128 + lane;   // the block's slots; the moving point in slot s is p_perm[s] (spatially sorted groups)
    int pi[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) pi[t] = fuse.p_perm ? fuse.p_perm[ibase + t * 64] : ibase + t * 64;
    float* stage = reinterpret_cast<float*>(lds_raw + STAGE_OFF) + w * (HB * STG);  // per wave: HB hits x {box 8, x 8, y 8, z 8, model index 8}
    float* msg = reinterpret_cast<float*>(lds_raw + MSG_OFF);
    float (*seedq)[128] = reinterpret_cast<float (*)[128]>(lds_raw + SEED_OFF);
    float (*mq)[NWS][128] = reinterpret_cast<float (*)[NWS][128]>(lds_raw + MQ_OFF);
    int phase_pass_ = 0;  // (phase log)
    constexpr int phase_nw_ = NWS;
    constexpr bool phase_diag_ = DIAG;
    ICP_PHASE(0)
    const int q0 = SP_SHARED ? 0 : (int)blockIdx.y * seg_len;
    const int c_lo = q0 / 8, c_hi = min(q0 + seg_len, m_pad) / 8;
    // issued first, with everything else that does not depend on the points:
    // the seed gather does not depend on the points (the compiler cannot move these loads above the
    // stores to P_out itself), and when the seeds are the correspondences the fused transform came from -- the
    // ordinary loop -- the same gathered q serves the error of that pass
    bool real[2], sok[2];
    float sq[2][3];
    const bool from_slots = fuse.slot_state != nullptr && fuse.slot_valid != 0;   // (the same values, without the chain of gathers)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = fresh(pi[t]);
        real[t] = i < fuse.n;
        sok[t] = false;
        sq[t][0] = sq[t][1] = sq[t][2] = 0.f;
        if (from_slots) {
            const float* ss = fuse.slot_state + 3 * (size_t)n_pad + (ibase + t * 64);
            sq[t][0] = ss[0]; sq[t][1] = ss[(size_t)n_pad]; sq[t][2] = ss[2 * (size_t)n_pad];
            sok[t] = real[t];
        } else {
            // no previous match (cold start): the model point at the same RELATIVE index -- consecutive scans of one
            // sensor, or a cloud and its moved copy, keep their order, and any valid index is a valid bound
            int j = !real[t] ? -1 : fuse.seed_idx ? fuse.seed_idx[i] : (int)(((long long)i * fuse.m) / fuse.n);
            sok[t] = (unsigned)j < (unsigned)fuse.m;  // a seed is trusted only if it is a real model index
            j = sok[t] ? j : 0;
            const float* Qg = fuse.Q_gather;
            sq[t][0] = Qg[j]; sq[t][1] = Qg[(size_t)m_pad + j]; sq[t][2] = Qg[2 * (size_t)m_pad + j];
        }
    }
    f2 px, py, pz;
    if (from_slots) {
        const float* ss = fuse.slot_state + (fuse.slot_flip ? 6 * (size_t)n_pad : 0) + ibase;
        px = f2{ss[0], ss[64]};
        py = f2{ss[(size_t)n_pad], ss[(size_t)n_pad + 64]};
        pz = f2{ss[2 * (size_t)n_pad], ss[2 * (size_t)n_pad + 64]};
    } else {
        px = f2{P[pi[0]], P[pi[1]]};
        py = f2{P[(size_t)n_pad + pi[0]], P[(size_t)n_pad + pi[1]]};
        pz = f2{P[2 * (size_t)n_pad + pi[0]], P[2 * (size_t)n_pad + pi[1]]};
    }
    // ---- the pass loop: one turn for an ordinary launch, one per ICP pass for a resident one -------------------
    // Armed launch: the kernel was enqueued while the previous pass was still running, so the launch and dispatch
    // latencies are behind it; what it lacks is the (R, t) the host is solving for.  Resident launch: the same,
    // carried through -- the blocks stay on the machine for the whole registration (cooperative launch: they are
    // all resident), keep their points in registers and their seeds in LDS, and every pass is one message from the
    // host: no launch, no dispatch, no kernel boundary between two passes.
    // Wave 0 of every block waits for the message (see below), the other waves sleep at the barrier.  The poll budget
    // (a few seconds) is the exit every wave reaches if the host never answers.
    // the chunk boxes of the wave's first find passes: fetched once, at kernel entry (with everything else that does not
    // depend on the points or on the message)
    float4 pb0 = float4{0.f, 0.f, 0.f, 0.f}, pb1 = pb0;   // hierarchical search: the wave's level-3 box, kept in registers
    float4 (*boxc)[2][64] = reinterpret_cast<float4 (*)[2][64]>(lds_raw + BOXC_OFF) + w * PRE;   // flat search: [pass][half][lane]
    if constexpr (HIER) {
        // (the upper levels follow the chunk boxes in the same array; the search starts at level 3: one pass per wave)
        const int n2_all = ((m_pad >> 3) + 63) >> 6;
        const int t_lo = c_lo >> 12, t_hi = ((((c_hi + 63) >> 6)) + 63) >> 6;
        const int tidx = t_lo + w * 64 + lane;
        const float4* bp = reinterpret_cast<const float4*>(fuse.boxes + (size_t)m_pad + ((size_t)n2_all + (size_t)(tidx < t_hi ? tidx : t_lo)) * 8);
        pb0 = bp[0];
        pb1 = bp[1];
    } else {
        static_assert(PRE == 2, "two passes are fetched together");
        const int tparts = SP_SHARED ? SP_PARTS : 1, tpart = SP_SHARED ? SP_PART : 0;   // (shared rows: the parts interleave the chunks, see find_round)
        const int ca = c_lo + (w * 64 + lane) * tparts + tpart, cb = ca + NWS * 64 * tparts;
        const float4* bpa = reinterpret_cast<const float4*>(fuse.boxes + (size_t)(ca < c_hi ? ca : 0) * 8);
        const float4* bpb = reinterpret_cast<const float4*>(fuse.boxes + (size_t)(cb < c_hi ? cb : 0) * 8);
        const float4 a0 = bpa[0], a1 = bpa[1], b0 = bpb[0], b1 = bpb[1];
        // (read back by this wave only: DS operations of a wave stay in order)
        boxc[0][0][lane] = a0; boxc[0][1][lane] = a1;
        boxc[1][0][lane] = b0; boxc[1][1][lane] = b1;
    }
    // (work-counting instantiation only) what this wave executes -- wave-uniform tallies, flushed once per pass
    unsigned int wk_find = 0, wk_upper = 0, wk_hit[3] = {0, 0, 0}, wk_samp = 0;
    // Speculative hit list of a resident launch (prepared at the end of a pass, see there): the group box and the bound
    // it was built for, and its length.  Wave-uniform, and the same in every wave of the block.
    // (kept in LDS, a private slot per wave {lo.xyz, B, hi.xyz, -}: the kernel has no registers to spare across the wait)
    bool spec_valid = false;
    float* spst = reinterpret_cast<float*>(lds_raw + SPST_OFF) + w * 8;
    for (int pass = 0;; ++pass) {
    phase_pass_ = pass;
    double err_row = 0.0;
    RT<float> rt = rt_arg;
    int cmd = fuse.apply ? ICP_CMD_TRANSFORM_MATCH : ICP_CMD_MATCH;
    double row_tag = tail.tag;
    unsigned int row_tag_lo = tail.tag_lo;
    const bool have_seeds = pass > 0 || fuse.seed_idx != nullptr;
    if (w == 0) {  // (wave 0 alone: in a resident launch it may still be reading last pass's keys when the others get here)
        smin[lane] = 0x7f800000u; smin[lane + 64] = 0x7f800000u;
        mkey[lane] = ~0ull; mkey[lane + 64] = ~0ull;
    }
    if (threadIdx.x == 0) {
        // (flat search: from the second pass of a resident launch on, the counter is looked after at the end of the pass
        // before -- it may hold the length of a speculative list)
        if (HIER || pass == 0) *hcount = 0;
        if constexpr (HIER) { hcount[1] = 0; hcount[2] = 0; }
    }
    if (fuse.mailbox != nullptr) {
        const double want = fuse.want + (double)pass;
        if (w == 0) {
            // One load fetches the whole line (lane l reads word l & 15): the message is there when both halves carry
            // the awaited tag, and then it has been received as well -- no second trip for the payload.
            // Where the mailbox is host memory, block 0 alone talks to the host (one reader: ~1.3 us each way; 128
            // readers would queue up to ~20 us, tools/mailbox_probe.hip) and relays the line through device memory,
            // where the other blocks wait for it with agent-scope loads.  (no relay: the mailbox itself is device
            // memory the host writes through the BAR, every block polls it)
            const bool first = (blockIdx.x == 0 && blockIdx.y == 0) || fuse.relay == nullptr;
            const uint32_t* src = (first ? fuse.mailbox : fuse.relay)->w + (lane & 15);
            const uint32_t want32 = (fuse.want_lo + (uint32_t)pass) | 0x80000000u;
            uint32_t word = 0u;
            bool ok = false;
            // the wait is bounded in wall-clock time (the device's constant 100 MHz counter, read every 64th poll): the
            // same budget whether the poll goes to host memory, to BAR-visible device memory or to the relay.  A block that
            // listens to the relay waits twice as long -- block 0 decides and publishes its verdict there.
            const long long give_up = (long long)wall_clock64() + (first ? ICP_MAILBOX_BUDGET_TICKS : 2 * ICP_MAILBOX_BUDGET_TICKS);
            for (unsigned int spins = 1;; ++spins) {
                word = first ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                             : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (uint32_t)__builtin_amdgcn_readlane((int)word, ICP_MB_TAG0) == want32 &&
                     (uint32_t)__builtin_amdgcn_readlane((int)word, ICP_MB_TAG1) == want32;
                if (ok) break;
                if ((spins & 63u) == 0u && (long long)wall_clock64() > give_up) break;
                __builtin_amdgcn_s_sleep(2);
            }
            // (a time-out reads as a withdrawal: the other blocks must end too)
            if (!ok) word = (lane & 15) == ICP_MB_CMD ? (uint32_t)ICP_CMD_EXIT : ((lane & 15) == ICP_MB_TAG0 || (lane & 15) == ICP_MB_TAG1) ? want32 : 0u;
            if (first && fuse.relay != nullptr && lane < 16)
                __hip_atomic_store(&fuse.relay->w[lane], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one line, one store
            if (lane < 16) reinterpret_cast<uint32_t*>(msg)[lane] = word;
            if constexpr (!HIER && TAIL != 0) {
                // shared rows, resident: the last pass of this row was closed by another of its blocks -- the matches are in the
                // row's publication (complete before the row's tag left, so before this message was written)
                if (pass > 0 && fuse.share_prev != nullptr && role[5] != 0) {
                    const float* pub = fuse.seed_pub + (size_t)SP_ROW * 384;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        seedq[a][lane] = __hip_atomic_load(&pub[a * 128 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        seedq[a][lane + 64] = __hip_atomic_load(&pub[a * 128 + lane + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        __syncthreads();
        cmd = reinterpret_cast<const int*>(msg)[ICP_MB_CMD];
        if (cmd == ICP_CMD_EXIT) return;  // withdrawn (the loop stopped) or timed out: nothing more is touched
#pragma unroll
        for (int k = 0; k < 9; ++k) rt.r[k] = msg[mailbox_rt_word(k)];
#pragma unroll
        for (int k = 0; k < 3; ++k) rt.t[k] = msg[mailbox_rt_word(9 + k)];
        row_tag = want;
        row_tag_lo = fuse.want_lo + (unsigned int)pass;
        if (pass > 0) {
            // the seeds of a resident pass are the matches of the one before: wave 0 left their coordinates in LDS
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                sok[t] = real[t];
                sq[t][0] = seedq[0][lane + t * 64]; sq[t][1] = seedq[1][lane + t * 64]; sq[t][2] = seedq[2][lane + t * 64];
            }
        }
    } else {
        __syncthreads();  // the list counter and the exchange minima are reset
    }
    const bool apply = cmd != ICP_CMD_MATCH;
    if (apply) {
        // every wave re-derives the moved points in registers (same instructions => same bits); wave 0 of the
        // grid.y == 0 block stores them and accounts the error of the pass that produced (R, t)
        double err = 0.0;
        const bool shared_gather = pass > 0 || from_slots || (fuse.seed_idx != nullptr && fuse.idx_prev == fuse.seed_idx);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float x = t ? px.y : px.x, y = t ? py.y : py.x, z = t ? pz.y : pz.x;
            apply_rt<float>(rt, x, y, z, x, y, z);
            if (t) { px.y = x; py.y = y; pz.y = z; } else { px.x = x; py.x = y; pz.x = z; }
            if (w == 0 && SP_PART == 0) {
                const int i = fresh(pi[t]);
                fuse.P_out[i] = x;
                fuse.P_out[(size_t)n_pad + i] = y;
                fuse.P_out[2 * (size_t)n_pad + i] = z;
                if (fuse.slot_state != nullptr) {   // (and in slot order, for the next pass's front end)
                    float* ss = fuse.slot_state + (fuse.slot_flip ? 0 : 6 * (size_t)n_pad) + (fresh(SP_ROW * 128) + lane + t * 64);   // (recomputed: no register held for it)
                    ss[0] = x; ss[(size_t)n_pad] = y; ss[2 * (size_t)n_pad] = z;
                }
                if (i < fuse.n) {
                    float qx = sq[t][0], qy = sq[t][1], qz = sq[t][2];
                    if (!(shared_gather && sok[t])) {
                        const int j = fuse.idx_prev[i];
                        const float* Qg = fuse.Q_gather;
                        qx = Qg[j]; qy = Qg[(size_t)m_pad + j]; qz = Qg[2 * (size_t)m_pad + j];
                    }
                    const double ex = (double)qx - (double)x, ey = (double)qy - (double)y, ez = (double)qz - (double)z;
                    err += ex * ex + ey * ey + ez * ez;
                }
            }
        }
        if (w == 0 && SP_PART == 0) {
            err_row = wave_sum(err);
            if (lane == 0) {
                if constexpr (TAIL != 0) {
                    // read by whichever block closes this row: agent-scope store, drained before our ticket
                    if (SP_PARTS > 1) __hip_atomic_store(&tail.err_tile[SP_ROW], err_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    fuse.err_rows[SP_ROW] = err_row;
                }
            }
        }
    }
    if (!apply && pass == 0 && fuse.store_first && w == 0 && SP_PART == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = fresh(pi[t]);
            fuse.P_out[i] = t ? px.y : px.x;
            fuse.P_out[(size_t)n_pad + i] = t ? py.y : py.x;
            fuse.P_out[2 * (size_t)n_pad + i] = t ? pz.y : pz.x;
        }
    }
    ICP_PHASE(1)
    if (cmd == ICP_CMD_TRANSFORM_ONLY) {
        const int row_ = SP_ROW;
        // the loop's last pass: nothing is matched any more, the row carries the error alone
        if constexpr (TAIL != 0) {
            if (w == 0 && SP_PART == 0) {
                if (TAIL == 1 && tail.compact != 0) {
                    double* row = tail.rows + (size_t)row_ * NN_CROW;
                    if (lane >= 1 && lane < NN_CROW) __hip_atomic_store(&row[lane], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&row[0], crow_pack(err_row, row_tag_lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    double* row = tail.rows + (size_t)row_ * ICP_NMOM;
                    if (lane < ICP_NMOM - 1) row[lane] = lane == ICP_MOM_ERR ? err_row : 0.0;
                    __threadfence_system();
                    if (lane == 0) __hip_atomic_store(&row[ICP_NMOM - 1], row_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        return;
    }
    float best[2];
    float bq[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // coordinates of the running minimum
    int bj[2];  // index of the running minimum; -1: this wave has not lowered the bound it started from
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bj[t] = -1;
        best[t] = inf_<float>();
        {
            // seeded bound: the distance to ANY model point (last pass's match) bumped by one ulp -- the true minimum
            // is <= that distance < bound, so the seed changes how much work is skipped, never the answer
            const float x = t ? px.y : px.x, y = t ? py.y : py.x, z = t ? pz.y : pz.x;
            const float d = dist2<float>(x, y, z, sq[t][0], sq[t][1], sq[t][2]);
            best[t] = (sok[t] && d < inf_<float>()) ? __uint_as_float(__float_as_uint(d) + 1u) : inf_<float>();
        }
        // padding lanes never ask for a chunk (their result, "nothing found", is never read)
        best[t] = real[t] ? best[t] : -1.f;
    }
    // Seeded, but poorly (hierarchical search): early in a registration of two clouds that are far apart the previous match is a
    // loose bound -- the 10 M-point pair, pass 2: the matches lie 0.23 away, the true neighbours 0.16, and what a row lists is the
    // model within sqrt(bound) of its box; it grows with (bound - true distance^2).  A sample of the thinned-out model that lies
    // within a sample spacing of the foot point brings the bound to within spacing^2 of the truth, for the price of ~16 hits per
    // wave.  So a seeded pass whose largest bound is well above the sample spacing squared takes the cold start's sample round
    // first.  The bound is the same in every wave (they hold the same points): the same decision everywhere.
    bool resample = false;
    if constexpr (HIER) {
        if (have_seeds && fuse.samples != nullptr && fuse.resample_bound > 0.f)
            resample = wave_minmax<true>(__builtin_fmaxf(best[0], best[1])) > fuse.resample_bound;
    }
    if ((!have_seeds || resample) && fuse.samples != nullptr) {
        // Cold start: no previous match to seed the bounds, so the block measures its points against a thinned-out
        // model first -- one point per chunk, at most 2048 of them, staged in LDS (over the hit list and the merge
        // scratch, both idle until later), a share per wave -- and every wave starts from the block-wide minimum
        // bumped by an ulp.  Any model point gives a valid bound; the scan below is then as selective as a seeded one.
        // A round measures the points against up to 2048 samples (a 10 M-point model: 7.4 -> 6.5 ms for the cold pass
        // with 2048 instead of 512).  On a small model that many samples cost as much as they save unless the
        // relative-index seeds are poor (hall scan against its slightly moved self: 13.8 -> 12.0 us with 64 samples;
        // a 128 x 128 grid against a copy 0.8 away: 73 -> 53 us with 2048), so a probe round of 8 groups decides.
        static_assert(3 * SMAX * 4 <= MKEY_OFF, "the staged samples overlay the hit list and merge scratch");
        const int ns8 = ((m_pad / 8) + 7) / 8;                 // groups of 8 samples in the array
        const int ns_pad = ns8 * 8;
        float* sl = reinterpret_cast<float*>(lds_raw);         // [3][SMAX]
        const int gfull = min(max(fuse.sample_groups, 1), SMAX / 8);
        constexpr int GPROBE = 8;
        // Small models (the full round would cost a good part of the pass itself): probe first -- did 64 samples cut
        // the bound of an eighth of the block's points to a quarter?  then the seeds were poor and the full round
        // follows.  Larger models: 64 samples are too coarse to tell, and the full round is cheap next to the pass.
        int gcap = (m_pad <= 32768 && gfull > GPROBE) ? GPROBE : gfull;
        for (;;) {   // (one body for both rounds: inlined twice it spilled registers in the sorted-view variant)
            // one round over <= gcap groups spread evenly over the model
            const int gs = (ns8 + gcap - 1) / gcap;            // group stride: <= gcap groups are staged
            int ng = (ns8 + gs - 1) / gs;
            if constexpr (HIER) {
                // a large model: the round's samples one by one, evenly spread (a group of 8 CONSECUTIVE samples is eight
                // neighbouring chunks -- 256 places, not 2048; with single samples the nearest one lies within
                // ~0.4 x sqrt(area / 2048) of a point's foot on the model, which is what makes the round worth taking
                // in a seeded pass too, see `resample`)
                // (the hierarchy's size class has >= 2^13 chunks; forced onto a small model -- the tests do -- the samples repeat)
                const int ns = m_pad >> 3, cnt = min(gcap * 8, max(8, ns & ~7)), stride = max(1, ns / cnt);
                ng = cnt >> 3;
                for (int a = 0; a < 3; ++a)
                    for (int i = threadIdx.x; i < cnt; i += NWS * 64) sl[a * SMAX + i] = fuse.samples[(size_t)a * ns_pad + (size_t)i * stride];
            } else
            for (int v = threadIdx.x; v < ng * 6; v += NWS * 64) {
                const int gp = v / 6, r = v % 6, a = r >> 1, hh = r & 1;
                *reinterpret_cast<float4*>(sl + a * SMAX + gp * 8 + hh * 4) =
                    *reinterpret_cast<const float4*>(fuse.samples + (size_t)a * ns_pad + (size_t)gp * gs * 8 + hh * 4);
            }
            __syncthreads();
            float sb[2] = {inf_<float>(), inf_<float>()};
            for (int gp = w; gp < ng; gp += NWS) {
                const float4* a = reinterpret_cast<const float4*>(sl + gp * 8);
                const float4* b = reinterpret_cast<const float4*>(sl + SMAX + gp * 8);
                const float4* c = reinterpret_cast<const float4*>(sl + 2 * SMAX + gp * 8);
                scan8_min(a[0], a[1], b[0], b[1], c[0], c[1], px, py, pz, sb);
                if constexpr (DIAG) ++wk_samp;
            }
            if (real[0]) atomicMin(&smin[lane], __float_as_uint(sb[0]));
            if (real[1]) atomicMin(&smin[lane + 64], __float_as_uint(sb[1]));
            __syncthreads();  // (also: the staging area is free again)
            int helped = 0;   // points whose bound fell to a quarter or less
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned int v = smin[lane + t * 64];
                const bool better = real[t] && v < 0x7f800000u && __uint_as_float(v + 1u) < best[t];
                const bool much = better && !(__uint_as_float(v + 1u) >= 0.25f * best[t]);
                if (better) best[t] = __uint_as_float(v + 1u);
                helped += (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(much));
            }
            // (every wave holds the same points and reads the same minima: the same count, the same decision everywhere)
            if (gcap == gfull || helped < 16) break;
            gcap = gfull;
        }
    }
    ICP_PHASE(2)

    // bounding box of the block's 128 moving points (every wave derives the same one) -- of its REAL points: the padding of a
    // partly filled last row never asks for a chunk, and wherever it lies it must not widen the box the chunks are listed by
    // (the 10 M-point share of one rank of 8 ends in such a row: 122 544 chunks on its list, 2.1 ms of a 3.0 ms pass, alone)
    const float binf = inf_<float>();
    float glo[3] = {__builtin_fminf(real[0] ? px.x : binf, real[1] ? px.y : binf), __builtin_fminf(real[0] ? py.x : binf, real[1] ? py.y : binf),
                    __builtin_fminf(real[0] ? pz.x : binf, real[1] ? pz.y : binf)};
    float ghi[3] = {__builtin_fmaxf(real[0] ? px.x : -binf, real[1] ? px.y : -binf), __builtin_fmaxf(real[0] ? py.x : -binf, real[1] ? py.y : -binf),
                    __builtin_fmaxf(real[0] ? pz.x : -binf, real[1] ? pz.y : -binf)};
    wave_box(glo, ghi);

    const int round_tiles = NWS * round_passes;                                         // 64-chunk tiles a round of the find covers
    // ... and the tiles this block searches (shared rows: every parts-th one)
    const int own_tiles = SP_SHARED ? (((c_hi - c_lo - SP_PART + SP_PARTS - 1) / SP_PARTS) + 63) >> 6 : ((c_hi - c_lo + 63) >> 6);
    float B0_pass = -1.f;   // (flat search) the largest starting bound of this pass
    // one find pass: the lane tests chunk cidx (box b0 = lo.xyz hi.x, b1 = hi.yz - -) and appends it to the hit list
    auto find_pass = [&](const int cidx, const float4 b0, const float4 b1, float B, const float (&gl)[3], const float (&gh)[3]) {
        const float gx = __builtin_fmaxf(__builtin_fmaxf(b0.x - gh[0], gl[0] - b0.w), 0.f);
        const float gy = __builtin_fmaxf(__builtin_fmaxf(b0.y - gh[1], gl[1] - b1.x), 0.f);
        const float gz = __builtin_fmaxf(__builtin_fmaxf(b0.z - gh[2], gl[2] - b1.y), 0.f);
        const float L = ((gx * gx + gy * gy) + gz * gz) * 0.99999905f;
        const bool pass = cidx < c_hi && L < B;  // every candidate winner lies strictly below its point's starting bound
        if constexpr (DIAG) wk_find += (unsigned int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(cidx < c_hi));
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
        if (mask != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(hcount, (int)__builtin_popcountll(mask));
            base = __builtin_amdgcn_readfirstlane(base);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (pass) hits[base + rank] = (hit_t)cidx;
        }
    };
    // The hits are dealt round-robin; a wave fetches the box and the coordinates of up to 8 of its hits with ONE
    // gather -- 8 lanes x 16 bytes per hit -- into its private LDS stage, so a batch of hits costs one trip to
    // memory instead of three or four each.
    // (exchanging minima between the batches of a cold pass was measured too: the barriers cost more than they save)
    auto gather_batch = [&](const int hb, const int h1) {
            {
                const int part = lane & 7;
                // (the loads of a trip are issued together; named values, not an array: the compiler moves private arrays to LDS)
                const int r0 = lane >> 3, h0 = hb + r0 * NWS + w, h8 = hb + (r0 + 8) * NWS + w;
                const bool on0 = h0 < h1, on8 = HB > 8 && h8 < h1;
                const int ch0 = on0 ? (int)hits[h0] : 0, ch8 = on8 ? (int)hits[h8] : 0;
                // (a large model -- the hierarchical search: the hit's record, 160 contiguous bytes in the stage's own layout, two cache
                // lines instead of 32-byte pieces of five arrays)
                const float* src0 = HIER ? fuse.records + (size_t)ch0 * NN_REC_WORDS + part * 4
                                         : part < 2 ? fuse.boxes + (size_t)ch0 * 8 + part * 4 : Q + (size_t)((part - 2) >> 1) * m_pad + (size_t)ch0 * 8 + (part & 1) * 4;
                const float* src8 = HIER ? fuse.records + (size_t)ch8 * NN_REC_WORDS + part * 4
                                         : part < 2 ? fuse.boxes + (size_t)ch8 * 8 + part * 4 : Q + (size_t)((part - 2) >> 1) * m_pad + (size_t)ch8 * 8 + (part & 1) * 4;
                float4 v0 = float4{0.f, 0.f, 0.f, 0.f}, v8 = v0;
                if (on0) v0 = *reinterpret_cast<const float4*>(src0);
                if constexpr (HB > 8) { if (on8) v8 = *reinterpret_cast<const float4*>(src8); }
                if (on0) *reinterpret_cast<float4*>(stage + r0 * STG + part * 4) = v0;
                if constexpr (HB > 8) { if (on8) *reinterpret_cast<float4*>(stage + (r0 + 8) * STG + part * 4) = v8; }
                // a sorted view: the elements' model indices (the sort permutation) are staged too
                if constexpr (PERM) {
                    const int r2 = lane >> 1, half = lane & 1;
                    const int h2 = hb + r2 * NWS + w;
                    if (lane < 2 * HB && h2 < h1)
                        *reinterpret_cast<int4*>(stage + r2 * STG + 32 + half * 4) =
                            HIER ? *reinterpret_cast<const int4*>(fuse.records + (size_t)(int)hits[h2] * NN_REC_WORDS + 32 + half * 4)
                                 : *reinterpret_cast<const int4*>(fuse.q_perm + (size_t)(int)hits[h2] * 8 + half * 4);
                }
            }
            lds_same_wave_order();
    };
    auto scan_batch = [&](const int hb, const int h1) {
            const int mine = (h1 - hb - w + NWS - 1) / NWS;     // this wave's hits in the batch
            const int cnt = mine < HB ? mine : HB;
            // A list of several batches per wave (a pair that is far apart: thousands of hits per row): the hits are dealt
            // round-robin, so the chunk that holds a point's nearest neighbour is worked on by ONE wave, and until the round's
            // exchange the other fifteen go on testing their hits against a bound that no longer holds.  So the waves leave the
            // minima they have reached in smin[] as they go (LDS atomic min, no barrier) and pick up what the others have left:
            // every value ever stored there is a distance some wave has MEASURED for that point, i.e. a valid bound whenever it
            // is read; taken over bumped by an ulp with "no candidate of my own", exactly as at the round's exchange.
            const bool share_minima = h1 > NWS * HB;
            if (share_minima && hb != 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned int v = smin[lane + t * 64];
                    if (real[t] && v < 0x7f800000u && v < __float_as_uint(best[t])) { best[t] = __uint_as_float(v + 1u); bj[t] = -1; }
                }
            }
            for (int rr = 0; rr < cnt; ++rr) {
                int stage_reached;
                if constexpr (PERM) {
                    stage_reached = scan_hit<true>(stage + rr * STG, 0, px, py, pz, best, bj, bq);
                } else {
                    const int ch = __builtin_amdgcn_readfirstlane((int)hits[hb + rr * NWS + w]);
                    stage_reached = scan_hit<false>(stage + rr * STG, ch, px, py, pz, best, bj, bq);
                }
                if constexpr (DIAG) {
                    ++wk_hit[0];
                    wk_hit[1] += stage_reached >= 1 ? 1u : 0u;
                    wk_hit[2] += stage_reached >= 2 ? 1u : 0u;
                }
            }
            if (share_minima && hb + NWS * HB < h1) {   // (the last batch: the round's exchange, or the merge, follows)
                if (real[0] && bj[0] >= 0) atomicMin(&smin[lane], __float_as_uint(best[0]));
                if (real[1] && bj[1] >= 0) atomicMin(&smin[lane + 64], __float_as_uint(best[1]));
            }
            lds_same_wave_order();
    };
    // (phase log, flat search: wave 2 of every block stamps "the list is complete" and "the first batch is fetched" in slots 6, 7)
    auto sub_stamp = [&](int k) {
        if constexpr (DIAG && !HIER) {
            if (fuse.tlog != nullptr && w == 2 && lane == 0 && (fuse.tlog_pass < 0 || fuse.tlog_pass == phase_pass_)) {
                const long long slot_ = (((long long)blockIdx.y * gridDim.x + blockIdx.x) * phase_nw_ + w) * 10 + k;
                if (slot_ < fuse.tlog_cap) fuse.tlog[slot_] = (long long)wall_clock64();
            }
        }
    };
    auto process_hits = [&](const int h1) {
        for (int hb = 0; hb < h1; hb += NWS * HB) {
            gather_batch(hb, h1);
            if (hb == 0) sub_stamp(7);
            scan_batch(hb, h1);
        }
    };
    // exchange before the next round: every wave goes on from the block's best minimum so far, bumped by
    // an ulp (d >= 0: the bit patterns order like the values, so this is an integer min)
    auto exchange = [&]() {
        if (real[0]) atomicMin(&smin[lane], __float_as_uint(best[0]));
        if (real[1]) atomicMin(&smin[lane + 64], __float_as_uint(best[1]));
        __syncthreads();
        if (threadIdx.x == 0) *hcount = 0;  // the list is consumed
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned int v = smin[lane + t * 64];
            if (real[t] && v < 0x7f800000u && v < __float_as_uint(best[t])) { best[t] = __uint_as_float(v + 1u); bj[t] = -1; }
        }
    };
    // one round of the find: every wave tests its share of the round's chunks against the group box (gl, gh) and the bound B
    // (tb: the round's first tile, counted in the block's OWN chunks, 64 to a tile: own chunk u is chunk c_lo + u of the segment,
    // or -- shared rows -- chunk u * parts + part of the model: the parts of a row interleave chunk by chunk, so that the hits
    // of a row, which cluster, fall to its parts evenly)
    auto find_round = [&](int tb, float B, const float (&gl)[3], const float (&gh)[3]) {
        const int tparts = SP_SHARED ? SP_PARTS : 1, tpart = SP_SHARED ? SP_PART : 0;
        int r = 0;
        if (tb == 0) {  // the first round's first passes use the boxes fetched at kernel entry
#pragma unroll
            for (; r < PRE; ++r) {
                const int cidx = c_lo + ((r * NWS + w) * 64 + lane) * tparts + tpart;
                if constexpr (!HIER)
                    if (r < round_passes && c_lo + ((r * NWS + w) * 64) * tparts + tpart < c_hi) find_pass(cidx, boxc[r][0][lane], boxc[r][1][lane], B, gl, gh);
            }
        }
        // (the boxes of four passes are requested together: one memory latency for all of them)
        for (; r < round_passes; r += 4) {
            int cidx[4];
            float4 b0[4], b1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cidx[q] = c_lo + ((tb + (r + q) * NWS + w) * 64 + lane) * tparts + tpart;
                const bool on = r + q < round_passes && cidx[q] < c_hi;
                const float4* bp = reinterpret_cast<const float4*>(fuse.boxes + (size_t)(on ? cidx[q] : c_lo) * 8);
                b0[q] = bp[0]; b1[q] = bp[1];
                cidx[q] = on ? cidx[q] : c_hi;   // (a pass that is not this round's lists nothing)
            }
            if (__builtin_amdgcn_readfirstlane(cidx[0]) >= c_hi) break;   // (lane 0 holds the pass's first chunk)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (__builtin_amdgcn_readfirstlane(cidx[q]) < c_hi) find_pass(cidx[q], b0[q], b1[q], B, gl, gh);
        }
    };
    if constexpr (HIER) {
        // A hierarchy of boxes, 64 to 1: chunks (8 model points) < super boxes (512 points) < level-3 boxes (32 768
        // points).  Every level is tested like the chunks of the flat search -- one box per lane against the group
        // box and the largest bound -- and only the children of the survivors are looked at: a survivor costs a wave ONE
        // pass over its 64 children.  A box contains its children and every operation of the test is monotonic, so
        // a chunk that passes its own test has ancestors that pass too: the hit list is the one the flat search
        // builds, the model is just not read where it cannot matter (10 M-point model: 40 MB of chunk boxes per
        // block -> 10 KB of level-3 boxes + the children of a few survivors).
        constexpr int SCAP = 2 * NWS * 64;   // super-box hit list: the children of 32 level-3 boxes
        constexpr int TCAP = NWS * 64;       // level-3 hit list = level-3 boxes per outermost round (33 M model points)
        static_assert(SCAP * 4 <= MKEY_OFF - HITS_BYTES, "the super-box hit list lies between the chunk hit list and the merge keys");
        int* shits = reinterpret_cast<int*>(lds_raw + HITS_BYTES);
        int* thits = reinterpret_cast<int*>(lds_raw + SPST_OFF + SPST_BYTES);
        int* scount = hcount + 1;
        int* tcount = hcount + 2;
        const int n2_all = ((m_pad >> 3) + 63) >> 6;
        const float* sboxes = fuse.boxes + (size_t)m_pad;         // (the chunk boxes take m_pad floats)
        const float* tboxes = sboxes + (size_t)n2_all * 8;
        const int s_lo = c_lo >> 6, s_hi = (c_hi + 63) >> 6;     // a segment starts on a super-box boundary (nn_plan)
        const int t_lo = s_lo >> 6, t_hi = (s_hi + 63) >> 6;
        // a split row: this block takes the super boxes whose number is its part modulo the parts (2, 4, .. 64 of them) --
        // the same lanes of every level-3 box's children
        const bool my_super = SP_ORDERED ? (lane & (SP_PARTS - 1)) == SP_PART : true;
        // one box per lane against the group box: true where the box may hold a winner
        auto near_box = [&](const float4 b0, const float4 b1, float B) {
            const float gx = __builtin_fmaxf(__builtin_fmaxf(b0.x - ghi[0], glo[0] - b0.w), 0.f);
            const float gy = __builtin_fmaxf(__builtin_fmaxf(b0.y - ghi[1], glo[1] - b1.x), 0.f);
            const float gz = __builtin_fmaxf(__builtin_fmaxf(b0.z - ghi[2], glo[2] - b1.y), 0.f);
            return ((gx * gx + gy * gy) + gz * gz) * 0.99999905f < B;
        };
        auto append = [&](bool pass, int value, int* list, int* count) {
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
            if (mask != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(count, (int)__builtin_popcountll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (pass) list[base + rank] = value;
            }
        };
        bool list_dirty = false;   // a round has been processed: the chunk list needs the barrier after its reset
        // (phase log: where the search spends its time -- ticks in the upper levels, the chunk find and the hit
        // processing, barriers included, and the two hit totals; waves other than 0 leave them in slots 6..9)
        long long dg_t[3] = {0, 0, 0}, dg_mark = 0;
        int dg_sh = 0, dg_h = 0;
        auto dg_lap = [&](int k) { if constexpr (DIAG) { const long long now = (long long)wall_clock64(); dg_t[k] += now - dg_mark; dg_mark = now; } };
        if constexpr (DIAG) dg_mark = (long long)wall_clock64();
        for (int tb = t_lo; tb < t_hi; tb += TCAP) {
            {   // level 3: one pass per wave (the first round's boxes were fetched at kernel entry)
                const float Bt = wave_minmax<true>(__builtin_fmaxf(best[0], best[1]));
                const int tidx = tb + w * 64 + lane;
                float4 b0 = pb0, b1 = pb1;
                if (tb != t_lo) {
                    const float4* bp = reinterpret_cast<const float4*>(tboxes + (size_t)(tidx < t_hi ? tidx : t_lo) * 8);
                    b0 = bp[0]; b1 = bp[1];
                }
                append(tidx < t_hi && near_box(b0, b1, Bt), tidx, thits, tcount);
                if constexpr (DIAG) wk_upper += (unsigned int)max(0, min(64, t_hi - (tb + w * 64)));
            }
            __syncthreads();   // the level-3 list is complete
            const int TH = *tcount;
            for (int tg = 0; tg < TH; tg += 2 * NWS) {
                // level 2: the children of up to 32 level-3 survivors, two per wave
                const float Bs = wave_minmax<true>(__builtin_fmaxf(best[0], best[1]));
                const int tend = min(tg + 2 * NWS, TH);
                for (int k = tg + w; k < tend; k += NWS) {
                    const int sidx = (thits[k] << 6) + lane;
                    const bool in = sidx >= s_lo && sidx < s_hi && my_super;
                    const float4* bp = reinterpret_cast<const float4*>(sboxes + (size_t)(in ? sidx : s_lo) * 8);
                    append(in && near_box(bp[0], bp[1], Bs), sidx, shits, scount);
                    if constexpr (DIAG) wk_upper += 64u;
                }
                __syncthreads();   // the super list is complete (and, after a processed round, the chunk list's reset is seen)
                const int SH = *scount;
                if (tb == t_lo && tg == 0 && fuse.samples != nullptr && fuse.refine_min > 0 && SH >= fuse.refine_min) {
                    // REFINEMENT ROUND (a pass whose bounds are loose lists many super boxes): before any chunk is listed, the row's
                    // points are measured against the chunk samples of the super boxes just listed -- every k-th of them, at most
                    // refine_cnt -- staged over the (still empty) chunk hit list.  Any model point gives a valid bound; these lie
                    // where the row's neighbours are, a few point spacings apart, so the rounds below start from near-final bounds
                    // instead of reaching them hit by hit.
                    constexpr int RSTRIDE = 1024;                      // (12 KB: below the super-box list in every block size)
                    const int ns_pad_r = (((m_pad / 8) + 7) / 8) * 8, ns_r = m_pad >> 3;
                    const int total = SH * 64, kstep = (total + fuse.refine_cnt - 1) / fuse.refine_cnt;
                    const int cnt_r = (total / kstep) & ~7, ng_r = cnt_r >> 3;
                    float* sl = reinterpret_cast<float*>(lds_raw);
                    for (int i = threadIdx.x; i < cnt_r; i += NWS * 64) {
                        const int e = i * kstep;
                        int ci = (shits[e >> 6] << 6) + (e & 63);
                        ci = ci < ns_r ? ci : ns_r - 1;
#pragma unroll
                        for (int a = 0; a < 3; ++a) sl[a * RSTRIDE + i] = fuse.samples[(size_t)a * ns_pad_r + ci];
                    }
                    __syncthreads();
                    float sbr[2] = {inf_<float>(), inf_<float>()};
                    for (int gp = w; gp < ng_r; gp += NWS) {
                        const float4* a = reinterpret_cast<const float4*>(sl + gp * 8);
                        const float4* b = reinterpret_cast<const float4*>(sl + RSTRIDE + gp * 8);
                        const float4* c = reinterpret_cast<const float4*>(sl + 2 * RSTRIDE + gp * 8);
                        scan8_min(a[0], a[1], b[0], b[1], c[0], c[1], px, py, pz, sbr);
                        if constexpr (DIAG) ++wk_samp;
                    }
                    if (real[0]) atomicMin(&smin[lane], __float_as_uint(sbr[0]));
                    if (real[1]) atomicMin(&smin[lane + 64], __float_as_uint(sbr[1]));
                    __syncthreads();   // (also: the staging area is the hit list again)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned int v = smin[lane + q * 64];
                        if (real[q] && v < 0x7f800000u && __uint_as_float(v + 1u) < best[q]) { best[q] = __uint_as_float(v + 1u); bj[q] = -1; }
                    }
                }
                dg_lap(0);
                dg_sh += SH;
                const bool more_above = tend < TH || tb + TCAP < t_hi;
                const int rs_ = fuse.round_supers;
                for (int sh0 = 0; sh0 < SH; sh0 += rs_) {
                    // one round: the chunks of up to 64 super boxes (<= SP_HCAP hits)
                    const float B = wave_minmax<true>(__builtin_fmaxf(best[0], best[1]));
                    if (list_dirty && sh0 != 0) __syncthreads();  // (sh0 == 0: the barrier above)
                    const int send = min(sh0 + rs_, SH);
                    for (int k = sh0 + w; k < send; k += NWS) {
                        const int c0 = shits[k] << 6;
                        const float4* bp = reinterpret_cast<const float4*>(fuse.boxes + (size_t)(c0 + lane < c_hi ? c0 + lane : c_lo) * 8);
                        find_pass(c0 + lane, bp[0], bp[1], B, glo, ghi);
                    }
                    __syncthreads();
                    dg_lap(1);
                    dg_h += *hcount;
                    if (SP_ORDERED && threadIdx.x == 0) hsum += (unsigned int)*hcount;
                    process_hits(*hcount);
                    if (send < SH || more_above) { exchange(); list_dirty = true; }
                    dg_lap(2);
                }
                if (more_above) {
                    __syncthreads();   // everybody has read the super list
                    if (threadIdx.x == 0) *scount = 0;
                    __syncthreads();
                }
            }
            if (tb + TCAP < t_hi) {
                __syncthreads();
                if (threadIdx.x == 0) *tcount = 0;
                __syncthreads();
            }
        }
        if constexpr (DIAG) {
            if (fuse.tlog != nullptr && lane == 0 && w != 0 && (fuse.tlog_pass < 0 || fuse.tlog_pass == phase_pass_)) {
                const long long slot_ = (((long long)blockIdx.y * gridDim.x + blockIdx.x) * phase_nw_ + w) * 10;
                if (slot_ + 9 < fuse.tlog_cap) {
                    fuse.tlog[slot_ + 6] = dg_t[0]; fuse.tlog[slot_ + 7] = dg_t[1]; fuse.tlog[slot_ + 8] = dg_t[2];
                    fuse.tlog[slot_ + 9] = ((long long)dg_sh << 32) | (long long)dg_h;
                    // wave 2: where the block ran (XCC_ID << 32 | HW_ID) -- tools/cu_gaps.py builds per-CU timelines from it
                    if (w == 2) fuse.tlog[slot_ + 6] = ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
                }
            }
        }
    } else {
    const float B0 = wave_minmax<true>(__builtin_fmaxf(best[0], best[1]));   // the largest starting bound of the block's points
    B0_pass = B0;
    bool searched = false;
    if (spec_valid) {
        // A hit list for this pass was prepared -- found AND fetched into the waves' stages -- while the block waited for
        // the message, for a group box and a bound that were guessed (end of the pass loop).  It holds every chunk the
        // find below would list if the guesses cover the real ones: the box test is monotonic in both, operation by
        // operation (a wider group box gives smaller gaps, a larger bound passes more).  Extra chunks cost time, never
        // the answer: every hit still goes through the exact per-point tests.
        const float4 s0 = *reinterpret_cast<const float4*>(spst), s1 = *reinterpret_cast<const float4*>(spst + 4);   // (broadcast reads)
        const bool covered = glo[0] >= s0.x && glo[1] >= s0.y && glo[2] >= s0.z &&
                             ghi[0] <= s1.x && ghi[1] <= s1.y && ghi[2] <= s1.z && B0 <= s0.w;
        if constexpr (DIAG) {
            if (fuse.work != nullptr && threadIdx.x == 0) {
                atomicAdd(&fuse.work[NN_WORK_SPEC_LISTS], 1ull);
                if (covered) { atomicAdd(&fuse.work[NN_WORK_SPEC_COVERED], 1ull); atomicAdd(&fuse.work[NN_WORK_SPEC_HITS], (unsigned long long)*hcount); }
            }
        }
        if (covered) {
            const int h1 = *hcount;   // (the list's length is still in the counter)
            scan_batch(0, h1);        // the first batch sits in the stages
            for (int hb = NWS * HB; hb < h1; hb += NWS * HB) { gather_batch(hb, h1); scan_batch(hb, h1); }
            searched = true;
        } else {
            if (threadIdx.x == 0) *hcount = 0;   // the guess did not hold: forget the list and search as usual
            __syncthreads();
        }
    }
    if (!searched) {
    for (int tb = 0; tb < own_tiles; tb += round_tiles) {
        // B only shrinks while the block works: refreshed once per round
        const float B = tb == 0 ? B0 : wave_minmax<true>(__builtin_fmaxf(best[0], best[1]));
        if (tb != 0) __syncthreads();  // the list is empty and its counter reset (first round: the barrier above)
        find_round(tb, B, glo, ghi);
        __syncthreads();
        if (tb == 0) sub_stamp(6);
        if (SP_SHARED && threadIdx.x == 0) {   // what the next launch shares the rows by
            __hip_atomic_fetch_add(&fuse.share_cur[SP_ROW], (unsigned int)*hcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fuse.share_cur2 != nullptr) __hip_atomic_fetch_add(&fuse.share_cur2[SP_ROW], (unsigned int)*hcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (DIAG) {
            // (phase log, flat search: wave 1 of every block leaves its role and the hits of its list in slots 6 and 7 -- tools/share_report.py)
            if (fuse.tlog != nullptr && w == 1 && lane == 0 && (fuse.tlog_pass < 0 || fuse.tlog_pass == phase_pass_)) {
                const long long slot_ = (((long long)blockIdx.y * gridDim.x + blockIdx.x) * phase_nw_ + w) * 10;
                if (slot_ + 9 < fuse.tlog_cap) {
                    fuse.tlog[slot_ + 6] = ((long long)SP_ROW << 32) | ((long long)SP_PART << 16) | (long long)SP_PARTS;
                    fuse.tlog[slot_ + 7] = (tb == 0 ? 0ll : fuse.tlog[slot_ + 7]) + (long long)*hcount;
                    if (SP_SHARED) fuse.tlog[slot_ + 8] = ((long long)role[4] << 32) | (long long)(unsigned int)role[3];   // (the target per block, the row's hits last time)
                }
            }
        }
        if constexpr (DIAG) { if (fuse.work != nullptr && threadIdx.x == 0) atomicAdd(&fuse.work[NN_WORK_LIST_HITS], (unsigned long long)*hcount); }
        process_hits(*hcount);
        if (tb + round_tiles < own_tiles) exchange();
    }
    }
    spec_valid = false;
    }
    ICP_PHASE(3)
    if constexpr (DIAG) {
        if (fuse.work != nullptr && lane == 0) {
            // slots: NN_WORK_* (icp_kernels.h)
            if (wk_find) atomicAdd(&fuse.work[NN_WORK_FIND_BOXES], (unsigned long long)wk_find);
            if (wk_upper) atomicAdd(&fuse.work[NN_WORK_UPPER_BOXES], (unsigned long long)wk_upper);
            if (wk_hit[0]) atomicAdd(&fuse.work[NN_WORK_HITS_BOX], (unsigned long long)wk_hit[0]);
            if (wk_hit[1]) atomicAdd(&fuse.work[NN_WORK_HITS_XY], (unsigned long long)wk_hit[1]);
            if (wk_hit[2]) atomicAdd(&fuse.work[NN_WORK_HITS_FULL], (unsigned long long)wk_hit[2]);
            if (wk_samp) atomicAdd(&fuse.work[NN_WORK_SAMPLE_GROUPS], (unsigned long long)wk_samp);
            if (w == 0) atomicAdd(&fuse.work[NN_WORK_BLOCK_PASSES], 1ull);
            if (w == 0 && apply) atomicAdd(&fuse.work[NN_WORK_BLOCK_TRANSFORMS], 1ull);
        }
        wk_find = wk_upper = wk_samp = 0; wk_hit[0] = wk_hit[1] = wk_hit[2] = 0;
    }

    if (SP_ORDERED && threadIdx.x == 0 && hsum != 0u) __hip_atomic_fetch_add(&fuse.row_hits[SP_ROW], hsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // in-block merge: every wave that lowered its bound folds its candidate into the point's key
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (bj[t] >= 0) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(best[t]) << 32) | ((unsigned int)bj[t] << 4) | (unsigned int)w;
            atomicMin(&mkey[lane + t * 64], key);
            mq[0][w][lane + t * 64] = bq[t][0]; mq[1][w][lane + t * 64] = bq[t][1]; mq[2][w][lane + t * 64] = bq[t][2];
        }
    }
    ICP_PHASE(4)
    __syncthreads();
    ICP_PHASE(5)
    // wave 0 finishes the row: it holds both of every lane's points in registers
    if (w != 0) {
        if (!fuse.resident) return;   // (resident: on to the speculative search below, then the next message)
    } else {
    float fb[2];
    int fj[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const unsigned long long key = mkey[lane + t * 64];
        const bool none = key == ~0ull;  // no wave found anything below the bound (padding lanes)
        const unsigned int lo = (unsigned int)key;
        fb[t] = none ? inf_<float>() : __uint_as_float((unsigned int)(key >> 32));
        fj[t] = none ? 0x7fffffff : (int)(lo >> 4);
        const int bw = none ? 0 : (int)(lo & 15u);
        sq[t][0] = mq[0][bw][lane + t * 64]; sq[t][1] = mq[1][bw][lane + t * 64]; sq[t][2] = mq[2][bw][lane + t * 64];
    }
    if constexpr (TAIL == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const size_t o = (size_t)blockIdx.y * n_pad + (size_t)pi[t];
            part_d[o] = fb[t];
            part_idx[o] = fj[t];
        }
        return;
    } else {
        const int parts = SP_PARTS, row = SP_ROW;
        bool closer = true;   // this block closes the row (always, unless the row is split)
        if (parts > 1) {
            // several segment blocks share the row: fold into the 64-bit (d, idx) keys and draw a ticket, the
            // last arriver closes the row (protocol as in nn_match_f32_v2; only this wave takes part)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(fb[t]) << 32) | (unsigned int)fj[t];
                __hip_atomic_fetch_min(&tail.keys[fresh(row * 128 + lane) + t * 64], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (fresh: no address held across the pass loop)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ICP_PHASE(6)
            unsigned int ticket = 0;
            if (lane == 0) ticket = __hip_atomic_fetch_add(&tail.tickets[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            ICP_PHASE(7)
            if (ticket != (unsigned int)(parts - 1)) {
                if (!fuse.resident) return;
                closer = false;   // (resident: on to the wait; the row's matches will be fetched from its publication)
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int i = fresh(ibase) + t * 64;  // keys live per slot
                    const unsigned long long key = __hip_atomic_load(&tail.keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // ready for the next pass / launch (nobody touches this row again in this one; agent scope: in a resident
                    // launch the next pass's atomics follow without a kernel boundary)
                    __hip_atomic_store(&tail.keys[i], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    fj[t] = (int)(unsigned int)(key & 0xffffffffull);
                }
                if (lane == 0) __hip_atomic_store(&tail.tickets[row], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fuse.apply) err_row = __hip_atomic_load(&tail.err_tile[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if constexpr (!HIER) { if (SP_SHARED && fuse.resident && lane == 0) role[5] = closer ? 0 : 1; }
        if (closer) {
#pragma unroll
        for (int t = 0; t < 2; ++t) fj[t] = ((unsigned)fj[t] < (unsigned)fuse.m) ? fj[t] : fuse.m - 1;  // unreachable clamp
        bool gather_in_tail = parts > 1;
        if constexpr (!HIER) {
            if (parts > 1 && fuse.resident && SP_SHARED) {
                // resident, split row: the coordinates of the matches (some were won by other blocks) are gathered here and
                // published for the row's other blocks BEFORE the row's tag leaves (the tail drains its stores first)
                float* pub = fuse.seed_pub + (size_t)row * 384;
                const float* Qg = fuse.Q_gather;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (fresh(pi[t]) < fuse.n) { sq[t][0] = Qg[fj[t]]; sq[t][1] = Qg[(size_t)m_pad + fj[t]]; sq[t][2] = Qg[2 * (size_t)m_pad + fj[t]]; }
#pragma unroll
                    for (int a = 0; a < 3; ++a) __hip_atomic_store(&pub[a * 128 + lane + t * 64], sq[t][a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                gather_in_tail = false;
            }
        }
        NNTail tl = tail;
        tl.tag = row_tag;
        tl.tag_lo = row_tag_lo;
        tl.idx_out = (pass & 1) ? tail.idx_out_odd : tail.idx_out;
        // (a row closed over several segment blocks may have been won elsewhere: its coordinates are gathered)
        tl.row = row;
        tl.idx_through = (SP_SHARED && fuse.resident) ? 1 : 0;
        if (parts == 1) { ICP_PHASE(6) }
        tail_close_row<TAIL, DIAG, NWS, false, true>(px, py, pz, fj, lane, pi, m_pad, fuse, tl, apply ? err_row : 0.0, lds_raw, sq, gather_in_tail, pass);
        ICP_PHASE(9)
        if (fuse.slot_state != nullptr) {   // the matched model points, in slot order, for the next pass (one segment; a shared row: gathered by the tail)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float* ss = fuse.slot_state + 3 * (size_t)n_pad + (fresh(SP_ROW * 128) + lane + t * 64);
                ss[0] = sq[t][0]; ss[(size_t)n_pad] = sq[t][1]; ss[2 * (size_t)n_pad] = sq[t][2];
            }
        }
        if (!fuse.resident) return;
        // the matches of this pass seed the next one and are what its error is measured against
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            sok[t] = real[t];
            seedq[0][lane + t * 64] = sq[t][0]; seedq[1][lane + t * 64] = sq[t][1]; seedq[2][lane + t * 64] = sq[t][2];
        }
        }  // (closer)
    }
    }  // (wave 0)

    // ---- resident launch: the wait for the next message is put to use -----------------------------------------------------
    // The rows are on their way to the host, which will add them up, solve and answer: 3-4 us during which the block
    // used to sleep, after which it searched the chunk boxes (find), fetched the hit chunks (one trip to memory) and only
    // then got to the arithmetic.  Instead the block now GUESSES where the next transform will put its points -- the group
    // box of this pass widened by twice the displacement this pass's transform caused (plus a thousandth of the box), and
    // a bound that grows with it by the triangle inequality -- and runs the find and the fetch for that guess right away.
    // When the message arrives the real group box and bound are compared with the guess (seven comparisons); if they are
    // covered, the list is a superset of the real one (see where it is used) and the pass goes straight to the exact
    // per-point tests on chunks that already sit in LDS.  If not -- or if the list outgrew one batch -- the pass searches
    // as before; nothing but idle time was spent.
    if constexpr (!HIER) {
        if (threadIdx.x == 0) *hcount = 0;   // (this pass's list is consumed; ordered before its next use by the barriers below / the message barrier)
        const bool single_round = own_tiles <= round_tiles;
        if (fuse.speculate && single_round && apply && B0_pass >= 0.f && B0_pass < inf_<float>()) {
            // displacement of the block's points under this pass's transform, per axis, bounded over their (new) group box:
            // p_old = R^T (p_new - t)  =>  p_new - p_old = (I - R^T) p_new + R^T t
            float dl[3], dn2 = 0.f, sp_lo[3], sp_hi[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float acc = 0.f, rt_t = 0.f;
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float m = (a == b ? 1.f : 0.f) - rt.r[b * 3 + a];           // (I - R^T)_ab
                    acc += __builtin_fabsf(m) * __builtin_fmaxf(__builtin_fabsf(glo[b]), __builtin_fabsf(ghi[b]));
                    rt_t += rt.r[b * 3 + a] * rt.t[b];                                // (R^T t)_a
                }
                dl[a] = fuse.spec_gain * (acc + __builtin_fabsf(rt_t)) + fuse.spec_floor * (ghi[a] - glo[a]) + 1e-6f;
                dn2 += dl[a] * dl[a];
                sp_lo[a] = glo[a] - dl[a];
                sp_hi[a] = ghi[a] + dl[a];
            }
            // every point's next starting bound is its distance to this pass's match after the move:
            // sqrt(d_new) <= sqrt(d_old) + |displacement|, and d_old < this pass's largest starting bound
            const float rB = __builtin_amdgcn_sqrtf(B0_pass) + __builtin_amdgcn_sqrtf(dn2);
            const float sp_B = rB * rB * 1.0001f;
            if (lane == 0) {
                *reinterpret_cast<float4*>(spst) = float4{sp_lo[0], sp_lo[1], sp_lo[2], sp_B};
                *reinterpret_cast<float4*>(spst + 4) = float4{sp_hi[0], sp_hi[1], sp_hi[2], 0.f};
            }
            __syncthreads();   // wave 0 is through with the row (its transpose buffer overlays the hit list); the counter is reset
            find_round(0, sp_B, sp_lo, sp_hi);
            __syncthreads();
            const int spec_n = *hcount;
            if (spec_n <= 4 * NWS * HB) {   // (up to four batches: the first is fetched now, the others when the list is used)
                gather_batch(0, spec_n);
                spec_valid = true;
            } else {
                __syncthreads();   // (everybody has read the count)
                if (threadIdx.x == 0) *hcount = 0;
            }
        }
    }
    }  // pass loop
}
#undef SP_SHARED
#undef SP_ORDERED
#undef SP_ROW
#undef SP_PART
#undef SP_PARTS

// ------------------------------------------------------------------------------------------------
// matching, fp32, sparse, 64-point rows -- the shipped kernel for clouds that cannot fill the machine with 128-point rows
// (up to 32 768 moving points: the hall scan, Bunny_res, the synthetic grids; the model searched flat, < 2^17 points).
//
// nn_match_sparse gives a hall-sized cloud 128 blocks: half the CUs idle, and on the other half 16 waves per CU, each of
// which transforms the block's 128 points, derives their bounds and group box and then works through its share of the
// hit chunks two moving points per lane.  The counters say the occupied VALUs are the limit during that phase (4 waves
// per SIMD x ~670 instructions a pass).  This kernel is the same search laid out for latency:
//   * a block is 8 waves holding the same 64 moving points, ONE per lane: 256 blocks for the hall scan -- every CU;
//   * the packed arithmetic runs over two MODEL points instead (v_pk_* on {q[2k], q[2k+1]} against the lane's point in
//     both halves): a hit chunk costs ~75 instructions instead of ~200, each half still an ordinary IEEE operation of
//     the reference's (dx*dx + dy*dy) + dz*dz;
//   * the group box of 64 points is tighter: fewer hits per block, fewer wasted pairs per hit;
//   * two waves per SIMD instead of four: the per-wave overhead (transform, bounds, group box, find) is paid 8 times
//     per 64 points where it was paid 16 times per 128, and a wave's dependent chain is half as long.
// Everything else is nn_match_sparse: seeds and ulp-bumped bounds, lane-parallel find over the chunk boxes, unordered hit
// list with the explicit (distance, model index) tie rule, LDS key merge, the row tail by wave 0, the mailbox protocol of
// armed and resident launches, the speculative search during the wait.  One segment only (gridDim.y == 1).
// ------------------------------------------------------------------------------------------------
constexpr int R64_NW = 8;                                   // waves per block

// one hit chunk (LDS stage {box 8, x 8, y 8, z 8 (, model index 8)}) against the lane's ONE point, two model points per
// packed operation; same tie rule as scan_hit.  The per-point box test is done by the caller for all of a wave's hits at
// once (their loads overlap); a hit that passes is evaluated in full here: with one or two waves per SIMD the early-out
// between the xy half and the z half saved less arithmetic than its ballot, its branch and the LDS round trip behind
// it cost (24 % of the hits that got there stopped there).
template <bool PERM>
__device__ __forceinline__ void scan_hit1(const float* sb, int ch, const f2 px, const f2 py, const f2 pz /* the point in both halves */,
                                          float& best, int& bj, float (&bq)[3])
{
    constexpr int C = 8;
    const float *qxp = sb + 8, *qyp = sb + 16, *qzp = sb + 24;
    // (all six loads are issued before the first use: one LDS latency per hit)
    const float4 qxa = *reinterpret_cast<const float4*>(qxp), qxb = *reinterpret_cast<const float4*>(qxp + 4);
    const float4 qya = *reinterpret_cast<const float4*>(qyp), qyb = *reinterpret_cast<const float4*>(qyp + 4);
    const float4 qza = *reinterpret_cast<const float4*>(qzp), qzb = *reinterpret_cast<const float4*>(qzp + 4);
    f2 d[C / 2];   // d[k] = {model point 2k, model point 2k+1}
    {
        f2 ax, ay, az;
        ax = f2{qxa.x, qxa.y} - px; ay = f2{qya.x, qya.y} - py; az = f2{qza.x, qza.y} - pz; d[0] = (ax * ax + ay * ay) + az * az;
        ax = f2{qxa.z, qxa.w} - px; ay = f2{qya.z, qya.w} - py; az = f2{qza.z, qza.w} - pz; d[1] = (ax * ax + ay * ay) + az * az;
        ax = f2{qxb.x, qxb.y} - px; ay = f2{qyb.x, qyb.y} - py; az = f2{qzb.x, qzb.y} - pz; d[2] = (ax * ax + ay * ay) + az * az;
        ax = f2{qxb.z, qxb.w} - px; ay = f2{qyb.z, qyb.w} - py; az = f2{qzb.z, qzb.w} - pz; d[3] = (ax * ax + ay * ay) + az * az;
    }
    float c0 = fmin_(fmin_(d[0].x, d[0].y), fmin_(d[1].x, d[1].y));   // the chunk's own minimum
    c0 = fmin_(c0, fmin_(fmin_(d[2].x, d[2].y), fmin_(d[3].x, d[3].y)));
    auto dk = [&](int k) { return (k & 1) ? d[k >> 1].y : d[k >> 1].x; };
    if constexpr (PERM) {
        const bool cand = c0 <= best;
        if (__builtin_amdgcn_ballot_w64(cand) != 0ull) {
            // lowest model index among the chunk elements at the chunk's minimum, and where it sits
            const int* qo = reinterpret_cast<const int*>(sb + 32);
            int o0 = 0x7fffffff, k0 = 0;
#pragma unroll
            for (int kk = C - 1; kk >= 0; --kk) {
                const int oj = qo[kk];
                const bool e0 = (dk(kk) == c0) & (oj < o0);
                o0 = e0 ? oj : o0; k0 = e0 ? kk : k0;
            }
            const bool take = cand & ((c0 < best) | (bj < 0) | (o0 < bj));   // bj < 0: nothing to tie with yet
            best = take ? c0 : best;
            bj = take ? o0 : bj;
            if (take) { bq[0] = qxp[k0]; bq[1] = qyp[k0]; bq[2] = qzp[k0]; }
        }
    } else {
        // identity order: chunks are disjoint index ranges, "lower model index" is "lower chunk, then lower k"
        const bool take = (c0 < best) | ((c0 == best) & (ch < (bj >> 3)));   // bj = -1: nothing to tie with
        if (__builtin_amdgcn_ballot_w64(take) != 0ull) {
            int k0 = C - 1;
#pragma unroll
            for (int kk = C - 2; kk >= 0; --kk) k0 = (dk(kk) == c0) ? kk : k0;
            best = take ? c0 : best;
            bj = take ? ch * C + k0 : bj;
            if (take) { bq[0] = qxp[k0]; bq[1] = qyp[k0]; bq[2] = qzp[k0]; }
        }
    }
}

// distances from the lane's point to 8 model points, folded into a running minimum (no index)
__device__ __forceinline__ void scan8_min1(const float4 qx0, const float4 qx1, const float4 qy0, const float4 qy1, const float4 qz0,
                                           const float4 qz1, const f2 px, const f2 py, const f2 pz, float& best)
{
    const f2 qx[4] = {f2{qx0.x, qx0.y}, f2{qx0.z, qx0.w}, f2{qx1.x, qx1.y}, f2{qx1.z, qx1.w}};
    const f2 qy[4] = {f2{qy0.x, qy0.y}, f2{qy0.z, qy0.w}, f2{qy1.x, qy1.y}, f2{qy1.z, qy1.w}};
    const f2 qz[4] = {f2{qz0.x, qz0.y}, f2{qz0.z, qz0.w}, f2{qz1.x, qz1.y}, f2{qz1.z, qz1.w}};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f2 ax = qx[k] - px, ay = qy[k] - py, az = qz[k] - pz;
        const f2 dd = (ax * ax + ay * ay) + az * az;
        best = fmin_(fmin_(best, dd.x), dd.y);
    }
}

// (Two blocks fit a CU: 512 rows = 32 768 points can stay on the machine for a whole registration.  Tried: a third
// instantiation squeezed to 80 VGPRs, three blocks per CU, so that Bunny.csv's 576 rows stay resident -- 47.3 us per
// iteration against 45.7 us with rows of 128 and one armed launch per pass: that cloud's passes are decided by a few
// hit-heavy blocks, which 8 waves work through more slowly than 16.  Not kept.)
// NW: waves per block.  8 is the shipped geometry (two blocks fit a CU).  16 (ICP_NN_WAVES=16, only while every block can have
// a CU of its own) halves the hits per wave once more, but such a block fills its CU: nothing else that must be resident --
// another context's registration, another rank rehearsed on the same device -- fits beside it.
template <int TAIL, bool DIAG, bool PERM, int NW = R64_NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 1) void nn_match_row64(const float* __restrict__ P, int n_pad, const float* __restrict__ Q,
                                                                 int m_pad, int round_passes, float* __restrict__ part_d,
                                                                 int32_t* __restrict__ part_idx, RT<float> rt_arg, NNFuse fuse, NNTail tail)
{
    constexpr int STG = PERM ? 40 : 32;  // floats per staged hit: box 8, x 8, y 8, z 8 (, model indices 8)
    constexpr int SMAX = 2048;           // cold start: samples staged per round
    constexpr int HITS_BYTES = SP_HCAP * 4, SAMPLE_BYTES = 3 * SMAX * 4;
    constexpr int TR_BYTES = TAIL ? ((TAIL == 2 ? 28 : 18) * 65 + 64) * 8 : 0;
    static_assert(TR_BYTES <= HITS_BYTES, "the tail's transpose buffer overlays the hit list");
    static_assert(HITS_BYTES <= SAMPLE_BYTES, "the staged samples overlay the hit list");
    // [0, 24 KB): hit list / tail transpose / cold-start samples (never live together); then the small arrays
    constexpr int MKEY_OFF = SAMPLE_BYTES, SMIN_OFF = MKEY_OFF + 64 * 8, HCNT_OFF = SMIN_OFF + 64 * 4;
    constexpr int STAGE_OFF = HCNT_OFF + 16, STAGE_BYTES = NW * 8 * STG * 4;
    constexpr int MSG_OFF = STAGE_OFF + STAGE_BYTES, SEED_OFF = MSG_OFF + 64;
    constexpr int MQ_OFF = SEED_OFF + 3 * 64 * 4, SPST_OFF = MQ_OFF + 3 * NW * 64 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SPST_OFF + NW * 8 * 4];
    int* hits = reinterpret_cast<int*>(lds_raw);
    unsigned long long* mkey = reinterpret_cast<unsigned long long*>(lds_raw + MKEY_OFF);   // (distance, index, wave) per point
    unsigned int* smin = reinterpret_cast<unsigned int*>(lds_raw + SMIN_OFF);
    int* hcount = reinterpret_cast<int*>(lds_raw + HCNT_OFF);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ibase = blockIdx.x * 64 + lane;   // the block's slots; the moving point in slot s is p_perm[s]
    const int pi = fuse.p_perm ? fuse.p_perm[ibase] : ibase;
    float* stage = reinterpret_cast<float*>(lds_raw + STAGE_OFF) + w * (8 * STG);
    float* msg = reinterpret_cast<float*>(lds_raw + MSG_OFF);
    float (*seedq)[64] = reinterpret_cast<float (*)[64]>(lds_raw + SEED_OFF);
    float (*mq)[NW][64] = reinterpret_cast<float (*)[NW][64]>(lds_raw + MQ_OFF);
    float* spst = reinterpret_cast<float*>(lds_raw + SPST_OFF) + w * 8;
    int phase_pass_ = 0;  // (phase log)
    constexpr int phase_nw_ = NW;
    constexpr bool phase_diag_ = DIAG;
    ICP_PHASE(0)
    const int c_lo = 0, c_hi = m_pad / 8;
    const bool real = pi < fuse.n;
    bool sok = false;
    float sq[3] = {0.f, 0.f, 0.f};
    {
        // the seed: last pass's match, or (cold start) the model point at the same RELATIVE index -- any valid index is a valid bound
        int j = !real ? -1 : fuse.seed_idx ? fuse.seed_idx[fresh(pi)] : (int)(((long long)pi * fuse.m) / fuse.n);
        sok = (unsigned)j < (unsigned)fuse.m;
        j = sok ? j : 0;
        const float* Qg = fuse.Q_gather;
        sq[0] = Qg[j]; sq[1] = Qg[(size_t)m_pad + j]; sq[2] = Qg[2 * (size_t)m_pad + j];
    }
    float x = P[pi], y = P[(size_t)n_pad + pi], z = P[2 * (size_t)n_pad + pi];
    unsigned int wk_find = 0, wk_hit[3] = {0, 0, 0}, wk_samp = 0;   // (work-counting instantiation only)
    bool spec_valid = false;   // a speculative hit list for the coming pass sits in LDS (see the end of the pass loop)
    for (int pass = 0;; ++pass) {
    phase_pass_ = pass;
    double err_row = 0.0;
    RT<float> rt = rt_arg;
    int cmd = fuse.apply ? ICP_CMD_TRANSFORM_MATCH : ICP_CMD_MATCH;
    double row_tag = tail.tag;
    unsigned int row_tag_lo = tail.tag_lo;
    const bool have_seeds = pass > 0 || fuse.seed_idx != nullptr;
    if (w == 0) { smin[lane] = 0x7f800000u; mkey[lane] = ~0ull; }
    if (threadIdx.x == 0 && pass == 0) *hcount = 0;   // (later passes: looked after at the end of the pass before)
    if (fuse.mailbox != nullptr) {
        const double want = fuse.want + (double)pass;
        if (w == 0) {
            // (the protocol of nn_match_sparse: one load fetches the line, both tags must be the awaited one; block 0 relays
            // a host-memory mailbox through device memory; the wait is bounded in wall-clock time)
            const bool first = blockIdx.x == 0 || fuse.relay == nullptr;
            const uint32_t* src = (first ? fuse.mailbox : fuse.relay)->w + (lane & 15);
            const uint32_t want32 = (fuse.want_lo + (uint32_t)pass) | 0x80000000u;
            uint32_t word = 0u;
            bool ok = false;
            const long long give_up = (long long)wall_clock64() + (first ? ICP_MAILBOX_BUDGET_TICKS : 2 * ICP_MAILBOX_BUDGET_TICKS);
            for (unsigned int spins = 1;; ++spins) {
                word = first ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                             : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (uint32_t)__builtin_amdgcn_readlane((int)word, ICP_MB_TAG0) == want32 &&
                     (uint32_t)__builtin_amdgcn_readlane((int)word, ICP_MB_TAG1) == want32;
                if (ok) break;
                if ((spins & 63u) == 0u && (long long)wall_clock64() > give_up) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) word = (lane & 15) == ICP_MB_CMD ? (uint32_t)ICP_CMD_EXIT : ((lane & 15) == ICP_MB_TAG0 || (lane & 15) == ICP_MB_TAG1) ? want32 : 0u;
            if (first && fuse.relay != nullptr && lane < 16)
                __hip_atomic_store(&fuse.relay->w[lane], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane < 16) reinterpret_cast<uint32_t*>(msg)[lane] = word;
        }
        __syncthreads();
        cmd = reinterpret_cast<const int*>(msg)[ICP_MB_CMD];
        if (cmd == ICP_CMD_EXIT) return;  // withdrawn (the loop stopped) or timed out: nothing more is touched
#pragma unroll
        for (int k = 0; k < 9; ++k) rt.r[k] = msg[mailbox_rt_word(k)];
#pragma unroll
        for (int k = 0; k < 3; ++k) rt.t[k] = msg[mailbox_rt_word(9 + k)];
        row_tag = want;
        row_tag_lo = fuse.want_lo + (unsigned int)pass;
        if (pass > 0) {   // the seeds of a resident pass are the matches of the one before: wave 0 left their coordinates in LDS
            sok = real;
            sq[0] = seedq[0][lane]; sq[1] = seedq[1][lane]; sq[2] = seedq[2][lane];
        }
    } else {
        __syncthreads();  // the list counter and the exchange minima are reset
    }
    const bool apply = cmd != ICP_CMD_MATCH;
    if (apply) {
        // every wave re-derives the moved point in registers (same instructions => same bits); wave 0 stores it and accounts
        // the error of the pass that produced (R, t)
        const bool shared_gather = pass > 0 || (fuse.seed_idx != nullptr && fuse.idx_prev == fuse.seed_idx);
        apply_rt<float>(rt, x, y, z, x, y, z);
        if (w == 0) {
            const int i = fresh(pi);
            fuse.P_out[i] = x;
            fuse.P_out[(size_t)n_pad + i] = y;
            fuse.P_out[2 * (size_t)n_pad + i] = z;
            double err = 0.0;
            if (i < fuse.n) {
                float qx = sq[0], qy = sq[1], qz = sq[2];
                if (!(shared_gather && sok)) {
                    const int j = fuse.idx_prev[i];
                    const float* Qg = fuse.Q_gather;
                    qx = Qg[j]; qy = Qg[(size_t)m_pad + j]; qz = Qg[2 * (size_t)m_pad + j];
                }
                const double ex = (double)qx - (double)x, ey = (double)qy - (double)y, ez = (double)qz - (double)z;
                err = ex * ex + ey * ey + ez * ez;
            }
            err_row = wave_sum(err);
            if constexpr (TAIL == 0) { if (lane == 0) fuse.err_rows[blockIdx.x] = err_row; }
        }
    }
    if (!apply && pass == 0 && fuse.store_first && w == 0) {
        const int i = fresh(pi);
        fuse.P_out[i] = x;
        fuse.P_out[(size_t)n_pad + i] = y;
        fuse.P_out[2 * (size_t)n_pad + i] = z;
    }
    ICP_PHASE(1)
    if (cmd == ICP_CMD_TRANSFORM_ONLY) {
        // the loop's last pass: nothing is matched any more, the row carries the error alone
        if constexpr (TAIL != 0) {
            if (w == 0) {
                if (TAIL == 1 && tail.compact != 0) {
                    double* row = tail.rows + (size_t)blockIdx.x * NN_CROW;
                    if (lane >= 1 && lane < NN_CROW) __hip_atomic_store(&row[lane], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&row[0], crow_pack(err_row, row_tag_lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    double* row = tail.rows + (size_t)blockIdx.x * ICP_NMOM;
                    if (lane < ICP_NMOM - 1) row[lane] = lane == ICP_MOM_ERR ? err_row : 0.0;
                    __threadfence_system();
                    if (lane == 0) __hip_atomic_store(&row[ICP_NMOM - 1], row_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        return;
    }
    const f2 px = f2{x, x}, py = f2{y, y}, pz = f2{z, z};   // the point in both halves of the packed operations
    float best = inf_<float>();
    float bq[3] = {0.f, 0.f, 0.f};  // coordinates of the running minimum
    int bj = -1;                    // its index; -1: this wave has not lowered the bound it started from
    {
        // seeded bound: the distance to ANY model point (last pass's match) bumped by one ulp -- the true minimum is <= that
        // distance < bound, so the seed changes how much work is skipped, never the answer
        const float d = dist2<float>(x, y, z, sq[0], sq[1], sq[2]);
        best = (sok && d < inf_<float>()) ? __uint_as_float(__float_as_uint(d) + 1u) : inf_<float>();
        best = real ? best : -1.f;   // padding lanes never ask for a chunk (their result, "nothing found", is never read)
    }
    if (!have_seeds && fuse.samples != nullptr) {
        // cold start (see nn_match_sparse): the block measures its points against a thinned-out model first -- a probe
        // round of 8 groups on a small model decides whether the full round is worth it
        const int ns8 = ((m_pad / 8) + 7) / 8;                 // groups of 8 samples in the array
        const int ns_pad = ns8 * 8;
        float* sl = reinterpret_cast<float*>(lds_raw);         // [3][SMAX]
        const int gfull = min(max(fuse.sample_groups, 1), SMAX / 8);
        constexpr int GPROBE = 8;
        int gcap = (m_pad <= 32768 && gfull > GPROBE) ? GPROBE : gfull;
        for (;;) {
            const int gs = (ns8 + gcap - 1) / gcap;            // group stride: <= gcap groups are staged
            const int ng = (ns8 + gs - 1) / gs;
            for (int v = threadIdx.x; v < ng * 6; v += NW * 64) {
                const int gp = v / 6, r = v % 6, a = r >> 1, hh = r & 1;
                *reinterpret_cast<float4*>(sl + a * SMAX + gp * 8 + hh * 4) =
                    *reinterpret_cast<const float4*>(fuse.samples + (size_t)a * ns_pad + (size_t)gp * gs * 8 + hh * 4);
            }
            __syncthreads();
            float sb = inf_<float>();
            for (int gp = w; gp < ng; gp += NW) {
                const float4* a = reinterpret_cast<const float4*>(sl + gp * 8);
                const float4* b = reinterpret_cast<const float4*>(sl + SMAX + gp * 8);
                const float4* c = reinterpret_cast<const float4*>(sl + 2 * SMAX + gp * 8);
                scan8_min1(a[0], a[1], b[0], b[1], c[0], c[1], px, py, pz, sb);
                if constexpr (DIAG) ++wk_samp;
            }
            if (real) atomicMin(&smin[lane], __float_as_uint(sb));
            __syncthreads();  // (also: the staging area is free again)
            const unsigned int v = smin[lane];
            const bool better = real && v < 0x7f800000u && __uint_as_float(v + 1u) < best;
            const bool much = better && !(__uint_as_float(v + 1u) >= 0.25f * best);
            if (better) best = __uint_as_float(v + 1u);
            const int helped = (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(much));   // points whose bound fell to a quarter or less
            if (gcap == gfull || helped < 8) break;   // (the same count, the same decision in every wave)
            gcap = gfull;
        }
    }
    ICP_PHASE(2)

    const float best0 = best;   // the point's starting bound (the block's largest one goes into the next speculative bound)
    const int round_chunks = NW * 64 * round_passes;
    // one find pass: lane l tests chunk c0 + l (box b0 = lo.xyz hi.x, b1 = hi.yz - -) against the group box (gl, gh) and appends it to the hit list
    auto find_pass = [&](int c0, const float4 b0, const float4 b1, float B, const float (&gl)[3], const float (&gh)[3]) {
        const int cidx = c0 + lane;
        const float gx = __builtin_fmaxf(__builtin_fmaxf(b0.x - gh[0], gl[0] - b0.w), 0.f);
        const float gy = __builtin_fmaxf(__builtin_fmaxf(b0.y - gh[1], gl[1] - b1.x), 0.f);
        const float gz = __builtin_fmaxf(__builtin_fmaxf(b0.z - gh[2], gl[2] - b1.y), 0.f);
        const float L = ((gx * gx + gy * gy) + gz * gz) * 0.99999905f;
        const bool pass_ = cidx < c_hi && L < B;  // every candidate winner lies strictly below its point's starting bound
        if constexpr (DIAG) wk_find += (unsigned int)max(0, min(64, c_hi - c0));
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass_);
        if (mask != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(hcount, (int)__builtin_popcountll(mask));
            base = __builtin_amdgcn_readfirstlane(base);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (pass_ && base + rank < SP_HCAP) hits[base + rank] = cidx;   // (a round never lists more; a speculative list that would is dropped)
        }
    };
    // one round of the find: every wave tests its share of the round's chunks (the boxes of two passes are requested
    // together: one memory latency for both)
    auto find_round = [&](int rb, float B, const float (&gl)[3], const float (&gh)[3]) {
        for (int r = 0; r < round_passes; r += 2) {
            const int c0 = rb + (r * NW + w) * 64, c1 = c0 + NW * 64;
            if (c0 >= c_hi) break;
            const bool two = r + 1 < round_passes && c1 < c_hi;
            const float4* bp0 = reinterpret_cast<const float4*>(fuse.boxes + (size_t)(c0 + lane < c_hi ? c0 + lane : c_lo) * 8);
            const float4* bp1 = reinterpret_cast<const float4*>(fuse.boxes + (size_t)(two && c1 + lane < c_hi ? c1 + lane : c_lo) * 8);
            const float4 a0 = bp0[0], a1 = bp0[1], d0 = bp1[0], d1 = bp1[1];
            find_pass(c0, a0, a1, B, gl, gh);
            if (two) find_pass(c1, d0, d1, B, gl, gh);
        }
    };
    // hits are dealt round-robin; a wave fetches the box and the coordinates of up to 8 of its hits with ONE gather -- 8 lanes x
    // 16 bytes per hit -- into its private LDS stage
    auto gather_batch = [&](const int hb, const int h1) {
        {
            const int r = lane >> 3, part = lane & 7;
            const int h = hb + r * NW + w;
            if (h < h1) {
                const int chl = hits[h];
                const float* src = part < 2 ? fuse.boxes + (size_t)chl * 8 + part * 4
                                            : Q + (size_t)((part - 2) >> 1) * m_pad + (size_t)chl * 8 + (part & 1) * 4;
                *reinterpret_cast<float4*>(stage + r * STG + part * 4) = *reinterpret_cast<const float4*>(src);
            }
            if constexpr (PERM) {   // a sorted view: the elements' model indices (the sort permutation) are staged too
                const int r2 = lane >> 1, half = lane & 1;
                const int h2 = hb + r2 * NW + w;
                if (lane < 16 && h2 < h1)
                    *reinterpret_cast<int4*>(stage + r2 * STG + 32 + half * 4) =
                        *reinterpret_cast<const int4*>(fuse.q_perm + (size_t)hits[h2] * 8 + half * 4);
            }
        }
        lds_same_wave_order();
    };
    auto scan_batch = [&](const int hb, const int h1) {
        const int mine = (h1 - hb - w + NW - 1) / NW;     // this wave's hits in the batch
        const int cnt = mine < 8 ? mine : 8;
        // stage 1: the per-point box test of ALL of the wave's hits (independent loads: their LDS latencies overlap),
        // against the bound the wave starts the batch with
        unsigned int alive = 0u;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            if (rr < cnt) {
                const float4 b0 = *reinterpret_cast<const float4*>(stage + rr * STG);          // lo.xyz hi.x
                const float2 b1 = *reinterpret_cast<const float2*>(stage + rr * STG + 4);      // hi.yz
                // (g = p - clamp(p, lo, hi), as box_may_improve: a v_med3 and a subtraction per axis)
                const float gx = x - __builtin_amdgcn_fmed3f(x, b0.x, b0.w);
                const float gy = y - __builtin_amdgcn_fmed3f(y, b0.y, b1.x);
                const float gz = z - __builtin_amdgcn_fmed3f(z, b0.z, b1.y);
                const float L = ((gx * gx + gy * gy) + gz * gz) * 0.99999905f;   // 1 - 2^-20, as box_may_improve
                if (__builtin_amdgcn_ballot_w64(L <= best) != 0ull) alive |= 1u << rr;   // (ties pass: the hits are unordered)
            }
        }
        if constexpr (DIAG) { wk_hit[0] += (unsigned int)cnt; wk_hit[1] += (unsigned int)__builtin_popcount(alive); wk_hit[2] += (unsigned int)__builtin_popcount(alive); }
        if (w != 0) { ICP_PHASE(7) }
        // stage 2: the survivors, in full
        while (alive != 0u) {
            const int rr = __builtin_ctz(alive);
            alive &= alive - 1u;
            if constexpr (PERM) {
                scan_hit1<true>(stage + rr * STG, 0, px, py, pz, best, bj, bq);
            } else {
                const int ch = __builtin_amdgcn_readfirstlane(hits[hb + rr * NW + w]);
                scan_hit1<false>(stage + rr * STG, ch, px, py, pz, best, bj, bq);
            }
        }
        lds_same_wave_order();
    };
    auto process_hits = [&](const int h1) {
        for (int hb = 0; hb < h1; hb += NW * 8) {
            gather_batch(hb, h1);
            scan_batch(hb, h1);
        }
    };
    // exchange before the next round: every wave goes on from the block's best minimum so far, bumped by an ulp
    auto exchange = [&]() {
        if (real) atomicMin(&smin[lane], __float_as_uint(best));
        __syncthreads();
        if (threadIdx.x == 0) *hcount = 0;  // the list is consumed
        const unsigned int v = smin[lane];
        if (real && v < 0x7f800000u && v < __float_as_uint(best)) { best = __uint_as_float(v + 1u); bj = -1; }
    };
    bool searched = false;
    if (spec_valid) {
        // the list prepared during the wait covers this pass if the guessed group box and bound cover the real ones (see
        // nn_match_sparse: the box test is monotonic in both); extra chunks cost time, never the answer.  "Every point
        // inside the guessed box, every bound below the guessed one" is one ballot -- the group box itself is not needed
        // on this path (it is derived after the pass, for the next guess)
        const float4 s0 = *reinterpret_cast<const float4*>(spst), s1 = *reinterpret_cast<const float4*>(spst + 4);   // (broadcast reads)
        const bool outside = x < s0.x || y < s0.y || z < s0.z || x > s1.x || y > s1.y || z > s1.z || best > s0.w;
        const bool covered = __builtin_amdgcn_ballot_w64(outside) == 0ull;
        if (w != 0) { ICP_PHASE(6) }
        if constexpr (DIAG) {
            if (fuse.work != nullptr && threadIdx.x == 0) {
                atomicAdd(&fuse.work[NN_WORK_SPEC_LISTS], 1ull);
                if (covered) { atomicAdd(&fuse.work[NN_WORK_SPEC_COVERED], 1ull); atomicAdd(&fuse.work[NN_WORK_SPEC_HITS], (unsigned long long)*hcount); }
            }
        }
        if (covered) {
            scan_batch(0, *hcount);   // (the list's length is still in the counter)
            searched = true;
        } else {
            if (threadIdx.x == 0) *hcount = 0;   // the guess did not hold: forget the list and search as usual
            __syncthreads();
        }
    }
    // bounding box of the block's 64 moving points (every wave derives the same one; of its real points: see nn_match_sparse)
    const float binf = inf_<float>();
    float glo[3] = {real ? x : binf, real ? y : binf, real ? z : binf}, ghi[3] = {real ? x : -binf, real ? y : -binf, real ? z : -binf};
    if (!searched) {
        wave_box(glo, ghi);
        for (int rb = c_lo; rb < c_hi; rb += round_chunks) {
            // B only shrinks while the block works: refreshed once per round
            const float B = wave_minmax<true>(best);
            if (rb != c_lo) __syncthreads();  // the list is empty and its counter reset (first round: the barrier above)
            find_round(rb, B, glo, ghi);
            __syncthreads();
            if constexpr (DIAG) { if (fuse.work != nullptr && threadIdx.x == 0) atomicAdd(&fuse.work[NN_WORK_LIST_HITS], (unsigned long long)*hcount); }
            process_hits(*hcount);
            if (rb + round_chunks < c_hi) exchange();
        }
    }
    spec_valid = false;
    ICP_PHASE(3)
    if constexpr (DIAG) {
        if (fuse.work != nullptr && lane == 0) {
            if (wk_find) atomicAdd(&fuse.work[NN_WORK_FIND_BOXES], (unsigned long long)wk_find);
            if (wk_hit[0]) atomicAdd(&fuse.work[NN_WORK_HITS_BOX], (unsigned long long)wk_hit[0]);
            if (wk_hit[1]) atomicAdd(&fuse.work[NN_WORK_HITS_XY], (unsigned long long)wk_hit[1]);
            if (wk_hit[2]) atomicAdd(&fuse.work[NN_WORK_HITS_FULL], (unsigned long long)wk_hit[2]);
            if (wk_samp) atomicAdd(&fuse.work[NN_WORK_SAMPLE_GROUPS], (unsigned long long)wk_samp);
            if (w == 0) atomicAdd(&fuse.work[NN_WORK_BLOCK_PASSES], 1ull);
            if (w == 0 && apply) atomicAdd(&fuse.work[NN_WORK_BLOCK_TRANSFORMS], 1ull);
        }
        wk_find = wk_samp = 0; wk_hit[0] = wk_hit[1] = wk_hit[2] = 0;
        // (phase log) wave 1 leaves where the block runs: XCC_ID << 32 | HW_ID -- tools/cu_usage.py counts the CUs in use
        if (fuse.tlog != nullptr && lane == 0 && w == 1 && (fuse.tlog_pass < 0 || fuse.tlog_pass == phase_pass_)) {
            const long long slot_ = ((long long)blockIdx.x * phase_nw_ + w) * 10 + 8;
            if (slot_ < fuse.tlog_cap) fuse.tlog[slot_] = ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
        }
    }

    // in-block merge: every wave that lowered its bound folds its candidate into the point's key (the 64-bit integer order
    // is the lexicographic (distance, index) order of the tie rule; the low bits name the wave whose coordinates to use)
    if (bj >= 0) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | ((unsigned int)bj << 4) | (unsigned int)w;
        atomicMin(&mkey[lane], key);
        mq[0][w][lane] = bq[0]; mq[1][w][lane] = bq[1]; mq[2][w][lane] = bq[2];
    }
    ICP_PHASE(4)
    __syncthreads();
    ICP_PHASE(5)
    if (w != 0) {
        if (!fuse.resident) return;   // (resident: on to the speculative search below, then the next message)
    } else {
        // wave 0 finishes the row
        const unsigned long long key = mkey[lane];
        const bool none = key == ~0ull;  // no wave found anything below the bound (padding lanes)
        const unsigned int lo = (unsigned int)key;
        const float fb = none ? inf_<float>() : __uint_as_float((unsigned int)(key >> 32));
        int fj = none ? 0x7fffffff : (int)(lo >> 4);
        const int bw = none ? 0 : (int)(lo & 15u);
        sq[0] = mq[0][bw][lane]; sq[1] = mq[1][bw][lane]; sq[2] = mq[2][bw][lane];
        if constexpr (TAIL == 0) {
            part_d[pi] = fb;
            part_idx[pi] = fj;
            return;
        } else {
            fj = ((unsigned)fj < (unsigned)fuse.m) ? fj : fuse.m - 1;  // unreachable clamp
            NNTail tl = tail;
            tl.tag = row_tag;
            tl.tag_lo = row_tag_lo;
            tl.idx_out = (pass & 1) ? tail.idx_out_odd : tail.idx_out;
            ICP_PHASE(6)
            // the row's moments by the two-points-per-lane routine with its second point switched off (an index beyond n)
            const int j2[2] = {fj, 0}, pi2[2] = {pi, 0x7fffffff};
            float qio[2][3] = {{sq[0], sq[1], sq[2]}, {0.f, 0.f, 0.f}};
            tail_close_row<TAIL, DIAG, NW, true>(f2{x, 0.f}, f2{y, 0.f}, f2{z, 0.f}, j2, lane, pi2, m_pad, fuse, tl, apply ? err_row : 0.0, lds_raw, qio,
                                                 false, pass);
            ICP_PHASE(9)
            if (!fuse.resident) return;
            // the matches of this pass seed the next one and are what its error is measured against
            sok = real;
            seedq[0][lane] = sq[0]; seedq[1][lane] = sq[1]; seedq[2][lane] = sq[2];
        }
    }

    // ---- resident launch: the wait for the next message is put to use (see nn_match_sparse) -----------------------------
    if (threadIdx.x == 0) *hcount = 0;   // (this pass's list is consumed; ordered before its next use by the barriers below / the message barrier)
    const float B0 = wave_minmax<true>(best0);   // the largest starting bound of the block's points
    if (fuse.speculate && apply && B0 >= 0.f && B0 < inf_<float>()) {
        if (searched) wave_box(glo, ghi);   // (a pass served by the speculative list has not derived its group box yet)
        // guess: the next transform moves the points no further than twice what this one did (per axis, bounded over the
        // group box: p_new - p_old = (I - R^T) p_new + R^T t), plus a thousandth of the box
        float dn2 = 0.f, sp_lo[3], sp_hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float acc = 0.f, rt_t = 0.f;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float mm = (a == b ? 1.f : 0.f) - rt.r[b * 3 + a];          // (I - R^T)_ab
                acc += __builtin_fabsf(mm) * __builtin_fmaxf(__builtin_fabsf(glo[b]), __builtin_fabsf(ghi[b]));
                rt_t += rt.r[b * 3 + a] * rt.t[b];                               // (R^T t)_a
            }
            const float dl = fuse.spec_gain * (acc + __builtin_fabsf(rt_t)) + fuse.spec_floor * (ghi[a] - glo[a]) + 1e-6f;
            dn2 += dl * dl;
            sp_lo[a] = glo[a] - dl;
            sp_hi[a] = ghi[a] + dl;
        }
        // a point's next starting bound is its distance to this pass's match after the move: sqrt(d_new) <= sqrt(d_old) + |move|
        const float rB = __builtin_amdgcn_sqrtf(B0) + __builtin_amdgcn_sqrtf(dn2);
        const float sp_B = rB * rB * 1.0001f;
        if (lane == 0) {
            *reinterpret_cast<float4*>(spst) = float4{sp_lo[0], sp_lo[1], sp_lo[2], sp_B};
            *reinterpret_cast<float4*>(spst + 4) = float4{sp_hi[0], sp_hi[1], sp_hi[2], 0.f};
        }
        __syncthreads();   // wave 0 is through with the row (its transpose buffer overlays the hit list); the counter is reset
        // (one list over ALL rounds of the model: the bound does not change between them, and a list that does not fit one batch is dropped anyway)
        for (int rb = c_lo; rb < c_hi; rb += round_chunks) find_round(rb, sp_B, sp_lo, sp_hi);
        __syncthreads();
        const int spec_n = *hcount;
        if (spec_n <= NW * 8) {
            gather_batch(0, spec_n);
            spec_valid = true;
        } else {
            __syncthreads();   // (everybody has read the count)
            if (threadIdx.x == 0) *hcount = 0;
        }
    }
    }  // pass loop
}

// ------------------------------------------------------------------------------------------------
// matching, fp64, sparse, 64-point rows -- the CPU path's precision (src/ICP_CPU.c:220-234) on the structure of
// nn_match_row64: chunk boxes (in double), seeded ulp-bumped bounds, lane-parallel find, unordered hit list with the
// explicit (distance, index) tie rule, one point per lane.  There is no packed fp64 arithmetic, so a hit chunk is eight
// scalar evaluations of (dx*dx + dy*dy) + dz*dz per lane, every operation rounded on its own.  One launch per pass
// (no mailbox: a message line holds twelve floats, not twelve doubles): [transform + error of the previous pass] ->
// matching -> moment row, where round 1 needed three launches per pass around a dense thread-per-point scan.
// A 64-bit distance and an index do not fit one LDS key: every wave leaves its candidate (distance, index,
// coordinates) in LDS and wave 0 takes the lexicographic minimum over the eight of them.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fmax(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double bump_ulp(double d)   // next double above d (d >= 0, finite)
{
    return __longlong_as_double(__double_as_longlong(d) + 1ll);
}

// the transpose buffer of a row tail, filled from ONE point per lane given in double (see tail_close_row)
template <int TAIL>
__device__ __forceinline__ void tail_fill_one(double (*tr)[65], int lane, bool live, double ppx, double ppy, double ppz, double qx, double qy,
                                              double qz, double nx, double ny, double nz)
{
    if constexpr (TAIL == 1) {
        tr[0][lane] = 0.0 + (live ? 1.0 : 0.0);
        tr[1][lane] = 0.0 + ppx; tr[2][lane] = 0.0 + ppy; tr[3][lane] = 0.0 + ppz;
        tr[4][lane] = 0.0 + qx; tr[5][lane] = 0.0 + qy; tr[6][lane] = 0.0 + qz;
        tr[7][lane] = 0.0 + qx * ppx; tr[8][lane] = 0.0 + qx * ppy; tr[9][lane] = 0.0 + qx * ppz;
        tr[10][lane] = 0.0 + qy * ppx; tr[11][lane] = 0.0 + qy * ppy; tr[12][lane] = 0.0 + qy * ppz;
        tr[13][lane] = 0.0 + qz * ppx; tr[14][lane] = 0.0 + qz * ppy; tr[15][lane] = 0.0 + qz * ppz;
        tr[16][lane] = 0.0 + (ppx * ppx + ppy * ppy + ppz * ppz);
        tr[17][lane] = 0.0 + (qx * qx + qy * qy + qz * qz);
    } else {
        double cn[6] = {0, 0, 0, 0, 0, 0}, bb = 0.0;
        if (live) {
            cn[0] = ppy * nz - ppz * ny;
            cn[1] = ppz * nx - ppx * nz;
            cn[2] = ppx * ny - ppy * nx;
            cn[3] = nx; cn[4] = ny; cn[5] = nz;
            bb = (ppx - qx) * nx + (ppy - qy) * ny + (ppz - qz) * nz;
        }
        tr[0][lane] = 0.0 + (live ? 1.0 : 0.0);
        int o = 1;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c2 = a; c2 < 6; ++c2) tr[o++][lane] = 0.0 + cn[a] * cn[c2];
#pragma unroll
        for (int a = 0; a < 6; ++a) tr[22 + a][lane] = 0.0 + -(cn[a] * bb);
    }
}

// NW: waves per block -- 16 while that still gives every block its own CU (a far-apart pair is hundreds of hits per
// block: the more waves share them the better), else 8 (two blocks per CU)
template <int TAIL, int NW, bool DIAG = false /* icp_set_work_counting: tallies of the executed work (NNFuse::work) */>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void nn_match_row64_f64(const double* __restrict__ P, int n_pad, const double* __restrict__ Q,
                                                                     int m_pad, int round_passes, double* __restrict__ part_d,
                                                                     int32_t* __restrict__ part_idx, RT<double> rt_arg, NNFuse fuse, NNTail tail)
{
    constexpr int STG = 32;              // doubles per staged hit: box 8, x 8, y 8, z 8
    constexpr int SMAX = 1024;           // cold start: samples staged per round
    constexpr int HITS_BYTES = SP_HCAP * 4, SAMPLE_BYTES = 3 * SMAX * 8;
    constexpr int TR_BYTES = TAIL ? ((TAIL == 2 ? 28 : 18) * 65 + 64) * 8 : 0;
    static_assert(TR_BYTES <= HITS_BYTES && HITS_BYTES <= SAMPLE_BYTES, "transpose buffer and staged samples overlay the hit list");
    constexpr int SMIN_OFF = SAMPLE_BYTES, HCNT_OFF = SMIN_OFF + 64 * 8, STAGE_OFF = HCNT_OFF + 16, STAGE_BYTES = NW * 8 * STG * 8;
    constexpr int CD_OFF = STAGE_OFF + STAGE_BYTES, CJ_OFF = CD_OFF + NW * 64 * 8, CQ_OFF = CJ_OFF + NW * 64 * 4;
    constexpr int MSG_OFF = CQ_OFF + 3 * NW * 64 * 8, SEED_OFF = MSG_OFF + 128;   // resident launch: the message (32 words), last pass's matches
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SEED_OFF + 3 * 64 * 8];
    int* hits = reinterpret_cast<int*>(lds_raw);
    unsigned long long* smin = reinterpret_cast<unsigned long long*>(lds_raw + SMIN_OFF);
    int* hcount = reinterpret_cast<int*>(lds_raw + HCNT_OFF);
    double (*cand_d)[64] = reinterpret_cast<double (*)[64]>(lds_raw + CD_OFF);
    int (*cand_j)[64] = reinterpret_cast<int (*)[64]>(lds_raw + CJ_OFF);
    double (*cand_q)[NW][64] = reinterpret_cast<double (*)[NW][64]>(lds_raw + CQ_OFF);
    uint32_t* msg = reinterpret_cast<uint32_t*>(lds_raw + MSG_OFF);
    double (*seedq)[64] = reinterpret_cast<double (*)[64]>(lds_raw + SEED_OFF);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pi = blockIdx.x * 64 + lane;
    double* stage = reinterpret_cast<double*>(lds_raw + STAGE_OFF) + w * (8 * STG);
    const double* boxes = reinterpret_cast<const double*>(fuse.boxes);
    const double* Qg = reinterpret_cast<const double*>(fuse.Q_gather);
    const int c_lo = 0, c_hi = m_pad / 8;
    const bool real = pi < fuse.n;
    constexpr double kInf = __builtin_huge_val();
    constexpr unsigned long long kInfBits = 0x7ff0000000000000ull;

    // the seed: last pass's match, or (cold start) the model point at the same RELATIVE index -- any valid index is a valid bound
    int js = !real ? -1 : fuse.seed_idx ? fuse.seed_idx[pi] : (int)(((long long)pi * fuse.m) / fuse.n);
    bool sok = (unsigned)js < (unsigned)fuse.m;
    js = sok ? js : 0;
    double sq[3] = {Qg[js], Qg[(size_t)m_pad + js], Qg[2 * (size_t)m_pad + js]};
    double x = P[pi], y = P[(size_t)n_pad + pi], z = P[2 * (size_t)n_pad + pi];
    unsigned int wk_find = 0, wk_hit[2] = {0, 0}, wk_samp = 0;   // (work-counting instantiation only)
    // ---- the pass loop: one turn for an ordinary launch, one per ICP pass for a resident one (fuse.mailbox: a message of
    // TWO cache lines -- twelve doubles do not fit one -- in four 32-byte parts {3 doubles, cmd, tag}: NNMailbox64) ----
    for (int pass = 0;; ++pass) {
    RT<double> rt = rt_arg;
    int cmd = fuse.apply ? ICP_CMD_TRANSFORM_MATCH : ICP_CMD_MATCH;
    double row_tag = tail.tag;
    unsigned int row_tag_lo = tail.tag_lo;
    const bool have_seeds = pass > 0 || fuse.seed_idx != nullptr;
    if (w == 0) smin[lane] = kInfBits;
    if (threadIdx.x == 0) *hcount = 0;
    if (fuse.mailbox != nullptr) {
        if (w == 0) {
            // one load fetches both lines (lane l reads word l & 31); the message is there when all four parts carry the
            // awaited tag.  Block 0 relays a host-memory mailbox through device memory; the wait is bounded in wall-clock time
            const bool first = blockIdx.x == 0 || fuse.relay == nullptr;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(first ? (const void*)fuse.mailbox : (const void*)fuse.relay) + (lane & 31);
            const uint32_t want32 = (fuse.want_lo + (uint32_t)pass) | 0x80000000u;
            uint32_t word = 0u;
            bool ok = false;
            const long long give_up = (long long)wall_clock64() + (first ? ICP_MAILBOX_BUDGET_TICKS : 2 * ICP_MAILBOX_BUDGET_TICKS);
            for (unsigned int spins = 1;; ++spins) {
                word = first ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                             : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (uint32_t)__builtin_amdgcn_readlane((int)word, 7) == want32 && (uint32_t)__builtin_amdgcn_readlane((int)word, 15) == want32 &&
                     (uint32_t)__builtin_amdgcn_readlane((int)word, 23) == want32 && (uint32_t)__builtin_amdgcn_readlane((int)word, 31) == want32;
                if (ok) break;
                if ((spins & 63u) == 0u && (long long)wall_clock64() > give_up) break;
                __builtin_amdgcn_s_sleep(2);
            }
            // (a time-out reads as a withdrawal)
            if (!ok) word = (lane & 7) == ICP_MB64_CMD ? (uint32_t)ICP_CMD_EXIT : (lane & 7) == 7 ? want32 : 0u;
            if (first && fuse.relay != nullptr && lane < 32)
                __hip_atomic_store(reinterpret_cast<uint32_t*>(fuse.relay) + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane < 32) msg[lane] = word;
        }
        __syncthreads();
        cmd = (int)msg[ICP_MB64_CMD];
        if (cmd == ICP_CMD_EXIT) return;  // withdrawn (the loop stopped) or timed out: nothing more is touched
#pragma unroll
        for (int k = 0; k < 12; ++k) {   // double k sits in part k / 3, words 2 (k % 3) and 2 (k % 3) + 1
            const int wd = (k / 3) * 8 + (k % 3) * 2;
            const double v = __hiloint2double((int)msg[wd + 1], (int)msg[wd]);
            if (k < 9) rt.r[k] = v; else rt.t[k - 9] = v;
        }
        row_tag = fuse.want + (double)pass;
        row_tag_lo = fuse.want_lo + (unsigned int)pass;
        if (pass > 0) {   // the seeds of a resident pass are the matches of the one before
            sok = real;
            sq[0] = seedq[0][lane]; sq[1] = seedq[1][lane]; sq[2] = seedq[2][lane];
        }
    } else {
        __syncthreads();
    }
    const bool apply = cmd != ICP_CMD_MATCH;

    double err_row = 0.0;
    if (apply) {
        // every wave re-derives the moved point (same instructions => same bits); wave 0 stores it and accounts the error
        // of the pass that produced (R, t): the statements of src/ICP_CPU.c:251-266
        apply_rt<double>(rt, x, y, z, x, y, z);
        if (w == 0) {
            double* Po = reinterpret_cast<double*>(fuse.P_out);
            Po[pi] = x; Po[(size_t)n_pad + pi] = y; Po[2 * (size_t)n_pad + pi] = z;
            double err = 0.0;
            if (real) {
                double qx = sq[0], qy = sq[1], qz = sq[2];
                if (!((pass > 0 || (fuse.seed_idx != nullptr && fuse.idx_prev == fuse.seed_idx)) && sok)) {
                    const int j = fuse.idx_prev[pi];
                    qx = Qg[j]; qy = Qg[(size_t)m_pad + j]; qz = Qg[2 * (size_t)m_pad + j];
                }
                const double ex = qx - x, ey = qy - y, ez = qz - z;
                err = ex * ex + ey * ey + ez * ez;
            }
            err_row = wave_sum(err);
            if constexpr (TAIL == 0) { if (lane == 0) fuse.err_rows[blockIdx.x] = err_row; }
        }
    }
    if (!apply && pass == 0 && fuse.store_first && w == 0) {   // resident launch reading a pristine copy
        double* Po = reinterpret_cast<double*>(fuse.P_out);
        Po[pi] = x; Po[(size_t)n_pad + pi] = y; Po[2 * (size_t)n_pad + pi] = z;
    }
    if (cmd == ICP_CMD_TRANSFORM_ONLY) {
        // the loop's last pass: nothing is matched any more, the row carries the error alone
        if constexpr (TAIL != 0) {
            if (w == 0) {
                if (TAIL == 1 && tail.compact != 0) {
                    double* row = tail.rows + (size_t)blockIdx.x * NN_CROW;
                    if (lane >= 1 && lane < NN_CROW) __hip_atomic_store(&row[lane], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&row[0], crow_pack(err_row, row_tag_lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    double* row = tail.rows + (size_t)blockIdx.x * ICP_NMOM;
                    if (lane < ICP_NMOM - 1) row[lane] = lane == ICP_MOM_ERR ? err_row : 0.0;
                    __threadfence_system();
                    if (lane == 0) __hip_atomic_store(&row[ICP_NMOM - 1], row_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        return;
    }
    double best = kInf;
    double bq[3] = {0.0, 0.0, 0.0};
    int bj = -1;   // index of the running minimum; -1: this wave has not lowered the bound it started from
    {
        const double d = dist2<double>(x, y, z, sq[0], sq[1], sq[2]);
        best = (sok && d < kInf) ? bump_ulp(d) : kInf;   // the true minimum is <= d < bound: the seed changes the work, never the answer
        best = real ? best : -1.0;                       // padding lanes never ask for a chunk
    }
    if (!have_seeds && fuse.samples != nullptr) {
        // cold start: the points are measured against a thinned-out model (one point per chunk, up to SMAX of them spread
        // evenly) and every wave starts from the block-wide minimum bumped by an ulp
        const double* samples = reinterpret_cast<const double*>(fuse.samples);
        const int ns8 = ((m_pad / 8) + 7) / 8, ns_pad = ns8 * 8;
        double* sl = reinterpret_cast<double*>(lds_raw);       // [3][SMAX]
        const int gcap = min(max(fuse.sample_groups, 1), SMAX / 8);
        const int gs = (ns8 + gcap - 1) / gcap, ng = (ns8 + gs - 1) / gs;
        for (int v = threadIdx.x; v < ng * 12; v += NW * 64) {
            const int gp = v / 12, r = v % 12, a = r >> 2, hh = r & 3;
            *reinterpret_cast<double2*>(sl + a * SMAX + gp * 8 + hh * 2) =
                *reinterpret_cast<const double2*>(samples + (size_t)a * ns_pad + (size_t)gp * gs * 8 + hh * 2);
        }
        __syncthreads();
        double sb = kInf;
        for (int gp = w; gp < ng; gp += NW) {
            if constexpr (DIAG) ++wk_samp;
#pragma unroll
            for (int k = 0; k < 8; ++k) sb = __builtin_fmin(sb, dist2<double>(x, y, z, sl[gp * 8 + k], sl[SMAX + gp * 8 + k], sl[2 * SMAX + gp * 8 + k]));
        }
        if (real && sb < kInf) atomicMin(&smin[lane], (unsigned long long)__double_as_longlong(sb));
        __syncthreads();  // (also: the staging area is free again)
        const unsigned long long v = smin[lane];
        if (real && v < kInfBits && __longlong_as_double((long long)(v + 1ull)) < best) best = __longlong_as_double((long long)(v + 1ull));
        __syncthreads();  // everybody has read the minima before they are used again
        if (w == 0) smin[lane] = kInfBits;
    }
    // bounding box of the block's 64 points (every wave derives the same one)
    const double binf = inf_<double>();   // (the box of the REAL points: see nn_match_sparse)
    const double glo[3] = {wave_min_f64(real ? x : binf), wave_min_f64(real ? y : binf), wave_min_f64(real ? z : binf)};
    const double ghi[3] = {wave_max_f64(real ? x : -binf), wave_max_f64(real ? y : -binf), wave_max_f64(real ? z : -binf)};
    const int round_chunks = NW * 64 * round_passes;
    for (int rb = c_lo; rb < c_hi; rb += round_chunks) {
        const double B = wave_max_f64(best);   // only shrinks while the block works: refreshed once per round
        if (rb != c_lo) __syncthreads();
        // find: lane l tests chunk c0 + l -- box {lo.xyz, hi.xyz, -, -} against the group box
        for (int r = 0; r < round_passes; ++r) {
            const int c0 = rb + (r * NW + w) * 64;
            if (c0 >= c_hi) break;
            const int cidx = c0 + lane;
            if constexpr (DIAG) wk_find += (unsigned int)max(0, min(64, c_hi - c0));
            const double* bp = boxes + (size_t)(cidx < c_hi ? cidx : c_lo) * 8;
            const double2 b01 = *reinterpret_cast<const double2*>(bp), b23 = *reinterpret_cast<const double2*>(bp + 2),
                          b45 = *reinterpret_cast<const double2*>(bp + 4);
            const double gx = __builtin_fmax(__builtin_fmax(b01.x - ghi[0], glo[0] - b23.y), 0.0);
            const double gy = __builtin_fmax(__builtin_fmax(b01.y - ghi[1], glo[1] - b45.x), 0.0);
            const double gz = __builtin_fmax(__builtin_fmax(b23.x - ghi[2], glo[2] - b45.y), 0.0);
            const double L = ((gx * gx + gy * gy) + gz * gz) * 0.99999999;   // rounding is monotonic; the shave is belt and braces
            const bool pass_ = cidx < c_hi && L < B;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass_);
            if (mask != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(hcount, (int)__builtin_popcountll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (pass_ && base + rank < SP_HCAP) hits[base + rank] = cidx;
            }
        }
        __syncthreads();
        const int h1 = *hcount;
        // hits are dealt round-robin; a wave fetches box + coordinates of up to 8 of its hits per batch (16 lanes x 16 bytes
        // per hit).  The fetch of the NEXT batch is in flight while the current one is scanned: a far-apart pair
        // (src/ICP_CPU.c's own: 870 of 1250 chunks survive the group test) is a dozen batches per wave, and their memory
        // round trips in a row were more than half of the pass.
        double2 nxt0 = double2{0.0, 0.0}, nxt1 = nxt0;   // (named, not an array: the compiler moved an array of two to LDS)
        const int part = lane & 15, r0 = lane >> 4, r1 = 4 + (lane >> 4);
        auto fetch_one = [&](int h, double2& dst) {
            if (h < h1) {
                const int chl = hits[h];
                const double* src = part < 4 ? boxes + (size_t)chl * 8 + part * 2
                                             : Q + (size_t)((part - 4) >> 2) * m_pad + (size_t)chl * 8 + ((part - 4) & 3) * 2;
                dst = *reinterpret_cast<const double2*>(src);
            }
        };
        if (h1 > 0) { fetch_one(r0 * NW + w, nxt0); fetch_one(r1 * NW + w, nxt1); }
        for (int hb = 0; hb < h1; hb += NW * 8) {
            if (hb + r0 * NW + w < h1) *reinterpret_cast<double2*>(stage + r0 * STG + part * 2) = nxt0;
            if (hb + r1 * NW + w < h1) *reinterpret_cast<double2*>(stage + r1 * STG + part * 2) = nxt1;
            lds_same_wave_order();
            if (hb + NW * 8 < h1) { fetch_one(hb + NW * 8 + r0 * NW + w, nxt0); fetch_one(hb + NW * 8 + r1 * NW + w, nxt1); }
            const int mine = (h1 - hb - w + NW - 1) / NW;
            const int cnt = mine < 8 ? mine : 8;
            for (int rr = 0; rr < cnt; ++rr) {
                const double* sb = stage + rr * STG;
                if constexpr (DIAG) ++wk_hit[0];
                {   // the chunk's box against the lane's point (ties pass: the hits are unordered)
                    const double gx = __builtin_fmax(__builtin_fmax(sb[0] - x, x - sb[3]), 0.0);
                    const double gy = __builtin_fmax(__builtin_fmax(sb[1] - y, y - sb[4]), 0.0);
                    const double gz = __builtin_fmax(__builtin_fmax(sb[2] - z, z - sb[5]), 0.0);
                    const double L = ((gx * gx + gy * gy) + gz * gz) * 0.99999999;
                    if (__builtin_amdgcn_ballot_w64(L <= best) == 0ull) continue;
                }
                if constexpr (DIAG) ++wk_hit[1];
                const int ch = __builtin_amdgcn_readfirstlane(hits[hb + rr * NW + w]);
                double d[8];
                double c0 = kInf;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    d[k] = dist2<double>(x, y, z, sb[8 + k], sb[16 + k], sb[24 + k]);
                    c0 = __builtin_fmin(c0, d[k]);
                }
                // identity order: chunks are disjoint index ranges, "lower model index" is "lower chunk, then lower k"
                const bool take = (c0 < best) | ((c0 == best) & (ch < (bj >> 3)));   // bj = -1: nothing to tie with
                if (__builtin_amdgcn_ballot_w64(take) != 0ull) {
                    int k0 = 7;
#pragma unroll
                    for (int k = 6; k >= 0; --k) k0 = (d[k] == c0) ? k : k0;
                    if (take) { best = c0; bj = ch * 8 + k0; bq[0] = sb[8 + k0]; bq[1] = sb[16 + k0]; bq[2] = sb[24 + k0]; }
                }
            }
            lds_same_wave_order();
        }
        if (rb + round_chunks < c_hi) {
            // exchange before the next round: every wave goes on from the block's best minimum so far, bumped by an ulp
            if (real && best >= 0.0 && best < kInf) atomicMin(&smin[lane], (unsigned long long)__double_as_longlong(best));
            __syncthreads();
            if (threadIdx.x == 0) *hcount = 0;
            const unsigned long long v = smin[lane];
            if (real && v < kInfBits && v < (unsigned long long)__double_as_longlong(best)) { best = __longlong_as_double((long long)(v + 1ull)); bj = -1; }
        }
    }
    if constexpr (DIAG) {
        if (fuse.work != nullptr && lane == 0) {
            if (wk_find) atomicAdd(&fuse.work[NN_WORK_FIND_BOXES], (unsigned long long)wk_find);
            if (wk_hit[0]) atomicAdd(&fuse.work[NN_WORK_HITS_BOX], (unsigned long long)wk_hit[0]);
            if (wk_hit[1]) { atomicAdd(&fuse.work[NN_WORK_HITS_XY], (unsigned long long)wk_hit[1]); atomicAdd(&fuse.work[NN_WORK_HITS_FULL], (unsigned long long)wk_hit[1]); }
            if (wk_samp) atomicAdd(&fuse.work[NN_WORK_SAMPLE_GROUPS], (unsigned long long)wk_samp);
            if (w == 0) atomicAdd(&fuse.work[NN_WORK_BLOCK_PASSES], 1ull);
            if (w == 0 && apply) atomicAdd(&fuse.work[NN_WORK_BLOCK_TRANSFORMS], 1ull);
        }
        wk_find = wk_samp = 0; wk_hit[0] = wk_hit[1] = 0;
    }
    // every wave leaves its candidate; wave 0 takes the lexicographic (distance, index) minimum
    cand_d[w][lane] = bj >= 0 ? best : kInf;
    cand_j[w][lane] = bj >= 0 ? bj : 0x7fffffff;
    cand_q[0][w][lane] = bq[0]; cand_q[1][w][lane] = bq[1]; cand_q[2][w][lane] = bq[2];
    __syncthreads();
    if (w != 0) {
        if (!fuse.resident) return;
        continue;   // resident: on to the next message (asleep at its barrier while wave 0 closes the row)
    }
    double fb = cand_d[0][lane];
    int fj = cand_j[0][lane], bw = 0;
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {
        const double dd = cand_d[ww][lane];
        const int jj = cand_j[ww][lane];
        const bool lower = (dd < fb) | ((dd == fb) & (jj < fj));
        fb = lower ? dd : fb; fj = lower ? jj : fj; bw = lower ? ww : bw;
    }
    if constexpr (TAIL == 0) {
        part_d[pi] = fb;
        part_idx[pi] = fj;
        return;
    } else {
        fj = ((unsigned)fj < (unsigned)fuse.m) ? fj : fuse.m - 1;  // unreachable clamp (padding lanes)
        const double qx = cand_q[0][bw][lane], qy = cand_q[1][bw][lane], qz = cand_q[2][bw][lane];
        if (real) ((pass & 1) ? tail.idx_out_odd : tail.idx_out)[pi] = fj;
        double nx = 0.0, ny = 0.0, nz = 0.0;
        if constexpr (TAIL == 2) {
            const double* Nr = reinterpret_cast<const double*>(tail.Nrm);
            if (real) { nx = Nr[fj]; ny = Nr[(size_t)m_pad + fj]; nz = Nr[2 * (size_t)m_pad + fj]; }
        }
        double (*tr)[65] = reinterpret_cast<double (*)[65]>(lds_raw);
        tail_fill_one<TAIL>(tr, lane, real, real ? x : 0.0, real ? y : 0.0, real ? z : 0.0, real ? qx : 0.0, real ? qy : 0.0, real ? qz : 0.0, nx, ny, nz);
        NNTail tl = tail;
        tl.tag = row_tag;
        tl.tag_lo = row_tag_lo;
        tail_reduce_store<TAIL, false, NW>(tr, lane, fuse, tl, apply ? err_row : 0.0, 0);
        if (!fuse.resident) return;
        // the matches of this pass seed the next one and are what its error is measured against
        seedq[0][lane] = qx; seedq[1][lane] = qy; seedq[2][lane] = qz;
    }
    }  // pass loop
}
#undef ICP_PHASE

// diagnostic (ICP_SELFTEST=1): does a running kernel see a store the host makes AFTER the kernel has started?
// The kernel reports that it runs (ack = 1), waits for mb->seq == 2 and answers ack = 2 (or -1 when its budget ends).
__global__ void mailbox_selftest_kernel(const NNMailbox* mb, double* ack)
{
    if (threadIdx.x == 0) __hip_atomic_store(ack, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t want32 = mailbox_tag(2.0);
    bool ok = false;
    for (int spins = 0; spins < (1 << 20) && !ok; ++spins) {
        const uint32_t word = __hip_atomic_load(&mb->w[threadIdx.x & 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = (uint32_t)__builtin_amdgcn_readlane((int)word, ICP_MB_TAG0) == want32 && (uint32_t)__builtin_amdgcn_readlane((int)word, ICP_MB_TAG1) == want32;
        if (!ok) __builtin_amdgcn_s_sleep(2);
    }
    if (threadIdx.x == 0) __hip_atomic_store(ack, ok ? 2.0 : -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_mailbox_selftest(const NNMailbox* mb, double* ack, hipStream_t st)
{
    hipLaunchKernelGGL(mailbox_selftest_kernel, dim3(1), dim3(64), 0, st, mb, ack);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// preparation of a cloud for the sparse kernel, on the device (once per icp_set_model / icp_set_moving):
// exact-duplicate flags (lexicographic order of the raw coordinate bits: three stable radix passes),
// Morton order, and the grouped-extent test that decides whether the Morton-ordered view is used.
// Everything is deterministic (fixed-order reductions): the decision must not change from run to run.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int canon_bits(float v) { return __float_as_uint(v == 0.0f ? 0.0f : v); }
__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    return x - x == 0.f && y - y == 0.f && z - z == 0.f;   // false for NaN and +-inf
}

// keys[s] = raw bits of coordinate `axis` of point order[s] (order == NULL: identity, and vals is initialised)
__global__ void prep_axis_keys_kernel(const float* __restrict__ X, int n, int n_pad, int axis, const int32_t* __restrict__ order,
                                      unsigned int* __restrict__ keys, int32_t* __restrict__ vals)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int i = order ? order[s] : s;
    keys[s] = canon_bits(X[(size_t)axis * n_pad + i]);
    if (!order) vals[s] = s;
}

// lex[s] ascending in (x, y, z, index): a point equal to its predecessor has a lower-index twin
__global__ void prep_mark_duplicates_kernel(const float* __restrict__ X, int n, int n_pad, const int32_t* __restrict__ lex,
                                            unsigned char* __restrict__ voided, int* __restrict__ count)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    bool v = false;
    if (s > 0) {
        const int a = lex[s - 1], b = lex[s];
        const float bx = X[b], by = X[(size_t)n_pad + b], bz = X[2 * (size_t)n_pad + b];
        const bool same = canon_bits(X[a]) == canon_bits(bx) && canon_bits(X[(size_t)n_pad + a]) == canon_bits(by) &&
                          canon_bits(X[2 * (size_t)n_pad + a]) == canon_bits(bz);
        const bool nan = bx != bx || by != by || bz != bz;
        v = same && !nan;
    }
    voided[lex[s]] = v ? 1 : 0;
    if (v) atomicAdd(count, 1);
}

// scan copy: the cloud with its flagged points (and the padding) voided to +inf
__global__ void prep_scan_copy_kernel(const float* __restrict__ X, int n, int n_pad, const unsigned char* __restrict__ voided,
                                      float* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pad) return;
    const bool keep = j < n && !voided[j];
#pragma unroll
    for (int a = 0; a < 3; ++a) out[(size_t)a * n_pad + j] = keep ? X[(size_t)a * n_pad + j] : inf_<float>();
}

// ---- the same for a cloud in double: a 64-bit key is two stable 32-bit passes (low word, then high word) --------------
__device__ __forceinline__ unsigned long long canon_bits64(double v) { return (unsigned long long)__double_as_longlong(v == 0.0 ? 0.0 : v); }

__global__ void prep_axis_keys_f64_kernel(const double* __restrict__ X, int n, int n_pad, int axis, int high, const int32_t* __restrict__ order,
                                          unsigned int* __restrict__ keys, int32_t* __restrict__ vals)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int i = order ? order[s] : s;
    const unsigned long long b = canon_bits64(X[(size_t)axis * n_pad + i]);
    keys[s] = high ? (unsigned int)(b >> 32) : (unsigned int)b;
    if (!order) vals[s] = s;
}

__global__ void prep_mark_duplicates_f64_kernel(const double* __restrict__ X, int n, int n_pad, const int32_t* __restrict__ lex,
                                                unsigned char* __restrict__ voided, int* __restrict__ count)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    bool v = false;
    if (s > 0) {
        const int a = lex[s - 1], b = lex[s];
        const double bx = X[b], by = X[(size_t)n_pad + b], bz = X[2 * (size_t)n_pad + b];
        const bool same = canon_bits64(X[a]) == canon_bits64(bx) && canon_bits64(X[(size_t)n_pad + a]) == canon_bits64(by) &&
                          canon_bits64(X[2 * (size_t)n_pad + a]) == canon_bits64(bz);
        const bool nan = bx != bx || by != by || bz != bz;
        v = same && !nan;
    }
    voided[lex[s]] = v ? 1 : 0;
    if (v) atomicAdd(count, 1);
}

__global__ void prep_scan_copy_f64_kernel(const double* __restrict__ X, int n, int n_pad, const unsigned char* __restrict__ voided,
                                          double* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pad) return;
    const bool keep = j < n && !voided[j];
#pragma unroll
    for (int a = 0; a < 3; ++a) out[(size_t)a * n_pad + j] = keep ? X[(size_t)a * n_pad + j] : inf_<double>();
}

// bounding cube of the finite points: box[0..2] = lo, box[3] = largest extent.  One block, or (partial != NULL) a grid of them
// that leave {lo, hi} per block for prep_bbox_final_kernel -- minima and maxima: the result does not depend on the split
// (a 10 M-point cloud through one block was 3.8 ms of a 37 ms set-up, twice)
__global__ __launch_bounds__(1024) void prep_bbox_kernel(const float* __restrict__ X, int n, int n_pad, float* __restrict__ box, float* __restrict__ partial)
{
    __shared__ float red[6][1024];
    float lo[3] = {inf_<float>(), inf_<float>(), inf_<float>()}, hi[3] = {-inf_<float>(), -inf_<float>(), -inf_<float>()};
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += 1024 * gridDim.x) {
        const float x = X[i], y = X[(size_t)n_pad + i], z = X[2 * (size_t)n_pad + i];
        if (!finite3(x, y, z)) continue;
        lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
        hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
    }
    for (int a = 0; a < 3; ++a) { red[a][threadIdx.x] = lo[a]; red[3 + a][threadIdx.x] = hi[a]; }
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int a = 0; a < 3; ++a) {
                red[a][threadIdx.x] = fminf(red[a][threadIdx.x], red[a][threadIdx.x + w]);
                red[3 + a][threadIdx.x] = fmaxf(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + w]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (partial != nullptr) {
            for (int a = 0; a < 6; ++a) partial[blockIdx.x * 6 + a] = red[a][0];
            return;
        }
        float ext = 0.f;
        for (int a = 0; a < 3; ++a) { box[a] = red[a][0]; ext = fmaxf(ext, red[3 + a][0] - red[a][0]); }
        box[3] = ext;   // -inf / NaN when there is no finite point: the codes below then all take the "last" value
    }
}

__global__ __launch_bounds__(64) void prep_bbox_final_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ box)
{
    float v[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) v[a] = a < 3 ? inf_<float>() : -inf_<float>();
    for (int b = threadIdx.x; b < blocks; b += 64)
#pragma unroll
        for (int a = 0; a < 6; ++a) v[a] = a < 3 ? fminf(v[a], partial[b * 6 + a]) : fmaxf(v[a], partial[b * 6 + a]);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(v[a], off, 64); v[a] = a < 3 ? fminf(v[a], o) : fmaxf(v[a], o); }
    if (threadIdx.x == 0) {
        float ext = 0.f;
        for (int a = 0; a < 3; ++a) { box[a] = v[a]; ext = fmaxf(ext, v[3 + a] - v[a]); }
        box[3] = ext;
    }
}

__device__ __forceinline__ unsigned int spread10(unsigned int v)
{
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// position of cell (x, y, z) of a 1024^3 grid along the Hilbert curve (J. Skilling, "Programming the Hilbert curve", AIP Conf.
// Proc. 707, 2004: axes -> transposed index, then the bits interleaved).  Consecutive positions are neighbouring cells -- a
// Z-order range of 128 points straddles the curve's jumps, and the group box is the union: on the 10 M-point surface the rows
// of 128 measure 0.027 x 0.027 x 0.031 in this order against 0.038 x 0.034 x 0.031 in Z-order, and every level of the search
// lists 10-19 % fewer boxes (tools/s5_hits_model.py)
__device__ __forceinline__ unsigned int hilbert30(unsigned int x, unsigned int y, unsigned int z)
{
    unsigned int X[3] = {x & 1023u, y & 1023u, z & 1023u};
#pragma unroll
    for (unsigned int Q = 512u; Q > 1u; Q >>= 1) {
        const unsigned int P = Q - 1u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned int t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned int t = 0u;
#pragma unroll
    for (unsigned int Q = 512u; Q > 1u; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1u;
    return (spread10(X[0] ^ t) << 2) | (spread10(X[1] ^ t) << 1) | spread10(X[2] ^ t);
}

// 30-bit space-filling-curve codes in one cube for all axes (cells stay cubic): the Hilbert curve, or Z-order (hilbert == 0:
// ICP_ORDER=morton, A/B runs); non-finite points go last
__global__ void prep_morton_keys_kernel(const float* __restrict__ X, int n, int n_pad, const float* __restrict__ box,
                                        unsigned int* __restrict__ keys, int32_t* __restrict__ vals, int hilbert)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = X[i], y = X[(size_t)n_pad + i], z = X[2 * (size_t)n_pad + i];
    unsigned int code = 0x7fffffffu;
    const float ext = box[3];
    if (finite3(x, y, z) && ext >= 0.f) {
        const double scale = ext > 0.f ? 1023.0 / (double)ext : 0.0;
        const unsigned int qx = (unsigned int)fmin(1023.0, fmax(0.0, ((double)x - (double)box[0]) * scale));
        const unsigned int qy = (unsigned int)fmin(1023.0, fmax(0.0, ((double)y - (double)box[1]) * scale));
        const unsigned int qz = (unsigned int)fmin(1023.0, fmax(0.0, ((double)z - (double)box[2]) * scale));
        code = hilbert ? hilbert30(qx, qy, qz) : (spread10(qx) | (spread10(qy) << 1) | (spread10(qz) << 2));
    }
    keys[i] = code;
    vals[i] = i;
}

// per group of `group` consecutive entries of an order (NULL: the cloud's own): extent dx + dy + dz of its finite points.
// One wave per group (a thread per group walks 128 gathered points one after the other: 50 us for 128 groups).
__global__ __launch_bounds__(64) void prep_group_extent_kernel(const float* __restrict__ X, int n, int n_pad, const int32_t* __restrict__ order,
                                                               int group, double* __restrict__ ext)
{
    const int g = blockIdx.x, g0 = g * group, lane = threadIdx.x;
    float lo[3] = {inf_<float>(), inf_<float>(), inf_<float>()}, hi[3] = {-inf_<float>(), -inf_<float>(), -inf_<float>()};
    for (int k = g0 + lane; k < min(n, g0 + group); k += 64) {
        const int i = order ? order[k] : k;
        const float p[3] = {X[i], X[(size_t)n_pad + i], X[2 * (size_t)n_pad + i]};
        if (!finite3(p[0], p[1], p[2])) continue;
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
        }
    if (lane == 0) ext[g] = hi[0] >= lo[0] ? (double)(hi[0] - lo[0]) + (double)(hi[1] - lo[1]) + (double)(hi[2] - lo[2]) : 0.0;
}

// out[which] = sum of ext[0..groups) in a fixed order (one block)
__global__ __launch_bounds__(256) void prep_sum_kernel(const double* __restrict__ ext, int groups, double* __restrict__ out, int which)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int g = threadIdx.x; g < groups; g += 256) s += ext[g];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[which] = red[0];
}

// Morton-ordered view of a scan copy + its permutation, padded (+inf / 0x7fffffff)
__global__ void prep_gather_sorted_kernel(const float* __restrict__ Qs, int m, int m_pad, const int32_t* __restrict__ perm,
                                          float* __restrict__ out, int32_t* __restrict__ perm_pad)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m_pad) return;
    const int j = k < m ? perm[k] : -1;
#pragma unroll
    for (int a = 0; a < 3; ++a) out[(size_t)a * m_pad + k] = j >= 0 ? Qs[(size_t)a * m_pad + j] : inf_<float>();
    perm_pad[k] = j >= 0 ? j : 0x7fffffff;
}

// slot -> point map of the moving cloud: the Morton order, padding slots keep themselves
__global__ void prep_slot_map_kernel(const int32_t* __restrict__ perm, int n, int n_pad, int32_t* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_pad) out[k] = k < n ? perm[k] : k;
}

__global__ void row_order_keys_kernel(unsigned int* __restrict__ hits, int rows, unsigned int* __restrict__ keys, int32_t* __restrict__ vals,
                                      unsigned long long* __restrict__ total_add, unsigned long long* __restrict__ total_zero)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int h = 0u;
    if (r < rows) {
        h = hits[r];
        hits[r] = 0u;                                          // (the next launch counts afresh)
        h = h > 0xfffffu ? 0xfffffu : h;
        keys[r] = 0xfffffu - h;                                // ascending sort of this = descending hits; ties keep the row order (stable)
        vals[r] = r;
    }
    // the sum of the counters (the target of the split rows derives from it): two words, this launch adds to one and clears
    // the other for the next
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += (unsigned int)__shfl_xor((int)h, off, 64);
    __shared__ unsigned int wsum[4];   // (one add per block: 1200 adds to one address were 17 us of every pass)
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t != 0ull) atomicAdd(total_add, t);
    }
    if (r == 0) *total_zero = 0ull;
}

// the roles of an ordered launch's blocks (NN_ORDER_*, icp_kernels.h): every block works the split of the NN_ORDER_HEAD heaviest
// rows out for itself (one scan of 1024 counters), block 0 writes their roles, all write the roles behind them
__global__ __launch_bounds__(1024) void row_roles_kernel(const unsigned int* __restrict__ keys, const int32_t* __restrict__ vals, int rows,
                                                         const unsigned long long* __restrict__ total, int min_part, int total_div, int32_t* __restrict__ roles)
{
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int head = rows < NN_ORDER_HEAD ? rows : NN_ORDER_HEAD;
    const unsigned int h = t < head ? 0xfffffu - keys[t] : 0u;
    auto block_sum = [&](int v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();   // (the words are free again)
        if (lane == 0) wsum[w] = v;
        __syncthreads();
        int s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += wsum[k];
        return s;
    };
    unsigned long long T = *total / (unsigned long long)(total_div > 0 ? total_div : 1024);
    if (T < (unsigned long long)min_part) T = (unsigned long long)min_part;
    int parts = t < head ? 1 : 0, E = head;
    if (min_part > 0) {
        for (int it = 0; it < 24; ++it) {   // (the target doubles until the parts fit the spare blocks: at most 20 times, the counters have 20 bits)
            unsigned int p = 1u;
            if ((unsigned long long)h > T) {
                const unsigned long long want = ((unsigned long long)h + T - 1ull) / T;   // >= 2
                p = want >= 64ull ? 64u : 1u << (32 - __builtin_clz((unsigned int)want - 1u));
            }
            parts = t < head ? (int)p : 0;
            E = block_sum(parts);
            if (E - head <= NN_ORDER_EXTRA) break;
            T *= 2ull;
            parts = t < head ? 1 : 0;
            E = head;
        }
    }
    int v = parts;   // inclusive running sum within the wave, then across the waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o, 64); v += lane >= o ? u : 0; }
    __syncthreads();
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) base += k < w ? wsum[k] : 0;
    const int excl = base + v - parts;
    if (blockIdx.x == 0 && t < head) {
        const int row = vals[t];
        const int lg = 31 - __builtin_clz((unsigned int)parts);
        for (int p = 0; p < parts; ++p) roles[excl + p] = row | (p << NN_ROLE_ROW_BITS) | (lg << (NN_ROLE_ROW_BITS + NN_ROLE_PART_BITS));
    }
    const int j = E + (int)blockIdx.x * 1024 + t;   // the roles behind the head: one block each, in the sorted order
    if (j < rows + NN_ORDER_EXTRA) {
        const int src = head + (j - E);
        roles[j] = src < rows ? vals[src] : -1;
    }
}

size_t row_order_temp_bytes(int rows)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned int*)nullptr, (unsigned int*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                    (unsigned int)(rows > 0 ? rows : 1), 0, 20);
    return bytes;
}

hipError_t launch_row_order(const RowOrderBuffers& b, unsigned int* hits, int rows, const int32_t** roles_out, hipStream_t st)
{
    if (rows <= 0 || rows >= (1 << NN_ROLE_ROW_BITS) || b.roles == nullptr || b.totals == nullptr) return hipErrorInvalidValue;
    unsigned long long* tot = b.totals + (b.seq & 1ull);
    hipLaunchKernelGGL(row_order_keys_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, hits, rows, b.keys[0], b.vals[0], tot, b.totals + ((b.seq + 1ull) & 1ull));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t bytes = b.temp_bytes;
    e = rocprim::radix_sort_pairs(b.temp, bytes, b.keys[0], b.keys[1], b.vals[0], b.vals[1], (unsigned int)rows, 0, 20, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(row_roles_kernel, dim3((rows + NN_ORDER_EXTRA + 1023) / 1024), dim3(1024), 0, st, b.keys[1], b.vals[1], rows,
                       (const unsigned long long*)tot, b.min_part, b.total_div, b.roles);
    *roles_out = b.roles;
    return hipGetLastError();
}

size_t prep_sort_temp_bytes(int count)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned int*)nullptr, (unsigned int*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                    (unsigned int)(count > 0 ? count : 1));
    return bytes;
}

static hipError_t sort_pairs(const PrepBuffers& b, int count, int from, int end_bit, hipStream_t st)
{
    size_t bytes = b.temp_bytes;
    return rocprim::radix_sort_pairs(b.temp, bytes, b.keys[from], b.keys[from ^ 1], b.vals[from], b.vals[from ^ 1], (unsigned int)count, 0,
                                     (unsigned int)end_bit, st);
}

// voided[j] = 1 for every point with an exact lower-index twin, *count_dev += their number; then the scan copy
hipError_t launch_duplicates_and_scan_copy(const PrepBuffers& b, const float* X, int n, int n_pad, unsigned char* voided, int* count_dev,
                                           float* scan_out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const dim3 blk(256), grd((n + 255) / 256);
    int cur = 0;
    for (int axis = 2; axis >= 0; --axis) {   // least significant key first, stable passes
        hipLaunchKernelGGL(prep_axis_keys_kernel, grd, blk, 0, st, X, n, n_pad, axis, axis == 2 ? (const int32_t*)nullptr : b.vals[cur],
                           b.keys[cur], b.vals[cur]);
        if (hipError_t e = sort_pairs(b, n, cur, 32, st)) return e;
        cur ^= 1;
    }
    hipLaunchKernelGGL(prep_mark_duplicates_kernel, grd, blk, 0, st, X, n, n_pad, b.vals[cur], voided, count_dev);
    hipLaunchKernelGGL(prep_scan_copy_kernel, dim3((n_pad + 255) / 256), blk, 0, st, X, n, n_pad, voided, scan_out);
    return hipGetLastError();
}

// the same for a cloud in double (six stable 32-bit passes: z low, z high, y low, ...)
hipError_t launch_duplicates_and_scan_copy_f64(const PrepBuffers& b, const double* X, int n, int n_pad, unsigned char* voided, int* count_dev,
                                               double* scan_out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const dim3 blk(256), grd((n + 255) / 256);
    int cur = 0;
    bool first = true;
    for (int axis = 2; axis >= 0; --axis)
        for (int high = 0; high < 2; ++high) {
            hipLaunchKernelGGL(prep_axis_keys_f64_kernel, grd, blk, 0, st, X, n, n_pad, axis, high, first ? (const int32_t*)nullptr : b.vals[cur],
                               b.keys[cur], b.vals[cur]);
            if (hipError_t e = sort_pairs(b, n, cur, 32, st)) return e;
            cur ^= 1;
            first = false;
        }
    hipLaunchKernelGGL(prep_mark_duplicates_f64_kernel, grd, blk, 0, st, X, n, n_pad, b.vals[cur], voided, count_dev);
    hipLaunchKernelGGL(prep_scan_copy_f64_kernel, dim3((n_pad + 255) / 256), blk, 0, st, X, n, n_pad, voided, scan_out);
    return hipGetLastError();
}

// perm_out[k] = k-th point in Morton order; totals[0] / totals[1] = summed group extents of the given / the Morton order
// (group2 > 0: totals[2] / totals[3] = the same for groups of group2 -- the upper search level of a large model)
hipError_t launch_morton_order(const PrepBuffers& b, const float* X, int n, int n_pad, int group, int group2, int32_t* perm_out,
                               double* totals, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const dim3 blk(256), grd((n + 255) / 256);
    {
        // (large clouds: a grid of blocks; their partial boxes lie in the sort's second key buffer, idle until the sort)
        const int bb = n >= (1 << 18) ? 256 : 1;
        float* partial = bb > 1 ? reinterpret_cast<float*>(b.keys[1]) : nullptr;
        hipLaunchKernelGGL(prep_bbox_kernel, dim3(bb), dim3(1024), 0, st, X, n, n_pad, b.box, partial);
        if (bb > 1) hipLaunchKernelGGL(prep_bbox_final_kernel, dim3(1), dim3(64), 0, st, (const float*)partial, bb, b.box);
    }
    const char* env_order = getenv("ICP_ORDER");   // (not cached: the tests switch it between contexts)
    const int hilbert = (env_order && env_order[0] == 'm') ? 0 : 1;
    hipLaunchKernelGGL(prep_morton_keys_kernel, grd, blk, 0, st, X, n, n_pad, b.box, b.keys[0], b.vals[0], hilbert);
    if (hipError_t e = sort_pairs(b, n, 0, 31, st)) return e;
    if (hipError_t e = hipMemcpyAsync(perm_out, b.vals[1], (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st)) return e;
    const int groups = (n + group - 1) / group;
    const dim3 ggrd(groups);
    hipLaunchKernelGGL(prep_group_extent_kernel, ggrd, dim3(64), 0, st, X, n, n_pad, (const int32_t*)nullptr, group, b.ext);
    hipLaunchKernelGGL(prep_sum_kernel, dim3(1), dim3(256), 0, st, b.ext, groups, totals, 0);
    hipLaunchKernelGGL(prep_group_extent_kernel, ggrd, dim3(64), 0, st, X, n, n_pad, (const int32_t*)perm_out, group, b.ext);
    hipLaunchKernelGGL(prep_sum_kernel, dim3(1), dim3(256), 0, st, b.ext, groups, totals, 1);
    if (group2 > 0) {
        const int groups2 = (n + group2 - 1) / group2;
        hipLaunchKernelGGL(prep_group_extent_kernel, dim3(groups2), dim3(64), 0, st, X, n, n_pad, (const int32_t*)nullptr, group2, b.ext);
        hipLaunchKernelGGL(prep_sum_kernel, dim3(1), dim3(256), 0, st, b.ext, groups2, totals, 2);
        hipLaunchKernelGGL(prep_group_extent_kernel, dim3(groups2), dim3(64), 0, st, X, n, n_pad, (const int32_t*)perm_out, group2, b.ext);
        hipLaunchKernelGGL(prep_sum_kernel, dim3(1), dim3(256), 0, st, b.ext, groups2, totals, 3);
    }
    return hipGetLastError();
}

hipError_t launch_gather_sorted(const float* Qs, int m, int m_pad, const int32_t* perm, float* out, int32_t* perm_pad, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    hipLaunchKernelGGL(prep_gather_sorted_kernel, dim3((m_pad + 255) / 256), dim3(256), 0, st, Qs, m, m_pad, perm, out, perm_pad);
    return hipGetLastError();
}

hipError_t launch_slot_map(const int32_t* perm, int n, int n_pad, int32_t* out, hipStream_t st)
{
    if (n_pad <= 0) return hipSuccess;
    hipLaunchKernelGGL(prep_slot_map_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, st, perm, n, n_pad, out);
    return hipGetLastError();
}

// bounding box of every 8-point chunk of the duplicate-voided scan copy (voided = +inf entries are ignored; an
// all-void chunk gets lo = +inf, hi = -inf and is skipped by construction).  Once per model.
__global__ void model_boxes_kernel(const float* __restrict__ Qs, int m_pad, float* __restrict__ boxes)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c * 8 >= m_pad) return;
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = inf_<float>(); hi[a] = -inf_<float>(); }
    for (int k = 0; k < 8; ++k) {
        const int j = c * 8 + k;
        if (j >= m_pad) break;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = Qs[(size_t)a * m_pad + j];
            if (v < inf_<float>() && v > -inf_<float>()) { lo[a] = __builtin_fminf(lo[a], v); hi[a] = __builtin_fmaxf(hi[a], v); }
        }
    }
    float* o = boxes + (size_t)c * 8;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = 0.f; o[7] = 0.f;
}

// one representative per chunk of the scan copy (its first point that is not voided; +inf if there is none),
// SoA over round_up(m_pad / 8, 8) entries: the thinned-out model of the sparse kernel's cold start
__global__ void model_samples_kernel(const float* __restrict__ Qs, int m_pad, int ns_pad, float* __restrict__ samples)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ns_pad) return;
    float v[3] = {inf_<float>(), inf_<float>(), inf_<float>()};
    for (int k = 0; k < 8; ++k) {
        const int j = c * 8 + k;
        if (j >= m_pad) break;
        const float x = Qs[j];
        if (x < inf_<float>() && x > -inf_<float>()) { v[0] = x; v[1] = Qs[(size_t)m_pad + j]; v[2] = Qs[2 * (size_t)m_pad + j]; break; }
    }
    samples[c] = v[0];
    samples[(size_t)ns_pad + c] = v[1];
    samples[2 * (size_t)ns_pad + c] = v[2];
}

// the same two tables in double, for the fp64 form of the search (the model itself is the scan copy: nothing is voided)
__global__ void model_boxes_f64_kernel(const double* __restrict__ Qs, int m_pad, double* __restrict__ boxes)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c * 8 >= m_pad) return;
    double lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = inf_<double>(); hi[a] = -inf_<double>(); }
    for (int k = 0; k < 8; ++k) {
        const int j = c * 8 + k;
        if (j >= m_pad) break;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double v = Qs[(size_t)a * m_pad + j];
            if (v < inf_<double>() && v > -inf_<double>()) { lo[a] = __builtin_fmin(lo[a], v); hi[a] = __builtin_fmax(hi[a], v); }
        }
    }
    double* o = boxes + (size_t)c * 8;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = 0.0; o[7] = 0.0;
}

__global__ void model_samples_f64_kernel(const double* __restrict__ Qs, int m_pad, int ns_pad, double* __restrict__ samples)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ns_pad) return;
    double v[3] = {inf_<double>(), inf_<double>(), inf_<double>()};
    for (int k = 0; k < 8; ++k) {
        const int j = c * 8 + k;
        if (j >= m_pad) break;
        const double x = Qs[j];
        if (x < inf_<double>() && x > -inf_<double>()) { v[0] = x; v[1] = Qs[(size_t)m_pad + j]; v[2] = Qs[2 * (size_t)m_pad + j]; break; }
    }
    samples[c] = v[0];
    samples[(size_t)ns_pad + c] = v[1];
    samples[2 * (size_t)ns_pad + c] = v[2];
}

hipError_t launch_model_tables_f64(const void* Q_soa, int m_pad, void* boxes, void* samples, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    const int chunks = (m_pad + 7) / 8, ns_pad = (chunks + 7) / 8 * 8;
    hipLaunchKernelGGL(model_boxes_f64_kernel, dim3((chunks + 255) / 256), dim3(256), 0, st, (const double*)Q_soa, m_pad, (double*)boxes);
    hipLaunchKernelGGL(model_samples_f64_kernel, dim3((ns_pad + 255) / 256), dim3(256), 0, st, (const double*)Q_soa, m_pad, ns_pad, (double*)samples);
    return hipGetLastError();
}
size_t model_boxes_f64_bytes(int m_pad) { return (size_t)((m_pad + 7) / 8) * 8 * sizeof(double); }
size_t model_samples_f64_bytes(int m_pad) { return 3 * (size_t)(((m_pad / 8) + 7) / 8 * 8) * sizeof(double); }

size_t model_samples_bytes(int m_pad) { return 3 * (size_t)(((m_pad / 8) + 7) / 8 * 8) * sizeof(float); }

hipError_t launch_model_samples(const void* Qs_soa, int m_pad, float* samples, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    const int ns_pad = ((m_pad / 8) + 7) / 8 * 8;
    hipLaunchKernelGGL(model_samples_kernel, dim3((ns_pad + 255) / 256), dim3(256), 0, st, (const float*)Qs_soa, m_pad, ns_pad, samples);
    return hipGetLastError();
}

// upper levels: the box of every 64 boxes of the level below (512, 32 768 model points), one wave per box; stored
// behind the chunk boxes, level after level
__global__ __launch_bounds__(256) void model_superboxes_kernel(const float* __restrict__ boxes, int chunks, int supers, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int sidx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sidx >= supers) return;
    const int c = sidx * 64 + lane;
    float lo[3] = {inf_<float>(), inf_<float>(), inf_<float>()}, hi[3] = {-inf_<float>(), -inf_<float>(), -inf_<float>()};
    if (c < chunks) {
        const float4* bp = reinterpret_cast<const float4*>(boxes + (size_t)c * 8);
        const float4 b0 = bp[0], b1 = bp[1];
        lo[0] = b0.x; lo[1] = b0.y; lo[2] = b0.z; hi[0] = b0.w; hi[1] = b1.x; hi[2] = b1.y;
    }
    wave_box(lo, hi);
    if (lane == 0) {
        float* o = out + (size_t)sidx * 8;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = 0.f; o[7] = 0.f;
    }
}

size_t model_boxes_bytes(int m_pad)
{
    const size_t chunks = (size_t)(m_pad + 7) / 8, supers = (chunks + 63) / 64, thirds = (supers + 63) / 64;
    return (chunks + supers + thirds) * 8 * sizeof(float);
}

hipError_t launch_model_boxes(const void* Qs_soa, int m_pad, float* boxes, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    const int chunks = (m_pad + 7) / 8;
    hipLaunchKernelGGL(model_boxes_kernel, dim3((chunks + 255) / 256), dim3(256), 0, st, (const float*)Qs_soa, m_pad, boxes);
    const int supers = (chunks + 63) / 64;
    float* l2 = boxes + (size_t)chunks * 8;
    hipLaunchKernelGGL(model_superboxes_kernel, dim3((supers + 3) / 4), dim3(256), 0, st, (const float*)boxes, chunks, supers, l2);
    const int thirds = (supers + 63) / 64;
    hipLaunchKernelGGL(model_superboxes_kernel, dim3((thirds + 3) / 4), dim3(256), 0, st, (const float*)l2, supers, thirds,
                       l2 + (size_t)supers * 8);
    return hipGetLastError();
}

// One record per chunk for the hierarchical search of a large model: {box lo.xyz hi.x | hi.yz - - | x[8] | y[8] | z[8] | index[8]}
// = 40 words = 160 contiguous bytes, the layout of a hit's stage in LDS.  A hit is then two cache lines instead of five
// 32-byte pieces of five arrays (a model of millions of points is not L2-resident: 10 M x 10 M fetched 23 GB per early pass).
__global__ void model_records_kernel(const float* __restrict__ Qs, const float* __restrict__ boxes, const int32_t* __restrict__ perm, int m_pad,
                                     float* __restrict__ rec)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte piece each
    const long long chunks = m_pad >> 3;
    if (t >= chunks * 10) return;
    const long long ch = t / 10;
    const int part = (int)(t % 10);
    float4 v;
    if (part < 2) v = *reinterpret_cast<const float4*>(boxes + ch * 8 + part * 4);
    else if (part < 8) v = *reinterpret_cast<const float4*>(Qs + (size_t)((part - 2) >> 1) * m_pad + ch * 8 + (part & 1) * 4);
    else {
        const int k0 = (int)(ch * 8) + (part - 8) * 4;
        int4 iv = perm ? *reinterpret_cast<const int4*>(perm + k0) : int4{k0, k0 + 1, k0 + 2, k0 + 3};
        v = float4{__int_as_float(iv.x), __int_as_float(iv.y), __int_as_float(iv.z), __int_as_float(iv.w)};
    }
    *reinterpret_cast<float4*>(rec + ch * NN_REC_WORDS + part * 4) = v;
}

size_t model_records_bytes(int m_pad) { return (size_t)(m_pad >> 3) * NN_REC_WORDS * sizeof(float); }

hipError_t launch_model_records(const void* Qs_soa, const float* boxes, const int32_t* perm, int m_pad, float* rec, hipStream_t st)
{
    if (m_pad <= 0) return hipSuccess;
    const long long pieces = (long long)(m_pad >> 3) * 10;
    hipLaunchKernelGGL(model_records_kernel, dim3((unsigned int)((pieces + 255) / 256)), dim3(256), 0, st, (const float*)Qs_soa, boxes, perm, m_pad, rec);
    return hipGetLastError();
}

// lexicographic (d, j) minimum over the S segment partials: segments are ascending model ranges,
// so the first strict minimum in segment order is the lowest index.
template <typename F>
__device__ __forceinline__ int merge_partials(const F* __restrict__ part_d, const int32_t* __restrict__ part_idx,
                                              int S, int n_pad, int i)
{
    F best = part_d[i];
    int bi = part_idx[i];
    int s = 1;
    for (; s + 8 <= S; s += 8) {  // 16 independent loads in flight, then the ordered compare chain
        F d[8];
        int j[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            d[u] = part_d[(size_t)(s + u) * n_pad + i];
            j[u] = part_idx[(size_t)(s + u) * n_pad + i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (d[u] < best) { best = d[u]; bi = j[u]; }
    }
    if (s < S) {  // tail: same 16 loads in flight, out-of-range slots replaced by +inf
        F d[8];
        int j[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ss = s + u < S ? s + u : S - 1;
            d[u] = part_d[(size_t)ss * n_pad + i];
            j[u] = part_idx[(size_t)ss * n_pad + i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s + u < S && d[u] < best) { best = d[u]; bi = j[u]; }
    }
    return bi;
}

template <typename F>
__global__ void merge_kernel(const F* __restrict__ part_d, const int32_t* __restrict__ part_idx, int S, int n_pad,
                             int n, int m, int32_t* __restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int j = merge_partials<F>(part_d, part_idx, S, n_pad, i);
    idx[i] = j < m ? j : m - 1;  // unreachable clamp (padding never wins); keeps idx in range by construction
}

// ------------------------------------------------------------------------------------------------
// fused merge + gather + moments.  HBM-bound: per moving point 12 B (p) + 8*S B (partials)
// + 4 B (idx store) + 12 B gathered (q) [+ 12 B normals], accumulated in fp64.
// ------------------------------------------------------------------------------------------------
constexpr int MOM_BLOCK = 64;  // one wave per block: no LDS, no barrier; 256 blocks already at 16 384 points

template <typename F, int METRIC>
__global__ __launch_bounds__(MOM_BLOCK) void moments_kernel(const F* __restrict__ P, int n, int n_pad,
                                                            const F* __restrict__ Q, int m, int m_pad,
                                                            const F* __restrict__ Nrm,
                                                            const F* __restrict__ part_d,
                                                            const int32_t* __restrict__ part_idx, int S,
                                                            int32_t* __restrict__ idx_out,
                                                            double* __restrict__ partials, double tag,
                                                            const double* __restrict__ err_rows, int err_count)
{
    constexpr int NACC = (METRIC == ICP_POINT_TO_POINT) ? 18 : 28;
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;

    for (int i = blockIdx.x * MOM_BLOCK + threadIdx.x; i < n; i += gridDim.x * MOM_BLOCK) {
        int j = merge_partials<F>(part_d, part_idx, S, n_pad, i);
        j = j < m ? j : m - 1;
        idx_out[i] = j;
        const double px = (double)P[i], py = (double)P[(size_t)n_pad + i], pz = (double)P[2 * (size_t)n_pad + i];
        const double qx = (double)Q[j], qy = (double)Q[(size_t)m_pad + j], qz = (double)Q[2 * (size_t)m_pad + j];
        acc[0] += 1.0;
        if constexpr (METRIC == ICP_POINT_TO_POINT) {
            acc[1] += px; acc[2] += py; acc[3] += pz;
            acc[4] += qx; acc[5] += qy; acc[6] += qz;
            acc[7] += qx * px; acc[8] += qx * py; acc[9] += qx * pz;
            acc[10] += qy * px; acc[11] += qy * py; acc[12] += qy * pz;
            acc[13] += qz * px; acc[14] += qz * py; acc[15] += qz * pz;
            acc[16] += px * px + py * py + pz * pz;
            acc[17] += qx * qx + qy * qy + qz * qz;
        } else {
            const double nx = (double)Nrm[j], ny = (double)Nrm[(size_t)m_pad + j], nz = (double)Nrm[2 * (size_t)m_pad + j];
            double cn[6];
            cn[0] = py * nz - pz * ny;
            cn[1] = pz * nx - px * nz;
            cn[2] = px * ny - py * nx;
            cn[3] = nx; cn[4] = ny; cn[5] = nz;
            const double bi = (px - qx) * nx + (py - qy) * ny + (pz - qz) * nz;
            int o = 1;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c = a; c < 6; ++c) acc[o++] += cn[a] * cn[c];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[22 + a] -= cn[a] * bi;
        }
    }
    // slot 0 of the moment vector is the error of the preceding transform (written by finalize)
    block_sum_store<NACC, MOM_BLOCK>(acc, partials + (size_t)blockIdx.x * ICP_NMOM + 1);
    // slot 0: this block's share of the error rows the preceding transform (fused into the matching
    // kernel, or its own launch) left in device memory -- fixed assignment, fixed order
    if (threadIdx.x == 0) {
        double e = 0.0;
        for (int r = blockIdx.x; r < err_count; r += gridDim.x) e += err_rows[r];
        partials[(size_t)blockIdx.x * ICP_NMOM + ICP_MOM_ERR] = e;
    }
    // completion tag for a host that polls the (pinned, mapped) rows instead of synchronising the
    // stream: the row's data is released to system scope before the tag becomes visible
    static_assert(MOM_BLOCK == 64, "the tag protocol assumes one wave per block");
    __threadfence_system();
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * ICP_NMOM + (ICP_NMOM - 1)] = tag;
}

// ------------------------------------------------------------------------------------------------
// in-place transform + error.  HBM-bound: 12 B read + 12 B written per moving point, + 4 B idx
// + 12 B gathered q.  The products and sums are rounded separately in the storage precision
// ((r0*x + r1*y) + r2*z) + t, the association of RyT (src/ICP_point_to_point.cu:85).
// ------------------------------------------------------------------------------------------------
constexpr int TR_BLOCK = 256;

template <typename F>
__global__ __launch_bounds__(TR_BLOCK) void transform_error_kernel(F* __restrict__ P, int n, int n_pad, RT<F> rt,
                                                                    const F* __restrict__ Q, int m_pad,
                                                                    const int32_t* __restrict__ idx,
                                                                    double* __restrict__ err_partials)
{
    double acc[1] = {0.0};
    for (int i = blockIdx.x * TR_BLOCK + threadIdx.x; i < n_pad; i += gridDim.x * TR_BLOCK) {
        const F x = P[i], y = P[(size_t)n_pad + i], z = P[2 * (size_t)n_pad + i];
        F o[3];
        apply_rt<F>(rt, x, y, z, o[0], o[1], o[2]);
        P[i] = o[0];
        P[(size_t)n_pad + i] = o[1];
        P[2 * (size_t)n_pad + i] = o[2];
        if (i < n) {
            const int j = idx[i];
            const double dx = (double)Q[j] - (double)o[0];
            const double dy = (double)Q[(size_t)m_pad + j] - (double)o[1];
            const double dz = (double)Q[2 * (size_t)m_pad + j] - (double)o[2];
            acc[0] += dx * dx + dy * dy + dz * dz;
        }
    }
    block_sum_store<1, TR_BLOCK>(acc, err_partials + blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// finalize: one block, fixed-order sums of the per-block partials -> the ICP_NMOM vector.
// thread (k = tid % 32, part = tid / 32) sums blocks part, part+8, ... of slot k.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_kernel(double* __restrict__ mom, const double* __restrict__ mom_partials,
                                                       int mom_blocks, const double* __restrict__ err_partials,
                                                       int err_blocks, int rows_have_err)
{
    __shared__ double red[8][ICP_NMOM];
    const int k = threadIdx.x & 31, part = threadIdx.x >> 5;
    double s = 0.0;
    if (k == 0) {
        for (int b = part; b < err_blocks; b += 8) s += err_partials[b];
        if (rows_have_err)
            for (int b = part; b < mom_blocks; b += 8) s += mom_partials[(size_t)b * ICP_NMOM];
    } else if (k == ICP_NMOM - 1) {
        s = 0.0;  // the rows' completion-tag slot is not a moment
    } else {
        for (int b = part; b < mom_blocks; b += 8) s += mom_partials[(size_t)b * ICP_NMOM + k];
    }
    red[part][k] = s;
    __syncthreads();
    if (threadIdx.x < ICP_NMOM) {
        double tot = red[0][k];
#pragma unroll
        for (int p = 1; p < 8; ++p) tot += red[p][k];
        mom[k] = tot;
    }
}

// ------------------------------------------------------------------------------------------------
// point-to-plane front end: 4 nearest model neighbours of every model point (self dropped).
// One lane per model point, whole model streamed through LDS; a sorted (d, j) top-5 lives in
// registers, insertion happens under a (rare) wave-level branch.  Candidates arrive in ascending
// j, so "insert after every entry with d_e <= d" reproduces the reference's k+1 passes of
// first-arg-min with overwrite (src/CUDA/GPU_point_to_plane_real.cu:83-89).
// ------------------------------------------------------------------------------------------------
template <typename F, int TQ>
__global__ __launch_bounds__(NN_BLOCK) void knn4_kernel(const F* __restrict__ Q, int m, int m_pad,
                                                        int32_t* __restrict__ nbr)
{
    using V = typename Vec16<F>::type;
    constexpr int VN = Vec16<F>::N;
    __shared__ __attribute__((aligned(16))) F sq[3 * TQ];
    const int i = blockIdx.x * NN_BLOCK + threadIdx.x;
    const int is = i < m ? i : m - 1;
    const F px = Q[is], py = Q[(size_t)m_pad + is], pz = Q[2 * (size_t)m_pad + is];
    F bd[5];
    int bj[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) { bd[r] = inf_<F>(); bj[r] = 0; }

    for (int tile = 0; tile < m; tile += TQ) {
        const int len = min(TQ, m_pad - tile);
        __syncthreads();
        for (int e = threadIdx.x * VN; e < len; e += NN_BLOCK * VN) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                *reinterpret_cast<V*>(&sq[a * TQ + e]) =
                    *reinterpret_cast<const V*>(&Q[(size_t)a * m_pad + tile + e]);
        }
        __syncthreads();
        const int real = min(len, m - tile);  // padded duplicates must not enter a top-k
        for (int c = 0; c < real; ++c) {
            const F d = dist2<F>(px, py, pz, sq[c], sq[TQ + c], sq[2 * TQ + c]);
            if (d < bd[4]) {
                const int j = tile + c;
                // insert keeping (d, j) ascending; equal d keeps the earlier (lower) j first
                F cd = d;
                int cj = j;
                bool shifting = false;  // once the new entry is placed, everything below moves down one slot
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const bool sw = shifting || (cd < bd[r]);
                    shifting = sw;
                    const F td = bd[r];
                    const int tj = bj[r];
                    bd[r] = sw ? cd : td;
                    bj[r] = sw ? cj : tj;
                    cd = sw ? td : cd;
                    cj = sw ? tj : cj;
                }
            }
        }
    }
    if (i < m) {
#pragma unroll
        for (int r = 1; r < 5; ++r) nbr[(size_t)i * 4 + (r - 1)] = bj[r];
    }
}

// ------------------------------------------------------------------------------------------------
// kNN(4), fp32, v2: the matching kernel's machinery (packed distances over two query points per lane, four
// waves splitting the block's model segment, grid.y segments, 8-point chunks with a wave-uniform early-out)
// carrying a sorted (d, j) top-5 per query instead of a single minimum.  A chunk is examined element-wise only
// when some lane's chunk minimum beats that lane's threshold = min(5th best so far, seeded bound).  The seeded
// bound is the largest distance to five DISTINCT model points around the query's own index, bumped one ulp: at
// least five points lie strictly under it, so the exact top-5 survives; the seed only prunes work.
// Per-wave lists are merged through LDS (ties -> the lower wave = lower indices), per-segment lists by
// knn4_merge_kernel (ties -> the lower segment).  Rank 0 (self or an equal-distance lower index) is dropped there.
// ------------------------------------------------------------------------------------------------
struct Top5 {
    float d[5];
    int j[5];
};

__device__ __forceinline__ void top5_insert(Top5& L, float d, int j)
{
    float cd = d;
    int cj = j;
    bool shifting = false;  // once placed, everything below moves down one slot (keeps equal-d entries index-ordered)
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const bool sw = shifting || (cd < L.d[r]);
        shifting = sw;
        const float td = L.d[r];
        const int tj = L.j[r];
        L.d[r] = sw ? cd : td;
        L.j[r] = sw ? cj : tj;
        cd = sw ? td : cd;
        cj = sw ? tj : cj;
    }
}

constexpr int KNN_C = 8;

__global__ __launch_bounds__(NN_BLOCK, 4) void knn4_f32_v2(const float* __restrict__ Q, int m, int m_pad, int n_pad,
                                                           int seg_len, float* __restrict__ part_d,
                                                           int32_t* __restrict__ part_j)
{
    constexpr int C = KNN_C;
    __shared__ __attribute__((aligned(16))) float sq[4][3][NN2_TQW];
    __shared__ float ld[4][128][5];
    __shared__ int lj[4][128][5];

    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int wseg = seg_len >> 2;
    const int q0 = blockIdx.y * seg_len;
    const int my0 = q0 + w * wseg;
    const int my1 = min(my0 + wseg, m_pad);
    const int ibase = blockIdx.x * 128 + lane;
    const int i0 = min(ibase, m - 1), i1 = min(ibase + 64, m - 1);  // queries are model points; padding lanes repeat the last

    const f2 px = f2{Q[i0], Q[i1]}, py = f2{Q[(size_t)m_pad + i0], Q[(size_t)m_pad + i1]},
             pz = f2{Q[2 * (size_t)m_pad + i0], Q[2 * (size_t)m_pad + i1]};
    Top5 L[2];
    float bound[2], thr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 5; ++r) { L[t].d[r] = inf_<float>(); L[t].j[r] = 0x7fffffff; }
        const int i = t ? i1 : i0;
        const float x = t ? px.y : px.x, y = t ? py.y : py.x, z = t ? pz.y : pz.x;
        const int lo = max(0, min(i - 2, m - 5));  // five distinct indices around the query's own
        float mx = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int j = lo + k;
            mx = fmaxf(mx, dist2<float>(x, y, z, Q[j], Q[(size_t)m_pad + j], Q[2 * (size_t)m_pad + j]));
        }
        bound[t] = (mx < inf_<float>()) ? __uint_as_float(__float_as_uint(mx) + 1u) : mx;
        thr[t] = bound[t];
    }

    const int ntile = (wseg + NN2_TQW - 1) / NN2_TQW;
    for (int k = 0; k < ntile; ++k) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int v = threadIdx.x + r * NN_BLOCK;
            const int ww = v / 192, rem = v % 192;
            const int a = rem / 64, e = (rem % 64) * 4;
            const int off = k * NN2_TQW + e;
            const int src = q0 + ww * wseg + off;
            if (off < wseg && src < m_pad)
                *reinterpret_cast<float4*>(&sq[ww][a][e]) = *reinterpret_cast<const float4*>(&Q[(size_t)a * m_pad + src]);
        }
        __syncthreads();
        const int tile0 = my0 + k * NN2_TQW;
        const int len = min(NN2_TQW, my1 - tile0);
        for (int c = 0; c < len; c += C) {
            f2 dd[C];
            float cmin0 = inf_<float>(), cmin1 = inf_<float>();
#pragma unroll
            for (int kk = 0; kk < C; kk += 4) {
                const float4 qx4 = *reinterpret_cast<const float4*>(&sq[w][0][c + kk]);
                const float4 qy4 = *reinterpret_cast<const float4*>(&sq[w][1][c + kk]);
                const float4 qz4 = *reinterpret_cast<const float4*>(&sq[w][2][c + kk]);
                const f2 qxa = f2{qx4.x, qx4.y}, qxb = f2{qx4.z, qx4.w};
                const f2 qya = f2{qy4.x, qy4.y}, qyb = f2{qy4.z, qy4.w};
                const f2 qza = f2{qz4.x, qz4.y}, qzb = f2{qz4.z, qz4.w};
                dd[kk + 0] = pk_dist2<0>(qxa, qya, qza, px, py, pz);
                dd[kk + 1] = pk_dist2<1>(qxa, qya, qza, px, py, pz);
                dd[kk + 2] = pk_dist2<0>(qxb, qyb, qzb, px, py, pz);
                dd[kk + 3] = pk_dist2<1>(qxb, qyb, qzb, px, py, pz);
                cmin0 = fmin_(fmin_(cmin0, dd[kk].x), dd[kk + 1].x);
                cmin0 = fmin_(fmin_(cmin0, dd[kk + 2].x), dd[kk + 3].x);
                cmin1 = fmin_(fmin_(cmin1, dd[kk].y), dd[kk + 1].y);
                cmin1 = fmin_(fmin_(cmin1, dd[kk + 2].y), dd[kk + 3].y);
            }
            const bool need = (cmin0 < thr[0]) | (cmin1 < thr[1]);
            if (__builtin_amdgcn_ballot_w64(need) == 0ull) continue;
#pragma unroll
            for (int kk = 0; kk < C; ++kk) {
                const int j = tile0 + c + kk;
                const bool real = j < m;  // padded duplicates of the last point must not enter a top-k
                if (real && dd[kk].x < thr[0]) { top5_insert(L[0], dd[kk].x, j); thr[0] = fmin_(bound[0], L[0].d[4]); }
                if (real && dd[kk].y < thr[1]) { top5_insert(L[1], dd[kk].y, j); thr[1] = fmin_(bound[1], L[1].d[4]); }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            ld[w][lane + t * 64][r] = L[t].d[r];
            lj[w][lane + t * 64][r] = L[t].j[r];
        }
    __syncthreads();
    if (threadIdx.x < 128) {
        // 4-way merge of sorted lists; on equal d the lower wave (lower indices) goes first
        int h[4] = {0, 0, 0, 0};
        const size_t o = ((size_t)blockIdx.y * n_pad + (size_t)blockIdx.x * 128 + threadIdx.x) * 5;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            float bd = inf_<float>();
            int bw = 0;
#pragma unroll
            for (int ww = 3; ww >= 0; --ww) {
                const float d = h[ww] < 5 ? ld[ww][threadIdx.x][h[ww]] : inf_<float>();
                if (d <= bd) { bd = d; bw = ww; }   // descending ww with <= : the lowest wave wins ties
            }
            const int hj = h[bw] < 5 ? lj[bw][threadIdx.x][h[bw]] : 0x7fffffff;
            part_d[o + r] = bd;
            part_j[o + r] = bd < inf_<float>() ? hj : 0x7fffffff;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) h[ww] += (ww == bw) ? 1 : 0;
        }
    }
}

// merge the S per-segment top-5 lists of every query (ascending segments, earlier segment first on equal d),
// drop rank 0, store the 4 neighbour indices
__global__ void knn4_merge_kernel(const float* __restrict__ part_d, const int32_t* __restrict__ part_j, int S, int n_pad,
                                  int m, int32_t* __restrict__ nbr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Top5 L;
#pragma unroll
    for (int r = 0; r < 5; ++r) { L.d[r] = part_d[(size_t)i * 5 + r]; L.j[r] = part_j[(size_t)i * 5 + r]; }
    for (int s = 1; s < S; ++s) {
        const size_t o = ((size_t)s * n_pad + i) * 5;
        for (int r = 0; r < 5; ++r) {
            const float d = part_d[o + r];
            if (!(d < L.d[4])) break;  // lists are sorted: nothing further in this segment can enter
            top5_insert(L, d, part_j[o + r]);
        }
    }
#pragma unroll
    for (int r = 1; r < 5; ++r) nbr[(size_t)i * 4 + (r - 1)] = L.j[r];
}

// PCA normal of every model point from its 4 neighbours, entirely on the device: float covariance in the
// order of src/CUDA/CPU_ICP_point_to-plane.cpp:217-246 (bar = sum * 0.25f, A += (x-bar)(y-bar), not divided by
// k), then a cyclic-Jacobi eigen-solve in fp64 registers (stands in for the reference's HOST loop of
// LAPACKE_ssyev, src/ICP_point_to_plane.cu:429-438) and the eigenvector of the eigenvalue of smallest magnitude
// (cblas_isamin over the ascending eigenvalues, first on ties).  Writes the padded SoA normal cloud directly.
template <typename F>
__global__ void normals_kernel(const F* __restrict__ Q, int m, int m_pad, const int32_t* __restrict__ nbr,
                               F* __restrict__ Nrm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m_pad) return;
    const int src = i < m ? i : m - 1;  // padding replicates the last point's normal (never referenced)
    float x[4], y[4], z[4];
    float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = nbr[(size_t)src * 4 + j];
        x[j] = (float)Q[s];
        y[j] = (float)Q[(size_t)m_pad + s];
        z[j] = (float)Q[2 * (size_t)m_pad + s];
        bx += x[j]; by += y[j]; bz += z[j];
    }
    const float qa = 1.0f / 4.0f;
    bx *= qa; by *= qa; bz *= qa;
    float A[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dx = x[j] - bx, dy = y[j] - by, dz = z[j] - bz;
        A[0] += dx * dx; A[1] += dx * dy; A[2] += dx * dz;
        A[3] += dy * dy; A[4] += dy * dz; A[5] += dz * dz;
    }
    // symmetric 3x3 in named scalars (no runtime-indexed arrays -> no scratch)
    double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = a01 * a01 + a02 * a02 + a12 * a12;
        const double dia = a00 * a00 + a11 * a11 + a22 * a22;
        if (off <= 1e-34 * dia || off == 0.0) break;
        // rotation (p,q) = (0,1): r = 2
        if (a01 != 0.0) {
            const double th = (a11 - a00) / (2.0 * a01);
            const double t = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            a00 -= t * a01; a11 += t * a01; a01 = 0.0;
            const double rp = a02, rq = a12;
            a02 = c * rp - s * rq; a12 = s * rp + c * rq;
            double p, q;
            p = v00; q = v01; v00 = c * p - s * q; v01 = s * p + c * q;
            p = v10; q = v11; v10 = c * p - s * q; v11 = s * p + c * q;
            p = v20; q = v21; v20 = c * p - s * q; v21 = s * p + c * q;
        }
        // (0,2): r = 1
        if (a02 != 0.0) {
            const double th = (a22 - a00) / (2.0 * a02);
            const double t = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            a00 -= t * a02; a22 += t * a02; a02 = 0.0;
            const double rp = a01, rq = a12;
            a01 = c * rp - s * rq; a12 = s * rp + c * rq;
            double p, q;
            p = v00; q = v02; v00 = c * p - s * q; v02 = s * p + c * q;
            p = v10; q = v12; v10 = c * p - s * q; v12 = s * p + c * q;
            p = v20; q = v22; v20 = c * p - s * q; v22 = s * p + c * q;
        }
        // (1,2): r = 0
        if (a12 != 0.0) {
            const double th = (a22 - a11) / (2.0 * a12);
            const double t = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            a11 -= t * a12; a22 += t * a12; a12 = 0.0;
            const double rp = a01, rq = a02;
            a01 = c * rp - s * rq; a02 = s * rp + c * rq;
            double p, q;
            p = v01; q = v02; v01 = c * p - s * q; v02 = s * p + c * q;
            p = v11; q = v12; v11 = c * p - s * q; v12 = s * p + c * q;
            p = v21; q = v22; v21 = c * p - s * q; v22 = s * p + c * q;
        }
    }
    // ascending eigenvalues (stable w.r.t. the original slot), then the first of smallest |w| as floats
    double w0 = a00, w1 = a11, w2 = a22;
    double e0x = v00, e0y = v10, e0z = v20, e1x = v01, e1y = v11, e1z = v21, e2x = v02, e2y = v12, e2z = v22;
#define ICP_SWAP_EIG(wa, ax, ay, az, wb, bx_, by_, bz_) \
    if (wb < wa) { double tw = wa; wa = wb; wb = tw; double tx = ax; ax = bx_; bx_ = tx; double ty = ay; ay = by_; by_ = ty; double tz = az; az = bz_; bz_ = tz; }
    ICP_SWAP_EIG(w0, e0x, e0y, e0z, w1, e1x, e1y, e1z)
    ICP_SWAP_EIG(w0, e0x, e0y, e0z, w2, e2x, e2y, e2z)
    ICP_SWAP_EIG(w1, e1x, e1y, e1z, w2, e2x, e2y, e2z)
#undef ICP_SWAP_EIG
    double nx = e0x, ny = e0y, nz = e0z;
    float wm = fabsf((float)w0);
    if (fabsf((float)w1) < wm) { wm = fabsf((float)w1); nx = e1x; ny = e1y; nz = e1z; }
    if (fabsf((float)w2) < wm) { nx = e2x; ny = e2y; nz = e2z; }
    Nrm[i] = (F)nx;
    Nrm[(size_t)m_pad + i] = (F)ny;
    Nrm[2 * (size_t)m_pad + i] = (F)nz;
}

// ------------------------------------------------------------------------------------------------
// OS1-16 polar -> Cartesian (mm), one range per lane
// ------------------------------------------------------------------------------------------------
__global__ void os1_conversion_kernel(const uint32_t* __restrict__ r, int n, uint32_t encoder0,
                                      const float* __restrict__ altitude, const float* __restrict__ azimuth,
                                      float* __restrict__ xyz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int azimuth_block = i / 16, channel = i % 16;
    const unsigned long long counter = ((unsigned long long)encoder0 + (unsigned long long)azimuth_block * 88ull) % 90112ull;
    const float theta = (float)(2.0 * M_PI * ((double)counter / 90112.0 + (double)azimuth[channel] / 360.0));
    const float phi = (float)(2.0 * M_PI * (double)altitude[channel] / 360.0);
    const float rr = (float)r[i];
    const float ct = cosf(theta), st = sinf(theta), cp = cosf(phi), sp = sinf(phi);
    xyz[3 * (size_t)i + 0] = rr * ct * cp;
    xyz[3 * (size_t)i + 1] = -rr * st * cp;
    xyz[3 * (size_t)i + 2] = rr * sp;
}

// raw OS1-16 packets (12 608 B each: 16 azimuth blocks x [16 B header | 64 channels x 12 B | 4 B status]) ->
// ranges [mm] + Cartesian points [mm] in one pass; one lane per (packet, block, beam).  Replaces the host
// parse loop + H2D + Conversion of src/CUDA/GPU_point_to_point_real.cu:457-487,538-563.  Byte-granular reads
// (the 20-bit range sits at an arbitrary byte offset); 3 bytes per lane, ~0.8 MB for the hall dump.
__global__ void os1_packets_kernel(const uint8_t* __restrict__ packets, int n_packets, const float* __restrict__ altitude,
                                   const float* __restrict__ azimuth, uint32_t* __restrict__ ranges,
                                   float* __restrict__ xyz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_packets * 256) return;
    const int packet = i / 256, blk = (i / 16) % 16, beam = i % 16;
    const int ch = 2 + 4 * beam;  // the 16 lasers of an OS1-16 sit in channels 2, 6, ..., 62
    const size_t w = (size_t)packet * 12608 + (size_t)blk * 788 + 16 + 12 * (size_t)ch;
    const uint32_t r = (uint32_t)packets[w] | ((uint32_t)packets[w + 1] << 8) | (((uint32_t)packets[w + 2] & 0xFu) << 16);
    const uint32_t encoder0 = (uint32_t)packets[12] | ((uint32_t)packets[13] << 8);  // first block of the first packet
    ranges[i] = r;
    const int azimuth_block = i / 16;
    const unsigned long long counter = ((unsigned long long)encoder0 + (unsigned long long)azimuth_block * 88ull) % 90112ull;
    const float theta = (float)(2.0 * M_PI * ((double)counter / 90112.0 + (double)azimuth[beam] / 360.0));
    const float phi = (float)(2.0 * M_PI * (double)altitude[beam] / 360.0);
    const float rr = (float)r;
    const float ct = cosf(theta), st = sinf(theta), cp = cosf(phi), sp = sinf(phi);
    xyz[3 * (size_t)i + 0] = rr * ct * cp;
    xyz[3 * (size_t)i + 1] = -rr * st * cp;
    xyz[3 * (size_t)i + 2] = rr * sp;
}

// ------------------------------------------------------------------------------------------------
// launch geometry + launchers
// ------------------------------------------------------------------------------------------------
template <typename F> struct NNCfg;
template <> struct NNCfg<float> { static constexpr int T = 4; static constexpr int TQ = 2048; };
template <> struct NNCfg<double> { static constexpr int T = 2; static constexpr int TQ = 1024; };

// tuning knobs (read once): ICP_NN_T = points per lane {1,2,4,8}, ICP_NN_SPLITS = forced segment
// count, ICP_NN_BLOCKS_PER_CU = occupancy target used to derive the segment count.
static float env_float(const char* name, float dflt)
{
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    return (float)atof(v);
}

static int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    return atoi(v);
}

unsigned int share_rows_plan(const unsigned int* hits, int rows, int blocks, int m_pad, int min_hits, int* parts_out)
{
    // (statement by statement what a block of nn_match_sparse computes; the block's reductions are plain loops here)
    unsigned int total = 0;
    for (int r = 0; r < rows; ++r) total += share_clamp(hits[r]);
    const unsigned int spare = blocks > rows ? (unsigned int)(blocks - rows) : 0u;
    const unsigned int T0 = share_first_target(total, spare), cap = share_cap(m_pad), Tmin = (unsigned int)min_hits;
    unsigned int T = T0 < Tmin ? Tmin : T0;
    if (share_tries_candidates(T0, Tmin, spare)) {
        unsigned int sums[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; ++k) {
            for (int r = 0; r < rows; ++r) sums[k] += share_parts(share_clamp(hits[r]), share_candidate(T0, k), cap);
            sums[k] &= 0xffffu;   // (16-bit fields in the kernel; <= 512 x 32 never reaches them)
        }
        T = share_pick(T0, Tmin, sums, (unsigned int)blocks);
    }
    long long all = 0;
    for (int r = 0; r < rows; ++r) { parts_out[r] = (int)share_parts(share_clamp(hits[r]), T, cap); all += parts_out[r]; }
    if (all > blocks)
        for (int r = 0; r < rows; ++r) parts_out[r] = 1;   // (cannot happen with clamped counts and blocks >= rows; the kernel falls back the same way)
    return T;
}

NNPlan nn_plan(int n, int m, int precision, int num_cus, int force_dense)
{
    NNPlan pl{};
    pl.precision = precision;
    pl.n = n;
    pl.m = m;
    pl.n_pad = pad_moving(n);
    pl.m_pad = pad_model(m);
    static const int env_T = env_int("ICP_NN_T", 0);
    static const int env_S = env_int("ICP_NN_SPLITS", 0);
    static const int env_bpc = env_int("ICP_NN_BLOCKS_PER_CU", 0);
    static const int env_v1 = env_int("ICP_NN_V1", 0);
    static const int env_C = env_int("ICP_NN_CHUNK", 0);
    if (num_cus <= 0) num_cus = 256;
    pl.version = (precision == ICP_F32 && !env_v1) ? 2 : 1;
    pl.chunk = NN_CHUNK;
    if (pl.version == 2) {
        // v2: a block (4 waves) owns 64*T moving points, each wave a quarter of the block's segment.
        // 8 resident waves per SIMD = 8 blocks per CU saturate the VALU (valu_rate probe).
        static const int env_sparse = env_int("ICP_NN_SPARSE", 1);
        static const int env_boxes = env_int("ICP_NN_BOXES", 1);
        if (env_sparse && env_boxes && !force_dense) {
            // sparse kernel: a block of 16 waves owns 128 moving points; split the model only while there are
            // fewer blocks than CUs, and never below 1024 model points per block
            pl.sparse = 1;
            pl.cull = 1;
            pl.chunk = 8;
            pl.pts_per_thread = 2;
            pl.row = 128;
            pl.blocks_x = pl.n_pad / 128;
            if (n <= 0 || m <= 0) { pl.splits = 0; pl.seg_len = 0; return pl; }
            {
                // 64-point rows (nn_match_row64: 8 waves per block, one point per lane, one segment) for clouds that cannot
                // fill the machine with 128-point rows and whose model is searched flat; ICP_NN_ROW = 64 / 128 overrides
                const int env_hier0 = env_int("ICP_NN_HIER", -1);
                const bool hier0 = env_hier0 >= 0 ? env_hier0 != 0 : pl.m_pad >= (1 << 17);
                const int env_row = env_int("ICP_NN_ROW", 0);   // (not cached: the tests switch it between contexts)
                // (two 8-wave blocks fit a CU: 512 rows of 64 = 32 768 points can stay on the machine for a whole registration)
                const bool row64 = !hier0 && (env_row == 64 || (env_row != 128 && pl.n_pad / 64 <= 2 * num_cus));
                if (row64) {
                    pl.row = 64;
                    pl.blocks_x = pl.n_pad / 64;
                    pl.hier = 0;
                    pl.splits = 1;
                    pl.seg_len = round_up(pl.m_pad, 8);
                    return pl;
                }
            }
            // (an unsplit row closes without the key/ticket exchange, worth ~3 us: prefer it from half a machine up)
            int S = (num_cus / 2 + pl.blocks_x - 1) / pl.blocks_x;
            const int max_S = (pl.m_pad + 1023) / 1024;
            if (S > max_S) S = max_S;
            if (env_S > 0) S = env_S;
            if (S < 1) S = 1;
            // large models are searched in two levels (boxes of 64 chunks first): from 2^17 points up, where the
            // flat pass over the chunk boxes starts to dominate (ICP_NN_HIER = 0 / 1 overrides)
            const int env_hier = env_int("ICP_NN_HIER", -1);   // (not cached: the tests switch it between contexts)
            // (round 2: with 16 hits per fetch and the rows taken heaviest first the hierarchy pays from 2^16 model points when
            // the cloud has more rows than shared 8-wave blocks could serve -- 90 000^2: 101.9 -> 87.4 us per iteration,
            // 131 044^2: 152.3 -> 117.9, 65 536^2: 80.6 -> 77.6)
            pl.hier = env_hier >= 0 ? (env_hier ? 1 : 0)
                                    : ((pl.m_pad >= (1 << 17) || (pl.m_pad >= (1 << 16) && pl.blocks_x > 2 * num_cus - num_cus / 4)) ? 1 : 0);
            if ((pl.m_pad >> 3) > 65536) pl.hier = 1;   // (the flat search lists 16-bit chunk numbers)
            int seg = round_up((pl.m_pad + S - 1) / S, pl.hier ? 512 : 8);   // (a segment starts on a super-box boundary)
            S = (pl.m_pad + seg - 1) / seg;
            pl.splits = S;
            pl.seg_len = seg;
            // Rows of 128 that outnumber the CUs (one 16-wave block each: a second round of blocks) but fit the machine as
            // 8-wave blocks, two to a CU: the 8-wave form, and -- one launch per pass -- the blocks the machine has room for
            // beyond the rows go to the heavy rows (shared rows, see nn_match_sparse).  ICP_NN_WAVES128 = 8 / 16 and
            // ICP_NN_SHARE = 0 override (not cached: the tests switch them between contexts).
            const int env_w128 = env_int("ICP_NN_WAVES128", 0), env_share = env_int("ICP_NN_SHARE", 1);
            pl.nw = 16;
            // (without spare blocks the 8-wave form loses: 65 536 points = 512 rows, 88 us per iteration against 79 with 16 waves
            // in two rounds; with an eighth of the machine to spare it wins -- 50 176 points: 39.6 against 53.4)
            // The hierarchical search with rows for several rounds of blocks runs them as 8-wave blocks as well, two to a CU: late in
            // a registration a block is a chain of short dependent steps (front end, three levels of boxes, a handful of hits, the
            // row's close: ~19 us for a median of 110 hits) and a second block on the CU fills the waits of the first -- 10 M x 10 M on
            // one GPU: 11.2 -> 8.3 ms per iteration, every pass faster (the first 35.0 -> 33.9 ms, the thirtieth 5.4 -> 3.3)
            // ... and 4-wave blocks, four to a CU: 8.4 -> 7.5 ms (the thirtieth pass 3.2 -> 2.5 ms; the first, cold, stays on 8 waves)
            if (pl.hier && S == 1 && env_w128 != 16 && (env_w128 == 8 || env_w128 == 4 || pl.blocks_x >= 2 * num_cus)) pl.nw = env_w128 == 8 ? 8 : 4;
            if (!pl.hier && S == 1 && (env_w128 == 8 || (env_w128 != 16 && pl.blocks_x > num_cus && pl.blocks_x <= 2 * num_cus - num_cus / 4))) {
                pl.nw = 8;
                if (env_share && pl.blocks_x < 2 * num_cus && pl.blocks_x <= 8 * 64) pl.share_blocks = 2 * num_cus;
            }
            // large models (hierarchical search), at least two rounds of blocks: the rows are taken heaviest first (launch_row_order)
            // (ICP_NN_ORDER = 0: index order; 2: also where the rows are few -- the parity tests; not cached)
            {
                const int env_order = env_int("ICP_NN_ORDER", 1);
                pl.order = (pl.hier && S == 1 && env_order && (env_order == 2 || pl.blocks_x >= 2 * num_cus) && pl.blocks_x < (1 << NN_ROLE_ROW_BITS)) ? 1 : 0;   // (a role holds 21 bits of row)
            }
            return pl;
        }
        const int bpc = env_bpc > 0 ? env_bpc : 8;
        const int target_blocks = num_cus * bpc;
        int T = (pl.n_pad / 256 >= target_blocks) ? 4 : 2;   // big clouds: 4 points per lane halve the LDS reads
        if (env_T == 2 || env_T == 4) T = env_T;
        pl.chunk = (env_C == 8 || env_C == 16) ? env_C : 16;
        static const int env_cull = env_int("ICP_NN_CULL", 1);
        pl.cull = (T == 2 && env_cull) ? 1 : 0;
        if (pl.cull && env_C == 0) pl.chunk = 8;  // 16 partial sums per chunk would spill under the 64-VGPR cap
        pl.pts_per_thread = T;
        pl.blocks_x = pl.n_pad / (64 * T);
        if (n <= 0 || m <= 0) { pl.splits = 0; pl.seg_len = 0; return pl; }
        const int gran = 4 * pl.chunk;                        // four wave quarters of whole chunks
        int S = (target_blocks + pl.blocks_x - 1) / pl.blocks_x;
        const int max_S = (pl.m_pad + 511) / 512;             // keep >= 128 model points per wave
        if (S > max_S) S = max_S;
        if (env_S > 0) S = env_S;
        if (S < 1) S = 1;
        int seg = round_up((pl.m_pad + S - 1) / S, gran);
        S = (pl.m_pad + seg - 1) / seg;
        pl.splits = S;
        pl.seg_len = seg;
        return pl;
    }
    if (precision == ICP_F64 && !force_dense && n > 0 && m > 0) {
        // fp64 on the sparse structure (nn_match_row64_f64: rows of 64 points, one launch per pass): up to two blocks per CU
        // and a model that is searched flat; ICP_F64_SPARSE=0 keeps the dense thread-per-point kernel
        static const int env_sparse = env_int("ICP_NN_SPARSE", 1);
        static const int env_boxes = env_int("ICP_NN_BOXES", 1);
        const int env_f64 = env_int("ICP_F64_SPARSE", 1);   // (not cached: the tests switch it between contexts)
        if (env_sparse && env_boxes && env_f64 && pl.n_pad / 64 <= 2 * num_cus && pl.m_pad < (1 << 17)) {
            pl.version = 3;
            pl.sparse = 1;
            pl.cull = 1;
            pl.chunk = 8;
            pl.row = 64;
            pl.pts_per_thread = 1;
            pl.blocks_x = pl.n_pad / 64;
            pl.splits = 1;
            pl.seg_len = round_up(pl.m_pad, 8);
            return pl;
        }
    }
    int T = precision == ICP_F64 ? NNCfg<double>::T : NNCfg<float>::T;
    if (env_T == 1 || env_T == 2 || env_T == 4 || env_T == 8) T = env_T;
    if (precision == ICP_F64 && T > 4) T = 4;
    pl.pts_per_thread = T;
    pl.blocks_x = pl.n_pad / (NN_BLOCK * pl.pts_per_thread);
    if (n <= 0 || m <= 0) { pl.splits = 0; pl.seg_len = 0; return pl; }
    // small clouds cannot fill 256 CUs along the moving axis alone: split the model range over
    // grid.y until every CU holds `bpc` blocks of 4 waves.
    const int bpc = env_bpc > 0 ? env_bpc : 2;
    const int target_blocks = num_cus * bpc;
    int S = (target_blocks + pl.blocks_x - 1) / pl.blocks_x;
    const int max_S = (pl.m_pad + 255) / 256;  // keep >= 256 model points per segment
    if (S > max_S) S = max_S;
    if (env_S > 0) S = env_S;
    if (S < 1) S = 1;
    int seg = round_up((pl.m_pad + S - 1) / S, NN_CHUNK);
    S = (pl.m_pad + seg - 1) / seg;
    pl.splits = S;
    pl.seg_len = seg;
    return pl;
}

template <typename F>
static hipError_t launch_nn_t(const NNPlan& pl, const void* P, const void* Q, void* part_d, int32_t* part_idx,
                              hipStream_t st)
{
    constexpr int TQ = NNCfg<F>::TQ;
    dim3 grid(pl.blocks_x, pl.splits);
#define ICP_LAUNCH_NN(TT)                                                                                          \
    hipLaunchKernelGGL((nn_match_kernel<F, TT, TQ>), grid, dim3(NN_BLOCK), 0, st, (const F*)P, pl.n_pad, (const F*)Q, \
                       pl.m_pad, pl.seg_len, (F*)part_d, part_idx)
    switch (pl.pts_per_thread) {
        case 1: ICP_LAUNCH_NN(1); break;
        case 2: ICP_LAUNCH_NN(2); break;
        case 8: if constexpr (sizeof(F) == 4) { ICP_LAUNCH_NN(8); break; }
        default: ICP_LAUNCH_NN(4); break;
    }
#undef ICP_LAUNCH_NN
    return hipGetLastError();
}

bool nn_can_fuse_tail(const NNPlan& pl)
{
    return ((pl.version == 2 && pl.pts_per_thread == 2 && pl.chunk == 8) || pl.version == 3) && pl.n > 0 && pl.m > 0;
}

int nn_block_threads(const NNPlan& pl) { return pl.sparse ? (pl.row == 64 ? R64_NW * 64 : (pl.nw == 8 ? 8 : (pl.nw == 4 && pl.hier) ? 4 : SP_NW) * 64) : NN_BLOCK; }

static long long* g_phase_log = nullptr;
static long long g_phase_log_cap = 0;
void set_phase_log(long long* dev, long long slots) { g_phase_log = dev; g_phase_log_cap = slots; }

static hipError_t launch_nn_v2(const NNPlan& pl, const void* P, const void* Q, void* part_d, int32_t* part_idx,
                               const NNFusedTransform* ft, const NNCullInputs* opt, const NNTailArgs* ta, hipStream_t st)
{
    dim3 grid(pl.blocks_x, pl.splits);
    RT<float> rt{};
    NNFuse fuse{};
    fuse.n = pl.n;
    fuse.m = pl.m;
    fuse.Q_gather = (const float*)Q;
    fuse.tlog = g_phase_log;
    fuse.tlog_cap = g_phase_log_cap;
    static const int env_tpass = env_int("ICP_NN_PHASE_PASS", -1);
    fuse.tlog_pass = env_tpass;
    // (diagnostic: ICP_NN_PHASE_WIPE=1 clears the log ahead of every launch -- launches of different block sizes, spare blocks
    // and the parts of split rows that do not close them would otherwise leave older stamps among the last launch's)
    static const int env_twipe = env_int("ICP_NN_PHASE_WIPE", 0);
    if (env_twipe && g_phase_log != nullptr && hipMemsetAsync(g_phase_log, 0, (size_t)g_phase_log_cap * sizeof(long long), st) != hipSuccess) return hipErrorInvalidValue;
    fuse.work = opt ? opt->work : nullptr;
    const void* Qscan = Q;
    if (pl.cull && opt && opt->Q_scan) {
        Qscan = opt->Q_scan;
        fuse.seed_idx = opt->seed_idx;
        // (the boxes describe the sparse kernel's view of the model: usable here only if that is the model's own order)
        fuse.boxes = (pl.chunk == 8 && !opt->Q_scan_sorted) ? (const float*)opt->boxes : nullptr;
    }
    if (ft) {
        if (ft->mailbox) {
            if (!pl.sparse) return hipErrorInvalidValue;  // only the sparse kernel can be armed
            fuse.mailbox = ft->mailbox;
            fuse.relay = ft->relay;
            fuse.want = ft->want;
            fuse.want_lo = (unsigned int)(unsigned long long)ft->want;
            fuse.slot_state = (pl.sparse && pl.row != 64 && pl.splits == 1 && !ft->resident) ? (float*)ft->slot_state : nullptr;
            fuse.slot_valid = (fuse.slot_state && ft->slot_valid) ? 1 : 0;
            fuse.slot_flip = ft->slot_flip ? 1 : 0;
            fuse.resident = ft->resident ? 1 : 0;
            const int env_spec = env_int("ICP_NN_SPECULATE", 1);   // (A/B runs and tests: 0 switches the speculative search of resident launches off)
            fuse.speculate = (ft->resident && env_spec) ? 1 : 0;
            {
                const char* g = getenv("ICP_SPEC_GAIN");
                const char* f = getenv("ICP_SPEC_FLOOR");
                fuse.spec_gain = g && *g ? (float)atof(g) : 2.0f;
                fuse.spec_floor = f && *f ? (float)atof(f) : 1e-3f;
            }
            fuse.store_first = ft->store_first ? 1 : 0;
        } else {
            for (int k = 0; k < 9; ++k) rt.r[k] = (float)ft->R9[k];
            for (int k = 0; k < 3; ++k) rt.t[k] = (float)ft->t3[k];
            // (round 3: a plain launch -- the loop of a cloud whose rows the device adds up -- leaves and finds its points and
            // matches in slot order too, as an armed one does)
            fuse.slot_state = (pl.sparse && pl.row != 64 && pl.splits == 1) ? (float*)ft->slot_state : nullptr;
            fuse.slot_valid = (fuse.slot_state && ft->slot_valid) ? 1 : 0;
            fuse.slot_flip = ft->slot_flip ? 1 : 0;
        }
        fuse.apply = 1;
        fuse.n = pl.n;
        fuse.idx_prev = ft->idx_prev;
        fuse.P_out = (float*)ft->P_out;
        fuse.err_rows = ft->err_rows;
    }
    NNTail tail{};
    tail.row = -1;
    if (ta) {
        if (!nn_can_fuse_tail(pl)) return hipErrorInvalidValue;
        tail.keys = ta->keys;
        tail.tickets = ta->tickets;
        tail.err_tile = ta->err_tile;
        tail.idx_out = ta->idx_out;
        tail.idx_out_odd = ta->idx_out_odd ? ta->idx_out_odd : ta->idx_out;
        tail.Nrm = (const float*)ta->Nrm_soa;
        tail.rows = ta->rows;
        tail.tag = ta->tag;
        tail.tag_lo = (unsigned int)(unsigned long long)ta->tag;
        tail.compact = (ta->compact && pl.sparse && ta->metric == ICP_POINT_TO_POINT) ? 1 : 0;
        tail.rows_on_device = (ta->rows_on_device && !tail.compact) ? 1 : 0;
    }
#define ICP_LAUNCH_NN2T(CU, TL)                                                                                     \
    hipLaunchKernelGGL((nn_match_f32_v2<2, 8, CU, TL>), grid, dim3(NN_BLOCK), 0, st, (const float*)P, pl.n_pad,      \
                       (const float*)Qscan, pl.m_pad, pl.seg_len, (float*)part_d, part_idx, rt, fuse, tail)
#define ICP_LAUNCH_NN2(TT, CC, CU)                                                                                  \
    hipLaunchKernelGGL((nn_match_f32_v2<TT, CC, CU, 0>), grid, dim3(NN_BLOCK), 0, st, (const float*)P, pl.n_pad,     \
                       (const float*)Qscan, pl.m_pad, pl.seg_len, (float*)part_d, part_idx, rt, fuse, tail)
    if (pl.sparse) {
        // the plan's geometry is the sparse kernel's: it needs the scan copy and its chunk boxes
        if (!(opt && opt->Q_scan && opt->boxes)) return hipErrorInvalidValue;
        if (pl.m_pad >= (1 << 28)) return hipErrorInvalidValue;  // the in-block merge key carries 28 index bits
        fuse.seed_idx = opt->seed_idx;
        fuse.boxes = (const float*)opt->boxes;
        fuse.q_perm = opt->Q_scan_sorted ? opt->q_perm : nullptr;
        fuse.p_perm = opt->p_perm;
        const void* Qsp = opt->Q_scan_sorted ? opt->Q_scan_sorted : opt->Q_scan;
        static const int env_samples = env_int("ICP_NN_SAMPLES", 1);
        fuse.samples = env_samples ? (const float*)opt->samples : nullptr;
        // the cold start's full round (its probe round is 8 groups): 64 groups = 512 samples on a small model -- measured
        // (round 2, rows of 64): as good as 2048 on the 128 x 128 grid, Bunny_res and a random cloud, and 4 us less of a cold
        // pass on the hall scan, where a few blocks take the full round without gaining from it; 2048 on large models
        static const int env_sgroups = env_int("ICP_NN_SAMPLE_GROUPS", 0);
        fuse.sample_groups = env_sgroups > 0 ? env_sgroups : (pl.m_pad <= 32768 ? 64 : 256);
        // (round 3, later: with the refinement round of the traversal -- local samples, where the row's neighbours are -- the coarse
        // round of a SEEDED pass costs more than it adds: off unless ICP_NN_RESAMPLE=k asks for it; 5.14 -> 5.10 ms)
        fuse.resample_bound = (opt->sample_spacing2 > 0.f && pl.hier) ? opt->sample_spacing2 * env_float("ICP_NN_RESAMPLE", 0.0f) : 0.f;
        static const int env_passes = env_int("ICP_NN_PASSES", 0);
        // seeded: few hits, long rounds; cold: short rounds so that the exchanged minima start pruning early
        static const int env_waves = env_int("ICP_NN_WAVES", 0);
        int cus64 = 0, dev64 = 0;
        if (pl.row == 64 && (hipGetDevice(&dev64) != hipSuccess || hipDeviceGetAttribute(&cus64, hipDeviceAttributeMultiprocessorCount, dev64) != hipSuccess)) cus64 = 256;
        const int nw64 = (pl.row == 64 && (env_waves == 16 || opt->waves64 == 16) && pl.blocks_x <= cus64) ? 16 : R64_NW;
        const int max_passes = pl.row == 64 ? SP_HCAP / (nw64 * 64) : pl.hier ? SP_MAX_PASSES : SP_HCAP_FLAT / (((pl.nw == 8 && ta) ? 8 : SP_NW) * 64);
        int passes = env_passes > 0 ? env_passes : ((fuse.seed_idx || fuse.samples) ? max_passes : 1);
        if (passes > max_passes) passes = max_passes;
        if (pl.row == 64) {
            // ---- 64-point rows ----
            if (pl.splits != 1 || pl.hier) return hipErrorInvalidValue;
            const bool diag = fuse.tlog != nullptr || fuse.work != nullptr, perm = fuse.q_perm != nullptr;
            const int tl = !ta ? 0 : (ta->metric == ICP_POINT_TO_PLANE ? 2 : 1);
#define ICP_R64_FN(TL, W) {{(const void*)nn_match_row64<TL, false, false, W>, (const void*)nn_match_row64<TL, false, true, W>},   \
                           {(const void*)nn_match_row64<TL, true, false, W>, (const void*)nn_match_row64<TL, true, true, W>}}
            static const void* const fns[2][3][2][2] = {{ICP_R64_FN(0, 8), ICP_R64_FN(1, 8), ICP_R64_FN(2, 8)},
                                                        {ICP_R64_FN(0, 16), ICP_R64_FN(1, 16), ICP_R64_FN(2, 16)}};
#undef ICP_R64_FN
            const void* fn = fns[nw64 == 16 ? 1 : 0][tl][diag ? 1 : 0][perm ? 1 : 0];
            const float* Pp = (const float*)P;
            const float* Qp = (const float*)Qsp;
            int n_pad = pl.n_pad, m_pad = pl.m_pad;
            float* pd = (float*)part_d;
            void* args[] = {&Pp, &n_pad, &Qp, &m_pad, &passes, &pd, &part_idx, &rt, &fuse, &tail};
            const dim3 g64(pl.blocks_x, 1);
            if (fuse.resident) {
                if (!ta) return hipErrorInvalidValue;
                // every block must be on the machine at once (see below): blocks <= CUs x resident blocks per CU
                const int variant = 100 + (((nw64 == 16 ? 3 : 0) + tl) * 2 + (diag ? 1 : 0)) * 2 + (perm ? 1 : 0);
                long long cap = 0;
                {
                    static std::mutex mu;
                    static std::map<std::pair<int, int>, long long> capacity;
                    int dev = 0;
                    if (hipGetDevice(&dev) != hipSuccess) return hipErrorCooperativeLaunchTooLarge;
                    std::lock_guard<std::mutex> lock(mu);
                    long long& slot = capacity[std::make_pair(dev, variant)];
                    if (slot <= 0) {
                        int per_cu = 0, cus = 0;
                        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nw64 * 64, 0) != hipSuccess ||
                            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
                            return hipErrorCooperativeLaunchTooLarge;
                        slot = (long long)per_cu * cus;
                    }
                    cap = slot;
                }
                if ((long long)g64.x > cap) return hipErrorCooperativeLaunchTooLarge;
            }
            return hipLaunchKernel(fn, g64, dim3(nw64 * 64), args, 0, st);
        }
        {
            // ---- 128-point rows: one table of instantiations; 8-wave blocks exist with a fused tail only, 4-wave blocks with a fused
            // tail and the hierarchical search only.  A plan of 4-wave blocks runs its COLD launches (no previous match: every block
            // starts from the sample round, and the rows are split by counters that are a registration old) on 8 waves: 10 M x 10 M,
            // first pass 34.9 ms against 44.1 ----
            const bool diag = fuse.tlog != nullptr || fuse.work != nullptr, perm = fuse.q_perm != nullptr, hier = pl.hier != 0;
            const int tl = !ta ? 0 : (ta->metric == ICP_POINT_TO_PLANE ? 2 : 1);
            const bool cold_launch = fuse.seed_idx == nullptr && fuse.slot_valid == 0;
            // (... unless the rows' counters hold a registration's history -- a context's second registration on: the cold pass of a
            // repeat then splits its heavy rows about right and the 4-wave form wins, 5.13 -> 5.04 ms per iteration; without history
            // the share of one rank of eight ran its first registration in 24.1 ms on 8 waves against 26.3 on 4)
            const bool cold8 = cold_launch && env_int("ICP_NN_COLD8", 1) && !(opt->row_order != nullptr && opt->order_history);
            const int nw = (pl.nw == 8 && tl != 0) ? 8 : (pl.nw == 4 && tl != 0 && hier) ? (cold8 ? 8 : 4) : SP_NW;
            if (passes > (hier ? SP_MAX_PASSES : SP_HCAP_FLAT / (nw * 64))) passes = hier ? SP_MAX_PASSES : SP_HCAP_FLAT / (nw * 64);
            if (!hier && (pl.m_pad >> 3) > 65536) return hipErrorInvalidValue;   // (the flat search lists 16-bit chunk numbers; nn_plan never asks for it)
#define ICP_SP_FN(TL, DG, PM) {(const void*)nn_match_sparse<TL, DG, PM, false>, (const void*)nn_match_sparse<TL, DG, PM, true>}
#define ICP_SP_FN8(TL, DG, PM) {(const void*)nn_match_sparse<TL, DG, PM, false, 8>, (const void*)nn_match_sparse<TL, DG, PM, true, 8>}
#define ICP_SP_FN4(TL, DG, PM) {nullptr, (const void*)nn_match_sparse<TL, DG, PM, true, 4>}
            static const void* const fns[3][3][2][2][2] = {
                {{{ICP_SP_FN(0, false, false), ICP_SP_FN(0, false, true)}, {ICP_SP_FN(0, true, false), ICP_SP_FN(0, true, true)}},
                 {{ICP_SP_FN(1, false, false), ICP_SP_FN(1, false, true)}, {ICP_SP_FN(1, true, false), ICP_SP_FN(1, true, true)}},
                 {{ICP_SP_FN(2, false, false), ICP_SP_FN(2, false, true)}, {ICP_SP_FN(2, true, false), ICP_SP_FN(2, true, true)}}},
                {{{{nullptr, nullptr}, {nullptr, nullptr}}, {{nullptr, nullptr}, {nullptr, nullptr}}},
                 {{ICP_SP_FN8(1, false, false), ICP_SP_FN8(1, false, true)}, {ICP_SP_FN8(1, true, false), ICP_SP_FN8(1, true, true)}},
                 {{ICP_SP_FN8(2, false, false), ICP_SP_FN8(2, false, true)}, {ICP_SP_FN8(2, true, false), ICP_SP_FN8(2, true, true)}}},
                {{{{nullptr, nullptr}, {nullptr, nullptr}}, {{nullptr, nullptr}, {nullptr, nullptr}}},
                 {{ICP_SP_FN4(1, false, false), ICP_SP_FN4(1, false, true)}, {ICP_SP_FN4(1, true, false), ICP_SP_FN4(1, true, true)}},
                 {{ICP_SP_FN4(2, false, false), ICP_SP_FN4(2, false, true)}, {ICP_SP_FN4(2, true, false), ICP_SP_FN4(2, true, true)}}}};
#undef ICP_SP_FN
#undef ICP_SP_FN8
#undef ICP_SP_FN4
            const void* fn = fns[nw == 8 ? 1 : nw == 4 ? 2 : 0][tl][diag ? 1 : 0][perm ? 1 : 0][hier ? 1 : 0];
            if (fn == nullptr) return hipErrorInvalidValue;
            const float* Pp = (const float*)P;
            const float* Qp = (const float*)Qsp;
            int n_pad = pl.n_pad, m_pad = pl.m_pad, seg = pl.seg_len;
            float* pd = (float*)part_d;
            void* args[] = {&Pp, &n_pad, &Qp, &m_pad, &seg, &passes, &pd, &part_idx, &rt, &fuse, &tail};
            // Shared rows: a grid of more blocks than rows, the roles dealt out inside the kernel from the
            // hits each row had in the previous launch (three count arrays in rotation: read, add to, zero for the next)
            dim3 g = grid;
            if (nw == 8 && pl.share_blocks > pl.blocks_x && pl.splits == 1 && opt->share_counts != nullptr && opt->share_seq != nullptr && opt->share_cold_seq != nullptr &&
                (!fuse.resident || (opt->seed_pub != nullptr && env_int("ICP_NN_SHARE_RESIDENT", 1))) &&   // (0: a resident launch keeps one block per row -- A/B runs)
               
                pl.blocks_x <= nw * 64) {
                const unsigned long long seq = (*opt->share_seq)++;
                const size_t R = (size_t)pl.blocks_x;
                fuse.share_prev = opt->share_counts + ((seq + 2) % 3) * R;
                fuse.share_cur = opt->share_counts + (seq % 3) * R;
                fuse.share_next = opt->share_counts + ((seq + 1) % 3) * R;
                // The first pass of a registration (no previous match: the launch has no seeds) has no previous pass to go by:
                // it takes the counts of the PREVIOUS registration's first pass (two more arrays, alternating) -- a sensor's
                // consecutive scans are heavy in the same places; a first registration finds zeros there and runs unshared.
                // The array the next first pass adds to is zeroed by every ordinary pass in between.
                unsigned int* cold = opt->share_counts + 3 * R;
                if (fuse.seed_idx == nullptr) {
                    const unsigned long long k = (*opt->share_cold_seq)++;
                    fuse.share_prev = cold + ((k + 1) % 2) * R;
                    fuse.share_cur2 = cold + (k % 2) * R;
                    // The array this launch adds to is zeroed on the stream, ahead of the launch: where every launch is a first pass
                    // (resident kernels, one per registration) no ordinary pass in between would do it, and the counts of all
                    // registrations would pile up.  (Not by a block of the launch itself: nothing orders block 0's stores before
                    // another block's adds, and lost counts would make the roles -- and the timings -- differ from run to run.)
                    if (hipMemsetAsync(fuse.share_cur2, 0, R * sizeof(unsigned int), st) != hipSuccess) return hipErrorInvalidValue;
                    fuse.share_zero2 = nullptr;
                } else {
                    fuse.share_zero2 = cold + (*opt->share_cold_seq % 2) * R;
                }
                fuse.share_rows = pl.blocks_x;
                {
                    const int env_min = env_int("ICP_NN_SHARE_MIN", 0);   // (A/B runs)
                    fuse.share_min = env_min > 0 ? env_min : 8 * nw;      // one batch for every wave
                }
                g = dim3(pl.share_blocks, 1);
            }
            fuse.seed_pub = opt->seed_pub;
            fuse.records = hier ? opt->records : nullptr;
            // (a round of the chunk find covers 16 listed super boxes -- the hit list would hold 64: every round starts from the largest
            // bound the rounds before have left, so shorter rounds list less; 10 M x 10 M, rounds of 64 / 32 / 16 / 8: 1.41 / 1.29 / 1.19 /
            // 1.13 G chunks listed per registration, 5.93 / 5.75 / 5.65 / 5.69 ms per iteration.  ICP_NN_ROUND_SUPERS for the A/B)
            // (the refinement round of a pass that lists >= 12 super boxes, over <= 256 of their chunk samples: 10 M x 10 M 5.68 -> 5.14 ms,
            // 1.19 -> 0.99 G chunks listed and 370 -> 327 M evaluated per registration; 24 / 512: 5.26, 12 / 128: 5.15, 8 / 256: 5.17.
            // ICP_NN_REFINE_MIN=0: never -- A/B runs; not cached)
            { const int rm = env_int("ICP_NN_REFINE_MIN", 12), rc = env_int("ICP_NN_REFINE_SAMPLES", 256); fuse.refine_min = rm > 0 ? rm : 0; fuse.refine_cnt = rc >= 64 && rc <= 1024 ? rc : 256; }
            { const int rs = env_int("ICP_NN_ROUND_SUPERS", 16); fuse.round_supers = rs >= 1 && rs <= 64 ? rs : 16; }
            if (hier && fuse.records == nullptr) return hipErrorInvalidValue;   // (the hierarchical search fetches its hits from the records)
            if (pl.order && ta && !fuse.resident && opt->row_order != nullptr && opt->row_hits != nullptr) {
                fuse.row_order = opt->row_order;
                fuse.row_hits = opt->row_hits;
                g = dim3(pl.blocks_x + NN_ORDER_EXTRA, 1);   // (the roles of the blocks beyond the rows: parts of split rows, or none)
            }
            if (fuse.resident) {
                if (!ta || pl.splits != 1) return hipErrorInvalidValue;
                const int variant = ((((nw == 8 ? 2 : nw == 4 ? 4 : 0) + (tl == 2 ? 1 : 0)) * 2 + (diag ? 1 : 0)) * 2 + (perm ? 1 : 0)) * 2 + (hier ? 1 : 0);
                // Every block must be on the machine at once (they all wait for the same host).  A cooperative launch
                // guarantees that or refuses, but costs ~13 us more per launch here; the same guarantee comes from the
                // occupancy query it is built on: the grid fits iff blocks <= CUs x resident blocks per CU.  Blocks that
                // start late (behind the previous kernel of the stream) only delay the first pass, nothing waits on them
                // that they cannot deliver.  ICP_COOP=1 uses the cooperative launch.
                static const int env_coop = env_int("ICP_COOP", 0);
                if (env_coop) return hipLaunchCooperativeKernel(fn, g, dim3(nw * 64), args, 0, st);
                // blocks the machine holds at once, per (device, variant): asked once, remembered under a lock (contexts of
                // several devices and threads share this table)
                long long cap = 0;
                {
                    static std::mutex mu;
                    static std::map<std::pair<int, int>, long long> capacity;
                    int dev = 0;
                    if (hipGetDevice(&dev) != hipSuccess) return hipErrorCooperativeLaunchTooLarge;
                    std::lock_guard<std::mutex> lock(mu);
                    long long& slot = capacity[std::make_pair(dev, variant)];
                    if (slot <= 0) {
                        int per_cu = 0, cus = 0;
                        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nw * 64, 0) != hipSuccess ||
                            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
                            return hipErrorCooperativeLaunchTooLarge;
                        slot = (long long)per_cu * cus;
                    }
                    cap = slot;
                }
                if ((long long)g.x * g.y > cap) return hipErrorCooperativeLaunchTooLarge;
                return hipLaunchKernel(fn, g, dim3(nw * 64), args, 0, st);
            }
            return hipLaunchKernel(fn, g, dim3(nw * 64), args, 0, st);
        }
    }
    // measured (profiles/r1/03_nn_sweep_cull.txt): without a seed the early-out variant loses to the plain
    // packed kernel on every cloud (its bound starts at +inf), with one it wins on every cloud
    const bool cull = pl.cull && Qscan != Q && fuse.seed_idx != nullptr;
    if (!cull) { Qscan = Q; fuse.seed_idx = nullptr; fuse.boxes = nullptr; }
    if (ta) {
        const bool plane = ta->metric == ICP_POINT_TO_PLANE;
        if (cull) { if (plane) ICP_LAUNCH_NN2T(1, 2); else ICP_LAUNCH_NN2T(1, 1); }
        else { if (plane) ICP_LAUNCH_NN2T(0, 2); else ICP_LAUNCH_NN2T(0, 1); }
    } else if (pl.pts_per_thread == 4) {
        if (pl.chunk == 8) ICP_LAUNCH_NN2(4, 8, 0); else ICP_LAUNCH_NN2(4, 16, 0);
    } else if (cull) {
        if (pl.chunk == 8) ICP_LAUNCH_NN2(2, 8, 1); else ICP_LAUNCH_NN2(2, 16, 1);
    } else {
        if (pl.chunk == 8) ICP_LAUNCH_NN2(2, 8, 0); else ICP_LAUNCH_NN2(2, 16, 0);
    }
#undef ICP_LAUNCH_NN2
#undef ICP_LAUNCH_NN2T
    return hipGetLastError();
}

bool nn_can_fuse_transform(const NNPlan& pl) { return (pl.version == 2 || pl.version == 3) && pl.n > 0 && pl.m > 0; }

// fp64, rows of 64 points: one launch per pass, no mailbox (NNFusedTransform::mailbox must be NULL)
static hipError_t launch_row64_f64(const NNPlan& pl, const void* P, const void* Q, void* part_d, int32_t* part_idx,
                                   const NNFusedTransform* ft, const NNCullInputs* opt, const NNTailArgs* ta, hipStream_t st)
{
    if (!(opt && opt->Q_scan && opt->boxes)) return hipErrorInvalidValue;
    if (ft && ft->mailbox && !ft->resident) return hipErrorInvalidValue;   // (no armed launches in double)
    RT<double> rt{};
    NNFuse fuse{};
    fuse.n = pl.n;
    fuse.m = pl.m;
    fuse.Q_gather = (const float*)Q;                 // (typed by the kernel: doubles)
    fuse.seed_idx = opt->seed_idx;
    fuse.boxes = (const float*)opt->boxes;
    static const int env_samples = env_int("ICP_NN_SAMPLES", 1);
    fuse.samples = env_samples ? (const float*)opt->samples : nullptr;
    static const int env_sgroups = env_int("ICP_NN_SAMPLE_GROUPS", 256);
    fuse.sample_groups = env_sgroups;
    if (ft) {
        if (ft->mailbox) {   // resident launch: (R, t) arrive as messages (NNMailbox64)
            fuse.mailbox = ft->mailbox;
            fuse.relay = ft->relay;
            fuse.want = ft->want;
            fuse.want_lo = (unsigned int)(unsigned long long)ft->want;
            fuse.resident = 1;
            fuse.store_first = ft->store_first ? 1 : 0;
        } else {
            for (int k = 0; k < 9; ++k) rt.r[k] = ft->R9[k];
            for (int k = 0; k < 3; ++k) rt.t[k] = ft->t3[k];
        }
        fuse.apply = 1;
        fuse.idx_prev = ft->idx_prev;
        fuse.P_out = (float*)ft->P_out;
        fuse.err_rows = ft->err_rows;
    }
    NNTail tail{};
    tail.row = -1;
    if (ta) {
        tail.idx_out = ta->idx_out;
        tail.idx_out_odd = ta->idx_out_odd ? ta->idx_out_odd : ta->idx_out;
        tail.Nrm = (const float*)ta->Nrm_soa;
        tail.rows = ta->rows;
        tail.tag = ta->tag;
        tail.tag_lo = (unsigned int)(unsigned long long)ta->tag;
        tail.compact = (ta->compact && ta->metric == ICP_POINT_TO_POINT) ? 1 : 0;
    }
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int nw = pl.blocks_x <= cus ? 16 : 8;
    const int passes = (fuse.seed_idx || fuse.samples) ? SP_HCAP / (nw * 64) : 1;
    const dim3 grid(pl.blocks_x), block(nw * 64);
    const double* Pp = (const double*)P;
    const double* Qs = (const double*)opt->Q_scan;
    const int tl = !ta ? 0 : (ta->metric == ICP_POINT_TO_PLANE ? 2 : 1);
    static const void* const fns[3][2] = {{(const void*)nn_match_row64_f64<0, 8>, (const void*)nn_match_row64_f64<0, 16>},
                                          {(const void*)nn_match_row64_f64<1, 8>, (const void*)nn_match_row64_f64<1, 16>},
                                          {(const void*)nn_match_row64_f64<2, 8>, (const void*)nn_match_row64_f64<2, 16>}};
    const void* fn = fns[tl][nw == 16 ? 1 : 0];
    // icp_set_work_counting: the instrumented instantiation (point-to-point rows only: what the fp64 loop of src/ICP_CPU.c runs)
    fuse.work = (opt->work != nullptr && tl == 1) ? opt->work : nullptr;
    if (fuse.work != nullptr) fn = nw == 16 ? (const void*)nn_match_row64_f64<1, 16, true> : (const void*)nn_match_row64_f64<1, 8, true>;
    if (fuse.resident) {
        // every block must be on the machine at once: blocks <= CUs x resident blocks per CU (the occupancy query)
        if (!ta) return hipErrorInvalidValue;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nw * 64, 0) != hipSuccess) return hipErrorCooperativeLaunchTooLarge;
        if ((long long)pl.blocks_x > (long long)per_cu * cus) return hipErrorCooperativeLaunchTooLarge;
    }
    int n_pad = pl.n_pad, m_pad = pl.m_pad, passes_ = passes;
    double* pd = (double*)part_d;
    void* args[] = {&Pp, &n_pad, &Qs, &m_pad, &passes_, &pd, &part_idx, &rt, &fuse, &tail};
    return hipLaunchKernel(fn, grid, block, args, 0, st);
}

hipError_t launch_nn(const NNPlan& pl, const void* P, const void* Q, void* part_d, int32_t* part_idx,
                     const NNFusedTransform* ft, const NNCullInputs* opt, const NNTailArgs* ta, hipStream_t st)
{
    if (pl.n <= 0 || pl.m <= 0) return hipSuccess;
    if (pl.version == 2) return launch_nn_v2(pl, P, Q, part_d, part_idx, ft, opt, ta, st);
    if (pl.version == 3) return launch_row64_f64(pl, P, Q, part_d, part_idx, ft, opt, ta, st);
    if (ft || ta) return hipErrorInvalidValue;  // only the packed fp32 kernel carries the fused front end
    return pl.precision == ICP_F64 ? launch_nn_t<double>(pl, P, Q, part_d, part_idx, st)
                                   : launch_nn_t<float>(pl, P, Q, part_d, part_idx, st);
}

hipError_t launch_merge(const NNPlan& pl, const void* part_d, const int32_t* part_idx, int32_t* idx, hipStream_t st)
{
    if (pl.n <= 0 || pl.m <= 0) return hipSuccess;
    const int blocks = (pl.n + 255) / 256;
    if (pl.precision == ICP_F64)
        hipLaunchKernelGGL((merge_kernel<double>), dim3(blocks), dim3(256), 0, st, (const double*)part_d, part_idx,
                           pl.splits, pl.n_pad, pl.n, pl.m, idx);
    else
        hipLaunchKernelGGL((merge_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)part_d, part_idx,
                           pl.splits, pl.n_pad, pl.n, pl.m, idx);
    return hipGetLastError();
}

hipError_t launch_moments(const NNPlan& pl, int metric, const void* P, const void* Q, const void* Nrm,
                          const void* part_d, const int32_t* part_idx, int32_t* idx, double* partials, int* blocks,
                          double tag, const double* err_rows, int err_count, hipStream_t st)
{
    int nb = (pl.n + MOM_BLOCK - 1) / MOM_BLOCK;
    if (nb > MOM_MAX_BLOCKS) nb = MOM_MAX_BLOCKS;
    *blocks = nb;
    if (nb <= 0) return hipSuccess;
#define ICP_LAUNCH_MOM(F, MET)                                                                                     \
    hipLaunchKernelGGL((moments_kernel<F, MET>), dim3(nb), dim3(MOM_BLOCK), 0, st, (const F*)P, pl.n, pl.n_pad,      \
                       (const F*)Q, pl.m, pl.m_pad, (const F*)Nrm, (const F*)part_d, part_idx, pl.splits, idx, partials, tag, err_rows, err_count)
    if (pl.precision == ICP_F64) {
        if (metric == ICP_POINT_TO_PLANE) ICP_LAUNCH_MOM(double, ICP_POINT_TO_PLANE);
        else ICP_LAUNCH_MOM(double, ICP_POINT_TO_POINT);
    } else {
        if (metric == ICP_POINT_TO_PLANE) ICP_LAUNCH_MOM(float, ICP_POINT_TO_PLANE);
        else ICP_LAUNCH_MOM(float, ICP_POINT_TO_POINT);
    }
#undef ICP_LAUNCH_MOM
    return hipGetLastError();
}

hipError_t launch_transform_error(int precision, void* P, int n, int n_pad, const double* R9, const double* t3,
                                  const void* Q, int m_pad, const int32_t* idx, double* err_partials, int* blocks,
                                  hipStream_t st)
{
    int nb = (n_pad + TR_BLOCK - 1) / TR_BLOCK;
    if (nb > MOM_MAX_BLOCKS) nb = MOM_MAX_BLOCKS;
    *blocks = nb;
    if (nb <= 0) return hipSuccess;
    if (precision == ICP_F64) {
        RT<double> rt;
        for (int k = 0; k < 9; ++k) rt.r[k] = R9[k];
        for (int k = 0; k < 3; ++k) rt.t[k] = t3[k];
        hipLaunchKernelGGL((transform_error_kernel<double>), dim3(nb), dim3(TR_BLOCK), 0, st, (double*)P, n, n_pad, rt,
                           (const double*)Q, m_pad, idx, err_partials);
    } else {
        RT<float> rt;
        for (int k = 0; k < 9; ++k) rt.r[k] = (float)R9[k];
        for (int k = 0; k < 3; ++k) rt.t[k] = (float)t3[k];
        hipLaunchKernelGGL((transform_error_kernel<float>), dim3(nb), dim3(TR_BLOCK), 0, st, (float*)P, n, n_pad, rt,
                           (const float*)Q, m_pad, idx, err_partials);
    }
    return hipGetLastError();
}

// many rows (a cloud of millions of points: 78 125 rows for 10 M): one block walking all of them is milliseconds -- the
// RCCL route of configs[4] spent 5.6 ms per iteration there.  Stage 1: up to 256 blocks each add a contiguous range of
// rows, in the same fixed order, into one row of `scratch`; stage 2: the block above adds those.  Fixed ranges, fixed
// order: the same bits on every rank and every run.
__global__ __launch_bounds__(256) void finalize_ranges_kernel(double* __restrict__ scratch, const double* __restrict__ mom_partials, int mom_blocks,
                                                              int per, int rows_have_err)
{
    __shared__ double red[8][ICP_NMOM];
    const int k = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int b0 = (int)blockIdx.x * per, b1 = min(b0 + per, mom_blocks);
    double s = 0.0;
    if (k != ICP_NMOM - 1 && (k != 0 || rows_have_err))
        for (int b = b0 + part; b < b1; b += 8) s += mom_partials[(size_t)b * ICP_NMOM + k];
    red[part][k] = s;
    __syncthreads();
    if (threadIdx.x < ICP_NMOM) {
        double tot = red[0][k];
#pragma unroll
        for (int p = 1; p < 8; ++p) tot += red[p][k];
        scratch[(size_t)blockIdx.x * ICP_NMOM + k] = tot;
    }
}

hipError_t launch_finalize(double* mom_out, const double* mom_partials, int mom_blocks, const double* err_partials,
                           int err_blocks, int rows_have_err, hipStream_t st, double* scratch)
{
    if (scratch != nullptr && mom_blocks > 2048) {
        const int groups = 256, per = (mom_blocks + groups - 1) / groups;
        hipLaunchKernelGGL(finalize_ranges_kernel, dim3(groups), dim3(256), 0, st, scratch, mom_partials, mom_blocks, per, rows_have_err);
        hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, mom_out, scratch, groups, err_partials, err_blocks, rows_have_err);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, mom_out, mom_partials, mom_blocks, err_partials,
                       err_blocks, rows_have_err);
    return hipGetLastError();
}

hipError_t launch_aos_to_soa(int precision, const void* aos, int n, int n_pad, void* soa, hipStream_t st, unsigned int* nonfinite)
{
    if (n <= 0) return hipSuccess;
    const int blocks = (n_pad + 255) / 256;
    if (precision == ICP_F64)
        hipLaunchKernelGGL((aos_to_soa_kernel<double>), dim3(blocks), dim3(256), 0, st, (const double*)aos, n, n_pad,
                           (double*)soa, nonfinite);
    else
        hipLaunchKernelGGL((aos_to_soa_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)aos, n, n_pad,
                           (float*)soa, nonfinite);
    return hipGetLastError();
}

hipError_t launch_soa_to_aos(int precision, const void* soa, int n, int n_pad, void* aos, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const int blocks = (n + 255) / 256;
    if (precision == ICP_F64)
        hipLaunchKernelGGL((soa_to_aos_kernel<double>), dim3(blocks), dim3(256), 0, st, (const double*)soa, n, n_pad,
                           (double*)aos);
    else
        hipLaunchKernelGGL((soa_to_aos_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)soa, n, n_pad,
                           (float*)aos);
    return hipGetLastError();
}

void knn4_v2_geometry(int m, int num_cus, int* n_pad, int* blocks_x, int* splits, int* seg_len)
{
    const int m_pad = pad_model(m);
    *n_pad = round_up(m, 128);
    *blocks_x = *n_pad / 128;
    if (num_cus <= 0) num_cus = 256;
    int S = (num_cus * 4 + *blocks_x - 1) / *blocks_x;        // 4 blocks (16 waves) per CU
    const int max_S = (m_pad + 511) / 512;
    if (S > max_S) S = max_S;
    if (S < 1) S = 1;
    int seg = round_up((m_pad + S - 1) / S, 4 * KNN_C);
    *splits = (m_pad + seg - 1) / seg;
    *seg_len = seg;
}

hipError_t launch_knn4_v2(const void* Q, int m, int num_cus, float* part_d, int32_t* part_j, int32_t* nbr, hipStream_t st)
{
    if (m <= 0) return hipSuccess;
    int n_pad, bx, S, seg;
    knn4_v2_geometry(m, num_cus, &n_pad, &bx, &S, &seg);
    hipLaunchKernelGGL(knn4_f32_v2, dim3(bx, S), dim3(NN_BLOCK), 0, st, (const float*)Q, m, pad_model(m), n_pad, seg, part_d,
                       part_j);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(knn4_merge_kernel, dim3((m + 255) / 256), dim3(256), 0, st, (const float*)part_d,
                       (const int32_t*)part_j, S, n_pad, m, nbr);
    return hipGetLastError();
}

hipError_t launch_knn4(const NNPlan& pl, const void* Q, int32_t* nbr, hipStream_t st)
{
    if (pl.m <= 0) return hipSuccess;
    const int blocks = (pl.m + NN_BLOCK - 1) / NN_BLOCK;
    if (pl.precision == ICP_F64)
        hipLaunchKernelGGL((knn4_kernel<double, 1024>), dim3(blocks), dim3(NN_BLOCK), 0, st, (const double*)Q, pl.m,
                           pl.m_pad, nbr);
    else
        hipLaunchKernelGGL((knn4_kernel<float, 2048>), dim3(blocks), dim3(NN_BLOCK), 0, st, (const float*)Q, pl.m,
                           pl.m_pad, nbr);
    return hipGetLastError();
}

hipError_t launch_normals(int precision, const void* Q, int m, int m_pad, const int32_t* nbr, void* Nrm_soa,
                          hipStream_t st)
{
    if (m <= 0) return hipSuccess;
    const int blocks = (m_pad + 127) / 128;
    if (precision == ICP_F64)
        hipLaunchKernelGGL((normals_kernel<double>), dim3(blocks), dim3(128), 0, st, (const double*)Q, m, m_pad, nbr,
                           (double*)Nrm_soa);
    else
        hipLaunchKernelGGL((normals_kernel<float>), dim3(blocks), dim3(128), 0, st, (const float*)Q, m, m_pad, nbr,
                           (float*)Nrm_soa);
    return hipGetLastError();
}

hipError_t launch_os1_packets(const uint8_t* packets, int n_packets, const float* alt16, const float* az16,
                              uint32_t* ranges, float* xyz_aos, hipStream_t st)
{
    if (n_packets <= 0) return hipSuccess;
    const int n = n_packets * 256;
    hipLaunchKernelGGL(os1_packets_kernel, dim3((n + 255) / 256), dim3(256), 0, st, packets, n_packets, alt16, az16, ranges,
                       xyz_aos);
    return hipGetLastError();
}

hipError_t launch_os1_conversion(const uint32_t* ranges, int n, uint32_t encoder0, const float* alt16,
                                 const float* az16, float* xyz_aos, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(os1_conversion_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ranges, n, encoder0, alt16, az16,
                       xyz_aos);
    return hipGetLastError();
}

}  // namespace icp
