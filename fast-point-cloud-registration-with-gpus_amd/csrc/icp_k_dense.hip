// icp_k_dense.hip -- gfx950 kernels that touch every point or every pair: layout conversion, the dense matching kernels (generic
// thread-per-point: fp64 fallback; packed fp32 v2: fallback + roofline reference), merge, fused moments, transform + error,
// finalize.  Reference statements: nn_match_* Matching<<<>>> src/CUDA/GPU_point_to_point_real.cu:38-79 / src/ICP_CPU.c:220-234;
// moments_kernel src/ICP_point_to_point.cu:308-357, src/CUDA/GPU_point_to_plane_real.cu:246-288,532-549; transform_error_kernel
// src/ICP_point_to_point.cu:81-88,403-416.
#include "icp_device.h"
#include <math.h>
#include <stdlib.h>
#include <cstring>

namespace icp {

// ------------------------------------------------------------------------------------------------
// layout conversion
// ------------------------------------------------------------------------------------------------
// `nonfinite` (pinned host memory, or NULL): counts the points with a NaN or an infinite coordinate -- icp_set_* refuse such a
// cloud (include/icp_mi355x.h).  Written only when there is something to count.
template <typename F>
__global__ void aos_to_soa_kernel(const F* __restrict__ aos, int n, int n_pad, F* __restrict__ soa, unsigned int* nonfinite, F* __restrict__ soa2, unsigned int* __restrict__ enc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n_pad;
    const int s = i < n ? i : n - 1;  // padding replicates the last real point
    F x = F(0), y = F(0), z = F(0);
    // the block's 256 points are 768 contiguous values: fetched as 16-byte vectors, coalesced (the source may be pinned HOST memory read
    // over PCIe -- the short set-up -- where three strided 4-byte loads per thread were 15 us for 196 KB), handed out through LDS; a block
    // that reaches beyond the cloud (its tail, the padding) reads its points one by one
    __shared__ __attribute__((aligned(16))) F buf[768];
    const size_t b0 = (size_t)blockIdx.x * 256;
    const bool whole = b0 + 256 <= (size_t)n;
    if (whole) {
        constexpr int PER = 16 / (int)sizeof(F), NV = 768 / PER;
        using V = typename Vec16<F>::type;
        const V* src = reinterpret_cast<const V*>(aos + 3 * b0);
        for (int k = threadIdx.x; k < NV; k += 256) reinterpret_cast<V*>(buf)[k] = src[k];
    }
    __syncthreads();
    if (in) {
        if (whole) { x = buf[3 * threadIdx.x]; y = buf[3 * threadIdx.x + 1]; z = buf[3 * threadIdx.x + 2]; }
        else { x = aos[3 * (size_t)s + 0]; y = aos[3 * (size_t)s + 1]; z = aos[3 * (size_t)s + 2]; }
        soa[i] = x;
        soa[(size_t)n_pad + i] = y;
        soa[2 * (size_t)n_pad + i] = z;
        if (soa2 != nullptr) { soa2[i] = x; soa2[(size_t)n_pad + i] = y; soa2[2 * (size_t)n_pad + i] = z; }
    }
    // (x - x is 0 for every finite x, NaN for NaN and for +-inf)
    const bool finite = (x - x) == F(0) && (y - y) == F(0) && (z - z) == F(0);
    if (in && nonfinite != nullptr && i < n && !finite)
        __hip_atomic_fetch_add(nonfinite, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if constexpr (sizeof(F) == 4) {
        // the bounding cube of the finite points, for the spatial order: a wave's minima and maxima, then six ordered-integer atomics
        if (enc != nullptr) {
            const bool use = in && i < n && finite;
            const float binf = __builtin_huge_valf();
            float lo[3] = {use ? (float)x : binf, use ? (float)y : binf, use ? (float)z : binf};
            float hi[3] = {use ? (float)x : -binf, use ? (float)y : -binf, use ? (float)z : -binf};
#pragma unroll
            for (int a = 0; a < 3; ++a)
                for (int off = 32; off > 0; off >>= 1) {
                    lo[a] = __builtin_fminf(lo[a], __shfl_xor(lo[a], off, 64));
                    hi[a] = __builtin_fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
                }
            if ((threadIdx.x & 63) == 0 && hi[0] >= lo[0]) {
                auto ord = [](float f) { const unsigned int b = __float_as_uint(f == 0.f ? 0.f : f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); };
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    atomicMax(&enc[a], ~ord(lo[a]));
                    atomicMax(&enc[3 + a], ord(hi[a]));
                }
            }
        }
    }
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
// launchers
// ------------------------------------------------------------------------------------------------
template <typename F> struct NNCfg;
template <> struct NNCfg<float> { static constexpr int T = 4; static constexpr int TQ = 2048; };
template <> struct NNCfg<double> { static constexpr int T = 2; static constexpr int TQ = 1024; };

template <typename F>
hipError_t launch_nn_t(const NNPlan& pl, const void* P, const void* Q, void* part_d, int32_t* part_idx,
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

hipError_t launch_aos_to_soa(int precision, const void* aos, int n, int n_pad, void* soa, hipStream_t st, unsigned int* nonfinite, void* soa2, unsigned int* enc)
{
    if (n <= 0) return hipSuccess;
    const int blocks = (n_pad + 255) / 256;
    if (precision == ICP_F64)
        hipLaunchKernelGGL((aos_to_soa_kernel<double>), dim3(blocks), dim3(256), 0, st, (const double*)aos, n, n_pad,
                           (double*)soa, nonfinite, (double*)soa2, (unsigned int*)nullptr);
    else
        hipLaunchKernelGGL((aos_to_soa_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)aos, n, n_pad,
                           (float*)soa, nonfinite, (float*)soa2, enc);
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


template hipError_t launch_nn_t<float>(const NNPlan&, const void*, const void*, void*, int32_t*, hipStream_t);
template hipError_t launch_nn_t<double>(const NNPlan&, const void*, const void*, void*, int32_t*, hipStream_t);

// the packed fp32 kernel nn_match_f32_v2 (the plan is the dense one's: pl.sparse == 0); tl: 0 no fused tail, 1 point-to-point, 2 point-to-plane
hipError_t launch_dense_v2(const NNPlan& pl, const void* P, const void* Qscan, void* part_d, int32_t* part_idx, const RT<float>& rt, const NNFuse& fuse,
                           const NNTail& tail, int tl, bool cull, hipStream_t st)
{
    const dim3 grid(pl.blocks_x, pl.splits);
#define ICP_LAUNCH_NN2T(CU, TL)                                                                                     \
    hipLaunchKernelGGL((nn_match_f32_v2<2, 8, CU, TL>), grid, dim3(NN_BLOCK), 0, st, (const float*)P, pl.n_pad,      \
                       (const float*)Qscan, pl.m_pad, pl.seg_len, (float*)part_d, part_idx, rt, fuse, tail)
#define ICP_LAUNCH_NN2(TT, CC, CU)                                                                                  \
    hipLaunchKernelGGL((nn_match_f32_v2<TT, CC, CU, 0>), grid, dim3(NN_BLOCK), 0, st, (const float*)P, pl.n_pad,     \
                       (const float*)Qscan, pl.m_pad, pl.seg_len, (float*)part_d, part_idx, rt, fuse, tail)
    if (tl != 0) {
        const bool plane = tl == 2;
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

}  // namespace icp
