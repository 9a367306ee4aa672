// icp_device.h -- device-side helpers shared by the gfx950 kernel families (icp_k_*.hip): small wave-level primitives, the
// packed (dx*dx + dy*dy) + dz*dz arithmetic, the box tests, and the argument blocks of the fused matching kernels (NNFuse:
// front end, NNTail: row tail).  Internal to libicp_mi355x.so.  Every translation unit that includes this is compiled with
// FP contraction OFF (-ffp-contract=off + the pragma below): the matching kernels must round every sub / mul / add separately
// (bit-exact correspondences against src/ICP_CPU.c:227-231); tests/test_host.py greps the ISA of every matching kernel.
#pragma once
#include "icp_kernels.h"

#pragma clang fp contract(off)

namespace icp {

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

// (q.lo - p.S, q.hi - p.S): ONE of the lane's two points (half S of p) against two model points -- src1's half by op_sel, negated
template <int S>
__device__ __forceinline__ f2 pk_sub_sel(f2 q, f2 p)
{
    f2 r;
    if constexpr (S == 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(q), "v"(p));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(q), "v"(p));
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
    // Rows added up INSIDE the launch (round 4; clouds of more than 1024 rows, nn_match_sparse): the rows stay in device memory, are
    // grouped into <= NN_FIN_GROUPS contiguous ranges, and whoever closes the LAST row of a range (a ticket per range) adds the
    // range's rows in index order into fin_scratch; whoever closes the last RANGE (one more ticket) adds the ranges in order and
    // leaves the launch's ICP_NMOM vector in fin_out -- pinned host memory with the pass's tag in its last slot (the host polls ONE
    // tag instead of walking thousands of rows; no finalize kernels, no copy back, no synchronisation), or the device vector a
    // collective follows on.  Fixed ranges, fixed order: the same bits whatever order the blocks ran in.  NULL: not used.
    unsigned int* fin_tickets; // [NN_FIN_GROUPS + 1], zero between launches (the closers reset them)
    double* fin_scratch;       // [NN_FIN_GROUPS][ICP_NMOM]
    double* fin_out;           // [ICP_NMOM]
    int fin_rows, fin_per, fin_groups;
    int fin_host;              // fin_out is host memory: system-scope stores, the tag last
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
    int xcd_shift;             // ordered rows: 2^xcd_shift consecutive positions of the order go to blocks of ONE XCD (0: position = block index)
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

// geometry of the sparse kernels' blocks and hit lists (the launchers in icp_launch.hip size their rounds by these)
constexpr int SP_NW = 16;                       // waves per block (the default; NWS = 8 is the other instantiation)
constexpr int SP_HCAP = 4096;                   // hit-list entries of the hierarchical search = chunks per round (SP_NW * 64 * passes <= this)
constexpr int SP_MAX_PASSES = SP_HCAP / (SP_NW * 64);
// flat search (models below 2^19 points = 65 536 chunks): the list holds 16-bit chunk numbers, twice as many in the same
// 16 KB -- a model of up to 65 536 points is one round of the find (Bunny.csv: 5040 chunks, two rounds with 4096 entries)
constexpr int SP_HCAP_FLAT = 2 * SP_HCAP;
constexpr int R64_NW = 8;                       // rows of 64 points: waves per block (16 with the device to itself)

}  // namespace icp
