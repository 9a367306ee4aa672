// icp_device_sparse.h -- what the sparse matching kernels share (nn_match_sparse: rows of 128 points, nn_match_row64: rows of
// 64): the hit-list constants, one hit chunk against a lane's packed pair of points, and the row tail (moment sums of a row by
// one wave, LDS transpose, system-scope row stores + tag).  See icp_k_sparse.hip for the search these belong to.
#pragma once
#include "icp_device.h"

namespace icp {

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

// ONE of the lane's two points (half S of px, py, pz; its best / bj / bq) against the chunk, two MODEL points per packed operation:
// half the arithmetic of the two-point form.  scan_hit takes it when only the row's first 64 slots, or only its last 64, ask for the
// chunk (the halves are the two halves of the row's stretch of the Hilbert curve: spatially apart).  Same roundings (every
// difference, product and sum is the one the two-point form computes for that pair), same tie rule.
template <bool PERM, int S>
__device__ __forceinline__ void scan_hit_half(const float* sb, int ch, const f2 px, const f2 py, const f2 pz, float& best, int& bj, float (&bq)[3])
{
    constexpr int C = 8;
    const float *qxp = sb + 8, *qyp = sb + 16, *qzp = sb + 24;
    const float4 qxa = *reinterpret_cast<const float4*>(qxp), qxb = *reinterpret_cast<const float4*>(qxp + 4);
    const float4 qya = *reinterpret_cast<const float4*>(qyp), qyb = *reinterpret_cast<const float4*>(qyp + 4);
    const float4 qza = *reinterpret_cast<const float4*>(qzp), qzb = *reinterpret_cast<const float4*>(qzp + 4);
    f2 d[C / 2];   // d[k] = {model point 2k, model point 2k+1}
    {
        f2 ax, ay, az;
        ax = pk_sub_sel<S>(f2{qxa.x, qxa.y}, px); ay = pk_sub_sel<S>(f2{qya.x, qya.y}, py); az = pk_sub_sel<S>(f2{qza.x, qza.y}, pz); d[0] = (ax * ax + ay * ay) + az * az;
        ax = pk_sub_sel<S>(f2{qxa.z, qxa.w}, px); ay = pk_sub_sel<S>(f2{qya.z, qya.w}, py); az = pk_sub_sel<S>(f2{qza.z, qza.w}, pz); d[1] = (ax * ax + ay * ay) + az * az;
        ax = pk_sub_sel<S>(f2{qxb.x, qxb.y}, px); ay = pk_sub_sel<S>(f2{qyb.x, qyb.y}, py); az = pk_sub_sel<S>(f2{qzb.x, qzb.y}, pz); d[2] = (ax * ax + ay * ay) + az * az;
        ax = pk_sub_sel<S>(f2{qxb.z, qxb.w}, px); ay = pk_sub_sel<S>(f2{qyb.z, qyb.w}, py); az = pk_sub_sel<S>(f2{qzb.z, qzb.w}, pz); d[3] = (ax * ax + ay * ay) + az * az;
    }
    float c0 = fmin_(fmin_(d[0].x, d[0].y), fmin_(d[1].x, d[1].y));   // the chunk's own minimum
    c0 = fmin_(c0, fmin_(fmin_(d[2].x, d[2].y), fmin_(d[3].x, d[3].y)));
    auto dk = [&](int k) { return (k & 1) ? d[k >> 1].y : d[k >> 1].x; };
    if constexpr (PERM) {
        const bool cand = c0 <= best;
        if (__builtin_amdgcn_ballot_w64(cand) != 0ull) {
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

template <bool PERM, bool HALVES = false>
__device__ __forceinline__ int scan_hit(const float* sb, int ch, const f2 px, const f2 py, const f2 pz, float (&best)[2], int (&bj)[2],
                                        float (&bq)[2][3])
{
    constexpr int C = 8;
    {
        // level 0: the chunk's bounding box against each of the lane's points (ties pass: the hits are unordered)
        if constexpr (HALVES) {
            // (the test of box_may_improve<1, true>, its two verdicts kept apart)
            const f2 gx = px - f2{__builtin_amdgcn_fmed3f(px.x, sb[0], sb[3]), __builtin_amdgcn_fmed3f(px.y, sb[0], sb[3])};
            const f2 gy = py - f2{__builtin_amdgcn_fmed3f(py.x, sb[1], sb[4]), __builtin_amdgcn_fmed3f(py.y, sb[1], sb[4])};
            const f2 gz = pz - f2{__builtin_amdgcn_fmed3f(pz.x, sb[2], sb[5]), __builtin_amdgcn_fmed3f(pz.y, sb[2], sb[5])};
            f2 L = (gx * gx + gy * gy) + gz * gz;
            L = L * f2{0.99999905f, 0.99999905f};
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(L.x <= best[0]), m1 = __builtin_amdgcn_ballot_w64(L.y <= best[1]);
            if ((m0 | m1) == 0ull) return 0;
            if (m1 == 0ull) { scan_hit_half<PERM, 0>(sb, ch, px, py, pz, best[0], bj[0], bq[0]); return 2; }
            if (m0 == 0ull) { scan_hit_half<PERM, 1>(sb, ch, px, py, pz, best[1], bj[1], bq[1]); return 2; }
        } else {
        const f2 pxa[1] = {px}, pya[1] = {py}, pza[1] = {pz};
        if (__builtin_amdgcn_ballot_w64(box_may_improve<1, true>(sb[0], sb[1], sb[2], sb[3], sb[4], sb[5], pxa, pya, pza, best)) == 0ull) return 0;
        }
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

// The in-launch finalize (NNTail::fin_*): called by the wave that has just stored row `rowi` (agent-scope stores, drained).  Lane
// (part, k) = (lane >> 5, lane & 31) adds slot k of every second row / range, eight loads in flight; part 0 + part 1 at the end.
__device__ __forceinline__ void fin_close(const NNTail& tail, unsigned int rowi, int lane)
{
    const int per = tail.fin_per, g = (int)rowi / per, b0 = g * per;
    const int cnt = min(per, tail.fin_rows - b0);
    unsigned int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(&tail.fin_tickets[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
    if (t != (unsigned int)(cnt - 1)) return;
    const int k = lane & 31, part = lane >> 5;
    auto add_in_order = [&](const double* src, int count) {   // src[r * ICP_NMOM + k], r = part, part + 2, ...
        double s = 0.0;
        if (k != ICP_NMOM - 1)                                  // (the rows' tag slot is not a moment)
            for (int r0 = part; r0 < count; r0 += 16) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + 2 * j;
                    v[j] = r < count ? __hip_atomic_load(&src[(size_t)r * ICP_NMOM + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
            }
        const double o = __shfl_xor(s, 32, 64);
        return part == 0 ? s + o : o + s;                       // even rows + odd rows, in both halves of the wave
    };
    const double tot = add_in_order(tail.rows + (size_t)b0 * ICP_NMOM, cnt);
    if (lane == 0) __hip_atomic_store(&tail.fin_tickets[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    if (lane < ICP_NMOM) __hip_atomic_store(&tail.fin_scratch[(size_t)g * ICP_NMOM + lane], lane == ICP_NMOM - 1 ? 0.0 : tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int t2 = 0;
    if (lane == 0) t2 = __hip_atomic_fetch_add(&tail.fin_tickets[NN_FIN_GROUPS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t2 = (unsigned int)__builtin_amdgcn_readfirstlane((int)t2);
    if (t2 != (unsigned int)(tail.fin_groups - 1)) return;
    const double all = add_in_order(tail.fin_scratch, tail.fin_groups);
    if (lane == 0) __hip_atomic_store(&tail.fin_tickets[NN_FIN_GROUPS], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tail.fin_host != 0) {
        // the host polls the tag in the vector's last slot: the sums first, drained, then the tag (as a row's)
        if (lane < ICP_NMOM - 1) __hip_atomic_store(&tail.fin_out[lane], all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&tail.fin_out[ICP_NMOM - 1], tail.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (lane < ICP_NMOM - 1) {
        __hip_atomic_store(&tail.fin_out[lane], all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by what follows on the stream)
    }
}

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
    if constexpr (ROWARG) {
        if (tail.rows_on_device != 0 && tail.fin_tickets != nullptr) {
            // (round 4: the rows are added up inside this launch -- agent-scope stores, drained, then the tickets: fin_close)
            if (lane < NACC) {
                double sum = tp[lane];
#pragma unroll
                for (int q = 1; q < PARTS; ++q) sum += tp[q * NACC + lane];
                __hip_atomic_store(&row[1 + lane], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) __hip_atomic_store(&row[ICP_MOM_ERR], err_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ICP_PHASE(8)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fin_close(tail, rowi, lane);
            return;
        }
    }
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
    if (compact) {
        // Round 4: the compact row leaves in ONE store instruction -- sixteen lanes x 8 bytes, 128 contiguous bytes -- and every 32-byte
        // sector of it carries the pass's tag in the low NN_CROW_TAG_BITS mantissa bits of its first slot (slots 0, 4, 8, 12: the error
        // share, sum q.x, two of the nine sums q p^T -- 2^-36 of a sum of products of floats is nothing, and every form of the loop that
        // uses this row format masks the same bits, so they stay bit-identical among themselves).  However the fabric splits the store
        // on its way, a sector whose tag is the awaited one holds this pass's sums: the host takes a row when all four tags match.  The
        // wait for the stores' acknowledgement ahead of a separate tag store (0.45 us of every pass) is gone.
        double v = 0.0;
        if (lane < NACC) {
            v = tp[lane];
#pragma unroll
            for (int q = 1; q < PARTS; ++q) v += tp[q * NACC + lane];   // (slot `lane` of the sums: the count for lane 0, then sum p, sum q, sum q p^T)
        }
        if (lane == 0) v = err_row;
        if ((lane & 3) == 0) v = crow_pack(v, tail.tag_lo);
        ICP_PHASE(8)
        if (lane < NN_CROW) __hip_atomic_store(&row[lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (lane < NACC) {
        double sum = tp[lane];
#pragma unroll
        for (int q = 1; q < PARTS; ++q) sum += tp[q * NACC + lane];
        __hip_atomic_store(&row[1 + lane], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (lane == 0) __hip_atomic_store(&row[ICP_MOM_ERR], err_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    ICP_PHASE(8)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&row[ICP_NMOM - 1], tail.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace icp
