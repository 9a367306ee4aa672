// icp_k_f64.hip -- nn_match_row64_f64: the CPU path's precision (src/ICP_CPU.c:220-234) on the sparse structure, rows of 64 points.
#include "icp_device_sparse.h"
#include <math.h>
#include <stdlib.h>
#include <cstring>

namespace icp {

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
                    // (never taken: the fp64 path keeps the full row format; kept in the format's own form -- tail_reduce_store)
                    if (lane < NN_CROW) __hip_atomic_store(&row[lane], (lane & 3) == 0 ? crow_pack(lane == 0 ? err_row : 0.0, row_tag_lo) : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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


const void* row64_f64_kernel_fn(int tl, bool waves16, bool diag)
{
    static const void* const fns[3][2] = {{(const void*)nn_match_row64_f64<0, 8>, (const void*)nn_match_row64_f64<0, 16>},
                                          {(const void*)nn_match_row64_f64<1, 8>, (const void*)nn_match_row64_f64<1, 16>},
                                          {(const void*)nn_match_row64_f64<2, 8>, (const void*)nn_match_row64_f64<2, 16>}};
    if (tl < 0 || tl > 2) return nullptr;
    if (diag) return waves16 ? (const void*)nn_match_row64_f64<1, 16, true> : (const void*)nn_match_row64_f64<1, 8, true>;
    return fns[tl][waves16 ? 1 : 0];
}

}  // namespace icp
