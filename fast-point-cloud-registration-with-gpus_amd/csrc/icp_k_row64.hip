// icp_k_row64.hip -- nn_match_row64: the same search laid out for latency -- rows of 64 moving points, one per lane (the hall scan and
// everything up to 32 768 points against a model below 2^17).  Replaces the same reference statements as icp_k_sparse.hip.
#include "icp_device_sparse.h"
#include <math.h>
#include <stdlib.h>
#include <cstring>

namespace icp {

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
                    // (one store, a tag in every 32-byte sector: tail_reduce_store)
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


const void* row64_kernel_fn(bool waves16, int tl, bool diag, bool perm)
{
#define ICP_R64_FN(TL, W) {{(const void*)nn_match_row64<TL, false, false, W>, (const void*)nn_match_row64<TL, false, true, W>},   \
                           {(const void*)nn_match_row64<TL, true, false, W>, (const void*)nn_match_row64<TL, true, true, W>}}
    static const void* const fns[2][3][2][2] = {{ICP_R64_FN(0, 8), ICP_R64_FN(1, 8), ICP_R64_FN(2, 8)},
                                                {ICP_R64_FN(0, 16), ICP_R64_FN(1, 16), ICP_R64_FN(2, 16)}};
#undef ICP_R64_FN
    if (tl < 0 || tl > 2) return nullptr;
    return fns[waves16 ? 1 : 0][tl][diag ? 1 : 0][perm ? 1 : 0];
}

}  // namespace icp
