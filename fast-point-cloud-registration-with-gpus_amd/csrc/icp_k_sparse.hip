// icp_k_sparse.hip -- nn_match_sparse: the pruned exact nearest-neighbour search over rows of 128 moving points (flat: Bunny.csv-sized
// clouds with shared rows; hierarchical: configs[4]).  Replaces Matching<<<>>> src/CUDA/GPU_point_to_point_real.cu:38-79 + the
// minimisation chain src/ICP_point_to_point.cu:308-357 (fused tail) + RyT / error :81-88,403-416 (fused front end).
#include "icp_device_sparse.h"
#include <math.h>
#include <stdlib.h>
#include <cstring>

namespace icp {

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
            // Blocks b and b + 8 run on one XCD (observed dispatch order; speed only): G = 2^xcd_shift CONSECUTIVE positions of the order --
            // rows of one weight class in Hilbert order, i.e. neighbours that list mostly the same chunks, and the parts of a split row --
            // go to blocks of one XCD, which then fetches their records into its L2 once instead of every XCD fetching every record;
            // every XCD still gets every eighth group, so the heaviest-first order holds per XCD.
            int bpos = (int)blockIdx.x;
            {
                const int sh = fuse.xcd_shift, span = 8 << sh;
                if (sh > 0 && bpos < ((int)gridDim.x / span) * span) {
                    const int x = bpos & 7, i = bpos >> 3;
                    bpos = ((((i >> sh) << 3) + x) << sh) + (i & ((1 << sh) - 1));
                }
            }
            const int ro = fuse.row_order[bpos];   // (one word for the whole block)
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
        // (a slot holds a real point iff its own number is below n: the slot map permutes [0, n) and leaves the padding slots to
        // themselves -- prep_slot_map_kernel -- so the test does not wait for the map's load, one dependent trip to memory less
        // in every block's front end; the point's number itself is needed at the stores only)
        real[t] = ibase + t * 64 < fuse.n;
        sok[t] = false;
        sq[t][0] = sq[t][1] = sq[t][2] = 0.f;
        if (from_slots) {
            const float* ss = fuse.slot_state + 3 * (size_t)n_pad + (ibase + t * 64);
            sq[t][0] = ss[0]; sq[t][1] = ss[(size_t)n_pad]; sq[t][2] = ss[2 * (size_t)n_pad];
            sok[t] = real[t];
        } else {
            // no previous match (cold start): the model point at the same RELATIVE index -- consecutive scans of one
            // sensor, or a cloud and its moved copy, keep their order, and any valid index is a valid bound
            const int i = fresh(pi[t]);
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
                    // (one store, a tag in every 32-byte sector: tail_reduce_store)
                    if (lane < NN_CROW) __hip_atomic_store(&row[lane], (lane & 3) == 0 ? crow_pack(lane == 0 ? err_row : 0.0, row_tag_lo) : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
    if (!have_seeds && fuse.samples != nullptr) {
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
            // (round 4 tried the hits two at a time -- both box tests first, their LDS reads independent, the second made once more when
            // the first went on to the exact arithmetic: 10 M x 10 M 4.93 against 4.90 ms per iteration one by one, the share of one rank of
            // eight 16.6 against 16.8 ms per 30 iterations, same box -- nothing: four blocks to a CU already cover an LDS round trip.
            // profiles/r4/r4_06_s5_ab_*.txt.  Removed.)
            for (int rr = 0; rr < cnt; ++rr) {
                int stage_reached;
                if constexpr (PERM) {
                    stage_reached = scan_hit<true, true>(stage + rr * STG, 0, px, py, pz, best, bj, bq);
                } else {
                    const int ch = __builtin_amdgcn_readfirstlane((int)hits[hb + rr * NWS + w]);
                    stage_reached = scan_hit<false, true>(stage + rr * STG, ch, px, py, pz, best, bj, bq);
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
                int SH = *scount;
                if (tb == t_lo && tg == 0 && fuse.samples != nullptr && fuse.refine_min > 0) {
                  // (round 4 tried a SECOND round -- the super boxes listed again with the bound the first round has left, compacted, their
                  // samples taken several times denser: 10 M x 10 M 4.884 against 4.891 ms per iteration with one round, the share of one
                  // rank of eight 16.5-16.8 against 16.6-16.7 ms per 30 iterations, same box: nothing; profiles/r4/r4_03_s5_ab_*.txt.  Removed.)
                  if (SH >= fuse.refine_min) {
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


// the instantiation a launch runs: nw waves per block (16, 8 or 4), tl = 0 no fused tail / 1 point-to-point / 2 point-to-plane rows,
// diag = phase stamps + work counters, perm = sorted views, hier = box hierarchy.  8-wave blocks exist with a fused tail only,
// 4-wave blocks with a fused tail and the hierarchical search only (NULL otherwise).
const void* sparse128_kernel_fn(int nw, int tl, bool diag, bool perm, bool hier)
{
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
    if (tl < 0 || tl > 2) return nullptr;
    return fns[nw == 8 ? 1 : nw == 4 ? 2 : 0][tl][diag ? 1 : 0][perm ? 1 : 0][hier ? 1 : 0];
}

}  // namespace icp
