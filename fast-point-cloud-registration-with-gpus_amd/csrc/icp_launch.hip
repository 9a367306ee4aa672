// icp_launch.hip -- the launch plan (nn_plan) and the dispatch of the matching kernels (launch_nn): host code only; the kernels
// themselves live in the family files (icp_k_*.hip), which hand out their instantiations as function pointers.  Compiled by
// hipcc because the kernels' argument blocks (RT, NNFuse, NNTail: icp_device.h) are built here.
#include "icp_device.h"

#include <math.h>
#include <stdlib.h>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

namespace icp {

size_t elem_size(int precision) { return precision == ICP_F64 ? sizeof(double) : sizeof(float); }

// the instantiations of the family files
const void* sparse128_kernel_fn(int nw, int tl, bool diag, bool perm, bool hier);
const void* row64_kernel_fn(bool waves16, int tl, bool diag, bool perm);
const void* row64_f64_kernel_fn(int tl, bool waves16, bool diag);
template <typename F> hipError_t launch_nn_t(const NNPlan& pl, const void* P, const void* Q, void* part_d, int32_t* part_idx, hipStream_t st);
hipError_t launch_dense_v2(const NNPlan& pl, const void* P, const void* Qscan, void* part_d, int32_t* part_idx, const RT<float>& rt, const NNFuse& fuse,
                           const NNTail& tail, int tl, bool cull, hipStream_t st);

// ------------------------------------------------------------------------------------------------
// launch geometry + launchers
// ------------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    return atoi(v);
}

// the context's switches, read once (icp_create); ICP_NN_PHASES is parsed by the caller, who owns the log's memory
NNTuning nn_tuning_from_env()
{
    NNTuning t;
    t.sparse = env_int("ICP_NN_SPARSE", 1) ? 1 : 0;
    t.cull = env_int("ICP_NN_CULL", 1) ? 1 : 0;
    t.row = env_int("ICP_NN_ROW", 0);
    t.waves64 = env_int("ICP_NN_WAVES", 0);
    t.waves128 = env_int("ICP_NN_WAVES128", 0);
    t.cold8 = env_int("ICP_NN_COLD8", 1) ? 1 : 0;
    t.hier = env_int("ICP_NN_HIER", -1);
    t.order = env_int("ICP_NN_ORDER", 1);
    t.share = env_int("ICP_NN_SHARE", 1) ? 1 : 0;
    t.share_resident = env_int("ICP_NN_SHARE_RESIDENT", 1) ? 1 : 0;
    t.speculate = env_int("ICP_NN_SPECULATE", 1) ? 1 : 0;
    t.f64_sparse = env_int("ICP_F64_SPARSE", 1) ? 1 : 0;
    t.sort = env_int("ICP_SORT", -1);
    return t;
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

NNPlan nn_plan(int n, int m, int precision, int num_cus, const NNTuning& tune, int force_dense)
{
    NNPlan pl{};
    pl.precision = precision;
    pl.n = n;
    pl.m = m;
    pl.n_pad = pad_moving(n);
    pl.m_pad = pad_model(m);
    if (num_cus <= 0) num_cus = 256;
    pl.version = precision == ICP_F32 ? 2 : 1;
    pl.chunk = NN_CHUNK;
    if (pl.version == 2) {
        // v2: a block (4 waves) owns 64*T moving points, each wave a quarter of the block's segment.
        // 8 resident waves per SIMD = 8 blocks per CU saturate the VALU (valu_rate probe).
        if (tune.sparse && !force_dense) {
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
                const int env_hier0 = tune.hier;
                const bool hier0 = env_hier0 >= 0 ? env_hier0 != 0 : pl.m_pad >= (1 << 17);
                const int env_row = tune.row;
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
            if (S < 1) S = 1;
            // large models are searched in two levels (boxes of 64 chunks first): from 2^17 points up, where the
            // flat pass over the chunk boxes starts to dominate (ICP_NN_HIER = 0 / 1 overrides)
            const int env_hier = tune.hier;
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
            // ICP_NN_SHARE = 0 override.
            const int env_w128 = tune.waves128, env_share = tune.share;
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
            // (ICP_NN_ORDER = 0: index order; 2: also where the rows are few -- the parity tests)
            {
                const int env_order = tune.order;
                pl.order = (pl.hier && S == 1 && env_order && (env_order == 2 || pl.blocks_x >= 2 * num_cus) && pl.blocks_x < (1 << NN_ROLE_ROW_BITS)) ? 1 : 0;   // (a role holds 21 bits of row)
            }
            return pl;
        }
        // (the sweeps that chose these -- points per lane, chunk, blocks per CU, segments: profiles/r1/03_nn_sweep_cull.txt,
        // profiles/r3/r3_09_dense_kernel_sweep.txt -- are settled; their switches are gone)
        const int bpc = 8;
        const int target_blocks = num_cus * bpc;
        int T = (pl.n_pad / 256 >= target_blocks) ? 4 : 2;   // big clouds: 4 points per lane halve the LDS reads
        pl.chunk = 16;
        pl.cull = (T == 2 && tune.cull) ? 1 : 0;
        if (pl.cull) pl.chunk = 8;  // 16 partial sums per chunk would spill under the 64-VGPR cap
        pl.pts_per_thread = T;
        pl.blocks_x = pl.n_pad / (64 * T);
        if (n <= 0 || m <= 0) { pl.splits = 0; pl.seg_len = 0; return pl; }
        const int gran = 4 * pl.chunk;                        // four wave quarters of whole chunks
        int S = (target_blocks + pl.blocks_x - 1) / pl.blocks_x;
        const int max_S = (pl.m_pad + 511) / 512;             // keep >= 128 model points per wave
        if (S > max_S) S = max_S;
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
        if (tune.sparse && tune.f64_sparse && pl.n_pad / 64 <= 2 * num_cus && pl.m_pad < (1 << 17)) {
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
    pl.pts_per_thread = precision == ICP_F64 ? 2 : 4;   // (NNCfg of icp_k_dense.hip)
    pl.blocks_x = pl.n_pad / (NN_BLOCK * pl.pts_per_thread);
    if (n <= 0 || m <= 0) { pl.splits = 0; pl.seg_len = 0; return pl; }
    // small clouds cannot fill 256 CUs along the moving axis alone: split the model range over
    // grid.y until every CU holds `bpc` blocks of 4 waves.
    const int bpc = 2;
    const int target_blocks = num_cus * bpc;
    int S = (target_blocks + pl.blocks_x - 1) / pl.blocks_x;
    const int max_S = (pl.m_pad + 255) / 256;  // keep >= 256 model points per segment
    if (S > max_S) S = max_S;
    if (S < 1) S = 1;
    int seg = round_up((pl.m_pad + S - 1) / S, NN_CHUNK);
    S = (pl.m_pad + seg - 1) / seg;
    pl.splits = S;
    pl.seg_len = seg;
    return pl;
}

bool nn_can_fuse_tail(const NNPlan& pl)
{
    return ((pl.version == 2 && pl.pts_per_thread == 2 && pl.chunk == 8) || pl.version == 3) && pl.n > 0 && pl.m > 0;
}

int nn_block_threads(const NNPlan& pl) { return pl.sparse ? (pl.row == 64 ? R64_NW * 64 : (pl.nw == 8 ? 8 : (pl.nw == 4 && pl.hier) ? 4 : SP_NW) * 64) : NN_BLOCK; }

// blocks the machine holds at once of a resident kernel, per (device, kernel): asked once (the occupancy query a cooperative
// launch is built on), remembered under a lock -- contexts of several devices and threads share this table.  <= 0: unknown.
static long long resident_capacity(const void* fn, int threads)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, long long> capacity;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(mu);
    long long& slot = capacity[std::make_pair(dev, fn)];
    if (slot <= 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return 0;
        slot = (long long)per_cu * cus;
    }
    return slot;
}

static hipError_t launch_nn_v2(const NNPlan& pl, const void* P, const void* Q, void* part_d, int32_t* part_idx,
                               const NNFusedTransform* ft, const NNCullInputs* opt, const NNTailArgs* ta, hipStream_t st)
{
    static const NNTuning defaults{};
    const NNTuning& tune = (opt && opt->tune) ? *opt->tune : defaults;
    dim3 grid(pl.blocks_x, pl.splits);
    RT<float> rt{};
    NNFuse fuse{};
    fuse.n = pl.n;
    fuse.m = pl.m;
    fuse.Q_gather = (const float*)Q;
    fuse.tlog = tune.phase_log;
    fuse.tlog_cap = tune.phase_cap;
    fuse.tlog_pass = tune.phase_pass;
    // (diagnostic: the log cleared ahead of every launch -- launches of different block sizes, spare blocks and the parts of split
    // rows that do not close them would otherwise leave older stamps among the last launch's)
    if (tune.phase_wipe && tune.phase_log != nullptr && hipMemsetAsync(tune.phase_log, 0, (size_t)tune.phase_cap * sizeof(long long), st) != hipSuccess) return hipErrorInvalidValue;
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
            fuse.speculate = (ft->resident && tune.speculate) ? 1 : 0;   // (ICP_NN_SPECULATE=0: A/B runs and tests)
            // the guess: the next displacement <= twice this one + a thousandth of the group box (profiles/r2: 90 % of the lists cover)
            fuse.spec_gain = 2.0f;
            fuse.spec_floor = 1e-3f;
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
        if (tail.rows_on_device && ta->fin_tickets != nullptr && ta->fin_scratch != nullptr && ta->fin_out != nullptr && pl.sparse && pl.row != 64) {
            // rows added up inside the launch: ~sqrt(rows) ranges of rows (at most NN_FIN_GROUPS), so that the sum of a range and the
            // sum of the ranges are equally short
            const int rows = pl.blocks_x;
            int groups = (int)ceil(sqrt((double)rows));
            groups = groups < 1 ? 1 : groups > NN_FIN_GROUPS ? NN_FIN_GROUPS : groups;
            const int per = (rows + groups - 1) / groups;
            tail.fin_tickets = ta->fin_tickets;
            tail.fin_scratch = ta->fin_scratch;
            tail.fin_out = ta->fin_out;
            tail.fin_host = ta->fin_host;
            tail.fin_rows = rows;
            tail.fin_per = per;
            tail.fin_groups = (rows + per - 1) / per;
        }
    }
    if (pl.sparse) {
        // the plan's geometry is the sparse kernel's: it needs the scan copy and its chunk boxes
        if (!(opt && opt->Q_scan && opt->boxes)) return hipErrorInvalidValue;
        if (pl.m_pad >= (1 << 28)) return hipErrorInvalidValue;  // the in-block merge key carries 28 index bits
        fuse.seed_idx = opt->seed_idx;
        fuse.boxes = (const float*)opt->boxes;
        fuse.q_perm = opt->Q_scan_sorted ? opt->q_perm : nullptr;
        fuse.p_perm = opt->p_perm;
        const void* Qsp = opt->Q_scan_sorted ? opt->Q_scan_sorted : opt->Q_scan;
        fuse.samples = (const float*)opt->samples;
        // the cold start's full round (its probe round is 8 groups): 64 groups = 512 samples on a small model -- measured
        // (round 2, rows of 64): as good as 2048 on the 128 x 128 grid, Bunny_res and a random cloud, and 4 us less of a cold
        // pass on the hall scan, where a few blocks take the full round without gaining from it; 2048 on large models
        fuse.sample_groups = pl.m_pad <= 32768 ? 64 : 256;
        // (round 3: with the refinement round of the traversal -- local samples, where the row's neighbours are -- the coarse sample
        // round of a SEEDED pass cost more than it added, 5.14 -> 5.10 ms without: removed in round 4)
        // seeded: few hits, long rounds; cold: short rounds so that the exchanged minima start pruning early
        const int env_waves = tune.waves64;
        int cus64 = 0, dev64 = 0;
        if (pl.row == 64 && (hipGetDevice(&dev64) != hipSuccess || hipDeviceGetAttribute(&cus64, hipDeviceAttributeMultiprocessorCount, dev64) != hipSuccess)) cus64 = 256;
        const int nw64 = (pl.row == 64 && (env_waves == 16 || opt->waves64 == 16) && pl.blocks_x <= cus64) ? 16 : R64_NW;
        const int max_passes = pl.row == 64 ? SP_HCAP / (nw64 * 64) : pl.hier ? SP_MAX_PASSES : SP_HCAP_FLAT / (((pl.nw == 8 && ta) ? 8 : SP_NW) * 64);
        int passes = (fuse.seed_idx || fuse.samples) ? max_passes : 1;
        if (pl.row == 64) {
            // ---- 64-point rows ----
            if (pl.splits != 1 || pl.hier) return hipErrorInvalidValue;
            const bool diag = fuse.tlog != nullptr || fuse.work != nullptr, perm = fuse.q_perm != nullptr;
            const int tl = !ta ? 0 : (ta->metric == ICP_POINT_TO_PLANE ? 2 : 1);
            const void* fn = row64_kernel_fn(nw64 == 16, tl, diag, perm);
            const float* Pp = (const float*)P;
            const float* Qp = (const float*)Qsp;
            int n_pad = pl.n_pad, m_pad = pl.m_pad;
            float* pd = (float*)part_d;
            void* args[] = {&Pp, &n_pad, &Qp, &m_pad, &passes, &pd, &part_idx, &rt, &fuse, &tail};
            const dim3 g64(pl.blocks_x, 1);
            if (fuse.resident) {
                if (!ta) return hipErrorInvalidValue;
                // every block must be on the machine at once (see below): blocks <= CUs x resident blocks per CU
                const long long cap = resident_capacity(fn, nw64 * 64);
                if (cap <= 0) return hipErrorCooperativeLaunchTooLarge;
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
            const bool cold8 = cold_launch && tune.cold8 && !(opt->row_order != nullptr && opt->order_history);
            const int nw = (pl.nw == 8 && tl != 0) ? 8 : (pl.nw == 4 && tl != 0 && hier) ? (cold8 ? 8 : 4) : SP_NW;
            if (passes > (hier ? SP_MAX_PASSES : SP_HCAP_FLAT / (nw * 64))) passes = hier ? SP_MAX_PASSES : SP_HCAP_FLAT / (nw * 64);
            if (!hier && (pl.m_pad >> 3) > 65536) return hipErrorInvalidValue;   // (the flat search lists 16-bit chunk numbers; nn_plan never asks for it)
            const void* fn = sparse128_kernel_fn(nw, tl, diag, perm, hier);
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
                (!fuse.resident || (opt->seed_pub != nullptr && tune.share_resident)) &&   // (ICP_NN_SHARE_RESIDENT=0: a resident launch keeps one block per row -- A/B runs)
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
                fuse.share_min = 8 * nw;      // one batch for every wave (32 .. 128 measured: no difference)
                g = dim3(pl.share_blocks, 1);
            }
            fuse.seed_pub = opt->seed_pub;
            fuse.records = hier ? opt->records : nullptr;
            // (a round of the chunk find covers 16 listed super boxes -- the hit list would hold 64: every round starts from the largest
            // bound the rounds before have left, so shorter rounds list less; 10 M x 10 M, rounds of 64 / 32 / 16 / 8: 1.41 / 1.29 / 1.19 /
            // 1.13 G chunks listed per registration, 5.93 / 5.75 / 5.65 / 5.69 ms per iteration.  ICP_NN_ROUND_SUPERS for the A/B)
            // (the refinement round of a pass that lists >= 12 super boxes, over <= 256 of their chunk samples: 10 M x 10 M 5.68 -> 5.14 ms,
            // 1.19 -> 0.99 G chunks listed and 370 -> 327 M evaluated per registration; 24 / 512: 5.26, 12 / 128: 5.15, 8 / 256: 5.17.
            // The switches of those A/B runs are gone with round 4.)
            fuse.refine_min = 12;
            fuse.refine_cnt = 256;
            fuse.round_supers = 16;
            if (hier && fuse.records == nullptr) return hipErrorInvalidValue;   // (the hierarchical search fetches its hits from the records)
            if (pl.order && ta && !fuse.resident && opt->row_order != nullptr && opt->row_hits != nullptr) {
                fuse.row_order = opt->row_order;
                fuse.row_hits = opt->row_hits;
                fuse.xcd_shift = NN_ORDER_XCD_SHIFT;
                g = dim3(pl.blocks_x + NN_ORDER_EXTRA, 1);   // (the roles of the blocks beyond the rows: parts of split rows, or none)
            }
            if (fuse.resident) {
                if (!ta || pl.splits != 1) return hipErrorInvalidValue;
                // Every block must be on the machine at once (they all wait for the same host).  A cooperative launch
                // guarantees that or refuses, but costs ~13 us more per launch here; the same guarantee comes from the
                // occupancy query it is built on: the grid fits iff blocks <= CUs x resident blocks per CU.  Blocks that
                // start late (behind the previous kernel of the stream) only delay the first pass, nothing waits on them
                // that they cannot deliver.
                const long long cap = resident_capacity(fn, nw * 64);
                if (cap <= 0) return hipErrorCooperativeLaunchTooLarge;
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
    return launch_dense_v2(pl, P, Qscan, part_d, part_idx, rt, fuse, tail, ta ? (ta->metric == ICP_POINT_TO_PLANE ? 2 : 1) : 0, cull, st);
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
    fuse.samples = (const float*)opt->samples;
    fuse.sample_groups = 256;
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
    // icp_set_work_counting: the instrumented instantiation (point-to-point rows only: what the fp64 loop of src/ICP_CPU.c runs)
    fuse.work = (opt->work != nullptr && tl == 1) ? opt->work : nullptr;
    const void* fn = row64_f64_kernel_fn(tl, nw == 16, fuse.work != nullptr);
    if (fuse.resident) {
        // every block must be on the machine at once: blocks <= CUs x resident blocks per CU (the occupancy query)
        if (!ta) return hipErrorInvalidValue;
        const long long cap = resident_capacity(fn, nw * 64);
        if (cap <= 0 || (long long)pl.blocks_x > cap) return hipErrorCooperativeLaunchTooLarge;
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

}  // namespace icp
