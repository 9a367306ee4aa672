// icp_kernels.h -- host-callable launchers of the gfx950 kernels.  The kernels live in one translation unit per family
// (same flags, gfx950 only; shared device helpers in icp_device.h / icp_device_sparse.h):
//   icp_k_sparse.hip  nn_match_sparse      rows of 128 points: shared rows (Bunny.csv), box hierarchy (configs[4])
//   icp_k_row64.hip   nn_match_row64       rows of 64 points: the hall scan and everything up to 32 768 points
//   icp_k_f64.hip     nn_match_row64_f64   the CPU path's precision on the same structure
//   icp_k_dense.hip   nn_match_kernel / nn_match_f32_v2 (every pair), merge, moments, transform + error, finalize, layout
//   icp_k_plane.hip   kNN(4) + normals, OS1 decode + conversion
//   icp_k_setup.hip   duplicates, spatial order, boxes / samples / records, row order + roles, the control block of a pass
//   icp_launch.hip    the plan (nn_plan) and the dispatch (launch_nn): host code only
// Internal to libicp_mi355x.so; the public surface is include/icp_mi355x.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <stdint.h>

#include "../../include/icp_mi355x.h"

namespace icp {

// Internal HBM layout of a cloud: SoA, x[pad] | y[pad] | z[pad], `pad` >= count.
//   moving cloud: pad = multiple of NN_POINT_ALIGN (whole NN blocks, no bounds checks in the hot loop)
//   model  cloud: pad = multiple of NN_CHUNK; entries [m, m_pad) replicate point m-1, which can
//                 never win a first-minimum search against its lower-index original.
constexpr int NN_BLOCK = 256;        // threads per matching block (4 wave64)
constexpr int NN_POINT_ALIGN = 1024; // moving-point padding granule
constexpr int NN_CHUNK = 16;         // model points per index-tracking chunk

struct NNPlan {
    int precision;  // ICP_F32 / ICP_F64
    int n, m;       // real counts
    int n_pad, m_pad;
    int pts_per_thread; // T
    int blocks_x;       // n_pad / (NN_BLOCK * T)
    int splits;         // S: model segments scanned by different blocks (grid.y)
    int seg_len;        // model points per segment (multiple of NN_CHUNK)
    int version;        // 1: generic kernel (dense fp64, A/B), 2: packed fp32 kernels, 3: fp64 on the sparse structure (rows of 64)
    int chunk;          // index-tracking chunk of the launched kernel
    int cull;           // the packed kernel may use the seeded-bound / xy early-out variant
    int sparse;         // the geometry is the sparse kernel's (16-wave blocks of 128 moving points; needs chunk boxes)
    int hier;           // sparse kernel: two-level search (boxes of 64 chunks first) -- large models
    int row;            // sparse geometry: moving points per block row -- 128 (nn_match_sparse, 16 waves) or 64 (nn_match_row64, 8 waves)
    int nw;             // rows of 128: waves per block -- 16, or 8 (two blocks per CU: clouds whose rows outnumber the CUs; launches with a fused tail),
                        // or 4 (four per CU: the hierarchical search of clouds with rows for several rounds of blocks)
    int share_blocks;   // rows of 128, 8 waves, one launch per pass: blocks of a launch (> blocks_x: the spare ones go to the heavy rows), or 0
    int order;          // rows of 128, many more rows than the machine holds at once: the blocks of a launch take the rows heaviest first
                        // (by the hits of the launch before) -- see launch_row_order
};
int nn_block_threads(const NNPlan& pl);

// Every switch the plan and the launchers look at, read ONCE per context (icp_create -> nn_tuning_from_env): nothing on the
// launch path scans the environment (the advisor's finding on round 3), and a test that wants another form creates another
// context.  Defaults = production.  All of these select among forms that give the same bits (INTEGRATION.md lists them).
struct NNTuning {
    int sparse = 1;          // ICP_NN_SPARSE=0: the dense packed kernel (every pair executed; no boxes, no hierarchy)
    int cull = 1;            // ICP_NN_CULL=0: ... without its seeded early-out
    int row = 0;             // ICP_NN_ROW=64 / 128: force the row size of the sparse kernels
    int waves64 = 0;         // ICP_NN_WAVES=16: rows of 64 points as 16-wave blocks (what icp_set_exclusive selects)
    int waves128 = 0;        // ICP_NN_WAVES128=4 / 8 / 16: waves per block of the rows of 128
    int cold8 = 1;           // ICP_NN_COLD8=0: a plan of 4-wave blocks runs its cold launches on 4 waves too
    int hier = -1;           // ICP_NN_HIER=0 / 1: box hierarchy never / always (-1: by the model's size)
    int order = 1;           // ICP_NN_ORDER=0 / 2: rows in index order / heaviest first also where the rows are few
    int share = 1;           // ICP_NN_SHARE=0: no shared rows
    int share_resident = 1;  // ICP_NN_SHARE_RESIDENT=0: a resident launch keeps one block per row
    int speculate = 1;       // ICP_NN_SPECULATE=0: resident launches without their speculative hit list
    int f64_sparse = 1;      // ICP_F64_SPARSE=0: ICP_F64 clouds on the dense thread-per-point kernel
    int sort = -1;           // ICP_SORT=0 / 1: spatially sorted views never / always (-1: by the extent test)
    // ICP_NN_PHASES=file[:pass[:slots[:wipe]]] -- per-wave phase stamps of the matching kernels (tools/phase_report.py); the
    // context owns the log
    long long* phase_log = nullptr;
    long long phase_cap = 0;
    int phase_pass = -1;
    int phase_wipe = 0;
};
NNTuning nn_tuning_from_env();

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }
inline int pad_moving(int n) { return n <= 0 ? 0 : round_up(n, NN_POINT_ALIGN); }
inline int pad_model(int m) { return m <= 0 ? 0 : round_up(m, NN_CHUNK); }

// Choose the launch geometry.  `num_cus` comes from hipDeviceProp_t::multiProcessorCount.
// force_dense != 0: the geometry of the dense packed kernel (every pair executed) even where the sparse kernel would run
NNPlan nn_plan(int n, int m, int precision, int num_cus, const NNTuning& tune, int force_dense = 0);

size_t elem_size(int precision);

// the transform of the previous pass, fused into the front of the matching kernel (fp32 kernel only):
// P_out <- R * P_in + t, err_rows[block_x] <- sum |p_new - q[idx_prev]|^2
// Mailbox of an armed / resident launch: ONE 64-byte line, written whole by the host and read whole by ONE load of the
// waiting wave (16 lanes x 4 bytes) -- detecting the message and receiving it are the same memory round trip.
// The line is two 32-byte halves, each carrying the message's tag in its last word next to its share of the payload:
//   w[0..6] = rt[0..6]   w[7] = tag   |   w[8..12] = rt[7..11]   w[13] = cmd   w[14] = tag   w[15] = 0
// (rt = R row-major, then t, rounded to the storage precision).  The host writes each half with one 32-byte vector
// store; whatever the fabric does with the line on its way (a write-combining buffer flushed in two pieces, a read split
// into sectors), a half whose tag is new carries new payload, and the reader accepts the line only when BOTH tags are
// the one it waits for.  tag = low 31 bits of the pass's sequence number, top bit set (a cleared mailbox never
// matches); cmd = ICP_CMD_EXIT under the awaited tag withdraws the kernel.  Where the mailbox lives in host memory,
// block 0 polls it and relays the line to the other blocks through a copy in device memory (`relay`).
#if defined(__HIPCC__)
#define ICP_HOST_DEVICE __host__ __device__
#else
#define ICP_HOST_DEVICE
#endif
struct alignas(64) NNMailbox {
    uint32_t w[16];
};
enum { ICP_MB_TAG0 = 7, ICP_MB_CMD = 13, ICP_MB_TAG1 = 14 };
ICP_HOST_DEVICE inline uint32_t mailbox_tag(double seq) { return (uint32_t)(unsigned long long)seq | 0x80000000u; }
ICP_HOST_DEVICE inline int mailbox_rt_word(int k) { return k < 7 ? k : k + 1; }   // the word rt[k] travels in
enum { ICP_CMD_EXIT = 0, ICP_CMD_MATCH = 1, ICP_CMD_TRANSFORM_MATCH = 2, ICP_CMD_TRANSFORM_ONLY = 3 };
// The same for a registration in double: twelve doubles do not fit one cache line, so the message is TWO lines in four
// 32-byte parts, part h = {rt[3h], rt[3h+1], rt[3h+2] (6 words), cmd, tag}; each part is written by one 32-byte store and
// the reader accepts the message only when all four tags are the awaited one.  A mailbox slot is 128 bytes for both
// precisions (the float message uses its first line).
struct alignas(128) NNMailbox64 {
    uint32_t w[32];
};
enum { ICP_MB64_CMD = 6 };   // (within each part; tags: word 7 of each part)
// How long a block waits for a message before it gives up (which reads as EXIT), in WALL-CLOCK seconds of the device's
// constant 100 MHz counter -- the same budget whatever memory the poll goes to.  Blocks that listen to block 0's relay
// wait twice as long: block 0 decides, and publishes its verdict (message or EXIT) through the relay.  The host side of
// the contract (icp_api.cpp, kMailLeaseS) never posts a message a block might no longer be waiting for.
#define ICP_MAILBOX_BUDGET_S 4
constexpr long long ICP_MAILBOX_BUDGET_TICKS = (long long)ICP_MAILBOX_BUDGET_S * 100000000ll;
struct NNFusedTransform {
    const double* R9;  // NULL with a mailbox
    const double* t3;
    const int32_t* idx_prev;
    void* P_out;       // SoA, same padding as the input; must not alias it
    double* err_rows;  // >= blocks_x doubles
    const NNMailbox* mailbox = nullptr;  // armed launch (sparse kernel only): R9/t3 are not read
    NNMailbox* relay = nullptr;          // device-memory copy for the blocks other than block 0 (one per context)
    double want = 0.0;                   // sequence number of the (first) message
    bool resident = false;               // the kernel stays for the whole registration: pass p is message want + p
    bool store_first = false;            // resident: the input is not P_out (pristine copy): store the cloud in pass 0 as well
    void* slot_state = nullptr;          // fused launches of the sparse kernels: 9 x n_pad floats -- points (two planes) and matched model points in slot order (or NULL)
    bool slot_valid = false;             // ... the previous pass wrote them
    bool slot_flip = false;              // ... which of the two planes of points this launch reads (it writes the other): 9 x n_pad floats in all
};
bool nn_can_fuse_transform(const NNPlan& pl);

// inputs of the early-out ("cull") variant of the packed kernel.  Q_scan is a copy of the model whose
// exact duplicates (same x,y,z as a LOWER index) are voided to +inf: such a point can never be the
// lowest-index minimum, and voiding it keeps one coincident cluster (e.g. the hall scan's 4361
// no-return points) from defeating the early-out for every chunk.  seed_idx: any valid model index
// per moving point (the previous pass's match) or NULL.
struct NNCullInputs {
    const void* Q_scan;
    const int32_t* seed_idx;
    const void* boxes;  // per 8-point chunk of Q_scan: {lo.xyz, hi.xyz, 0, 0} floats (launch_model_boxes), or NULL
    const void* samples = nullptr;  // one point per chunk of Q_scan (launch_model_samples): the sparse kernel's cold start
    int waves64 = 0;                // rows of 64 points: 16 = sixteen waves per block where every row has a CU to itself (icp_set_exclusive)
    const NNTuning* tune = nullptr; // the context's switches (NULL: the defaults)
    // sparse kernel only: when the model's own order has no locality its scan copy is kept in Morton order instead
    // (boxes and samples then describe THAT copy) with q_perm[sorted j] = model index; likewise the moving points are
    // dealt to the blocks in Morton order of their initial positions, p_perm[slot] = point.  NULL = identity.
    const void* Q_scan_sorted = nullptr;
    const int32_t* q_perm = nullptr;
    const int32_t* p_perm = nullptr;
    // diagnostic (sparse kernel): NN_WORK_SLOTS device counters of the work the kernel executes -- the launch then uses
    // the instrumented instantiation.  NULL (the default): the production kernel, nothing is counted.
    unsigned long long* work = nullptr;
    // shared rows (NNPlan::share_blocks): 5 x blocks_x device counters, zero before the first launch (three in rotation from
    // launch to launch, two alternating between the first passes of registrations), and the context's counters of both kinds
    // of launches (host; the launcher advances them).  NULL: every row is one block.
    unsigned int* share_counts = nullptr;
    unsigned long long* share_seq = nullptr;
    unsigned long long* share_cold_seq = nullptr;
    float* seed_pub = nullptr;   // resident launches with shared rows: blocks_x x 384 floats (NNFuse::seed_pub)
    // ordered rows (NNPlan::order): block b of the launch works on row row_order[b]; every block adds the hits of its lists to row_hits[row]
    const int32_t* row_order = nullptr;
    unsigned int* row_hits = nullptr;
    bool order_history = false;      // ... the counters behind row_order were filled by launches of an EARLIER registration too
    const float* records = nullptr;   // hierarchical search: one 160-byte record per chunk of the searched view (launch_model_records) or NULL
};
// What the sparse kernel EXECUTED (it returns the brute-force answer without evaluating most pairs): wave-level tallies.
// One "hit" = one 8-point model chunk processed by one wave = 64 lanes x 2 moving points against that chunk.
enum {
    NN_WORK_FIND_BOXES = 0,       // chunk boxes tested against a block's group box (one lane each, ~12 flop)
    NN_WORK_UPPER_BOXES = 1,      // boxes of the upper hierarchy levels tested the same way
    NN_WORK_HITS_BOX = 2,         // hits that went through the per-point box test (128 points x 12 flop)
    NN_WORK_HITS_XY = 3,          // ... that went on to the xy half of the distances (128 x 8 x 5 flop)
    NN_WORK_HITS_FULL = 4,        // ... whose 128 x 8 distances were evaluated in full (128 x 8 x 3 flop more)
    NN_WORK_SAMPLE_GROUPS = 5,    // cold start: groups of 8 samples scanned by a wave (128 x 8 x 8 flop)
    NN_WORK_BLOCK_PASSES = 6,     // (block, pass) pairs, for normalisation
    NN_WORK_BLOCK_TRANSFORMS = 7, // ... of which applied a transform first (16 waves x 128 points x 15 flop, redundantly)
    NN_WORK_SPEC_LISTS = 8,       // resident launches: (block, pass) pairs that entered a pass with a speculative hit list
    NN_WORK_SPEC_COVERED = 9,     // ... whose list covered the real group box and bound (the pass skipped find + fetch)
    NN_WORK_SPEC_HITS = 10,       // hits on the speculative lists that were used
    NN_WORK_LIST_HITS = 11,       // hits on the lists built by the ordinary find
    NN_WORK_SLOTS = 12
};
// Shared rows: how a launch of `blocks` blocks deals itself to `rows` rows, given the hits every row had in the launch
// before (nn_match_sparse computes this in every block, from the same counters: the pieces below are what it is made of,
// and share_rows_plan is the same computation on the host -- icp_share_rows_plan, tests/test_host.py).
//   parts(row) = clamp(ceil(h / T), 1, cap),  cap = min(64-chunk tiles of the model, 32)
//   T0 = ceil(total / spare) always fits (sum of ceil(h / T0) <= total / T0 + rows = blocks); four tighter targets, 4/8 .. 7/8
//   of T0, are tried and the smallest that fits is taken; never below `min_hits`.  The counts are clamped to 2^20 before
//   anything is added up: 512 rows x 2^20 stays within 32 bits, and with it the guarantee that the parts fit the grid.
constexpr unsigned int SHARE_COUNT_CLAMP = 1u << 20;
constexpr unsigned int SHARE_MAX_PARTS = 32u;
ICP_HOST_DEVICE inline unsigned int share_clamp(unsigned int h) { return h > SHARE_COUNT_CLAMP ? SHARE_COUNT_CLAMP : h; }
ICP_HOST_DEVICE inline unsigned int share_cap(int m_pad)
{
    const unsigned int tiles = (unsigned int)((((m_pad >> 3) + 63) >> 6));
    return tiles < SHARE_MAX_PARTS ? (tiles < 1u ? 1u : tiles) : SHARE_MAX_PARTS;
}
ICP_HOST_DEVICE inline unsigned int share_parts(unsigned int h, unsigned int T, unsigned int cap)
{
    unsigned int S = (h + T - 1u) / T;
    S = S > cap ? cap : S;
    return S < 1u ? 1u : S;
}
ICP_HOST_DEVICE inline unsigned int share_first_target(unsigned int total, unsigned int spare) { return spare ? (total + spare - 1u) / spare : 0xffffffffu; }
ICP_HOST_DEVICE inline unsigned int share_candidate(unsigned int T0, int k) { return (T0 * (unsigned int)(4 + k) + 7u) / 8u; }   // k = 0..3
ICP_HOST_DEVICE inline bool share_tries_candidates(unsigned int T0, unsigned int Tmin, unsigned int spare) { return spare != 0u && T0 > Tmin && T0 < 0x10000000u; }
// sums[k] = blocks the launch would need with candidate k
ICP_HOST_DEVICE inline unsigned int share_pick(unsigned int T0, unsigned int Tmin, const unsigned int (&sums)[4], unsigned int blocks)
{
    unsigned int T = T0;
    for (int k = 3; k >= 0; --k) T = sums[k] <= blocks ? share_candidate(T0, k) : T;   // (the smallest candidate that fits: k = 0 last)
    return T < Tmin ? Tmin : T;
}
// the same on the host: parts_out[rows]; returns the target (hits per block)
unsigned int share_rows_plan(const unsigned int* hits, int rows, int blocks, int m_pad, int min_hits, int* parts_out);

// Ordered rows.  A cloud with many more rows than the machine holds blocks runs them in rounds, in index order, and a pass
// lasts until its last block ends: a heavy row that starts late is the tail of the pass (10 M x 10 M, one GPU, pass 12:
// one row lists 468 000 chunks and takes 8.3 ms; it started at 5.9 ms of a 14.2 ms pass whose blocks add up to 10.0 ms per
// CU).  So the rows are taken heaviest first: every block adds the hits of its lists to its row's counter, and before the
// next launch the counters are sorted (stable, descending; 20 bits) into the order its blocks follow.  Any order is exact.
// SPLIT ROWS.  With the rows in that order and everything else faster, a late pass of the same registration lasts exactly as
// long as its heaviest row (pass 24: 8.2 ms for the row that lists 476 000 chunks, 6.0 ms of work per CU in the whole
// launch).  So the launch gets NN_ORDER_EXTRA blocks beyond its rows, and the rows whose counters exceed a target
// T = max(total / (4 x the blocks the machine holds at once), min) are dealt 2, 4, .. 64 of them: block b finds its role
// -- row, part, parts -- in roles[b]
// (launch_row_order writes them: the split rows first, then the others, heaviest first; -1 = no role).  The parts of a row
// interleave the model's super boxes (part p takes those whose number is p modulo parts), fold their minima into the row's
// 64-bit keys and the last to arrive closes the row -- the protocol of the segment blocks.  Any assignment is exact.
constexpr int NN_ORDER_EXTRA = 4096;   // blocks of an ordered launch beyond its rows
// (round 4) the order is by weight CLASSES -- a count's leading one and NN_ORDER_CLASS_BITS bits behind it; the rows of a class keep the
// curve's order -- and 2^NN_ORDER_XCD_SHIFT consecutive positions of it go to blocks of one XCD (blocks b and b + 8 share one): neighbours
// on the curve list mostly the same chunks, and run side by side behind ONE L2.  10 M x 10 M: 10.6 -> 3.1 GB fetched per pass (raw
// FETCH_SIZE), 4.88 -> 4.77 ms per iteration; classes of 0 / 1 / 2 bits, groups of 8 / 16 / 32: the same within 0.5 %
// (profiles/r4/r4_12_s5_order_classes_xcd_groups.txt)
constexpr int NN_ORDER_CLASS_BITS = 1;
constexpr int NN_ORDER_XCD_SHIFT = 4;
constexpr int NN_ORDER_HEAD = 1024;    // rows (the heaviest) that may be split
constexpr int NN_ROLE_ROW_BITS = 21, NN_ROLE_PART_BITS = 6;   // role = row | part << 21 | log2(parts) << 27
// the hierarchical search fetches a hit from one 160-byte record per chunk (icp_kernels.hip, model_records_kernel)
constexpr int NN_REC_WORDS = 40;
size_t model_records_bytes(int m_pad);
hipError_t launch_model_records(const void* Qs_soa, const float* boxes, const int32_t* perm, int m_pad, float* rec, hipStream_t st);

struct RowOrderBuffers {
    unsigned int* keys[2];   // >= rows each
    int32_t* vals[2];        // >= rows each; the order ends up in vals[1]
    void* temp;              // rocPRIM scratch, row_order_temp_bytes(rows)
    size_t temp_bytes;
    int32_t* roles;          // >= rows + NN_ORDER_EXTRA
    unsigned long long* totals;   // 2, zero before the first launch (the sum of the counters, alternating)
    unsigned long long seq;  // launches so far (the caller counts)
    int min_part;            // no part is meant to be smaller than this many hits (0: no row is split)
    int total_div;           // the target: the sum of the counters over this (4 x the blocks the machine holds at once)
    int control = 1;         // up to 16 384 rows: ONE single-workgroup launch (LDS counting sort + roles) instead of keys + rocPRIM sort + roles
    int coarse = NN_ORDER_CLASS_BITS;   // the order's weight classes: bits kept behind a count's leading one (0..3; order_class) -- rows of one class keep the curve's order
};
size_t row_order_temp_bytes(int rows);
// reads AND zeroes hits[rows] (the next launch counts afresh); *roles_out = the roles of rows + NN_ORDER_EXTRA blocks (device pointer)
hipError_t launch_row_order(const RowOrderBuffers& b, unsigned int* hits, int rows, const int32_t** roles_out, hipStream_t st);

// device-side preparation of the sparse kernel's views (icp_set_model / icp_set_moving): scratch owned by the caller
struct PrepBuffers {
    unsigned int* keys[2];   // >= count each
    int32_t* vals[2];        // >= count each
    void* temp;              // rocPRIM radix-sort scratch, prep_sort_temp_bytes(count)
    size_t temp_bytes;
    float* box;              // 4 floats
    double* ext;             // >= ceil(count / smallest group) doubles
};
size_t prep_sort_temp_bytes(int count);
hipError_t launch_duplicates_and_scan_copy(const PrepBuffers& b, const float* X_soa, int n, int n_pad, unsigned char* voided, int* count_dev,
                                           float* scan_out_soa, hipStream_t st);
hipError_t launch_duplicates_and_scan_copy_f64(const PrepBuffers& b, const double* X_soa, int n, int n_pad, unsigned char* voided, int* count_dev,
                                               double* scan_out_soa, hipStream_t st);
// totals_dev: 4 doubles -- summed extents of the groups in the given / the Morton order, for `group` and (if > 0) `group2`
hipError_t launch_morton_order(const PrepBuffers& b, const float* X_soa, int n, int n_pad, int group, int group2, int32_t* perm_out,
                               double* totals_dev, hipStream_t st);
hipError_t launch_gather_sorted(const float* Qs_soa, int m, int m_pad, const int32_t* perm, float* out_soa, int32_t* perm_pad, hipStream_t st);
hipError_t launch_slot_map(const int32_t* perm, int n, int n_pad, int32_t* out, hipStream_t st);
// round 4, the set-up of a cloud in a handful of launches (icp_k_setup.hip): exact duplicates by hashing (table: a power of two of
// >= 2 n 32-bit words, never cleared; gen: the upload's number), the scan copy in the same pass; group extents summed in fixed
// point relative to the bounding cube the layout kernel left in enc[6] (out[2 * which] given order, [2 * which + 1] `order`);
// the curve order of a small cloud; chunk boxes + samples of a flat model in one launch
hipError_t launch_duplicates_hashed(const float* X_soa, int n, int n_pad, unsigned int* table, unsigned int table_entries, unsigned int gen, unsigned char* voided,
                                    int* count_dev, float* scan_out_soa, hipStream_t st);
// (ticket / voided_count / report / seq: the launch's last block leaves the four sums, the duplicate count and `seq` in pinned host
// memory -- the host spins on report->seq; all NULL: nothing is reported)
struct PrepReport { unsigned long long fixed[4]; int voided; unsigned int seq; };
hipError_t launch_extents_fixed(const float* X_soa, int n, int n_pad, const int32_t* order, int group, const unsigned int* enc, unsigned long long* out, int which, hipStream_t st,
                                unsigned int* ticket = nullptr, const int* voided_count = nullptr, PrepReport* report = nullptr, unsigned int seq = 0);
hipError_t launch_curve_order_small(const PrepBuffers& b, const float* X_soa, int n, int n_pad, const unsigned int* enc, int32_t* perm_out, hipStream_t st);
hipError_t launch_model_boxes_samples(const void* Qs_soa, int m_pad, float* boxes, float* samples, hipStream_t st);
size_t model_samples_bytes(int m_pad);
// fp64 form of the sparse search: chunk boxes {lo.xyz, hi.xyz, -, -} and one sample per chunk, in double, of the model itself
size_t model_boxes_f64_bytes(int m_pad);
size_t model_samples_f64_bytes(int m_pad);
hipError_t launch_model_tables_f64(const void* Q_soa, int m_pad, void* boxes, void* samples, hipStream_t st);
hipError_t launch_model_samples(const void* Qs_soa, int m_pad, float* samples, hipStream_t st);
// boxes: model_boxes_bytes(m_pad) -- the chunk boxes, followed by the boxes of every 64 of them and of every 64 of
// those (the upper search levels of a large model)
size_t model_boxes_bytes(int m_pad);
hipError_t launch_model_boxes(const void* Qs_soa, int m_pad, float* boxes, hipStream_t st);

// fused tail of the packed kernel: atomic (d, idx) keys + row tickets, the row's last block produces idx and the
// moment row (see NNTail in icp_kernels.hip).  keys must be all-ones and tickets zero before the first launch;
// the kernel leaves them that way.
struct NNTailArgs {
    int metric;                 // ICP_POINT_TO_POINT / ICP_POINT_TO_PLANE
    unsigned long long* keys;   // [n_pad]
    unsigned int* tickets;      // [blocks_x]
    double* err_tile;           // [blocks_x] device
    int32_t* idx_out;           // [n_pad]
    int32_t* idx_out_odd = nullptr;  // resident launch: odd passes write here (NULL: idx_out)
    const void* Nrm_soa;        // normals (plane)
    double* rows;               // [blocks_x][ICP_NMOM], pinned host or device
    double tag;
    // sparse kernels, point-to-point, rows read by the host: rows of NN_CROW doubles {error share + tag, sum p, sum q,
    // sum q p^T} -- see NNTail in icp_kernels.hip
    int compact = 0;
    int rows_on_device = 0;     // the rows stay in device memory for a later kernel (finalize): no drain, no tag to wait for
    // rows_on_device, nn_match_sparse: the rows are added up inside the launch (NNTail::fin_*, icp_device.h) into fin_out -- pinned
    // host memory (fin_host: the pass's tag lands in its last slot) or a device vector; NULL: a finalize launch follows
    unsigned int* fin_tickets = nullptr;   // [NN_FIN_GROUPS + 1], zero before the first launch
    double* fin_scratch = nullptr;         // [NN_FIN_GROUPS][ICP_NMOM]
    double* fin_out = nullptr;             // [ICP_NMOM]
    int fin_host = 0;
};
constexpr int NN_FIN_GROUPS = 256;     // ranges of rows of an in-launch finalize, at most
constexpr int NN_CROW = 16;            // doubles per compact row
constexpr int NN_CROW_TAG_BITS = 16;   // low mantissa bits of slot 0 that carry the row's tag (mod 2^16)
bool nn_can_fuse_tail(const NNPlan& pl);

// matching: per (segment, point) partial minimum + index.  `ft` (optional) = fused transform,
// `opt` (optional) = early-out inputs, `ta` (optional) = fused tail (then no partials are written).
hipError_t launch_nn(const NNPlan& pl, const void* P_soa, const void* Q_soa, void* part_d, int32_t* part_idx,
                     const NNFusedTransform* ft, const NNCullInputs* opt, const NNTailArgs* ta, hipStream_t st);
// stand-alone merge of the segment partials into idx (icp_nn_match_* only; the ICP loop merges
// inside the moments kernel)
hipError_t launch_merge(const NNPlan& pl, const void* part_d, const int32_t* part_idx, int32_t* idx, hipStream_t st);

constexpr int MOM_MAX_BLOCKS = 1024;
// fused: merge segment partials -> idx, gather q[idx], accumulate the metric's moments in fp64.
// partials: [blocks][ICP_NMOM] doubles; returns the number of blocks used in *blocks.
hipError_t launch_moments(const NNPlan& pl, int metric, const void* P_soa, const void* Q_soa, const void* N_soa,
                          const void* part_d, const int32_t* part_idx, int32_t* idx, double* partials, int* blocks,
                          double tag /* stored in slot ICP_NMOM-1 of every row once the row is complete */,
                          const double* err_rows /* device */, int err_count /* folded into slot ICP_MOM_ERR */,
                          hipStream_t st);
// p <- R p + t in place (storage precision, separately rounded mul/add), and
// sum |p_new - q[idx]|^2 in fp64 -> err_partials[block]
hipError_t launch_transform_error(int precision, void* P_soa, int n, int n_pad, const double* R9, const double* t3,
                                  const void* Q_soa, int m_pad, const int32_t* idx, double* err_partials,
                                  int* blocks, hipStream_t st);
// deterministic fixed-order reduction of the per-block partials into the ICP_NMOM-vector
hipError_t launch_finalize(double* mom_out, const double* mom_partials, int mom_blocks, const double* err_partials,
                           int err_blocks, int rows_have_err /* slot 0 of the rows carries error shares */, hipStream_t st, double* scratch = nullptr /* >= 256 x ICP_NMOM doubles: many rows are added in two stages */);

// soa2 (optional): a second copy of the result (the pristine moving cloud); enc (optional, fp32): the cloud's bounding cube as six
// ordered-integer words {~ord(min xyz), ord(max xyz)}, zero before the launch
hipError_t launch_aos_to_soa(int precision, const void* aos, int n, int n_pad, void* soa, hipStream_t st, unsigned int* nonfinite = nullptr, void* soa2 = nullptr,
                             unsigned int* enc = nullptr);
hipError_t launch_soa_to_aos(int precision, const void* soa, int n, int n_pad, void* aos, hipStream_t st);

// model-on-model 4 nearest neighbours (self / rank 0 dropped)
hipError_t launch_knn4(const NNPlan& pl, const void* Q_soa, int32_t* nbr /*[m][4]*/, hipStream_t st);
// fp32: packed, wave-/segment-split top-5 kernel + merge.  part_* hold splits * n_pad * 5 entries.
void knn4_v2_geometry(int m, int num_cus, int* n_pad, int* blocks_x, int* splits, int* seg_len);
hipError_t launch_knn4_v2(const void* Q_soa, int m, int num_cus, float* part_d, int32_t* part_j, int32_t* nbr,
                          hipStream_t st);
// covariance of the 4 neighbours + fp64 Jacobi eigen-solve per lane -> padded SoA normals on the device
hipError_t launch_normals(int precision, const void* Q_soa, int m, int m_pad, const int32_t* nbr, void* Nrm_soa,
                          hipStream_t st);
hipError_t launch_os1_packets(const uint8_t* packets, int n_packets, const float* alt16, const float* az16,
                              uint32_t* ranges, float* xyz_aos, hipStream_t st);
hipError_t launch_os1_conversion(const uint32_t* ranges, int n, uint32_t encoder0, const float* alt16,
                                 const float* az16, float* xyz_aos, hipStream_t st);

}  // namespace icp
