// icp_api.cpp -- the C ABI of libicp_mi355x.so (include/icp_mi355x.h): context, HBM residency,
// and the ICP driver loops that replace the reference's main() while-loops
//   src/ICP_CPU.c:217-271, src/ICP_point_to_point.cu:295-423, src/ICP_point_to_plane.cu:517-631.
//
// Loop shape (one host round trip per iteration, no H2D traffic at all):
//
//   enqueue k:  [transform_error(R_{k-1}, t_{k-1})]  ->  nn_match  ->  moments  ->  finalize
//               (R, t travel as kernel arguments)        P_k vs Q      fused       32 doubles
//   <optional all-reduce of the 32-double vector across ranks, in place, on the same stream>
//   complete k: D2H 256 B, E[k] and the stop rule on the host, 3x3 SVD / 6x6 Cholesky -> R_k, t_k
//
// The error of transform k-1 rides in slot 0 of the vector produced by enqueue k, so matching pass k
// is issued speculatively before the stop rule for E[k] is known; when the rule fires that one
// pass is discarded (it never touched P).  Correspondences ping-pong between two buffers so the
// indices of the last CONTRIBUTING pass survive the speculative one.
#include <hip/hip_runtime.h>

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <unistd.h>
#include <sched.h>
#include <limits>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/icp_mi355x.h"
#include "../../include/icp_mi355x_diag.h"
#include "icp_comm.h"
#include "icp_lcomm.h"
#include "icp_host_loop.h"
#include "icp_host_math.h"
#include "icp_kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail(ICP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

// orders the stores of a mailbox message before its sequence number (and pushes them out, should the mailbox ever
// live in write-combining memory: the `lock or` compilers emit for a seq_cst fence does not do that)
static inline void bar_fence()
{
#if defined(__x86_64__) || defined(__i386__)
    _mm_sfence();
#else
    __sync_synchronize();
#endif
}

// One message = one 64-byte line (layout: icp_kernels.h, NNMailbox): each 32-byte half is written by ONE vector store
// and carries the tag in its last word, then one fence pushes the line out.  rt may be NULL (commands that carry no
// transform); seq = 0 clears the mailbox (no tag ever equals 0).
#if defined(__x86_64__)
// The compact rows of a pass added up, rows in block order: four 4-double accumulators take a row's sixteen slots at once.
// Every slot is still the sum of its 256 values in block order, starting from zero -- the bits of the scalar loop -- but the
// sixteen chains advance together instead of one after the other (hall: 256 rows, once per pass, on the path between the last
// row's arrival and the next message).  The first slot of every 32-byte sector (0, 4, 8, 12) carries the row's tag in its low
// mantissa bits (round 4: one store per row, no drain -- tail_reduce_store): masked off as it is loaded.
__attribute__((target("avx"))) static void add_compact_rows_avx(const double* rows, int count, unsigned long long tag_mask, double (&out)[16])
{
    const __m256d keep = _mm256_castsi256_pd(_mm256_set_epi64x(-1ll, -1ll, -1ll, (long long)~tag_mask));
    __m256d a0 = _mm256_setzero_pd(), a1 = a0, a2 = a0, a3 = a0;
    for (int b = 0; b < count; ++b) {
        const double* r = rows + (size_t)b * 16;
        a0 = _mm256_add_pd(a0, _mm256_and_pd(_mm256_loadu_pd(r), keep));
        a1 = _mm256_add_pd(a1, _mm256_and_pd(_mm256_loadu_pd(r + 4), keep));
        a2 = _mm256_add_pd(a2, _mm256_and_pd(_mm256_loadu_pd(r + 8), keep));
        a3 = _mm256_add_pd(a3, _mm256_and_pd(_mm256_loadu_pd(r + 12), keep));
    }
    _mm256_storeu_pd(out, a0); _mm256_storeu_pd(out + 4, a1); _mm256_storeu_pd(out + 8, a2); _mm256_storeu_pd(out + 12, a3);
}

// the same for rows in the full format (ICP_NMOM = 32 doubles, the last one the row's tag: not a moment -- left out)
__attribute__((target("avx"))) static void add_full_rows_avx(const double* rows, int count, double (&out)[32])
{
    const __m256d keep = _mm256_castsi256_pd(_mm256_set_epi64x(0ll, -1ll, -1ll, -1ll));
    __m256d a[8];
    for (int v = 0; v < 8; ++v) a[v] = _mm256_setzero_pd();
    for (int b = 0; b < count; ++b) {
        const double* r = rows + (size_t)b * 32;
        for (int v = 0; v < 7; ++v) a[v] = _mm256_add_pd(a[v], _mm256_loadu_pd(r + 4 * v));
        a[7] = _mm256_add_pd(a[7], _mm256_and_pd(_mm256_loadu_pd(r + 28), keep));
    }
    for (int v = 0; v < 8; ++v) _mm256_storeu_pd(out + 4 * v, a[v]);
}

__attribute__((target("avx"))) static void store_line_avx(uint32_t* dst, const uint32_t* line)
{
    _mm256_store_si256(reinterpret_cast<__m256i*>(dst), _mm256_load_si256(reinterpret_cast<const __m256i*>(line)));
    _mm256_store_si256(reinterpret_cast<__m256i*>(dst + 8), _mm256_load_si256(reinterpret_cast<const __m256i*>(line + 8)));
}
#endif
static inline void post_message(icp::NNMailbox* mb, const double* R9, const double* t3, int cmd, double seq, bool wide_stores = true)
{
    alignas(32) uint32_t line[16];
    std::memset(line, 0, sizeof line);
    const uint32_t tag = seq == 0.0 ? 0u : icp::mailbox_tag(seq);
    if (R9 && t3) {
        for (int k = 0; k < 12; ++k) {
            const float f = (float)(k < 9 ? R9[k] : t3[k - 9]);
            std::memcpy(&line[icp::mailbox_rt_word(k)], &f, sizeof f);
        }
    }
    line[icp::ICP_MB_CMD] = (uint32_t)cmd;
    line[icp::ICP_MB_TAG0] = tag;
    line[icp::ICP_MB_TAG1] = tag;
#if defined(__x86_64__)
    static const bool have_avx = __builtin_cpu_supports("avx");
    if (have_avx && wide_stores) {
        store_line_avx(mb->w, line);
        bar_fence();
        return;
    }
#endif
    // no 32-byte stores: the payload first, then (fenced) the two tags -- the reader still accepts only a line whose
    // tags both match, so the order of the words within a half does not matter
    volatile uint32_t* dst = mb->w;
    for (int k = 0; k < 16; ++k)
        if (k != icp::ICP_MB_TAG0 && k != icp::ICP_MB_TAG1) dst[k] = line[k];
    bar_fence();
    dst[icp::ICP_MB_TAG0] = tag;
    dst[icp::ICP_MB_TAG1] = tag;
    bar_fence();
}

// the message of a registration in double (NNMailbox64): four 32-byte parts {3 doubles, cmd, tag}, one vector store each
static inline void post_message64(icp::NNMailbox* mb32, const double* R9, const double* t3, int cmd, double seq, bool wide_stores = true)
{
    alignas(32) uint32_t line[32];
    std::memset(line, 0, sizeof line);
    const uint32_t tag = seq == 0.0 ? 0u : icp::mailbox_tag(seq);
    for (int h = 0; h < 4; ++h) {
        if (R9 && t3)
            for (int k = 0; k < 3; ++k) {
                const int i = 3 * h + k;
                const double v = i < 9 ? R9[i] : t3[i - 9];
                std::memcpy(&line[h * 8 + 2 * k], &v, sizeof v);
            }
        line[h * 8 + icp::ICP_MB64_CMD] = (uint32_t)cmd;
        line[h * 8 + 7] = tag;
    }
    uint32_t* dstw = reinterpret_cast<uint32_t*>(mb32);
#if defined(__x86_64__)
    static const bool have_avx = __builtin_cpu_supports("avx");
    if (have_avx && wide_stores) {
        store_line_avx(dstw, line);
        store_line_avx(dstw + 16, line + 16);
        bar_fence();
        return;
    }
#endif
    volatile uint32_t* dst = dstw;
    for (int k = 0; k < 32; ++k)
        if ((k & 7) != 7) dst[k] = line[k];
    bar_fence();
    for (int h = 0; h < 4; ++h) dst[h * 8 + 7] = tag;
    bar_fence();
}

static constexpr int kMailSlots = 4;  // armed launches: ring of mailboxes (one is live at a time)
static constexpr size_t kMailSlotBytes = sizeof(icp::NNMailbox64);   // a slot holds a float message (one line) or a double one (two)
static inline icp::NNMailbox* mail_slot(icp::NNMailbox* base, int slot) { return reinterpret_cast<icp::NNMailbox*>(reinterpret_cast<char*>(base) + (size_t)slot * kMailSlotBytes); }
// Time budgets of a kernel that waits for the host, ordered so that a late host and a waiting kernel can never disagree:
//   * a waiting block gives up (and reads that as EXIT) only after ICP_MAILBOX_BUDGET_S of WALL-CLOCK time (icp_kernels.h);
//   * the host posts a message only while at most kMailLeaseS have passed since it last knew the kernel to be waiting (the
//     rows of the previous pass complete / the launch); when it is later than that -- descheduled, or held up in the
//     inter-rank exchange, whose own limit is longer -- it sends EXIT instead (an EXIT is consistent at any time: a block
//     that has already given up did exactly that) and relaunches.  kMailLeaseS < budget / 2;
//   * the host's wait for a pass's rows (kRowPollS) is shorter than the budget too: when it gives up it withdraws the
//     kernel and lets the runtime report what happened.
static constexpr double kMailLeaseS = 1.5;
static constexpr double kRowPollS = 2.0;
static_assert(kMailLeaseS * 2.0 < (double)ICP_MAILBOX_BUDGET_S && kRowPollS < (double)ICP_MAILBOX_BUDGET_S, "host budgets must stay inside the kernel's");
static constexpr size_t kPhaseSlots = 512 * 1024;  // ICP_NN_PHASES: 10 stamps per wave

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes < 256 ? 256 : bytes;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct LoopState {
    bool active = false;
    bool pending = false;   // an enqueue awaits its complete
    icp::HostLoop H;        // error series, stop rule, minimisation, transform composition (host only)
    int applied_idx = 0;    // idx buffer used by the last applied transform
    int mom_blocks = 0, err_blocks = 0;
    double seconds_nn = 0.0;
    double seconds_host = 0.0;  // host half of the passes (error, stop rule, solve), summed while profiling is on
    int nn_launches = 0;
    bool timed_nn = false;
    bool numeric_failure = false;  // the minimisation refused the last pass's moments: the loop is over, its state stays readable
    bool host_reduce = false;  // how the pending enqueue's partial rows are being reduced
    bool final_poll = false;   // ... inside the launch itself, which leaves the vector and the pass's tag in c->h_final (the host polls ONE tag)
    bool matched = false;      // a matching pass of THIS loop has filled idx[cur]
    bool rows_have_err = false; // slot 0 of the pending moment rows carries the error shares (fused tail)
    bool rows_compact = false;  // the pending rows are compact (NN_CROW doubles; slot 0 = error share with the tag in its low mantissa bits)
    double wait_tag = 0.0;      // completion tag of the pending enqueue's rows
    // armed launch: the matching pass AFTER the pending one is already enqueued and waits for its (R, t)
    bool armed = false;
    bool slot_written = false;   // the pending (or last completed) pass was an armed launch that left points and matches in slot order
    bool slot_flip = false;      // ... in this plane of the slot-order points (the next such launch reads it and writes the other)
    double armed_tag = 0.0;
    int armed_slot = 0;
    int armed_prev_cur = 0;
    bool armed_compact = false;
    std::chrono::steady_clock::time_point armed_at{};   // when the armed pass was launched (mailbox lease)
    icp::NNMailbox* live_mailbox = nullptr;             // a resident kernel is running and listens here
    bool from_pristine = false;  // the loop started from the cloud icp_set_moving uploaded: it can be run again from the copy
    long long steps = 0;         // completed (enqueue + complete) steps of this loop
};

// the calling thread's affinity, narrowed to the device's NUMA node for the duration of one entry point (see icp_create)
struct ScopedPin {
    bool restore = false;
    cpu_set_t saved;
    explicit ScopedPin(const icp_ctx* c);
    ~ScopedPin() { if (restore) (void)sched_setaffinity(0, sizeof saved, &saved); }
    ScopedPin(const ScopedPin&) = delete;
    ScopedPin& operator=(const ScopedPin&) = delete;
};

}  // namespace

struct icp_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool profiling = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    int prec = -1;  // precision of the resident clouds (model and moving must agree)
    int n = 0, m = 0;
    bool have_model = false, have_moving = false, have_normals = false;
    DevBuf P0;  // pristine copy of the moving cloud as uploaded (icp_reset_moving)
    DevBuf Qbox;  // chunk bounding boxes of Qs
    DevBuf Qrec;  // large models (hierarchical search): one 160-byte record per chunk -- box, coordinates, indices (launch_model_records)
    bool have_records = false;
    DevBuf Qsamp; // one point per chunk of Qs
    DevBuf Qss;   // Morton-ordered scan copy (sparse kernel), when the model's own order has no locality
    DevBuf Qperm; // ... and its permutation: sorted position -> model index
    DevBuf Pperm; // slot -> moving point (Morton order of the initial positions), when the cloud's own order has no locality
    DevBuf slot_state; // fused launches of the sparse kernels: moving points (two planes) + matched model points in slot order (9 x n_pad floats)
    DevBuf share_counts;                        // shared rows (NNPlan::share_blocks): 5 x blocks_x hit counters (3 in rotation from launch to launch, 2 for first passes)
    mutable unsigned long long share_seq = 0;   // ... the launches so far (advanced by the launcher)
    mutable unsigned long long share_cold_seq = 0;   // ... and those that were the first pass of a registration
    DevBuf order_roles, order_totals;  // ... the roles of the launch's blocks (split rows: icp_kernels.h, NN_ORDER_*), the sum of the counters
    unsigned long long order_seq = 0;
    int order_regs = 0;                // registrations (loops) that have run ordered launches with these counters
    int order_launches = 0;            // ... ordered launches of the loop that is running
    int split_min = -1;                // the smallest part of a split row, in hits of the launch before (ICP_NN_SPLIT_MIN; 0: no row is split; -1: 512 per wave of a block)
    DevBuf row_hits, order_keys[2], order_vals[2], order_tmp;   // ordered rows (NNPlan::order): hits per row, and the sort that turns them into the next launch's order
    const int32_t* row_order = nullptr;          // ... the order the next launch follows (device; NULL: index order)
    DevBuf seed_pub;                            // ... resident launches: blocks_x x 384 floats, the matches of split rows for their other blocks
    bool exclusive = false;                     // icp_set_exclusive: the caller owns the device -- rows of 64 points run as 16-wave blocks, one to a CU
    bool share_auto = true;                     // ... ICP_SHARE_AUTO=0: never resident of its own accord (see share_wants_resident)
    int share_resident_after = -1;              // ... ICP_SHARE_RESIDENT_AFTER=n: a registration runs armed launches for n passes, then one resident kernel (< 0, the default: armed throughout)
    bool model_sorted = false, moving_sorted = false;
    // scratch of the device-side preparation (duplicate flags, Morton order, extent test)
    DevBuf prep_keys[2], prep_vals[2], prep_tmp, prep_small, prep_ext, prep_voided, prep_perm;
    struct PrepSmall { float box[4]; double totals[4]; int voided; int pad_; unsigned int enc[6]; unsigned int ticket; unsigned int pad2_; unsigned long long fixed[4]; };
    icp::PrepReport* h_prep = nullptr;   // pinned, coherent: where the short set-up's last launch leaves its sums (the host spins on its seq word)
    unsigned int prep_seq = 0;
    void* h_stage = nullptr;             // pinned, mapped: small clouds are laid out straight from here (no separate copy command)
    size_t h_stage_cap = 0;
    // round 4, the short set-up of clouds of up to kPrepSmallMax points: exact duplicates by hashing (a table that is never cleared:
    // entries carry the upload's generation), and the spatial-order decision remembered per (cloud kind, size, group) -- a sensor's next
    // scan has the order of the one before: while the given order's summed group extent stays within a quarter of the remembered one
    // and the remembered decision was "own order", the curve sort that only served to confirm it is not run (ICP_SORT overrides)
    static constexpr int kPrepSmallMax = 65536;
    DevBuf dup_table;
    unsigned int dup_gen = 0;
    struct OrderMemo { bool valid = false; int count = 0, group = 0; bool sorted = false; double given_rel = 0.0; };
    OrderMemo memo_model, memo_moving;
    DevBuf fin_scratch;      // finalize in two stages (many rows): 256 x ICP_NMOM doubles
    DevBuf fin_tickets;      // rows added up inside the matching launch (NNTail::fin_*): NN_FIN_GROUPS + 1 tickets, zero between launches
    double* h_final = nullptr;   // ... and where the launch leaves its ICP_NMOM vector for the host: pinned, coherent; the pass's tag in the last slot
    DevBuf work;             // icp_set_work_counting: NN_WORK_SLOTS counters of the work the sparse kernel executes
    bool count_work = false;
    DevBuf phase_log;        // ICP_NN_PHASES diagnostic
    size_t phase_slots = 0;
    std::string phase_path;
    DevBuf P, P2, Q, Qs, Nrm, stage;  // Qs: duplicate-voided scan copy of the model (fp32 early-out kernel)
    bool have_scan_copy = false;
    int voided = 0;  // P2: ping-pong target of the transform fused into the matching kernel
    DevBuf part_d, part_idx, idx[2];
    int cur = 0;  // idx buffer written by the most recent matching pass
    bool idx_valid = false;  // idx[cur] holds matches of the resident clouds
    DevBuf mom_partials, err_partials, mom_own, nbr;
    DevBuf keys, tickets;              // fused tail of the matching kernel: (d, idx) keys per moving point, row tickets
    size_t rows_cap = 0;               // rows available in mom_partials / h_mom_partials
    int rows_format = -1;              // format of the rows last written to h_mom_partials: 1 compact, 0 full, -1 none yet
    bool fused_tail = true;            // ICP_FUSED_TAIL=0 keeps matching and moments as two kernels
    bool use_boxes = true;             // (false with ICP_NN_SPARSE=0: the dense kernels, no boxes)
    bool mail_wide = true;             // ICP_MAILBOX=plain: write the mailbox line word by word (payload, fence, tags) -- the path of a CPU without AVX
    icp::NNTuning tune{};              // every switch the plan and the launchers look at, read once in icp_create
    double* mom_dev = nullptr;
    double* h_mom = nullptr;  // pinned: the reduced ICP_NMOM vector as the host solve reads it
    unsigned int* h_nonfinite = nullptr;  // pinned, coherent: points with a NaN / infinite coordinate seen by the last upload
    // single-GPU fast path: the moments / transform kernels store their per-block partial rows straight
    // into mapped pinned host memory and the host adds them in block order -- no finalize launch, no
    // D2H blit.  (With an external moments buffer, i.e. the multi-GPU driver, the device finalize runs.)
    double* h_mom_partials = nullptr;  // [MOM_MAX_BLOCKS][ICP_NMOM]
    double* h_err_partials = nullptr;  // [err_cap]
    size_t err_cap = 0;                // rows available in err_partials / h_err_partials
    uint64_t tag_seq = 0;              // completion tag of the most recent moments launch (exact in a double)
    int profile_stride = 0;            // time every n-th matching launch (0 = never)
    uint64_t nn_launch_count = 0;
    double prof_seconds_nn = 0.0;      // cumulative over loops since icp_set_profiling
    int prof_nn_launches = 0;
    long long prof_nn_passes = 0;      // matching passes inside those launches (resident kernels run many)
    uint64_t resident_launch_count = 0;
    // ICP_TRACE=1: host-side time split of the loop, printed by icp_destroy
    bool trace = false;
    double tr_first_row = 0.0, tr_last_row = 0.0;
    std::chrono::steady_clock::time_point tr_rows_done{};
    bool trace_passes = false;         // ICP_TRACE=2: one line per pass of a resident registration
    double tr_enqueue = 0, tr_wait = 0, tr_reduce = 0, tr_solve = 0;
    uint64_t tr_n = 0;
    void* comm = nullptr;              // RCCL communicator (icp_comm_init): the loop all-reduces its vector itself
    icp::LocalComm* lcomm = nullptr;   // host-memory communicator (icp_comm_init_local): the vector is summed over the node's ranks on the host
    // test hook (ICP_DEBUG="stall=pass:seconds", read by icp_create): the host sleeps once, right before it would publish
    // the message of that pass of a registration -- a descheduled host thread, as the mailbox lease has to survive it
    int debug_stall_pass = -1;
    double debug_stall_s = 0.0;
    int debug_lose_pass = -1;          // test hook (ICP_DEBUG=lose=pass): the message of that pass is never posted, once
    bool debug_shared_resident = false; // test hook (ICP_DEBUG=shared_resident): ranks that share a device may keep resident kernels
    int moving_group = 0;              // group size the moving cloud's order was judged on (0: not judged)
    std::chrono::steady_clock::time_point posted_at{};   // resident loop: when the pending pass's message went out (the row poll's time-out counts from here)
    bool moving_untouched = false;     // c->P (or the pristine copy standing in for it) still holds what icp_set_moving uploaded
    bool rows_timed_out = false;       // the last failure of icp_loop_complete was a pass that never delivered its rows
    int recoveries = 0;                // registrations finished step-wise after such a time-out (icp_recoveries)
    int pin_mode = 1;                  // ICP_PIN: 0 never, 1 scoped (default), 2 narrowed once and kept
    bool have_local_cpus = false;
    cpu_set_t local_cpus;              // CPUs of the device's NUMA node (sysfs local_cpulist)
    std::chrono::steady_clock::time_point rows_done_at{};   // when the host last saw a pass's rows complete (mailbox lease)
    bool poll = true;                  // (false: the host waits for a pass with a stream synchronisation instead of polling the row tags; no switch any more)
    bool arm = true;                   // ICP_ARMED=0: icp_loop_run never enqueues a pass ahead of its (R, t)
    bool shares_device = false;        // a rank of the attached node communicator runs on the same device: nothing is armed ahead (see icp_comm_init_local)
    int resident = 1;                  // ICP_RESIDENT=0: icp_loop_run never keeps one kernel for a whole registration; 2: also where shared rows are preferred
    bool resident_refused = false;     // the resident kernel does not fit the machine with this plan: do not try again
    // ring of mailboxes for armed / resident launches, in pinned mapped host memory, and the device-memory relay.
    // (Fine-grained device memory written through the PCIe BAR is ~0.5 us faster per message and needs no relay --
    // tools/mailbox_probe.hip -- but with the HIP runtime that PyTorch bundles the waiting kernel never sees a
    // store made after it started; host memory polled by ONE block works with every runtime.)
    icp::NNMailbox* h_mail = nullptr;
    bool mail_in_bar = false;
    bool moving_is_pristine = false;   // icp_reset_moving: P is stale, the cloud to use is P0 (copied on first need)
    // fine-grained device memory: ordinary (coarse-grained) device memory is cached per XCD L2, and a block polling
    // it from another XCD keeps reading its stale line (seen as 24 of 128 blocks never receiving the message)
    icp::NNMailbox* relay = nullptr;
    uint64_t mail_seq = 0;
    // Who adds up the moment rows: the host, as their tags arrive in pinned memory (no synchronisation, and what armed and
    // resident launches need) -- or, for clouds of more than kHostRowsMax rows, the device (two-stage finalize, 256 bytes come
    // back): 78 125 rows of the 10 M-point cloud are 20 MB over PCIe and a pass through them on one core per iteration,
    // 0.7 ms of 13 (profiles/r3: the library-issued RCCL route, which reduces on the device, was FASTER than the default).
    // Round 4: from 1 025 rows up (beyond what the host's sweep takes) the sparse kernels add their rows up INSIDE the launch (two
    // levels of tickets, NNTail::fin_*) and leave the vector with the pass's tag in pinned memory: no finalize launches, no copy, no
    // synchronisation, and such a pass can be armed ahead like any other.
    static constexpr int kHostRowsMax = 1024;
    bool host_reduce() const { return !comm && mom_dev == (double*)mom_own.p && h_mom_partials != nullptr && (plan.blocks_x <= host_rows_max || plan.n == 0); }
    int host_rows_max = kHostRowsMax;   // (ICP_HOST_ROWS_MAX: A/B runs)
    icp::NNPlan plan{};
    LoopState loop;
};

extern "C++" __attribute__((visibility("hidden"))) int decide_moving_order(icp_ctx* c, const void* P_soa, int grp, bool have_enc = false);   // (below: it needs the prep helpers)

namespace {

ScopedPin::ScopedPin(const icp_ctx* c)
{
    if (!c || c->pin_mode != 1 || !c->have_local_cpus) return;
    const int cpu = sched_getcpu();
    if (cpu >= 0 && cpu < CPU_SETSIZE && CPU_ISSET(cpu, &c->local_cpus)) return;   // already next to the device: nothing to do
    cpu_set_t both;
    if (sched_getaffinity(0, sizeof saved, &saved) != 0) return;
    CPU_AND(&both, &saved, &c->local_cpus);
    if (CPU_COUNT(&both) == 0) return;                                              // the caller may not run there at all
    if (sched_setaffinity(0, sizeof both, &both) == 0) restore = true;
}

int use(icp_ctx* c)
{
    if (!c) return fail(ICP_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return ICP_OK;
}

int ensure_work_buffers(icp_ctx* c)
{
    const icp::NNPlan before = c->plan;
    c->plan = icp::nn_plan(c->n, c->m, c->prec, c->num_cus, c->tune);
    const icp::NNPlan& pl = c->plan;
    // the moving cloud's order was judged when it was uploaded, possibly before the model was known: now that the plan is
    // fixed, judge it again if the kernel works on groups of another size than the one assumed then
    if (c->prec == ICP_F32 && c->have_moving && c->n > 128 && pl.sparse && c->moving_group != 0 && c->moving_group != (pl.row == 64 ? 64 : 128) && c->P0.p)
        if (int rc = decide_moving_order(c, c->P0.p, pl.row == 64 ? 64 : 128)) return rc;
    if (before.n_pad != pl.n_pad || before.m_pad != pl.m_pad) c->resident_refused = false;  // another geometry: ask again
    if (before.n_pad != pl.n_pad || before.blocks_x != pl.blocks_x) c->rows_format = -1;     // (rows that were not in use keep old tags: wiped before the next launch)
    const size_t es = icp::elem_size(c->prec);
    const size_t S = pl.splits > 0 ? (size_t)pl.splits : 1;
    HIP_TRY(c->part_d.ensure(S * (size_t)pl.n_pad * es));
    HIP_TRY(c->part_idx.ensure(S * (size_t)pl.n_pad * sizeof(int32_t)));
    HIP_TRY(c->idx[0].ensure((size_t)pl.n_pad * sizeof(int32_t)));
    HIP_TRY(c->idx[1].ensure((size_t)pl.n_pad * sizeof(int32_t)));
    const bool fresh = c->mom_partials.cap == 0;
    size_t rows = (size_t)icp::MOM_MAX_BLOCKS;
    if ((size_t)pl.blocks_x > rows) rows = (size_t)pl.blocks_x;
    if (rows > c->rows_cap) {
        if (c->h_mom_partials) { (void)hipHostFree(c->h_mom_partials); c->h_mom_partials = nullptr; }
        HIP_TRY(hipHostMalloc((void**)&c->h_mom_partials, rows * ICP_NMOM * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(c->h_mom_partials, 0, rows * ICP_NMOM * sizeof(double));
        c->rows_cap = rows;
        c->rows_format = -1;
    }
    HIP_TRY(c->mom_partials.ensure(rows * ICP_NMOM * sizeof(double)));
    if (icp::nn_can_fuse_tail(pl)) {
        const size_t kb = (size_t)pl.n_pad * sizeof(unsigned long long), tb = (size_t)pl.blocks_x * sizeof(unsigned int);
        if (kb > c->keys.cap) {
            HIP_TRY(c->keys.ensure(kb));
            HIP_TRY(hipMemsetAsync(c->keys.p, 0xFF, c->keys.cap, c->stream));   // "no candidate yet"
        }
        if (tb > c->tickets.cap) {
            HIP_TRY(c->tickets.ensure(tb));
            HIP_TRY(hipMemsetAsync(c->tickets.p, 0, c->tickets.cap, c->stream));
        }
    }
    if (pl.share_blocks > 0) {
        const size_t sb = 5 * (size_t)pl.blocks_x * sizeof(unsigned int);
        if (sb > c->share_counts.cap || before.blocks_x != pl.blocks_x || before.share_blocks != pl.share_blocks) {
            HIP_TRY(c->share_counts.ensure(sb));
            HIP_TRY(hipMemsetAsync(c->share_counts.p, 0, c->share_counts.cap, c->stream));   // "nothing known": every row is one block
            c->share_seq = 0;
            c->share_cold_seq = 0;
        }
        HIP_TRY(c->seed_pub.ensure((size_t)pl.blocks_x * 384 * sizeof(float)));
    }
    if (pl.sparse && pl.version == 2 && pl.row != 64 && pl.blocks_x > c->host_rows_max && icp::nn_can_fuse_tail(pl)) {
        // rows added up inside the launch
        if (c->fin_tickets.cap == 0) {
            HIP_TRY(c->fin_tickets.ensure((icp::NN_FIN_GROUPS + 1) * sizeof(unsigned int)));
            HIP_TRY(hipMemsetAsync(c->fin_tickets.p, 0, c->fin_tickets.cap, c->stream));
        }
        HIP_TRY(c->fin_scratch.ensure((size_t)icp::NN_FIN_GROUPS * ICP_NMOM * sizeof(double)));
    }
    c->row_order = nullptr;
    if (pl.order) {
        const size_t rb = (size_t)pl.blocks_x * sizeof(unsigned int);
        if (rb > c->row_hits.cap || before.blocks_x != pl.blocks_x) {
            HIP_TRY(c->row_hits.ensure(rb));
            HIP_TRY(hipMemsetAsync(c->row_hits.p, 0, c->row_hits.cap, c->stream));   // "nothing known": index order
            c->order_regs = 0;
            c->order_launches = 0;
        }
        for (int k = 0; k < 2; ++k) { HIP_TRY(c->order_keys[k].ensure(rb)); HIP_TRY(c->order_vals[k].ensure(rb)); }
        HIP_TRY(c->order_roles.ensure(((size_t)pl.blocks_x + icp::NN_ORDER_EXTRA) * sizeof(int32_t)));
        if (c->order_totals.cap == 0) {
            HIP_TRY(c->order_totals.ensure(2 * sizeof(unsigned long long)));
            HIP_TRY(hipMemsetAsync(c->order_totals.p, 0, c->order_totals.cap, c->stream));
            c->order_seq = 0;
        }
        HIP_TRY(c->order_tmp.ensure(icp::row_order_temp_bytes(pl.blocks_x)));
    }
    // one error row per matching block row (fused transform) or per transform block
    size_t err_rows = (size_t)icp::MOM_MAX_BLOCKS;
    if ((size_t)pl.blocks_x > err_rows) err_rows = (size_t)pl.blocks_x;
    if (err_rows > c->err_cap) {
        if (c->h_err_partials) { (void)hipHostFree(c->h_err_partials); c->h_err_partials = nullptr; }
        HIP_TRY(hipHostMalloc((void**)&c->h_err_partials, err_rows * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(c->h_err_partials, 0, err_rows * sizeof(double));
        c->err_cap = err_rows;
    }
    HIP_TRY(c->err_partials.ensure(err_rows * sizeof(double)));
    if (icp::nn_can_fuse_transform(pl)) HIP_TRY(c->P2.ensure(3 * (size_t)pl.n_pad * es));
    HIP_TRY(c->mom_own.ensure(ICP_NMOM * sizeof(double)));
    if (fresh) {
        HIP_TRY(hipMemsetAsync(c->mom_partials.p, 0, c->mom_partials.cap, c->stream));
        HIP_TRY(hipMemsetAsync(c->err_partials.p, 0, c->err_partials.cap, c->stream));
        HIP_TRY(hipMemsetAsync(c->mom_own.p, 0, c->mom_own.cap, c->stream));
    }
    if (!c->mom_dev) c->mom_dev = (double*)c->mom_own.p;
    return ICP_OK;
}

// upload a host AoS cloud and convert it to the padded SoA layout
static int check_nonfinite(icp_ctx* c, int count);

// deferred: no synchronisation here -- the caller synchronises once, at the end of its set-up, and asks check_nonfinite then
// (soa2: a second copy of the converted cloud; enc: the bounding cube's six words, see launch_aos_to_soa)
int upload_cloud(icp_ctx* c, const void* aos, int count, int pad, int precision, DevBuf& dst, bool deferred = false, void* soa2 = nullptr, unsigned int* enc = nullptr)
{
    const size_t es = icp::elem_size(precision);
    HIP_TRY(dst.ensure(3 * (size_t)pad * es));
    if (count <= 0) return ICP_OK;
    const size_t bytes = 3 * (size_t)count * es;
    const void* src = nullptr;
    if (deferred && bytes <= (4u << 20)) {
        // a small cloud whose set-up ends with a wait anyway: copied by this thread into pinned, mapped memory and laid out straight
        // from there by the layout kernel (one pass over PCIe) -- no copy command, no runtime staging of a pageable source
        if (bytes > c->h_stage_cap) {
            if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_stage_cap = 0; }
            const size_t want = std::max(bytes, (size_t)1 << 20);
            HIP_TRY(hipHostMalloc(&c->h_stage, want, hipHostMallocMapped | hipHostMallocCoherent));
            c->h_stage_cap = want;
        }
        std::memcpy(c->h_stage, aos, bytes);
        src = c->h_stage;
    } else {
        HIP_TRY(c->stage.ensure(bytes));
        HIP_TRY(hipMemcpyAsync(c->stage.p, aos, bytes, hipMemcpyHostToDevice, c->stream));
        src = c->stage.p;
    }
    *(volatile unsigned int*)c->h_nonfinite = 0u;
    HIP_TRY(icp::launch_aos_to_soa(precision, src, count, pad, dst.p, c->stream, c->h_nonfinite, soa2, enc));
    if (deferred) return ICP_OK;
    // the staging buffer is reused by the next upload: order them on the stream, and make sure the
    // pageable host source has been consumed before returning
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_nonfinite(c, count);
}

static int check_nonfinite(icp_ctx* c, int count)
{
    // Non-finite coordinates are refused (include/icp_mi355x.h, "non-finite input": a deliberate deviation).  The reference
    // does not look at its input: the match of such a point is whatever cblas_idamin (src/ICP_CPU.c:232) answers for a vector
    // that holds NaN -- MKL documents nothing -- and the centroid sums (:342-366) then turn the whole transform into NaN:
    // nothing a caller could use, and the pruned search has no bound to go by.
    if (const unsigned int bad = *(volatile unsigned int*)c->h_nonfinite) {
        char msg[160];
        std::snprintf(msg, sizeof msg, "%u of the %d points have a NaN or infinite coordinate: non-finite input is refused", bad, count);
        return fail(ICP_ERR_INVALID, msg);
    }
    return ICP_OK;
}

// ---- spatial order and duplicate flags, on the device ---------------------------------------------------------------
// The sparse matching kernel prunes by bounding boxes of 8 consecutive model points and of 128 consecutive moving
// points: it needs clouds whose index order has spatial locality.  A LiDAR scan has it; a mesh's vertex list
// (Bunny) does not.  Where Morton order makes the groups clearly tighter than the given order, the kernel works
// on a Morton-ordered view (a permutation: the clouds at the ABI and every index it returns stay in user order).
// Sorting and the extent test run on the device (rocPRIM radix sorts, fixed-order reductions): a few dozen
// microseconds per cloud instead of milliseconds of std::sort on the host.
static int prep_buffers(icp_ctx* c, int count, icp::PrepBuffers& b)
{
    const size_t tb = icp::prep_sort_temp_bytes(count);
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(c->prep_keys[k].ensure((size_t)count * sizeof(unsigned int)));
        HIP_TRY(c->prep_vals[k].ensure((size_t)count * sizeof(int32_t)));
    }
    HIP_TRY(c->prep_tmp.ensure(tb));
    HIP_TRY(c->prep_small.ensure(sizeof(icp_ctx::PrepSmall)));
    HIP_TRY(c->prep_ext.ensure((size_t)((count + 7) / 8) * sizeof(double)));
    HIP_TRY(c->prep_voided.ensure((size_t)icp::round_up(count, 16) + 16));
    HIP_TRY(c->prep_perm.ensure((size_t)count * sizeof(int32_t)));
    b.keys[0] = (unsigned int*)c->prep_keys[0].p; b.keys[1] = (unsigned int*)c->prep_keys[1].p;
    b.vals[0] = (int32_t*)c->prep_vals[0].p; b.vals[1] = (int32_t*)c->prep_vals[1].p;
    b.temp = c->prep_tmp.p;
    b.temp_bytes = tb;
    b.box = (float*)c->prep_small.p;
    b.ext = (double*)c->prep_ext.p;
    HIP_TRY(hipMemsetAsync(c->prep_small.p, 0, sizeof(icp_ctx::PrepSmall), c->stream));
    return ICP_OK;
}

// reads the extent totals back and decides: true when Morton order makes the groups at least 3x tighter.  A scan that
// already has locality must keep its order even if Morton cells are tighter: the hall scan's model chunks are 2.1x
// tighter in Morton order, yet matching gets 20 % slower -- its 8-point half columns line up with the moving groups
// (8 columns), compact Morton cells do not; the Bunny vertex list is 10x / 5.8x looser than Morton order.
static int morton_decision(icp_ctx* c, int count, int group, int group2, bool* use_sorted, int* voided_out)
{
    icp_ctx::PrepSmall h{};
    HIP_TRY(hipMemcpyAsync(&h, c->prep_small.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (voided_out) *voided_out = h.voided;
    const int force = c->tune.sort;   // ICP_SORT=0 never, =1 always (A/B runs, tests)
    if (force == 0) *use_sorted = false;
    else if (count <= group) *use_sorted = false;
    else if (force == 1) *use_sorted = true;
    else {
        *use_sorted = 3.0 * h.totals[1] < h.totals[0];
        // a model searched through the box hierarchy: the order also has to serve the level above the chunks (a
        // row-major grid has tight 8-point chunks but 512-point boxes one row thin and a fifth of the cloud long)
        if (group2 > 0 && 3.0 * h.totals[3] < h.totals[2] && h.totals[1] <= h.totals[0]) *use_sorted = true;
    }
    if (c->trace) {
        std::fprintf(stderr, "[icp trace] %d points, groups of %d: extent %.4g in the given order, %.4g in Morton order", count, group, h.totals[0], h.totals[1]);
        if (group2 > 0) std::fprintf(stderr, "; groups of %d: %.4g, %.4g", group2, h.totals[2], h.totals[3]);
        std::fprintf(stderr, " -> %s; %d exact duplicates voided\n", *use_sorted ? "Morton view" : "own order", h.voided);
    }
    return ICP_OK;
}

// The order decision of a small cloud (<= kPrepSmallMax points) with ONE synchronisation: summed group extents of the given order
// and -- unless the remembered decision says it is not needed -- of the Hilbert-curve order, in fixed point relative to the bounding
// cube the layout kernel left in PrepSmall::enc.  Ends the deferred upload: the non-finite count is checked here.
static int decide_order_small(icp_ctx* c, const icp::PrepBuffers& pb, const void* X_soa, int count, int pad, int group, icp_ctx::OrderMemo& memo, bool* use_sorted,
                              int* voided_out, const char* what)
{
    icp_ctx::PrepSmall* small = (icp_ctx::PrepSmall*)c->prep_small.p;
    const int force = c->tune.sort;
    const bool trivial = count <= group || force == 0;                       // never sorted: nothing to measure
    const bool fast = !trivial && force < 0 && memo.valid && memo.count == count && memo.group == group && !memo.sorted;
    bool have_sorted = false;
    unsigned int seq = 0;
    auto sorted_extents = [&](int which) -> int {
        HIP_TRY(icp::launch_curve_order_small(pb, (const float*)X_soa, count, pad, small->enc, (int32_t*)c->prep_perm.p, c->stream));
        seq = ++c->prep_seq ? c->prep_seq : ++c->prep_seq;
        HIP_TRY(icp::launch_extents_fixed((const float*)X_soa, count, pad, (const int32_t*)c->prep_perm.p, group, small->enc, small->fixed, which, c->stream,
                                          &small->ticket, &small->voided, c->h_prep, seq));
        have_sorted = true;
        return ICP_OK;
    };
    // the launch's last block leaves the sums in pinned memory: the host spins on the sequence word (a copy back and a stream
    // synchronisation cost 15-20 us more); should the word never come, the runtime says why
    struct Report { unsigned long long fixed[4]; int voided; };
    auto wait_report = [&](Report& h) -> int {
        if (seq != 0) {
            const auto t0 = std::chrono::steady_clock::now();
            const volatile unsigned int* w = &c->h_prep->seq;
            bool there = false;
            for (unsigned spins = 1; !(there = *w == seq); ++spins)
                if ((spins & 0x3ff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
            if (there) {
                std::atomic_thread_fence(std::memory_order_acquire);
                for (int k = 0; k < 4; ++k) h.fixed[k] = c->h_prep->fixed[k];
                h.voided = c->h_prep->voided;
                return ICP_OK;
            }
        }
        icp_ctx::PrepSmall full{};
        HIP_TRY(hipMemcpyAsync(&full, c->prep_small.p, sizeof full, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int k = 0; k < 4; ++k) h.fixed[k] = full.fixed[k];
        h.voided = full.voided;
        return ICP_OK;
    };
    if (!trivial) {
        if (fast) {
            seq = ++c->prep_seq ? c->prep_seq : ++c->prep_seq;
            HIP_TRY(icp::launch_extents_fixed((const float*)X_soa, count, pad, nullptr, group, small->enc, small->fixed, 0, c->stream, &small->ticket, &small->voided, c->h_prep, seq));
        } else if (int rc = sorted_extents(0)) return rc;
    }
    Report h{};
    if (int rc = wait_report(h)) return rc;
    if (int rc = check_nonfinite(c, count)) return rc;
    constexpr double kFix = 1.0 / 68719476736.0;   // 2^-36
    double given = (double)h.fixed[0] * kFix, sorted = (double)h.fixed[1] * kFix;
    if (fast && !(given <= 1.25 * memo.given_rel)) {
        // the cloud is not what the one before was: measure the curve order after all (a second short round trip, once)
        if (int rc = sorted_extents(1)) return rc;
        if (int rc = wait_report(h)) return rc;
        given = (double)h.fixed[2] * kFix;
        sorted = (double)h.fixed[3] * kFix;
    }
    if (voided_out) *voided_out = h.voided;
    if (trivial) *use_sorted = false;
    else if (force == 1) *use_sorted = true;
    else if (!have_sorted) *use_sorted = false;                              // (remembered: own order, and the cloud still looks the same)
    else *use_sorted = 3.0 * sorted < given;
    if (c->trace) {
        std::fprintf(stderr, "[icp trace] %s: %d points, groups of %d: extent %.4g of the bounding cube's edge in the given order", what, count, group, given);
        if (have_sorted) std::fprintf(stderr, ", %.4g along the Hilbert curve", sorted); else std::fprintf(stderr, " (curve order not measured: %s)", trivial ? "not applicable" : "as the cloud before");
        std::fprintf(stderr, " -> %s; %d exact duplicates voided\n", *use_sorted ? "sorted view" : "own order", h.voided);
    }
    if (!trivial && force < 0) { memo.valid = true; memo.count = count; memo.group = group; memo.sorted = *use_sorted; if (have_sorted || !memo.given_rel) memo.given_rel = given; }
    return ICP_OK;
}

int check_precision(int precision)
{
    if (precision != ICP_F32 && precision != ICP_F64) return fail(ICP_ERR_INVALID, "unknown precision");
    return ICP_OK;
}

}  // namespace

extern "C" {

int icp_abi_version(void) { return ICP_ABI_VERSION; }

const char* icp_strerror(int code)
{
    switch (code) {
        case ICP_OK: return "ok";
        case ICP_ERR_INVALID: return "invalid argument";
        case ICP_ERR_NO_DEVICE: return "no usable gfx950 HIP device (there is no CPU fallback)";
        case ICP_ERR_HIP: return "HIP runtime error";
        case ICP_ERR_EMPTY: return "empty model cloud";
        case ICP_ERR_SINGULAR: return "point-to-plane system is not positive definite";
        case ICP_ERR_IO: return "dataset file missing or malformed";
        case ICP_ERR_STATE: return "call sequence error";
        case ICP_ERR_NOMEM: return "out of memory";
        default: return "unknown error";
    }
}

const char* icp_last_error(void) { return g_last_error.c_str(); }

int icp_device_count(void)
{
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ICP_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    return n;
}

// The loop is a conversation between one host thread and the GPU (mailbox through the PCIe BAR, moment rows through pinned
// host memory): on a two-socket machine every message of a thread running on the other socket crosses the socket link
// too -- measured on the hall loop 13.05 us per iteration from the GPU's own NUMA node, 14.4-14.6 us from the other one,
// and a coin toss when the scheduler chooses (tools/numa_probe.py).  The library therefore wants the calling thread on a
// CPU that sysfs lists as local to the device -- but a drop-in library must not leave its caller's affinity changed.
// So the narrowing is SCOPED: an entry point that talks to the GPU in a loop (icp_create while it allocates and first
// touches the pinned buffers, icp_loop_run, icp_loop_complete, icp_point_to_*) narrows the mask only if the thread is
// currently running on a remote CPU, and puts the caller's mask back before it returns.
//   ICP_PIN=0  never touch the affinity;  ICP_PIN=1 (default) scoped as above;
//   ICP_PIN=2  narrow once in icp_create and leave it narrowed (the behaviour of round 1; a dedicated worker thread).
static bool parse_local_cpus(int device, cpu_set_t* local)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return false; }
    for (char* p = bus; *p; ++p) *p = (char)std::tolower((unsigned char)*p);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
    std::FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    char line[4096] = {0};
    const bool got = std::fgets(line, sizeof line, f) != nullptr;
    std::fclose(f);
    if (!got) return false;
    CPU_ZERO(local);
    for (const char* p = line; *p;) {                      // "0-63,128-191"
        char* end = nullptr;
        const long a = std::strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') { b = std::strtol(p + 1, &end, 10); p = end; }
        for (long k = a; k <= b && k < CPU_SETSIZE; ++k) if (k >= 0) CPU_SET((int)k, local);
        if (*p == ',') ++p; else break;
    }
    return CPU_COUNT(local) > 0;
}

int icp_create(int device, icp_ctx** out)
{
    if (!out) return fail(ICP_ERR_INVALID, "out == NULL");
    *out = nullptr;
    const int nd = icp_device_count();
    if (nd <= 0) return fail(ICP_ERR_NO_DEVICE, "no HIP device visible: " + g_last_error);
    if (device < 0 || device >= nd) return fail(ICP_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ICP_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    icp_ctx* c = new (std::nothrow) icp_ctx();
    if (!c) return fail(ICP_ERR_NOMEM, "context allocation failed");
    c->device = device;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* v = std::getenv("ICP_PIN")) c->pin_mode = (v[0] == '0') ? 0 : (v[0] == '2') ? 2 : 1;
    c->have_local_cpus = c->pin_mode != 0 && parse_local_cpus(device, &c->local_cpus);
    if (c->pin_mode == 2 && c->have_local_cpus) {   // narrowed once and kept (a thread dedicated to this context)
        cpu_set_t cur, both;
        if (sched_getaffinity(0, sizeof cur, &cur) == 0) {
            CPU_AND(&both, &cur, &c->local_cpus);
            if (CPU_COUNT(&both) != 0 && CPU_COUNT(&both) != CPU_COUNT(&cur)) (void)sched_setaffinity(0, sizeof both, &both);
        }
    }
    ScopedPin pin(c);   // (the pinned host buffers below are allocated and first touched next to the device)
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_mom, ICP_NMOM * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_nonfinite, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_final, ICP_NMOM * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) std::memset(c->h_final, 0, ICP_NMOM * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_prep, sizeof(icp::PrepReport), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) std::memset(c->h_prep, 0, sizeof(icp::PrepReport));
    if (e == hipSuccess) {
        // mailbox: fine-grained device memory written through the PCIe BAR when the machine allows it (every block
        // polls its own memory), else pinned host memory polled by block 0 and relayed (ICP_MAILBOX=host forces that)
        // (ICP_MAILBOX: a comma list of `host` -- pinned memory + relay -- and `plain` -- the line written word by word)
        const char* mv = std::getenv("ICP_MAILBOX");
        if (mv && std::strstr(mv, "plain")) c->mail_wide = false;
        int large_bar = 0;
        if (!(mv && std::strstr(mv, "host")) && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) == hipSuccess && large_bar &&
            hipExtMallocWithFlags((void**)&c->h_mail, kMailSlots * kMailSlotBytes, hipDeviceMallocFinegrained) == hipSuccess) {
            c->mail_in_bar = true;
        } else {
            (void)hipGetLastError();
            e = hipHostMalloc((void**)&c->h_mail, kMailSlots * kMailSlotBytes, hipHostMallocMapped | hipHostMallocCoherent);
        }
        if (e == hipSuccess) { std::memset(c->h_mail, 0, kMailSlots * kMailSlotBytes); bar_fence(); }
    }
    if (e == hipSuccess) {
        if (hipExtMallocWithFlags((void**)&c->relay, kMailSlotBytes, hipDeviceMallocFinegrained) == hipSuccess) {
            e = hipMemset(c->relay, 0, kMailSlotBytes);
        } else {
            (void)hipGetLastError();
            c->relay = nullptr;  // no armed / resident launches on this device: every pass is launched after its solve
        }
    }

    if (e != hipSuccess) {
        const std::string msg = std::string("context setup: ") + hipGetErrorString(e);
        icp_destroy(c);
        return fail(ICP_ERR_HIP, msg);
    }
    c->stream = c->own_stream;
    // test hooks, one variable: ICP_DEBUG="stall=pass:seconds,lose=pass,shared_resident" (tests/test_gpu_runtime.py, tests/test_gpu_parity.py)
    if (const char* v = std::getenv("ICP_DEBUG")) {
        if (const char* q = std::strstr(v, "lose=")) c->debug_lose_pass = std::atoi(q + 5);
        if (const char* q = std::strstr(v, "stall=")) {
            int pass = -1;
            double sec = 0.0;
            if (std::sscanf(q + 6, "%d:%lf", &pass, &sec) == 2 && pass >= 0 && sec > 0.0 && sec < 30.0) { c->debug_stall_pass = pass; c->debug_stall_s = sec; }
        }
        c->debug_shared_resident = std::strstr(v, "shared_resident") != nullptr;
    }
    c->tune = icp::nn_tuning_from_env();
    c->use_boxes = c->tune.sparse != 0;
    if (const char* v = std::getenv("ICP_ARMED")) c->arm = !(v[0] == '0');
    if (const char* v = std::getenv("ICP_RESIDENT")) c->resident = v[0] == '0' ? 0 : (v[0] == '2' ? 2 : 1);
    if (const char* v = std::getenv("ICP_SHARE_RESIDENT_AFTER")) c->share_resident_after = std::atoi(v);
    if (const char* v = std::getenv("ICP_SHARE_AUTO")) c->share_auto = !(v[0] == '0');
    if (const char* v = std::getenv("ICP_NN_SPLIT_MIN")) c->split_min = std::max(0, std::atoi(v));   // (A/B runs and tests)
    if (const char* v = std::getenv("ICP_HOST_ROWS_MAX")) c->host_rows_max = std::max(1, std::atoi(v));
    if (const char* v = std::getenv("ICP_TRACE")) { c->trace = v[0] == '1' || v[0] == '2'; c->trace_passes = v[0] == '2'; }
    if (const char* v = std::getenv("ICP_FUSED_TAIL")) c->fused_tail = !(v[0] == '0');
    if (const char* v = std::getenv("ICP_NN_PHASES")) {
        // diagnostic, ICP_NN_PHASES=file[:pass[:slots[:wipe]]] -- the matching kernel stamps its phases per wave; the last launch's
        // stamps are written to the named file (raw int64) when the context is destroyed -- tools/phase_report.py reads it.
        // pass: stamp this pass of a resident launch only (-1 / empty: every pass, the last one survives); slots: room for more than
        // the default 3277 sixteen-wave blocks (160 stamps a block); wipe = 1: the log is cleared ahead of every launch
        std::string spec = v;
        std::vector<std::string> part;
        for (size_t at = 0;;) { const size_t q = spec.find(':', at); part.push_back(spec.substr(at, q == std::string::npos ? q : q - at)); if (q == std::string::npos) break; at = q + 1; }
        size_t slots = kPhaseSlots;
        if (part.size() > 1 && !part[1].empty()) c->tune.phase_pass = std::atoi(part[1].c_str());
        if (part.size() > 2 && !part[2].empty()) { const long long w = std::atoll(part[2].c_str()); if (w > 0 && w <= (1ll << 28)) slots = (size_t)w; }
        if (part.size() > 3 && !part[3].empty()) c->tune.phase_wipe = std::atoi(part[3].c_str()) != 0;
        if (!part[0].empty() && c->phase_log.ensure(slots * sizeof(long long)) == hipSuccess &&
            hipMemset(c->phase_log.p, 0, slots * sizeof(long long)) == hipSuccess) {
            c->phase_path = part[0];
            c->phase_slots = slots;
            c->tune.phase_log = (long long*)c->phase_log.p;
            c->tune.phase_cap = (long long)slots;
        }
    }
    *out = c;
    return ICP_OK;
}

void icp_destroy(icp_ctx* c)
{
    if (!c) return;
    if (c->trace && c->tr_n)
        std::fprintf(stderr, "[icp trace] %llu iterations: enqueue %.2f us, wait %.2f us, reduce %.2f us, solve %.2f us (host, per iteration)\n",
                     (unsigned long long)c->tr_n, 1e6 * c->tr_enqueue / c->tr_n, 1e6 * c->tr_wait / c->tr_n,
                     1e6 * c->tr_reduce / c->tr_n, 1e6 * c->tr_solve / c->tr_n);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) { icp::comm_destroy(c->comm); c->comm = nullptr; }
    if (c->lcomm) { icp::lcomm_destroy(c->lcomm); c->lcomm = nullptr; }
    if (c->phase_log.p && !c->phase_path.empty()) {
        std::vector<long long> h(c->phase_slots);
        if (hipMemcpy(h.data(), c->phase_log.p, c->phase_slots * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* f = std::fopen(c->phase_path.c_str(), "wb")) { std::fwrite(h.data(), sizeof(long long), h.size(), f); std::fclose(f); }
        }
        c->phase_log.release();
    }
    DevBuf* bufs[] = {&c->work, &c->fin_scratch, &c->fin_tickets, &c->dup_table, &c->slot_state, &c->share_counts, &c->seed_pub, &c->row_hits, &c->order_keys[0], &c->order_keys[1], &c->order_vals[0], &c->order_vals[1], &c->order_tmp, &c->order_roles, &c->order_totals, &c->P0, &c->P, &c->P2, &c->Q, &c->Qs, &c->Qbox, &c->Qrec, &c->Qsamp, &c->Qss, &c->Qperm, &c->Pperm, &c->prep_keys[0], &c->prep_keys[1], &c->prep_vals[0], &c->prep_vals[1], &c->prep_tmp, &c->prep_small, &c->prep_ext, &c->prep_voided, &c->prep_perm, &c->Nrm, &c->stage, &c->part_d, &c->part_idx, &c->idx[0], &c->idx[1],
                      &c->mom_partials, &c->err_partials, &c->mom_own, &c->nbr, &c->keys, &c->tickets};
    for (DevBuf* b : bufs) b->release();
    if (c->h_mom) (void)hipHostFree(c->h_mom);
    if (c->h_nonfinite) (void)hipHostFree(c->h_nonfinite);
    if (c->h_final) (void)hipHostFree(c->h_final);
    if (c->h_prep) (void)hipHostFree(c->h_prep);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_mail) { if (c->mail_in_bar) (void)hipFree(c->h_mail); else (void)hipHostFree(c->h_mail); }
    if (c->relay) (void)hipFree(c->relay);
    if (c->h_mom_partials) (void)hipHostFree(c->h_mom_partials);
    if (c->h_err_partials) (void)hipHostFree(c->h_err_partials);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int icp_comm_unique_id(void* out_bytes)
{
    if (!out_bytes) return fail(ICP_ERR_INVALID, "out == NULL");
    std::string err;
    const int rc = icp::comm_unique_id(out_bytes, err);
    return rc == ICP_OK ? ICP_OK : fail(rc, err);
}

int icp_comm_init(icp_ctx* c, const void* id_bytes, int rank, int world)
{
    if (int rc = use(c)) return rc;
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) return fail(ICP_ERR_INVALID, "bad communicator arguments");
    if (c->loop.pending) return fail(ICP_ERR_STATE, "an enqueue is in flight");
    if (c->comm) { icp::comm_destroy(c->comm); c->comm = nullptr; }
    HIP_TRY(c->mom_own.ensure(ICP_NMOM * sizeof(double)));
    if (!c->mom_dev) c->mom_dev = (double*)c->mom_own.p;
    std::string err;
    const int rc = icp::comm_init(id_bytes, rank, world, &c->comm, err);
    return rc == ICP_OK ? ICP_OK : fail(rc, err);
}

int icp_comm_destroy(icp_ctx* c)
{
    if (int rc = use(c)) return rc;
    if (c->loop.pending) return fail(ICP_ERR_STATE, "an enqueue is in flight");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->comm) { icp::comm_destroy(c->comm); c->comm = nullptr; }
    if (c->lcomm) { icp::lcomm_destroy(c->lcomm); c->lcomm = nullptr; c->shares_device = false; }
    return ICP_OK;
}

int icp_comm_random_id(void* out_bytes)
{
    if (!out_bytes) return fail(ICP_ERR_INVALID, "out == NULL");
    std::memset(out_bytes, 0, ICP_COMM_ID_BYTES);
    FILE* f = std::fopen("/dev/urandom", "rb");
    size_t got = f ? std::fread(out_bytes, 1, 16, f) : 0;
    if (f) std::fclose(f);
    if (got != 16) {  // fall back to clock + pid: unique enough for a segment name on one node
        const uint64_t a = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count(), b = (uint64_t)getpid();
        std::memcpy(out_bytes, &a, 8);
        std::memcpy((char*)out_bytes + 8, &b, 8);
    }
    return ICP_OK;
}

int icp_comm_init_local(icp_ctx* c, const void* id_bytes, int rank, int world)
{
    if (int rc = use(c)) return rc;
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) return fail(ICP_ERR_INVALID, "bad communicator arguments");
    if (c->loop.pending) return fail(ICP_ERR_STATE, "an enqueue is in flight");
    if (c->comm) return fail(ICP_ERR_STATE, "a device communicator is attached: destroy it first");
    // (a caller that installed its own moments buffer sums it across ranks itself, between enqueue and complete: with the node
    // communicator on top the vector would be summed twice -- the advisor's finding on round 3)
    if (c->mom_dev != nullptr && c->mom_dev != (double*)c->mom_own.p)
        return fail(ICP_ERR_STATE, "an external moments buffer is installed (icp_loop_set_moments_dev): its owner reduces it; remove it first");
    if (c->lcomm) { icp::lcomm_destroy(c->lcomm); c->lcomm = nullptr; }
    std::string err;
    const int rc = icp::lcomm_create(id_bytes, rank, world, &c->lcomm, err);
    if (rc != ICP_OK) return fail(rc, err);
    // Do two ranks of this communicator sit on ONE device (a rehearsal; the deployment is one process per GPU)?  Then no pass is
    // armed ahead of its (R, t): the waiting blocks of one rank can keep the running pass of the other off the CUs, and with an
    // exchange between the ranks that is a circular wait (seen with two ranks of the 10 M-point configuration on one GPU: a
    // pass missing its last rows after the 2 s poll budget).  Every rank leaves its device's PCI address in its slot of one
    // sum; plain launches wait on nothing that is not running.
    c->shares_device = false;
    if (world > 1 && world <= ICP_NMOM) {
        hipDeviceProp_t prop{};
        double ids[ICP_NMOM] = {0};
        // (an address of all zeros is "unknown": such ranks are not taken to share anything)
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && (prop.pciDomainID | prop.pciBusID | prop.pciDeviceID) != 0)
            ids[rank] = (double)(((long long)prop.pciDomainID << 16) | ((long long)prop.pciBusID << 8) | (long long)prop.pciDeviceID);
        if (int rc2 = icp::lcomm_allreduce_sum_f64(c->lcomm, ids, world, err)) return fail(rc2, err);
        for (int r = 0; r < world; ++r)
            if (r != rank && ids[r] != 0.0 && ids[r] == ids[rank]) c->shares_device = true;
    }
    return ICP_OK;
}

struct icp_lcomm { icp::LocalComm* p; };

int icp_lcomm_create(const void* id_bytes, int rank, int world, icp_lcomm** out)
{
    if (!out) return fail(ICP_ERR_INVALID, "out == NULL");
    *out = nullptr;
    std::string err;
    icp::LocalComm* p = nullptr;
    if (int rc = icp::lcomm_create(id_bytes, rank, world, &p, err)) return fail(rc, err);
    *out = new icp_lcomm{p};
    return ICP_OK;
}

int icp_lcomm_allreduce(icp_lcomm* h, double* v, int count)
{
    if (!h) return fail(ICP_ERR_INVALID, "null communicator");
    std::string err;
    const int rc = icp::lcomm_allreduce_sum_f64(h->p, v, count, err);
    return rc == ICP_OK ? ICP_OK : fail(rc, err);
}

void icp_lcomm_destroy(icp_lcomm* h)
{
    if (!h) return;
    icp::lcomm_destroy(h->p);
    delete h;
}

int icp_set_stream(icp_ctx* c, void* hip_stream)
{
    if (int rc = use(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return ICP_OK;
}

int icp_set_profiling(icp_ctx* c, int enable)
{
    if (!c) return fail(ICP_ERR_INVALID, "null context");
    c->profiling = enable != 0;
    c->profile_stride = enable > 0 ? enable : 0;
    c->prof_seconds_nn = 0.0;
    c->prof_nn_launches = 0;
    c->prof_nn_passes = 0;
    // the stride counts from here: the first launch after this call is a timed one
    c->nn_launch_count = 0;
    c->resident_launch_count = 0;
    return ICP_OK;
}

int icp_set_exclusive(icp_ctx* c, int on)
{
    if (int rc = use(c)) return rc;
    if (c->loop.pending) return fail(ICP_ERR_STATE, "an enqueue is in flight");
    if (c->exclusive != (on != 0)) c->resident_refused = false;   // (another kernel variant: the occupancy question is asked again)
    c->exclusive = on != 0;
    return ICP_OK;
}

int icp_set_work_counting(icp_ctx* c, int enable)
{
    if (int rc = use(c)) return rc;
    if (c->loop.pending) return fail(ICP_ERR_STATE, "an enqueue is in flight");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (enable) {
        HIP_TRY(c->work.ensure(icp::NN_WORK_SLOTS * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(c->work.p, 0, icp::NN_WORK_SLOTS * sizeof(unsigned long long)));
    }
    c->count_work = enable != 0;
    return ICP_OK;
}

int icp_get_work_counters(icp_ctx* c, uint64_t* out, int reset)
{
    if (int rc = use(c)) return rc;
    if (!out) return fail(ICP_ERR_INVALID, "out == NULL");
    if (!c->work.p) return fail(ICP_ERR_STATE, "work counting was never enabled");
    if (c->loop.pending) return fail(ICP_ERR_STATE, "an enqueue is in flight");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->work.p, icp::NN_WORK_SLOTS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(c->work.p, 0, icp::NN_WORK_SLOTS * sizeof(unsigned long long)));
    return ICP_OK;
}

int icp_set_model(icp_ctx* c, const void* xyz, int m, int precision)
{
    if (int rc = use(c)) return rc;
    if (int rc = check_precision(precision)) return rc;
    if (m < 0 || (m > 0 && !xyz)) return fail(ICP_ERR_INVALID, "bad model cloud");
    if (c->have_moving && c->prec != precision) { c->have_moving = false; c->n = 0; }
    c->prec = precision;
    c->m = m;
    c->have_normals = false;
    c->loop.active = false;
    c->idx_valid = false;
    c->have_model = false;     // (until the upload has been accepted)
    c->have_scan_copy = false;
    c->have_records = false;
    // (the model's size decides the search form -- together with the CLOUD's: a model of 2^16 .. 2^17 points is searched through the
    // hierarchy by a cloud of more rows than shared 8-wave blocks serve, flat by a smaller one, and the model is set before the cloud is
    // known: it gets the upper levels and the records whenever SOME cloud would ask for them.  Round 3 built them by the plan of a
    // one-row cloud; a 65 536-point grid against itself then failed with "invalid argument" at its first pass.)
    const int group2 = (precision == ICP_F32 && m > 0 && (icp::nn_plan(128, m, precision, c->num_cus, c->tune).hier ||
                                                          icp::nn_plan(1 << 22, m, precision, c->num_cus, c->tune).hier)) ? 512 : 0;
    const bool short_setup = precision == ICP_F32 && m > 0 && m <= icp_ctx::kPrepSmallMax && group2 == 0;
    icp::PrepBuffers pb{};
    if (short_setup) {
        // (round 4: upload, layout and bounding cube without a synchronisation of their own; see decide_order_small)
        if (int rc = prep_buffers(c, m, pb)) return rc;
        icp_ctx::PrepSmall* small0 = (icp_ctx::PrepSmall*)c->prep_small.p;
        if (int rc = upload_cloud(c, xyz, m, icp::pad_model(m), precision, c->Q, true, nullptr, small0->enc)) return rc;
    } else if (int rc = upload_cloud(c, xyz, m, icp::pad_model(m), precision, c->Q)) return rc;
    if (precision == ICP_F32 && m > 0) {
        // scan copy for the early-out matching kernels: exact duplicates of a lower-index point (and the padding)
        // voided to +inf -- they can never be the lowest-index minimum (see NNCullInputs).  Flags, Morton order and
        // the extent test are computed on the device from the uploaded cloud.
        const int m_pad = icp::pad_model(m);
        if (!short_setup) if (int rc = prep_buffers(c, m, pb)) return rc;
        icp_ctx::PrepSmall* small = (icp_ctx::PrepSmall*)c->prep_small.p;
        HIP_TRY(c->Qs.ensure(3 * (size_t)m_pad * sizeof(float)));
        if (m <= (1 << 21)) {
            // exact duplicates by hashing: two launches instead of three radix sorts (icp_k_setup.hip)
            unsigned int entries = 1024u;
            while (entries < 2u * (unsigned int)m) entries <<= 1;
            if ((size_t)entries * sizeof(unsigned int) > c->dup_table.cap) {
                HIP_TRY(c->dup_table.ensure((size_t)entries * sizeof(unsigned int)));
                HIP_TRY(hipMemsetAsync(c->dup_table.p, 0, c->dup_table.cap, c->stream));
                c->dup_gen = 0;
            }
            if (++c->dup_gen > 255u) {   // (generation 0 is "never written")
                HIP_TRY(hipMemsetAsync(c->dup_table.p, 0, c->dup_table.cap, c->stream));
                c->dup_gen = 1;
            }
            HIP_TRY(icp::launch_duplicates_hashed((const float*)c->Q.p, m, m_pad, (unsigned int*)c->dup_table.p, entries, c->dup_gen, (unsigned char*)c->prep_voided.p,
                                                  &small->voided, (float*)c->Qs.p, c->stream));
        } else {
            HIP_TRY(icp::launch_duplicates_and_scan_copy(pb, (const float*)c->Q.p, m, m_pad, (unsigned char*)c->prep_voided.p, &small->voided,
                                                         (float*)c->Qs.p, c->stream));
        }
        if (short_setup) {
            if (int rc = decide_order_small(c, pb, c->Q.p, m, m_pad, 8, c->memo_model, &c->model_sorted, &c->voided, "model")) return rc;
        } else {
            HIP_TRY(icp::launch_morton_order(pb, (const float*)c->Q.p, m, m_pad, 8, group2, (int32_t*)c->prep_perm.p, small->totals, c->stream));
            if (int rc = morton_decision(c, m, 8, group2, &c->model_sorted, &c->voided)) return rc;
        }
        // the sparse kernel's view: the same voided copy, in Morton order if the model's own order has no locality
        const void* view = c->Qs.p;
        if (c->model_sorted) {
            HIP_TRY(c->Qss.ensure(3 * (size_t)m_pad * sizeof(float)));
            HIP_TRY(c->Qperm.ensure((size_t)m_pad * sizeof(int32_t)));
            HIP_TRY(icp::launch_gather_sorted((const float*)c->Qs.p, m, m_pad, (const int32_t*)c->prep_perm.p, (float*)c->Qss.p,
                                              (int32_t*)c->Qperm.p, c->stream));
            view = c->Qss.p;
        }
        // bounding boxes of its 8-point chunks (the first, cheapest level of the early-out) and one point per chunk
        HIP_TRY(c->Qbox.ensure(icp::model_boxes_bytes(m_pad)));
        HIP_TRY(c->Qsamp.ensure(icp::model_samples_bytes(m_pad)));
        if (group2 == 0) {   // (searched flat: the upper box levels are never read -- one launch for boxes and samples)
            HIP_TRY(icp::launch_model_boxes_samples(view, m_pad, (float*)c->Qbox.p, (float*)c->Qsamp.p, c->stream));
        } else {
            HIP_TRY(icp::launch_model_boxes(view, m_pad, (float*)c->Qbox.p, c->stream));
            HIP_TRY(icp::launch_model_samples(view, m_pad, (float*)c->Qsamp.p, c->stream));
        }
        c->have_records = false;
        if (group2 > 0) {   // a model searched through the box hierarchy: the hits are fetched from per-chunk records
            HIP_TRY(c->Qrec.ensure(icp::model_records_bytes(m_pad)));
            HIP_TRY(icp::launch_model_records(view, (const float*)c->Qbox.p, c->model_sorted ? (const int32_t*)c->Qperm.p : nullptr, m_pad, (float*)c->Qrec.p, c->stream));
            c->have_records = true;
        }
        c->have_scan_copy = true;
    }
    if (precision == ICP_F64 && m > 0) {
        // fp64 on the sparse structure: the scan copy (exact duplicates of a lower-index point and the padding voided to
        // +inf -- the hall scan's 4361 coincident points would otherwise put 545 chunks on every origin point's hit list),
        // its chunk boxes and the cold-start samples, all in double.  No Morton view is built: the CPU path's clouds are
        // grids and scans, which have locality.
        const int m_pad = icp::pad_model(m);
        icp::PrepBuffers pb{};
        if (int rc = prep_buffers(c, m, pb)) return rc;
        icp_ctx::PrepSmall* small = (icp_ctx::PrepSmall*)c->prep_small.p;
        HIP_TRY(c->Qs.ensure(3 * (size_t)m_pad * sizeof(double)));
        HIP_TRY(icp::launch_duplicates_and_scan_copy_f64(pb, (const double*)c->Q.p, m, m_pad, (unsigned char*)c->prep_voided.p, &small->voided,
                                                         (double*)c->Qs.p, c->stream));
        HIP_TRY(c->Qbox.ensure(icp::model_boxes_f64_bytes(m_pad)));
        HIP_TRY(c->Qsamp.ensure(icp::model_samples_f64_bytes(m_pad)));
        HIP_TRY(icp::launch_model_tables_f64(c->Qs.p, m_pad, c->Qbox.p, c->Qsamp.p, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->model_sorted = false;
        c->have_scan_copy = true;
    }
    c->have_model = true;
    return ICP_OK;
}

int icp_set_moving(icp_ctx* c, const void* xyz, int n, int precision)
{
    if (int rc = use(c)) return rc;
    if (int rc = check_precision(precision)) return rc;
    if (n < 0 || (n > 0 && !xyz)) return fail(ICP_ERR_INVALID, "bad moving cloud");
    if (c->have_model && c->prec != precision)
        return fail(ICP_ERR_INVALID, "moving cloud precision differs from the resident model");
    c->prec = precision;
    c->n = n;
    c->loop.active = false;
    c->idx_valid = false;
    c->have_moving = false;    // (until the upload has been accepted)
    const bool judged = precision == ICP_F32 && n > 128;
    const bool short_setup = judged && n <= icp_ctx::kPrepSmallMax;
    if (n > 0) HIP_TRY(c->P0.ensure(3 * (size_t)icp::pad_moving(n) * icp::elem_size(precision)));
    if (short_setup) {
        // (round 4: one layout launch writes the cloud, its pristine copy and the bounding cube; no synchronisation until the order is decided)
        icp::PrepBuffers pb{};
        if (int rc = prep_buffers(c, n, pb)) return rc;
        if (int rc = upload_cloud(c, xyz, n, icp::pad_moving(n), precision, c->P, true, c->P0.p, ((icp_ctx::PrepSmall*)c->prep_small.p)->enc)) return rc;
    } else if (int rc = upload_cloud(c, xyz, n, icp::pad_moving(n), precision, c->P, false, n > 0 ? c->P0.p : nullptr)) return rc;
    c->moving_sorted = false;
    c->moving_group = 0;
    if (judged) {
        // judged on the groups the matching kernel will work on (rows of 64 or of 128 points: nn_plan's rule, overrides
        // included); with no model resident yet the plan assumes one of the moving cloud's size -- ensure_work_buffers looks again
        const icp::NNPlan guess = icp::nn_plan(n, c->have_model && c->m > 0 ? c->m : n, precision, c->num_cus, c->tune);
        if (int rc = decide_moving_order(c, c->P.p, (guess.sparse && guess.row == 64) ? 64 : 128, short_setup)) return rc;
    }
    c->have_moving = true;
    c->moving_is_pristine = false;
    c->moving_untouched = true;
    return ICP_OK;
}

// Morton order or the given one for the moving cloud's slots ("spatial order"): decided on groups of `grp`
extern "C++" int decide_moving_order(icp_ctx* c, const void* P_soa, int grp, bool have_enc)
{
    const int n = c->n, n_pad = icp::pad_moving(n);
    icp::PrepBuffers pb{};
    if (have_enc) {
        // (the upload has prepared the buffers and left the bounding cube: the short form, which also ends the deferred upload)
        pb.keys[0] = (unsigned int*)c->prep_keys[0].p; pb.keys[1] = (unsigned int*)c->prep_keys[1].p;
        pb.vals[0] = (int32_t*)c->prep_vals[0].p; pb.vals[1] = (int32_t*)c->prep_vals[1].p;
        pb.temp = c->prep_tmp.p; pb.temp_bytes = icp::prep_sort_temp_bytes(n);
        pb.box = (float*)c->prep_small.p; pb.ext = (double*)c->prep_ext.p;
        if (int rc = decide_order_small(c, pb, P_soa, n, n_pad, grp, c->memo_moving, &c->moving_sorted, nullptr, "moving cloud")) return rc;
    } else {
    if (int rc = prep_buffers(c, n, pb)) return rc;
    icp_ctx::PrepSmall* small = (icp_ctx::PrepSmall*)c->prep_small.p;
    HIP_TRY(icp::launch_morton_order(pb, (const float*)P_soa, n, n_pad, grp, 0, (int32_t*)c->prep_perm.p, small->totals, c->stream));
    if (int rc = morton_decision(c, n, grp, 0, &c->moving_sorted, nullptr)) return rc;
    }
    if (c->moving_sorted) {
        HIP_TRY(c->Pperm.ensure((size_t)n_pad * sizeof(int32_t)));
        HIP_TRY(icp::launch_slot_map((const int32_t*)c->prep_perm.p, n, n_pad, (int32_t*)c->Pperm.p, c->stream));
    }
    c->moving_group = grp;
    return ICP_OK;
}

int icp_reset_moving(icp_ctx* c)
{
    if (int rc = use(c)) return rc;
    if (!c->have_moving) return fail(ICP_ERR_STATE, "no moving cloud resident");
    if (c->loop.pending) return fail(ICP_ERR_STATE, "an enqueue is in flight");
    c->moving_is_pristine = true;
    c->moving_untouched = true;
    c->loop.active = false;
    c->idx_valid = false;
    return ICP_OK;
}

int icp_set_model_normals(icp_ctx* c, const void* nxyz, int m)
{
    if (int rc = use(c)) return rc;
    if (!c->have_model) return fail(ICP_ERR_STATE, "set the model before its normals");
    if (m != c->m || (m > 0 && !nxyz)) return fail(ICP_ERR_INVALID, "normal count must equal the model size");
    c->have_normals = false;
    if (int rc = upload_cloud(c, nxyz, m, icp::pad_model(m), c->prec, c->Nrm)) return rc;
    c->have_normals = true;
    return ICP_OK;
}

// Rows added up inside the matching launch (NNTail::fin_*, icp_device.h): the sparse kernels with rows of 128 points, a fused tail,
// and more rows than the host takes (host_rows_max).  to_host: the vector lands in pinned memory with the pass's tag (the host
// polls it) -- else in the device vector a collective, or the caller, goes on from.
static bool fin_in_launch(const icp_ctx* c, const icp::NNPlan& pl)
{
    return c->fused_tail && pl.sparse && pl.version == 2 && pl.row != 64 && icp::nn_can_fuse_tail(pl) && pl.blocks_x > c->host_rows_max &&
           c->fin_tickets.p != nullptr && c->fin_scratch.p != nullptr && c->h_final != nullptr;
}
static bool fin_to_host(const icp_ctx* c) { return !c->comm && c->mom_dev == (double*)c->mom_own.p; }
static void fill_fin(const icp_ctx* c, icp::NNTailArgs& ta)
{
    ta.rows = (double*)c->mom_partials.p;
    ta.rows_on_device = 1;
    ta.compact = 0;
    ta.fin_tickets = (unsigned int*)c->fin_tickets.p;
    ta.fin_scratch = (double*)c->fin_scratch.p;
    ta.fin_host = fin_to_host(c) ? 1 : 0;
    ta.fin_out = ta.fin_host ? c->h_final : c->mom_dev;
}

// rows the host itself adds up (single GPU, or ranks meeting in host memory) leave the sparse point-to-point kernels in
// the compact two-cache-line form (icp_kernels.h, NNTailArgs)
static bool use_compact_rows(const icp_ctx* c, const icp::NNPlan& pl, int metric, const double* rows)
{
    // (fp32 only: the compact row spends the last 16 mantissa bits of the error share on its tag -- 2^-36 of a sum of squares
    // of floats is nothing, but the fp64 path is held to 1e-12 against src/ICP_CPU.c's arithmetic)
    return c->prec == ICP_F32 && pl.sparse && metric == ICP_POINT_TO_POINT && rows == c->h_mom_partials;
}

// Completion tags are consecutive integers.  A compact row shows only the low NN_CROW_TAG_BITS bits of its tag, and a
// wiped row shows zero: no tag that is ever waited for may have those bits all zero.  Returns the first of `count`
// consecutive tags that are safe in that sense and reserves them.
static uint64_t take_tags(icp_ctx* c, uint64_t count)
{
    constexpr uint64_t kMod = 1ull << icp::NN_CROW_TAG_BITS;
    uint64_t first = c->tag_seq + 1;
    if (first % kMod == 0 || first / kMod != (first + count - 1) / kMod) first = (first / kMod + 1) * kMod + 1;   // (count << kMod)
    c->tag_seq = first + count - 1;
    return first;
}

// The two row formats keep their completion tags in different places of the same pinned buffer: when the format changes
// (another metric, a communicator attached or removed -- never inside a loop) the buffer is wiped, so that no sum left by
// the other format can ever be mistaken for a tag.
static void prepare_rows_format(icp_ctx* c, bool compact)
{
    const int want = compact ? 1 : 0;
    if (c->rows_format == want || !c->h_mom_partials) { c->rows_format = want; return; }
    std::memset(c->h_mom_partials, 0, c->rows_cap * ICP_NMOM * sizeof(double));
    bar_fence();
    c->rows_format = want;
}

// ordered rows: sort the rows by the hits of the launch before (and zero the counters) -- enqueued right before a pass of the loop
static int prepare_row_order(icp_ctx* c)
{
    if (!c->plan.order || c->row_hits.p == nullptr) { c->row_order = nullptr; return ICP_OK; }
    icp::RowOrderBuffers b{};
    for (int k = 0; k < 2; ++k) { b.keys[k] = (unsigned int*)c->order_keys[k].p; b.vals[k] = (int32_t*)c->order_vals[k].p; }
    b.temp = c->order_tmp.p;
    b.temp_bytes = c->order_tmp.cap;
    b.roles = (int32_t*)c->order_roles.p;
    b.totals = (unsigned long long*)c->order_totals.p;
    b.seq = c->order_seq++;
    c->order_launches++;
    // (8192 hits for a 16-wave block, and in proportion for smaller ones; the target itself: a quarter of a block slot's mean load)
    const int nw = c->plan.nw > 0 ? c->plan.nw : 16;
    b.min_part = c->split_min >= 0 ? c->split_min : 512 * nw;
    b.total_div = 4 * c->num_cus * (16 / nw);
    HIP_TRY(icp::launch_row_order(b, (unsigned int*)c->row_hits.p, c->plan.blocks_x, &c->row_order, c->stream));
    return ICP_OK;
}

static icp::NNCullInputs make_cull(const icp_ctx* c, const int32_t* seed)
{
    if (c->prec == ICP_F64) {   // (fp64: no sorted views)
        icp::NNCullInputs o{c->have_scan_copy ? c->Qs.p : nullptr, seed, c->use_boxes ? c->Qbox.p : nullptr, c->use_boxes ? c->Qsamp.p : nullptr};
        o.tune = &c->tune;
        if (c->count_work) o.work = (unsigned long long*)c->work.p;
        return o;
    }
    icp::NNCullInputs o{c->have_scan_copy ? c->Qs.p : nullptr, seed, c->use_boxes ? c->Qbox.p : nullptr, c->use_boxes ? c->Qsamp.p : nullptr};
    if (c->have_scan_copy && c->model_sorted) { o.Q_scan_sorted = c->Qss.p; o.q_perm = (const int32_t*)c->Qperm.p; }
    o.tune = &c->tune;
    o.waves64 = c->exclusive ? 16 : 0;
    if (c->moving_sorted) o.p_perm = (const int32_t*)c->Pperm.p;
    if (c->count_work) o.work = (unsigned long long*)c->work.p;
    if (c->plan.order && c->row_order != nullptr) { o.row_order = c->row_order; o.row_hits = (unsigned int*)c->row_hits.p; o.order_history = c->order_regs > 0; }
    if (c->have_records && c->use_boxes && c->plan.hier) o.records = (const float*)c->Qrec.p;
    if (c->plan.share_blocks > 0 && c->share_counts.p != nullptr) { o.share_counts = (unsigned int*)c->share_counts.p; o.share_seq = &c->share_seq; o.share_cold_seq = &c->share_cold_seq; o.seed_pub = (float*)c->seed_pub.p; }
    return o;
}

// icp_reset_moving is lazy: whoever needs the moving cloud in c->P asks for it here (the resident kernel does not --
// it reads the pristine copy directly and writes c->P itself, which saves a device-to-device copy and a dependent
// dispatch per registration)
static int materialize_moving(icp_ctx* c)
{
    if (c->moving_is_pristine && c->n > 0) {
        const size_t bytes = 3 * (size_t)icp::pad_moving(c->n) * icp::elem_size(c->prec);
        HIP_TRY(hipMemcpyAsync(c->P.p, c->P0.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    c->moving_is_pristine = false;
    return ICP_OK;
}

int icp_get_moving(icp_ctx* c, void* out)
{
    if (int rc = use(c)) return rc;
    if (int rc = materialize_moving(c)) return rc;
    if (!c->have_moving) return fail(ICP_ERR_STATE, "no moving cloud resident");
    if (c->n == 0) return ICP_OK;
    if (!out) return fail(ICP_ERR_INVALID, "out == NULL");
    const size_t bytes = 3 * (size_t)c->n * icp::elem_size(c->prec);
    HIP_TRY(c->stage.ensure(bytes));
    HIP_TRY(icp::launch_soa_to_aos(c->prec, c->P.p, c->n, icp::pad_moving(c->n), c->stage.p, c->stream));
    HIP_TRY(hipMemcpyAsync(out, c->stage.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ICP_OK;
}

static int download_idx(icp_ctx* c, int which, int32_t* out)
{
    if (c->n == 0) return ICP_OK;
    if (!out) return fail(ICP_ERR_INVALID, "idx_out == NULL");
    if (!c->idx[which].p) return fail(ICP_ERR_STATE, "no matching pass has run");
    HIP_TRY(hipMemcpyAsync(out, c->idx[which].p, (size_t)c->n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ICP_OK;
}

int icp_get_indices(icp_ctx* c, int32_t* out)
{
    if (int rc = use(c)) return rc;
    return download_idx(c, c->cur, out);
}

static int require_clouds(icp_ctx* c)
{
    if (!c->have_model || !c->have_moving) return fail(ICP_ERR_STATE, "model and moving clouds must be resident");
    if (c->n > 0 && c->m == 0) return fail(ICP_ERR_EMPTY, "empty model cloud");
    return ICP_OK;
}

int icp_nn_match_resident(icp_ctx* c, float* kernel_ms)
{
    if (int rc = use(c)) return rc;
    if (int rc = require_clouds(c)) return rc;
    if (int rc = ensure_work_buffers(c)) return rc;
    if (int rc = materialize_moving(c)) return rc;
    if (kernel_ms) HIP_TRY(hipEventRecord(c->ev0, c->stream));
    const icp::NNCullInputs cull = make_cull(c, nullptr);
    HIP_TRY(icp::launch_nn(c->plan, c->P.p, c->Q.p, c->part_d.p, (int32_t*)c->part_idx.p, nullptr, &cull, nullptr, c->stream));
    if (kernel_ms) HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(icp::launch_merge(c->plan, c->part_d.p, (const int32_t*)c->part_idx.p, (int32_t*)c->idx[c->cur].p, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->idx_valid = true;
    if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, c->ev0, c->ev1));
    return ICP_OK;
}

int icp_nn_match_bench(icp_ctx* c, int reps, float* total_ms) { return icp_nn_match_bench_ex(c, reps, 1, total_ms); }

int icp_nn_match_bench_ex(icp_ctx* c, int reps, int seeded, float* total_ms)
{
    if (int rc = use(c)) return rc;
    if (int rc = require_clouds(c)) return rc;
    if (reps <= 0 || !total_ms) return fail(ICP_ERR_INVALID, "reps/total_ms");
    if (int rc = ensure_work_buffers(c)) return rc;
    if (int rc = materialize_moving(c)) return rc;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    // seeded with the most recent correspondences when there are any: this is how the loop launches it
    const icp::NNCullInputs cull = make_cull(c, (seeded && c->idx_valid) ? (const int32_t*)c->idx[c->cur].p : nullptr);
    for (int r = 0; r < reps; ++r)
        HIP_TRY(icp::launch_nn(c->plan, c->P.p, c->Q.p, c->part_d.p, (int32_t*)c->part_idx.p, nullptr, &cull, nullptr, c->stream));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipEventElapsedTime(total_ms, c->ev0, c->ev1));
    return ICP_OK;
}

// the reference's methodology (src/CUDA/Matching_opt.cu:213-226): events around EVERY launch, warm-ups first, the caller
// takes the minimum (and the mean) of the `reps` durations
int icp_nn_match_bench_launches(icp_ctx* c, int reps, int warmups, int mode, float* each_ms)
{
    if (int rc = use(c)) return rc;
    if (int rc = require_clouds(c)) return rc;
    if (reps <= 0 || warmups < 0 || !each_ms || mode < 0 || mode > 2) return fail(ICP_ERR_INVALID, "reps/warmups/mode/each_ms");
    if (c->loop.pending) return fail(ICP_ERR_STATE, "an enqueue is in flight");
    if (int rc = ensure_work_buffers(c)) return rc;
    if (int rc = materialize_moving(c)) return rc;
    icp::NNPlan pl = c->plan;
    const bool dense = mode == 2;
    if (dense) {
        if (c->prec != ICP_F32) return fail(ICP_ERR_INVALID, "the dense packed kernel is fp32");
        pl = icp::nn_plan(c->n, c->m, c->prec, c->num_cus, c->tune, 1);
        const size_t S = pl.splits > 0 ? (size_t)pl.splits : 1;
        HIP_TRY(c->part_d.ensure(S * (size_t)pl.n_pad * sizeof(float)));
        HIP_TRY(c->part_idx.ensure(S * (size_t)pl.n_pad * sizeof(int32_t)));
    }
    const icp::NNCullInputs cull = make_cull(c, (mode == 0 && c->idx_valid) ? (const int32_t*)c->idx[c->cur].p : nullptr);
    for (int r = -warmups; r < reps; ++r) {
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
        HIP_TRY(icp::launch_nn(pl, c->P.p, c->Q.p, c->part_d.p, (int32_t*)c->part_idx.p, nullptr, dense ? nullptr : &cull, nullptr, c->stream));
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        HIP_TRY(hipEventSynchronize(c->ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        if (r >= 0) each_ms[r] = ms;
    }
    if (dense) {   // the partial buffers may have grown: nothing else depends on the dense plan
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return ICP_OK;
}

int icp_share_rows_plan(const uint32_t* hits, int rows, int blocks, int model_points, int min_hits, int32_t* parts, uint32_t* target)
{
    if (!hits || !parts || rows < 1 || blocks < rows || model_points < 1 || min_hits < 1) return fail(ICP_ERR_INVALID, "icp_share_rows_plan: bad arguments");
    const unsigned int T = icp::share_rows_plan(hits, rows, blocks, icp::pad_model(model_points), min_hits, parts);
    if (target) *target = T;
    return ICP_OK;
}

int icp_diag_row_roles(icp_ctx* c, uint32_t* hits_io, int rows, int min_part, int total_div, int control, int32_t* roles_out)
{
    if (int rc = use(c)) return rc;
    if (!hits_io || !roles_out || rows < 1 || rows >= (1 << icp::NN_ROLE_ROW_BITS)) return fail(ICP_ERR_INVALID, "icp_diag_row_roles: bad arguments");
    static_assert(ICP_ROLES_EXTRA == icp::NN_ORDER_EXTRA, "the header's constant is the kernels'");
    DevBuf hits, keys[2], vals[2], tmp, roles, totals;
    const size_t rb = (size_t)rows * sizeof(unsigned int);
    auto body = [&]() -> int {
        HIP_TRY(hits.ensure(rb));
        for (int k = 0; k < 2; ++k) { HIP_TRY(keys[k].ensure(rb)); HIP_TRY(vals[k].ensure(rb)); }
        HIP_TRY(tmp.ensure(icp::row_order_temp_bytes(rows)));
        HIP_TRY(roles.ensure(((size_t)rows + icp::NN_ORDER_EXTRA) * sizeof(int32_t)));
        HIP_TRY(totals.ensure(2 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(totals.p, 0, totals.cap, c->stream));
        HIP_TRY(hipMemcpyAsync(hits.p, hits_io, rb, hipMemcpyHostToDevice, c->stream));
        icp::RowOrderBuffers b{};
        for (int k = 0; k < 2; ++k) { b.keys[k] = (unsigned int*)keys[k].p; b.vals[k] = (int32_t*)vals[k].p; }
        b.temp = tmp.p; b.temp_bytes = tmp.cap; b.roles = (int32_t*)roles.p; b.totals = (unsigned long long*)totals.p;
        b.seq = 0; b.min_part = min_part; b.total_div = total_div; b.control = control;
        const int32_t* out = nullptr;
        HIP_TRY(icp::launch_row_order(b, (unsigned int*)hits.p, rows, &out, c->stream));
        HIP_TRY(hipMemcpyAsync(roles_out, out, ((size_t)rows + icp::NN_ORDER_EXTRA) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(hits_io, hits.p, rb, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return ICP_OK;
    };
    const int rc = body();
    DevBuf* all[] = {&hits, &keys[0], &keys[1], &vals[0], &vals[1], &tmp, &roles, &totals};
    for (DevBuf* d : all) d->release();
    return rc;
}

int icp_nn_launch_info_ex(icp_ctx* c, int dense, int* splits, int* blocks, int* threads, int* n_pad, int* m_pad)
{
    if (!c) return fail(ICP_ERR_INVALID, "null context");
    const icp::NNPlan pl = icp::nn_plan(c->n, c->m, c->prec, c->num_cus, c->tune, dense ? 1 : 0);
    if (splits) *splits = pl.splits;
    if (blocks) *blocks = pl.blocks_x * pl.splits;
    if (threads) *threads = icp::nn_block_threads(pl);
    if (n_pad) *n_pad = pl.n_pad;
    if (m_pad) *m_pad = pl.m_pad;
    return ICP_OK;
}

int icp_nn_launch_info(icp_ctx* c, int* splits, int* blocks, int* threads, int* n_pad, int* m_pad)
{
    if (!c) return fail(ICP_ERR_INVALID, "null context");
    // (the geometry of the resident clouds, also before their first launch has fixed the plan)
    const icp::NNPlan pl = (c->plan.n == c->n && c->plan.m == c->m && c->plan.precision == c->prec) ? c->plan
                                                                                                       : icp::nn_plan(c->n, c->m, c->prec, c->num_cus, c->tune);
    if (splits) *splits = pl.splits;
    if (blocks) *blocks = pl.blocks_x * pl.splits;
    if (threads) *threads = icp::nn_block_threads(pl);
    if (n_pad) *n_pad = pl.n_pad;
    if (m_pad) *m_pad = pl.m_pad;
    return ICP_OK;
}

static int nn_match_host(icp_ctx* c, const void* P, int n, const void* Q, int m, int precision, int32_t* idx)
{
    if (int rc = use(c)) return rc;
    if (n < 0 || m < 0) return fail(ICP_ERR_INVALID, "negative size");
    if (n == 0) return ICP_OK;
    if (m == 0) return fail(ICP_ERR_EMPTY, "empty model cloud");
    if (!P || !Q || !idx) return fail(ICP_ERR_INVALID, "null pointer");
    if (int rc = icp_set_model(c, Q, m, precision)) return rc;
    if (int rc = icp_set_moving(c, P, n, precision)) return rc;
    if (int rc = icp_nn_match_resident(c, nullptr)) return rc;
    return download_idx(c, c->cur, idx);
}

int icp_nn_match_f32(icp_ctx* c, const float* P, int n, const float* Q, int m, int32_t* idx)
{
    return nn_match_host(c, P, n, Q, m, ICP_F32, idx);
}

int icp_nn_match_f64(icp_ctx* c, const double* P, int n, const double* Q, int m, int32_t* idx)
{
    return nn_match_host(c, P, n, Q, m, ICP_F64, idx);
}

// ---- normals -----------------------------------------------------------------------------------
int icp_estimate_normals(icp_ctx* c, void* nxyz_out, int32_t* nbr_out)
{
    if (int rc = use(c)) return rc;
    if (!c->have_model) return fail(ICP_ERR_STATE, "no model resident");
    const int m = c->m;
    if (m == 0) return fail(ICP_ERR_EMPTY, "empty model cloud");
    if (m < 5) return fail(ICP_ERR_INVALID, "normals need at least 5 model points (k = 4 neighbours + self)");
    icp::NNPlan pl = icp::nn_plan(m, m, c->prec, c->num_cus, c->tune);
    HIP_TRY(c->nbr.ensure((size_t)m * 4 * sizeof(int32_t)));
    const size_t es = icp::elem_size(c->prec);
    HIP_TRY(c->Nrm.ensure(3 * (size_t)pl.m_pad * es));
    if (c->prec == ICP_F32) {
        int n_pad, bx, S, seg;
        icp::knn4_v2_geometry(m, c->num_cus, &n_pad, &bx, &S, &seg);
        // the per-segment top-5 lists reuse the matching partial buffers
        HIP_TRY(c->part_d.ensure((size_t)S * n_pad * 5 * sizeof(float)));
        HIP_TRY(c->part_idx.ensure((size_t)S * n_pad * 5 * sizeof(int32_t)));
        HIP_TRY(icp::launch_knn4_v2(c->Q.p, m, c->num_cus, (float*)c->part_d.p, (int32_t*)c->part_idx.p, (int32_t*)c->nbr.p,
                                    c->stream));
    } else {
        HIP_TRY(icp::launch_knn4(pl, c->Q.p, (int32_t*)c->nbr.p, c->stream));
    }
    // covariance + eigen-solve on the device, straight into the resident (padded SoA) normal cloud
    HIP_TRY(icp::launch_normals(c->prec, c->Q.p, m, pl.m_pad, (const int32_t*)c->nbr.p, c->Nrm.p, c->stream));
    if (nxyz_out) {
        HIP_TRY(c->stage.ensure(3 * (size_t)m * es));
        HIP_TRY(icp::launch_soa_to_aos(c->prec, c->Nrm.p, m, pl.m_pad, c->stage.p, c->stream));
        HIP_TRY(hipMemcpyAsync(nxyz_out, c->stage.p, 3 * (size_t)m * es, hipMemcpyDeviceToHost, c->stream));
    }
    if (nbr_out)
        HIP_TRY(hipMemcpyAsync(nbr_out, c->nbr.p, (size_t)m * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_normals = true;
    return ICP_OK;
}

// ---- the loop ----------------------------------------------------------------------------------
int icp_loop_begin(icp_ctx* c, const icp_params* prm)
{
    if (int rc = use(c)) return rc;
    if (!prm) return fail(ICP_ERR_INVALID, "params == NULL");
    if (int rc = require_clouds(c)) return rc;
    if (prm->max_iter < 1) return fail(ICP_ERR_INVALID, "max_iter must be >= 1");
    if (prm->metric != ICP_POINT_TO_POINT && prm->metric != ICP_POINT_TO_PLANE) return fail(ICP_ERR_INVALID, "unknown metric");
    if (prm->precision != c->prec) return fail(ICP_ERR_INVALID, "params precision differs from the resident clouds");
    if (prm->metric == ICP_POINT_TO_PLANE && !c->have_normals) return fail(ICP_ERR_STATE, "point-to-plane needs model normals");
    if (c->n == 0) return fail(ICP_ERR_INVALID, "empty moving cloud");
    if (int rc = ensure_work_buffers(c)) return rc;
    LoopState& L = c->loop;
    L = LoopState();
    if (c->order_launches > 0) { c->order_regs++; c->order_launches = 0; }   // (the loop before left its rows' counters: history for this one's cold pass)
    // (the tickets of the in-launch finalize are zero between launches; a loop that was abandoned in mid-pass may have left some drawn)
    if (c->fin_tickets.p != nullptr && fin_in_launch(c, c->plan)) HIP_TRY(hipMemsetAsync(c->fin_tickets.p, 0, c->fin_tickets.cap, c->stream));
    if (int rc = L.H.begin(*prm)) return fail(rc, "bad loop parameters");
    L.active = true;
    L.from_pristine = c->moving_untouched;
    c->moving_untouched = false;   // (a loop moves the cloud)
    return ICP_OK;
}

static int loop_enqueue_body(icp_ctx* c);

int icp_loop_enqueue(icp_ctx* c)
{
    if (int rc = use(c)) return rc;
    return loop_enqueue_body(c);
}

static int loop_enqueue_body(icp_ctx* c)
{
    LoopState& L = c->loop;
    if (!L.active || L.H.done || L.pending) return fail(ICP_ERR_STATE, "enqueue: loop not ready");
    if (int rc = materialize_moving(c)) return rc;
    const auto tr0 = std::chrono::steady_clock::now();
    const icp::NNPlan& pl = c->plan;
    L.err_blocks = 0;
    L.mom_blocks = 0;
    L.rows_have_err = false;
    L.rows_compact = false;
    const bool host_reduce = c->host_reduce();
    double* mom_rows = host_reduce ? c->h_mom_partials : (double*)c->mom_partials.p;
    double* err_rows = (double*)c->err_partials.p;  // device: the moments kernel folds them into its rows
    const bool apply = L.H.have_rt;
    const bool final_only = L.H.next_is_final();  // the loop ends after this error whatever it is
    // the transform of the previous pass rides in the front of the matching kernel when that kernel
    // supports it; otherwise (fp64, or nothing left to match) it is its own launch
    const bool fused = apply && !final_only && icp::nn_can_fuse_transform(pl);
    if (apply) {
        if (!fused)  // with nothing left to match, the last pass's rows go straight to the host
            HIP_TRY(icp::launch_transform_error(c->prec, c->P.p, c->n, pl.n_pad, L.H.R, L.H.t, c->Q.p, pl.m_pad,
                                                (const int32_t*)c->idx[c->cur].p,
                                                (final_only && host_reduce) ? c->h_err_partials : err_rows,
                                                &L.err_blocks, c->stream));
        L.applied_idx = c->cur;
        L.H.note_applied();
    }
    L.timed_nn = false;
    L.final_poll = false;
    bool slots_written = false;
    bool fin = false;   // this pass's rows are added up inside its launch
    if (!final_only) {
        // the previous pass's matches seed the early-out bound (any valid index would do)
        if (c->fused_tail && icp::nn_can_fuse_tail(pl)) { if (int rc = prepare_row_order(c)) return rc; } else c->row_order = nullptr;
        const icp::NNCullInputs cull = make_cull(c, L.matched ? (const int32_t*)c->idx[c->cur].p : nullptr);
        c->cur ^= 1;
        L.matched = true;
        c->idx_valid = true;
        const bool time_this = c->profile_stride > 0 && (c->nn_launch_count++ % (uint64_t)c->profile_stride) == 0;
        if (time_this) { HIP_TRY(hipEventRecord(c->ev0, c->stream)); }
        // fused tail: the matching kernel itself merges the segments (atomic keys), stores idx and produces the
        // moment rows -- no partial arrays, no second launch.  ICP_FUSED_TAIL=0 keeps the two-kernel form.
        const bool tail = c->fused_tail && icp::nn_can_fuse_tail(pl);
        icp::NNTailArgs ta{};
        if (tail) {
            ta.metric = L.H.prm.metric;
            ta.keys = (unsigned long long*)c->keys.p;
            ta.tickets = (unsigned int*)c->tickets.p;
            ta.err_tile = (double*)c->err_partials.p;
            ta.idx_out = (int32_t*)c->idx[c->cur].p;
            ta.Nrm_soa = c->Nrm.p;
            ta.rows = mom_rows;
            ta.tag = (double)take_tags(c, 1);
            ta.compact = use_compact_rows(c, pl, ta.metric, mom_rows) ? 1 : 0;
            ta.rows_on_device = host_reduce ? 0 : 1;
            if (!host_reduce && fin_in_launch(c, pl)) { fill_fin(c, ta); fin = true; }
        }
        L.rows_compact = tail && ta.compact != 0;
        if (host_reduce) prepare_rows_format(c, L.rows_compact);   // (also the two-kernel form: launch_moments writes full rows)
        bool slots = false;
        if (fused) {
            icp::NNFusedTransform ft{L.H.R, L.H.t, (const int32_t*)c->idx[L.applied_idx].p, c->P2.p, err_rows};
            // every fused pass of the sparse kernels leaves its points and matches in slot order; the next one starts from them
            // (one level of coalesced loads instead of slot -> point -> seed -> model point), as the armed launches do
            if (tail && pl.sparse && pl.row != 64 && pl.splits == 1 && c->slot_state.ensure(9 * (size_t)pl.n_pad * sizeof(float)) == hipSuccess) {
                ft.slot_state = c->slot_state.p;
                ft.slot_valid = L.slot_written;
                ft.slot_flip = L.slot_flip;
                slots = true;
            }
            HIP_TRY(icp::launch_nn(pl, c->P.p, c->Q.p, c->part_d.p, (int32_t*)c->part_idx.p, &ft, &cull, tail ? &ta : nullptr, c->stream));
            std::swap(c->P, c->P2);  // the moved cloud is the current one from here on
            slots_written = slots;
            L.err_blocks = pl.blocks_x;
        } else {
            HIP_TRY(icp::launch_nn(pl, c->P.p, c->Q.p, c->part_d.p, (int32_t*)c->part_idx.p, nullptr, &cull, tail ? &ta : nullptr, c->stream));
        }
        if (time_this) { HIP_TRY(hipEventRecord(c->ev1, c->stream)); L.timed_nn = true; }
        if (tail) {
            L.mom_blocks = pl.blocks_x;   // one row per row of matching blocks, error share in slot 0
            L.err_blocks = 0;
            L.rows_have_err = true;
        } else {
            HIP_TRY(icp::launch_moments(pl, L.H.prm.metric, c->P.p, c->Q.p, c->Nrm.p, c->part_d.p,
                                        (const int32_t*)c->part_idx.p, (int32_t*)c->idx[c->cur].p, mom_rows,
                                        &L.mom_blocks, (double)take_tags(c, 1), err_rows, L.err_blocks, c->stream));
            if (host_reduce) L.err_blocks = 0;  // already inside the moment rows
        }
    }
    if (!host_reduce) {
        if (!fin) {
            if (L.mom_blocks > 2048) HIP_TRY(c->fin_scratch.ensure(256 * ICP_NMOM * sizeof(double)));
            HIP_TRY(icp::launch_finalize(c->mom_dev, (const double*)c->mom_partials.p, L.mom_blocks,
                                         (const double*)c->err_partials.p, L.err_blocks, L.rows_have_err ? 1 : 0, c->stream, (double*)c->fin_scratch.p));
        }
        if (c->comm) {  // the iteration's one collective: 32 doubles, in place, on the loop's stream
            std::string err;
            if (int rc = icp::comm_allreduce_sum_f64(c->comm, c->mom_dev, ICP_NMOM, c->stream, err)) return fail(rc, err);
        }
    }
    L.host_reduce = host_reduce;
    L.final_poll = fin && fin_to_host(c);
    L.wait_tag = (double)c->tag_seq;
    L.pending = true;
    L.slot_written = slots_written;   // (a fused pass of the sparse kernels left its points and matches in slot order)
    if (slots_written) L.slot_flip = !L.slot_flip;
    if (c->trace) c->tr_enqueue += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
    return ICP_OK;
}

void* icp_loop_moments_dev(icp_ctx* c) { return c ? (void*)c->mom_dev : nullptr; }

int icp_loop_set_moments_dev(icp_ctx* c, void* dev_ptr)
{
    if (int rc = use(c)) return rc;
    if (c->loop.pending) return fail(ICP_ERR_STATE, "an enqueue is in flight");
    if (!dev_ptr) {
        HIP_TRY(c->mom_own.ensure(ICP_NMOM * sizeof(double)));
        c->mom_dev = (double*)c->mom_own.p;
    } else {
        // (whoever owns the buffer reduces it across ranks; a library-side exchange on top would add the ranks up twice)
        if (c->lcomm) return fail(ICP_ERR_STATE, "a node communicator is attached (icp_comm_init_local): the library exchanges the vector itself");
        if (c->comm) return fail(ICP_ERR_STATE, "a device communicator is attached (icp_comm_init): the library all-reduces its own vector");
        c->mom_dev = (double*)dev_ptr;
    }
    return ICP_OK;
}

static int loop_complete_body(icp_ctx* c, int* done);

int icp_loop_complete(icp_ctx* c, int* done)
{
    if (int rc = use(c)) return rc;
    ScopedPin pin(c);
    return loop_complete_body(c, done);
}

// (icp_loop_run's forms call this once per pass: the device was selected and the thread placed when the call came in --
// a hipSetDevice and a sched_getcpu per pass are a measurable part of a 9 us iteration)
static int loop_complete_body(icp_ctx* c, int* done)
{
    LoopState& L = c->loop;
    if (!L.active || !L.pending) return fail(ICP_ERR_STATE, "complete without enqueue");
    // (clock reads cost ~25 ns apiece and there were seven per pass: the ones that only feed ICP_TRACE are taken when it is on)
    const bool tracing = c->trace || c->trace_passes;
    const auto tr0 = tracing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point{};
    auto tr1 = tr0;
    if (L.host_reduce) {
        // the kernels wrote their partial rows into mapped pinned memory.  Instead of a stream
        // synchronisation the host polls the per-row completion tags (each row is released to system
        // scope before its tag); the matching kernel's error rows were complete before the moments
        // kernel started.  Fixed block order => the same bits every run.
        // Rows are summed in block order AS their tags arrive, so the reduction overlaps the kernel's last blocks.
        double* mom = c->h_mom;
        const bool compact = L.rows_compact;
        const size_t stride = compact ? (size_t)icp::NN_CROW : (size_t)ICP_NMOM, tag_slot = compact ? 0 : ICP_NMOM - 1;
        constexpr unsigned long long kTagMask = (1ull << icp::NN_CROW_TAG_BITS) - 1ull;
        // the tag a row carries now: a double of its own (full rows), or the low mantissa bits of slot 0 (compact rows)
        // (compact rows: a tag in every 32-byte sector -- slots 0, 4, 8, 12; the row's tag is what all four agree on, else "none")
        auto row_tag = [&](int b) -> double {
            const volatile double* p = c->h_mom_partials + (size_t)b * stride + tag_slot;
            if (!compact) return *p;
            const volatile unsigned long long* q = reinterpret_cast<const volatile unsigned long long*>(p);
            const unsigned long long t0 = q[0] & kTagMask, t1 = q[4] & kTagMask, t2 = q[8] & kTagMask, t3 = q[12] & kTagMask;
            return (t0 == t1 && t0 == t2 && t0 == t3) ? (double)t0 : -1.0;
        };
        auto tag_value = [&](double tag) { return compact ? (double)((unsigned long long)tag & kTagMask) : tag; };
        auto start_sum = [&]() {
            for (int k = 0; k < ICP_NMOM; ++k) mom[k] = 0.0;
            for (int b = 0; b < L.err_blocks; ++b) mom[ICP_MOM_ERR] += c->h_err_partials[b];
            // (a compact row does not carry its point count: a row of the sparse kernels holds the real points of its slots)
            if (compact) mom[ICP_MOM_CNT] = (double)c->n;
        };
        auto add_row = [&](int b) {
            const double* row = c->h_mom_partials + (size_t)b * stride;
            if (compact) {   // {error share + tag, sum p, sum q, sum q p^T} -> slots ICP_MOM_SP .. ICP_MOM_SQP + 8, ICP_MOM_ERR
                auto untagged = [&](int k) {
                    unsigned long long bits;
                    std::memcpy(&bits, &row[k], sizeof bits);
                    if ((k & 3) == 0) bits &= ~kTagMask;   // (the first slot of every 32-byte sector carries the tag)
                    double v;
                    std::memcpy(&v, &bits, sizeof v);
                    return v;
                };
                for (int k = 1; k < icp::NN_CROW; ++k) mom[ICP_MOM_SP - 1 + k] += untagged(k);
                mom[ICP_MOM_ERR] += untagged(0);
            } else {
                for (int k = 0; k < ICP_NMOM - 1; ++k) mom[k] += row[k];  // the last slot is the completion tag
            }
        };
        bool polled = false;
        if (L.mom_blocks > 0 && !L.timed_nn && c->poll && L.err_blocks == 0) {
            const double want = tag_value(L.wait_tag);
            // (the poll's start: what the 2 s time-out counts from; the host got here right after posting the message, whose
            // time the resident loop has just read -- an armed or plain pass reads the clock itself)
            const auto t0 = L.live_mailbox != nullptr && !tracing ? c->posted_at : std::chrono::steady_clock::now();
            int b = 0;
            unsigned spins = 0;
            start_sum();
            static_assert(ICP_NMOM == 32, "add_full_rows_avx takes rows of 32 doubles");
            if (L.mom_blocks <= 1024) {
                // Compact rows: SWEEP over the rows whose tag is still missing -- the cache misses of different rows overlap,
                // where polling row b to completion before looking at row b + 1 takes them one after the other -- fetch a
                // row's second line as soon as its tag is seen, and add the rows up in block order once all are there
                // (tools/rows_probe.hip: 256 rows 6.6 -> 5.8 us; the order of the additions, and with it every bit of
                // the sums, is the same as before).
                // (round 3 tried a LIST of the rows still missing instead of the flags -- a sweep then costs what is missing, not the
                // row count: no difference on the hall loop, 8.99-9.07 against 8.92-9.04 us per iteration on one box; the tags are
                // compared as the integers they are)
                // Round 3: rows in the full format (point-to-plane, fp64) are swept and added the same way; their tag is a double
                // of its own in the row's last slot, compared by its bits.
                unsigned char seen[1024];
                std::memset(seen, 0, (size_t)L.mom_blocks);
                int left = L.mom_blocks;
                bool first = true;
                unsigned long long want_bits = (unsigned long long)want, tag_bits_mask = kTagMask;
                if (!compact) { std::memcpy(&want_bits, &want, sizeof want_bits); tag_bits_mask = ~0ull; }
                const volatile unsigned long long* tags = reinterpret_cast<const volatile unsigned long long*>(c->h_mom_partials) + tag_slot;
                while (left > 0) {
                    for (int r = 0; r < L.mom_blocks; ++r) {
                        if (seen[r] || (tags[(size_t)r * stride] & tag_bits_mask) != want_bits) continue;
                        // (compact rows: every 32-byte sector carries the tag; the row is there when all four do)
                        if (compact && ((tags[(size_t)r * stride + 4] & tag_bits_mask) != want_bits || (tags[(size_t)r * stride + 8] & tag_bits_mask) != want_bits ||
                                        (tags[(size_t)r * stride + 12] & tag_bits_mask) != want_bits)) continue;
                        seen[r] = 1;
                        --left;
                        if (compact) __builtin_prefetch(reinterpret_cast<const char*>(c->h_mom_partials + (size_t)r * stride) + 64);
                        else for (int l = 0; l < 3; ++l) __builtin_prefetch(reinterpret_cast<const char*>(c->h_mom_partials + (size_t)r * stride) + 64 * l);   // (the tag sits in the row's fourth line)
                        if (first && c->trace_passes) c->tr_first_row = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                        first = false;
                    }
                    if (left > 0 && (++spins & 0x3f) == 0 &&
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kRowPollS)
                        break;  // something is wrong (fault, hang): let the runtime report it
                }
                if (left == 0) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    static const bool have_avx = __builtin_cpu_supports("avx");
                    static_assert(icp::NN_CROW == 16, "add_compact_rows_avx takes rows of sixteen doubles");
                    if (have_avx && c->mail_wide && compact) {   // (ICP_MAILBOX_AVX=0 keeps the scalar loop: the same bits, for the A/B)
                        double sum[16];
                        add_compact_rows_avx(c->h_mom_partials, L.mom_blocks, kTagMask, sum);
                        mom[ICP_MOM_ERR] += sum[0];
                        for (int k = 1; k < icp::NN_CROW; ++k) mom[ICP_MOM_SP - 1 + k] += sum[k];
                        b = L.mom_blocks;
                    } else if (have_avx && c->mail_wide) {
                        double sum[32];
                        add_full_rows_avx(c->h_mom_partials, L.mom_blocks, sum);
                        for (int k = 0; k < ICP_NMOM - 1; ++k) mom[k] += sum[k];
                        b = L.mom_blocks;
                    } else
                    for (b = 0; b < L.mom_blocks; ++b) add_row(b);
                }
            } else
            while (b < L.mom_blocks) {
                if (row_tag(b) == want) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    if (c->trace_passes && b == 0) c->tr_first_row = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    add_row(b++);
                    continue;
                }
                if ((++spins & 0x3ff) == 0 &&
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kRowPollS)
                    break;  // something is wrong (fault, hang): let the runtime report it
            }
            polled = b == L.mom_blocks;
            c->rows_done_at = std::chrono::steady_clock::now();
            if (c->trace_passes) { c->tr_last_row = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); c->tr_rows_done = std::chrono::steady_clock::now(); }
            if (!polled && c->trace) {
                std::fprintf(stderr, "[icp trace]   poll gave up at row %d; rows still missing:", b);
                int shown = 0;
                for (int r = 0; r < L.mom_blocks && shown < 40; ++r)
                    if (row_tag(r) != want) { std::fprintf(stderr, " %d", r); ++shown; }
                std::fprintf(stderr, "\n");
            }
        }
        if (tracing) tr1 = std::chrono::steady_clock::now();
        if (!polled) {
            // a resident kernel would go on waiting for its next message: withdraw it (under the tag it will wait for)
            if (L.live_mailbox) {
                if (c->prec == ICP_F64) post_message64(L.live_mailbox, nullptr, nullptr, icp::ICP_CMD_EXIT, L.wait_tag + 1.0, c->mail_wide);
                else post_message(L.live_mailbox, nullptr, nullptr, icp::ICP_CMD_EXIT, L.wait_tag + 1.0, c->mail_wide);
            }
            HIP_TRY(hipStreamSynchronize(c->stream));
            tr1 = std::chrono::steady_clock::now();
            for (int b = 0; b < L.mom_blocks; ++b)
                if (row_tag(b) != tag_value(L.wait_tag)) {
                    L.pending = false;
                    char msg[240];
                    int have = 0;
                    for (int r = 0; r < L.mom_blocks; ++r) have += row_tag(r) == tag_value(L.wait_tag) ? 1 : 0;
                    std::snprintf(msg, sizeof msg, "a matching pass ended without producing its rows: row %d of %d carries tag %.0f, expected %.0f; %d rows arrived (armed / resident launch timed out?)",
                                  b, L.mom_blocks, row_tag(b), tag_value(L.wait_tag), have);
                    c->rows_timed_out = true;
                    return fail(ICP_ERR_HIP, msg);
                }
            start_sum();
            for (int b = 0; b < L.mom_blocks; ++b) add_row(b);
        }
    } else if (L.final_poll) {
        // the launch itself added its rows up and leaves the vector in pinned memory, the pass's tag in its last slot
        const volatile double* fin = c->h_final;
        bool there = false;
        if (!L.timed_nn && c->poll) {
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 1; !(there = fin[ICP_NMOM - 1] == L.wait_tag); ++spins)
                if ((spins & 0x3ff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kRowPollS) break;
            c->rows_done_at = std::chrono::steady_clock::now();
        }
        if (!there) {
            // (a timed pass is completed with a synchronisation; so is one whose tag never came: the runtime says what happened)
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (fin[ICP_NMOM - 1] != L.wait_tag) {
                L.pending = false;
                char msg[200];
                std::snprintf(msg, sizeof msg, "a matching pass ended without leaving its sums: the vector carries tag %.0f, expected %.0f (armed launch timed out?)",
                              fin[ICP_NMOM - 1], L.wait_tag);
                c->rows_timed_out = true;
                return fail(ICP_ERR_HIP, msg);
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int k = 0; k < ICP_NMOM - 1; ++k) c->h_mom[k] = fin[k];
        c->h_mom[ICP_NMOM - 1] = 0.0;
        tr1 = std::chrono::steady_clock::now();
    } else {
        HIP_TRY(hipMemcpyAsync(c->h_mom, c->mom_dev, ICP_NMOM * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        tr1 = std::chrono::steady_clock::now();
    }
    const auto tr2 = tracing ? std::chrono::steady_clock::now() : tr1;
    L.pending = false;
    if (L.timed_nn) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        L.seconds_nn += 1e-3 * ms;
        L.nn_launches += 1;
        c->prof_seconds_nn += 1e-3 * ms;
        c->prof_nn_launches += 1;
        c->prof_nn_passes += 1;
    }
    if (c->lcomm && c->mom_dev == (double*)c->mom_own.p) {  // the node's ranks exchange their sums (rank order: identical on every rank)
        std::string err;
        // (only the entries the metric uses travel: 19 doubles = 3 cache lines per slot instead of 5)
        const int used = L.H.prm.metric == ICP_POINT_TO_PLANE ? ICP_MOM_B + 6 : ICP_MOM_SQQ + 1;
        if (int rc = icp::lcomm_allreduce_sum_f64(c->lcomm, c->h_mom, used, err)) return fail(rc, err);
    }
    // (the host half of a pass is timed only while profiling is on: two clock reads are 0.5 % of a 9 us iteration)
    const bool time_host = c->profile_stride > 0;
    const auto th0 = time_host ? std::chrono::steady_clock::now() : tr2;
    const int adv = L.H.advance(c->h_mom);
    if (time_host) L.seconds_host += std::chrono::duration<double>(std::chrono::steady_clock::now() - th0).count();
    if (adv != ICP_OK) {
        if (done) *done = 1;
        L.numeric_failure = true;   // (the loop is over; what its completed passes produced stays readable: icp_loop_state)
        return fail(adv, "minimisation failed (degenerate correspondences)");
    }
    if (c->trace) {
        const auto tr3 = std::chrono::steady_clock::now();
        c->tr_wait += std::chrono::duration<double>(tr1 - tr0).count();
        c->tr_reduce += std::chrono::duration<double>(tr2 - tr1).count();
        c->tr_solve += std::chrono::duration<double>(tr3 - tr2).count();
        c->tr_n += 1;
    }
    L.steps += 1;
    if (done) *done = L.H.done ? 1 : 0;
    return ICP_OK;
}

// ---- armed launches ------------------------------------------------------------------------------
// icp_loop_run keeps one matching pass enqueued AHEAD of the (R, t) it will apply: the kernel is launched and
// dispatched while the previous pass still runs and the host still solves, waits on a mailbox in pinned memory
// and starts the moment the solution is published -- the launch + dispatch latency (~8 us of a ~23 us iteration
// on the hall cloud) leaves the critical path.  If the loop stops instead, the pass is withdrawn and exits
// without having touched anything.
namespace {

// A plan with shared rows (33-57 k moving points) runs armed launches: every launch deals its blocks anew, by the hits of the
// launch before.  A resident kernel can share its rows too -- its blocks keep, for the whole launch, the roles the counts at
// its start give them; whichever block closes a split row publishes the matches for the others -- and is what ICP_RESIDENT=2
// (from the first pass, by the counts of the registration before) and ICP_SHARE_RESIDENT_AFTER=n (after n armed passes) select.
// Measured on Bunny.csv, registrations repeated in one context: 32.9 us per iteration armed, 30.7 resident from the start,
// 33.2 switching after 6 passes; a context's FIRST registration has no counts and runs a resident launch unshared (late passes
// of 42 us instead of 24), which is why armed is the default.
bool share_wants_resident(const icp_ctx* c)
{
    if (!(c->plan.share_blocks > 0 && c->resident == 1 && !c->resident_refused)) return false;
    // Round 3 default: a context's FIRST registration of a geometry has no counts to deal the roles by -- it runs armed launches,
    // which adapt within one pass; from the second registration on the counts of the one before are there, and the whole
    // registration is ONE resident kernel with shared rows (Bunny.csv, registrations repeated in one context: 32.9 -> 30.7 us per
    // iteration, profiles/r2/r2_03_bunny_shared_rows.txt).  ICP_SHARE_AUTO=0: armed throughout, as in round 2.
    if (c->share_auto && c->share_cold_seq >= 1 && c->loop.H.applied == 0 && !c->loop.matched) return true;
    // Round 4: the FIRST registration does not stay armed to its end either -- its cold pass has left counts (share_cold_seq == 1),
    // and from its second pass on it is one resident kernel dealt by them: 33.1 -> 31.7 us per iteration for that one registration
    // (profiles/r4/r4_12_bunny_first_registration_anatomy.txt, ICP_SHARE_RESIDENT_AFTER=2; = 3, 4: the same).
    if (c->share_auto && c->share_resident_after < 0 && c->share_cold_seq == 1 && c->loop.matched && c->loop.H.applied >= 1) return true;
    return c->share_resident_after >= 0 && c->loop.H.applied + 1 >= c->share_resident_after;
}

bool can_arm(icp_ctx* c)
{
    const LoopState& L = c->loop;
    const icp::NNPlan& pl = c->plan;
    return c->arm && !c->shares_device && !share_wants_resident(c) && c->prec == ICP_F32 && c->h_mail && (c->relay || c->mail_in_bar) && c->poll && (c->host_reduce() || (fin_in_launch(c, pl) && fin_to_host(c))) && c->fused_tail && pl.sparse && icp::nn_can_fuse_tail(pl) &&
           icp::nn_can_fuse_transform(pl) && c->have_scan_copy && c->use_boxes && L.active && L.pending && !L.armed &&
           L.matched && !L.H.done && !L.H.have_rt &&
           !L.timed_nn &&  // a timed pass is completed with a stream synchronisation: nothing may wait behind it
           L.H.applied + 1 < L.H.prm.max_iter &&  // the pass after the pending one still matches (it is not the final, error-only one)
           !(c->profile_stride > 0 && (c->nn_launch_count % (uint64_t)c->profile_stride) == 0);  // timed launches stay plain
}

int loop_arm(icp_ctx* c)
{
    LoopState& L = c->loop;
    const icp::NNPlan& pl = c->plan;
    if (int rc = prepare_row_order(c)) return rc;
    const icp::NNCullInputs cull = make_cull(c, (const int32_t*)c->idx[c->cur].p);
    const int prev_cur = c->cur;
    const int slot = (int)(c->mail_seq++ % kMailSlots);
    icp::NNMailbox* mb = mail_slot(c->h_mail, slot);
    const double tag = (double)take_tags(c, 1);
    post_message(mb, nullptr, nullptr, icp::ICP_CMD_EXIT, 0.0);   // cleared: nothing to act on yet
    icp::NNTailArgs ta{};
    ta.metric = L.H.prm.metric;
    ta.keys = (unsigned long long*)c->keys.p;
    ta.tickets = (unsigned int*)c->tickets.p;
    ta.err_tile = (double*)c->err_partials.p;
    ta.idx_out = (int32_t*)c->idx[prev_cur ^ 1].p;
    ta.Nrm_soa = c->Nrm.p;
    ta.rows = c->h_mom_partials;
    ta.tag = tag;
    ta.compact = use_compact_rows(c, pl, ta.metric, ta.rows) ? 1 : 0;
    const bool fin = !c->host_reduce();   // (can_arm: then the rows are added up inside the launch, the vector comes back in pinned memory)
    if (fin) fill_fin(c, ta);
    L.armed_compact = ta.compact != 0;
    if (!fin) prepare_rows_format(c, L.armed_compact);
    icp::NNFusedTransform ft{nullptr, nullptr, (const int32_t*)c->idx[prev_cur].p, c->P2.p, (double*)c->err_partials.p, mb, c->mail_in_bar ? nullptr : c->relay, tag};
    // every armed pass leaves its points and matches in slot order; the next one starts from them (one level of
    // coalesced loads instead of slot -> point -> seed -> model point) if the pass before it was such a pass
    if (pl.splits == 1 && c->slot_state.ensure(9 * (size_t)pl.n_pad * sizeof(float)) == hipSuccess) {
        ft.slot_state = c->slot_state.p;
        ft.slot_valid = L.slot_written;
        ft.slot_flip = L.slot_flip;
    }
    if (c->profile_stride > 0) c->nn_launch_count++;
    HIP_TRY(icp::launch_nn(pl, c->P.p, c->Q.p, c->part_d.p, (int32_t*)c->part_idx.p, &ft, &cull, &ta, c->stream));
    std::swap(c->P, c->P2);
    c->cur = prev_cur ^ 1;
    L.armed = true;
    L.armed_at = std::chrono::steady_clock::now();
    L.armed_tag = tag;
    L.armed_slot = slot;
    L.armed_prev_cur = prev_cur;
    return ICP_OK;
}

// the solution is in: publish it to the waiting kernel, which becomes the pending pass
void loop_release_armed(icp_ctx* c)
{
    LoopState& L = c->loop;
    icp::NNMailbox* mb = mail_slot(c->h_mail, L.armed_slot);
    if (c->debug_lose_pass >= 0 && L.H.applied == c->debug_lose_pass) c->debug_lose_pass = -1;   // (test hook: this message is lost)
    else post_message(mb, L.H.R, L.H.t, icp::ICP_CMD_TRANSFORM_MATCH, L.armed_tag, c->mail_wide);
    L.applied_idx = L.armed_prev_cur;
    L.H.note_applied();
    L.mom_blocks = c->plan.blocks_x;
    L.err_blocks = 0;
    L.rows_have_err = true;
    L.rows_compact = L.armed_compact;
    L.host_reduce = c->host_reduce();
    L.final_poll = !L.host_reduce;
    L.timed_nn = false;
    L.wait_tag = L.armed_tag;
    L.pending = true;
    L.armed = false;
    L.slot_written = c->plan.splits == 1 && c->slot_state.p != nullptr;
    if (L.slot_written) L.slot_flip = !L.slot_flip;
}

// the loop ended (or failed): the waiting kernel exits without touching anything; undo the bookkeeping
void loop_withdraw_armed(icp_ctx* c)
{
    LoopState& L = c->loop;
    if (!L.armed) return;
    icp::NNMailbox* mb = mail_slot(c->h_mail, L.armed_slot);
    post_message(mb, nullptr, nullptr, icp::ICP_CMD_EXIT, L.armed_tag, c->mail_wide);
    std::swap(c->P, c->P2);
    c->cur = L.armed_prev_cur;
    L.armed = false;
}

}  // namespace

// ---- resident registration ---------------------------------------------------------------------------
// One launch (every block resident) carries the whole loop: the blocks keep their points in registers and their seeds in
// LDS, every pass is one mailbox message (command + R, t) and one set of rows coming back.  No launch, no
// dispatch and no kernel boundary between two passes; what is left of an iteration is the pass itself plus one
// host <-> device round trip (~2 us, tools/mailbox_probe.hip).  The host side is the step-wise loop unchanged:
// the same HostLoop decides, the same rows are reduced in the same order -- the results are bit-identical.
namespace {

bool can_reside(icp_ctx* c)
{
    const LoopState& L = c->loop;
    const icp::NNPlan& pl = c->plan;
    // (a plan with shared rows starts with armed launches, see share_wants_resident; ICP_RESIDENT=2: resident from the first pass)
    // (ranks of one node communicator that share a DEVICE never reside: each fits the machine alone, the two together need
    // not -- one rank's waiting blocks would hold the CUs the other's rows are waited for on, the circular wait of can_arm)
    // ICP_DEBUG=shared_resident (tests: two hall-sized ranks, 2 x 256 half-CU blocks, known to fit together) lifts it.
    const bool shared_ok = c->debug_shared_resident;
    return c->resident && (!c->shares_device || shared_ok) && (c->resident > 1 || pl.share_blocks == 0 || share_wants_resident(c)) && (c->prec == ICP_F32 || pl.version == 3) && !c->resident_refused && c->h_mail && (c->relay || c->mail_in_bar) && c->poll && c->host_reduce() && c->fused_tail && pl.sparse &&
           icp::nn_can_fuse_tail(pl) && c->have_scan_copy && c->use_boxes && L.active && !L.pending && !L.H.done;
}

// returns ICP_OK with *fell_back = true when the resident kernel could not be launched (nothing has been done)
int loop_run_resident(icp_ctx* c, int max_steps, int* k_io, int* d_io, bool* fell_back)
{
    LoopState& L = c->loop;
    icp::NNPlan rp = c->plan;   // the resident kernel closes every row inside its block: one segment
    rp.splits = 1;
    rp.seg_len = icp::round_up(rp.m_pad, 8);
    icp::NNMailbox* mb = mail_slot(c->h_mail, (int)(c->mail_seq++ % kMailSlots));
    const int pass_cap = L.H.prm.max_iter + 2;
    const double base = (double)take_tags(c, (uint64_t)pass_cap + 1);
    const bool f64 = c->prec == ICP_F64;
    auto post = [&](const double* R9, const double* t3, int cmd, double seq) {
        if (f64) post_message64(mb, R9, t3, cmd, seq, c->mail_wide);
        else post_message(mb, R9, t3, cmd, seq, c->mail_wide);
    };
    post(nullptr, nullptr, icp::ICP_CMD_EXIT, 0.0);   // cleared
    const int c0 = c->cur;
    const icp::NNCullInputs cull = make_cull(c, L.matched ? (const int32_t*)c->idx[c0].p : nullptr);
    icp::NNTailArgs ta{};
    ta.metric = L.H.prm.metric;
    ta.keys = (unsigned long long*)c->keys.p;
    ta.tickets = (unsigned int*)c->tickets.p;
    ta.err_tile = (double*)c->err_partials.p;
    ta.idx_out = (int32_t*)c->idx[c0 ^ 1].p;   // pass 0, 2, ... (the step-wise loop flips before it writes, too)
    ta.idx_out_odd = (int32_t*)c->idx[c0].p;
    ta.Nrm_soa = c->Nrm.p;
    ta.rows = c->h_mom_partials;
    ta.tag = 0.0;
    ta.compact = use_compact_rows(c, rp, ta.metric, ta.rows) ? 1 : 0;
    prepare_rows_format(c, ta.compact != 0);
    icp::NNFusedTransform ft{nullptr, nullptr, (const int32_t*)c->idx[c0].p, c->P.p /* in place */, (double*)c->err_partials.p, mb, c->mail_in_bar ? nullptr : c->relay, base, true};
    // icp_set_profiling(n): every n-th resident kernel is bracketed by events (read after it has ended)
    const bool time_this = c->profile_stride > 0 && (c->resident_launch_count++ % (uint64_t)c->profile_stride) == 0;
    if (time_this) HIP_TRY(hipEventRecord(c->ev0, c->stream));
    // after icp_reset_moving the kernel reads the pristine copy and (re)writes c->P itself -- no copy is enqueued
    const void* P_in = c->moving_is_pristine ? c->P0.p : c->P.p;
    ft.store_first = c->moving_is_pristine;
    // shared rows: several blocks read a row's points at kernel entry, one of them stores the moved points in pass 0 -- not into
    // the buffer a block that starts late is still reading: the cloud goes to the second buffer (as an armed launch does)
    const bool two_buffers = rp.share_blocks > 0 && !c->moving_is_pristine && c->P2.p != nullptr;
    if (two_buffers) { ft.P_out = c->P2.p; ft.store_first = true; }
    const hipError_t le = icp::launch_nn(rp, P_in, c->Q.p, c->part_d.p, (int32_t*)c->part_idx.p, &ft, &cull, &ta, c->stream);
    if (le != hipSuccess) {
        (void)hipGetLastError();
        c->resident_refused = true;   // does not fit the machine
        *fell_back = true;
        return ICP_OK;
    }
    c->moving_is_pristine = false;
    if (two_buffers) std::swap(c->P, c->P2);
    *fell_back = false;
    if (time_this) HIP_TRY(hipEventRecord(c->ev1, c->stream));
    if (c->trace_passes) std::fprintf(stderr, "[icp trace] resident launch: mailbox %p relay %p base %.0f\n", (void*)mb, (void*)c->relay, base);
    auto send = [&](int cmd, double seq) { post(L.H.R, L.H.t, cmd, seq); };   // (R, t: ignored by a plain MATCH)
    int k = *k_io, d = *d_io, sent = 0, matched = 0, rc = ICP_OK;
    bool alive = true;
    c->rows_done_at = std::chrono::steady_clock::now();   // (the launch: the kernel waits from now on at the earliest)
    L.live_mailbox = mb;
    while (!d && k < max_steps && sent < pass_cap) {
        if (c->debug_stall_pass >= 0 && L.H.applied == c->debug_stall_pass) {
            c->debug_stall_pass = -1;
            usleep((useconds_t)(c->debug_stall_s * 1e6));
        }
        const auto tr0 = std::chrono::steady_clock::now();
        if (std::chrono::duration<double>(tr0 - c->rows_done_at).count() > kMailLeaseS) {
            // this thread was away for too long (descheduled, or held up in the inter-rank exchange): the kernel may have
            // given up waiting.  EXIT is consistent whatever each block has decided; icp_loop_run launches a new kernel,
            // which resumes from the state the last complete pass left (P in place, idx ping-pong).
            if (c->trace) std::fprintf(stderr, "[icp trace] resident kernel withdrawn: the host was %.2f s late\n",
                                       std::chrono::duration<double>(tr0 - c->rows_done_at).count());
            break;
        }
        const bool apply = L.H.have_rt;
        const bool final_only = L.H.next_is_final();
        const int cmd = !apply ? icp::ICP_CMD_MATCH : (final_only ? icp::ICP_CMD_TRANSFORM_ONLY : icp::ICP_CMD_TRANSFORM_MATCH);
        if (apply) {
            L.applied_idx = c->cur;
            L.H.note_applied();
        }
        if (cmd != icp::ICP_CMD_TRANSFORM_ONLY) {
            c->cur ^= 1;
            L.matched = true;
            c->idx_valid = true;
            ++matched;
        }
        if (c->debug_lose_pass >= 0 && L.H.applied == c->debug_lose_pass + (apply ? 1 : 0)) c->debug_lose_pass = -1;   // (test hook: this message is lost)
        else send(cmd, base + (double)sent);
        if (c->trace_passes && sent > 0)
            std::fprintf(stderr, "[icp trace]   host turnaround (last row seen -> next message out): %.2f us\n",
                         1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - c->tr_rows_done).count());
        L.mom_blocks = rp.blocks_x;
        L.err_blocks = 0;
        L.rows_have_err = true;
        L.rows_compact = ta.compact != 0;
        L.host_reduce = true;
        L.timed_nn = false;
        L.wait_tag = base + (double)sent;
        L.pending = true;
        ++sent;
        if (c->trace) c->tr_enqueue += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
        c->posted_at = tr0;
        const auto tc0 = c->trace_passes ? std::chrono::steady_clock::now() : tr0;
        rc = loop_complete_body(c, &d);
        if (c->trace_passes)
            std::fprintf(stderr, "[icp trace] resident pass %d cmd %d: %.2f us from message to reduced rows + solve (row 0 after %.2f us, all rows after %.2f us)\n", sent - 1, cmd,
                         1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count(), 1e6 * c->tr_first_row, 1e6 * c->tr_last_row);
        if (rc != ICP_OK) {
            if (c->trace) std::fprintf(stderr, "[icp trace] resident pass %d failed: mailbox %p reads back tags %08x %08x cmd %d (sent tag %08x)\n",
                                       sent - 1, (void*)mb, *(volatile uint32_t*)&mb->w[icp::ICP_MB_TAG0], *(volatile uint32_t*)&mb->w[icp::ICP_MB_TAG1],
                                       (int)*(volatile uint32_t*)&mb->w[icp::ICP_MB_CMD], icp::mailbox_tag(base + (double)(sent - 1)));
            break;
        }
        ++k;
        if (cmd == icp::ICP_CMD_TRANSFORM_ONLY) { alive = false; break; }  // the kernel ends itself after that pass
    }
    if (alive) send(icp::ICP_CMD_EXIT, base + (double)sent);
    L.live_mailbox = nullptr;
    if (rc != ICP_OK && L.numeric_failure) {
        // the pass completed on the device and the MINIMISATION refused its sums: the kernel has been told to exit, the cloud
        // is in the state the last applied transform left, the loop's counters and error series stay readable
        (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
    } else if (rc != ICP_OK) {
        // a pass did not complete: blocks may have applied its transform to their part of the cloud and others not.
        // Nothing of that state is offered to the caller: the loop is over, the moving cloud goes back to what
        // icp_set_moving uploaded (materialised from the pristine copy on its next use), the matches are void.
        (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
        c->moving_is_pristine = true;
        c->idx_valid = false;
        L.active = false;
        L.pending = false;
        g_last_error += " [the loop was abandoned; the moving cloud is reset to its uploaded state]";
    }
    if (time_this && rc == ICP_OK) {
        float ms = 0.f;
        // the kernel ends within microseconds of the exit message: spin on the event instead of a blocking wait
        // (whose wake-up alone costs tens of microseconds of the loop being measured)
        const auto tq = std::chrono::steady_clock::now();
        hipError_t qe = hipErrorNotReady;
        while ((qe = hipEventQuery(c->ev1)) == hipErrorNotReady)
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - tq).count() > 5.0) break;
        if (qe != hipSuccess) { (void)hipGetLastError(); HIP_TRY(hipEventSynchronize(c->ev1)); }
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        L.seconds_nn += 1e-3 * ms;
        L.nn_launches += 1;
        c->prof_seconds_nn += 1e-3 * ms;
        c->prof_nn_launches += 1;
        c->prof_nn_passes += matched;
    }
    *k_io = k;
    *d_io = d;
    return rc;
}

}  // namespace

// A pass that never delivered its rows (a message that no block saw, blocks another process kept off the machine) ends the
// resident / armed conversation -- but not necessarily the registration: when the loop started from the uploaded cloud
// (icp_set_moving / icp_reset_moving) the copy is still there, and the same registration is run again with plain launches,
// one per pass, nothing resident and nothing armed.  Every loop form produces the same bits, so the caller gets what an
// undisturbed run would have returned; only when that fails too (a device that is really gone) does the error surface.
static int loop_run_inner(icp_ctx* c, int max_steps, int* k_out, int* d_out);

static int redo_stepwise(icp_ctx* c, const icp_params& prm, long long target_steps, int* d_out)
{
    if (hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return ICP_ERR_HIP; }
    const int keep_resident = c->resident;
    const bool keep_arm = c->arm;
    c->resident = 0;
    c->arm = false;
    c->moving_is_pristine = true;
    c->moving_untouched = true;
    int d = 0;
    int rc = icp_loop_begin(c, &prm);
    while (rc == ICP_OK && !d && c->loop.steps < target_steps) {
        rc = loop_enqueue_body(c);
        if (rc == ICP_OK) rc = loop_complete_body(c, &d);
    }
    c->resident = keep_resident;
    c->arm = keep_arm;
    *d_out = d;
    return rc;
}

int icp_loop_run(icp_ctx* c, int max_steps, int* steps_done, int* done)
{
    if (max_steps < 0) return fail(ICP_ERR_INVALID, "max_steps < 0");
    if (int rc = use(c)) return rc;
    ScopedPin pin(c);
    const bool can_redo = c->loop.active && c->loop.from_pristine && !c->comm && !c->lcomm;   // (ranks of a communicator must move together)
    const icp_params prm = c->loop.H.prm;
    const long long steps_before = c->loop.steps;
    c->rows_timed_out = false;
    // (should the registration have to be run again, the aborted attempt's share of the profiling and trace totals is taken back)
    const double keep_nn = c->prof_seconds_nn, keep_tr[4] = {c->tr_enqueue, c->tr_wait, c->tr_reduce, c->tr_solve};
    const int keep_launches = c->prof_nn_launches;
    const long long keep_passes = c->prof_nn_passes;
    const uint64_t keep_tr_n = c->tr_n;
    int k = 0, d = 0;
    int rc = loop_run_inner(c, max_steps, &k, &d);
    if (rc == ICP_ERR_HIP && c->rows_timed_out && can_redo) {
        const std::string first = g_last_error;
        c->prof_seconds_nn = keep_nn; c->prof_nn_launches = keep_launches; c->prof_nn_passes = keep_passes;
        c->tr_enqueue = keep_tr[0]; c->tr_wait = keep_tr[1]; c->tr_reduce = keep_tr[2]; c->tr_solve = keep_tr[3]; c->tr_n = keep_tr_n;
        if (c->trace) std::fprintf(stderr, "[icp trace] %s -- running the registration again step-wise\n", first.c_str());
        c->loop.active = false;
        c->loop.pending = false;
        c->idx_valid = false;
        rc = redo_stepwise(c, prm, steps_before + (long long)max_steps, &d);
        if (rc == ICP_OK) {
            c->recoveries += 1;
            k = (int)std::max<long long>(0, c->loop.steps - steps_before);
        } else {
            g_last_error = first + " [the step-wise re-run failed as well: " + g_last_error + "]";
        }
    }
    if (rc != ICP_OK) return rc;
    if (steps_done) *steps_done = k;
    if (done) *done = d;
    return ICP_OK;
}

int icp_recoveries(icp_ctx* c) { return c ? c->recoveries : ICP_ERR_INVALID; }

static int loop_run_inner(icp_ctx* c, int max_steps, int* k_out, int* d_out)
{
    int d = c->loop.active && c->loop.H.done ? 1 : 0, k = 0;
    while (!d && k < max_steps) {
        if (can_reside(c)) {
            bool fell_back = false;
            if (int rc = loop_run_resident(c, max_steps, &k, &d, &fell_back)) return rc;
            if (!fell_back) continue;
        }
        if (!c->loop.pending)
            if (int rc = loop_enqueue_body(c)) return rc;
        if (k + 1 < max_steps && can_arm(c))
            if (int rc = loop_arm(c)) return rc;
        if (int rc = loop_complete_body(c, &d)) {
            loop_withdraw_armed(c);
            (void)hipStreamSynchronize(c->stream);
            (void)hipGetLastError();
            if (!c->loop.numeric_failure) {
                // (as after a failed resident pass: nothing half-transformed is offered to the caller; a numeric failure -- the
                // minimisation refused the sums of a pass that completed -- leaves the loop's state readable instead)
                c->moving_is_pristine = true;
                c->idx_valid = false;
                c->loop.active = false;
                c->loop.pending = false;
            }
            return rc;
        }
        if (c->loop.armed) {
            // (a host that comes back too late may not publish any more: the waiting kernel may have given up -- it is
            // withdrawn, which is consistent either way, and the pass is launched afresh)
            if (c->debug_stall_pass >= 0 && c->loop.H.applied == c->debug_stall_pass) {
                c->debug_stall_pass = -1;
                usleep((useconds_t)(c->debug_stall_s * 1e6));
            }
            const bool late = std::chrono::duration<double>(std::chrono::steady_clock::now() - c->loop.armed_at).count() > kMailLeaseS;
            if (d || late) loop_withdraw_armed(c);
            else loop_release_armed(c);
        }
        ++k;
    }
    *k_out = k;
    *d_out = d;
    return ICP_OK;
}

int icp_loop_state(icp_ctx* c, int* iterations, int* passes, double* err, int err_cap, double* T16)
{
    if (!c) return fail(ICP_ERR_INVALID, "null context");
    const LoopState& L = c->loop;
    if (!L.active) return fail(ICP_ERR_STATE, "no loop");
    if (iterations) *iterations = L.H.iterations;
    if (passes) *passes = L.H.applied;
    if (err) {
        const int cnt = (int)L.H.err.size() < err_cap ? (int)L.H.err.size() : err_cap;
        for (int i = 0; i < cnt; ++i) err[i] = L.H.err[i];
    }
    if (T16) std::memcpy(T16, L.H.T, sizeof L.H.T);
    return ICP_OK;
}

int icp_loop_timing(icp_ctx* c, double* seconds_nn, int* nn_launches)
{
    if (!c) return fail(ICP_ERR_INVALID, "null context");
    if (seconds_nn) *seconds_nn = c->prof_seconds_nn;
    if (nn_launches) *nn_launches = c->prof_nn_launches;
    return ICP_OK;
}

int icp_loop_phase_seconds(icp_ctx* c, double* seconds_nn, double* seconds_host)
{
    if (!c) return fail(ICP_ERR_INVALID, "null context");
    if (seconds_nn) *seconds_nn = c->loop.seconds_nn;
    if (seconds_host) *seconds_host = c->loop.seconds_host;
    return ICP_OK;
}

int icp_loop_timing_passes(icp_ctx* c, long long* passes)
{
    if (!c || !passes) return fail(ICP_ERR_INVALID, "null argument");
    *passes = c->prof_nn_passes;
    return ICP_OK;
}

int icp_loop_indices(icp_ctx* c, int32_t* out)
{
    if (int rc = use(c)) return rc;
    if (!c->loop.active) return fail(ICP_ERR_STATE, "no loop");
    return download_idx(c, c->loop.H.applied > 0 ? c->loop.applied_idx : c->cur, out);
}

static int run_loop(icp_ctx* c, const icp_params* prm, icp_result* out, double seconds_setup)
{
    ScopedPin pin(c);
    if (int rc = icp_loop_begin(c, prm)) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    int done = 0;
    while (!done)
        if (int rc = icp_loop_run(c, 1 << 20, nullptr, &done)) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    const LoopState& L = c->loop;
    if (out) {
        std::memcpy(out->T, L.H.T, sizeof L.H.T);
        out->iterations = L.H.iterations;
        out->passes = L.H.applied;
        out->seconds_total = std::chrono::duration<double>(t1 - t0).count();
        out->seconds_nn = L.seconds_nn;
        out->seconds_host = L.seconds_host;
        out->seconds_setup = seconds_setup;
        if (out->err)
            for (size_t i = 0; i < L.H.err.size(); ++i) out->err[i] = L.H.err[i];
        if (out->idx)
            if (int rc = icp_loop_indices(c, out->idx)) return rc;
        if (out->moved)
            if (int rc = icp_get_moving(c, out->moved)) return rc;
    }
    return ICP_OK;
}

int icp_point_to_point(icp_ctx* c, const void* data, int n, const void* model, int m, const icp_params* prm,
                       icp_result* out)
{
    if (int rc = use(c)) return rc;
    if (!prm) return fail(ICP_ERR_INVALID, "params == NULL");
    if (n <= 0) return fail(ICP_ERR_INVALID, "empty moving cloud");
    if (m <= 0) return fail(ICP_ERR_EMPTY, "empty model cloud");
    icp_params p = *prm;
    p.metric = ICP_POINT_TO_POINT;
    const auto s0 = std::chrono::steady_clock::now();
    if (int rc = icp_set_model(c, model, m, p.precision)) return rc;
    if (int rc = icp_set_moving(c, data, n, p.precision)) return rc;
    return run_loop(c, &p, out, std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count());
}

int icp_point_to_plane(icp_ctx* c, const void* data, int n, const void* model, int m, const void* normals,
                       const icp_params* prm, icp_result* out)
{
    if (int rc = use(c)) return rc;
    if (!prm) return fail(ICP_ERR_INVALID, "params == NULL");
    if (n <= 0) return fail(ICP_ERR_INVALID, "empty moving cloud");
    if (m <= 0) return fail(ICP_ERR_EMPTY, "empty model cloud");
    icp_params p = *prm;
    p.metric = ICP_POINT_TO_PLANE;
    const auto s0 = std::chrono::steady_clock::now();
    if (int rc = icp_set_model(c, model, m, p.precision)) return rc;
    if (normals) {
        if (int rc = icp_set_model_normals(c, normals, m)) return rc;
    } else {
        if (int rc = icp_estimate_normals(c, nullptr, nullptr)) return rc;
    }
    if (int rc = icp_set_moving(c, data, n, p.precision)) return rc;
    return run_loop(c, &p, out, std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count());
}

int icp_os1_packets_to_cartesian(icp_ctx* c, const uint8_t* packets, int n_packets, const float alt16[16],
                                 const float az16[16], float* xyz_out, uint32_t* ranges_out)
{
    if (int rc = use(c)) return rc;
    if (n_packets < 0 || (n_packets > 0 && (!packets || !xyz_out)) || !alt16 || !az16) return fail(ICP_ERR_INVALID, "bad arguments");
    if (n_packets == 0) return ICP_OK;
    const size_t n = (size_t)n_packets * 256, bytes = (size_t)n_packets * 12608;
    DevBuf d_pk, d_ang, d_r, d_xyz;
    auto body = [&]() -> int {
        HIP_TRY(d_pk.ensure(bytes));
        HIP_TRY(d_ang.ensure(32 * sizeof(float)));
        HIP_TRY(d_r.ensure(n * sizeof(uint32_t)));
        HIP_TRY(d_xyz.ensure(3 * n * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(d_pk.p, packets, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_ang.p, alt16, 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync((float*)d_ang.p + 16, az16, 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(icp::launch_os1_packets((const uint8_t*)d_pk.p, n_packets, (const float*)d_ang.p, (const float*)d_ang.p + 16,
                                        (uint32_t*)d_r.p, (float*)d_xyz.p, c->stream));
        HIP_TRY(hipMemcpyAsync(xyz_out, d_xyz.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        if (ranges_out) HIP_TRY(hipMemcpyAsync(ranges_out, d_r.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return ICP_OK;
    };
    const int rc = body();
    d_pk.release(); d_ang.release(); d_r.release(); d_xyz.release();
    return rc;
}

int icp_os1_to_cartesian(icp_ctx* c, const uint32_t* ranges, int n, uint32_t encoder0, const float alt16[16],
                         const float az16[16], float* xyz_out)
{
    if (int rc = use(c)) return rc;
    if (n < 0 || (n > 0 && (!ranges || !xyz_out)) || !alt16 || !az16) return fail(ICP_ERR_INVALID, "bad arguments");
    if (n == 0) return ICP_OK;
    DevBuf d_r, d_ang, d_xyz;
    int rc = ICP_OK;
    auto body = [&]() -> int {
        HIP_TRY(d_r.ensure((size_t)n * sizeof(uint32_t)));
        HIP_TRY(d_ang.ensure(32 * sizeof(float)));
        HIP_TRY(d_xyz.ensure(3 * (size_t)n * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(d_r.p, ranges, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_ang.p, alt16, 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync((float*)d_ang.p + 16, az16, 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(icp::launch_os1_conversion((const uint32_t*)d_r.p, n, encoder0, (const float*)d_ang.p,
                                           (const float*)d_ang.p + 16, (float*)d_xyz.p, c->stream));
        HIP_TRY(hipMemcpyAsync(xyz_out, d_xyz.p, 3 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return ICP_OK;
    };
    rc = body();
    d_r.release();
    d_ang.release();
    d_xyz.release();
    return rc;
}

}  // extern "C"
