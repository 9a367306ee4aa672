/*
 * icp_mi355x_diag.h -- measurement and diagnostic entry points of libicp_mi355x.so: what bench.py, the sweep program
 * and the tests use to time the matching kernel alone, to count the work it executes and to inspect launch geometry.
 * None of this is on the registration path and nothing here replaces a reference statement other than the reference's
 * own timing harness (src/CUDA/Matching_opt.cu:213-226).  Same conventions as icp_mi355x.h.
 */
#ifndef ICP_MI355X_DIAG_H
#define ICP_MI355X_DIAG_H

#include "icp_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* `reps` back-to-back launches of the matching kernel alone between two hipEvents on the context's
 * stream; total_ms / reps is the kernel's average launch duration (bench.py roofline leg) */
int icp_nn_match_bench(icp_ctx* ctx, int reps, float* total_ms);
/* same; seeded != 0 hands the kernel the most recent correspondences as its starting bound (what the ICP
 * loop does from its second pass on), seeded == 0 starts it cold (what icp_nn_match_* does) */
int icp_nn_match_bench_ex(icp_ctx* ctx, int reps, int seeded, float* total_ms);
/* The reference's own kernel-timing method (src/CUDA/Matching_opt.cu:213-226: cudaEventRecord around every launch,
 * minimum of 10 after warm-up): `warmups` untimed launches, then `reps` launches with a hipEvent pair around each one;
 * each_ms[r] receives the duration of launch r.  mode 0: the matching kernel as the loop launches it (seeded with the
 * most recent correspondences), 1: cold (no seed), 2: the dense packed kernel, which EXECUTES every one of the
 * n_pad x m_pad pairs (fp32 only; no boxes, no early-out) -- the brute-force scan the roofline arithmetic is about. */
int icp_nn_match_bench_launches(icp_ctx* ctx, int reps, int warmups, int mode, float* each_ms);
/* ... of the launch the resident clouds get from the production plan (dense == 0) or from the dense packed kernel */
int icp_nn_launch_info_ex(icp_ctx* ctx, int dense, int* splits, int* blocks, int* threads, int* n_pad, int* m_pad);
/* Executed-work accounting of the sparse matching kernel (it returns the brute-force answer of Matching<<<>>> without
 * evaluating most pairs, so the roofline of EXECUTED arithmetic needs a count).  enable != 0: every following sparse
 * launch of this context runs its instrumented instantiation and adds wave-level tallies to 8 device counters;
 * icp_get_work_counters reads them (uint64 x ICP_WORK_SLOTS) and optionally zeroes them.  Slots:
 *   0 chunk boxes tested against a block's group box (one lane each)      1 upper-level boxes (large models)
 *   2 hits = (wave, 8-point chunk) pairs through the per-point box test   3 ... through the xy half of the distances
 *   4 ... evaluated in full (each hit: 64 lanes x 2 moving points x 8 model points)
 *   5 cold-start sample groups scanned (128 x 8 pairs each)   6 (block, pass) pairs   7 ... that applied a transform
 * Timing with counting on is not representative (atomics, extra registers): count in a separate run. */
#define ICP_WORK_SLOTS 12 /* 8..11: speculative lists entered / that covered the pass / their hits / hits of ordinarily built lists */
int icp_set_work_counting(icp_ctx* ctx, int enable);
int icp_get_work_counters(icp_ctx* ctx, uint64_t* out_slots, int reset);

/* shared rows (clouds of 33-57 k moving points): how a matching launch of `blocks` blocks deals itself to `rows`
 * rows of 128 moving points, given the hit chunks every row listed in the launch before -- parts[r] blocks search row r
 * (>= 1 each, their sum <= blocks whatever the counts hold), *target = hits per block the split aims at.  The kernel computes
 * exactly this in every block; no device is involved here (no reference counterpart: its kernels are thread-per-point). */
int icp_share_rows_plan(const uint32_t* hits, int rows, int blocks, int model_points, int min_hits, int32_t* parts, uint32_t* target);
/* ordered + split rows (large clouds): the roles a launch's blocks get from the hit counters of the launch before,
 * computed ON THE DEVICE by the kernels the loop uses -- control != 0: the single-workgroup launch (rows <= 16 384: quantised
 * counting sort + roles), else keys + rocPRIM radix sort + roles.  roles_out: rows + ICP_ROLES_EXTRA entries, each
 * row | part << 21 | log2(parts) << 27, or -1 for a block nothing needs; hits_io is read AND zeroed, as by the loop.
 * min_part / total_div: the smallest part of a split row in hits, and what the counters' sum is divided by for the target. */
#define ICP_ROLES_EXTRA 4096
int icp_diag_row_roles(icp_ctx* ctx, uint32_t* hits_io, int rows, int min_part, int total_div, int control, int32_t* roles_out);

#ifdef __cplusplus
}
#endif
#endif /* ICP_MI355X_DIAG_H */
