/*
 * icp_mi355x.h -- C ABI of libicp_mi355x.so, the MI355X (gfx950) ICP registration hot path.
 *
 * The reference (Carlos310197/Fast-Point-Cloud-Registration-with-GPUs) has no library, plugin or
 * FFI interface: the path sits behind three main() programs (CMakeLists.txt:26, README.md:12).
 * The narrowest seams that exist are its kernel signatures and its driver loops; every entry
 * point below names the reference statement group it replaces (file:line relative to the
 * reference root).  Plain C: opaque handle, caller-owned buffers, int return codes, no
 * exceptions, no exit().  Nothing here takes or returns a torch type.
 *
 * Conventions
 *   - clouds cross the boundary in the reference GPU layout: AoS "xyzxyz..." (column-major 3xN,
 *     src/ICP_point_to_point.cu:139-152), float or double, HOST pointers unless a name says _dev.
 *   - correspondence semantics are those of the CPU path (src/ICP_CPU.c:227-232): squared
 *     distance (dx*dx + dy*dy) + dz*dz with every operation rounded separately (no FMA), the
 *     LOWEST model index wins ties, every idx[i] is always written.
 *   - rotation matrices are row-major 3x3 (R maps moving -> model), transforms row-major 4x4.
 *   - non-finite input -- a DELIBERATE DEVIATION from the reference, listed in INTEGRATION.md under "entry points that change
 *     behaviour": a cloud (or normal set) with a NaN or an infinite coordinate is REFUSED -- icp_set_model, icp_set_moving,
 *     icp_set_model_normals and everything built on them (icp_nn_match_*, icp_point_to_*) return ICP_ERR_INVALID and the
 *     context keeps no such cloud.  The reference does not check its input: its distances come from vdSub / vdSqr / vdAdd
 *     (src/ICP_CPU.c:227-231) and the match from cblas_idamin (:232), whose answer for a vector that holds NaN is whatever
 *     the BLAS at hand does (MKL documents none); whichever index comes back, the centroid sums (:342-366) then turn the
 *     whole transform into NaN.  There is nothing usable to reproduce, so the library says so at the door.
 *   - a context is bound to one HIP device; calls on one context are not re-entrant, distinct
 *     contexts are independent.  Every device entry point fails with ICP_ERR_NO_DEVICE when no
 *     gfx950 device is usable -- there is no CPU fallback.
 */
#ifndef ICP_MI355X_H
#define ICP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICP_ABI_VERSION 2 /* 2: icp_result gained seconds_host / seconds_setup, icp_loop_phase_seconds */

/* return codes */
#define ICP_OK 0
#define ICP_ERR_INVALID (-1)   /* bad argument (NULL pointer, negative size, unknown enum) */
#define ICP_ERR_NO_DEVICE (-2) /* no usable HIP device / device index out of range */
#define ICP_ERR_HIP (-3)       /* a HIP runtime call or kernel launch failed (see icp_last_error) */
#define ICP_ERR_EMPTY (-4)     /* empty model cloud (m == 0) where a match is required */
#define ICP_ERR_SINGULAR (-5)  /* 6x6 point-to-plane system not positive definite */
#define ICP_ERR_IO (-6)        /* dataset file missing / malformed */
#define ICP_ERR_STATE (-7)     /* call sequence error (e.g. step before begin, clouds not set) */
#define ICP_ERR_NOMEM (-8)

typedef enum { ICP_F32 = 0, ICP_F64 = 1 } icp_precision;
typedef enum { ICP_POINT_TO_POINT = 0, ICP_POINT_TO_PLANE = 1 } icp_metric;

typedef struct icp_ctx icp_ctx; /* opaque: owns device buffers, stream, pinned staging */

/* number of doubles in the per-iteration moment vector (the only data that ever crosses ranks) */
#define ICP_NMOM 32
/* moment vector slots (doubles).  Point-to-point uses [0..18], point-to-plane [0..1] + [2..28]. */
#define ICP_MOM_ERR 0  /* sum |p_new - q[idx_prev]|^2 of the transform applied in this enqueue */
#define ICP_MOM_CNT 1  /* number of moving points that contributed */
#define ICP_MOM_SP 2   /* sum p (3) */
#define ICP_MOM_SQ 5   /* sum q[idx] (3) */
#define ICP_MOM_SQP 8  /* sum q_a * p_b, a-major (9) */
#define ICP_MOM_SPP 17 /* sum |p|^2   (informational: nothing on the path reads these two; the single-GPU point-to-point */
#define ICP_MOM_SQQ 18 /* sum |q[idx]|^2  fast path, whose rows the host adds itself, leaves them 0)                         */
#define ICP_MOM_C 2    /* point-to-plane: upper triangle of C, row-major (21) */
#define ICP_MOM_B 23   /* point-to-plane: b (6) */

typedef struct icp_params {
    int max_iter;         /* MAX_ITER: 200 src/ICP_CPU.c:17, 40 src/ICP_point_to_point.cu:24, 50 ICP_point_to_plane.cu:23 */
    double tol;           /* 1e-5 src/ICP_CPU.c:267, 1e-6 src/ICP_point_to_point.cu:420 */
    int fixed_iterations; /* != 0: no tolerance test, run exactly max_iter passes (src/ICP_standard.cu:369) */
    int precision;        /* icp_precision of the device arithmetic */
    int metric;           /* icp_metric */
} icp_params;

typedef struct icp_result {
    double T[16];        /* composed transform, row-major 4x4: T = [R_k|t_k] * ... * [R_0|t_0] */
    int iterations;      /* the reference's `iteration` / `num_iterations` at loop exit */
    int passes;          /* matching passes that contributed to T */
    double* err;         /* caller-allocated max_iter+1 doubles (or NULL); err[0] = 0, err[k] = RMS after pass k-1 */
    int32_t* idx;        /* caller-allocated n int32 (or NULL): correspondences of the last contributing pass */
    void* moved;         /* caller-allocated 3n values of the run's precision, AoS (or NULL): final moving cloud */
    double seconds_total;   /* wall-clock of the loop (upload/download excluded) */
    double seconds_nn;      /* device time of the matching kernels, summed (0 unless profiling enabled) */
    /* per-phase seconds (the reference's match_time / minimization_time / transf_time / error_time,
     * src/CUDA/GPU_point_to_point_real.cu:241,386-403).  Transformation and error estimation have no time of their own
     * here: they are the front end of the matching kernel (same launch) and are inside seconds_nn. */
    double seconds_host;    /* host half of the iterations, summed: error + stop rule + 3x3 SVD / 6x6 solve (0 unless profiling enabled) */
    double seconds_setup;   /* icp_point_to_*: icp_set_model (+ normals) + icp_set_moving of this call -- upload, layout, order, boxes */
} icp_result;

/* ---- library / context ------------------------------------------------------------------- */
int icp_abi_version(void);
const char* icp_strerror(int code);
/* last error text of the calling thread (HIP error strings etc.), never NULL */
const char* icp_last_error(void);
/* number of usable HIP devices, or a negative error code */
int icp_device_count(void);
/* The loop is a host-thread <-> GPU conversation and every message from the other socket costs ~0.5 us more, so the entry
 * points that hold that conversation (icp_create while it allocates its pinned buffers, icp_loop_run, icp_loop_complete,
 * icp_point_to_*) narrow the calling thread's CPU affinity to the device's NUMA node (sysfs local_cpulist) if the thread
 * currently runs elsewhere -- and put the caller's mask back before they return.  ICP_PIN=0 never touches the affinity;
 * ICP_PIN=2 narrows once in icp_create and keeps it (a thread dedicated to the context). */
int icp_create(int device, icp_ctx** out);
void icp_destroy(icp_ctx* ctx);
/* run all work of this context on an externally owned hipStream_t (e.g. torch's current stream);
 * NULL restores the context's own stream */
int icp_set_stream(icp_ctx* ctx, void* hip_stream);
/* hipEvent timing of the matching kernel inside the loop: 0 = off, n > 0 = time every n-th launch
 * (two event records + a stream synchronisation on the timed launches only; with a resident registration
 * kernel the launch is the whole registration).  The call restarts the stride and the accumulators: the first
 * launch after it is a timed one. */
int icp_set_profiling(icp_ctx* ctx, int every_nth);

/* ---- matching seam: replaces  Matching<<<>>>(n, P, Q, q_points, idx)
 *      src/CUDA/GPU_point_to_point_real.cu:38-79, src/ICP_point_to_point.cu:31-57 (fp32) and the
 *      MKL loop src/ICP_CPU.c:220-234 (fp64).  Host pointers, AoS.  idx[i] in [0, m). --------- */
int icp_nn_match_f32(icp_ctx* ctx, const float* P_aos, int n, const float* Q_aos, int m, int32_t* idx);
int icp_nn_match_f64(icp_ctx* ctx, const double* P_aos, int n, const double* Q_aos, int m, int32_t* idx);

/* ---- resident clouds (data stays in HBM between calls) ------------------------------------- */
/* upload + convert to the internal padded SoA layout.  precision: ICP_F32 / ICP_F64 selects the
 * element type of `xyz_aos` AND of the device arithmetic. */
int icp_set_model(icp_ctx* ctx, const void* xyz_aos, int m, int precision);
int icp_set_moving(icp_ctx* ctx, const void* xyz_aos, int n, int precision);
/* unit normals of the model points, AoS, same precision as the model (point-to-plane) */
int icp_set_model_normals(icp_ctx* ctx, const void* nxyz_aos, int m);
/* put the moving cloud back to the state icp_set_moving uploaded (device-to-device copy of a resident pristine
 * copy): lets a caller register the same pair repeatedly without touching PCIe */
int icp_reset_moving(icp_ctx* ctx);
int icp_get_moving(icp_ctx* ctx, void* xyz_aos_out);        /* 3n values, precision of the cloud */
int icp_get_indices(icp_ctx* ctx, int32_t* idx_out);        /* n int32: the most recent matching pass */
/* one matching pass over the resident clouds; indices stay on the device.  kernel_ms (optional)
 * receives the hipEvent time of the matching kernel(s) alone. */
int icp_nn_match_resident(icp_ctx* ctx, float* kernel_ms);
/* geometry of the last matching launch (the programs print it as the reference prints its Grid Size / Block Size,
 * src/CUDA/GPU_point_to_point_real.cu:237) */
int icp_nn_launch_info(icp_ctx* ctx, int* splits, int* blocks, int* threads, int* n_pad, int* m_pad);
/* The caller owns the device (no other context of this or any other process keeps kernels resident on it): clouds of up to
 * 16 384 moving points (one row of 64 per CU) then run their rows as 16-wave blocks, one to a CU, instead of 8-wave blocks that
 * leave room for a second resident context -- hall pair 9.5 -> 9.15 us per iteration (profiles/r2/r2_02_waves_8_vs_16.txt).
 * The results are the same bits.  Off by default: the library cannot see who else is on the device.  (The reference's
 * programs own their GPU implicitly, src/ICP_point_to_point.cu:90.) */
int icp_set_exclusive(icp_ctx* ctx, int on);

/* ---- model normals: replaces knn + Normals + host ssyev loop
 *      src/CUDA/GPU_point_to_plane_real.cu:54-188,391-423 (k = 4 neighbours, self excluded).
 *      Works on the resident model; results stay resident and are optionally returned. ------- */
int icp_estimate_normals(icp_ctx* ctx, void* nxyz_aos_out /*3m or NULL*/, int32_t* neighbours_out /*4m or NULL*/);

/* ---- the ICP loops: replace main()'s while-loops
 *      point-to-point  src/ICP_CPU.c:217-271, src/ICP_point_to_point.cu:295-423
 *      point-to-plane  src/ICP_point_to_plane.cu:517-631, src/CUDA/CPU_ICP_point_to-plane.cpp:309-428
 *      Clouds are host AoS arrays of prm->precision.  For point-to-plane the model normals are
 *      estimated on the device first unless `normals_aos` is given. ---------------------------- */
int icp_point_to_point(icp_ctx* ctx, const void* data_aos, int n, const void* model_aos, int m,
                       const icp_params* prm, icp_result* out);
int icp_point_to_plane(icp_ctx* ctx, const void* data_aos, int n, const void* model_aos, int m,
                       const void* normals_aos /*may be NULL*/, const icp_params* prm, icp_result* out);

/* ---- step-wise loop over the resident clouds (what the loops above are built from; used by the
 *      multi-GPU driver, which all-reduces the moment vector between enqueue and complete) ---- */
int icp_loop_begin(icp_ctx* ctx, const icp_params* prm);
/* enqueue (asynchronously): [transform + error of the previous pass] -> matching -> fused
 * gather/moments -> finalize into the ICP_NMOM-double device vector. */
int icp_loop_enqueue(icp_ctx* ctx);
/* device address of that vector (valid until icp_destroy); to be summed across ranks in place */
void* icp_loop_moments_dev(icp_ctx* ctx);
/* optional: make the loop write its moments into caller-owned device memory (e.g. a torch tensor) */
int icp_loop_set_moments_dev(icp_ctx* ctx, void* dev_ptr_32_doubles);
/* copy the (reduced) vector back, evaluate the stop rule, solve R,t for the next pass.
 * *done != 0 when the loop has ended. */
int icp_loop_complete(icp_ctx* ctx, int* done);
/* up to max_steps x (enqueue + complete) without returning to the caller in between (single GPU, or a
 * communicator attached with icp_comm_init); stops early when the loop ends */
int icp_loop_run(icp_ctx* ctx, int max_steps, int* steps_done, int* done);
/* A pass of icp_loop_run's resident / armed conversation that never delivers its rows (a lost message, blocks that another
 * process kept off the machine) does not end the registration when the loop started from the cloud icp_set_moving uploaded
 * (or icp_reset_moving restored) and no communicator is attached: the kernel is withdrawn and the same registration is run
 * again from the copy with plain launches, one per pass -- every loop form yields the same bits.  ICP_ERR_HIP is returned
 * only if that fails as well.  icp_recoveries: how often this context has done so (the reference's loops,
 * src/ICP_point_to_point.cu:308-421, are plain launches throughout and have nothing to recover from). */
int icp_recoveries(icp_ctx* ctx);
/* current state: iterations so far, error series (count doubles), composed transform.
 * After a FAILED icp_loop_run / icp_loop_complete: a numeric failure (degenerate correspondences: ICP_ERR_SINGULAR /
 * ICP_ERR_INVALID from the minimisation) ends the loop but keeps its state readable -- this call, icp_loop_indices and
 * icp_get_moving answer for the passes that completed.  A device failure (ICP_ERR_HIP, a pass that never delivered its rows)
 * discards the loop: this call then returns ICP_ERR_STATE and icp_get_moving the cloud as icp_set_moving uploaded it. */
int icp_loop_state(icp_ctx* ctx, int* iterations, int* passes, double* err, int err_cap, double* T16);
/* summed hipEvent time and count of the matching-kernel launches timed since icp_set_profiling was last called
 * (cumulative over loops; the bench's roofline leg reads the timed region through this) */
int icp_loop_timing(icp_ctx* ctx, double* seconds_nn, int* nn_launches);
/* matching passes executed by those timed launches: equal to their count when every pass is its own launch, larger
 * when icp_loop_run keeps ONE resident kernel for a whole registration (that kernel is then the timed launch, every
 * n-th one, host round trips between its passes included) */
int icp_loop_timing_passes(icp_ctx* ctx, long long* passes);
/* the current (or last) loop's own phase sums: matching-kernel seconds and host-solve seconds, as icp_result reports them
 * (both 0 unless icp_set_profiling is on); either pointer may be NULL */
int icp_loop_phase_seconds(icp_ctx* ctx, double* seconds_nn, double* seconds_host);
/* correspondences of the last pass that contributed to T (ping-pong buffer), n int32 */
int icp_loop_indices(icp_ctx* ctx, int32_t* idx_out);

/* ---- multi-GPU: the loop's single collective issued by the library (RCCL over xGMI, bound at run time) ----
 * One process per GPU.  Rank 0 obtains an id (icp_comm_unique_id), the host application distributes those
 * ICP_COMM_ID_BYTES bytes by any means (MPI, a torch.distributed broadcast, a file), every rank calls
 * icp_comm_init on its context.  From then on icp_loop_enqueue all-reduces (sum, in place) the ICP_NMOM vector
 * right behind the finalize kernel on the loop's stream; icp_loop_complete sees the global sums.  Shard the
 * MOVING cloud with icp_shard_range and give every rank the full model. */
#define ICP_COMM_ID_BYTES 128
int icp_comm_unique_id(void* out_id_bytes);
int icp_comm_init(icp_ctx* ctx, const void* id_bytes, int rank, int world);
int icp_comm_destroy(icp_ctx* ctx);
/* The same exchange for ranks on ONE node, through POSIX shared memory instead of a device collective: in the
 * single-node fast path the loop's vector is already in host memory when the rows have been added, and 256 bytes
 * between processes of a node take ~1 us (RCCL: ~15-20 us, more than the iteration).  Every rank adds the ranks'
 * vectors in rank order, so all ranks continue from bit-identical sums; the resident-kernel loop stays available.
 * id_bytes: ICP_COMM_ID_BYTES random bytes from rank 0 (icp_comm_random_id), distributed by the application. */
int icp_comm_random_id(void* out_id_bytes);
int icp_comm_init_local(icp_ctx* ctx, const void* id_bytes, int rank, int world);
/* the host-memory communicator on its own (no device needed: multi-process CPU tests, custom drivers);
 * icp_lcomm_allreduce: v[0..count) <- sum over ranks in rank order, count <= ICP_NMOM */
typedef struct icp_lcomm icp_lcomm;
int icp_lcomm_create(const void* id_bytes, int rank, int world, icp_lcomm** out);
int icp_lcomm_allreduce(icp_lcomm* comm, double* v, int count);
void icp_lcomm_destroy(icp_lcomm* comm);

/* ---- host-only pieces (no device needed; exercised by the CPU test-suite) ------------------- */
/* 3x3 cross-covariance solve from raw moments: replaces cublasSgemm + cusolverDnSgesvd + 2 gemm
 * (src/ICP_point_to_point.cu:356-397) / dgemm + LAPACKE_dgesvd (src/ICP_CPU.c:239-248).
 * R = U*Vt with NO reflection fix (the reference has none).  Returns ICP_OK. */
int icp_solve_point_to_point(const double* mom /*ICP_NMOM*/, double* R9, double* t3);
/* 6x6 normal equations: replaces cusolverDnSpotrf/Spotrs UPPER (src/ICP_point_to_plane.cu:576-581)
 * and LAPACKE_ssysv (CPU_ICP_point_to-plane.cpp:371); x = (alpha,beta,gamma,tx,ty,tz), then the
 * full (non-linearised) R = Rz(gamma) Ry(beta) Rx(alpha) (src/ICP_point_to_plane.cu:585-593). */
int icp_solve_point_to_plane(const double* mom /*ICP_NMOM*/, double* R9, double* t3, double* x6);
/* The host half of the loops above as a device-free state machine (the device loop runs this very
 * code): feed it the rank-reduced ICP_NMOM vector of each pass, it returns the stop decision and the
 * next R, t; tell it when that motion has been applied.  Sequence per pass:
 *   advance(mom_k) -> [done?] -> apply R,t to the shard -> note_applied() -> (next pass' moments) ... */
typedef struct icp_host_loop icp_host_loop;
int icp_host_loop_create(const icp_params* prm, icp_host_loop** out);
void icp_host_loop_destroy(icp_host_loop* h);
int icp_host_loop_advance(icp_host_loop* h, const double* mom /*ICP_NMOM*/, int* done, double* R9, double* t3);
int icp_host_loop_note_applied(icp_host_loop* h);
int icp_host_loop_state(icp_host_loop* h, int* iterations, int* passes, double* err, int err_cap, double* T16);
/* contiguous shard [begin, begin+count) of n moving points for `rank` of `world`.  Any partition of the moving points is the same
 * registration, and a registration is as slow as its slowest rank: where the work per point varies over a LARGE cloud (BASELINE
 * configs[4]: contiguous eighths take 16 to 30 ms) deal compact blocks of the cloud to the ranks instead (distributed.curve_order +
 * shard_cyclic_index of the Python mirror) -- every rank simply passes its own points to icp_set_moving. */
int icp_shard_range(int64_t n, int rank, int world, int64_t* begin, int64_t* count);
/* symmetric 3x3 eigen-solve used for the normals (upper triangle of row-major A read);
 * w ascending, Z[i*3+k] = component i of eigenvector k */
int icp_eigh3(const double* A9, double* w3, double* Z9);

/* ---- datasets: the reference's input formats --------------------------------- */
/* synthetic z = x^2 - y^2 grid, fp32 AoS: src/ICP_point_to_point.cu:103-152 (W*W points) */
int icp_synthetic_grid_f32(int W, float xy_min, float xy_max, float* D_aos);
/* fp64 AoS variant of src/ICP_CPU.c:51-95 */
int icp_synthetic_grid_f64(int W, double xy_min, double xy_max, double* D_aos);
/* model = R*D + t with the closed-form column-major rotation of the GPU programs
 * (src/ICP_point_to_point.cu:157-190), fp32 */
int icp_make_model_f32(const float* D_aos, int n, const float angles_xyz[3], const float t[3], float* M_aos);
/* model of src/ICP_CPU.c:100-149 (r = rx*ry*rz, +sin above the diagonal), fp64 */
int icp_make_model_cpu_f64(const double* D_aos, int n, const double angles_xyz[3], const double t[3], double* M_aos);
/* the hard-coded rotation of src/ICP_standard.cu:247-249 */
int icp_make_model_standard_f32(const float* D_aos, int n, float* M_aos);
/* "x y z" / "x;y;z" text (Bunny_res.csv / Bunny.csv): src/CUDA/GPU_point_to_point_bunny.cu:463-497.
 * Returns the number of POINTS read (>= 0) or a negative error; at most cap_points are stored. */
int icp_read_xyz_text(const char* path, float* out_aos, int cap_points);
/* Ouster OS1-16 dump, one byte value per text line (Donut_1024x16.csv) or raw binary packets
 * (12608 B each): src/CUDA/GPU_point_to_point_real.cu:432-488.  16 beams x 16 azimuth blocks per
 * packet; ranges in mm.  Returns the number of ranges or a negative error. */
int icp_read_os1_ranges(const char* path, uint32_t* ranges_out, int cap, uint32_t* encoder_count0);
/* beam_intrinsics.csv: 64 altitude + 64 azimuth angles (deg), the 16 used beams selected as the
 * reference does (every 4th from the 3rd): src/CUDA/GPU_point_to_point_real.cu:503-527 */
int icp_read_os1_intrinsics(const char* path, float altitude16[16], float azimuth16[16]);
/* polar -> Cartesian on the device, replaces Conversion<<<>>> src/CUDA/GPU_point_to_point_real.cu:20-36.
 * Output AoS fp32 in mm (host). */
int icp_os1_to_cartesian(icp_ctx* ctx, const uint32_t* ranges, int n, uint32_t encoder_count0,
                         const float altitude16[16], const float azimuth16[16], float* xyz_aos_mm);

/* raw OS1-16 packets (n_packets x 12608 bytes, as captured from the sensor) -> ranges + Cartesian points in one
 * device pass: replaces the host parse loop + H2D + Conversion<<<>>> of
 * src/CUDA/GPU_point_to_point_real.cu:457-487,538-563.  256 points per packet; ranges_out may be NULL. */
int icp_os1_packets_to_cartesian(icp_ctx* ctx, const uint8_t* packets, int n_packets, const float altitude16[16],
                                 const float azimuth16[16], float* xyz_aos_mm, uint32_t* ranges_out);

#ifdef __cplusplus
}
#endif
#endif /* ICP_MI355X_H */
