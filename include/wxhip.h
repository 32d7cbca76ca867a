/*
 * wxhip.h - C ABI of the MI355X-native RHS / JVP engine (libwxhip.so).
 *
 * Drop-in boundary for ONE hot path of WxFactory: evaluation of the spatial
 * right-hand side R(Q) of the DFR discretisation on the cubed sphere.
 * Every entry point cites the reference interface it replaces (paths relative to
 * the WxFactory tree, wx_factory/...).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions cross the boundary.
 *   - every call returns a wx_status; wx_last_error() gives the message (thread local).
 *   - all array arguments are DEVICE pointers unless marked [host]; the caller owns them
 *     (same ownership rule as the reference's pybind module, pde/interface.hpp:158-171).
 *   - all work is enqueued on the caller's hipStream_t; no evaluation entry point synchronises the
 *     device, allocates after plan creation, or keeps a pointer to q / rhs / halo.  The stream's
 *     device is made current for the duration of a call (and restored), so a process may drive
 *     several GPUs, or call with another current device, without invalid-handle errors.
 *   - SETUP-TIME calls are the exception: wx_*_plan_create*, wx_*_batch_create, wx_expfilter_create and
 *     wx_euler3d_set_exp_filter allocate, use the null stream and blocking copies, and SYNCHRONISE THE
 *     DEVICE first (so that static fields still being produced on a non-blocking stream are complete,
 *     and no kernel is still reading constants about to be replaced).  Do not call them while a
 *     stream is being captured into a graph.
 *   - what only some callers need is allocated by wx_euler3d_plan_reserve at setup time: the second interface slot of the
 *     stage pipeline (wx_euler3d_stage / wx_euler3d_extrap_pack_slot) and the face-value cache of the prepared JVP
 *     (wx_euler3d_jvp_prepare).  Without it those entry points return WX_ERR_INVALID; with it their FIRST call may sit inside
 *     a stream capture like any later one.
 *   - device flags (nan_flag arguments) are only ever raised to 1 by plain stores from many threads - one
 *     value, so no atomic is needed - and never cleared by the library.
 *   - arrays are C-contiguous float64 (WX_F64) or complex128 (WX_C128, interleaved
 *     re,im) in the reference's element-blocked layout (geometry/cubed_sphere_3d.py:187-205):
 *         state     q[var][ek][ej][ei][p],  p = (kl*n + jl)*n + il
 *         faces     [..][2*n*n]  first n*n = minus side (W/S/bottom), then plus side
 */
#ifndef WXHIP_H
#define WXHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* wx_stream; /* hipStream_t */

typedef enum {
    WX_OK = 0,
    WX_ERR_INVALID = 1,     /* bad argument (null pointer, unsupported n, shape mismatch) */
    WX_ERR_UNSUPPORTED = 2, /* valid request this build cannot serve (dtype, case) */
    WX_ERR_HIP = 3,         /* a HIP runtime call failed; message has hipGetErrorString */
    WX_ERR_NOMEM = 4,
    WX_ERR_COMM = 5         /* an RCCL call failed; message has ncclGetErrorString */
} wx_status;

typedef enum {
    WX_F64 = 0,
    WX_C128 = 1,   /* complex128, true complex arithmetic with NumPy's abs / maximum rules */
    /* complex128 STORAGE with first-order (dual-number) arithmetic: re = value, im = tangent.  For the
     * complex-step Jacobian-vector product of solvers/matvec.py:56-61 it returns the same real and
     * imaginary parts as WX_C128 up to O(eps^2) ~ 2e-16 at a fraction of the cost; it is NOT complex
     * arithmetic for inputs whose imaginary part is not a small perturbation. */
    WX_DUAL128 = 2
} wx_dtype;

/* Which elements of the tile an RHS launch covers (lets the caller overlap the halo
 * exchange with the interior, the ordering of rhs/rhs.py:88-118). */
typedef enum {
    WX_REGION_ALL = 0,
    WX_REGION_INTERIOR = 1, /* elements that touch no lateral tile edge: need no halo */
    WX_REGION_BOUNDARY = 2  /* the ring of elements on the four lateral tile edges   */
} wx_region;

const char* wx_last_error(void);
/* "wxhip <version> gfx950" */
const char* wx_version(void);
/* the build-time switches of this library, e.g. "WX_MFMA=1 WX_K2_DIAG=0" (product build); A/B and diagnostic variants
 * differ here, so that a measurement can name the variant it was taken on */
const char* wx_build_info(void);
/* number of HIP devices visible; <0 on error.  Does not create a context. */
int wx_device_count(void);
/* dst = src, `bytes` of them (a multiple of 16, 16-byte aligned device buffers), as one streaming kernel - 16 bytes per lane,
 * every wave slot of the chip: the measured read-once / write-once rate a kernel's roofline fraction is put beside
 * (bench.py: roofline.ceiling).  Moves 2 x bytes through HBM. */
wx_status wx_stream_copy(const void* src, void* dst, size_t bytes, wx_stream stream);
/* The read side alone: `bytes` of src (a multiple of 16) streamed once and summed; sink receives
 * wx_stream_read_sink_doubles() partial sums (their total = the sum of src: the check that everything was read).  A kernel
 * whose traffic is mostly reads (the fused RHS kernel: 89 %) is put beside THIS rate, a balanced one beside the copy's. */
int wx_stream_read_sink_doubles(void);
wx_status wx_stream_read(const void* src, size_t bytes, double* sink, wx_stream stream);

/* ------------------------------------------------------------------------------------------
 * The reference's per-evaluation timing row without torch: RHS.timestamps / retrieve_last_times (rhs/rhs.py:39-41, 68-118;
 * device.timestamp / device.elapsed, device/device.py) - nine device timestamps per evaluation, consumed as eight
 * intervals + total (rhs.py:191-200, output/solver_stats.py:158-175: the `rhs_timing` table).
 *   wx_phase_timer_stamp(t, slot, stream)   records an event on `stream` for slot 0..8 (a caller that drives the two
 *       kernels itself - INTEGRATION.md section 4, the MPI route - stamps: 0 start, 1 after wx_euler3d_extrap_pack,
 *       2 exchange posted, 3 = 4 after the INTERIOR launch, 5 exchange complete, 6 = 7 = 8 after the BOUNDARY / ALL
 *       launch: pointwise fluxes + divergence, and Riemann + correction + forcing, are one kernel each here)
 *   wx_phase_timer_elapsed(t, out[9])       waits for every stamped event (they may sit on different streams) and writes
 *       the eight intervals between consecutive stamps and their total, in SECONDS like device.elapsed; slots never
 *       stamped repeat their predecessor (interval 0), and so does a stamp that another stream reached before its
 *       predecessor.  Returns WX_ERR_INVALID when slot 0 or 8 was not stamped.
 * ------------------------------------------------------------------------------------------ */
typedef struct wx_phase_timer wx_phase_timer;
wx_status wx_phase_timer_create(wx_phase_timer** timer);
wx_status wx_phase_timer_destroy(wx_phase_timer* timer);
wx_status wx_phase_timer_stamp(wx_phase_timer* timer, int slot, wx_stream stream);
wx_status wx_phase_timer_elapsed(wx_phase_timer* timer, double seconds[9]);
/* seconds from stamp 0 to every stamped slot (a slot without a stamp: -1): where two streams' phases lie against each other -
 * did the exchange on the compute stream (slot 5) complete before the INTERIOR launches on the second stream ended (slot 3)? */
wx_status wx_phase_timer_since_start(wx_phase_timer* timer, double seconds[9]);

/* Streams with a scheduling priority, for the overlapped evaluation (wx_*_rhs_overlapped, wx_exchange_fork): the INTERIOR launches
 * fill every CU, and the grouped ncclSend / ncclRecv that should run beside them are a handful of workgroups on the compute stream.
 * MEASURED on MI355X (profiles/r06_overlap_priority_ab.txt, the loopback rehearsal of bench.py with the stamps of
 * wx_phase_timer_since_start): with both streams at normal priority the exchange enqueued BEHIND the INTERIOR launches completes
 * 50 us after INTERIOR starts - the dispatcher serves the hardware queues in turn and a workgroup of the fused kernel lives a few
 * microseconds, so the exchange's workgroups are not starved -, while INTERIOR on a lowest-priority stream runs 23 % slower even
 * alone on the device.  No priority is required; these entry points let a caller choose one (torch.cuda.Stream offers normal
 * and high only, no low).
 * priority_class: +1 lowest, 0 normal, -1 highest (mapped onto hipDeviceGetStreamPriorityRange).  Non-blocking streams of the
 * current device; wx_stream_destroy synchronises the stream first. */
wx_status wx_stream_priority_range(int* least, int* greatest);
wx_status wx_stream_create(wx_stream* stream, int priority_class);
wx_status wx_stream_destroy(wx_stream stream);

/* ------------------------------------------------------------------------------------------
 * 3-D Euler on a cubed-sphere tile.
 * Replaces  rhs/rhs_dfr.py:48-313  (RHSDirecFluxReconstruction_mpi, phases 1-8 of
 *           rhs/rhs.py:75-122) together with pde/pde_euler_cubesphere.py:72-290 and
 *           pde/fluxes.py:150-222, 326-403, 507-582.
 * ------------------------------------------------------------------------------------------ */

/* Static fields read by the path, exactly the reference's arrays, read IN PLACE (no copy):
 * geometry/metric3d.py:1108-1157.  nh = H+2, nv = V+2 (halo-padded interface arrays). */
typedef struct {
    const double* sqrtG;          /* (V,H,H,n^3)        metric.sqrtG_new                 */
    const double* h_contra;       /* (3,3,V,H,H,n^3)    metric.h_contra_new              */
    const double* christoffel;    /* (3,9,V,H,H,n^3)    metric.christoffel               */
    const double* inv_dzdeta;     /* (V,H,H,n^3)        metric.inv_dzdeta_new            */
    const double* sqrtG_itf_i;    /* (V,H,nh,2n^2)      metric.sqrtG_itf_i_new           */
    const double* sqrtG_itf_j;    /* (V,nh,H,2n^2)      metric.sqrtG_itf_j_new           */
    const double* sqrtG_itf_k;    /* (nv,H,H,2n^2)      metric.sqrtG_itf_k_new           */
    const double* h_contra_itf_i; /* (3,3,V,H,nh,2n^2)  metric.h_contra_itf_i_new        */
    const double* h_contra_itf_j; /* (3,3,V,nh,H,2n^2)  metric.h_contra_itf_j_new        */
    const double* h_contra_itf_k; /* (3,3,nv,H,H,2n^2)  metric.h_contra_itf_k_new        */
    /* Rayleigh sponge of DCMIP 2-1/2-2 (init/dcmip.py:676-757): forcing_i += damp_coef*rho*
     * (u_i - damp_uref_i).  Both NULL unless case_number is 21 or 22. */
    const double* damp_coef;      /* (V,H,H,n^3)   sin^2(pi/2 (z-zh)/(ztop-zh))/tau0 above zh, else 0 */
    const double* damp_uref;      /* (3,V,H,H,n^3) contravariant reference wind          */
    /* tan of the edge coordinate used by the vector rotation (process_topology.py:137-175):
     * geom.boundary_sn (H*n values along x1), geom.boundary_we (H*n values along x2). */
    const double* boundary_sn;
    const double* boundary_we;
} wx_euler3d_metric;

/* 1-D operator pieces [host], geometry/operators.py:55-80, 144-148 (row-major). */
typedef struct {
    const double* extrap_neg; /* (n)    ops.extrap_west  */
    const double* extrap_pos; /* (n)    ops.extrap_east  */
    const double* diff_solpt; /* (n,n)  ops.diff_solpt   */
    const double* correction; /* (n,2)  ops.correction   */
    const double* highfilter; /* (n,n)  ops.highfilter   */
} wx_dfr_ops;

typedef struct wx_euler3d_plan wx_euler3d_plan;

/* Build a plan for one tile.  `panel` (0..5) selects the flip/rotation tables of
 * process_topology.py:105-175 for the four lateral edges (one tile per panel).
 * Allocates the plan's private interface buffer (6*5*n^2 values per element). */
wx_status wx_euler3d_plan_create(wx_euler3d_plan** plan, int n, int H, int V, int case_number, wx_dtype dtype,
                                 int panel, const wx_dfr_ops* ops, const wx_euler3d_metric* metric);
/* The same for one of k x k tiles of a panel (the reference's 6 k^2-rank decomposition,
 * process_topology.py:69-94): on_panel_edge[e] != 0 where edge e (S,N,W,E) of the tile lies on the
 * panel's edge and takes the panel's rotation / flip; interior tile edges exchange unrotated, unflipped
 * (process_topology.py:219-228).  H, the metric arrays and boundary_sn/we are the TILE's. */
wx_status wx_euler3d_plan_create_tile(wx_euler3d_plan** plan, int n, int H, int V, int case_number, wx_dtype dtype,
                                      int panel, const int on_panel_edge[4], const wx_dfr_ops* ops,
                                      const wx_euler3d_metric* metric);
wx_status wx_euler3d_plan_destroy(wx_euler3d_plan* plan);
/* Setup-time allocation of the buffers only some callers need (a bit mask); idempotent.  wx_euler3d_plan_reserved: what the
 * plan holds. */
typedef enum {
    WX_RESERVE_STAGE = 1, /* second interface slot: wx_euler3d_stage with prepare_next / itf_in = 1, wx_euler3d_extrap_pack_slot(.., 1) */
    WX_RESERVE_JVP = 2    /* face-value cache of the prepared complex-step JVP (WX_DUAL128 plans): wx_euler3d_jvp_prepare */
} wx_reserve;
wx_status wx_euler3d_plan_reserve(wx_euler3d_plan* plan, int what);
int wx_euler3d_plan_reserved(const wx_euler3d_plan* plan);
/* the dtype the plan was created with (WX_F64 when plan is NULL) */
wx_dtype wx_euler3d_plan_dtype(const wx_euler3d_plan* plan);

/* Values per face point in an edge message: the 5 prognostic variables the reference exchanges
 * (rho, rho u1, rho u2, rho w, rho theta). */
#define WX_EULER3D_EDGE_FIELDS 5
/* Compulsory HBM bytes per solution point of one wx_euler3d_rhs launch on this plan (SURVEY.md 8d figure:
 * 8 (5 Q + 5 R + sqrtG + 6 h + 27 Gamma + inv_dzdeta) + interface metric = 384 B at n = 8), after the plan-time
 * specialisations: plan creation scans the nine rotation Christoffel symbols christoffel[:, 0:3] once and, when
 * they are identically zero (non-rotating planets: DCMIP 2-x / 3-1), the kernels never read them (312 B). */
double wx_euler3d_bytes_per_point(const wx_euler3d_plan* plan);

/* Column form of the metric (WX_F64 and WX_DUAL128 plans).  On a shallow atmosphere without topography every metric array of
 * geometry/metric3d.py takes the same values on all levels (config/dcmip31.ini, the reference's tests/rhs_benchmark): give
 * the plan ONE slab per column and field - point fields [..][H][H][n^2] (the lowest level of the lowest element),
 * sqrtG_itf_i / h_contra_itf_i [..][H][H+2][2][n], sqrtG_itf_j / h_contra_itf_j [..][H+2][H][2][n] (the n values along
 * the face's horizontal direction), sqrtG_itf_k / h_contra_itf_k [..][H][H][n^2]; damp_*, boundary_* are not read - and
 * wx_euler3d_rhs / _rhs_axpy* / wx_euler3d_stage (WX_F64; the stage pipeline incl. its fused filter) and wx_euler3d_jvp /
 * _jvp_prepared (WX_DUAL128), every region, read those instead of the full arrays, the elements of a column following
 * each other so that its slabs are fetched once.  The CALLER vouches for the invariance (the Python host checks it:
 * rhs_euler3d.column_metric_slabs); the full arrays stay in use for every other entry point.  NULL: back to the full
 * arrays.  The slabs are borrowed like the plan's metric. */
wx_status wx_euler3d_plan_set_column_metric(wx_euler3d_plan* plan, const wx_euler3d_metric* column_metric);
int wx_euler3d_plan_has_column_metric(const wx_euler3d_plan* plan);

/* Which engine the derivative-matrix x nodal-field contractions (geometry/operators.py:157-183 in sum-factorised
 * form) of a launch on this plan run on: 1 = the matrix cores (v_mfma_f64_4x4x4_4b_f64), 0 = the vector pipe, < 0 on
 * a bad argument.  A compile-time property of the instantiation the library selects for (n, dtype, kernel); the parity
 * tests assert it so that "the matrix-core kernels are pinned to the reference" is a checked statement. */
typedef enum {
    WX_KERNEL_RHS = 0,       /* wx_euler3d_rhs / _rhs_axpy* / _shifted_rhs_axpy2 */
    WX_KERNEL_STAGE = 1,     /* wx_euler3d_stage with prepare_next != 0 (incl. the fused exponential filter) */
    WX_KERNEL_JVP = 2,       /* wx_euler3d_jvp / _jvp_prepared */
    WX_KERNEL_BATCH_RHS = 3, /* wx_euler3d_batch_rhs_axpy2 */
    WX_KERNEL_BATCH_JVP = 4  /* wx_euler3d_batch_jvp / _batch_kiops_vector */
} wx_kernel;
int wx_euler3d_uses_matrix_cores(const wx_euler3d_plan* plan, wx_kernel kernel);

/* Low orders (WX_F64; default at num_solpts 2, available for 3 and 4 - the orders of the reference's shipped configurations
 * and of its RHS benchmark, tests/rhs_benchmark/run.sh:67-71): the evaluation is ONE kernel that keeps the face states on chip (a workgroup owns a
 * brick of elements, solves every Riemann problem of the brick once, and extrapolates the states beyond the brick's surface
 * from the neighbour elements' nodal values; csrc/euler3d_brick.h).  On such a plan wx_euler3d_extrap_pack* write the edge
 * messages only - there is no interface buffer - and every wx_euler3d_rhs* / _stage / _shifted_* call reads q alone; the
 * calling sequence and the results (to rounding) are those of the two-kernel form.  wx_euler3d_plan_one_kernel: 1 when the
 * plan takes this form (0 otherwise, < 0 on a null plan); wx_euler3d_plan_set_one_kernel(plan, 0 / 1) switches it at setup
 * time (0 = the two-kernel form; 1 is refused for plans the form does not serve).  The environment variable
 * WXHIP_DIRECT=0, read when a plan is created, makes 0 the default. */
/* The one-kernel form takes its float64 logarithms (log rho, log rho theta of rhs_dfr.py:50-71) from a 38-instruction form
 * (x = 2^k m; s = (m - 1) / (m + 1); the classical degree-7 series in s^2; error < 1 ulp for positive normal arguments) instead of
 * the device library's 84-instruction one: its face stage is bound by its instruction count.  wx_lean_log applies that function to
 * an array (device pointers): the accuracy claim is a test (tests/test_low_order_gpu.py). */
wx_status wx_lean_log(const double* x, double* y, size_t n, wx_stream stream);
int wx_euler3d_plan_one_kernel(const wx_euler3d_plan* plan);
wx_status wx_euler3d_plan_set_one_kernel(wx_euler3d_plan* plan, int on);

/* Number of ELEMENTS of dtype in one edge message: 5*V*H*n^2, layout [var][ek][along][n^2] =
 * exactly the reference's q_itf_{s,n,w,e} after ExchangeRequest.wait(). */
size_t wx_euler3d_edge_count(const wx_euler3d_plan* plan);

/* Phases 1-2, sender side (rhs_dfr.py:50-71, 141-172; process_topology.py:269-386):
 * extrapolate q to all element faces (log-space for rho, rho*theta) into the plan's
 * interface buffer, and write the four outward tile-edge faces - rotated into the
 * neighbour's basis and flipped as the reference does before MPI - to send[e]
 * (e = S,N,W,E; each wx_euler3d_edge_count() values).  send[e] is exactly what the neighbour's
 * q_itf_{s,n,w,e} holds after ExchangeRequest.wait(). */
wx_status wx_euler3d_extrap_pack(wx_euler3d_plan* plan, const void* q, void* const send[4], wx_stream stream);

/* Phases 3-8 (rhs_dfr.py:73-139, 203-313): pointwise fluxes, derivatives, Rusanov fluxes
 * with the received halo faces halo[e] (a neighbour's send[] message, receiver-local ordering),
 * corrections, forcing; writes rhs (same layout/dtype as q) for the elements of `region`.
 * Requires wx_euler3d_extrap_pack(plan, q, ...) to have run on the same stream. */
wx_status wx_euler3d_rhs(wx_euler3d_plan* plan, const void* q, const void* const halo[4], void* rhs,
                         wx_region region, wx_stream stream);

/* Same evaluation with the stage update of an explicit Runge-Kutta scheme fused into the store:
 *     out = a*y + b*q + c*R(q)          (y may be NULL: then out = b*q + c*R(q))
 * e.g. the three stages of integrators/tvdrk3.py:12-19 are (y,a,b,c) = (-,0,1,dt), (Q,3/4,1/4,dt/4),
 * (Q,1/3,2/3,2dt/3).  q is already in registers when R is formed, so a stage costs one extra
 * stream (y) instead of three passes over the state.  out must not alias q (neighbours read q's
 * faces through the interface buffer, but the INTERIOR/BOUNDARY launches both read q itself). */
wx_status wx_euler3d_rhs_axpy(wx_euler3d_plan* plan, const void* q, const void* const halo[4], const void* y, void* out,
                              double a, double b, double c, wx_region region, wx_stream stream);
/* ... and with a second array:  out = a*y + b*q + c*R(q) + d*z.  With q = Q + eps*v, y = v, z = R(Q),
 * (a,b,c,d) = (1, 0, -dt/(2 eps), +dt/(2 eps)) this is the Rosenbrock operator of solvers/matvec.py:76-88
 * (matvec_rat) in one launch; (0, 0, dt/eps, -dt/eps) is the finite-difference matvec_fun (:62-66). */
wx_status wx_euler3d_rhs_axpy2(wx_euler3d_plan* plan, const void* q, const void* const halo[4], const void* y,
                               const void* z, void* out, double a, double b, double c, double d, wx_region region,
                               wx_stream stream);

/* Finite-difference Jacobian products (solvers/matvec.py:62-66, 76-88) without materialising Q + eps v: the
 * two kernels of a WX_F64 plan evaluate on the shifted state  q + eps * v  formed on load;
 *   out = a*y + b*(q + eps v) + c*R(q + eps v) + d*z.
 * matvec_fun("fd"):  a = -dt/eps, y = R(q), b = 0, c = dt/eps;   matvec_rat: additionally d = 1, z = v, c = -dt/(2 eps). */
wx_status wx_euler3d_shifted_extrap_pack(wx_euler3d_plan* plan, const double* q, const double* v, double eps,
                                         void* const send[4], wx_stream stream);
wx_status wx_euler3d_shifted_rhs_axpy2(wx_euler3d_plan* plan, const double* q, const double* v, double eps,
                                       const void* const halo[4], const double* y, const double* z, double* out, double a,
                                       double b, double c, double d, wx_region region, wx_stream stream);

/* Stage pipeline for explicit Runge-Kutta loops.  The plan owns two interface buffers (slots 0, 1).
 * wx_euler3d_stage evaluates  out = a*y + b*q + c*R(q) + d*z  reading q's faces from slot `itf_in`
 * and - when prepare_next != 0 - extrapolates `out` (the next stage's state, still in registers) to
 * the element faces into the OTHER slot and packs its tile-edge faces into next_send[e]: the next
 * stage then needs no wx_euler3d_extrap_pack (one read of Q and a launch saved per stage).
 * next_send must not be the buffers the current stage's halos alias.
 * prepare_next == 2 additionally applies the per-step exponential filter (operators.apply_filter_3d; the nodal
 * 1-D matrix is given once by wx_euler3d_set_exp_filter: host pointer, n x n row-major) to `out` before it is
 * stored and extrapolated - the last stage of a step then produces the filtered new state - and raises *nan_flag
 * (device int, nullable) when the stored values hold a NaN: simulation.py:147-155 in one kernel.
 * wx_euler3d_extrap_pack_slot is wx_euler3d_extrap_pack into a chosen slot (pipeline start-up). */
wx_status wx_euler3d_extrap_pack_slot(wx_euler3d_plan* plan, const void* q, void* const send[4], int slot,
                                      wx_stream stream);
wx_status wx_euler3d_set_exp_filter(wx_euler3d_plan* plan, const double* filter);
wx_status wx_euler3d_stage(wx_euler3d_plan* plan, const void* q, const void* const halo[4], const void* y, const void* z,
                           void* out, double a, double b, double c, double d, wx_region region, int itf_in,
                           void* const next_send[4], int prepare_next, int* nan_flag, wx_stream stream);

/* All tiles of a rank in ONE launch per phase (for small tiles an evaluation is launch-bound: 2 launches instead of
 * 2 per tile).  The plans must agree in n, H, V, dtype and case; send[i] / halo[i] are the persistent edge buffers of
 * tile i (they are static: the exchange owns them); the states of the tiles are consecutive slices, panel_stride
 * elements apart, of one stacked array (q, v, y, z, out alike; v, y, z nullable).  axpy = 0: out = R(.);
 * axpy = 1: out = a*y + b*(.) + c*R(.) + d*z.  With v: a WX_F64 batch evaluates on the shifted state q + eps v
 * (wx_euler3d_shifted_*); a WX_DUAL128 batch forms the dual state (q, eps v) from the two REAL arrays
 * (wx_euler3d_jvp_extrap_pack / wx_euler3d_jvp; panel_stride then counts doubles).  The plans must outlive the batch. */
typedef struct wx_euler3d_batch wx_euler3d_batch;
wx_status wx_euler3d_batch_create(wx_euler3d_batch** out, wx_euler3d_plan* const* plans, int count, void* const (*send)[4],
                                  const void* const (*halo)[4]);
/* 1 when the batch evaluates in the one-kernel form AND every tile's four neighbours are tiles of this batch (one rank owns the
 * sphere: each halo line handed to wx_euler3d_batch_create IS another tile's send line): wx_euler3d_batch_extrap_pack then launches
 * nothing - the evaluation forms a tile-edge state from the neighbour tile's nodal values itself, the sender's extrapolation,
 * rotation and flip (process_topology.py:269-386) - and R(Q) of the whole sphere is ONE launch.  Taken by default for spheres of
 * at most 262 144 points (launch-bound: the sizes of the shipped .ini files); WXHIP_BRICK_PULLS=1 / 0 (read at batch creation)
 * forces it on / off. */
int wx_euler3d_batch_pulls(const wx_euler3d_batch* batch);
wx_status wx_euler3d_batch_destroy(wx_euler3d_batch* batch);
wx_status wx_euler3d_batch_extrap_pack(const wx_euler3d_batch* batch, const void* q, const double* v, double eps,
                                       size_t panel_stride, wx_stream stream);
wx_status wx_euler3d_batch_rhs_axpy2(const wx_euler3d_batch* batch, const void* q, const double* v, double eps, const void* y,
                                     const void* z, void* out, size_t panel_stride, int axpy, double a, double b, double c,
                                     double d, wx_region region, wx_stream stream);
wx_status wx_euler3d_batch_jvp(const wx_euler3d_batch* batch, const double* q, const double* v, double eps, double* out,
                               double scale, size_t panel_stride, wx_region region, wx_stream stream);
/* One Krylov vector of KIOPS with the complex-step Jacobian from one host call (solvers/kiops.py:170-207 with
 * solvers/matvec.py:56-61): aw = scale Im R(q + i eps V[j-1][:n]) through the two batched JVP launches, then
 * wx_kiops_finish on row j (see there for V, uflip, hcol, workspace).  WX_DUAL128 batch of a rank that owns the whole
 * sphere (no exchange between the launches); n = tiles x panel_stride; aw: n doubles of scratch on the device. */
wx_status wx_euler3d_batch_kiops_vector(const wx_euler3d_batch* b, const double* q, double* V, size_t ldv, int j, size_t n,
                                        int p, int iop, double eps, double scale, const double* uflip, double* hcol,
                                        double* aw, double* workspace, size_t panel_stride, wx_stream stream);

/* Complex-step Jacobian-vector product (solvers/matvec.py:56-61) with no complex array in HBM.
 * The plan must be WX_DUAL128.  q and v are REAL (n-double) arrays in the state layout; the kernels form
 * the dual state (q, eps*v) on load, exchange dual faces as usual (send/halo buffers are those of a
 * WX_DUAL128 plan: wx_euler3d_edge_count() 16-byte values) and store only  out = scale * Im R(q + i eps v)
 * as a REAL array - e.g. scale = dt/eps gives matvec_fun's result directly. */
wx_status wx_euler3d_jvp_extrap_pack(wx_euler3d_plan* plan, const double* q, const double* v, double eps,
                                     void* const send[4], wx_stream stream);
wx_status wx_euler3d_jvp(wx_euler3d_plan* plan, const double* q, const double* v, double eps, const void* const halo[4],
                         double* out, double scale, wx_region region, wx_stream stream);
/* Prepared complex-step JVP: one linearisation state q, many products (every matvec of an FGMRES / KIOPS solve,
 * solvers/matvec.py:56-61 called from solvers/fgmres.py:150-200 and solvers/kiops.py:170-207).  WX_DUAL128 plans.
 *   wx_euler3d_jvp_prepare              face VALUES of q into the plan's cache (exactly the float64 extrapolation) and the
 *                                       value edge messages into send_val (REAL, wx_euler3d_edge_count doubles each): the
 *                                       caller exchanges them once and keeps the received value halos
 *   wx_euler3d_jvp_tangent_extrap_pack  per product: only the face TANGENTS of (q, eps v) and the tangent edge messages
 *                                       (REAL); reads v and the two log-extrapolated rows of q
 *   wx_euler3d_jvp_prepared             out = scale * Im R(q + i eps v) from cached values + this product's tangents;
 *                                       halo_val / halo_tan: the four received REAL edge messages of each kind
 * Same results as wx_euler3d_jvp_extrap_pack + wx_euler3d_jvp, bit for bit, with 60 B/point less HBM traffic per
 * product and half the exchange volume. */
wx_status wx_euler3d_jvp_prepare(wx_euler3d_plan* pl, const double* q, void* const send_val[4], wx_stream stream);
wx_status wx_euler3d_jvp_tangent_extrap_pack(wx_euler3d_plan* pl, const double* q, const double* v, double eps,
                                             void* const send_tan[4], wx_stream stream);
/* ... with the tangent CORRECTED IN PLACE first (KIOPS on long vectors: the subtraction and the norm of the previous Krylov
 * vector's incomplete orthogonalisation, solvers/kiops.py:176-207, folded into the kernel that reads the whole vector anyway):
 * v <- v - h[0] s[0] row0 [- h[1] s[1] row1]  (h, s: device memory, s NULL = 1; row1 NULL: one row; the order of
 * wx_kiops_long_b), then extrapolated; |v|^2 is left as wx_euler3d_jvp_workgroups(plan, WX_REGION_ALL) partial sums in
 * `partials` (wx_kiops_long_b_fold_finish sums them).  row0 NULL: the plain call. */
wx_status wx_euler3d_jvp_tangent_extrap_pack_fix(wx_euler3d_plan* pl, const double* q, double* v, double eps,
                                                 void* const send_tan[4], const double* row0, const double* row1,
                                                 const double* h, const double* s, double* partials, wx_stream stream);
wx_status wx_euler3d_jvp_prepared(wx_euler3d_plan* pl, const double* q, const double* v, double eps,
                                  const void* const halo_val[4], const void* const halo_tan[4], double* out, double scale,
                                  wx_region region, wx_stream stream);
/* ... with the store  out = *z_scale * (scale * Im R) + *z_coef * z  (z_scale null: 1; z null: the plain product).  The two
 * coefficients are read from DEVICE memory by the kernel.  KIOPS (solvers/kiops.py:170-176: V[j] = A V[j-1] + u a) forms the
 * n-long part of its next Krylov vector here, in the product's own store, instead of in a sweep of its own
 * (wx_kiops_long_a_formed then only takes the products). */
/* With row0 (row1 nullable) and partials, the launch also leaves the products <row_r, out> of the vector it stored, as one
 * pair of partial sums per workgroup: partials[2 * w + r], w < wx_euler3d_jvp_workgroups(pl, region) - the iop = 2 products
 * of KIOPS' incomplete orthogonalisation (solvers/kiops.py:178-186) without a sweep of their own; wx_kiops_long_a_finish
 * sums the partials of all launches of a product. */
wx_status wx_euler3d_jvp_prepared_axpy(wx_euler3d_plan* pl, const double* q, const double* v, double eps,
                                       const void* const halo_val[4], const void* const halo_tan[4], double* out, double scale,
                                       const double* z, const double* z_scale, const double* z_coef, const double* row0,
                                       const double* row1, double* partials, wx_region region, wx_stream stream);
size_t wx_euler3d_jvp_workgroups(const wx_euler3d_plan* pl, wx_region region);

/* ------------------------------------------------------------------------------------------
 * Shallow water on a cubed-sphere tile.
 * Replaces  rhs/rhs_sw.py:38-240 (RhsShallowWater.__call__ / __compute_rhs__).
 * Layout: q[var][ej][ei][p], p = jl*n + il (geometry/cubed_sphere_2d.py:134-170); variables
 * (h, h u1, h u2); halo-padded interface arrays (H, H+2, 2n) / (H+2, H, 2n).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const double* sqrtG;             /* (H,H,n^2)    geometry/metric2d.py:17-167           */
    const double* H_contra_11;       /* (H,H,n^2)                                          */
    const double* H_contra_12;
    const double* H_contra_21;
    const double* H_contra_22;
    const double* christoffel_1_01;  /* (H,H,n^2) the eight symbols rhs_sw.py:223-235 reads */
    const double* christoffel_1_02;
    const double* christoffel_1_11;
    const double* christoffel_1_12;
    const double* christoffel_2_01;
    const double* christoffel_2_02;
    const double* christoffel_2_12;
    const double* christoffel_2_22;
    const double* sqrtG_itf_i;       /* (H,H+2,2n)                                         */
    const double* sqrtG_itf_j;       /* (H+2,H,2n)                                         */
    const double* H_contra_11_itf_i; /* (H,H+2,2n)                                         */
    const double* H_contra_21_itf_i;
    const double* H_contra_12_itf_j; /* (H+2,H,2n)                                         */
    const double* H_contra_22_itf_j;
    /* bottom topography (init Topo: rhs_sw.py:80-82, 153-155, 213-220); all five NULL when flat */
    const double* hsurf;             /* (H,H,n^2)  */
    const double* dzdx1;
    const double* dzdx2;
    const double* hsurf_itf_i;       /* (H,H+2,2n) */
    const double* hsurf_itf_j;       /* (H+2,H,2n) */
    const double* boundary_sn;       /* (H*n) tan(x1) along S/N edges: geom.boundary_sn */
    const double* boundary_we;       /* (H*n) tan(x2) along W/E edges: geom.boundary_we */
} wx_sw_metric;

typedef struct wx_sw_plan wx_sw_plan;

wx_status wx_sw_plan_create(wx_sw_plan** plan, int n, int H, wx_dtype dtype, int panel, const wx_dfr_ops* ops,
                            const wx_sw_metric* metric);
wx_status wx_sw_plan_create_tile(wx_sw_plan** plan, int n, int H, wx_dtype dtype, int panel, const int on_panel_edge[4],
                                 const wx_dfr_ops* ops, const wx_sw_metric* metric);
wx_status wx_sw_plan_destroy(wx_sw_plan* plan);
/* Elements of dtype per edge message: 3*H*n, layout [var][along][n]: exactly what the reference's
 * exchange delivers (rhs_sw.py:103-117, 138-150): h (+ surface height), then the rotated (hu1, hu2). */
size_t wx_sw_edge_count(const wx_sw_plan* plan);
wx_dtype wx_sw_plan_dtype(const wx_sw_plan* plan);
/* rhs_sw.py:76-117 sender side: extrapolate (h+hsurf, hu1, hu2) to the element faces into the plan's
 * interface buffer; rotate + flip the four tile-edge lines into send[e] (e = S,N,W,E). */
wx_status wx_sw_extrap_pack(wx_sw_plan* plan, const void* q, void* const send[4], wx_stream stream);
/* rhs_sw.py:119-240: fluxes, derivatives, AUSM common fluxes with the received halo lines,
 * corrections, Coriolis/metric/topography forcing; writes rhs for the elements of `region`. */
wx_status wx_sw_rhs(wx_sw_plan* plan, const void* q, const void* const halo[4], void* rhs, wx_region region,
                    wx_stream stream);

/* out = a*y + b*q + c*R(q): explicit Runge-Kutta stage fused into the store (see wx_euler3d_rhs_axpy) */
wx_status wx_sw_rhs_axpy(wx_sw_plan* plan, const void* q, const void* const halo[4], const void* y, void* out, double a,
                         double b, double c, wx_region region, wx_stream stream);

/* Several tiles of one rank (e.g. all six panels on one GPU) in ONE launch per phase: the shallow-water
 * workload is launch-latency bound (5.5 MB of state per panel at n=8, H=60).  The states / results are
 * slices q + i*panel_stride of one stacked array (panel_stride in elements of dtype); the edge buffers
 * send[i][e] / halo[i][e] are fixed at creation (they are the exchange's persistent buffers). */
typedef struct wx_sw_batch wx_sw_batch;
wx_status wx_sw_batch_create(wx_sw_batch** batch, wx_sw_plan* const plans[], int count, void* const send[][4],
                             const void* const halo[][4]);
wx_status wx_sw_batch_destroy(wx_sw_batch* batch);
wx_status wx_sw_batch_extrap_pack(wx_sw_batch* batch, const void* q, size_t panel_stride, wx_stream stream);
wx_status wx_sw_batch_rhs(wx_sw_batch* batch, const void* q, void* rhs, size_t panel_stride, wx_region region,
                          wx_stream stream);
/* y (nullable) and out are stacked like q */
wx_status wx_sw_batch_rhs_axpy(wx_sw_batch* batch, const void* q, const void* y, void* out, size_t panel_stride, double a,
                               double b, double c, wx_region region, wx_stream stream);

/* Stage pipeline for explicit Runge-Kutta loops (integrators/tvdrk3.py:12-19 over rhs/rhs_sw.py:38-240), the twin of
 * wx_euler3d_stage: the plan owns two interface buffers (slot 1 after wx_sw_plan_reserve(plan, WX_RESERVE_STAGE): setup
 * time).  wx_sw_stage evaluates  out = a*y + b*q + c*R(q)  reading q's faces from slot itf_in and - prepare_next != 0 -
 * extrapolates `out` (the next stage's state, still in registers) to the element faces of the OTHER slot and packs its
 * tile-edge lines into next_send[e]: the next stage needs no wx_sw_extrap_pack (at the benchmark's S7 size the
 * extrapolation launch is a fifth of an evaluation).  next_send must not be the buffers the current halos alias.
 * wx_sw_extrap_pack_slot: wx_sw_extrap_pack into a chosen slot (pipeline start-up).
 * Batches: wx_sw_batch_create_pipelined fixes both edge-buffer sets of every tile (send / halo: slot 0, send2 / halo2:
 * slot 1; the plans must have been reserved); wx_sw_batch_stage reads slot itf_in and prepares the other one. */
/* The direct form of the evaluation (no interface buffer): the RHS kernel extrapolates the own face states from the element's
 * nodal values and the neighbour's from the NEIGHBOUR ELEMENT's nodal values in memory; only the tile-edge lines are packed
 * beforehand - wx_sw_extrap_pack_ring: the ring of elements on the four tile edges - and exchanged.  One launch instead of
 * the extrapolation + RHS pair, no 36 B/point round trip of face values; same results term by term.
 * wx_sw_rhs_direct: axpy = 0: out = R(q); axpy != 0: out = a*y + b*q + c*R(q) (y nullable).  Batches alike. */
wx_status wx_sw_extrap_pack_ring(wx_sw_plan* plan, const void* q, void* const send[4], wx_stream stream);
wx_status wx_sw_rhs_direct(wx_sw_plan* plan, const void* q, const void* const halo[4], const void* y, void* out, double a,
                           double b, double c, int axpy, wx_region region, wx_stream stream);
wx_status wx_sw_batch_extrap_pack_ring(wx_sw_batch* batch, const void* q, size_t panel_stride, wx_stream stream);
wx_status wx_sw_batch_rhs_direct(wx_sw_batch* batch, const void* q, const void* y, void* out, size_t panel_stride, double a,
                                 double b, double c, int axpy, wx_region region, wx_stream stream);
/* *pulls = 1: every halo line of every tile of the batch IS the send line of a tile of the batch (same-rank neighbours: found by
 * the addresses given to wx_sw_batch_create) - wx_sw_batch_rhs_direct then forms the tile-edge lines itself from the neighbour
 * tiles' nodal values (sum, rotation and flip as the ring pack: rhs_sw.py:76-117, process_topology.py:318-384) and
 * wx_sw_batch_extrap_pack_ring need not be called; 0: call it (and exchange) first.  The stacked state must then hold the tiles in
 * the order of `plans`, panel_stride apart. */
wx_status wx_sw_batch_direct_pulls(const wx_sw_batch* batch, int* pulls);
wx_status wx_sw_plan_reserve(wx_sw_plan* plan, int what);
wx_status wx_sw_extrap_pack_slot(wx_sw_plan* plan, const void* q, void* const send[4], int slot, wx_stream stream);
wx_status wx_sw_stage(wx_sw_plan* plan, const void* q, const void* const halo[4], const void* y, void* out, double a,
                      double b, double c, wx_region region, int itf_in, void* const next_send[4], int prepare_next,
                      wx_stream stream);
wx_status wx_sw_batch_create_pipelined(wx_sw_batch** batch, wx_sw_plan* const plans[], int count, void* const send[][4],
                                       const void* const halo[][4], void* const send2[][4], const void* const halo2[][4]);
wx_status wx_sw_batch_extrap_pack_slot(wx_sw_batch* batch, const void* q, size_t panel_stride, int slot, wx_stream stream);
wx_status wx_sw_batch_stage(wx_sw_batch* batch, const void* q, const void* y, void* out, size_t panel_stride, double a,
                            double b, double c, wx_region region, int itf_in, int prepare_next, wx_stream stream);

/* ------------------------------------------------------------------------------------------
 * Panel-edge halo exchange over RCCL point-to-point (xGMI inside a node).
 * Replaces  process_topology.py:259-261 (Create_dist_graph_adjacent), :269-386 (start_exchange_scalars / _vectors:
 *           rotate, flip, pack, device.synchronize(), Ineighbor_alltoall) and :564-606 (ExchangeRequest.wait).
 * Rotation, flip and packing are done by the extrapolation kernels (wx_*_extrap_pack write exactly what the neighbour's
 * q_itf_{s,n,w,e} holds after wait()); what remains is movement: ONE message per tile edge (the reference sends three),
 * every message of a rank in one ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd on a communication stream forked
 * from the compute stream by an event - no device synchronisation - and joined again by wx_exchange_wait, the elements
 * that need no halo being evaluated in between (rhs/rhs.py:88-118).  A message between two tiles of one rank never moves:
 * the receiver's halo pointer is the sender's slot.  The fork / join records into a HIP-graph capture of the compute
 * stream like any other launch, so an evaluation WITH the exchange in flight beside the interior launch replays from one
 * graph (BASELINE config 5).
 *
 * wx_comm: an RCCL communicator.  Either the caller's own ncclComm_t (wx_comm_adopt: borrowed, never destroyed here), or
 *   one this library makes: rank 0 calls wx_comm_unique_id, hands the 128 bytes to every rank by whatever means it has
 *   (MPI_Bcast of the reference's communicator, torch.distributed, a file), and every rank calls wx_comm_init_rank with
 *   ITS GPU current (collective, blocking; setup time).
 * wx_exchange: the tile graph of the 6 k^2-tile decomposition (process_topology.py:69-125: neighbour table, the tile
 *   across each panel edge, the edge a message lands on), tile ownership (contiguous equal runs of tiles per rank where
 *   that divides, round-robin otherwise; ranks beyond the tile count own nothing but take part), and the two edge
 *   buffers of this rank: messages are laid out by destination rank, inside a rank pair by (destination tile,
 *   destination edge).  edge_doubles: float64 words per edge message (wx_euler3d_edge_count / wx_sw_edge_count, doubled
 *   for 16-byte dtypes).  loopback != 0: same-rank messages go through the communicator too (a one-GPU rehearsal of the
 *   several-GPU path).  comm may be NULL when nothing travels (world == 1, no loopback) - or for layout queries only:
 *   wx_exchange_start then refuses.  Creation is host-only.
 *     wx_exchange_bind        gives the exchange its buffers: the caller's (wx_exchange_send_doubles / _recv_doubles
 *                             doubles each, device memory, alive as long as the exchange) or - both NULL - the library's
 *                             own (hipMalloc, zeroed: the only allocation, at setup time)
 *     wx_exchange_send_ptr    where wx_*_extrap_pack must write edge e (S, N, W, E) of tile t: pass as send[e]
 *     wx_exchange_halo_ptr    where the message for edge e of tile t is found after wx_exchange_wait: pass as halo[e]
 *     wx_exchange_start       enqueue the exchange of everything packed on `compute` so far.  comm_stream != compute:
 *                             event fork, the group on comm_stream, event record for the join (returns at once: the
 *                             caller launches the WX_REGION_INTERIOR work on `compute` now).  comm_stream NULL or ==
 *                             compute: the group in stream order on `compute`, nothing to wait for.
 *     wx_exchange_wait        make `compute` wait for the halos (a stream wait, not a host wait)
 *     wx_exchange_fork / _join  the mirror arrangement: `side` is forked off `compute` by an event (fork), takes the
 *                             WX_REGION_INTERIOR launches, and is joined into `compute` again (join), while the exchange
 *                             (wx_exchange_start(ex, compute, compute)) and the BOUNDARY launches stay on `compute`.
 *   Nothing travels (wx_exchange_needs_comm == 0): start and wait are no-ops.  One exchange in flight per object.
 *   STREAM CAPTURE.  Both arrangements record into a HIP-graph capture of `compute` on ROCm 7.2 (HIP 7.2, RCCL 2.27:
 *   tools/rccl_capture_probe.c).  The HIP 7.0.2 runtime that ships inside torch 2.10 wheels survives an RCCL launch in a
 *   capture only on the capture's ORIGIN stream: hipStreamWaitEvent there lets every non-origin stream that waits on a
 *   captured event join again (parent + parallel-stream list), RCCL forks and joins its own internal stream around each
 *   launch, so a user stream that is itself forked and RCCL's stream end up in each other's lists and
 *   hip::Stream::EndCapture - which walks those lists before it clears them - recurses until the stack is gone
 *   (profiles/r04_capture_crash.md).  There, capture the fork / join arrangement with `compute` = the origin stream
 *   (wx_*_rhs_overlapped does exactly that), or the stream-ordered form.
 * wx_euler3d_rhs_overlapped / wx_sw_rhs_overlapped: the whole evaluation of a rank from one call - pack every local tile on
 *   `compute`, fork `side` for the INTERIOR launches, the exchange and then the BOUNDARY launches on `compute`, join (one
 *   WX_REGION_ALL launch per tile when nothing travels; side NULL or == compute: everything in stream order);
 *   plans[i], q[i], rhs[i] belong to the i-th tile of wx_exchange_local_tiles.
 * ------------------------------------------------------------------------------------------ */
#define WX_COMM_ID_BYTES 128
typedef struct wx_comm wx_comm;
typedef struct wx_exchange wx_exchange;
/* ncclGetVersion (e.g. 22606), < 0 on error */
int wx_comm_rccl_version(void);
wx_status wx_comm_unique_id(unsigned char id[WX_COMM_ID_BYTES]);
wx_status wx_comm_init_rank(wx_comm** comm, int nranks, const unsigned char id[WX_COMM_ID_BYTES], int rank);
wx_status wx_comm_adopt(wx_comm** comm, void* nccl_comm /* ncclComm_t */, int nranks, int rank);
/* WX_ERR_INVALID while an exchange made on it is alive (wx_exchange_destroy them first: an exchange keeps its communicator
 * and a later wx_exchange_start on a destroyed one would use freed memory) */
wx_status wx_comm_destroy(wx_comm* comm);
/* exchanges alive on this communicator (wx_exchange_create counts up, wx_exchange_destroy down); < 0: null */
int wx_comm_users(const wx_comm* comm);
/* The small reductions of the Krylov callers on the SAME communicator as the halo exchange - reference
 * solvers/global_operations.py:14-36 (global_norm / global_dotprod / global_inf_norm: MPI allreduce), solvers/kiops.py:165-200
 * and solvers/pmex.py:150-173 (the products of a new Krylov vector with the basis), solvers/fgmres.py:41 (the one
 * reduction of the low-synchronisation Gram-Schmidt), simulation.py:399-408 (the NaN flag, MAX).  In place on `count`
 * doubles of device memory, ncclAllReduce enqueued on `stream` - the caller's compute stream, so inside a HIP-graph capture
 * the reduction becomes a node of the graph on the capture's origin stream (see STREAM CAPTURE above) and a whole Krylov
 * pass - matvec, exchange, both reductions per vector - replays from one graph on every rank.  No host synchronisation. */
typedef enum wx_reduce_op { WX_REDUCE_SUM = 0, WX_REDUCE_MAX = 1, WX_REDUCE_MIN = 2 } wx_reduce_op;
wx_status wx_comm_allreduce(wx_comm* comm, double* buf, size_t count, wx_reduce_op op, wx_stream stream);
/* versions this process actually BOUND to (the library is built with the image's hipcc but runs on whatever libamdhip64 /
 * librccl the process loaded first - inside torch: the wheel's): hipRuntimeGetVersion, hipDriverGetVersion; < 0 on error */
int wx_hip_runtime_version(void);
int wx_hip_driver_version(void);

wx_status wx_exchange_create(wx_exchange** exchange, wx_comm* comm, int rank, int world, int tiles_per_side,
                             size_t edge_doubles, int loopback);
wx_status wx_exchange_destroy(wx_exchange* exchange);
/* number of tiles this rank owns; the first `capacity` ids (ascending) into tiles (nullable) */
int wx_exchange_local_tiles(const wx_exchange* exchange, int* tiles, int capacity);
/* the tile across edge `edge` (S, N, W, E = 0..3) of `tile`, the edge of that tile the message lands on, and the rank
 * that owns it (process_topology.py:105-125, 259-261); outputs nullable */
wx_status wx_exchange_neighbor(const wx_exchange* exchange, int tile, int edge, int* neighbor_tile, int* landing_edge,
                               int* neighbor_rank);
int wx_exchange_needs_comm(const wx_exchange* exchange);
size_t wx_exchange_send_doubles(const wx_exchange* exchange);
size_t wx_exchange_recv_doubles(const wx_exchange* exchange);
/* doubles sent to / received from every rank per exchange [host arrays of `world` entries, nullable] */
wx_status wx_exchange_peer_counts(const wx_exchange* exchange, size_t* send_doubles, size_t* recv_doubles);
wx_status wx_exchange_bind(wx_exchange* exchange, double* send_buf, double* recv_buf);
void* wx_exchange_send_ptr(const wx_exchange* exchange, int tile, int edge);
const void* wx_exchange_halo_ptr(const wx_exchange* exchange, int tile, int edge);
/* timer != NULL: the *_rhs_overlapped calls on this exchange stamp the reference's nine-slot timing row on the compute
 * stream (rhs/rhs.py:88-118): 0 start, 1 packed, 2 / 3 around the INTERIOR launches (on the stream that carries them),
 * 5 halos received, 8 BOUNDARY / ALL launches done and the streams joined; read it with wx_phase_timer_elapsed.
 * NULL: no stamps (default).  The timer is borrowed. */
wx_status wx_exchange_set_timer(wx_exchange* exchange, wx_phase_timer* timer);
wx_status wx_exchange_start(wx_exchange* exchange, wx_stream compute, wx_stream comm_stream);
wx_status wx_exchange_wait(wx_exchange* exchange, wx_stream compute);
wx_status wx_exchange_fork(wx_exchange* exchange, wx_stream compute, wx_stream side);
wx_status wx_exchange_join(wx_exchange* exchange, wx_stream compute, wx_stream side);
wx_status wx_euler3d_rhs_overlapped(wx_euler3d_plan* const plans[], int count, wx_exchange* exchange, const void* const q[],
                                    void* const rhs[], wx_stream compute, wx_stream side);
wx_status wx_sw_rhs_overlapped(wx_sw_plan* const plans[], int count, wx_exchange* exchange, const void* const q[],
                               void* const rhs[], wx_stream compute, wx_stream side);

/* ------------------------------------------------------------------------------------------
 * The reference's compiled `pde` module, function for function (pde/interface.cpp:282-302,
 * pde/interface.cu:433-442): same argument order and array layouts (C-contiguous, variable-major
 * q[var][elem_x3][elem_x1][pt]); the dtype that pybind11 / the CUDA dispatcher derives from the
 * array objects is an explicit argument here.  Unlike interface.cu:190-210, 328-350 an unknown
 * dtype is an error, not a silent no-op.
 * ------------------------------------------------------------------------------------------ */
/* pde/kernels/pointwise_flux.hpp:3-31 via interface.cpp:11-47 */
wx_status wx_pointwise_eulercartesian_2d(const void* q, void* flux_x1, void* flux_x3, int num_elem_x1, int num_elem_x3,
                                         int num_solpts_tot, wx_dtype dtype, wx_stream stream);
/* pde/kernels/riemann_flux.hpp:5-80 + boundary_flux.hpp:3-24 via interface.cpp:127-238
 * (interface arrays (4, nz, nx, 2n); entries the reference never writes are left untouched) */
wx_status wx_riemann_eulercartesian_ausm_2d(const void* q_itf_x1, const void* q_itf_x3, void* flux_itf_x1,
                                            void* flux_itf_x3, int num_elem_x1, int num_elem_x3, int num_solpts,
                                            wx_dtype dtype, wx_stream stream);
/* pde/kernels/forcing.hpp:6-100 via interface.cpp:241-279: q (5,.), pressure, sqrt_g (unused, as in the
 * reference), h (9,.), christoffel (27,.) -> forcing rows rho_u, rho_v, rho_w (rows rho, rho_theta untouched) */
wx_status wx_forcing_euler_cubesphere_3d(const void* q, const void* pressure, const double* sqrt_g, const double* h,
                                         const double* christoffel, void* forcing, int num_elem_x1, int num_elem_x2,
                                         int num_elem_x3, int num_solpts, wx_dtype dtype, wx_stream stream);

/* The whole 2-D Cartesian Euler RHS in one launch: rhs/rhs_dfr.py:8-45 (RHSDirecFluxReconstruction)
 * with pde/pde_euler_cartesian.py:24-48: extrapolate, pointwise fluxes, derivative, AUSM + solid walls,
 * correction, scale by -2/dx, gravity.  q, rhs: (4, nz, nx, n^2). */
typedef struct wx_cart2d_plan wx_cart2d_plan;
wx_status wx_cart2d_plan_create(wx_cart2d_plan** plan, int n, int num_elem_x1, int num_elem_x3, double dx1, double dx3,
                                wx_dtype dtype, const wx_dfr_ops* ops);
wx_status wx_cart2d_plan_destroy(wx_cart2d_plan* plan);
wx_status wx_cart2d_rhs(wx_cart2d_plan* plan, const void* q, void* rhs, wx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Per-step filters of the explicit time loop (simulation/simulation.py:147-155).
 *
 * wx_expfilter_*: DFROperators.apply_filter_3d (geometry/operators.py:114-119, 257-261),
 *   out = ((sqrtG * q) @ (Fx Fy Fz)) * (1 / sqrtG)  with the nodal 1-D exponential modal filter F (n x n,
 *   row-major, host pointer at creation: operators.make_filter, :208-233) applied along the three local
 *   axes of every element.  q, out: (nvar <= 5, nelem, n^3) of dtype; out may be q (in place).
 *   sqrtG: (nelem, n^3) float64 on the device.  nan_flag (device int, nullable) is set to 1 when the result
 *   holds a NaN - simulation._check_for_nan (:399-408) without a second pass over the state.
 * wx_check_nan: the same flag for a state that is not filtered; count = number of dtype scalars.
 * wx_cart2d_sponge: rho_w *= 1 / (1 + beta dt)  (operators.py:242-253; beta float64 on the device).
 * The flag is only ever raised, never cleared: the caller zeroes it. */
typedef struct wx_expfilter wx_expfilter;
wx_status wx_expfilter_create(wx_expfilter** out, int n, const double* filter);
wx_status wx_expfilter_destroy(wx_expfilter* h);
/* 1 when wx_expfilter_apply* of this dtype runs its three passes on the matrix cores (see wx_euler3d_uses_matrix_cores) */
int wx_expfilter_uses_matrix_cores(const wx_expfilter* h, wx_dtype dtype);
wx_status wx_expfilter_apply(const wx_expfilter* h, const void* q, void* out, const double* sqrtG, int nvar, size_t nelem,
                             wx_dtype dtype, int* nan_flag, wx_stream stream);
/* The same filter on npanels stacked panels in one launch: q, out (npanels, nvar, nelem, n^3), sqrtG (npanels, nelem, n^3). */
wx_status wx_expfilter_apply_stacked(const wx_expfilter* h, const void* q, void* out, const double* sqrtG, int nvar,
                                     size_t nelem, int npanels, wx_dtype dtype, int* nan_flag, wx_stream stream);
wx_status wx_check_nan(const void* q, size_t count, wx_dtype dtype, int* flag, wx_stream stream);
wx_status wx_cart2d_sponge(void* rho_w, const double* beta, double dt, size_t count, wx_dtype dtype, wx_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Vector kernels of the matrix-free Krylov solvers (solvers/fgmres.py:150-200, solvers/kiops.py:170-200): the
 * Gram-Schmidt passes over a basis of m device vectors of n doubles (rows of V, row stride ldv >= n), each in one
 * pass over the data and without temporaries:
 *   wx_multi_dot   out[k] = <V[k], w>, k < m   (out: m doubles on the device; workspace: wx_multi_dot_workspace(m)
 *                  doubles on the device; deterministic two-stage reduction, no floating-point atomics)
 *   wx_multi_axpy  w -= sum_k h[k] V[k]        (h: m doubles on the device) */
size_t wx_multi_dot_workspace(int m);
wx_status wx_multi_dot(const double* V, size_t ldv, int m, const double* w, size_t n, double* out, double* workspace,
                       wx_stream stream);
wx_status wx_multi_axpy(double* w, const double* V, size_t ldv, int m, const double* h, size_t n, wx_stream stream);
/* ... and w = (w - sum_k h[k] V[k]) * scale in the same pass (the correction and the normalisation of a Krylov vector of
 * solvers/pmex.py:193-233 when its norm is known from the products already). */
wx_status wx_multi_axpy_scaled(double* w, const double* V, size_t ldv, int m, const double* h, size_t n, double scale,
                               wx_stream stream);
/* The augmented update of the phi-function Krylov methods (solvers/kiops.py:170-173, solvers/pmex.py:160-163) for row j of
 * the basis V (rows of n + p doubles, row stride ldv):  V[j, :n] = aw + uflip @ V[j-1, n:n+p]  (uflip: n x p, row-major;
 * aw: the operator's product with V[j-1, :n]),  V[j, n:n+p] = V[j-1, n+1:n+p], 0.  1 <= p <= 16. */
wx_status wx_krylov_aug_update(double* V, size_t ldv, int j, size_t n, int p, const double* aw, const double* uflip,
                               wx_stream stream);
/* One Krylov vector of PMEX (solvers/pmex.py:157-233) from the operator's product `aw` with V[j-1, :n], with no host round
 * trip: the augmented update, the (j+1) x 2 block of products, the coefficients of the reference's low-synchronisation
 * projector (LT, Linv: ld x ld, row-major, persistent over the solve - LT zeros and Linv the identity at its start), the
 * norm from the same reduction (squares accumulated in double-double where the reference uses the platform's extended
 * precision) or, where that difference is negative, the corrected vector's own norm; the correction and the
 * normalisation.  hcol[0 .. j-1] = the projection coefficients, hcol[j] = the norm (below `tol`: a happy breakdown, the
 * vector is left unnormalised and the caller discards what follows), *own = 1 when the own norm was needed.
 * 1 <= j <= mmax <= 128, mmax <= ld, 1 <= p <= 16.  workspace: wx_pmex_workspace(mmax) doubles. */
size_t wx_pmex_workspace(int mmax);
/* ... with the complex-step matvec of a dual batch in front (wx_euler3d_batch_extrap_pack + wx_euler3d_batch_jvp applied to
 * V[j-1, :n]; aw: n doubles of scratch for the product), all from one host call: the PMEX twin of
 * wx_euler3d_batch_kiops_vector, same conditions (the rank owns the whole sphere; n = tiles x panel_stride). */
wx_status wx_euler3d_batch_pmex_vector(const wx_euler3d_batch* b, const double* q, double* V, size_t ldv, int j, size_t n,
                                       int p, double eps, double scale, const double* uflip, double* LT, double* Linv, int ld,
                                       double tol, double* hcol, double* own, double* aw, double* workspace, int mmax,
                                       size_t panel_stride, wx_stream stream);
wx_status wx_pmex_vector(double* V, size_t ldv, int j, size_t n, int p, const double* aw, const double* uflip, double* LT,
                         double* Linv, int ld, double tol, double* hcol, double* own, double* workspace, int mmax,
                         wx_stream stream);
/* ... with the vectors split over ranks (solvers/pmex.py:150-173, 194-218): this rank holds n of the components, the p
 * augmented ones are replicated.  The (j+1) x 2 block of products and the vector's own norm are formed over the n-long parts
 * and completed by wx_comm_allreduce on `comm` in stream order - two graph nodes under capture, no host round trip -, the
 * augmented components enter once afterwards.  comm NULL: no reduction (one rank taking the several-rank code path). */
wx_status wx_pmex_vector_split(double* V, size_t ldv, int j, size_t n, int p, const double* aw, const double* uflip, double* LT,
                               double* Linv, int ld, double tol, double* hcol, double* own, double* workspace, int mmax,
                               wx_comm* comm, wx_stream stream);
/* The low-synchronisation Gram-Schmidt step of solvers/fgmres.py:16-73 (_ortho_1_sync_igs: all rows against the last
 * two in ONE fused reduction, then both rows corrected, scaled and orthogonalised against each other):
 *   wx_multi_dot2   out[k] = <V[k], a>, out[m + k] = <V[k], b>, k < m, one pass over the rows
 *                   (workspace: wx_multi_dot_workspace(2 m) doubles)
 *   wx_pair_update  a -= sum_k ha[k] V[k];  b -= sum_k hb[k] V[k];  a *= scale_a;  b = (b - cross a) * scale_b
 *                   in one pass (m may be 0; ha, hb: m doubles on the device) */
/* One Krylov vector of KIOPS (solvers/kiops.py:170-207) finished in three short launches instead of the seven or more
 * of the array-expression form, for SHORT vectors (launch-bound sizes such as the shipped .ini files): row j of the
 * basis V (rows of n + p doubles, stride ldv), aw = the matvec's output A V[j-1][:n]:
 *   V[j][:n] = aw + uflip (n x p row-major) @ V[j-1][n:];  V[j][n:] = V[j-1][n+1:], 0;
 *   hcol[r] = <V[r], V[j]>, max(0, j - iop) <= r < j;  V[j] -= sum_r hcol[r] V[r];  hcol[j] = |V[j]|;  V[j] /= hcol[j]
 * p <= 16, iop <= 4; workspace: wx_kiops_finish_workspace(n + p) doubles on the device.  Deterministic reductions. */
size_t wx_kiops_finish_workspace(size_t len);
wx_status wx_kiops_finish(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* aw, const double* uflip,
                          double* hcol, double* workspace, wx_stream stream);
/* The same Krylov vector for LONG vectors (the E7 sphere: 442 M doubles), in three streaming stages whose reductions the
 * caller completes in between - over ranks (all-reduce: BASELINE config 5 runs KIOPS on 6 GPUs) and with the p replicated
 * augmented components - 11 vector sweeps per Krylov vector instead of 14 and a tall-skinny gemv (9 with the lazy
 * normalisation of the *_scaled forms below):
 *   wx_kiops_long_a  V[j][:n] = aw + uflip @ V[j-1][n:];  V[j][n:] = V[j-1][n+1:], 0;
 *                    dots[r - ilow] = <V[r][:n], V[j][:n]>,  ilow = max(0, j - iop) <= r < j      (dots: device, iop doubles)
 *   wx_kiops_long_b  V[j][:] -= sum_r h[r - ilow] V[r][:]  (h: device, the completed products);  *nrm2 = |V[j][:n]|^2
 *   wx_kiops_long_c  V[j][:] /= sqrt(*nrm2)  (nrm2: device, the completed squared norm);  hcol[j] = that root
 * workspace: wx_kiops_long_workspace() doubles on the device.  p <= 16, iop <= 4.  Deterministic reductions. */
size_t wx_kiops_long_workspace(void);
wx_status wx_kiops_long_a(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* aw, const double* uflip,
                          double* dots, double* workspace, wx_stream stream);
wx_status wx_kiops_long_b(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* h, double* nrm2,
                          double* workspace, wx_stream stream);
wx_status wx_kiops_long_c(double* V, size_t ldv, int j, size_t n, int p, const double* nrm2, double* hcol, wx_stream stream);
/* The same three stages with LAZY normalisation: the n-long part of a basis row is never rewritten for its norm - row r stays
 * unscaled, scales[r] = 1 / |V[r]| (device, one double per row; the caller sets scales[0] = 1 for the normalised start row) is
 * applied wherever the row is used, and `aw` is A applied to the UNSCALED row j-1.  The p augmented components are kept
 * scaled.  9 vector sweeps per Krylov vector instead of 11.  A linear combination of rows must carry the scales:
 * sum_r c[r] scales[r] V[r][:n]. */
wx_status wx_kiops_long_a_scaled(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* aw, const double* uflip,
                                 double* dots, double* workspace, const double* scales, wx_stream stream);
/* ... when V[j][:n] = scales[j-1] * aw + uflip @ V[j-1][n:] has been formed already (by the matvec's own store,
 * wx_euler3d_jvp_prepared_axpy): the augmented components and the products only - 3 sweeps instead of 5 at iop = 2 */
wx_status wx_kiops_long_a_formed(double* V, size_t ldv, int j, size_t n, int p, int iop, double* dots, double* workspace,
                                 const double* scales, wx_stream stream);
/* ... and when the matvec has left the products too (partials: nblocks pairs, wx_euler3d_jvp_prepared_axpy): the augmented
 * components of row j and dots[r] = scales[ilow + r] * sum_b partials[2 b + r], r < min(iop, j) <= 2 - no sweep at all
 * (workspace: wx_kiops_long_workspace() doubles, as for the other stages) */
wx_status wx_kiops_long_a_finish(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* partials, size_t nblocks,
                                 double* dots, double* workspace, const double* scales, wx_stream stream);
wx_status wx_kiops_long_b_scaled(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* h, double* nrm2,
                                 double* workspace, const double* scales, wx_stream stream);
/* ... when the subtraction and the norm of row j were done by the NEXT product's tangent extrapolation
 * (wx_euler3d_jvp_tangent_extrap_pack_fix: `nblocks` partial squared norms in `partials`): *nrm2 = their sum over the n-long part,
 * the p augmented components of row j corrected with the same coefficients h[0 .. nr-1]; the caller all-reduces nrm2, adds
 * the augmented components' share and calls wx_kiops_long_c_lazy, as after wx_kiops_long_b_scaled. */
wx_status wx_kiops_long_b_fold_finish(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* h,
                                      const double* partials, size_t nblocks, double* nrm2, double* workspace, wx_stream stream);
wx_status wx_kiops_long_c_lazy(double* V, size_t ldv, int j, size_t n, int p, const double* nrm2, double* hcol, double* scales,
                               wx_stream stream);
wx_status wx_multi_dot2(const double* V, size_t ldv, int m, const double* a, const double* b, size_t n, double* out,
                        double* workspace, wx_stream stream);
wx_status wx_pair_update(double* a, double* b, const double* V, size_t ldv, int m, const double* ha, const double* hb,
                         size_t n, double scale_a, double cross, double scale_b, wx_stream stream);
/* fgmres without a host round trip per Krylov vector (solvers/fgmres.py:16-73, 160-210: the lagged one-synchronisation
 * Gram-Schmidt).  wx_fgmres_vector = step J of that scheme on rows a = V[J-2], b = V[J-1]: the 2 J products <V[k], a>, <V[k], b>
 * (all-reduced over `comm` when given), the step's small algebra in a one-wave kernel - Hessenberg column J-2 finished, column
 * J-1 started, the second-pass corrections T, the lagged products K, all (ld x ld) row-major in DEVICE memory - and the update
 * of the two rows with the coefficients that kernel left at `coef` (3 ld doubles).  vn[J-2] = the norm of row J-2 (the scale of
 * the next operator application).  *flag (device int, zeroed by the caller before a pass) becomes J at the first step whose norm
 * estimate falls under the host algorithm's thresholds (a breakdown, a suspect cancellation, a NaN): that step and the later
 * ones of the pass leave the rows untouched, and the caller redoes them on the host.  The host reads R once per PASS of
 * several vectors and finds the iteration the reference would have stopped at from its columns; vectors built past it are
 * discarded.  workspace: wx_fgmres_workspace(ld) doubles.  On one rank and up to 262 144 components the step is three launches
 * (products of all rows, algebra, update of both rows); with WXHIP_FGMRES_ONE_LAUNCH=1 in the environment ONE, its <= 128
 * workgroups meeting at a barrier inside it (never under stream capture): the same bits, the same time - and a launch that needs
 * all its workgroups resident at once, which a GPU shared with other processes does not promise; a workgroup that waits in vain
 * sets *flag = -1.
 * wx_euler3d_batch_fgmres_vector: the same with the finite-difference Rosenbrock operator in front (integrators/ros2.py:27-30,
 * solvers/matvec.py:76-88): row J-1 = A(row J-2 / s) s with s = vn[J-3] read by the kernels from device memory. */
size_t wx_fgmres_workspace(int rows);
/* fgmres' host side of the same scheme (solvers/fgmres.py:75-94, 202-262), plain host code on HOST arrays - the reference does it in
 * the interpreter, which at the shipped .ini sizes was a sixth of a Rosenbrock step once the vectors came from device passes.
 * wx_fgmres_rotate_columns: Hessenberg columns j0 .. j1-1 (column j = R[0 .. j+1][j+1], R row-major ld x ld) through the stored
 * rotations cs / sn[0 .. j), a new rotation each (fgmres.py's _rotg), g updated, Hm[j][0 .. j+1] = the rotated column, res[j - j0]
 * = |g[j+1]|, *rate = the running decay of that estimate (NaN on entry: none yet); stops behind the first column whose estimate is
 * below tol_abs or NaN or whose row norm vn[j+1] is zero (*stopped = 1).  Returns the number of columns taken, -1 on bad arguments.
 * wx_fgmres_back_substitute: y[0 .. k) from the rotated columns and g.  Same IEEE operations in the same order as the interpreted
 * loops (tests/test_solvers.py holds them bit for bit). */
int wx_fgmres_rotate_columns(const double* R, int ld, int j0, int j1, int restart, const double* vn, double* cs, double* sn, double* g,
                             double* Hm, double tol_abs, double* rate, double* res, int* stopped);
int wx_fgmres_back_substitute(const double* Hm, int ld, int k, const double* g, double* y);
wx_status wx_fgmres_vector(double* V, size_t ldv, int J, size_t n, double* R, double* T, double* K, int ld, double* coef,
                           double* vn, int* flag, double* workspace, wx_comm* comm, wx_stream stream);
wx_status wx_euler3d_batch_fgmres_vector(const wx_euler3d_batch* batch, const double* q, const double* rq, double* V, size_t ldv,
                                         int J, size_t n, double eps, double half_dt_over_eps, double* R, double* T, double* K,
                                         int ld, double* coef, double* vn, int* flag, double* workspace, size_t panel_stride,
                                         wx_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* WXHIP_H */
