// 3-D Euler RHS on one cubed-sphere tile: hand-written HIP for gfx950 (MI355X).
//
// Replaces reference wx_factory/rhs/rhs_dfr.py:48-313 + pde/pde_euler_cubesphere.py:72-290 +
// pde/fluxes.py:150-222,326-403,507-582 (phases 1-8 of rhs/rhs.py:75-122).
//
// Two kernels per evaluation (DESIGN.md has the traffic model):
//   K1 euler_extrap_kernel  phase 1-2: Q -> face states of every element (log-space for rho,
//                           rho*theta) into the plan's interface buffer + rotated/flipped
//                           tile-edge faces into the send buffers.
//   K2 euler_rhs_kernel     phases 3-8 fused, one pass over Q, the 35 static metric fields and
//                           the interface buffer; writes R.  Nothing else touches HBM.
// Both are element-blocked: a workgroup owns EPB whole elements, one thread per solution
// point, so every global access is a contiguous n^3 (or n^2) run of doubles.  The dense
// Kronecker operators of the reference become 1-D contractions: on the matrix cores for n = 8 float64
// (wx_mfma.h), staged through LDS on the vector pipe otherwise.
//
// Build-time switches - two, both for measurements, never for behaviour:
//   WX_MFMA     (wx_mfma.h) 1: n = 8 contractions on v_mfma_f64_4x4x4_4b_f64 (default); 0: vector pipe everywhere -
//               the A/B build of profiles/r02_k2_valu_*
//   WX_K2_DIAG  0: the product (default).  Diagnostic builds of the fused kernel, wrong results / extra stores:
//               1 per-workgroup phase stamps (tools/kstamps.py); 2 skeleton: every load, LDS write and store but no
//               arithmetic; 3 without the Riemann arithmetic only; 4 without the three directional passes only; 5 the
//               vertical face states neither stored nor loaded (the ceiling of column-walking designs)
// Everything else that was A/B-tested in rounds 1-2 (LDS swizzles, element orders, load orders, own/neighbour flux
// forms, non-temporal stores, extra face fields, the 16x16x4 MFMA shape ...) is decided and gone: DESIGN.md 4.1
// keeps the measurements.
//
// This file: the plan, the batch and the C ABI.  The device code is in the headers included below, one per kernel family:
//   euler3d_common.h (parameter block, launch shapes, value types), euler3d_extrap.h (K1), euler3d_rhs.h (K2),
//   euler3d_brick.h (the one-kernel form of the low orders), euler3d_jvp.h (the JVP specialisation), euler3d_launch.h
//   (launchers, the column-form entry points).
#include "wx_common.h"
#include "wx_math.h"
#include "wx_mfma.h"
#include "wx_panels.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#ifndef WX_K2_DIAG
#define WX_K2_DIAG 0
#endif


#include "euler3d_common.h"
#include "euler3d_extrap.h"
#include "euler3d_rhs.h"
#include "euler3d_brick.h"
#include "euler3d_jvp.h"
#include "euler3d_brick_jvp.h"
#include "euler3d_launch.h"

// ------------------------------------------------------------------------------------------------
// plan + C ABI
// ------------------------------------------------------------------------------------------------
using namespace wx;

struct wx_euler3d_plan {
    int n, H, V, case_number, panel;
    wx_dtype dtype;
    size_t nelem;
    void* itf;         // device: [elem][6][5][n^2] of dtype (interface slot 0)
    void* itf2 = nullptr;  // interface slot 1, allocated on first use of the stage pipeline
    size_t itf_bytes;
    EulerConsts* consts;  // device
    unsigned long long* stamps = nullptr;  // device, diagnostic builds only
    double* face_val = nullptr;   // prepared JVP: face values of the linearisation state, [elem][6][5][n^2] doubles
    EulerParams<double> base;  // pointer-free parts + metric pointers (q/rhs/halo/send filled per call)
    // column form (wx_euler3d_plan_set_column_metric): the metric of a column-invariant geometry as (n x n) slabs
    // low orders, float64: the one-kernel form (euler3d_brick.h) - the pack entry points write the edge messages only,
    // the evaluation reads no interface buffer.  WXHIP_DIRECT=0 (environment, read at plan creation): the two-kernel form.
    bool direct = false;
    int flip[4] = {0, 0, 0, 0};   // what the pack of edge e does to its line (the device constants' flip[e]) ...
    double rot[4][8];             // ... and its rotation table (the device constants' rot[e])
    bool column = false;
    const double *c_sg = nullptr, *c_h = nullptr, *c_chr = nullptr, *c_idz = nullptr, *c_sgi = nullptr, *c_sgj = nullptr,
                 *c_sgk = nullptr, *c_hi = nullptr, *c_hj = nullptr, *c_hk = nullptr;
};

namespace {

template <typename T>
EulerParams<T> make_params(const wx_euler3d_plan* pl) {
    // EulerParams<double> and <cplx> differ only in pointer value types: copy field by field
    EulerParams<T> P;
    const EulerParams<double>& b = pl->base;
    P.H = b.H; P.V = b.V; P.nelem = b.nelem; P.count = b.count; P.region = b.region;
    P.advection_only = b.advection_only; P.has_damp = b.has_damp; P.rot_zero = b.rot_zero;
    P.md_v = b.md_v; P.md_h = b.md_h; P.md_w = b.md_w; P.md_ring = b.md_ring;
    P.q = nullptr; P.rhs = nullptr; P.itf = static_cast<T*>(pl->itf);
    P.axpy = 0; P.ca = P.cb = P.cd = 0.0; P.cc = 1.0; P.y = nullptr; P.z = nullptr;
    P.itf_out = nullptr; P.nsend_s = P.nsend_n = P.nsend_w = P.nsend_e = nullptr;
    P.efilter = 0; P.nan_flag = nullptr;
    P.jvp = 0; P.q_re = P.q_tan = nullptr; P.out_tan = nullptr; P.jvp_eps = 0.0; P.jvp_scale = 1.0;
    P.dscale = nullptr;
    for (int e = 0; e < 4; ++e) { P.pull_tile[e] = -1; P.pull_edge[e] = 0; P.pull_flip[e] = 0; }
    P.split = 0; P.ft = nullptr; P.fv = nullptr; P.hv_s = P.hv_n = P.hv_w = P.hv_e = nullptr;
    P.halo_s = P.halo_n = P.halo_w = P.halo_e = nullptr;
    P.send_s = P.send_n = P.send_w = P.send_e = nullptr;
    P.K = pl->consts;
    P.stamps = pl->stamps;
    P.jz = nullptr; P.jzs = nullptr; P.jza = nullptr; P.jr0 = nullptr; P.jr1 = nullptr; P.jpart = nullptr;
    P.tc_r0 = nullptr; P.tc_r1 = nullptr; P.tch = nullptr; P.tcs = nullptr; P.tc_out = nullptr; P.tc_part = nullptr;
    P.sg = b.sg; P.h = b.h; P.chr = b.chr; P.idz = b.idz;
    P.sgi = b.sgi; P.sgj = b.sgj; P.sgk = b.sgk; P.hi = b.hi; P.hj = b.hj; P.hk = b.hk;
    P.dcoef = b.dcoef; P.duref = b.duref; P.bsn = b.bsn; P.bwe = b.bwe;
    return P;
}

template <typename T>
wx_status dispatch_extrap(int n, const EulerParams<T>& P, hipStream_t st) {
    switch (n) {
        case 2: return launch_extrap<2, T>(P, st);
        case 3: return launch_extrap<3, T>(P, st);
        case 4: return launch_extrap<4, T>(P, st);
        case 5: return launch_extrap<5, T>(P, st);
        case 6: return launch_extrap<6, T>(P, st);
        case 7: return launch_extrap<7, T>(P, st);
        case 8: return launch_extrap<8, T>(P, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", n);
}

template <typename T>
wx_status dispatch_rhs(int n, const EulerParams<T>& P, hipStream_t st) {
    switch (n) {
        case 2: return launch_rhs<2, T>(P, st);
        case 3: return launch_rhs<3, T>(P, st);
        case 4: return launch_rhs<4, T>(P, st);
        case 5: return launch_rhs<5, T>(P, st);
        case 6: return launch_rhs<6, T>(P, st);
        case 7: return launch_rhs<7, T>(P, st);
        case 8: return launch_rhs<8, T>(P, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", n);
}

int region_count(int region, int H, int V);

bool plan_direct(const wx_euler3d_plan* pl) { return pl->direct && !pl->column; }

// the pack half of an evaluation in the one-kernel form (float64)
wx_status dispatch_pack(int n, const EulerParams<double>& P, hipStream_t st) {
    switch (n) {
        case 2: return launch_pack<2, double>(P, st);
        case 3: return launch_pack<3, double>(P, st);
        case 4: return launch_pack<4, double>(P, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "the one-kernel form serves num_solpts 2..4, not %d", n);
}

wx_status dispatch_brick(int n, const EulerParams<double>& P, bool epi, hipStream_t st) {
    switch (n) {
        case 2: return launch_brick<2>(P, epi, st);
        case 3: return launch_brick<3>(P, epi, st);
        case 4: return launch_brick<4>(P, epi, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "the one-kernel form serves num_solpts 2..4, not %d", n);
}

template <typename T>
wx_status run_extrap(wx_euler3d_plan* pl, const void* q, void* const send[4], hipStream_t st, int slot = 0) {
    EulerParams<T> P = make_params<T>(pl);
    if (slot == 1) P.itf = static_cast<T*>(pl->itf2);
    P.q = static_cast<const T*>(q);
    if (send) {
        P.send_s = static_cast<T*>(send[0]); P.send_n = static_cast<T*>(send[1]);
        P.send_w = static_cast<T*>(send[2]); P.send_e = static_cast<T*>(send[3]);
    }
    if constexpr (std::is_same<T, double>::value) {
        if (plan_direct(pl)) return send ? dispatch_pack(pl->n, P, st) : WX_OK;   // (no interface buffer to fill)
    }
    return dispatch_extrap<T>(pl->n, P, st);
}

template <typename T>
wx_status run_rhs(wx_euler3d_plan* pl, const void* q, const void* const halo[4], void* out, wx_region region,
                  hipStream_t st, int axpy, const void* y, double ca, double cb, double cc, const void* z, double cd,
                  int itf_in = 0, void* const next_send[4] = nullptr, bool epilogue = false, bool efilter = false,
                  int* nan_flag = nullptr) {
    EulerParams<T> P = make_params<T>(pl);
    P.efilter = (epilogue && efilter) ? 1 : 0;
    P.nan_flag = nan_flag;
    if (itf_in == 1) P.itf = static_cast<T*>(pl->itf2);
    if (epilogue) {
        P.itf_out = static_cast<T*>(itf_in == 1 ? pl->itf : pl->itf2);
        if (next_send) {
            P.nsend_s = static_cast<T*>(next_send[0]); P.nsend_n = static_cast<T*>(next_send[1]);
            P.nsend_w = static_cast<T*>(next_send[2]); P.nsend_e = static_cast<T*>(next_send[3]);
        }
    }
    P.q = static_cast<const T*>(q); P.rhs = static_cast<T*>(out);
    P.region = region; P.count = region_count(region, pl->H, pl->V);
    P.axpy = axpy; P.ca = ca; P.cb = cb; P.cc = cc; P.cd = cd;
    P.y = static_cast<const T*>(y); P.z = static_cast<const T*>(z);
    if (halo) {
        P.halo_s = static_cast<const T*>(halo[0]); P.halo_n = static_cast<const T*>(halo[1]);
        P.halo_w = static_cast<const T*>(halo[2]); P.halo_e = static_cast<const T*>(halo[3]);
    }
    if constexpr (std::is_same<T, double>::value) {
        // column form: launches on a plan that holds the column slabs (the plain kernel and the stage pipeline's)
        if (pl->column && P.q_tan == nullptr) {
            P.sg = pl->c_sg; P.h = pl->c_h; P.chr = pl->c_chr; P.idz = pl->c_idz;
            P.sgi = pl->c_sgi; P.sgj = pl->c_sgj; P.sgk = pl->c_sgk; P.hi = pl->c_hi; P.hj = pl->c_hj; P.hk = pl->c_hk;
            return dispatch_rhs_column(pl->n, P, st);
        }
        if (plan_direct(pl)) return dispatch_brick(pl->n, P, epilogue, st);
    }
    return dispatch_rhs<T>(pl->n, P, st);
}

int region_count(int region, int H, int V) {
    const int w = H > 2 ? H - 2 : 0;
    if (region == WX_REGION_ALL) return V * H * H;
    if (region == WX_REGION_INTERIOR) return V * w * w;
    return V * (H * H - w * w);
}

}  // namespace

extern "C" {

wx_status wx_euler3d_plan_create(wx_euler3d_plan** out, int n, int H, int V, int case_number, wx_dtype dtype,
                                 int panel, const wx_dfr_ops* ops, const wx_euler3d_metric* m) {
    static const int all_edges[4] = {1, 1, 1, 1};
    return wx_euler3d_plan_create_tile(out, n, H, V, case_number, dtype, panel, all_edges, ops, m);
}

wx_status wx_euler3d_plan_create_tile(wx_euler3d_plan** out, int n, int H, int V, int case_number, wx_dtype dtype,
                                      int panel, const int on_panel_edge[4], const wx_dfr_ops* ops,
                                      const wx_euler3d_metric* m) {
    if (!on_panel_edge) return fail(WX_ERR_INVALID, "wx_euler3d_plan_create_tile: null on_panel_edge");
    if (!out || !ops || !m) return fail(WX_ERR_INVALID, "wx_euler3d_plan_create: null argument");
    *out = nullptr;
    if (n < 2 || n > kMaxN) return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..%d", n, kMaxN);
    if (H < 1 || V < 1) return fail(WX_ERR_INVALID, "bad tile size H=%d V=%d", H, V);
    if (panel < 0 || panel > 5) return fail(WX_ERR_INVALID, "panel %d not in 0..5", panel);
    if (dtype != WX_F64 && dtype != WX_C128 && dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "unknown dtype %d", (int)dtype);
    if (!ops->extrap_neg || !ops->extrap_pos || !ops->diff_solpt || !ops->correction || !ops->highfilter)
        return fail(WX_ERR_INVALID, "wx_dfr_ops has a null member");
    if (!m->sqrtG || !m->h_contra || !m->christoffel || !m->inv_dzdeta || !m->sqrtG_itf_i || !m->sqrtG_itf_j ||
        !m->sqrtG_itf_k || !m->h_contra_itf_i || !m->h_contra_itf_j || !m->h_contra_itf_k || !m->boundary_sn ||
        !m->boundary_we)
        return fail(WX_ERR_INVALID, "wx_euler3d_metric has a null member");
    const bool damp = (case_number == 21 || case_number == 22);
    if (damp && (!m->damp_coef || !m->damp_uref))
        return fail(WX_ERR_INVALID, "case %d needs damp_coef and damp_uref", case_number);

    wx_euler3d_plan* pl = new (std::nothrow) wx_euler3d_plan();
    if (!pl) return fail(WX_ERR_NOMEM, "out of host memory");
    pl->n = n; pl->H = H; pl->V = V; pl->case_number = case_number; pl->panel = panel; pl->dtype = dtype;
    pl->nelem = (size_t)V * H * H;
    const size_t esz = dtype == WX_F64 ? 8 : 16;
    pl->itf_bytes = pl->nelem * 6 * NQ * n * n * esz;
    hipError_t e = hipMalloc(&pl->itf, pl->itf_bytes);
    if (e != hipSuccess) {
        const size_t want = pl->itf_bytes;
        delete pl;
        return fail(WX_ERR_NOMEM, "hipMalloc(%zu bytes) for the interface buffer failed: %s", want, hipGetErrorString(e));
    }
    EulerParams<double>& b = pl->base;
    b.H = H; b.V = V; b.nelem = (int)pl->nelem; b.count = 0; b.region = 0;
    {   // multipliers of fast_div (exact while slot * d < 2^32)
        auto magic = [&](int d) -> unsigned {
            return (d >= 2 && (unsigned long long)pl->nelem * (unsigned long long)d < (1ull << 32)) ? (unsigned)((1ull << 32) / (unsigned)d) + 1u : 0u;
        };
        b.md_v = magic(V); b.md_h = magic(H); b.md_w = magic(H - 2);
        b.md_ring = magic(H * H - (H > 2 ? (H - 2) * (H - 2) : 0));
    }
    {
        const char* env = getenv("WXHIP_DIRECT");
        // default at num_solpts 2, where it wins (profiles/r06_low_order_forms.txt); 3 and 4 on request (set_one_kernel)
        pl->direct = (dtype == WX_F64 || dtype == WX_DUAL128) && n == 2 && !(env && env[0] == '0');   // (dual plans: the JVP entry points)
    }
    b.advection_only = case_number < 13; b.has_damp = damp;
    b.sg = m->sqrtG; b.h = m->h_contra; b.chr = m->christoffel; b.idz = m->inv_dzdeta;
    b.sgi = m->sqrtG_itf_i; b.sgj = m->sqrtG_itf_j; b.sgk = m->sqrtG_itf_k;
    b.hi = m->h_contra_itf_i; b.hj = m->h_contra_itf_j; b.hk = m->h_contra_itf_k;
    b.dcoef = damp ? m->damp_coef : nullptr; b.duref = damp ? m->damp_uref : nullptr;
    b.bsn = m->boundary_sn; b.bwe = m->boundary_we;
    {   // static-field specialisation: are the nine rotation Christoffel symbols identically zero (cases on a
        // non-rotating planet: DCMIP 2-x, 3-1)?  One pass over them now saves 72 B/point in every evaluation.
        // Plan creation is a SETUP-TIME call: it synchronises the device first (the metric may still be in the making
        // on a non-blocking stream, which the null stream used here does not wait for) and must not be captured.
        (void)hipDeviceSynchronize();
        int* flag = nullptr;
        int any = 1;
        e = hipMalloc((void**)&flag, sizeof(int));
        if (e == hipSuccess) e = hipMemset(flag, 0, sizeof(int));
        if (e == hipSuccess) {
            const size_t fs = pl->nelem * (size_t)n * n * n;
            for (int r = 0; r < 3; ++r)
                hipLaunchKernelGGL(any_nonzero_kernel, dim3(4096), dim3(256), 0, 0, m->christoffel + (size_t)(r * 9) * fs,
                                   3 * fs, flag);
            e = hipMemcpy(&any, flag, sizeof(int), hipMemcpyDeviceToHost);
        }
        if (flag) (void)hipFree(flag);
        if (e != hipSuccess) {
            (void)hipFree(pl->itf);
            delete pl;
            return fail(WX_ERR_HIP, "scan of the rotation Christoffel symbols failed: %s", hipGetErrorString(e));
        }
        b.rot_zero = any ? 0 : 1;
    }
    EulerConsts hc;
    memset(&hc, 0, sizeof(hc));
    for (int i = 0; i < n; ++i) {
        hc.em[i] = ops->extrap_neg[i]; hc.ep[i] = ops->extrap_pos[i];
        hc.cm[i] = ops->correction[2 * i]; hc.cp[i] = ops->correction[2 * i + 1];
        for (int j = 0; j < n; ++j) { hc.D[i * n + j] = ops->diff_solpt[i * n + j]; hc.HF[i * n + j] = ops->highfilter[i * n + j]; }
        hc.EF[i * n + i] = 1.0;
    }
    static const double identity[8] = {1, 0, 0, 0, 0, 1, 0, 0};
    for (int ed = 0; ed < 4; ++ed) {  // interior tile edges: no flip, no rotation (process_topology.py:219-228)
        hc.flip[ed] = on_panel_edge[ed] ? kFlip[panel][ed] : 0;
        pl->flip[ed] = hc.flip[ed];
        for (int i = 0; i < 8; ++i) pl->rot[ed][i] = hc.rot[ed][i] = on_panel_edge[ed] ? kRot[panel][ed][i] : identity[i];
    }
    e = hipMalloc((void**)&pl->consts, sizeof(EulerConsts));
    if (e == hipSuccess) e = hipMemcpy(pl->consts, &hc, sizeof(hc), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(pl->itf);
        if (pl->consts) (void)hipFree(pl->consts);
        delete pl;
        return fail(WX_ERR_HIP, "constant upload failed: %s", hipGetErrorString(e));
    }
    *out = pl;
    return WX_OK;
}

wx_status wx_euler3d_plan_destroy(wx_euler3d_plan* pl) {
    if (!pl) return WX_OK;
    hipError_t e = hipFree(pl->itf);
    if (pl->itf2) (void)hipFree(pl->itf2);
    if (pl->face_val) (void)hipFree(pl->face_val);
    hipError_t e2 = hipFree(pl->consts);
    if (e == hipSuccess) e = e2;
    delete pl;
    if (e != hipSuccess) return fail(WX_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
    return WX_OK;
}

wx_dtype wx_euler3d_plan_dtype(const wx_euler3d_plan* pl) { return pl ? pl->dtype : WX_F64; }

// Diagnostic (not in wxhip.h): give the plan a device buffer of 8 uint64 per workgroup for phase stamps.
wx_status wx_euler3d_debug_set_stamps(wx_euler3d_plan* pl, void* dev_buffer) {
    if (!pl) return fail(WX_ERR_INVALID, "null plan");
    pl->stamps = static_cast<unsigned long long*>(dev_buffer);
    return WX_OK;
}

wx_status wx_euler3d_plan_set_column_metric(wx_euler3d_plan* pl, const wx_euler3d_metric* cm) {
    if (!pl) return fail(WX_ERR_INVALID, "wx_euler3d_plan_set_column_metric: null plan");
    if (!cm) { pl->column = false; return WX_OK; }
    if (pl->dtype != WX_F64 && pl->dtype != WX_DUAL128)
        return fail(WX_ERR_INVALID, "wx_euler3d_plan_set_column_metric: the plan must be WX_F64 or WX_DUAL128");
    if (!cm->sqrtG || !cm->h_contra || !cm->christoffel || !cm->inv_dzdeta || !cm->sqrtG_itf_i || !cm->sqrtG_itf_j ||
        !cm->sqrtG_itf_k || !cm->h_contra_itf_i || !cm->h_contra_itf_j || !cm->h_contra_itf_k)
        return fail(WX_ERR_INVALID, "wx_euler3d_plan_set_column_metric: the column metric has a null member");
    pl->c_sg = cm->sqrtG; pl->c_h = cm->h_contra; pl->c_chr = cm->christoffel; pl->c_idz = cm->inv_dzdeta;
    pl->c_sgi = cm->sqrtG_itf_i; pl->c_sgj = cm->sqrtG_itf_j; pl->c_sgk = cm->sqrtG_itf_k;
    pl->c_hi = cm->h_contra_itf_i; pl->c_hj = cm->h_contra_itf_j; pl->c_hk = cm->h_contra_itf_k;
    pl->column = true;
    return WX_OK;
}

int wx_euler3d_plan_has_column_metric(const wx_euler3d_plan* pl) { return pl && pl->column ? 1 : 0; }

int wx_euler3d_plan_one_kernel(const wx_euler3d_plan* pl) { return pl ? (plan_direct(pl) ? 1 : 0) : -1; }

wx_status wx_euler3d_plan_set_one_kernel(wx_euler3d_plan* pl, int on) {
    if (!pl) return fail(WX_ERR_INVALID, "wx_euler3d_plan_set_one_kernel: null plan");
    if (on && !((pl->dtype == WX_F64 || pl->dtype == WX_DUAL128) && pl->n <= 4))
        return fail(WX_ERR_UNSUPPORTED, "the one-kernel form serves WX_F64 and WX_DUAL128 plans of num_solpts 2..4");
    pl->direct = on != 0;
    return WX_OK;
}

double wx_euler3d_bytes_per_point(const wx_euler3d_plan* pl) {
    if (!pl) return 0.0;
    // compulsory HBM traffic of one RHS-kernel launch per solution point: Q, R, sqrtG, 6 h^ij,
    // the Christoffel fields this plan reads, inv_dzdeta, the interface metric, the sponge fields
    const int n = pl->n;
    const double gammas = pl->base.rot_zero ? 18.0 : 27.0;
    return 8.0 * (5 + 5 + 1 + 6 + gammas + 1) + 3.0 * 2 * 4 * 8 / n + (pl->base.has_damp ? 32.0 : 0.0);
}

static bool jvp_lean() {
    static const bool lean = [] { const char* e = getenv("WXHIP_JVP_LEAN"); return !(e && e[0] == '0'); }();
    return lean;
}

int wx_euler3d_uses_matrix_cores(const wx_euler3d_plan* pl, wx_kernel kernel) {
    if (!pl) return -1;
    const bool rhs_mf = WX_MFMA && pl->n == 8 && pl->dtype == WX_F64;
    const bool jvp_mf = WX_MFMA && pl->n == 8 && pl->dtype == WX_DUAL128;
    switch (kernel) {
        case WX_KERNEL_RHS: case WX_KERNEL_STAGE: case WX_KERNEL_BATCH_RHS: return rhs_mf ? 1 : 0;
        case WX_KERNEL_JVP: return (jvp_mf && jvp_lean()) ? 1 : 0;   // WXHIP_JVP_LEAN=0: the generic dual instantiation
        case WX_KERNEL_BATCH_JVP: return jvp_mf ? 1 : 0;
    }
    return -1;
}

size_t wx_euler3d_edge_count(const wx_euler3d_plan* pl) {
    return pl ? (size_t)NQ * pl->V * pl->H * pl->n * pl->n : 0;
}

wx_status wx_euler3d_extrap_pack(wx_euler3d_plan* pl, const void* q, void* const send[4], wx_stream stream) {
    if (!pl || !q) return fail(WX_ERR_INVALID, "wx_euler3d_extrap_pack: null argument");
    WX_STREAM(st, stream);
    switch (pl->dtype) {
        case WX_F64: return run_extrap<double>(pl, q, send, st);
        case WX_C128: return run_extrap<cplx>(pl, q, send, st);
        case WX_DUAL128: return run_extrap<dual>(pl, q, send, st);
    }
    return fail(WX_ERR_INVALID, "bad plan dtype");
}

static wx_status euler3d_rhs_impl(wx_euler3d_plan* pl, const void* q, const void* const halo[4], void* out,
                                  wx_region region, wx_stream stream, int axpy, const void* y, double ca, double cb,
                                  double cc, const void* z, double cd) {
    if (!pl || !q || !out) return fail(WX_ERR_INVALID, "wx_euler3d_rhs: null argument");
    if (out == q) return fail(WX_ERR_INVALID, "wx_euler3d_rhs: output must not alias the state");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    if (region != WX_REGION_INTERIOR) {
        if (!halo) return fail(WX_ERR_INVALID, "wx_euler3d_rhs: halo is required for this region");
        for (int e = 0; e < 4; ++e)
            if (!halo[e]) return fail(WX_ERR_INVALID, "wx_euler3d_rhs: halo[%d] is null", e);
    }
    WX_STREAM(st, stream);
    switch (pl->dtype) {
        case WX_F64: return run_rhs<double>(pl, q, halo, out, region, st, axpy, y, ca, cb, cc, z, cd);
        case WX_C128: return run_rhs<cplx>(pl, q, halo, out, region, st, axpy, y, ca, cb, cc, z, cd);
        case WX_DUAL128: return run_rhs<dual>(pl, q, halo, out, region, st, axpy, y, ca, cb, cc, z, cd);
    }
    return fail(WX_ERR_INVALID, "bad plan dtype");
}

wx_status wx_euler3d_rhs(wx_euler3d_plan* pl, const void* q, const void* const halo[4], void* rhs, wx_region region,
                         wx_stream stream) {
    return euler3d_rhs_impl(pl, q, halo, rhs, region, stream, 0, nullptr, 0.0, 0.0, 1.0, nullptr, 0.0);
}

wx_status wx_euler3d_rhs_axpy(wx_euler3d_plan* pl, const void* q, const void* const halo[4], const void* y, void* out,
                              double a, double b, double c, wx_region region, wx_stream stream) {
    return euler3d_rhs_impl(pl, q, halo, out, region, stream, 1, y, a, b, c, nullptr, 0.0);
}

// Complex-step Jacobian-vector product without complex arrays in HBM (WX_DUAL128 plans only).
wx_status wx_euler3d_jvp_extrap_pack(wx_euler3d_plan* pl, const double* q, const double* v, double eps, void* const send[4],
                                     wx_stream stream) {
    if (!pl || !q || !v) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_extrap_pack: null argument");
    if (pl->dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_*: the plan must be WX_DUAL128");
    EulerParams<dual> P = make_params<dual>(pl);
    P.jvp = 1; P.q_re = q; P.q_tan = v; P.jvp_eps = eps;
    if (send) {
        P.send_s = static_cast<dual*>(send[0]); P.send_n = static_cast<dual*>(send[1]);
        P.send_w = static_cast<dual*>(send[2]); P.send_e = static_cast<dual*>(send[3]);
    }
    WX_STREAM(st, stream);
    if (plan_direct(pl) && jvp_lean()) {   // one-kernel form: the edge messages only (no interface buffer)
        if (!send) return WX_OK;
        switch (pl->n) {
            case 2: return launch_pack<2, dual>(P, st);
            case 3: return launch_pack<3, dual>(P, st);
            case 4: return launch_pack<4, dual>(P, st);
        }
    }
    return dispatch_extrap<dual>(pl->n, P, st);
}

wx_status wx_euler3d_jvp(wx_euler3d_plan* pl, const double* q, const double* v, double eps, const void* const halo[4],
                         double* out, double scale, wx_region region, wx_stream stream) {
    if (!pl || !q || !v || !out) return fail(WX_ERR_INVALID, "wx_euler3d_jvp: null argument");
    if (pl->dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_*: the plan must be WX_DUAL128");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    if (region != WX_REGION_INTERIOR) {
        if (!halo) return fail(WX_ERR_INVALID, "wx_euler3d_jvp: halo is required for this region");
        for (int e = 0; e < 4; ++e)
            if (!halo[e]) return fail(WX_ERR_INVALID, "wx_euler3d_jvp: halo[%d] is null", e);
    }
    EulerParams<dual> P = make_params<dual>(pl);
    P.jvp = 1; P.q_re = q; P.q_tan = v; P.jvp_eps = eps; P.out_tan = out; P.jvp_scale = scale;
    P.region = region; P.count = region_count(region, pl->H, pl->V);
    if (halo) {
        P.halo_s = static_cast<const dual*>(halo[0]); P.halo_n = static_cast<const dual*>(halo[1]);
        P.halo_w = static_cast<const dual*>(halo[2]); P.halo_e = static_cast<const dual*>(halo[3]);
    }
    WX_STREAM(st, stream);
    if (!jvp_lean()) return dispatch_rhs<dual>(pl->n, P, st);
    if (plan_direct(pl)) {
        switch (pl->n) {
            case 2: return launch_jvp_brick<2>(P, st);
            case 3: return launch_jvp_brick<3>(P, st);
            case 4: return launch_jvp_brick<4>(P, st);
        }
    }
    if (pl->column) {   // column form of the metric: the launch reads the slabs
        P.sg = pl->c_sg; P.h = pl->c_h; P.chr = pl->c_chr; P.idz = pl->c_idz;
        P.sgi = pl->c_sgi; P.sgj = pl->c_sgj; P.sgk = pl->c_sgk; P.hi = pl->c_hi; P.hj = pl->c_hj; P.hk = pl->c_hk;
        return dispatch_jvp_column(pl->n, P, st);
    }
    switch (pl->n) {
        case 2: return launch_jvp<2>(P, st);
        case 3: return launch_jvp<3>(P, st);
        case 4: return launch_jvp<4>(P, st);
        case 5: return launch_jvp<5>(P, st);
        case 6: return launch_jvp<6>(P, st);
        case 7: return launch_jvp<7>(P, st);
        case 8: return launch_jvp<8>(P, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", pl->n);
}

// ---- prepared complex-step JVP: one linearisation state, many products (a Krylov solve).
// wx_euler3d_jvp_prepare: the face VALUES of q (the plain float64 extrapolation: bit for bit what a float64 plan's
// wx_euler3d_extrap_pack writes) into the plan's value cache, and the value edge messages into send_val (real,
// wx_euler3d_edge_count doubles each) for the caller to exchange ONCE and keep.
wx_status wx_euler3d_jvp_prepare(wx_euler3d_plan* pl, const double* q, void* const send_val[4], wx_stream stream) {
    if (!pl || !q) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_prepare: null argument");
    if (pl->dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_*: the plan must be WX_DUAL128");
    static_assert(NQ == 5, "the value cache holds the five prognostic face values");
    if (!pl->face_val)
        return fail(WX_ERR_INVALID, "wx_euler3d_jvp_prepare: no face-value cache - call wx_euler3d_plan_reserve(plan, "
                                    "WX_RESERVE_JVP) at setup time (evaluation entry points do not allocate)");
    EulerParams<double> P = make_params<double>(pl);
    P.itf = pl->face_val;
    P.q = q;
    if (send_val) {
        P.send_s = static_cast<double*>(send_val[0]); P.send_n = static_cast<double*>(send_val[1]);
        P.send_w = static_cast<double*>(send_val[2]); P.send_e = static_cast<double*>(send_val[3]);
    }
    WX_STREAM(st, stream);
    return dispatch_extrap<double>(pl->n, P, st);
}

// Per product: only the TANGENTS of the face values (real; the plan's interface buffer serves as their store) and the
// tangent edge messages (send_tan: real, wx_euler3d_edge_count doubles each).  Reads v, and of q the two rows that are
// extrapolated in log space.
wx_status wx_euler3d_jvp_tangent_extrap_pack(wx_euler3d_plan* pl, const double* q, const double* v, double eps,
                                             void* const send_tan[4], wx_stream stream) {
    return wx_euler3d_jvp_tangent_extrap_pack_fix(pl, q, const_cast<double*>(v), eps, send_tan, nullptr, nullptr, nullptr, nullptr,
                                                  nullptr, stream);
}

// ... with the tangent corrected in place first: v <- v - h[0] s[0] row0 [- h[1] s[1] row1] (h, s: device memory; s null: 1;
// row1 null: one row), |v|^2 left as wx_euler3d_jvp_workgroups(plan, WX_REGION_ALL) partial sums in `partials`.
wx_status wx_euler3d_jvp_tangent_extrap_pack_fix(wx_euler3d_plan* pl, const double* q, double* v, double eps,
                                                 void* const send_tan[4], const double* row0, const double* row1,
                                                 const double* h, const double* s, double* partials, wx_stream stream) {
    if (!pl || !q || !v) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_tangent_extrap_pack: null argument");
    if (row0 && (!h || !partials)) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_tangent_extrap_pack_fix: rows without coefficients / partials");
    if (pl->dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_*: the plan must be WX_DUAL128");
    if (!pl->face_val) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_tangent_extrap_pack: call wx_euler3d_jvp_prepare first");
    EulerParams<dual> P = make_params<dual>(pl);
    P.jvp = 1; P.q_re = q; P.q_tan = v; P.jvp_eps = eps;
    P.split = 2; P.ft = static_cast<double*>(pl->itf); P.fv = pl->face_val;
    if (send_tan) {
        P.send_s = static_cast<dual*>(send_tan[0]); P.send_n = static_cast<dual*>(send_tan[1]);
        P.send_w = static_cast<dual*>(send_tan[2]); P.send_e = static_cast<dual*>(send_tan[3]);
    }
    const bool fix = row0 != nullptr;
    if (fix) { P.tc_r0 = row0; P.tc_r1 = row1; P.tch = h; P.tcs = s; P.tc_out = v; P.tc_part = partials; }
    WX_STREAM(st, stream);
    const int nelem = (int)pl->nelem;
#define WX_TAN_CASE(NN)                                                                                                \
    case NN: {                                                                                                         \
        const dim3 grid = grid3_for<NN>() ? region_grid(WX_REGION_ALL, pl->H, pl->V)                                    \
                                          : dim3((nelem + Cfg<NN>::EPB - 1) / Cfg<NN>::EPB);                           \
        if (fix) hipLaunchKernelGGL((euler_tan_extrap_kernel<NN, true>), grid, dim3(Cfg<NN>::BS), 0, st, P);            \
        else hipLaunchKernelGGL((euler_tan_extrap_kernel<NN, false>), grid, dim3(Cfg<NN>::BS), 0, st, P);              \
        break;                                                                                                         \
    }
    switch (pl->n) {
        WX_TAN_CASE(2) WX_TAN_CASE(3) WX_TAN_CASE(4) WX_TAN_CASE(5) WX_TAN_CASE(6) WX_TAN_CASE(7) WX_TAN_CASE(8)
        default: return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", pl->n);
    }
#undef WX_TAN_CASE
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

// out (real) = scale * Im R(q + i eps v) from the cached face values (+ value halos the caller kept) and this product's
// face tangents (+ tangent halos).  halo_val / halo_tan: four REAL edge messages each (may be null for INTERIOR).
wx_status wx_euler3d_jvp_prepared(wx_euler3d_plan* pl, const double* q, const double* v, double eps,
                                  const void* const halo_val[4], const void* const halo_tan[4], double* out, double scale,
                                  wx_region region, wx_stream stream) {
    return wx_euler3d_jvp_prepared_axpy(pl, q, v, eps, halo_val, halo_tan, out, scale, nullptr, nullptr, nullptr, nullptr,
                                        nullptr, nullptr, region, stream);
}

// workgroups of one wx_euler3d_jvp_prepared* launch over `region` (= the pairs of partial products such a launch writes)
size_t wx_euler3d_jvp_workgroups(const wx_euler3d_plan* pl, wx_region region) {
    if (!pl || (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)) return 0;
    const int count = region_count(region, pl->H, pl->V);
    if (count == 0) return 0;
    const int n3 = pl->n * pl->n * pl->n, epb = n3 >= 216 ? 1 : 256 / n3;   // (Cfg<N>::EPB)
    const size_t grid = ((size_t)count + epb - 1) / epb;
    if (pl->column) return 8 * ((grid + 7) / 8);                          // (launch_jvp_column)
    if (epb == 1) { const dim3 g = region_grid(region, pl->H, pl->V); return (size_t)g.x * g.y * g.z; }
    return grid;
}

// ... with the store  out = *z_scale * (scale * Im R) + *z_coef * z  (z null: the plain product; the two coefficients are
// read from device memory by the kernel: a Krylov solver that keeps them there needs no host synchronisation)
// ... and, with row0 (row1 nullable) and partials: the products <row_r, out> of the stored vector as one pair of partial sums per
// workgroup, partials[2 * w + r], w < wx_euler3d_jvp_workgroups(pl, region) (the caller sums them: wx_kiops_long_a_finish)
wx_status wx_euler3d_jvp_prepared_axpy(wx_euler3d_plan* pl, const double* q, const double* v, double eps,
                                       const void* const halo_val[4], const void* const halo_tan[4], double* out, double scale,
                                       const double* z, const double* z_scale, const double* z_coef, const double* row0,
                                       const double* row1, double* partials, wx_region region, wx_stream stream) {
    if (!pl || !q || !v || !out) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_prepared: null argument");
    if (z && !z_coef) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_prepared_axpy: z without its coefficient");
    if ((row0 || row1 || partials) && !(z && row0 && partials))
        return fail(WX_ERR_INVALID, "wx_euler3d_jvp_prepared_axpy: products need z, row0 and the partials buffer");
    if (pl->dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_*: the plan must be WX_DUAL128");
    if (!pl->face_val) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_prepared: call wx_euler3d_jvp_prepare first");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    if (region != WX_REGION_INTERIOR) {
        if (!halo_val || !halo_tan) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_prepared: halos are required for this region");
        for (int e = 0; e < 4; ++e)
            if (!halo_val[e] || !halo_tan[e]) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_prepared: halo[%d] is null", e);
    }
    EulerParams<dual> P = make_params<dual>(pl);
    P.jvp = 1; P.q_re = q; P.q_tan = v; P.jvp_eps = eps; P.out_tan = out; P.jvp_scale = scale;
    P.split = 1; P.ft = static_cast<double*>(pl->itf); P.fv = pl->face_val;
    P.jz = z; P.jzs = z ? z_scale : nullptr; P.jza = z ? z_coef : nullptr;
    P.jr0 = row0; P.jr1 = row0 ? row1 : nullptr; P.jpart = row0 ? partials : nullptr;
    P.region = region; P.count = region_count(region, pl->H, pl->V);
    if (halo_val && halo_tan) {
        P.halo_s = static_cast<const dual*>(halo_tan[0]); P.halo_n = static_cast<const dual*>(halo_tan[1]);
        P.halo_w = static_cast<const dual*>(halo_tan[2]); P.halo_e = static_cast<const dual*>(halo_tan[3]);
        P.hv_s = static_cast<const double*>(halo_val[0]); P.hv_n = static_cast<const double*>(halo_val[1]);
        P.hv_w = static_cast<const double*>(halo_val[2]); P.hv_e = static_cast<const double*>(halo_val[3]);
    }
    WX_STREAM(st, stream);
    if (pl->column) {   // column form of the metric: the launch reads the slabs
        P.sg = pl->c_sg; P.h = pl->c_h; P.chr = pl->c_chr; P.idz = pl->c_idz;
        P.sgi = pl->c_sgi; P.sgj = pl->c_sgj; P.sgk = pl->c_sgk; P.hi = pl->c_hi; P.hj = pl->c_hj; P.hk = pl->c_hk;
        return dispatch_jvp_column(pl->n, P, st);
    }
    switch (pl->n) {
        case 2: return launch_jvp<2>(P, st);
        case 3: return launch_jvp<3>(P, st);
        case 4: return launch_jvp<4>(P, st);
        case 5: return launch_jvp<5>(P, st);
        case 6: return launch_jvp<6>(P, st);
        case 7: return launch_jvp<7>(P, st);
        case 8: return launch_jvp<8>(P, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", pl->n);
}

wx_status wx_euler3d_rhs_axpy2(wx_euler3d_plan* pl, const void* q, const void* const halo[4], const void* y,
                               const void* z, void* out, double a, double b, double c, double d, wx_region region,
                               wx_stream stream) {
    return euler3d_rhs_impl(pl, q, halo, out, region, stream, 1, y, a, b, c, z, d);
}

// Finite-difference Jacobian products: the same two kernels on the shifted state q + eps v, formed on load.
wx_status wx_euler3d_shifted_extrap_pack(wx_euler3d_plan* pl, const double* q, const double* v, double eps,
                                         void* const send[4], wx_stream stream) {
    if (!pl || !q || !v) return fail(WX_ERR_INVALID, "wx_euler3d_shifted_extrap_pack: null argument");
    if (pl->dtype != WX_F64) return fail(WX_ERR_INVALID, "wx_euler3d_shifted_*: the plan must be WX_F64");
    EulerParams<double> P = make_params<double>(pl);
    P.q = q; P.q_tan = v; P.jvp_eps = eps;
    if (send) {
        P.send_s = static_cast<double*>(send[0]); P.send_n = static_cast<double*>(send[1]);
        P.send_w = static_cast<double*>(send[2]); P.send_e = static_cast<double*>(send[3]);
    }
    WX_STREAM(st, stream);
    if (plan_direct(pl)) return send ? dispatch_pack(pl->n, P, st) : WX_OK;
    return dispatch_extrap<double>(pl->n, P, st);
}

wx_status wx_euler3d_shifted_rhs_axpy2(wx_euler3d_plan* pl, const double* q, const double* v, double eps,
                                       const void* const halo[4], const double* y, const double* z, double* out, double a,
                                       double b, double c, double d, wx_region region, wx_stream stream) {
    if (!pl || !q || !v || !out) return fail(WX_ERR_INVALID, "wx_euler3d_shifted_rhs_axpy2: null argument");
    if (pl->dtype != WX_F64) return fail(WX_ERR_INVALID, "wx_euler3d_shifted_*: the plan must be WX_F64");
    if (out == q || out == v) return fail(WX_ERR_INVALID, "wx_euler3d_shifted_rhs_axpy2: output must not alias q or v");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    if (region != WX_REGION_INTERIOR) {
        if (!halo) return fail(WX_ERR_INVALID, "wx_euler3d_shifted_rhs_axpy2: halo is required for this region");
        for (int e = 0; e < 4; ++e)
            if (!halo[e]) return fail(WX_ERR_INVALID, "wx_euler3d_shifted_rhs_axpy2: halo[%d] is null", e);
    }
    EulerParams<double> P = make_params<double>(pl);
    P.q = q; P.q_tan = v; P.jvp_eps = eps; P.rhs = out;
    P.region = region; P.count = region_count(region, pl->H, pl->V);
    P.axpy = 1; P.ca = a; P.cb = b; P.cc = c; P.cd = d; P.y = y; P.z = z;
    if (halo) {
        P.halo_s = static_cast<const double*>(halo[0]); P.halo_n = static_cast<const double*>(halo[1]);
        P.halo_w = static_cast<const double*>(halo[2]); P.halo_e = static_cast<const double*>(halo[3]);
    }
    WX_STREAM(st, stream);
    if (plan_direct(pl)) return dispatch_brick(pl->n, P, false, st);
    return dispatch_rhs<double>(pl->n, P, st);
}

static wx_status ensure_slot1(wx_euler3d_plan* pl) {
    if (pl->itf2 || plan_direct(pl)) return WX_OK;   // (the one-kernel form has no interface buffer)
    return fail(WX_ERR_INVALID, "the stage pipeline needs the second interface buffer - call wx_euler3d_plan_reserve(plan, "
                                "WX_RESERVE_STAGE) at setup time (evaluation entry points do not allocate)");
}

// Setup-time allocation of what only some callers need: the second interface slot of the stage pipeline, the face-value
// cache of the prepared JVP.  Evaluation entry points never allocate; they refuse when their buffer is absent.
wx_status wx_euler3d_plan_reserve(wx_euler3d_plan* pl, int what) {
    if (!pl) return fail(WX_ERR_INVALID, "wx_euler3d_plan_reserve: null plan");
    if (what & ~(WX_RESERVE_STAGE | WX_RESERVE_JVP)) return fail(WX_ERR_INVALID, "wx_euler3d_plan_reserve: unknown flags %d", what);
    if ((what & WX_RESERVE_JVP) && pl->dtype != WX_DUAL128)
        return fail(WX_ERR_INVALID, "wx_euler3d_plan_reserve: WX_RESERVE_JVP is for WX_DUAL128 plans");
    if ((what & WX_RESERVE_STAGE) && !pl->itf2) {
        hipError_t e = hipMalloc(&pl->itf2, pl->itf_bytes);
        if (e != hipSuccess) return fail(WX_ERR_NOMEM, "hipMalloc(%zu bytes) for the second interface buffer failed: %s",
                                         pl->itf_bytes, hipGetErrorString(e));
    }
    if ((what & WX_RESERVE_JVP) && !pl->face_val) {
        hipError_t e = hipMalloc((void**)&pl->face_val, pl->itf_bytes / 2);
        if (e != hipSuccess) return fail(WX_ERR_NOMEM, "hipMalloc(%zu bytes) for the face-value cache failed: %s",
                                         pl->itf_bytes / 2, hipGetErrorString(e));
    }
    return WX_OK;
}

int wx_euler3d_plan_reserved(const wx_euler3d_plan* pl) {
    return pl ? ((pl->itf2 ? WX_RESERVE_STAGE : 0) | (pl->face_val ? WX_RESERVE_JVP : 0)) : 0;
}

wx_status wx_euler3d_extrap_pack_slot(wx_euler3d_plan* pl, const void* q, void* const send[4], int slot,
                                      wx_stream stream) {
    if (!pl || !q) return fail(WX_ERR_INVALID, "wx_euler3d_extrap_pack_slot: null argument");
    if (slot != 0 && slot != 1) return fail(WX_ERR_INVALID, "interface slot %d not in {0,1}", slot);
    if (slot == 1) { wx_status s1 = ensure_slot1(pl); if (s1 != WX_OK) return s1; }
    WX_STREAM(st, stream);
    switch (pl->dtype) {
        case WX_F64: return run_extrap<double>(pl, q, send, st, slot);
        case WX_C128: return run_extrap<cplx>(pl, q, send, st, slot);
        case WX_DUAL128: return run_extrap<dual>(pl, q, send, st, slot);
    }
    return fail(WX_ERR_INVALID, "bad plan dtype");
}

wx_status wx_euler3d_set_exp_filter(wx_euler3d_plan* pl, const double* filter) {
    if (!pl || !filter) return fail(WX_ERR_INVALID, "wx_euler3d_set_exp_filter: null argument");
    double ef[kMaxN * kMaxN] = {0.0};
    for (int i = 0; i < pl->n * pl->n; ++i) ef[i] = filter[i];
    // setup-time call: kernels still reading the old matrix on any stream (a non-blocking one included) finish first
    WX_HIP_TRY(hipDeviceSynchronize());
    WX_HIP_TRY(hipMemcpy(pl->consts->EF, ef, sizeof(ef), hipMemcpyHostToDevice));
    return WX_OK;
}

wx_status wx_euler3d_stage(wx_euler3d_plan* pl, const void* q, const void* const halo[4], const void* y, const void* z,
                           void* out, double a, double b, double c, double d, wx_region region, int itf_in,
                           void* const next_send[4], int prepare_next, int* nan_flag, wx_stream stream) {
    if (!pl || !q || !out) return fail(WX_ERR_INVALID, "wx_euler3d_stage: null argument");
    if (out == q) return fail(WX_ERR_INVALID, "wx_euler3d_stage: output must not alias the state");
    if (itf_in != 0 && itf_in != 1) return fail(WX_ERR_INVALID, "interface slot %d not in {0,1}", itf_in);
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    if (region != WX_REGION_INTERIOR) {
        if (!halo) return fail(WX_ERR_INVALID, "wx_euler3d_stage: halo is required for this region");
        for (int e = 0; e < 4; ++e)
            if (!halo[e]) return fail(WX_ERR_INVALID, "wx_euler3d_stage: halo[%d] is null", e);
    }
    if (prepare_next || itf_in == 1) { wx_status s1 = ensure_slot1(pl); if (s1 != WX_OK) return s1; }
    WX_STREAM(st, stream);
    const bool ep = prepare_next != 0, ef = prepare_next == 2;
    switch (pl->dtype) {
        case WX_F64: return run_rhs<double>(pl, q, halo, out, region, st, 1, y, a, b, c, z, d, itf_in, next_send, ep, ef, nan_flag);
        case WX_C128: return run_rhs<cplx>(pl, q, halo, out, region, st, 1, y, a, b, c, z, d, itf_in, next_send, ep, ef, nan_flag);
        case WX_DUAL128: return run_rhs<dual>(pl, q, halo, out, region, st, 1, y, a, b, c, z, d, itf_in, next_send, ep, ef, nan_flag);
    }
    return fail(WX_ERR_INVALID, "bad plan dtype");
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// batch: all tiles of a rank in one launch per phase
// ------------------------------------------------------------------------------------------------
struct wx_euler3d_batch {
    int n, H, V, count, nelem;
    wx_dtype dtype;
    bool direct = false;    // every plan takes the one-kernel form (euler3d_brick.h)
    bool pulls = false;     // ... and every tile's neighbours are tiles of this batch: no pack launch (EulerParams::pull_tile)
    void* table = nullptr;  // device: EulerParams<T>[count]
};

namespace {

template <typename T>
wx_status batch_upload(wx_euler3d_batch* b, wx_euler3d_plan* const* plans, void* const (*send)[4],
                       const void* const (*halo)[4]) {
    std::vector<EulerParams<T>> host(b->count);
    for (int i = 0; i < b->count; ++i) {
        EulerParams<T> P = make_params<T>(plans[i]);
        P.send_s = static_cast<T*>(send[i][0]); P.send_n = static_cast<T*>(send[i][1]);
        P.send_w = static_cast<T*>(send[i][2]); P.send_e = static_cast<T*>(send[i][3]);
        P.halo_s = static_cast<const T*>(halo[i][0]); P.halo_n = static_cast<const T*>(halo[i][1]);
        P.halo_w = static_cast<const T*>(halo[i][2]); P.halo_e = static_cast<const T*>(halo[i][3]);
        if (b->pulls) {
            double prot[4][8];
            const double* px[4];
            for (int e = 0; e < 4; ++e)
                for (int j = 0; j < b->count; ++j)
                    for (int e2 = 0; e2 < 4; ++e2)
                        if (send[j][e2] == halo[i][e]) {
                            P.pull_tile[e] = j; P.pull_edge[e] = e2; P.pull_flip[e] = plans[j]->flip[e2];
                            for (int k = 0; k < 8; ++k) prot[e][k] = plans[j]->rot[e2][k];
                            px[e] = e2 >= E_W ? plans[j]->base.bwe : plans[j]->base.bsn;
                        }
            // (setup time: the plan's device constants learn what its four neighbours do to their lines)
            hipError_t e1 = hipMemcpy(&plans[i]->consts->prot, prot, sizeof(prot), hipMemcpyHostToDevice);
            if (e1 == hipSuccess) e1 = hipMemcpy(&plans[i]->consts->pull_x, px, sizeof(px), hipMemcpyHostToDevice);
            if (e1 != hipSuccess) return fail(WX_ERR_HIP, "batch neighbour tables: %s", hipGetErrorString(e1));
        }
        host[i] = P;
    }
    const size_t bytes = sizeof(EulerParams<T>) * b->count;
    hipError_t e = hipMalloc(&b->table, bytes);
    if (e == hipSuccess) e = hipMemcpy(b->table, host.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(WX_ERR_HIP, "batch table upload failed: %s", hipGetErrorString(e));
    return WX_OK;
}

template <typename T>
wx_status batch_extrap(const wx_euler3d_batch* b, const void* q, const double* v, double eps, size_t stride, hipStream_t st,
                       const double* dscale = nullptr) {
    EulerBatchDyn<T> dyn{};
    dyn.dscale = dscale;
    dyn.stride = stride;
    dyn.stride_re = stride;
    if (v != nullptr && std::is_same<T, dual>::value) {  // dual state (q, eps v) from two real arrays
        dyn.q_re = static_cast<const double*>(q); dyn.q_tan = v; dyn.eps = eps; dyn.jvp = 1;
    } else {
        dyn.q = static_cast<const T*>(q);
        dyn.q_tan = v; dyn.eps = eps;  // float64: shifted state q + eps v (null: plain)
    }
    const EulerParams<T>* t = static_cast<const EulerParams<T>*>(b->table);
    if constexpr (std::is_same<T, dual>::value) {
        if (b->direct && dyn.jvp && jvp_lean()) {   // the one-kernel form of the complex-step product: edge messages only, or nothing
            if (b->pulls) return WX_OK;
            switch (b->n) {
                case 2: return launch_pack_batch<2, T>(t, dyn, b->H, b->V, b->count, st);
                case 3: return launch_pack_batch<3, T>(t, dyn, b->H, b->V, b->count, st);
                case 4: return launch_pack_batch<4, T>(t, dyn, b->H, b->V, b->count, st);
            }
        }
    }
    if constexpr (std::is_same<T, double>::value) {
        if (b->direct && b->pulls) return WX_OK;   // (the evaluation forms the tile-edge states itself: nothing to pack)
        if (b->direct) {
            switch (b->n) {
                case 2: return launch_pack_batch<2, T>(t, dyn, b->H, b->V, b->count, st);
                case 3: return launch_pack_batch<3, T>(t, dyn, b->H, b->V, b->count, st);
                case 4: return launch_pack_batch<4, T>(t, dyn, b->H, b->V, b->count, st);
            }
        }
    }
    switch (b->n) {
        case 2: return launch_extrap_batch<2, T>(t, dyn, b->nelem, b->count, st);
        case 3: return launch_extrap_batch<3, T>(t, dyn, b->nelem, b->count, st);
        case 4: return launch_extrap_batch<4, T>(t, dyn, b->nelem, b->count, st);
        case 5: return launch_extrap_batch<5, T>(t, dyn, b->nelem, b->count, st);
        case 6: return launch_extrap_batch<6, T>(t, dyn, b->nelem, b->count, st);
        case 7: return launch_extrap_batch<7, T>(t, dyn, b->nelem, b->count, st);
        case 8: return launch_extrap_batch<8, T>(t, dyn, b->nelem, b->count, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", b->n);
}

template <typename T>
wx_status batch_rhs(const wx_euler3d_batch* b, const void* q, const double* v, double eps, const void* y, const void* z,
                    void* out, size_t stride, int axpy, double ca, double cb, double cc, double cd, wx_region region,
                    hipStream_t st, const double* dscale = nullptr) {
    EulerBatchDyn<T> dyn{};
    dyn.dscale = dscale;
    dyn.pulls = (b->direct && b->pulls) ? 1 : 0;
    dyn.q = static_cast<const T*>(q); dyn.y = static_cast<const T*>(y); dyn.z = static_cast<const T*>(z);
    dyn.rhs = static_cast<T*>(out);
    dyn.stride = stride;
    dyn.stride_re = stride;
    dyn.q_tan = v; dyn.eps = eps;
    dyn.region = region; dyn.count = region_count(region, b->H, b->V);
    dyn.axpy = axpy; dyn.ca = ca; dyn.cb = cb; dyn.cc = cc; dyn.cd = cd;
    const EulerParams<T>* t = static_cast<const EulerParams<T>*>(b->table);
    if constexpr (std::is_same<T, double>::value) {
        if (b->direct) {
            switch (b->n) {
                case 2: return launch_brick_batch<2>(t, dyn, b->H, b->V, b->count, st);
                case 3: return launch_brick_batch<3>(t, dyn, b->H, b->V, b->count, st);
                case 4: return launch_brick_batch<4>(t, dyn, b->H, b->V, b->count, st);
            }
        }
    }
    switch (b->n) {
        case 2: return launch_rhs_batch<2, T>(t, dyn, b->count, st);
        case 3: return launch_rhs_batch<3, T>(t, dyn, b->count, st);
        case 4: return launch_rhs_batch<4, T>(t, dyn, b->count, st);
        case 5: return launch_rhs_batch<5, T>(t, dyn, b->count, st);
        case 6: return launch_rhs_batch<6, T>(t, dyn, b->count, st);
        case 7: return launch_rhs_batch<7, T>(t, dyn, b->count, st);
        case 8: return launch_rhs_batch<8, T>(t, dyn, b->count, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", b->n);
}

}  // namespace

extern "C" {

wx_status wx_euler3d_batch_create(wx_euler3d_batch** out, wx_euler3d_plan* const* plans, int count, void* const (*send)[4],
                                  const void* const (*halo)[4]) {
    if (!out || !plans || count < 1 || !send || !halo) return fail(WX_ERR_INVALID, "wx_euler3d_batch_create: bad argument");
    for (int i = 0; i < count; ++i) {
        if (!plans[i]) return fail(WX_ERR_INVALID, "wx_euler3d_batch_create: plan %d is null", i);
        if (plans[i]->n != plans[0]->n || plans[i]->H != plans[0]->H || plans[i]->V != plans[0]->V ||
            plans[i]->dtype != plans[0]->dtype || plans[i]->case_number != plans[0]->case_number)
            return fail(WX_ERR_INVALID, "wx_euler3d_batch_create: plan %d differs in shape, dtype or case", i);
        for (int e = 0; e < 4; ++e)
            if (!send[i][e] || !halo[i][e]) return fail(WX_ERR_INVALID, "wx_euler3d_batch_create: null edge buffer");
    }
    wx_euler3d_batch* b = new (std::nothrow) wx_euler3d_batch;
    if (!b) return fail(WX_ERR_NOMEM, "out of host memory");
    b->n = plans[0]->n; b->H = plans[0]->H; b->V = plans[0]->V; b->count = count; b->nelem = (int)plans[0]->nelem;
    b->dtype = plans[0]->dtype;
    b->direct = true;
    for (int i = 0; i < count; ++i) b->direct = b->direct && plan_direct(plans[i]);
    {   // every halo line IS the send line of a tile of this batch (one rank owns the sphere: the exchange aliases): pulls
        // Taken at LAUNCH-BOUND sizes only (the shipped .ini files: the pack launch is a fifth of an evaluation there); at the
        // reference's benchmark size the tile-edge items cost the kernel more (+12 us) than the pack launch it saves (8 us):
        // profiles/r06_low_order_forms.txt.  WXHIP_BRICK_PULLS=1 / 0 forces it on / off.
        const char* env = getenv("WXHIP_BRICK_PULLS");
        const size_t points = (size_t)count * plans[0]->nelem * (size_t)(b->n * b->n * b->n);
        const bool wanted = env ? env[0] != '0' : points <= 262144;
        bool all = b->direct && (b->dtype == WX_F64 || b->dtype == WX_DUAL128) && count <= 30 && wanted;   // (30: a tile index + 1 in five bits)
        for (int i = 0; i < count && all; ++i)
            for (int e = 0; e < 4 && all; ++e) {
                bool found = false;
                for (int j = 0; j < count && !found; ++j)
                    for (int e2 = 0; e2 < 4 && !found; ++e2) found = send[j][e2] == halo[i][e];
                all = found;
            }
        b->pulls = all;
    }
    wx_status s = WX_ERR_INVALID;
    try {   // (the host copy of the table is a std::vector: no exception crosses the C boundary)
        switch (b->dtype) {
            case WX_F64: s = batch_upload<double>(b, plans, send, halo); break;
            case WX_C128: s = batch_upload<cplx>(b, plans, send, halo); break;
            case WX_DUAL128: s = batch_upload<dual>(b, plans, send, halo); break;
        }
    } catch (...) {
        s = fail(WX_ERR_NOMEM, "wx_euler3d_batch_create: out of host memory");
    }
    if (s != WX_OK) {
        if (b->table) (void)hipFree(b->table);
        delete b;
        return s;
    }
    *out = b;
    return WX_OK;
}

int wx_euler3d_batch_pulls(const wx_euler3d_batch* b) { return b ? ((b->direct && b->pulls) ? 1 : 0) : -1; }

wx_status wx_euler3d_batch_destroy(wx_euler3d_batch* b) {
    if (!b) return WX_OK;
    hipError_t e = hipFree(b->table);
    delete b;
    if (e != hipSuccess) return fail(WX_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
    return WX_OK;
}

wx_status wx_euler3d_batch_extrap_pack(const wx_euler3d_batch* b, const void* q, const double* v, double eps,
                                       size_t panel_stride, wx_stream stream) {
    if (!b || !q) return fail(WX_ERR_INVALID, "wx_euler3d_batch_extrap_pack: null argument");
    if (v && b->dtype == WX_C128) return fail(WX_ERR_INVALID, "a shift / tangent vector needs a WX_F64 or WX_DUAL128 batch");
    WX_STREAM(st, stream);
    switch (b->dtype) {
        case WX_F64: return batch_extrap<double>(b, q, v, eps, panel_stride, st);
        case WX_C128: return batch_extrap<cplx>(b, q, nullptr, 0.0, panel_stride, st);
        case WX_DUAL128: return batch_extrap<dual>(b, q, v, eps, panel_stride, st);
    }
    return fail(WX_ERR_INVALID, "bad batch dtype");
}

wx_status wx_euler3d_batch_jvp(const wx_euler3d_batch* b, const double* q, const double* v, double eps, double* out,
                               double scale, size_t panel_stride, wx_region region, wx_stream stream) {
    if (!b || !q || !v || !out) return fail(WX_ERR_INVALID, "wx_euler3d_batch_jvp: null argument");
    if (b->dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "wx_euler3d_batch_jvp: the batch must be WX_DUAL128");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    EulerBatchDyn<dual> dyn{};
    dyn.q_re = q; dyn.q_tan = v; dyn.out_tan = out; dyn.eps = eps; dyn.scale = scale; dyn.jvp = 1;
    dyn.stride_re = panel_stride;
    dyn.region = region; dyn.count = region_count(region, b->H, b->V);
    const EulerParams<dual>* t = static_cast<const EulerParams<dual>*>(b->table);
    WX_STREAM(st, stream);
    if (b->direct && jvp_lean()) {
        dyn.pulls = b->pulls ? 1 : 0;
        switch (b->n) {
            case 2: return launch_jvp_brick_batch<2>(t, dyn, b->H, b->V, b->count, st);
            case 3: return launch_jvp_brick_batch<3>(t, dyn, b->H, b->V, b->count, st);
            case 4: return launch_jvp_brick_batch<4>(t, dyn, b->H, b->V, b->count, st);
        }
    }
    switch (b->n) {
        case 2: return launch_jvp_batch<2>(t, dyn, b->count, st);
        case 3: return launch_jvp_batch<3>(t, dyn, b->count, st);
        case 4: return launch_jvp_batch<4>(t, dyn, b->count, st);
        case 5: return launch_jvp_batch<5>(t, dyn, b->count, st);
        case 6: return launch_jvp_batch<6>(t, dyn, b->count, st);
        case 7: return launch_jvp_batch<7>(t, dyn, b->count, st);
        case 8: return launch_jvp_batch<8>(t, dyn, b->count, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", b->n);
}

// One Krylov vector of KIOPS with the complex-step Jacobian (solvers/kiops.py:170-207 + solvers/matvec.py:56-61) from
// ONE host call: the two launches of the dual-number JVP of all tiles (aw = scale Im R(q + i eps v), v = V[j-1][:n]) and
// the three of wx_kiops_finish.  For a rank that owns the whole sphere (halos alias the packed edges: no exchange
// between the two JVP launches) at launch-bound sizes - the shipped .ini files - where the host side of five separate
// calls costs more than the kernels.
wx_status wx_euler3d_batch_kiops_vector(const wx_euler3d_batch* b, const double* q, double* V, size_t ldv, int j, size_t n,
                                        int p, int iop, double eps, double scale, const double* uflip, double* hcol,
                                        double* aw, double* workspace, size_t panel_stride, wx_stream stream) {
    if (!b || !q || !V || !aw) return fail(WX_ERR_INVALID, "wx_euler3d_batch_kiops_vector: null argument");
    if (b->dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "wx_euler3d_batch_kiops_vector: the batch must be WX_DUAL128");
    if (j < 1 || n != (size_t)b->count * panel_stride)
        return fail(WX_ERR_INVALID, "wx_euler3d_batch_kiops_vector: vector length %zu != %d tiles x %zu", n, b->count, panel_stride);
    const double* v = V + (size_t)(j - 1) * ldv;
    wx_status s = wx_euler3d_batch_extrap_pack(b, q, v, eps, panel_stride, stream);
    if (s != WX_OK) return s;
    s = wx_euler3d_batch_jvp(b, q, v, eps, aw, scale, panel_stride, WX_REGION_ALL, stream);
    if (s != WX_OK) return s;
    return wx_kiops_finish(V, ldv, j, n, p, iop, aw, uflip, hcol, workspace, stream);
}

// One Krylov vector of fgmres with the finite-difference Rosenbrock operator (integrators/ros2.py:27-30 + solvers/matvec.py:76-88
// + solvers/fgmres.py:160-210) from ONE host call and with no host round trip: row J-1 = A(row J-2 / s) s, A v = v - dt/2
// (R(q + eps v) - R(q)) / eps, s = the lagged norm vn[J-3] read from device memory by the kernels (EulerParams::dscale), then
// wx_fgmres_vector's products, step and update.  For a rank that owns the whole sphere at launch-bound sizes (the shipped .ini
// files).  rq = R(q); half_dt_over_eps = dt / (2 eps).
wx_status wx_euler3d_batch_fgmres_vector(const wx_euler3d_batch* b, const double* q, const double* rq, double* V, size_t ldv, int J,
                                         size_t n, double eps, double half_dt_over_eps, double* R, double* T, double* K, int ld,
                                         double* coef, double* vn, int* flag, double* workspace, size_t panel_stride,
                                         wx_stream stream) {
    if (!b || !q || !rq || !V || !vn) return fail(WX_ERR_INVALID, "wx_euler3d_batch_fgmres_vector: null argument");
    if (b->dtype != WX_F64) return fail(WX_ERR_INVALID, "wx_euler3d_batch_fgmres_vector: the batch must be WX_F64");
    if (J < 3 || n != (size_t)b->count * panel_stride)
        return fail(WX_ERR_INVALID, "wx_euler3d_batch_fgmres_vector: J = %d, vector length %zu != %d tiles x %zu", J, n, b->count, panel_stride);
    WX_STREAM(st, stream);
    const double* z = V + (size_t)(J - 2) * ldv;
    double* w = V + (size_t)(J - 1) * ldv;
    const double* s = vn + (J - 3);
    wx_status st1 = batch_extrap<double>(b, q, z, eps, panel_stride, st, s);
    if (st1 != WX_OK) return st1;
    // out = 1 * z + 0 * (q + eps z / s) - c s R(q + eps z / s) + c s R(q)
    st1 = batch_rhs<double>(b, q, z, eps, z, rq, w, panel_stride, 1, 1.0, 0.0, -half_dt_over_eps, half_dt_over_eps, WX_REGION_ALL, st, s);
    if (st1 != WX_OK) return st1;
    return wx_fgmres_vector(V, ldv, J, n, R, T, K, ld, coef, vn, flag, workspace, nullptr, stream);
}

wx_status wx_euler3d_batch_rhs_axpy2(const wx_euler3d_batch* b, const void* q, const double* v, double eps, const void* y,
                                     const void* z, void* out, size_t panel_stride, int axpy, double a, double bq, double c,
                                     double d, wx_region region, wx_stream stream) {
    if (!b || !q || !out) return fail(WX_ERR_INVALID, "wx_euler3d_batch_rhs_axpy2: null argument");
    if (v && b->dtype != WX_F64) return fail(WX_ERR_INVALID, "a shifted state needs a WX_F64 batch");
    if (out == q) return fail(WX_ERR_INVALID, "wx_euler3d_batch_rhs_axpy2: output must not alias the state");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    WX_STREAM(st, stream);
    switch (b->dtype) {
        case WX_F64: return batch_rhs<double>(b, q, v, eps, y, z, out, panel_stride, axpy, a, bq, c, d, region, st);
        case WX_C128: return batch_rhs<cplx>(b, q, nullptr, 0.0, y, z, out, panel_stride, axpy, a, bq, c, d, region, st);
        case WX_DUAL128: return batch_rhs<dual>(b, q, nullptr, 0.0, y, z, out, panel_stride, axpy, a, bq, c, d, region, st);
    }
    return fail(WX_ERR_INVALID, "bad batch dtype");
}

}  // extern "C"
