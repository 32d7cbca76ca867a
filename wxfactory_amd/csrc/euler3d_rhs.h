// K2 of the 3-D Euler path: phases 3-8 fused (face stage, point stage, the three directional passes, epilogue), the
// general and the column-metric instantiations, the stage-pipeline epilogue, the batched form.
#pragma once

namespace wx {

// ------------------------------------------------------------------------------------------------
// Rusanov common flux at one face point (fluxes.py:326-403 and its j / vertical twins).
// Outputs the seven face quantities the element on the `own` side needs:
//   out[0..3] F* for rho, rho u1, rho u2, rho theta;  out[4] A* (rho w advective);
//   out[5] B*_own = 1/2 (P_L + P_R) / p_own;          out[6] log p_own
// (the common flux of the rho w row itself is never used: rhs_dfr.py:139 overwrites that row).
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void rusanov_face(const T* qL, const T* qR, T unL, T unR, T rL, T rR, double sg, double h0,
                                             double h1, double h2, double hdd, bool own_is_L, bool advection_only,
                                             T* out) {
    // q[0..4] state, q[5] pressure, q[6] log pressure; rL, rR = 1/rho
    const T pL = qL[5], pR = qR[5];
    T eL, eR;
    if (advection_only) {
        eL = T(w_abs(unL));
        eR = T(w_abs(unR));
    } else {
        eL = w_abs(unL) + w_sqrt((hdd * kGamma) * pL * rL);
        eR = w_abs(unR) + w_sqrt((hdd * kGamma) * pR * rR);
    }
    const T eig = w_max(eL, eR);
    const T sguL = sg * unL, sguR = sg * unR;
    const T es = eig * sg;
    const double sgh0 = sg * h0, sgh1 = sg * h1, sgh2 = sg * h2;

    out[0] = 0.5 * (sguL * qL[0] + sguR * qR[0] - es * (qR[0] - qL[0]));
    out[1] = 0.5 * ((sguL * qL[1] + sgh0 * pL) + (sguR * qR[1] + sgh0 * pR) - es * (qR[1] - qL[1]));
    out[2] = 0.5 * ((sguL * qL[2] + sgh1 * pL) + (sguR * qR[2] + sgh1 * pR) - es * (qR[2] - qL[2]));
    out[3] = 0.5 * (sguL * qL[4] + sguR * qR[4] - es * (qR[4] - qL[4]));
    out[4] = 0.5 * (sguL * qL[3] + sguR * qR[3] - es * (qR[3] - qL[3]));
    out[5] = 0.5 * (sgh2 * pL + sgh2 * pR) / w_sel(own_is_L, pL, pR);
    out[6] = w_sel(own_is_L, qL[6], qR[6]);
}

// The same common flux written from the point of view of the element that owns the face point (own / neighbour
// instead of left / right): no left-right copies of the two 7-value states, which is what keeps the dual-number
// JVP kernel inside 128 registers.  plus = the face is the element's upper one (own state is the left one).
template <typename T>
__device__ __forceinline__ void rusanov_own(const T* qo, const T* qn, T uo, T un, T ro, T rn, double sg, double h0, double h1,
                                            double h2, double hdd, bool plus, bool advection_only, T* out) {
    const T po = qo[5], pn = qn[5];
    T eo, en;
    if (advection_only) {
        eo = T(w_abs(uo));
        en = T(w_abs(un));
    } else {
        eo = w_abs(uo) + w_sqrt((hdd * kGamma) * po * ro);
        en = w_abs(un) + w_sqrt((hdd * kGamma) * pn * rn);
    }
    const T eig = w_max(w_sel(plus, eo, en), w_sel(plus, en, eo));  // (left, right) order: the tie-break of numpy.maximum
    const T sguo = sg * uo, sgun = sg * un;
    const T es = (plus ? sg : -sg) * eig;   // eig sqrtG (q_R - q_L) = +-(q_n - q_o)
    const double sgh0 = sg * h0, sgh1 = sg * h1, sgh2 = sg * h2;
    out[0] = 0.5 * (sguo * qo[0] + sgun * qn[0] - es * (qn[0] - qo[0]));
    out[1] = 0.5 * ((sguo * qo[1] + sgh0 * po) + (sgun * qn[1] + sgh0 * pn) - es * (qn[1] - qo[1]));
    out[2] = 0.5 * ((sguo * qo[2] + sgh1 * po) + (sgun * qn[2] + sgh1 * pn) - es * (qn[2] - qo[2]));
    out[3] = 0.5 * (sguo * qo[4] + sgun * qn[4] - es * (qn[4] - qo[4]));
    out[4] = 0.5 * (sguo * qo[3] + sgun * qn[3] - es * (qn[3] - qo[3]));
    out[5] = 0.5 * (sgh2 * po + sgh2 * pn) / po;
    out[6] = qo[6];
}

// Inputs of one face point: the two face states and the interface metric, as loaded.
template <typename T>
struct FaceIn {
    T qo[5], qn[5];
    double sg, h0, h1, h2;
    bool mirror;
};

// (diagnostic build 7: the interface metric from a window too)
template <int N, bool COLM>
__device__ __forceinline__ size_t diag_fold(size_t o) { return (kDiagCacheAll && !COLM) ? o % (size_t)(128 * N * N) : o; }

// The loads of one face point of one element: own slot of the interface buffer; the neighbour element's slot,
// the received halo on a lateral tile edge, or the own state again (mirrored later) at ground / top; the
// interface metric.  Separate from the arithmetic so that a kernel can issue them early.
template <int N, typename T, bool COLM = false, bool G>
__device__ __forceinline__ void face_load(const EulerParams<T, G>& P, const Elem& el, int f, int fp, FaceIn<T>& in) {
    constexpr int N2 = N * N;
    const int H = P.H, V = P.V;
    const int d = f >> 1, plus = f & 1;
    const size_t vsh = (size_t)V * H * N2;  // var stride in a halo edge message

    pp<T, const T, G> own = P.itf + ((size_t)(kDiagCacheState ? el.e & 63 : el.e) * 6 + f) * NQ * N2 + fp;
    pp<T, const T, G> nbr;
    size_t nstride = N2;
    bool mirror = false, from_halo = false;
    pp<T, const double, G> sgp, hp;
    size_t hfs;  // field stride of the h_contra_itf array
    if (d == 0) {
        const int ne = el.ei + (plus ? 1 : -1);
        if (ne >= 0 && ne < H) nbr = P.itf + ((size_t)(kDiagCacheState ? (el.e + 1) & 63 : el.e + (plus ? 1 : -1)) * 6 + (f ^ 1)) * NQ * N2 + fp;
        else { nbr = (plus ? P.halo_e : P.halo_w) + ((size_t)el.ek * H + el.ej) * N2 + fp; nstride = vsh; from_halo = true; }
        // (column form: the interface metric of a lateral face does not depend on the level - one row of n values per
        // face side instead of V n of them; that of a horizontal face neither on the level nor on the side)
        const size_t o = diag_fold<N, COLM>(COLM ? (((size_t)el.ej * (H + 2) + el.ei + 1) * 2 + plus) * N + fp % N
                              : (((size_t)el.ek * H + el.ej) * (H + 2) + el.ei + 1) * 2 * N2 + plus * N2 + fp);
        hfs = COLM ? (size_t)H * (H + 2) * 2 * N : (size_t)V * H * (H + 2) * 2 * N2;
        sgp = P.sgi + o;
        hp = P.hi + 0 * 3 * hfs + o;
    } else if (d == 1) {
        const int ne = el.ej + (plus ? 1 : -1);
        if (ne >= 0 && ne < H) nbr = P.itf + ((size_t)(kDiagCacheState ? (el.e + 2) & 63 : el.e + (plus ? H : -H)) * 6 + (f ^ 1)) * NQ * N2 + fp;
        else { nbr = (plus ? P.halo_n : P.halo_s) + ((size_t)el.ek * H + el.ei) * N2 + fp; nstride = vsh; from_halo = true; }
        const size_t o = diag_fold<N, COLM>(COLM ? ((((size_t)el.ej + 1) * H + el.ei) * 2 + plus) * N + fp % N
                              : (((size_t)el.ek * (H + 2) + el.ej + 1) * H + el.ei) * 2 * N2 + plus * N2 + fp);
        hfs = COLM ? (size_t)(H + 2) * H * 2 * N : (size_t)V * (H + 2) * H * 2 * N2;
        sgp = P.sgj + o;
        hp = P.hj + 1 * 3 * hfs + o;
    } else {
        const int ne = el.ek + (plus ? 1 : -1);
        if (ne >= 0 && ne < V) nbr = P.itf + ((size_t)(kDiagCacheState ? (el.e + 3) & 63 : el.e + (plus ? H * H : -H * H)) * 6 + (f ^ 1)) * NQ * N2 + fp;
        else { nbr = own; mirror = true; }
        const size_t o = diag_fold<N, COLM>(COLM ? ((size_t)el.ej * H + el.ei) * N2 + fp
                              : ((((size_t)el.ek + 1) * H + el.ej) * H + el.ei) * 2 * N2 + plus * N2 + fp);
        hfs = COLM ? (size_t)H * H * N2 : (size_t)(V + 2) * H * H * 2 * N2;
        sgp = P.sgk + o;
        hp = P.hk + 2 * 3 * hfs + o;
    }
    bool split = false;
    if constexpr (std::is_same<T, dual>::value) split = P.split == 1;
    if constexpr (std::is_same<T, dual>::value) {
        if (split) {
            // prepared JVP: values from the cache of the linearisation state, tangents from this product's buffers.
            // The pointers computed above index [..][5][n^2] arrays of T; the same offsets address the real arrays.
            const size_t oo = (size_t)(own - P.itf);
            const double *ov = P.fv + oo, *ot = P.ft + oo, *nv, *nt;
            if (mirror) { nv = ov; nt = ot; }
            else if (!from_halo) { const size_t no = (size_t)(nbr - P.itf); nv = P.fv + no; nt = P.ft + no; }
            else {
                const T* hb = d == 0 ? (plus ? P.halo_e : P.halo_w) : (plus ? P.halo_n : P.halo_s);
                const double* hvb = d == 0 ? (plus ? P.hv_e : P.hv_w) : (plus ? P.hv_n : P.hv_s);
                const size_t no = (size_t)(nbr - hb);
                nv = hvb + no; nt = reinterpret_cast<const double*>(hb) + no;
            }
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                in.qo[v] = dual(ov[v * N2], ot[v * N2]);
                in.qn[v] = dual(nv[v * nstride], nt[v * nstride]);
            }
        }
    }
    if (!split) {
        if (kNoVertFaces && d == 2) {   // diagnostic: a plausible state without a load
#pragma unroll
            for (int v = 0; v < 5; ++v) { in.qo[v] = T(v == 0 ? 1.0 : (v == 4 ? 300.0 : 1e-5)); in.qn[v] = in.qo[v]; }
        } else {
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                in.qo[v] = own[v * N2];
                in.qn[v] = nbr[v * nstride];
            }
        }
    }
    in.sg = *sgp; in.h0 = hp[0]; in.h1 = hp[hfs]; in.h2 = hp[2 * hfs];
    in.mirror = mirror;
}

// The Rusanov problem of one face point from its loaded inputs.  out[0..6] as in rusanov_face.
template <typename T, bool OWN_FORM = false>
__device__ __forceinline__ void face_flux(const FaceIn<T>& in, int f, bool advection_only, T* out) {
    const int d = f >> 1, plus = f & 1;
    T qo[7], qn[7];
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        qo[v] = in.qo[v];
        qn[v] = in.qn[v];
    }
    // pressures from rho*theta on both sides (pde_euler_cubesphere.py:158-160)
    const T go = kGamma * w_log(qo[4] * kRdOverP0), gn = kGamma * w_log(qn[4] * kRdOverP0);
    qo[5] = kP0 * w_exp(go); qn[5] = kP0 * w_exp(gn);
    qo[6] = kLogP0 + go; qn[6] = kLogP0 + gn;
    const double sg = in.sg, h0 = in.h0, h1 = in.h1, h2 = in.h2;
    if (kSkelFace) {   // diagnostic builds: every load consumed, no Riemann arithmetic
        T sum = T(sg + h0 + h1 + h2);
#pragma unroll
        for (int v = 0; v < 5; ++v) sum += in.qo[v] + in.qn[v];
#pragma unroll
        for (int c = 0; c < 7; ++c) out[c] = sum;
        return;
    }
    const double hdd = d == 0 ? h0 : (d == 1 ? h1 : h2);
    const T ro = 1.0 / qo[0], rn = 1.0 / qn[0];
    // (explicit selects: a run-time index into a register array of 16-byte values goes to scratch)
    T uo = w_sel(d == 0, qo[1], w_sel(d == 1, qo[2], qo[3])) * ro;
    T un = w_sel(d == 0, qn[1], w_sel(d == 1, qn[2], qn[3])) * rn;
    if (in.mirror) un = -uo;  // no-flow wall: odd symmetry of w (pde_euler_cubesphere.py:150-156)
    if (OWN_FORM) {
        rusanov_own<T>(qo, qn, uo, un, ro, rn, sg, h0, h1, h2, hdd, plus != 0, advection_only, out);
        return;
    }
    // left = plus-side state of the lower element, right = minus-side state of the upper one
    // (by value with selects: passing swapped array pointers would push both arrays to scratch)
    T qL[7], qR[7];
#pragma unroll
    for (int v = 0; v < 7; ++v) {
        qL[v] = w_sel(plus != 0, qo[v], qn[v]);
        qR[v] = w_sel(plus != 0, qn[v], qo[v]);
    }
    const bool pl = plus != 0;
    rusanov_face<T>(qL, qR, w_sel(pl, uo, un), w_sel(pl, un, uo), w_sel(pl, ro, rn), w_sel(pl, rn, ro), sg, h0, h1, h2, hdd,
                    pl, advection_only, out);
}

// One face point of one element, loads + arithmetic.  Shared by the fused RHS kernel and the JVP kernel.
template <int N, typename T, bool OWN_FORM = false, bool COLM = false, bool G>
__device__ __forceinline__ void face_problem(const EulerParams<T, G>& P, const Elem& el, int f, int fp, T* out) {
    FaceIn<T> in;
    face_load<N, T, COLM>(P, el, f, fp, in);
    face_flux<T, OWN_FORM>(in, f, P.advection_only, out);
}

// ------------------------------------------------------------------------------------------------
// K2: fused phases 3-8.  The body is a sequence of stages, each a device function below:
//   face loads (n = 8: issued first)  ->  point loads  ->  face stage (Riemann problems -> LDS)  ->  pointwise
//   quantities + forcing  ->  three directional passes (matrix cores or vector pipe)  ->  epilogue (scaling, fused
//   stage update, optional filter + NaN flag, store, optional extrapolation of the output for the next stage).
// ------------------------------------------------------------------------------------------------
// what a thread holds of its solution point after the loads
template <typename T>
struct PointIn {
    T q0, q1, q2, q3, q4;
    double sg, h00, h01, h02, h11, h12, h22;
};

// (om, fsm: offset and field stride of the point in the metric arrays - those of the state, or of the column slabs)
template <typename T, bool CACHED = false, bool G>
__device__ __forceinline__ void k2_point_loads(const EulerParams<T, G>& P, bool active, size_t o, size_t fs, PointIn<T>& S,
                                               size_t om, size_t fsm) {
    S.q0 = T(1.0); S.q1 = T(0.0); S.q2 = T(0.0); S.q3 = T(0.0); S.q4 = T(1.0);
    S.sg = 1.0; S.h00 = S.h01 = S.h02 = S.h11 = S.h12 = S.h22 = 0.0;
    if (active) {
        load_state<T>(P, o, fs, S.q0, S.q1, S.q2, S.q3, S.q4);
        S.sg = ldm_if<CACHED>(P.sg + om);
        S.h00 = ldm_if<CACHED>(P.h + 0 * fsm + om); S.h01 = ldm_if<CACHED>(P.h + 1 * fsm + om);
        S.h02 = ldm_if<CACHED>(P.h + 2 * fsm + om); S.h11 = ldm_if<CACHED>(P.h + 4 * fsm + om);
        S.h12 = ldm_if<CACHED>(P.h + 5 * fsm + om); S.h22 = ldm_if<CACHED>(P.h + 8 * fsm + om);
    }
}
template <typename T, bool G>
__device__ __forceinline__ void k2_point_loads(const EulerParams<T, G>& P, bool active, size_t o, size_t fs, PointIn<T>& S) {
    k2_point_loads<T, false>(P, active, o, fs, S, o, fs);
}

// forcing of the three momentum rows, all but the gravity filter (pde_euler_cubesphere.py:12-25, 203-290), from the 27
// (18 on a non-rotating planet) Christoffel fields, all loads in flight together; gcoef = inv_dzdeta * g
template <typename T, bool CACHED = false, bool WITH_IDZ = true, bool G>
__device__ __forceinline__ void k2_forcing(const EulerParams<T, G>& P, bool active, size_t o, size_t fs, const PointIn<T>& S, T u1,
                                           T u2, T u3, T p, T& fc0, T& fc1, T& fc2, double& gcoef, size_t om, size_t fsm) {
    double cg[27], idzv = 0.0;
#if defined(WX_BRICK_DIAG) && WX_BRICK_DIAG == 4
    active = false;
#endif
    if (active && P.rot_zero) {   // non-rotating planet: the 9 rotation symbols are identically zero
#pragma unroll
        for (int i = 0; i < 27; ++i) cg[i] = (i % 9) < 3 ? 0.0 : ldm_if<CACHED>(P.chr + (size_t)i * fsm + om);
        if (WITH_IDZ) idzv = ldm_if<CACHED>(P.idz + om);
    } else if (active) {
#pragma unroll
        for (int i = 0; i < 27; ++i) cg[i] = ldm_if<CACHED>(P.chr + (size_t)i * fsm + om);
        if (WITH_IDZ) idzv = ldm_if<CACHED>(P.idz + om);
    } else {
#pragma unroll
        for (int i = 0; i < 27; ++i) cg[i] = 0.0;
    }
    fc0 = T(0.0); fc1 = T(0.0); fc2 = T(0.0);
    gcoef = 0.0;
    if (active) {
        const T q0 = S.q0;
        T fc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double* c = cg + i * 9;
            const double c01 = c[0], c02 = c[1], c03 = c[2], c11 = c[3], c12 = c[4], c13 = c[5],
                         c22 = c[6], c23 = c[7], c33 = c[8];
            fc[i] = 2.0 * q0 * (c01 * u1 + c02 * u2 + c03 * u3) + c11 * (q0 * u1 * u1 + S.h00 * p) +
                    2.0 * c12 * (q0 * u1 * u2 + S.h01 * p) + 2.0 * c13 * (q0 * u1 * u3 + S.h02 * p) +
                    c22 * (q0 * u2 * u2 + S.h11 * p) + 2.0 * c23 * (q0 * u2 * u3 + S.h12 * p) +
                    c33 * (q0 * u3 * u3 + S.h22 * p);
        }
        if (P.has_damp) {
            const T dw = P.dcoef[o] * q0;
            fc[0] += dw * (u1 - P.duref[o]);
            fc[1] += dw * (u2 - P.duref[fs + o]);
            fc[2] += dw * (u3 - P.duref[2 * fs + o]);
        }
        fc0 = fc[0]; fc1 = fc[1]; fc2 = fc[2];
        gcoef = idzv * kGravity;
    }
}
template <typename T, bool G>
__device__ __forceinline__ void k2_forcing(const EulerParams<T, G>& P, bool active, size_t o, size_t fs, const PointIn<T>& S, T u1,
                                           T u2, T u3, T p, T& fc0, T& fc1, T& fc2, double& gcoef) {
    k2_forcing<T, false>(P, active, o, fs, S, u1, u2, u3, p, fc0, fc1, fc2, gcoef, o, fs);
}

template <int N, typename T, bool PIPE, bool COLM = false, bool G3 = false, bool G>
__device__ __forceinline__ void euler_rhs_body(const EulerParams<T, G>& P) {
    using C = Cfg<N>;
    static_assert(!COLM || std::is_same<T, double>::value, "the column form: float64");
    const int bx = COLM ? xcd_slab_block(blockIdx.x, gridDim.x >> 3) : (int)blockIdx.x;
    constexpr int N2 = C::N2, N3 = C::N3, EPB = C::EPB, BS = C::BS;
    constexpr int NF = 8;   // staged fields: 4 F rows, A, B (per direction) + log p + sqrtG*rho
    constexpr int NC = 7;   // face quantities, see rusanov_face
    // matrix-core path for the derivative contractions (n = 8, float64); everything else keeps the vector path
    constexpr bool MF = WX_MFMA && N == 8 && std::is_same<T, double>::value;
    static_assert(!MF || (EPB == 1 && C::LE == kMfLE), "the MFMA pass owns one n = 8 element per workgroup");
    constexpr int FST = NC * N2 + (MF ? kMfFS - 7 * 64 : 0);   // doubles per face in the face-flux image
    // One LDS block: the field images, then the face-flux image.  Matrix-core path: 7 images suffice - the eighth field
    // (sqrtG rho, vertical pass only) lands on the face fluxes of the first direction, which are dead by then - and the
    // operator tables are not needed (they sit in the lanes' MFMA operands): 54.5 KB.
    constexpr int NFI = MF ? 7 : NF;
    __shared__ T smem[NFI * EPB * C::LE + EPB * 6 * FST];
    T(*fld)[EPB * C::LE] = reinterpret_cast<T(*)[EPB * C::LE]>(smem);
    T* frs = smem + NFI * EPB * C::LE;
#define WX_FR(le_, f_, c_, fp_) frs[((le_) * 6 + (f_)) * FST + (c_) * N2 + (fp_)]
    __shared__ double sD[MF ? 1 : N * N], sHF[MF ? 1 : N * N], sCm[MF ? 1 : N], sCp[MF ? 1 : N];
    __shared__ double sEF[(PIPE && !MF) ? N * N : 1];

    const int tid = threadIdx.x;
    __builtin_assume(tid < (int)Cfg<N>::BS);   // (the launch bounds: lets one-element workgroups drop their `le < EPB` guards)
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;
    if (PIPE && !MF && P.efilter)
        for (int i = tid; i < N * N; i += BS) sEF[i] = P.K->EF[i];
#if WX_K2_DIAG == 1
#define WX_STAMP(i)                                                                           \
    do {                                                                                      \
        __syncthreads();                                                                      \
        if (tid == 0 && P.stamps)                                                             \
            P.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64();     \
    } while (0)
#else
#define WX_STAMP(i)
#endif
    WX_STAMP(0);

    if (!MF) {
        for (int i = tid; i < N * N; i += BS) {
            sD[i] = P.K->D[i];
            sHF[i] = P.K->HF[i];
        }
        if (tid < N) {
            sCm[tid] = P.K->cm[tid];
            sCp[tid] = P.K->cp[tid];
        }
    }

    const int le = tid / N3, pt = tid % N3;
    const Elem el = COLM ? decode_elem_col(bx * EPB + le, P.count, P.region, H, V, P.md_v, P.md_h, P.md_w) : decode_blk<EPB, G3>(P, bx * EPB + le, P.count, P.region);
    const bool active = (le < EPB) && el.valid;
    const int kl = pt / N2, jl = (pt / N) % N, il = pt % N;
    const int lb = (le < EPB ? le : 0) * C::LE;  // LDS base of this thread's element
    const int lpt = lb + C::lidx(kl, jl, il);    // this thread's node in the LDS image
    const int lptm = mf_idx(kl, jl, il);         // ... and in the image of the matrix-core passes
    MfOps4 mops4{0.0, 0.0, 0.0, 0.0, 0.0};
    const size_t o = (size_t)el.e * N3 + pt;

    // ---- loads.  n = 8: one face point per thread (384 of 512) and the face loads go FIRST: vector-memory results return
    // in issue order, so the face stage (the first consumer) does not wait for the twelve point loads queued behind
    constexpr bool FACE_FIRST = N == 8 && EPB == 1;
    FaceIn<T> fin_first;
    int ff_first = 0;
    if constexpr (FACE_FIRST) {
        ff_first = __builtin_amdgcn_readfirstlane(tid / N2);
        if (tid < 6 * N2 && el.valid) face_load<N, T, COLM>(P, el, ff_first, tid % N2, fin_first);
    }
    PointIn<T> S;
    // metric offsets: the point's own, or - column form - its place in the column's (n x n) slab
    const size_t om = COLM ? ((size_t)el.ej * H + el.ei) * N2 + pt % N2 : o;
    const size_t fsm = COLM ? (size_t)H * H * N2 : fs;
    // (diagnostic builds 6 / 7: the loads come from a window of 64 elements; the stores go where they belong)
    const size_t o_ld = kDiagCacheState ? (size_t)(el.e & 63) * N3 + pt : o;
    const size_t om_ld = (kDiagCacheAll && !COLM) ? o_ld : om;
    k2_point_loads<T, COLM>(P, active, o_ld, fs, S, om_ld, fsm);   // in flight while the face stage computes
    const T q0 = S.q0, q1 = S.q1, q2 = S.q2, q3 = S.q3, q4 = S.q4;
    const double sg = S.sg;

    // ---- face stage: Riemann problems of all 6 faces of the block's elements -> LDS
    if constexpr (FACE_FIRST) {
        if (tid < 6 * N2 && el.valid) {
            T out[NC];
            face_flux<T>(fin_first, ff_first, P.advection_only, out);
#pragma unroll
            for (int c = 0; c < NC; ++c) WX_FR(0, ff_first, c, tid % N2) = out[c];
        }
    }
    for (int fi = tid; fi < (FACE_FIRST ? 0 : EPB * 6 * N2); fi += BS) {
        const int fle = fi / (6 * N2);
        const int r = fi % (6 * N2);
        int f = r / N2;
        const int fp = r % N2;
        if (N2 % 64 == 0 && BS % 64 == 0) f = __builtin_amdgcn_readfirstlane(f);
        const Elem fel = COLM ? decode_elem_col(bx * EPB + fle, P.count, P.region, H, V, P.md_v, P.md_h, P.md_w) : decode_blk<EPB, G3>(P, bx * EPB + fle, P.count, P.region);
        if (!fel.valid) continue;
        T out[NC];
        face_problem<N, T, false, COLM>(P, fel, f, fp, out);
#pragma unroll
        for (int c = 0; c < NC; ++c) WX_FR(fle, f, c, fp) = out[c];
    }
    WX_STAMP(1);
    // the operator fragments of the matrix-core passes (five doubles per lane, from cache): loaded HERE, behind the face stage -
    // ten registers fewer across the kernel's register peak (at the top of the kernel the stage-pipeline instantiation spilled
    // one of them behind a full vector-memory wait, in front of its first loads), their latency under the point stage
    // (the two high-filter fragments, wanted by the third pass only, follow at the top of the second)
    if (MF) mops4 = mf4_load_ops(P.K->D, P.K->cm, P.K->cp, nullptr, tid & 63);

    // ---- pointwise quantities
    const T rinv = 1.0 / q0;
    const T u1 = q1 * rinv, u2 = q2 * rinv, u3 = q3 * rinv;
    const T glog = kGamma * w_log(kRdOverP0 * q4);
    const T p = kP0 * w_exp(glog);
    const T logp = kLogP0 + glog;  // log p, without a second logarithm
    if (!MF && le < EPB) {
        fld[6][lpt] = logp;
        fld[7][lpt] = sg * q0;
    }

    // ---- forcing
    T fc0, fc1, fc2;
    double gcoef;
    // (matrix-core path: inv_dzdeta, wanted by the epilogue only, is loaded at the top of the third pass - a value that lives
    // from the forcing to the epilogue was the one the stage-pipeline instantiation spilled, behind a full vector-memory wait)
    k2_forcing<T, COLM, !MF>(P, active, o, fs, S, u1, u2, u3, p, fc0, fc1, fc2, gcoef, om_ld, fsm);
    WX_STAMP(2);

    // accumulators of sum_d dF^d; the forcing is folded in as sqrtG*f so that the final
    // -1/sqrtG scaling yields  -1/sqrtG sum_d dF^d - f  (keeps 4 values out of the hot loop)
    T acc0 = T(0.0), acc1 = sg * fc0, acc2 = sg * fc1, acc4 = T(0.0), accw = sg * fc2;
    T hf = T(0.0);

    if (kSkelDirs) {   // diagnostic builds: the staged data consumed, no passes
        if (MF) fld[6][lpt] = logp;
        __syncthreads();
        acc0 += WX_FR(le < EPB ? le : 0, 0, 0, pt % N2) + fld[6][lpt];
    }
    // ---- three directional passes, one copy per direction (constant LDS strides: the reads pair up as ds_read2_b64)
#pragma unroll
    for (int d = 0; d < (kSkelDirs ? 0 : 3); ++d) {
        const T ud = w_sel(d == 0, u1, w_sel(d == 1, u2, u3));
        const double hd0 = d == 0 ? S.h00 : (d == 1 ? S.h01 : S.h02);
        const double hd1 = d == 0 ? S.h01 : (d == 1 ? S.h11 : S.h12);
        const double hd2 = d == 0 ? S.h02 : (d == 1 ? S.h12 : S.h22);
        const T sgu = sg * ud;
        const T Bd = T(sg * hd2);
        if constexpr (MF) {
            if (d == 1) {
                const MfOps4 hf = mf4_load_ops(P.K->D, nullptr, nullptr, P.K->HF, tid & 63);
                mops4.h0 = hf.h0; mops4.h1 = hf.h1;
            }
            if (d == 2 && active) gcoef = ldm_if<COLM>(P.idz + om_ld) * kGravity;
            // matrix-core pass (mf4_dir_pass): each thread stages its own node, the 8 waves contract all lines in place -
            // D | cm | cp with the two common face values as a third k-step -, each thread picks its own node up again:
            // no barrier between a thread's read and its next write
            double* fm = reinterpret_cast<double*>(&fld[0][0]);
            const double* fq = reinterpret_cast<const double*>(&frs[0]);
            fm[0 * kMfLE + lptm] = sgu * q0;
            fm[1 * kMfLE + lptm] = sgu * q1 + (sg * hd0) * p;
            fm[2 * kMfLE + lptm] = sgu * q2 + (sg * hd1) * p;
            fm[3 * kMfLE + lptm] = sgu * q4;
            fm[4 * kMfLE + lptm] = sgu * q3;
            fm[5 * kMfLE + lptm] = Bd;
            fm[6 * kMfLE + lptm] = logp;
            if (d == 2) fm[7 * kMfLE + lptm] = sg * q0;
            __syncthreads();
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            if (d == 0) mf4_dir_pass<0, true, 7, true, kMfFS, kMfFieldBatch>(fm, fq, mops4, wave, tid & 63);
            else if (d == 1) mf4_dir_pass<1, true, 7, true, kMfFS, kMfFieldBatch>(fm, fq, mops4, wave, tid & 63);
            else mf4_dir_pass<2, true, 7, true, kMfFS, kMfFieldBatch>(fm, fq, mops4, wave, tid & 63);
            __syncthreads();
            const double r0 = fm[0 * kMfLE + lptm], r1 = fm[1 * kMfLE + lptm], r2 = fm[2 * kMfLE + lptm],
                         r3 = fm[3 * kMfLE + lptm], r4 = fm[4 * kMfLE + lptm], r5 = fm[5 * kMfLE + lptm],
                         r6 = fm[6 * kMfLE + lptm];
            // W^d = [A@D + A*@C] + p [B@D + B*@C] + p B [log p@D + log p^@C]  (rhs_dfr.py:113-136)
            acc0 += r0; acc1 += r1; acc2 += r2; acc4 += r3;
            accw += r4 + p * r5 + (p * Bd) * r6;
            if (d == 2) hf = fm[7 * kMfLE + lptm];
            WX_STAMP(3 + d);
            continue;
        }
        if (d > 0) __syncthreads();  // previous direction's reads are done
        if (le < EPB) {
            fld[0][lpt] = sgu * q0;
            fld[1][lpt] = sgu * q1 + (sg * hd0) * p;
            fld[2][lpt] = sgu * q2 + (sg * hd1) * p;
            fld[3][lpt] = sgu * q4;
            fld[4][lpt] = sgu * q3;
            fld[5][lpt] = Bd;
        }
        __syncthreads();

        int idx, fp, base, stride;
        if (d == 0) { idx = il; fp = kl * N + jl; base = lb + C::lidx(kl, jl, 0); stride = 1; }
        else if (d == 1) { idx = jl; fp = kl * N + il; base = lb + C::lidx(kl, 0, il); stride = C::NP; }
        else { idx = kl; fp = jl * N + il; base = lb + C::lidx(0, jl, il); stride = N * C::NP; }

        double dm[N];
#pragma unroll
        for (int m = 0; m < N; ++m) dm[m] = sD[idx * N + m];
        const double cm = sCm[idx], cp = sCp[idx];
        const int lf = le < EPB ? le : 0;
        // rolled over field batches: bounds the LDS reads in flight (register pressure); fully
        // unrolled, the compiler clusters 70 LDS reads and needs 241 VGPRs (1 workgroup/CU)
        constexpr int FB = is_complex<T>::value ? kFieldBatchWide : kFieldBatch;
        const T pB = p * Bd;
#pragma unroll 1
        for (int c0 = 0; c0 < 7; c0 += FB) {
#pragma unroll
            for (int cc = 0; cc < FB; ++cc) {
                const int c = c0 + cc;
                if (c < 7) {
                    T a = cm * WX_FR(lf, 2 * d, c, fp) + cp * WX_FR(lf, 2 * d + 1, c, fp);
#pragma unroll
                    for (int m = 0; m < N; ++m) a += dm[m] * fld[c][base + m * stride];
                    // W^d = [A@D + A*@C] + p [B@D + B*@C] + p B [log p@D + log p^@C]  (rhs_dfr.py:113-136)
                    if (c == 0) acc0 += a;
                    else if (c == 1) acc1 += a;
                    else if (c == 2) acc2 += a;
                    else if (c == 3) acc4 += a;
                    else if (c == 4) accw += a;
                    else if (c == 5) accw += a * p;
                    else accw += a * pB;
                }
            }
        }
        if (d == 2) {
#pragma unroll
            for (int m = 0; m < N; ++m) hf += sHF[idx * N + m] * fld[7][base + m * stride];
        }
        WX_STAMP(3 + d);
    }

    // ---- epilogue
    const double inv_sg = 1.0 / sg;
    accw += gcoef * hf;  // gravity: inv_dzdeta * g * 1/sqrtG * HF_k(sqrtG rho)
    T r0 = -inv_sg * acc0, r1 = -inv_sg * acc1, r2 = -inv_sg * acc2, r3 = -inv_sg * accw, r4 = -inv_sg * acc4;
    if (P.advection_only) { r0 = r1 = r2 = r3 = r4 = T(0.0); }
    if (active && P.axpy) {  // fused stage update of an explicit Runge-Kutta scheme (integrators/tvdrk3.py:12-19)
        const double sdev = P.dscale ? *P.dscale : 1.0;   // (fgmres' device pass: see EulerParams::dscale)
        const double cc = P.cc * sdev, cd = P.cd * sdev;
        r0 = P.cb * q0 + cc * r0; r1 = P.cb * q1 + cc * r1; r2 = P.cb * q2 + cc * r2;
        r3 = P.cb * q3 + cc * r3; r4 = P.cb * q4 + cc * r4;
        if (P.y != nullptr) {
            r0 += P.ca * P.y[o]; r1 += P.ca * P.y[fs + o]; r2 += P.ca * P.y[2 * fs + o];
            r3 += P.ca * P.y[3 * fs + o]; r4 += P.ca * P.y[4 * fs + o];
        }
        if (P.z != nullptr) {
            r0 += cd * P.z[o]; r1 += cd * P.z[fs + o]; r2 += cd * P.z[2 * fs + o];
            r3 += cd * P.z[3 * fs + o]; r4 += cd * P.z[4 * fs + o];
        }
    }
    if (PIPE && P.efilter) {
        // the per-step exponential filter (operators.py:114-119, 257-261) on the stage's output while it is in
        // registers: ((sqrtG q) F_i F_j F_k) / sqrtG through the LDS images the directional passes are done with
        T t0 = active ? sg * r0 : T(0.0), t1 = active ? sg * r1 : T(0.0), t2 = active ? sg * r2 : T(0.0),
          t3 = active ? sg * r3 : T(0.0), t4 = active ? sg * r4 : T(0.0);
        if constexpr (MF) {   // the three filter passes on the matrix cores, like the derivative passes above
            double* fm = reinterpret_cast<double*>(&fld[0][0]);
            const MfOps4 fops = mf4_load_ops(P.K->EF, nullptr, nullptr, nullptr, tid & 63);
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                fm[0 * kMfLE + lptm] = t0; fm[1 * kMfLE + lptm] = t1; fm[2 * kMfLE + lptm] = t2;
                fm[3 * kMfLE + lptm] = t3; fm[4 * kMfLE + lptm] = t4;
                __syncthreads();
                if (d == 0) mf4_dir_pass<0, false, 5, false>(fm, fm, fops, wave, tid & 63);
                else if (d == 1) mf4_dir_pass<1, false, 5, false>(fm, fm, fops, wave, tid & 63);
                else mf4_dir_pass<2, false, 5, false>(fm, fm, fops, wave, tid & 63);
                __syncthreads();
                t0 = fm[0 * kMfLE + lptm]; t1 = fm[1 * kMfLE + lptm]; t2 = fm[2 * kMfLE + lptm];
                t3 = fm[3 * kMfLE + lptm]; t4 = fm[4 * kMfLE + lptm];
            }
        }
#pragma unroll
        for (int d = 0; d < (MF ? 0 : 3); ++d) {
            __syncthreads();  // previous reads of fld are done
            if (le < EPB) {
                fld[0][lpt] = t0; fld[1][lpt] = t1; fld[2][lpt] = t2; fld[3][lpt] = t3; fld[4][lpt] = t4;
            }
            __syncthreads();
            int base, stride, idx;
            if (d == 0) { base = lb + C::lidx(kl, jl, 0); stride = 1; idx = il; }
            else if (d == 1) { base = lb + C::lidx(kl, 0, il); stride = C::NP; idx = jl; }
            else { base = lb + C::lidx(0, jl, il); stride = N * C::NP; idx = kl; }
            t0 = t1 = t2 = t3 = t4 = T(0.0);
#pragma unroll
            for (int m = 0; m < N; ++m) {
                const double w = sEF[idx * N + m];
                t0 += w * fld[0][base + m * stride]; t1 += w * fld[1][base + m * stride];
                t2 += w * fld[2][base + m * stride]; t3 += w * fld[3][base + m * stride];
                t4 += w * fld[4][base + m * stride];
            }
        }
        r0 = t0 * inv_sg; r1 = t1 * inv_sg; r2 = t2 * inv_sg; r3 = t3 * inv_sg; r4 = t4 * inv_sg;
        if (active && P.nan_flag != nullptr &&
            (w_real(r0) != w_real(r0) || w_real(r1) != w_real(r1) || w_real(r2) != w_real(r2) ||
             w_real(r3) != w_real(r3) || w_real(r4) != w_real(r4)))
            *P.nan_flag = 1;   // many writers, one value: a plain store is as good as an atomic OR (never cleared here)
    }
    if (active) {
        store_r<T>(P, o, r0);
        store_r<T>(P, fs + o, r1);
        store_r<T>(P, 2 * fs + o, r2);
        store_r<T>(P, 3 * fs + o, r3);
        store_r<T>(P, 4 * fs + o, r4);
    }
    WX_STAMP(6);
    // ---- stage pipeline: the output is the next stage's state; extrapolate it to the faces now, while it
    // is in registers (saves the next evaluation's K1: one read of Q and a launch)
    if (PIPE) {  // (a separate instantiation: the plain kernel keeps its instruction schedule)
        __syncthreads();  // the last directional pass has finished reading fld
        if (le < EPB) {
            fld[0][lpt] = active ? w_log(r0) : T(0.0);
            fld[1][lpt] = r1;
            fld[2][lpt] = r2;
            fld[3][lpt] = r3;
            fld[4][lpt] = active ? w_log(r4) : T(0.0);
        }
        __syncthreads();
        extrap_faces<N, T, COLM, G3>(P, fld, bx * EPB, P.count, P.region, P.itf_out, P.nsend_s, P.nsend_n, P.nsend_w, P.nsend_e);
    }
#undef WX_STAMP
#undef WX_FR
}

template <int N, typename T>
constexpr int k2_waves() { return is_complex<T>::value ? 2 : kK2Waves; }

template <int N, typename T, bool PIPE>
__global__ __launch_bounds__(Cfg<N>::BS, (k2_waves<N, T>())) void euler_rhs_kernel(const EulerParams<T> P) {
    euler_rhs_body<N, T, PIPE, false, grid3_for<N>()>(P);
}

template <int N, typename T>
__global__ __launch_bounds__(Cfg<N>::BS, (k2_waves<N, T>())) void euler_rhs_batch_kernel(const EulerParams<T>* table,
                                                                                         const EulerBatchDyn<T> dyn) {
    auto patch = [&](auto& P) {
        const size_t off = (size_t)blockIdx.y * dyn.stride;
        batch_state<T>(P, dyn);
        P.rhs = dyn.rhs ? dyn.rhs + off : (T*)nullptr;
        P.y = dyn.y ? dyn.y + off : (const T*)nullptr;
        P.z = dyn.z ? dyn.z + off : (const T*)nullptr;
        P.region = dyn.region; P.count = dyn.count;
        P.axpy = dyn.axpy; P.ca = dyn.ca; P.cb = dyn.cb; P.cc = dyn.cc; P.cd = dyn.cd;
    };
    if constexpr (std::is_same<T, double>::value) {   // (float64: the register copy fits and is faster; 16-byte dtypes spilled)
        // the block comes out of memory: typed so that every access through its pointers is a global one (wx_common.h: gp)
        EulerParams<T, true> P = *reinterpret_cast<const EulerParams<T, true>*>(table + blockIdx.y);
        patch(P);
        euler_rhs_body<N, T, false>(P);
    } else {
        __shared__ EulerParams<T> sP;
        euler_rhs_body<N, T, false>(batch_params<T>(sP, table, patch));
    }
}

}  // namespace wx
