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

namespace wx {

constexpr int kMaxN = 8;
constexpr int NQ = 5;   // values per face point in the interface buffer and in the edge messages: the prognostic variables

// measured optima (A/B on MI355X): compile-time constants, not build knobs
constexpr int kK1Waves = 1;           // min waves per SIMD requested for the extrapolation kernel
constexpr int kK2Waves = 4;           // ... for the fused kernel, 8-byte dtypes (n = 8: two workgroups of 8 waves per CU)
constexpr int kJvpWaves = 4;          // ... for the JVP kernel
constexpr int kFieldBatch = 3;        // vector-pipe passes: fields contracted per rolled batch (all 7 at once: 241 VGPRs)
constexpr int kFieldBatchWide = 4;    // ... 16-byte dtypes (one workgroup per CU: a little more ILP pays)
constexpr int kMfFieldBatch = 8;      // matrix-core passes of the fused kernel: fields whose operands are in flight together
constexpr int kJvpMfFieldBatch = 4;   // ... of the JVP kernel
constexpr bool kSkelFace = WX_K2_DIAG == 2 || WX_K2_DIAG == 3;
constexpr bool kSkelDirs = WX_K2_DIAG == 2 || WX_K2_DIAG == 4;
// 5: the ceiling of every design that keeps the VERTICAL face states on chip (a workgroup walking a column, the top face handed
// to the next element through LDS): the extrapolation kernel neither computes nor stores them, the fused kernel never loads
// them (wrong results) - what such a design could gain before it pays for its own work (profiles/r04_vertical_faces_ceiling.txt)
constexpr bool kNoVertFaces = WX_K2_DIAG == 5;

// the streamed-once static fields go through non-temporal loads
__device__ __forceinline__ double ldm(const double* p) { return __builtin_nontemporal_load(p); }
// ... unless they are REUSED: the column slabs of a column-invariant metric are read by all n levels of an element (the
// same CU) and by the V elements of the column (the same XCD): cached loads
template <bool CACHED>
__device__ __forceinline__ double ldm_if(const double* p) { return CACHED ? *p : __builtin_nontemporal_load(p); }
template <int N>
struct Cfg {
    static constexpr int N2 = N * N;
    static constexpr int N3 = N * N * N;
    // elements per workgroup: whole elements, <= 256 points unless one element is larger
    static constexpr int EPB = (N3 >= 216) ? 1 : (256 / N3);   // (n = 3 with 8 instead of 9 elements: fused kernel -5 %, extrapolation +9 %)
    static constexpr int BS = ((EPB * N3 + 63) / 64) * 64;
    // LDS image of one element's nodal field: rows of N nodes padded to an odd length so that
    // line reads along i (lane stride = one row) do not pile onto a few banks (N=8: 8-way -> none).
    // Not for n = 2, 4: there the padding's LDS costs a workgroup per CU (n = 4: 42.3 KB -> 38.2 KB, 3 -> 4 workgroups,
    // fused kernel 113 -> 99 us on the reference's benchmark size; n = 2 JVP kernel 1 -> 2 workgroups, matvec 0.48 -> 0.33 ms)
    // and the conflicts it would avoid are at most two-way
    static constexpr int NP = (N % 2 == 0 && N >= 6) ? N + 1 : N;
    static constexpr int LE = N2 * NP;  // doubles per element image
    __host__ __device__ static constexpr int lidx(int kl, int jl, int il) { return (kl * N + jl) * NP + il; }
};

enum { F_W = 0, F_E = 1, F_S = 2, F_N = 3, F_B = 4, F_T = 5 };

// 1-D operator pieces and the tile's edge tables, in device memory (one copy per plan): dynamic
// indexing into a by-value kernel argument would force the whole struct into scratch.
struct EulerConsts {
    double em[kMaxN], ep[kMaxN], cm[kMaxN], cp[kMaxN];
    double D[kMaxN * kMaxN], HF[kMaxN * kMaxN];
    double EF[kMaxN * kMaxN];  // nodal exponential filter (wx_euler3d_set_exp_filter), identity until set
    double rot[4][8];
    int flip[4];
};

template <typename T>
struct EulerParams {
    int H, V, nelem, count, region;
    int advection_only, has_damp;
    int rot_zero;          // plan-time finding: christoffel[:, 0:3] (the rotation symbols) is identically zero
    int axpy;              // 1: out = ca*y + cb*q + cc*R(q) + cd*z (RK stage / FD Jacobian operator), 0: out = R(q)
    double ca, cb, cc, cd;
    const T* y;            // nullable (then ca is ignored)
    const T* z;            // nullable (then cd is ignored)
    // stage pipeline: when itf_out != null the kernel also extrapolates ITS OUTPUT (the next stage's state)
    // to the element faces (phase 1-2 of the NEXT evaluation) into itf_out / nsend_*: no separate K1 pass
    T* itf_out;
    T *nsend_s, *nsend_n, *nsend_w, *nsend_e;
    int efilter;           // stage pipeline only: apply the exponential filter to the stage's output before storing it
    int* nan_flag;         // ... and raise this device flag when the stored values hold a NaN (nullable)
    // JVP mode (T = dual only): the state is formed on load as (q_re, jvp_eps * q_tan) from two REAL arrays
    // and only jvp_scale * tangent(R) is stored, as a real array - no complex temporaries in HBM
    int jvp;
    const double *q_re, *q_tan;
    double* out_tan;
    double jvp_eps, jvp_scale;
    // prepared JVP (wx_euler3d_jvp_prepare): the face VALUES of the linearisation state stay in fv (real,
    // [elem][6][5][n^2]) and in the value halos hv_* for a whole Krylov solve; per product only the face TANGENTS are
    // extrapolated (ft, real, same layout; tangent edge messages through send_* / halo_* as REAL arrays)
    int split;   // 0: off; 1: the JVP kernel reads (fv, ft), written by euler_tan_extrap_kernel (split = 2 there: unused flag)
    double* ft;
    const double* fv;
    const double *hv_s, *hv_n, *hv_w, *hv_e;
    const T* q;
    T* rhs;
    T* itf;  // [elem][6 faces][NQ vars][N2]
    const T *halo_s, *halo_n, *halo_w, *halo_e;
    T *send_s, *send_n, *send_w, *send_e;
    const double *sg, *h, *chr, *idz;
    const double *sgi, *sgj, *sgk, *hi, *hj, *hk;
    const double *dcoef, *duref, *bsn, *bwe;
    const EulerConsts* K;  // device memory
    unsigned long long* stamps;  // WX_K2_DIAG == 1 only, else null
};

struct Elem {
    int ek, ej, ei, e;
    bool valid;
};

// slot (position in this launch's processing order = memory order of the region's elements) -> element of the tile
__device__ __forceinline__ Elem decode_elem(int slot, int count, int region, int H, int V) {
    Elem r;
    r.valid = slot < count;
    if (!r.valid) slot = 0;
    if (region == WX_REGION_ALL) {
        r.ei = slot % H;
        r.ej = (slot / H) % H;
        r.ek = slot / (H * H);
    } else if (region == WX_REGION_INTERIOR) {
        const int w = H - 2;
        r.ei = 1 + slot % w;
        r.ej = 1 + (slot / w) % w;
        r.ek = slot / (w * w);
    } else {
        const int w = H > 2 ? H - 2 : 0;
        const int ring = H * H - w * w;
        r.ek = slot / ring;
        int s = slot % ring;
        if (s < H) {
            r.ej = 0;
            r.ei = s;
        } else if (s < 2 * H) {
            r.ej = H - 1;
            r.ei = s - H;
        } else {
            s -= 2 * H;
            r.ej = 1 + s % w;
            r.ei = (s / w) ? H - 1 : 0;
        }
    }
    r.e = (r.ek * H + r.ej) * H + r.ei;
    return r;
}

// COLUMN form (plans with a column-invariant metric): the V elements of a column follow each other,
// so that the column's metric - one (n x n) slab per field instead of V n of them - is fetched once and found in cache
// by the rest of the column
__device__ __forceinline__ Elem decode_elem_col(int slot, int count, int region, int H, int V) {
    Elem r;
    r.valid = slot < count;
    if (!r.valid) slot = 0;
    const int c = slot / V;   // the column within the region, in the order decode_elem walks one level of it
    r.ek = slot % V;
    if (region == WX_REGION_ALL) {
        r.ei = c % H;
        r.ej = c / H;
    } else if (region == WX_REGION_INTERIOR) {
        const int w = H - 2;
        r.ei = 1 + c % w;
        r.ej = 1 + c / w;
    } else {
        const int w = H > 2 ? H - 2 : 0;
        int s = c;
        if (s < H) {
            r.ej = 0;
            r.ei = s;
        } else if (s < 2 * H) {
            r.ej = H - 1;
            r.ei = s - H;
        } else {
            s -= 2 * H;
            r.ej = 1 + s % w;
            r.ei = (s / w) ? H - 1 : 0;
        }
    }
    r.e = (r.ek * H + r.ej) * H + r.ei;
    return r;
}
// (workgroups go to the eight XCDs round-robin: give each XCD a contiguous eighth of the launch, so that a column stays in
// one L2; the launch has a multiple of eight workgroups, the surplus finds no element)
__device__ __forceinline__ int xcd_slab_block(int b, int nblocks8) { return (b & 7) * nblocks8 + (b >> 3); }

template <typename T>
__device__ __forceinline__ T load_q(const EulerParams<T>& P, size_t i) {
    return P.q[i];
}
// float64 plans: the state may be a shifted one, q + eps * v formed on load (finite-difference Jacobian
// products: no pass that materialises Q + eps v)
template <>
__device__ __forceinline__ double load_q<double>(const EulerParams<double>& P, size_t i) {
    if (P.q_tan != nullptr) return P.q[i] + P.jvp_eps * P.q_tan[i];  // (same expression as load_state<double>)
    return P.q[i];
}
template <>
__device__ __forceinline__ dual load_q<dual>(const EulerParams<dual>& P, size_t i) {
    if (P.jvp) return dual(P.q_re[i], P.jvp_eps * P.q_tan[i]);
    return P.q[i];
}
// the five prognostic values of one point, the mode decided ONCE (a branch per load costs the extrapolation
// kernel 12 %: the compiler no longer issues the five loads back to back)
template <typename T>
__device__ __forceinline__ void load_state(const EulerParams<T>& P, size_t o, size_t fs, T& a0, T& a1, T& a2, T& a3, T& a4) {
    a0 = load_q<T>(P, o); a1 = load_q<T>(P, fs + o); a2 = load_q<T>(P, 2 * fs + o);
    a3 = load_q<T>(P, 3 * fs + o); a4 = load_q<T>(P, 4 * fs + o);
}
template <>
__device__ __forceinline__ void load_state<double>(const EulerParams<double>& P, size_t o, size_t fs, double& a0, double& a1,
                                                   double& a2, double& a3, double& a4) {
    const double* q = P.q;
    a0 = q[o]; a1 = q[fs + o]; a2 = q[2 * fs + o]; a3 = q[3 * fs + o]; a4 = q[4 * fs + o];
    if (P.q_tan != nullptr) {
        const double* v = P.q_tan;
        const double e = P.jvp_eps;
        a0 += e * v[o]; a1 += e * v[fs + o]; a2 += e * v[2 * fs + o]; a3 += e * v[3 * fs + o]; a4 += e * v[4 * fs + o];
    }
}
template <>
__device__ __forceinline__ void load_state<dual>(const EulerParams<dual>& P, size_t o, size_t fs, dual& a0, dual& a1, dual& a2,
                                                 dual& a3, dual& a4) {
    if (P.jvp) {
        const double *r = P.q_re, *t = P.q_tan;
        const double e = P.jvp_eps;
        a0 = dual(r[o], e * t[o]); a1 = dual(r[fs + o], e * t[fs + o]); a2 = dual(r[2 * fs + o], e * t[2 * fs + o]);
        a3 = dual(r[3 * fs + o], e * t[3 * fs + o]); a4 = dual(r[4 * fs + o], e * t[4 * fs + o]);
    } else {
        const dual* q = P.q;
        a0 = q[o]; a1 = q[fs + o]; a2 = q[2 * fs + o]; a3 = q[3 * fs + o]; a4 = q[4 * fs + o];
    }
}

template <typename T>
__device__ __forceinline__ void store_r(const EulerParams<T>& P, size_t i, T r) {
    P.rhs[i] = r;
}
template <>
__device__ __forceinline__ void store_r<dual>(const EulerParams<dual>& P, size_t i, dual r) {
    if (P.jvp) P.out_tan[i] = P.jvp_scale * r.im;
    else P.rhs[i] = r;
}

// Phase 1-2 on nodal values already staged in LDS (log rho, rho u1, rho u2, rho w, log rho*theta):
// one thread per face point extrapolates, exponentiates, writes the interface buffer and, on outward
// tile-edge faces, the rotated / flipped edge message.  Shared by K1 and by K2's stage-pipeline epilogue.
template <int N, typename T, bool COLM = false>
__device__ __forceinline__ void extrap_faces(const EulerParams<T>& P, T (*fld)[Cfg<N>::EPB * Cfg<N>::LE], int slot0,
                                             int count, int region, T* itf_dst, T* ss, T* sn, T* sw, T* se) {
    using C = Cfg<N>;
    constexpr int N2 = C::N2, EPB = C::EPB, BS = C::BS;
    const int tid = threadIdx.x;
    const int H = P.H, V = P.V;
    for (int fi = tid; fi < EPB * 6 * N2; fi += BS) {
        const int le = fi / (6 * N2);
        const int r = fi % (6 * N2);
        int f = r / N2;
        const int fp = r % N2;
        // a face is a whole number of waves when n^2 is a multiple of 64 (n = 8): tell the compiler, so that the
        // face's direction, strides and weights live in scalar registers
        if (N2 % 64 == 0 && BS % 64 == 0) f = __builtin_amdgcn_readfirstlane(f);
        const Elem el = COLM ? decode_elem_col(slot0 + le, count, region, H, V) : decode_elem(slot0 + le, count, region, H, V);
        if (!el.valid) continue;
        if (kNoVertFaces && f >= 4) continue;
        const int d = f >> 1, plus = f & 1;
        const int a = fp / N, b = fp % N;
        // point index of m-th node on the line normal to the face, and its stride
        int base, stride;
        if (d == 0) { base = C::lidx(a, b, 0); stride = 1; }           // (kl=a, jl=b, il=m)
        else if (d == 1) { base = C::lidx(a, 0, b); stride = C::NP; }  // (kl=a, jl=m, il=b)
        else { base = C::lidx(0, a, b); stride = N * C::NP; }          // (kl=m, jl=a, il=b)
        const double* w = plus ? P.K->ep : P.K->em;
        T s[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) s[v] = T(0.0);
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const double wm = w[m];
#pragma unroll
            for (int v = 0; v < 5; ++v) s[v] += wm * fld[v][le * C::LE + base + m * stride];
        }
        s[0] = w_exp(s[0]);
        s[4] = w_exp(s[4]);
        T* dst = itf_dst + ((size_t)el.e * 6 + f) * NQ * N2 + fp;
#pragma unroll
        for (int v = 0; v < 5; ++v) dst[v * N2] = s[v];

        // outward faces of the tile edge: rotate into the neighbour's basis, flip, pack
        int edge = -1, along = 0;
        double X = 0.0;
        if (d == 0 && ((plus && el.ei == H - 1) || (!plus && el.ei == 0))) {
            edge = plus ? E_E : E_W;
            along = el.ej;
            X = P.bwe[el.ej * N + b];
        } else if (d == 1 && ((plus && el.ej == H - 1) || (!plus && el.ej == 0))) {
            edge = plus ? E_N : E_S;
            along = el.ei;
            X = P.bsn[el.ei * N + b];
        }
        T* sendp = edge == E_S ? ss : (edge == E_N ? sn : (edge == E_W ? sw : se));
        if (edge >= 0 && sendp != nullptr) {
            rotate_contra<T>(P.K->rot[edge], X, s[1], s[2]);
            int al = along, bb = b;
            if (P.K->flip[edge]) { al = H - 1 - along; bb = N - 1 - b; }
            const size_t eo = ((size_t)el.ek * H + al) * N2 + a * N + bb;
            const size_t vs = (size_t)V * H * N2;
            T* out = sendp + eo;
#pragma unroll
            for (int v = 0; v < 5; ++v) out[v * vs] = s[v];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: extrapolation to element faces + tile-edge pack
// ------------------------------------------------------------------------------------------------
template <int N, typename T>
__device__ __forceinline__ void euler_extrap_body(const EulerParams<T>& P) {
    using C = Cfg<N>;
    constexpr int N2 = C::N2, N3 = C::N3, EPB = C::EPB;
    __shared__ T fld[5][EPB * C::LE];

    const int tid = threadIdx.x;
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;

    {
        const int le = tid / N3, pt = tid % N3;
        const Elem el = decode_elem(blockIdx.x * EPB + le, P.nelem, WX_REGION_ALL, H, V);
        if (le < EPB && el.valid) {
            const size_t o = (size_t)el.e * N3 + pt;
            const int lp = le * C::LE + C::lidx(pt / N2, (pt / N) % N, pt % N);
            T a0, a1, a2, a3, a4;
            load_state<T>(P, o, fs, a0, a1, a2, a3, a4);
            fld[0][lp] = w_log(a0);
            fld[1][lp] = a1;
            fld[2][lp] = a2;
            fld[3][lp] = a3;
            fld[4][lp] = w_log(a4);
        }
    }
    __syncthreads();

    extrap_faces<N, T>(P, fld, blockIdx.x * EPB, P.nelem, WX_REGION_ALL, P.itf, P.send_s, P.send_n, P.send_w, P.send_e);
}

template <int N, typename T>
__global__ __launch_bounds__(Cfg<N>::BS, kK1Waves) void euler_extrap_kernel(const EulerParams<T> P) {
    euler_extrap_body<N, T>(P);
}

// K1 for the prepared complex-step JVP (wx_euler3d_jvp_tangent_extrap_pack): only the TANGENTS of the face states of
// (q, eps v) are wanted - the values are cached.  The same arithmetic as the dual-number instantiation above, term by
// term, on seven real planes (log rho and log rho*theta; their tangents t / q; the three momentum tangents) instead of
// five 16-byte ones: 32 KB of LDS instead of 46 (n = 8), no value parts carried for the momentum rows - 0.283 -> 0.235 ms per
// E7 panel (5.4 TB/s, the float64 K1's rate), bit-identical tangents (the prepared and unprepared products still agree to
// the last bit: tests/test_n8_kernels_gpu.py).
template <int N>
__global__ __launch_bounds__(Cfg<N>::BS, kK1Waves) void euler_tan_extrap_kernel(const EulerParams<dual> P) {
    using C = Cfg<N>;
    constexpr int N2 = C::N2, N3 = C::N3, EPB = C::EPB, BS = C::BS;
    __shared__ double pl[7][EPB * C::LE];
    const int tid = threadIdx.x;
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;
    {
        const int le = tid / N3, pt = tid % N3;
        const Elem el = decode_elem(blockIdx.x * EPB + le, P.nelem, WX_REGION_ALL, H, V);
        if (le < EPB && el.valid) {
            const size_t o = (size_t)el.e * N3 + pt;
            const int lp = le * C::LE + C::lidx(pt / N2, (pt / N) % N, pt % N);
            const double *r = P.q_re, *t = P.q_tan;
            const double e = P.jvp_eps;
            const double r0 = r[o], r4 = r[4 * fs + o];
            pl[0][lp] = log(r0);
            pl[1][lp] = (e * t[o]) / r0;
            pl[2][lp] = e * t[fs + o];
            pl[3][lp] = e * t[2 * fs + o];
            pl[4][lp] = e * t[3 * fs + o];
            pl[5][lp] = log(r4);
            pl[6][lp] = (e * t[4 * fs + o]) / r4;
        }
    }
    __syncthreads();
    for (int fi = tid; fi < EPB * 6 * N2; fi += BS) {
        const int le = fi / (6 * N2);
        const int r = fi % (6 * N2);
        int f = r / N2;
        const int fp = r % N2;
        if (N2 % 64 == 0 && BS % 64 == 0) f = __builtin_amdgcn_readfirstlane(f);
        const Elem el = decode_elem(blockIdx.x * EPB + le, P.nelem, WX_REGION_ALL, H, V);
        if (!el.valid) continue;
        const int d = f >> 1, plus = f & 1;
        const int a = fp / N, b = fp % N;
        int base, stride;
        if (d == 0) { base = C::lidx(a, b, 0); stride = 1; }
        else if (d == 1) { base = C::lidx(a, 0, b); stride = C::NP; }
        else { base = C::lidx(0, a, b); stride = N * C::NP; }
        const double* w = plus ? P.K->ep : P.K->em;
        double s[7];
#pragma unroll
        for (int v = 0; v < 7; ++v) s[v] = 0.0;
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const double wm = w[m];
#pragma unroll
            for (int v = 0; v < 7; ++v) s[v] += wm * pl[v][le * C::LE + base + m * stride];
        }
        double tn[5];
        tn[0] = exp(s[0]) * s[1];
        tn[1] = s[2]; tn[2] = s[3]; tn[3] = s[4];
        tn[4] = exp(s[5]) * s[6];
        double* dt = P.ft + ((size_t)el.e * 6 + f) * 5 * N2 + fp;
#pragma unroll
        for (int v = 0; v < 5; ++v) dt[v * N2] = tn[v];
        int edge = -1, along = 0;
        double X = 0.0;
        if (d == 0 && ((plus && el.ei == H - 1) || (!plus && el.ei == 0))) {
            edge = plus ? E_E : E_W;
            along = el.ej;
            X = P.bwe[el.ej * N + b];
        } else if (d == 1 && ((plus && el.ej == H - 1) || (!plus && el.ej == 0))) {
            edge = plus ? E_N : E_S;
            along = el.ei;
            X = P.bsn[el.ei * N + b];
        }
        dual* sendp = edge == E_S ? P.send_s : (edge == E_N ? P.send_n : (edge == E_W ? P.send_w : P.send_e));
        if (edge >= 0 && sendp != nullptr) {
            rotate_contra<double>(P.K->rot[edge], X, tn[1], tn[2]);
            int al = along, bb = b;
            if (P.K->flip[edge]) { al = H - 1 - along; bb = N - 1 - b; }
            const size_t eo = ((size_t)el.ek * H + al) * N2 + a * N + bb;
            const size_t vs = (size_t)V * H * N2;
            double* out = reinterpret_cast<double*>(sendp) + eo;
#pragma unroll
            for (int v = 0; v < 5; ++v) out[v * vs] = tn[v];
        }
    }
}

// All tiles of a rank in one launch (blockIdx.y = tile): the static per-tile parameters come from a device table,
// the state is a slice of one stacked tensor.  For small tiles the evaluation is launch-bound.
template <typename T>
struct EulerBatchDyn {
    const T *q, *y, *z;
    T* rhs;
    size_t stride;  // elements of T between consecutive tiles' states
    int region, count, axpy;
    double ca, cb, cc, cd;
    // shifted state q + eps v (float64) or the dual state (q, eps v) formed on load from REAL arrays (dual):
    const double *q_re, *q_tan;  // stride_re doubles apart per tile
    double* out_tan;             // dual JVP output (real)
    size_t stride_re;
    double eps, scale;
    int jvp;
};

template <typename T>
__device__ __forceinline__ void batch_state(EulerParams<T>& P, const EulerBatchDyn<T>& dyn) {
    const size_t off = (size_t)blockIdx.y * dyn.stride, offr = (size_t)blockIdx.y * dyn.stride_re;
    P.q = dyn.q ? dyn.q + off : nullptr;
    P.q_re = dyn.q_re ? dyn.q_re + offr : nullptr;
    P.q_tan = dyn.q_tan ? dyn.q_tan + offr : nullptr;
    P.out_tan = dyn.out_tan ? dyn.out_tan + offr : nullptr;
    P.jvp = dyn.jvp; P.jvp_eps = dyn.eps; P.jvp_scale = dyn.scale;
}

// The parameters of tile blockIdx.y for a batched launch: the table entry goes to LDS (one 8-byte word per thread), one
// thread patches in the per-launch fields, and the body reads what it needs where it needs it.  (Round 2 copied the
// entry into registers - `EulerParams<T> P = table[blockIdx.y]` -: about 120 values live for the whole kernel, which no
// register file holds beside the kernel's own state.  The batched JVP kernels spilled 470-600 bytes per lane and took
// 2.6 x the batched RHS kernel at the reference's benchmark sizes.)
template <typename T, typename Patch>
__device__ __forceinline__ const EulerParams<T>& batch_params(EulerParams<T>& sP, const EulerParams<T>* table, Patch patch) {
    static_assert(sizeof(EulerParams<T>) % 8 == 0, "copied in 8-byte words");
    constexpr int W = sizeof(EulerParams<T>) / 8;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(table + blockIdx.y);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&sP);
    for (int i = threadIdx.x; i < W; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) patch(sP);
    __syncthreads();
    return sP;
}

template <int N, typename T>
__global__ __launch_bounds__(Cfg<N>::BS, kK1Waves) void euler_extrap_batch_kernel(const EulerParams<T>* table,
                                                                                  const EulerBatchDyn<T> dyn) {
    if constexpr (std::is_same<T, double>::value) {   // (float64: the register copy fits - 66 VGPRs, nothing spills - and is faster)
        EulerParams<T> P = table[blockIdx.y];
        batch_state<T>(P, dyn);
        euler_extrap_body<N, T>(P);
    } else {
        __shared__ EulerParams<T> sP;
        euler_extrap_body<N, T>(batch_params<T>(sP, table, [&](EulerParams<T>& P) { batch_state<T>(P, dyn); }));
    }
}

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

// The loads of one face point of one element: own slot of the interface buffer; the neighbour element's slot,
// the received halo on a lateral tile edge, or the own state again (mirrored later) at ground / top; the
// interface metric.  Separate from the arithmetic so that a kernel can issue them early.
template <int N, typename T, bool COLM = false>
__device__ __forceinline__ void face_load(const EulerParams<T>& P, const Elem& el, int f, int fp, FaceIn<T>& in) {
    constexpr int N2 = N * N;
    const int H = P.H, V = P.V;
    const int d = f >> 1, plus = f & 1;
    const size_t vsh = (size_t)V * H * N2;  // var stride in a halo edge message

    const T* own = P.itf + ((size_t)el.e * 6 + f) * NQ * N2 + fp;
    const T* nbr;
    size_t nstride = N2;
    bool mirror = false, from_halo = false;
    const double *sgp, *hp;
    size_t hfs;  // field stride of the h_contra_itf array
    if (d == 0) {
        const int ne = el.ei + (plus ? 1 : -1);
        if (ne >= 0 && ne < H) nbr = P.itf + ((size_t)(el.e + (plus ? 1 : -1)) * 6 + (f ^ 1)) * NQ * N2 + fp;
        else { nbr = (plus ? P.halo_e : P.halo_w) + ((size_t)el.ek * H + el.ej) * N2 + fp; nstride = vsh; from_halo = true; }
        // (column form: the interface metric of a lateral face does not depend on the level - one row of n values per
        // face side instead of V n of them; that of a horizontal face neither on the level nor on the side)
        const size_t o = COLM ? (((size_t)el.ej * (H + 2) + el.ei + 1) * 2 + plus) * N + fp % N
                              : (((size_t)el.ek * H + el.ej) * (H + 2) + el.ei + 1) * 2 * N2 + plus * N2 + fp;
        hfs = COLM ? (size_t)H * (H + 2) * 2 * N : (size_t)V * H * (H + 2) * 2 * N2;
        sgp = P.sgi + o;
        hp = P.hi + 0 * 3 * hfs + o;
    } else if (d == 1) {
        const int ne = el.ej + (plus ? 1 : -1);
        if (ne >= 0 && ne < H) nbr = P.itf + ((size_t)(el.e + (plus ? H : -H)) * 6 + (f ^ 1)) * NQ * N2 + fp;
        else { nbr = (plus ? P.halo_n : P.halo_s) + ((size_t)el.ek * H + el.ei) * N2 + fp; nstride = vsh; from_halo = true; }
        const size_t o = COLM ? ((((size_t)el.ej + 1) * H + el.ei) * 2 + plus) * N + fp % N
                              : (((size_t)el.ek * (H + 2) + el.ej + 1) * H + el.ei) * 2 * N2 + plus * N2 + fp;
        hfs = COLM ? (size_t)(H + 2) * H * 2 * N : (size_t)V * (H + 2) * H * 2 * N2;
        sgp = P.sgj + o;
        hp = P.hj + 1 * 3 * hfs + o;
    } else {
        const int ne = el.ek + (plus ? 1 : -1);
        if (ne >= 0 && ne < V) nbr = P.itf + ((size_t)(el.e + (plus ? H * H : -H * H)) * 6 + (f ^ 1)) * NQ * N2 + fp;
        else { nbr = own; mirror = true; }
        const size_t o = COLM ? ((size_t)el.ej * H + el.ei) * N2 + fp
                              : ((((size_t)el.ek + 1) * H + el.ej) * H + el.ei) * 2 * N2 + plus * N2 + fp;
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
template <int N, typename T, bool OWN_FORM = false, bool COLM = false>
__device__ __forceinline__ void face_problem(const EulerParams<T>& P, const Elem& el, int f, int fp, T* out) {
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
template <typename T, bool CACHED = false>
__device__ __forceinline__ void k2_point_loads(const EulerParams<T>& P, bool active, size_t o, size_t fs, PointIn<T>& S,
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
template <typename T>
__device__ __forceinline__ void k2_point_loads(const EulerParams<T>& P, bool active, size_t o, size_t fs, PointIn<T>& S) {
    k2_point_loads<T, false>(P, active, o, fs, S, o, fs);
}

// forcing of the three momentum rows, all but the gravity filter (pde_euler_cubesphere.py:12-25, 203-290), from the 27
// (18 on a non-rotating planet) Christoffel fields, all loads in flight together; gcoef = inv_dzdeta * g
template <typename T, bool CACHED = false>
__device__ __forceinline__ void k2_forcing(const EulerParams<T>& P, bool active, size_t o, size_t fs, const PointIn<T>& S, T u1,
                                           T u2, T u3, T p, T& fc0, T& fc1, T& fc2, double& gcoef, size_t om, size_t fsm) {
    double cg[27], idzv = 0.0;
    if (active && P.rot_zero) {   // non-rotating planet: the 9 rotation symbols are identically zero
#pragma unroll
        for (int i = 0; i < 27; ++i) cg[i] = (i % 9) < 3 ? 0.0 : ldm_if<CACHED>(P.chr + (size_t)i * fsm + om);
        idzv = ldm_if<CACHED>(P.idz + om);
    } else if (active) {
#pragma unroll
        for (int i = 0; i < 27; ++i) cg[i] = ldm_if<CACHED>(P.chr + (size_t)i * fsm + om);
        idzv = ldm_if<CACHED>(P.idz + om);
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
template <typename T>
__device__ __forceinline__ void k2_forcing(const EulerParams<T>& P, bool active, size_t o, size_t fs, const PointIn<T>& S, T u1,
                                           T u2, T u3, T p, T& fc0, T& fc1, T& fc2, double& gcoef) {
    k2_forcing<T, false>(P, active, o, fs, S, u1, u2, u3, p, fc0, fc1, fc2, gcoef, o, fs);
}

template <int N, typename T, bool PIPE, bool COLM = false>
__device__ __forceinline__ void euler_rhs_body(const EulerParams<T>& P) {
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
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;
    if (PIPE && !MF && P.efilter)
        for (int i = tid; i < N * N; i += BS) sEF[i] = P.K->EF[i];
#if WX_K2_DIAG == 1
#define WX_STAMP(i)                                                                           \
    do {                                                                                      \
        __syncthreads();                                                                      \
        if (tid == 0 && P.stamps) P.stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64();    \
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
    const Elem el = COLM ? decode_elem_col(bx * EPB + le, P.count, P.region, H, V) : decode_elem(bx * EPB + le, P.count, P.region, H, V);
    const bool active = (le < EPB) && el.valid;
    const int kl = pt / N2, jl = (pt / N) % N, il = pt % N;
    const int lb = (le < EPB ? le : 0) * C::LE;  // LDS base of this thread's element
    const int lpt = lb + C::lidx(kl, jl, il);    // this thread's node in the LDS image
    const int lptm = mf_idx(kl, jl, il);         // ... and in the image of the matrix-core passes
    MfOps4 mops4{0.0, 0.0, 0.0, 0.0, 0.0};
    if (MF) mops4 = mf4_load_ops(P.K->D, P.K->cm, P.K->cp, P.K->HF, tid & 63);
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
    k2_point_loads<T, COLM>(P, active, o, fs, S, om, fsm);   // in flight while the face stage computes
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
        const Elem fel = COLM ? decode_elem_col(bx * EPB + fle, P.count, P.region, H, V) : decode_elem(bx * EPB + fle, P.count, P.region, H, V);
        if (!fel.valid) continue;
        T out[NC];
        face_problem<N, T, false, COLM>(P, fel, f, fp, out);
#pragma unroll
        for (int c = 0; c < NC; ++c) WX_FR(fle, f, c, fp) = out[c];
    }
    WX_STAMP(1);

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
    k2_forcing<T, COLM>(P, active, o, fs, S, u1, u2, u3, p, fc0, fc1, fc2, gcoef, om, fsm);
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
        r0 = P.cb * q0 + P.cc * r0; r1 = P.cb * q1 + P.cc * r1; r2 = P.cb * q2 + P.cc * r2;
        r3 = P.cb * q3 + P.cc * r3; r4 = P.cb * q4 + P.cc * r4;
        if (P.y != nullptr) {
            r0 += P.ca * P.y[o]; r1 += P.ca * P.y[fs + o]; r2 += P.ca * P.y[2 * fs + o];
            r3 += P.ca * P.y[3 * fs + o]; r4 += P.ca * P.y[4 * fs + o];
        }
        if (P.z != nullptr) {
            r0 += P.cd * P.z[o]; r1 += P.cd * P.z[fs + o]; r2 += P.cd * P.z[2 * fs + o];
            r3 += P.cd * P.z[3 * fs + o]; r4 += P.cd * P.z[4 * fs + o];
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
        extrap_faces<N, T, COLM>(P, fld, bx * EPB, P.count, P.region, P.itf_out, P.nsend_s, P.nsend_n, P.nsend_w, P.nsend_e);
    }
#undef WX_STAMP
#undef WX_FR
}

template <int N, typename T>
constexpr int k2_waves() { return is_complex<T>::value ? 2 : kK2Waves; }

template <int N, typename T, bool PIPE>
__global__ __launch_bounds__(Cfg<N>::BS, (k2_waves<N, T>())) void euler_rhs_kernel(const EulerParams<T> P) {
    euler_rhs_body<N, T, PIPE>(P);
}

template <int N, typename T>
__global__ __launch_bounds__(Cfg<N>::BS, (k2_waves<N, T>())) void euler_rhs_batch_kernel(const EulerParams<T>* table,
                                                                                         const EulerBatchDyn<T> dyn) {
    auto patch = [&](EulerParams<T>& P) {
        const size_t off = (size_t)blockIdx.y * dyn.stride;
        batch_state<T>(P, dyn);
        P.rhs = dyn.rhs ? dyn.rhs + off : nullptr;
        P.y = dyn.y ? dyn.y + off : nullptr;
        P.z = dyn.z ? dyn.z + off : nullptr;
        P.region = dyn.region; P.count = dyn.count;
        P.axpy = dyn.axpy; P.ca = dyn.ca; P.cb = dyn.cb; P.cc = dyn.cc; P.cd = dyn.cd;
    };
    if constexpr (std::is_same<T, double>::value) {   // (float64: the register copy fits and is faster; 16-byte dtypes spilled)
        EulerParams<T> P = table[blockIdx.y];
        patch(P);
        euler_rhs_body<N, T, false>(P);
    } else {
        __shared__ EulerParams<T> sP;
        euler_rhs_body<N, T, false>(batch_params<T>(sP, table, patch));
    }
}

// ------------------------------------------------------------------------------------------------
// K2-JVP: the fused phases 3-8 specialised for the complex-step Jacobian-vector product (wx_euler3d_jvp).
// Only the TANGENT of R is wanted, and the derivative contractions are linear, so of the eight fields the generic
// dual-number kernel stages through LDS as (value, tangent) pairs, six need their tangent only (the four flux
// rows, the advective rho*w flux, sqrtG*rho), one is a pure metric quantity with no tangent (B = sqrtG h^{d3})
// and one needs both (log p, multiplied by p B afterwards).  The same holds for the face quantities.  LDS per
// element 118 KB -> 70 KB and half the registers in the accumulators: TWO workgroups per CU instead of one,
// and 40 % fewer LDS bytes and contraction flops.  Arithmetic is the generic kernel's, term by term; the Riemann
// problems use the own / neighbour form (rusanov_own) and the Christoffel rows are read one at a time (rolled loop):
// with everything in flight the compiler wanted 184 VGPRs.  WXHIP_JVP_LEAN=0 (environment, read once) sends
// wx_euler3d_jvp through the generic kernel instead.
// ------------------------------------------------------------------------------------------------
// tangent of the forcing of the three momentum rows, times sqrtG (.f1, .f2, .fw), and gcoef = inv_dzdeta * g
struct JvpForcing { double f1, f2, fw, gcoef; };
template <bool CACHED>
__device__ __forceinline__ JvpForcing jvp_forcing(const EulerParams<dual>& P, size_t o, size_t fs, double sg, double h00,
                                                  double h01, double h02, double h11, double h12, double h22, dual q0,
                                                  dual u1, dual u2, dual u3, dual p, size_t om, size_t fsm) {
    JvpForcing r{0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
        const double* c = P.chr + (size_t)(i * 9) * fsm + om;
        double c01 = 0.0, c02 = 0.0, c03 = 0.0;
        if (!P.rot_zero) { c01 = ldm_if<CACHED>(c); c02 = ldm_if<CACHED>(c + fsm); c03 = ldm_if<CACHED>(c + 2 * fsm); }
        const double c11 = ldm_if<CACHED>(c + 3 * fsm), c12 = ldm_if<CACHED>(c + 4 * fsm), c13 = ldm_if<CACHED>(c + 5 * fsm),
                     c22 = ldm_if<CACHED>(c + 6 * fsm), c23 = ldm_if<CACHED>(c + 7 * fsm), c33 = ldm_if<CACHED>(c + 8 * fsm);
        dual f = 2.0 * q0 * (c01 * u1 + c02 * u2 + c03 * u3) + c11 * (q0 * u1 * u1 + h00 * p) +
                 2.0 * c12 * (q0 * u1 * u2 + h01 * p) + 2.0 * c13 * (q0 * u1 * u3 + h02 * p) +
                 c22 * (q0 * u2 * u2 + h11 * p) + 2.0 * c23 * (q0 * u2 * u3 + h12 * p) +
                 c33 * (q0 * u3 * u3 + h22 * p);
        if (P.has_damp) f += (P.dcoef[o] * q0) * ((i == 0 ? u1 : (i == 1 ? u2 : u3)) - P.duref[(size_t)i * fs + o]);
        if (i == 0) r.f1 = sg * f.im;
        else if (i == 1) r.f2 = sg * f.im;
        else r.fw = sg * f.im;
    }
    r.gcoef = ldm_if<CACHED>(P.idz + om) * kGravity;
    return r;
}

template <int N, bool COLM = false>
__device__ __forceinline__ void euler_jvp_body(const EulerParams<dual>& P) {
    using C = Cfg<N>;
    using T = dual;
    const int bx = COLM ? xcd_slab_block(blockIdx.x, gridDim.x >> 3) : (int)blockIdx.x;
    constexpr int N2 = C::N2, N3 = C::N3, EPB = C::EPB, BS = C::BS;
    __shared__ double ft[6][EPB * C::LE];     // tangents: F rows rho, rho u1, rho u2, rho theta; A; sqrtG*rho
    __shared__ double fx[3][EPB * C::LE];     // B = sqrtG h^{d3} (metric only); log p, value and tangent planes
    __shared__ double frt[EPB][6][5][N2];     // tangents of the face quantities 0..4 of rusanov_face
    __shared__ T frf[EPB][6][2][N2];          // B*_own, log p_own
    __shared__ double sD[N * N], sHF[N * N], sCm[N], sCp[N];

    const int tid = threadIdx.x;
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;
    for (int i = tid; i < N * N; i += BS) {
        sD[i] = P.K->D[i];
        sHF[i] = P.K->HF[i];
    }
    if (tid < N) {
        sCm[tid] = P.K->cm[tid];
        sCp[tid] = P.K->cp[tid];
    }

    const int le = tid / N3, pt = tid % N3;
    const Elem el = COLM ? decode_elem_col(bx * EPB + le, P.count, P.region, H, V) : decode_elem(bx * EPB + le, P.count, P.region, H, V);
    const bool active = (le < EPB) && el.valid;
    const int kl = pt / N2, jl = (pt / N) % N, il = pt % N;
    const int lb = (le < EPB ? le : 0) * C::LE;
    const int lpt = lb + C::lidx(kl, jl, il);
    const size_t o = (size_t)el.e * N3 + pt;

    // ---- face stage
    for (int fi = tid; fi < EPB * 6 * N2; fi += BS) {
        const int fle = fi / (6 * N2);
        const int r = fi % (6 * N2);
        const int f = r / N2, fp = r % N2;
        const Elem fel = COLM ? decode_elem_col(bx * EPB + fle, P.count, P.region, H, V) : decode_elem(bx * EPB + fle, P.count, P.region, H, V);
        if (!fel.valid) continue;
        T out[7];
        face_problem<N, T, true, COLM>(P, fel, f, fp, out);
#pragma unroll
        for (int c = 0; c < 5; ++c) frt[fle][f][c][fp] = out[c].im;
        frf[fle][f][0][fp] = out[5];
        frf[fle][f][1][fp] = out[6];
    }

    PointIn<T> S;
    const size_t om = COLM ? ((size_t)el.ej * H + el.ei) * N2 + pt % N2 : o;
    const size_t fsm = COLM ? (size_t)H * H * N2 : fs;
    k2_point_loads<T, COLM>(P, active, o, fs, S, om, fsm);
    const T q0 = S.q0, q1 = S.q1, q2 = S.q2, q3 = S.q3, q4 = S.q4;
    const double sg = S.sg;
    // ---- pointwise quantities
    const T rinv = 1.0 / q0;
    const T u1 = q1 * rinv, u2 = q2 * rinv, u3 = q3 * rinv;
    const T glog = kGamma * w_log(kRdOverP0 * q4);
    const T p = kP0 * w_exp(glog);
    if (le < EPB) {
        const T lp = kLogP0 + glog;
        fx[1][lpt] = lp.re;
        fx[2][lpt] = lp.im;
        ft[5][lpt] = sg * q0.im;
    }

    // ---- forcing (tangent)
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc4 = 0.0, accw = 0.0, hf = 0.0, gcoef = 0.0;
    if (active) {
        const JvpForcing F = jvp_forcing<COLM>(P, o, fs, sg, S.h00, S.h01, S.h02, S.h11, S.h12, S.h22, q0, u1, u2, u3, p, om, fsm);
        acc1 = F.f1; acc2 = F.f2; accw = F.fw; gcoef = F.gcoef;
    }

#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const T ud = w_sel(d == 0, u1, w_sel(d == 1, u2, u3));
        const double hd0 = d == 0 ? S.h00 : (d == 1 ? S.h01 : S.h02);
        const double hd1 = d == 0 ? S.h01 : (d == 1 ? S.h11 : S.h12);
        const double hd2 = d == 0 ? S.h02 : (d == 1 ? S.h12 : S.h22);
        const T sgu = sg * ud;
        const double Bd = sg * hd2;
        __syncthreads();  // face stage / previous direction's reads are done
        if (le < EPB) {
            ft[0][lpt] = (sgu * q0).im;
            ft[1][lpt] = (sgu * q1 + (sg * hd0) * p).im;
            ft[2][lpt] = (sgu * q2 + (sg * hd1) * p).im;
            ft[3][lpt] = (sgu * q4).im;
            ft[4][lpt] = (sgu * q3).im;
            fx[0][lpt] = Bd;
        }
        __syncthreads();

        int base, stride, idx, fp;
        if (d == 0) { base = lb + C::lidx(kl, jl, 0); stride = 1; idx = il; fp = kl * N + jl; }
        else if (d == 1) { base = lb + C::lidx(kl, 0, il); stride = C::NP; idx = jl; fp = kl * N + il; }
        else { base = lb + C::lidx(0, jl, il); stride = N * C::NP; idx = kl; fp = jl * N + il; }
        double dm[N];
#pragma unroll
        for (int m = 0; m < N; ++m) dm[m] = sD[idx * N + m];
        const double cm = sCm[idx], cp = sCp[idx];
        const int lf = le < EPB ? le : 0;
        constexpr int FB = kFieldBatch;
#pragma unroll 1
        for (int c0 = 0; c0 < 5; c0 += FB) {
#pragma unroll
            for (int cc = 0; cc < FB; ++cc) {
                const int c = c0 + cc;
                if (c < 5) {
                    double a = cm * frt[lf][2 * d][c][fp] + cp * frt[lf][2 * d + 1][c][fp];
#pragma unroll
                    for (int m = 0; m < N; ++m) a += dm[m] * ft[c][base + m * stride];
                    if (c == 0) acc0 += a;
                    else if (c == 1) acc1 += a;
                    else if (c == 2) acc2 += a;
                    else if (c == 3) acc4 += a;
                    else accw += a;
                }
            }
        }
        // W^d = [A@D + A*@C] + p [B@D + B*@C] + p B [log p@D + log p^@C]  (rhs_dfr.py:113-136): tangent of the
        // two products; B@D is a metric-only number.  One plane at a time: bounds the LDS reads in flight
        {
            double xs0 = 0.0, xs1 = 0.0, xs2 = 0.0;
#pragma unroll 1
            for (int w = 0; w < 3; ++w) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < N; ++m) acc += dm[m] * fx[w][base + m * stride];
                if (w == 0) xs0 = acc;
                else if (w == 1) xs1 = acc;
                else xs2 = acc;
            }
            const T a5 = cm * frf[lf][2 * d][0][fp] + cp * frf[lf][2 * d + 1][0][fp] + xs0;
            const T a6 = cm * frf[lf][2 * d][1][fp] + cp * frf[lf][2 * d + 1][1][fp] + T(xs1, xs2);
            accw += (a5 * p).im + (a6 * (p * Bd)).im;
        }
        if (d == 2) {
#pragma unroll
            for (int m = 0; m < N; ++m) hf += sHF[idx * N + m] * ft[5][base + m * stride];
        }
    }

    if (active) {
        const double s = P.advection_only ? 0.0 : -P.jvp_scale / sg;
        accw += gcoef * hf;  // gravity: inv_dzdeta * g * 1/sqrtG * HF_k(sqrtG rho)
        P.out_tan[o] = s * acc0;
        P.out_tan[fs + o] = s * acc1;
        P.out_tan[2 * fs + o] = s * acc2;
        P.out_tan[3 * fs + o] = s * accw;
        P.out_tan[4 * fs + o] = s * acc4;
    }
}

// The same kernel with the contractions on the matrix cores (n = 8; mf4_dir_pass, the layout and the in-place scheme of
// the fused RHS kernel).  Nine real planes per direction: the five flux tangents, B (metric), log p value and tangent -
// eight take D | cm | cp with their face pairs as the third k-step - and the tangent of sqrtG rho for the vertical
// high-filter; the tangent of B* has no nodal part and keeps its two-term correction on the vector pipe.
constexpr int kJvFS = 9 * 64 + 16;   // doubles per face of the JVP kernel's face image (9 quantities)

template <bool COLM = false>
__device__ __forceinline__ void euler_jvp_body_mf(const EulerParams<dual>& P) {
    using T = dual;
    const int bx = COLM ? xcd_slab_block(blockIdx.x, gridDim.x >> 3) : (int)blockIdx.x;
    constexpr int N = 8, N2 = 64, N3 = 512;
    __shared__ double pl[9 * kMfLE];   // 0-4 flux tangents (rho, rho u1, rho u2, rho theta, A); 5 B; 6, 7 log p (value, tangent); 8 (sqrtG rho)'
    __shared__ double fq[6 * kJvFS];   // per face: 0-4 tangents of F*; 5 B*.re; 6, 7 log p_own (value, tangent); 8 B*.im
    __shared__ double sCm[N], sCp[N];
    const int tid = threadIdx.x;
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;
    if (tid < N) {
        sCm[tid] = P.K->cm[tid];
        sCp[tid] = P.K->cp[tid];
    }
    const MfOps4 mops = mf4_load_ops(P.K->D, P.K->cm, P.K->cp, P.K->HF, tid & 63);
    const Elem el = COLM ? decode_elem_col(bx, P.count, P.region, H, V) : decode_elem(bx, P.count, P.region, H, V);
    const bool active = el.valid;
    const int kl = tid / N2, jl = (tid / N) % N, il = tid % N;
    const int lptm = mf_idx(kl, jl, il);
    const size_t o = (size_t)el.e * N3 + tid;

    // ---- face stage
    for (int fi = tid; fi < 6 * N2; fi += 512) {
        const int f = fi / N2, fp = fi % N2;
        if (!el.valid) continue;
        T out[7];
        face_problem<N, T, true, COLM>(P, el, f, fp, out);
        double* q = fq + f * kJvFS + fp;
#pragma unroll
        for (int c = 0; c < 5; ++c) q[c * N2] = out[c].im;
        q[5 * N2] = out[5].re; q[8 * N2] = out[5].im;
        q[6 * N2] = out[6].re; q[7 * N2] = out[6].im;
    }

    PointIn<T> S;
    // (column form: a 32-bit offset into the slabs, which are small - one register instead of two beside `o`)
    const unsigned om32 = (unsigned)((el.ej * H + el.ei) * N2 + tid % N2);
    const size_t om = COLM ? (size_t)om32 : o;
    const size_t fsm = COLM ? (size_t)H * H * N2 : fs;
    k2_point_loads<T, COLM>(P, active, o, fs, S, om, fsm);
    const T q0 = S.q0, q1 = S.q1, q2 = S.q2, q3 = S.q3, q4 = S.q4;
    const double sg = S.sg;
    const T rinv = 1.0 / q0;
    const T u1 = q1 * rinv, u2 = q2 * rinv, u3 = q3 * rinv;
    const T glog = kGamma * w_log(kRdOverP0 * q4);
    const T p = kP0 * w_exp(glog);
    const T lp = kLogP0 + glog;

    // ---- forcing (tangent)
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc4 = 0.0, accw = 0.0, hf = 0.0, gcoef = 0.0;
    if (active) {
        const JvpForcing F = jvp_forcing<COLM>(P, o, fs, sg, S.h00, S.h01, S.h02, S.h11, S.h12, S.h22, q0, u1, u2, u3, p, om, fsm);
        acc1 = F.f1; acc2 = F.f2; accw = F.fw; gcoef = F.gcoef;
    }

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (kSkelDirs) {   // diagnostic builds: the staged data consumed, no passes
        pl[lptm] = lp.im;
        __syncthreads();
        acc0 += fq[tid & 63] + pl[lptm] + u1.im + u2.im + u3.im;
    }
#pragma unroll
    for (int d = 0; d < (kSkelDirs ? 0 : 3); ++d) {
        const T ud = w_sel(d == 0, u1, w_sel(d == 1, u2, u3));
        const double hd0 = d == 0 ? S.h00 : (d == 1 ? S.h01 : S.h02);
        const double hd1 = d == 0 ? S.h01 : (d == 1 ? S.h11 : S.h12);
        const double hd2 = d == 0 ? S.h02 : (d == 1 ? S.h12 : S.h22);
        const T sgu = sg * ud;
        const double Bd = sg * hd2;
        // each thread stages its own node of the nine planes, the eight waves contract all lines in place, each thread
        // picks its own node up again (no barrier between a thread's read and its next write)
        pl[0 * kMfLE + lptm] = (sgu * q0).im;
        pl[1 * kMfLE + lptm] = (sgu * q1 + (sg * hd0) * p).im;
        pl[2 * kMfLE + lptm] = (sgu * q2 + (sg * hd1) * p).im;
        pl[3 * kMfLE + lptm] = (sgu * q4).im;
        pl[4 * kMfLE + lptm] = (sgu * q3).im;
        pl[5 * kMfLE + lptm] = Bd;
        pl[6 * kMfLE + lptm] = lp.re;
        pl[7 * kMfLE + lptm] = lp.im;
        if (d == 2) pl[8 * kMfLE + lptm] = sg * q0.im;
        __syncthreads();
        if (d == 0) mf4_dir_pass<0, true, 8, true, kJvFS, kJvpMfFieldBatch>(pl, fq, mops, wave, tid & 63);
        else if (d == 1) mf4_dir_pass<1, true, 8, true, kJvFS, kJvpMfFieldBatch>(pl, fq, mops, wave, tid & 63);
        else mf4_dir_pass<2, true, 8, true, kJvFS, kJvpMfFieldBatch>(pl, fq, mops, wave, tid & 63);
        __syncthreads();
        acc0 += pl[0 * kMfLE + lptm];
        acc1 += pl[1 * kMfLE + lptm];
        acc2 += pl[2 * kMfLE + lptm];
        acc4 += pl[3 * kMfLE + lptm];
        accw += pl[4 * kMfLE + lptm];
        // W^d = [A@D + A*@C] + p [B@D + B*@C] + p B [log p@D + log p^@C]  (rhs_dfr.py:113-136): tangent of the two products
        const int fp = d == 0 ? kl * N + jl : (d == 1 ? kl * N + il : jl * N + il);
        const int ix = d == 0 ? il : (d == 1 ? jl : kl);
        const T a5(pl[5 * kMfLE + lptm], sCm[ix] * fq[(2 * d) * kJvFS + 8 * N2 + fp] + sCp[ix] * fq[(2 * d + 1) * kJvFS + 8 * N2 + fp]);
        const T a6(pl[6 * kMfLE + lptm], pl[7 * kMfLE + lptm]);
        accw += (a5 * p).im + (a6 * (p * Bd)).im;
        if (d == 2) hf = pl[8 * kMfLE + lptm];
    }

    if (active) {
        const double sc = P.advection_only ? 0.0 : -P.jvp_scale / sg;
        accw += gcoef * hf;  // gravity: inv_dzdeta * g * 1/sqrtG * HF_k(sqrtG rho)
        P.out_tan[o] = sc * acc0;
        P.out_tan[fs + o] = sc * acc1;
        P.out_tan[2 * fs + o] = sc * acc2;
        P.out_tan[3 * fs + o] = sc * accw;
        P.out_tan[4 * fs + o] = sc * acc4;
    }
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::BS, kJvpWaves) void euler_jvp_kernel(const EulerParams<dual> P) {
    if constexpr (N == 8 && WX_MFMA) euler_jvp_body_mf<false>(P);
    else euler_jvp_body<N, false>(P);
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::BS, kJvpWaves) void euler_jvp_batch_kernel(const EulerParams<dual>* table,
                                                                                const EulerBatchDyn<dual> dyn) {
    __shared__ EulerParams<dual> sP;
    const EulerParams<dual>& P = batch_params<dual>(sP, table, [&](EulerParams<dual>& Q) {
        batch_state<dual>(Q, dyn);
        Q.region = dyn.region; Q.count = dyn.count;
    });
    if constexpr (N == 8 && WX_MFMA) euler_jvp_body_mf<false>(P);
    else euler_jvp_body<N, false>(P);
}

// plan-time scan of a static field: raises *flag when any value differs from (+/-) zero
__global__ __launch_bounds__(256) void any_nonzero_kernel(const double* __restrict__ x, size_t count, int* flag) {
    bool any = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        any = any || (x[i] != 0.0);
    if (any) *flag = 1;   // many writers, one value: benign
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int N, typename T>
static wx_status launch_extrap(const EulerParams<T>& P, hipStream_t st) {
    using C = Cfg<N>;
    const int grid = (P.nelem + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_extrap_kernel<N, T>), dim3(grid), dim3(C::BS), 0, st, P);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N, bool PIPE>
__global__ __launch_bounds__(Cfg<N>::BS, kK2Waves) void euler_rhs_column_kernel(const EulerParams<double> P) {
    euler_rhs_body<N, double, PIPE, true>(P);
}

template <int N>
static wx_status launch_rhs_column(const EulerParams<double>& P, hipStream_t st) {
    using C = Cfg<N>;
    if (P.count == 0) return WX_OK;
    const int grid = (P.count + C::EPB - 1) / C::EPB;
    if (P.itf_out != nullptr) hipLaunchKernelGGL((euler_rhs_column_kernel<N, true>), dim3(8 * ((grid + 7) / 8)), dim3(C::BS), 0, st, P);
    else hipLaunchKernelGGL((euler_rhs_column_kernel<N, false>), dim3(8 * ((grid + 7) / 8)), dim3(C::BS), 0, st, P);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

static wx_status dispatch_rhs_column(int n, const EulerParams<double>& P, hipStream_t st) {
    switch (n) {
        case 2: return launch_rhs_column<2>(P, st);
        case 3: return launch_rhs_column<3>(P, st);
        case 4: return launch_rhs_column<4>(P, st);
        case 5: return launch_rhs_column<5>(P, st);
        case 6: return launch_rhs_column<6>(P, st);
        case 7: return launch_rhs_column<7>(P, st);
        case 8: return launch_rhs_column<8>(P, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", n);
}

template <int N, typename T>
static wx_status launch_rhs(const EulerParams<T>& P, hipStream_t st) {
    using C = Cfg<N>;
    if (P.count == 0) return WX_OK;
    const int grid = (P.count + C::EPB - 1) / C::EPB;
    if (P.itf_out != nullptr) hipLaunchKernelGGL((euler_rhs_kernel<N, T, true>), dim3(grid), dim3(C::BS), 0, st, P);
    else hipLaunchKernelGGL((euler_rhs_kernel<N, T, false>), dim3(grid), dim3(C::BS), 0, st, P);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N, typename T>
static wx_status launch_extrap_batch(const EulerParams<T>* table, const EulerBatchDyn<T>& dyn, int nelem, int ntiles,
                                     hipStream_t st) {
    using C = Cfg<N>;
    const int grid = (nelem + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_extrap_batch_kernel<N, T>), dim3(grid, ntiles), dim3(C::BS), 0, st, table, dyn);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N, typename T>
static wx_status launch_rhs_batch(const EulerParams<T>* table, const EulerBatchDyn<T>& dyn, int ntiles, hipStream_t st) {
    using C = Cfg<N>;
    if (dyn.count == 0) return WX_OK;
    const int grid = (dyn.count + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_rhs_batch_kernel<N, T>), dim3(grid, ntiles), dim3(C::BS), 0, st, table, dyn);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N>
static wx_status launch_jvp_batch(const EulerParams<dual>* table, const EulerBatchDyn<dual>& dyn, int ntiles, hipStream_t st) {
    using C = Cfg<N>;
    if (dyn.count == 0) return WX_OK;
    const int grid = (dyn.count + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_jvp_batch_kernel<N>), dim3(grid, ntiles), dim3(C::BS), 0, st, table, dyn);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N>
static wx_status launch_jvp(const EulerParams<dual>& P, hipStream_t st) {
    using C = Cfg<N>;
    if (P.count == 0) return WX_OK;
    const int grid = (P.count + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_jvp_kernel<N>), dim3(grid), dim3(C::BS), 0, st, P);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::BS, kJvpWaves) void euler_jvp_column_kernel(const EulerParams<dual> P) {
    if constexpr (N == 8 && WX_MFMA) euler_jvp_body_mf<true>(P);
    else euler_jvp_body<N, true>(P);
}

template <int N>
static wx_status launch_jvp_column(const EulerParams<dual>& P, hipStream_t st) {
    using C = Cfg<N>;
    if (P.count == 0) return WX_OK;
    const int grid = (P.count + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_jvp_column_kernel<N>), dim3(8 * ((grid + 7) / 8)), dim3(C::BS), 0, st, P);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

static wx_status dispatch_jvp_column(int n, const EulerParams<dual>& P, hipStream_t st) {
    switch (n) {
        case 2: return launch_jvp_column<2>(P, st);
        case 3: return launch_jvp_column<3>(P, st);
        case 4: return launch_jvp_column<4>(P, st);
        case 5: return launch_jvp_column<5>(P, st);
        case 6: return launch_jvp_column<6>(P, st);
        case 7: return launch_jvp_column<7>(P, st);
        case 8: return launch_jvp_column<8>(P, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", n);
}

}  // namespace wx

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
    P.q = nullptr; P.rhs = nullptr; P.itf = static_cast<T*>(pl->itf);
    P.axpy = 0; P.ca = P.cb = P.cd = 0.0; P.cc = 1.0; P.y = nullptr; P.z = nullptr;
    P.itf_out = nullptr; P.nsend_s = P.nsend_n = P.nsend_w = P.nsend_e = nullptr;
    P.efilter = 0; P.nan_flag = nullptr;
    P.jvp = 0; P.q_re = P.q_tan = nullptr; P.out_tan = nullptr; P.jvp_eps = 0.0; P.jvp_scale = 1.0;
    P.split = 0; P.ft = nullptr; P.fv = nullptr; P.hv_s = P.hv_n = P.hv_w = P.hv_e = nullptr;
    P.halo_s = P.halo_n = P.halo_w = P.halo_e = nullptr;
    P.send_s = P.send_n = P.send_w = P.send_e = nullptr;
    P.K = pl->consts;
    P.stamps = pl->stamps;
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

template <typename T>
wx_status run_extrap(wx_euler3d_plan* pl, const void* q, void* const send[4], hipStream_t st, int slot = 0) {
    EulerParams<T> P = make_params<T>(pl);
    if (slot == 1) P.itf = static_cast<T*>(pl->itf2);
    P.q = static_cast<const T*>(q);
    if (send) {
        P.send_s = static_cast<T*>(send[0]); P.send_n = static_cast<T*>(send[1]);
        P.send_w = static_cast<T*>(send[2]); P.send_e = static_cast<T*>(send[3]);
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
        for (int i = 0; i < 8; ++i) hc.rot[ed][i] = on_panel_edge[ed] ? kRot[panel][ed][i] : identity[i];
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
    if (!pl || !q || !v) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_tangent_extrap_pack: null argument");
    if (pl->dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_*: the plan must be WX_DUAL128");
    if (!pl->face_val) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_tangent_extrap_pack: call wx_euler3d_jvp_prepare first");
    EulerParams<dual> P = make_params<dual>(pl);
    P.jvp = 1; P.q_re = q; P.q_tan = v; P.jvp_eps = eps;
    P.split = 2; P.ft = static_cast<double*>(pl->itf); P.fv = pl->face_val;
    if (send_tan) {
        P.send_s = static_cast<dual*>(send_tan[0]); P.send_n = static_cast<dual*>(send_tan[1]);
        P.send_w = static_cast<dual*>(send_tan[2]); P.send_e = static_cast<dual*>(send_tan[3]);
    }
    WX_STREAM(st, stream);
    const int nelem = (int)pl->nelem;
#define WX_TAN_CASE(NN)                                                                                                \
    case NN:                                                                                                           \
        hipLaunchKernelGGL((euler_tan_extrap_kernel<NN>), dim3((nelem + Cfg<NN>::EPB - 1) / Cfg<NN>::EPB),              \
                           dim3(Cfg<NN>::BS), 0, st, P);                                                               \
        break;
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
    if (!pl || !q || !v || !out) return fail(WX_ERR_INVALID, "wx_euler3d_jvp_prepared: null argument");
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
    return dispatch_rhs<double>(pl->n, P, st);
}

static wx_status ensure_slot1(wx_euler3d_plan* pl) {
    if (pl->itf2) return WX_OK;
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
        host[i] = P;
    }
    const size_t bytes = sizeof(EulerParams<T>) * b->count;
    hipError_t e = hipMalloc(&b->table, bytes);
    if (e == hipSuccess) e = hipMemcpy(b->table, host.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(WX_ERR_HIP, "batch table upload failed: %s", hipGetErrorString(e));
    return WX_OK;
}

template <typename T>
wx_status batch_extrap(const wx_euler3d_batch* b, const void* q, const double* v, double eps, size_t stride, hipStream_t st) {
    EulerBatchDyn<T> dyn{};
    dyn.stride = stride;
    dyn.stride_re = stride;
    if (v != nullptr && std::is_same<T, dual>::value) {  // dual state (q, eps v) from two real arrays
        dyn.q_re = static_cast<const double*>(q); dyn.q_tan = v; dyn.eps = eps; dyn.jvp = 1;
    } else {
        dyn.q = static_cast<const T*>(q);
        dyn.q_tan = v; dyn.eps = eps;  // float64: shifted state q + eps v (null: plain)
    }
    const EulerParams<T>* t = static_cast<const EulerParams<T>*>(b->table);
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
                    hipStream_t st) {
    EulerBatchDyn<T> dyn{};
    dyn.q = static_cast<const T*>(q); dyn.y = static_cast<const T*>(y); dyn.z = static_cast<const T*>(z);
    dyn.rhs = static_cast<T*>(out);
    dyn.stride = stride;
    dyn.stride_re = stride;
    dyn.q_tan = v; dyn.eps = eps;
    dyn.region = region; dyn.count = region_count(region, b->H, b->V);
    dyn.axpy = axpy; dyn.ca = ca; dyn.cb = cb; dyn.cc = cc; dyn.cd = cd;
    const EulerParams<T>* t = static_cast<const EulerParams<T>*>(b->table);
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
    wx_status s = WX_ERR_INVALID;
    switch (b->dtype) {
        case WX_F64: s = batch_upload<double>(b, plans, send, halo); break;
        case WX_C128: s = batch_upload<cplx>(b, plans, send, halo); break;
        case WX_DUAL128: s = batch_upload<dual>(b, plans, send, halo); break;
    }
    if (s != WX_OK) {
        if (b->table) (void)hipFree(b->table);
        delete b;
        return s;
    }
    *out = b;
    return WX_OK;
}

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
