// Shared by every 3-D Euler kernel: launch shape (Cfg), the parameter block (EulerParams), constants, the element
// descriptor and the value-type helpers (double, complex128, dual).  Included by euler3d.hip only, in this order:
// euler3d_common.h, euler3d_extrap.h, euler3d_rhs.h, euler3d_jvp.h, euler3d_launch.h.
#pragma once

namespace wx {

constexpr int kMaxN = 8;
constexpr int NQ = 5;   // values per face point in the interface buffer and in the edge messages: the prognostic variables

// measured optima (A/B on MI355X): compile-time constants, not build knobs
constexpr int kK1Waves = 1;           // min waves per SIMD requested for the extrapolation kernel
#ifndef WX_K2_WAVES
#define WX_K2_WAVES 4
#endif
constexpr int kK2Waves = WX_K2_WAVES;   // ... for the fused kernel, 8-byte dtypes (n = 8: two workgroups of 8 waves per CU; A/B builds: -DWX_K2_WAVES=6)
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
// 6: the ceiling of every scheme that hides the latency of the fused kernel's state-dependent loads (a workgroup prefetching its
// next element's face and state values under this element's passes): those loads are folded onto the first 64 elements - they
// hit the L2 instead of HBM (wrong results).  7: the static fields too - what the kernel's own structure (barriers, LDS, the
// vector pipe) costs with every load served from cache.  profiles/r05_k2_latency_ceiling.txt
constexpr bool kDiagCacheState = WX_K2_DIAG == 6 || WX_K2_DIAG == 7;
constexpr bool kDiagCacheAll = WX_K2_DIAG == 7;

// the streamed-once static fields go through non-temporal loads
__device__ __forceinline__ double ldm(const double* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ double ldm(gp<const double> p) { return __builtin_nontemporal_load(p.g()); }
// ... unless they are REUSED: the column slabs of a column-invariant metric are read by all n levels of an element (the
// same CU) and by the V elements of the column (the same XCD): cached loads
template <bool CACHED>
__device__ __forceinline__ double ldm_if(const double* p) { return CACHED ? *p : __builtin_nontemporal_load(p); }
template <bool CACHED>
__device__ __forceinline__ double ldm_if(gp<const double> p) { return CACHED ? *p : __builtin_nontemporal_load(p.g()); }
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
    // one-kernel form with pulls (EulerParams::pull_tile): what the tile ACROSS edge e does to the line it would send through its
    // edge towards this tile - its rotation table for that edge and its array of edge coordinates (boundary_sn / _we)
    double prot[4][8];
    const double* pull_x[4];
};

// G: the pointers carry the global address space in device code (wx_common.h: gp) - the batched float64 kernels, whose
// parameter block is loaded from a device table; every other kernel (by-value arguments, known to be global) and the host
// use G = false: plain pointers, the same layout
template <typename T, bool G = false>
struct EulerParams {
    int H, V, nelem, count, region;
    int advection_only, has_damp;
    int rot_zero;          // plan-time finding: christoffel[:, 0:3] (the rotation symbols) is identically zero
    unsigned md_v, md_h, md_w, md_ring;   // floor(2^32 / d) + 1 for d = V, H, H - 2, H^2 - (H-2)^2, or 0: the decodes' fast_div
    int axpy;              // 1: out = ca*y + cb*q + cc*R(q) + cd*z (RK stage / FD Jacobian operator), 0: out = R(q)
    double ca, cb, cc, cd;
    pp<T, const T, G> y;      // nullable (then ca is ignored)
    pp<T, const T, G> z;      // nullable (then cd is ignored)
    // stage pipeline: when itf_out != null the kernel also extrapolates ITS OUTPUT (the next stage's state)
    // to the element faces (phase 1-2 of the NEXT evaluation) into itf_out / nsend_*: no separate K1 pass
    pp<T, T, G> itf_out;
    pp<T, T, G> nsend_s, nsend_n, nsend_w, nsend_e;
    int efilter;           // stage pipeline only: apply the exponential filter to the stage's output before storing it
    int* nan_flag;         // ... and raise this device flag when the stored values hold a NaN (nullable)
    // JVP mode (T = dual only): the state is formed on load as (q_re, jvp_eps * q_tan) from two REAL arrays
    // and only jvp_scale * tangent(R) is stored, as a real array - no complex temporaries in HBM
    int jvp;
    pp<T, const double, G> q_re, q_tan;
    pp<T, double, G> out_tan;
    double jvp_eps, jvp_scale;
    // shifted float64 state with a scale read from DEVICE memory (fgmres' device pass: `A(z / s) * s` of solvers/fgmres.py:172
    // with s = the lagged norm the Gram-Schmidt kernel left on the device): the state is q + (jvp_eps / *dscale) q_tan and the
    // store coefficients cc, cd are multiplied by *dscale.  Null: off.
    const double* dscale;
    // one-kernel form, all tiles of a sphere in one batch (euler3d_brick.h): the tile across edge e is tile pull_tile[e] of the
    // same batch, reached through ITS edge pull_edge[e], which it flips or not (pull_flip[e]) - the launch then forms a tile-edge
    // state from that tile's nodal values itself (the sender's sum, rotation and flip) and no pack launch runs.  -1: no such tile.
    int pull_tile[4], pull_edge[4], pull_flip[4];
    // prepared JVP (wx_euler3d_jvp_prepare): the face VALUES of the linearisation state stay in fv (real,
    // [elem][6][5][n^2]) and in the value halos hv_* for a whole Krylov solve; per product only the face TANGENTS are
    // extrapolated (ft, real, same layout; tangent edge messages through send_* / halo_* as REAL arrays)
    int split;   // 0: off; 1: the JVP kernel reads (fv, ft), written by euler_tan_extrap_kernel (split = 2 there: unused flag)
    pp<T, double, G> ft;
    pp<T, const double, G> fv;
    pp<T, const double, G> hv_s, hv_n, hv_w, hv_e;
    pp<T, const T, G> q;
    pp<T, T, G> rhs;
    pp<T, T, G> itf;  // [elem][6 faces][NQ vars][N2]
    pp<T, const T, G> halo_s, halo_n, halo_w, halo_e;
    pp<T, T, G> send_s, send_n, send_w, send_e;
    // (the float64 kernels see every buffer through global-address-space pointers - wx_common.h: gp -; the complex and dual
    // instantiations keep plain pointers throughout: their code, and with it the last bit of the Jacobian-vector products that
    // the reference's KIOPS statistics are matched on, stays what it was)
    pp<T, const double, G> sg, h, chr, idz;
    pp<T, const double, G> sgi, sgj, sgk, hi, hj, hk;
    pp<T, const double, G> dcoef, duref, bsn, bwe;
    pp<T, const EulerConsts, G> K;  // device memory
    // JVP mode, optional (KIOPS: the n-long part of the next Krylov vector formed in the product's epilogue instead of in a sweep
    // of its own): out_tan = *jzs * (jvp_scale * tangent) + *jza * jz, both coefficients read from DEVICE memory (jzs null: 1)
    pp<T, const double, G> jz;
    const double* jzs;
    const double* jza;
    // ... and, with it, the products of that stored vector with up to two other vectors (the basis rows KIOPS orthogonalises
    // against; jr1 nullable), summed over the workgroup: jpart[2 * w + r], w = the workgroup's linear index in the launch
    pp<T, const double, G> jr0, jr1;
    double* jpart;
    // tangent extrapolation with the input CORRECTED IN PLACE first (KIOPS: the subtraction and the norm of the previous Krylov
    // vector's orthogonalisation, solvers/kiops.py:176-207, done by the kernel that reads the whole vector anyway):
    // t <- t - tch[0] * tcs[0] * tc_r0 [- tch[1] * tcs[1] * tc_r1], written back to tc_out (= q_tan's buffer), |t|^2 summed per
    // workgroup into tc_part[linear workgroup index].  tc_r0 null: off.  tcs null: scales 1.
    pp<T, const double, G> tc_r0, tc_r1;
    const double* tch;
    const double* tcs;
    double* tc_out;
    double* tc_part;
    unsigned long long* stamps;  // WX_K2_DIAG == 1 only, else null
};

static_assert(sizeof(EulerParams<double, true>) == sizeof(EulerParams<double, false>), "one layout, two views");

struct Elem {
    int ek, ej, ei, e;
    bool valid;
};

// n / d with m = floor(2^32 / d) + 1 from the host: one multiply, exact while n d < 2^32 (the host passes m = 0 otherwise, and
// for d < 2: then the division itself)
__device__ __forceinline__ int fast_div(int n, int d, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n / d; }

// slot (position in this launch's processing order = memory order of the region's elements) -> element of the tile
__device__ __forceinline__ Elem decode_elem(int slot, int count, int region, int H, int V, unsigned md_h, unsigned md_w,
                                           unsigned md_ring) {
    Elem r;
    r.valid = slot < count;
    if (!r.valid) slot = 0;
    if (region == WX_REGION_ALL) {
        const int row = fast_div(slot, H, md_h);   // = ek * H + ej
        r.ei = slot - row * H;
        r.ek = fast_div(row, H, md_h);
        r.ej = row - r.ek * H;
    } else if (region == WX_REGION_INTERIOR) {
        const int w = H - 2;
        const int row = fast_div(slot, w, md_w);
        r.ei = 1 + slot - row * w;
        r.ek = fast_div(row, w, md_w);
        r.ej = 1 + row - r.ek * w;
    } else {
        const int w = H > 2 ? H - 2 : 0;
        const int ring = H * H - w * w;
        r.ek = fast_div(slot, ring, md_ring);
        int s = slot - r.ek * ring;
        if (s < H) {
            r.ej = 0;
            r.ei = s;
        } else if (s < 2 * H) {
            r.ej = H - 1;
            r.ei = s - H;
        } else {
            s -= 2 * H;
            const int east = s >= w;
            r.ej = 1 + (east ? s - w : s);
            r.ei = east ? H - 1 : 0;
        }
    }
    r.e = (r.ek * H + r.ej) * H + r.ei;
    return r;
}

// One element per workgroup, one tile per launch: the launchers give the grid the region's own shape - (H, H, V) for ALL,
// (H-2, H-2, V) for INTERIOR, (ring, 1, V) for BOUNDARY - in the same linear order as the slots above, and the element is
// read off the block index: none of the integer divisions of decode_elem (a reciprocal and ~60 vector instructions) stand
// between the start of a wave and its first load.
__device__ __forceinline__ Elem decode_elem_grid(int region, int H) {
    // selects, not branches: the prologue stays one basic block, so that the kernel arguments it needs load in one batch
    const int bx = blockIdx.x, by = blockIdx.y;
    const int w = H > 2 ? H - 2 : 0;
    const int t = bx - 2 * H;             // ring position past the south and north rows: west column, then east column
    const int east = t >= w;
    const int ring_ej = bx < H ? 0 : (bx < 2 * H ? H - 1 : 1 + (east ? t - w : t));
    const int ring_ei = bx < H ? bx : (bx < 2 * H ? bx - H : (east ? H - 1 : 0));
    const bool ring = region == WX_REGION_BOUNDARY;
    const int off = region == WX_REGION_INTERIOR ? 1 : 0;
    Elem r;
    r.valid = true;
    r.ek = blockIdx.z;
    r.ei = ring ? ring_ei : bx + off;
    r.ej = ring ? ring_ej : by + off;
    r.e = (r.ek * H + r.ej) * H + r.ei;
    return r;
}

// the element of slot `slot` of this launch (general form of the plan).  G3: the kernel is a single-tile one launched on
// the region's grid (a compile-time property: the prologue stays straight-line and the kernel arguments load in one batch)
template <int EPB, bool G3, typename T, bool G>
__device__ __forceinline__ Elem decode_blk(const EulerParams<T, G>& P, int slot, int count, int region) {
    if constexpr (EPB == 1 && G3) return decode_elem_grid(region, P.H);
    else return decode_elem(slot, count, region, P.H, P.V, P.md_h, P.md_w, P.md_ring);
}
// single-tile kernels of one-element workgroups run on the region's grid
template <int N>
constexpr bool grid3_for() { return Cfg<N>::EPB == 1; }


// COLUMN form (plans with a column-invariant metric): the V elements of a column follow each other,
// so that the column's metric - one (n x n) slab per field instead of V n of them - is fetched once and found in cache
// by the rest of the column
__device__ __forceinline__ Elem decode_elem_col(int slot, int count, int region, int H, int V, unsigned md_v, unsigned md_h,
                                               unsigned md_w) {
    Elem r;
    r.valid = slot < count;
    if (!r.valid) slot = 0;
    const int c = fast_div(slot, V, md_v);   // the column within the region, in the order decode_elem walks one level of it
    r.ek = slot - c * V;
    if (region == WX_REGION_ALL) {
        r.ej = fast_div(c, H, md_h);
        r.ei = c - r.ej * H;
    } else if (region == WX_REGION_INTERIOR) {
        const int w = H - 2;
        const int row = fast_div(c, w, md_w);
        r.ei = 1 + c - row * w;
        r.ej = 1 + row;
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
            const int east = s >= w;
            r.ej = 1 + (east ? s - w : s);
            r.ei = east ? H - 1 : 0;
        }
    }
    r.e = (r.ek * H + r.ej) * H + r.ei;
    return r;
}
// (workgroups go to the eight XCDs round-robin: give each XCD a contiguous eighth of the launch, so that a column stays in
// one L2; the launch has a multiple of eight workgroups, the surplus finds no element)
__device__ __forceinline__ int xcd_slab_block(int b, int nblocks8) { return (b & 7) * nblocks8 + (b >> 3); }

// float64 plans: the state may be a shifted one, q + eps * v formed on load (finite-difference Jacobian products: no pass
// that materialises Q + eps v); dual plans in JVP mode form (q, eps v) from two real arrays
template <typename T, bool G>
__device__ __forceinline__ T load_q(const EulerParams<T, G>& P, size_t i) {
    if constexpr (std::is_same<T, double>::value) {
        if (P.q_tan != nullptr) return P.q[i] + (P.dscale ? P.jvp_eps / *P.dscale : P.jvp_eps) * P.q_tan[i];  // (same expression as load_state)
        return P.q[i];
    } else if constexpr (std::is_same<T, dual>::value) {
        if (P.jvp) return dual(P.q_re[i], P.jvp_eps * P.q_tan[i]);
        return P.q[i];
    } else {
        return P.q[i];
    }
}
// the five prognostic values of one point, the mode decided ONCE (a branch per load costs the extrapolation
// kernel 12 %: the compiler no longer issues the five loads back to back)
template <typename T, bool G>
__device__ __forceinline__ void load_state(const EulerParams<T, G>& P, size_t o, size_t fs, T& a0, T& a1, T& a2, T& a3, T& a4) {
    if constexpr (std::is_same<T, double>::value) {
        const auto q = P.q;
        a0 = q[o]; a1 = q[fs + o]; a2 = q[2 * fs + o]; a3 = q[3 * fs + o]; a4 = q[4 * fs + o];
        if (P.q_tan != nullptr) {
            const auto v = P.q_tan;
            const double e = P.dscale ? P.jvp_eps / *P.dscale : P.jvp_eps;
            a0 += e * v[o]; a1 += e * v[fs + o]; a2 += e * v[2 * fs + o]; a3 += e * v[3 * fs + o]; a4 += e * v[4 * fs + o];
        }
    } else if constexpr (std::is_same<T, dual>::value) {
        if (P.jvp) {
            const auto r = P.q_re, t = P.q_tan;
            const double e = P.jvp_eps;
            a0 = dual(r[o], e * t[o]); a1 = dual(r[fs + o], e * t[fs + o]); a2 = dual(r[2 * fs + o], e * t[2 * fs + o]);
            a3 = dual(r[3 * fs + o], e * t[3 * fs + o]); a4 = dual(r[4 * fs + o], e * t[4 * fs + o]);
        } else {
            const dual* q = P.q;
            a0 = q[o]; a1 = q[fs + o]; a2 = q[2 * fs + o]; a3 = q[3 * fs + o]; a4 = q[4 * fs + o];
        }
    } else {
        a0 = load_q<T>(P, o); a1 = load_q<T>(P, fs + o); a2 = load_q<T>(P, 2 * fs + o);
        a3 = load_q<T>(P, 3 * fs + o); a4 = load_q<T>(P, 4 * fs + o);
    }
}

template <typename T, bool G>
__device__ __forceinline__ void store_r(const EulerParams<T, G>& P, size_t i, T r) {
    if constexpr (std::is_same<T, dual>::value) {
        if (P.jvp) P.out_tan[i] = P.jvp_scale * r.im;
        else P.rhs[i] = r;
    } else {
        P.rhs[i] = r;
    }
}

}  // namespace wx
