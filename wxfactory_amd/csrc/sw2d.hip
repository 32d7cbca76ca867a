// Shallow-water RHS on one cubed-sphere tile: hand-written HIP for gfx950 (MI355X).
//
// Replaces reference wx_factory/rhs/rhs_sw.py:38-240.  Same two-kernel, element-blocked shape
// as the 3-D Euler path (euler3d.hip):
//   K1 sw_extrap_kernel  rhs_sw.py:76-117  faces of (h+hsurf, hu1, hu2) -> interface buffer,
//                        rotated/flipped tile-edge lines -> send buffers
//   K2 sw_rhs_kernel     rhs_sw.py:119-240 fluxes, derivatives, AUSM, corrections, forcing -> R
// An element is n^2 points (one wave at n=8), so a 256-thread workgroup owns EPB = 256/n^2
// whole elements.  The S7 workload (60x60 elements, n=8) is 5.5 MB of state per panel: the path is
// launch-latency bound, which is why everything after the exchange is ONE kernel.
#include "wx_common.h"
#include "wx_math.h"
#include "wx_panels.h"

#include <cstdlib>
#include <cstring>
#include <new>

// No fused multiply-adds formed BY THE COMPILER in this file: which products and sums it fuses depends on how an expression
// is spread over basic blocks, so the same source line gave different last bits in the two-kernel and in the direct form as
// soon as one of them moved a load (and a balanced state - Williamson 2 - turns an ulp of the cancelling terms into 1e-7 of
// R).  With contraction off every form of the evaluation - extrapolation kernel + RHS kernel, stage pipeline, direct,
// per tile or batched, whole tile or INTERIOR + BOUNDARY - performs the same IEEE operations in source order and agrees to
// the last bit by construction; it is also what the reference's NumPy expressions do.  (The kernels are bound by memory
// round trips, not by the vector pipe: the unfused multiplies cost nothing measurable - profiles/r05_sw_s7_ab.txt.)
#pragma clang fp contract(off)

namespace wx {

constexpr int kMaxN2 = 8;
// min waves per SIMD requested for the RHS kernels (an element is one wave at n = 8; the evaluation is short, a workgroup's
// life is one memory round trip after the other: the more of them a CU holds, the more round trips overlap)
#ifndef WX_SW_WAVES
#define WX_SW_WAVES 1
#endif
constexpr int kSwWaves = WX_SW_WAVES;
#ifndef WX_SW_LEAN
#define WX_SW_LEAN 1       // 0: the round-4 schedule everywhere (A/B builds)
#endif
#ifndef WX_SW_LEAN_WAVES
#define WX_SW_LEAN_WAVES 1
#endif
constexpr int kSwLeanWaves = WX_SW_LEAN_WAVES;

template <int N>
struct Cfg2 {
    static constexpr int N2 = N * N;
    static constexpr int EPB = 256 / N2;
    static constexpr int BS = ((EPB * N2 + 63) / 64) * 64;
    static constexpr int NP = (N % 2 == 0) ? N + 1 : N;  // padded LDS row (see Cfg<N>::NP in euler3d.hip)
    static constexpr int LE = N * NP;
    __host__ __device__ static constexpr int lidx(int jl, int il) { return jl * NP + il; }
};

struct SwConsts {
    double em[kMaxN2], ep[kMaxN2], cm[kMaxN2], cp[kMaxN2];
    double D[kMaxN2 * kMaxN2];
    double rot[4][8];
    int flip[4];
};

// per-launch part (state, result, which elements); everything else is static per plan
template <typename T>
struct SwDyn {
    const T* q;
    T* rhs;
    int count, region;
    // fused explicit Runge-Kutta stage: out = ca*y + cb*q + cc*R(q)  (y nullable); axpy = 0: out = R(q)
    int axpy;
    const T* y;
    double ca, cb, cc;
    // stage pipeline: `slot` = the interface buffer / edge-buffer set that holds the faces of q (extrapolation: that it
    // writes); prepare != 0: the RHS kernel also extrapolates ITS OUTPUT - the next stage's state - to the faces of the
    // OTHER slot and packs its tile-edge lines into the other send set (no separate extrapolation pass for the next stage)
    int slot, prepare;
    // direct form of a batch whose tiles are all each other's neighbours (SwParams::pull): the stacked state, so that a
    // tile-edge face can read the NEIGHBOUR TILE's nodal values itself (null: the edge lines come from the halo buffers)
    const T* q_all = nullptr;
    size_t q_stride = 0;
};

// a tile edge as the SOURCE of a neighbour's halo line: its rotation matrix, the tile's topography (null: flat).  (The boundary
// coordinates along the edge - boundary_we / boundary_sn - are COPIED behind the table, H * N per edge: SwParams::pull_x; a
// pointer to them here would put a second round trip in front of the rotation.)
struct SwPullEdge {
    double rot[8];
    const double* hs;
};

template <typename T>
struct SwParams {
    int H, nelem, has_topo;
    unsigned md_h, md_w;   // floor(2^32 / d) + 1 for d = H, H - 2, or 0 (sw_fast_div)
    tp<T, T> itf;  // [elem][4 faces W,E,S,N][3 vars][N]
    tp<T, const T> halo_s, halo_n, halo_w, halo_e;
    tp<T, T> send_s, send_n, send_w, send_e;
    // slot 1 of the stage pipeline (wx_sw_plan_reserve; null until then)
    tp<T, T> itf2;
    tp<T, const T> halo2_s, halo2_n, halo2_w, halo2_e;
    tp<T, T> send2_s, send2_n, send2_w, send2_e;
    gp<const double> sg, h11, h12, h21, h22;
    gp<const double> c101, c102, c111, c112, c201, c202, c212, c222;
    gp<const double> sgi, sgj, h11i, h21i, h12j, h22j;
    gp<const double> hsurf, dz1, dz2, hsi, hsj;
    gp<const double> bsn, bwe;
    gp<const SwConsts> K;
    // batches only: for each edge (S, N, W, E) 4 * s + 2 * (the source has topography) + (its edge flips the line), s = 4 * (index
    // of the tile in the batch whose send line this halo line is) + that tile's edge, or -1 (sw_batch_build: found by the
    // addresses - a halo line that IS another tile's send line); pull_tab[s]: what the kernel needs of that tile's edge at one
    // load's distance
    int pull[4];
    gp<const SwPullEdge> pull_tab;
    gp<const double> pull_x;   // [4 * tiles][H * N]
};

struct Elem2 {
    int ej, ei, e;
    bool valid;
};

// n / d with m = floor(2^32 / d) + 1 from the host: one multiply, exact while n d < 2^32 (m = 0: the division itself)
__device__ __forceinline__ int sw_fast_div(int n, int d, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n / d; }

__device__ __forceinline__ Elem2 decode_elem2(int slot, int count, int region, int H, unsigned md_h, unsigned md_w) {
    Elem2 r;
    r.valid = slot < count;
    if (!r.valid) slot = 0;
    if (region == WX_REGION_ALL) {
        r.ej = sw_fast_div(slot, H, md_h);
        r.ei = slot - r.ej * H;
    } else if (region == WX_REGION_INTERIOR) {
        const int w = H - 2;
        const int row = sw_fast_div(slot, w, md_w);
        r.ei = 1 + slot - row * w;
        r.ej = 1 + row;
    } else {
        const int w = H > 2 ? H - 2 : 0;
        int s = slot;
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
    r.e = r.ej * H + r.ei;
    return r;
}

// the interface buffer and the edge buffers of one slot
template <typename T>
struct SwSlot {
    tp<T, T> itf;
    tp<T, const T> halo_s, halo_n, halo_w, halo_e;
    tp<T, T> send_s, send_n, send_w, send_e;
};
template <typename T>
__device__ __forceinline__ SwSlot<T> sw_slot(const SwParams<T>& P, int slot) {
    if (slot == 1) return {P.itf2, P.halo2_s, P.halo2_n, P.halo2_w, P.halo2_e, P.send2_s, P.send2_n, P.send2_w, P.send2_e};
    return {P.itf, P.halo_s, P.halo_n, P.halo_w, P.halo_e, P.send_s, P.send_n, P.send_w, P.send_e};
}

// ------------------------------------------------------------------------------------------------
// rhs_sw.py:76-117 on nodal values staged in LDS (h + hsurf, hu1, hu2): one thread per face point extrapolates, writes
// the interface buffer and - on outward tile-edge faces - the rotated / flipped edge line.  Shared by the extrapolation
// kernel and by the RHS kernel's stage-pipeline epilogue.
template <int N, typename T>
__device__ __forceinline__ void sw_extrap_faces(const SwParams<T>& P, T (*fld)[Cfg2<N>::EPB * Cfg2<N>::LE], int slot0, int count,
                                                int region, const SwSlot<T>& S) {
    using C = Cfg2<N>;
    constexpr int EPB = C::EPB, BS = C::BS;
    const int tid = threadIdx.x;
    const int H = P.H;
    for (int fi = tid; fi < EPB * 4 * N; fi += BS) {
        const int le = fi / (4 * N);
        const int r = fi % (4 * N);
        const int f = r / N, k = r % N;
        const Elem2 el = decode_elem2(slot0 + le, count, region, H, P.md_h, P.md_w);
        if (!el.valid) continue;
        const int d = f >> 1, plus = f & 1;
        const int base = d == 0 ? C::lidx(k, 0) : C::lidx(0, k);
        const int stride = d == 0 ? 1 : C::NP;
        gp<const double> w = plus ? P.K->ep : P.K->em;
        T s[3] = {T(0.0), T(0.0), T(0.0)};
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const double wm = w[m];
#pragma unroll
            for (int v = 0; v < 3; ++v) s[v] += wm * fld[v][le * C::LE + base + m * stride];
        }
        tp<T, T> dst = S.itf + ((size_t)el.e * 4 + f) * 3 * N + k;
#pragma unroll
        for (int v = 0; v < 3; ++v) dst[v * N] = s[v];

        int edge = -1, along = 0;
        double X = 0.0;
        if (d == 0 && ((plus && el.ei == H - 1) || (!plus && el.ei == 0))) {
            edge = plus ? E_E : E_W;
            along = el.ej;
            X = P.bwe[el.ej * N + k];
        } else if (d == 1 && ((plus && el.ej == H - 1) || (!plus && el.ej == 0))) {
            edge = plus ? E_N : E_S;
            along = el.ei;
            X = P.bsn[el.ei * N + k];
        }
        tp<T, T> sendp = edge == E_S ? S.send_s : (edge == E_N ? S.send_n : (edge == E_W ? S.send_w : S.send_e));
        if (edge >= 0 && sendp != nullptr) {
            rotate_contra<T>(P.K->rot[edge], X, s[1], s[2]);
            int pos = along * N + k;
            if (P.K->flip[edge]) pos = H * N - 1 - pos;
            const size_t vs = (size_t)H * N;
#pragma unroll
            for (int v = 0; v < 3; ++v) sendp[v * vs + pos] = s[v];
        }
    }
}

template <int N, typename T>
__device__ __forceinline__ void sw_extrap_body(const SwParams<T> P, const SwDyn<T> D) {
    using C = Cfg2<N>;
    constexpr int N2 = C::N2, EPB = C::EPB;
    __shared__ T fld[3][EPB * C::LE];
    const int tid = threadIdx.x;
    const int H = P.H;
    const size_t fs = (size_t)P.nelem * N2;
    {
        const int le = tid / N2, pt = tid % N2;
        const Elem2 el = decode_elem2(blockIdx.x * EPB + le, P.nelem, WX_REGION_ALL, H, P.md_h, P.md_w);
        if (le < EPB && el.valid) {
            const size_t o = (size_t)el.e * N2 + pt;
            const int lp = le * C::LE + C::lidx(pt / N, pt % N);
            T h = D.q[o];
            if (P.has_topo) h = h + P.hsurf[o];
            fld[0][lp] = h;
            fld[1][lp] = D.q[fs + o];
            fld[2][lp] = D.q[2 * fs + o];
        }
    }
    __syncthreads();
    sw_extrap_faces<N, T>(P, fld, blockIdx.x * EPB, P.nelem, WX_REGION_ALL, sw_slot<T>(P, D.slot));
}

// ------------------------------------------------------------------------------------------------
// DIRECT: no interface buffer - the face stage extrapolates the own face states from the element's nodal values (staged in
// LDS) and the neighbour's from the NEIGHBOUR ELEMENT's nodal values in memory (an element of the same tile: its lines are
// one or two cache lines, read by the neighbour's own workgroup too - the launch deals its workgroups to the XCDs in
// contiguous slabs so that the two meet in one L2), tile-edge faces from the received halo lines as ever.  One launch does
// what the extrapolation kernel + the RHS kernel do (the tile-edge lines alone are packed by a ring-only extrapolation
// launch in front of the exchange): no 12 + 24 B/point round trip of the face values, one launch boundary less in a
// 60 us evaluation.  Same arithmetic term by term (the face sums run in the extrapolation kernel's order).
// LEAN (float64, one face pass, interface buffer): the same arithmetic, expression for expression, scheduled for REGISTERS
// instead of for loads in flight per wave - the eight Christoffel fields are loaded behind the first directional pass instead
// of with the state, and the compiler is fenced from hoisting them back - so that the kernel fits 64 registers and a CU
// holds eight workgroups (32 elements) instead of four: the evaluation is a chain of memory round trips per workgroup,
// and what hides them is other workgroups (107 registers: 16 elements per CU, 50 us at S7; see profiles/r05_sw_lean_ab.txt)
template <int N, typename T, bool PIPE = false, bool DIRECT = false, bool LEAN = false>
__device__ __forceinline__ void sw_rhs_body(const SwParams<T> P, const SwDyn<T> D) {
    using C = Cfg2<N>;
    constexpr int N2 = C::N2, EPB = C::EPB, BS = C::BS;
    static_assert(!(PIPE && DIRECT), "the stage pipeline prepares an interface buffer: not for the direct form");
    static_assert(!LEAN || (std::is_same<T, double>::value && EPB * 4 * N <= BS), "the lean schedule: float64, one face pass");
    const SwSlot<T> S = sw_slot<T>(P, PIPE ? D.slot : 0);   // (the plain kernel reads slot 0: its schedule is untouched)
    __shared__ T fld[3][EPB * C::LE];
    __shared__ T fr[EPB][4][3][N];
    __shared__ double sD[N * N], sCm[N], sCp[N];
    // DIRECT: the extrapolated states of both sides of every face point (own side from LDS, neighbour side from the neighbour
    // element's nodal values), formed by ALL threads - one side of one face point each - before the flux stage reads them:
    // half the loads in flight per thread of the one-thread-per-face-point form (160 registers, 3 waves per SIMD)
    __shared__ T fside[DIRECT ? 2 * EPB * 4 * 3 * N : 1];
    __shared__ double sEm[DIRECT ? N : 1], sEp[DIRECT ? N : 1];
#define WX_FSIDE(side_, le_, f_, v_, k_) fside[((((side_) * EPB + (le_)) * 4 + (f_)) * 3 + (v_)) * N + (k_)]
    const int tid = threadIdx.x;
    const int H = P.H;
    const size_t fs = (size_t)P.nelem * N2;
    const int bx = DIRECT ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;   // XCD slabs
    for (int i = tid; i < N * N; i += BS) sD[i] = P.K->D[i];
    if (tid < N) {
        sCm[tid] = P.K->cm[tid];
        sCp[tid] = P.K->cp[tid];
        if constexpr (DIRECT) { sEm[tid] = P.K->em[tid]; sEp[tid] = P.K->ep[tid]; }
    }
    // ---- DIRECT: every load that depends on nothing is issued HERE, before the first barrier - the point's state, metric
    // tensor and Christoffel fields, the neighbour elements' nodal values of the face sides, the interface metric of the face
    // points - so that a workgroup's life is ONE memory round trip and not a chain of five (nodal values | neighbour values |
    // interface metric | point metric | Christoffel fields: each behind a barrier or a dependent stage, with four or five
    // workgroups per CU to hide them - the direct form then ran at 4.6 TB/s on its 239 MB where the rate allows 5.5)
    T dq0 = T(1.0), dq1 = T(0.0), dq2 = T(0.0);
    double e_sg = 1.0, e_h11 = 0, e_h12 = 0, e_h21 = 0, e_h22 = 0;
    double e_c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, e_dz1 = 0.0, e_dz2 = 0.0;
    struct FaceInSw {
        T qo[3], qn[3];
        double sg, hdd, hod;
        T hs_o, hs_n;      // DIRECT: the interface topography of the two sides (subtracted once the sides are there)
        int le, f, k, d, plus;
        bool valid;
    };
    constexpr bool DIRECT_ONE = DIRECT && EPB * 4 * N <= BS;   // one flux task per thread: its interface metric is prefetched
    FaceInSw fin_d;
    fin_d.valid = false;
    // (the interface metric of face point fi; defined here because the direct form calls it before its barriers)
    auto face_meta = [&](int fi, FaceInSw& in) {
        in.valid = false;
        const int le = fi / (4 * N), r = fi % (4 * N), f = r / N, k = r % N;
        const Elem2 el = decode_elem2(bx * EPB + le, D.count, D.region, H, P.md_h, P.md_w);
        if (!el.valid) return;
        const int d = f >> 1, plus = f & 1;
        size_t o_own, o_nbr;
        if (d == 0) {
            const size_t row = (size_t)el.ej * (H + 2);
            o_own = (row + el.ei + 1) * 2 * N + plus * N + k;
            o_nbr = (row + el.ei + 1 + (plus ? 1 : -1)) * 2 * N + (1 - plus) * N + k;
            in.sg = P.sgi[o_own]; in.hdd = P.h11i[o_own]; in.hod = P.h21i[o_own];
            if (P.has_topo) { in.hs_o = T(P.hsi[o_own]); in.hs_n = T(P.hsi[o_nbr]); }
        } else {
            o_own = ((size_t)(el.ej + 1) * H + el.ei) * 2 * N + plus * N + k;
            o_nbr = ((size_t)(el.ej + 1 + (plus ? 1 : -1)) * H + el.ei) * 2 * N + (1 - plus) * N + k;
            in.sg = P.sgj[o_own]; in.hdd = P.h22j[o_own]; in.hod = P.h12j[o_own];
            if (P.has_topo) { in.hs_o = T(P.hsj[o_own]); in.hs_n = T(P.hsj[o_nbr]); }
        }
        in.le = le; in.f = f; in.k = k; in.d = d; in.plus = plus;
        in.valid = true;
    };
    if constexpr (DIRECT) {
        const int le0 = tid / N2, pt0 = tid % N2;
        const Elem2 el0 = decode_elem2(bx * EPB + le0, D.count, D.region, H, P.md_h, P.md_w);
        const bool act0 = le0 < EPB && el0.valid;
        const size_t o0 = (size_t)el0.e * N2 + pt0;
        T hs0 = T(0.0);
        if (act0) {
            dq0 = D.q[o0]; dq1 = D.q[fs + o0]; dq2 = D.q[2 * fs + o0];
            if (P.has_topo) hs0 = T(P.hsurf[o0]);
            e_sg = P.sg[o0];
            e_h11 = P.h11[o0]; e_h12 = P.h12[o0]; e_h21 = P.h21[o0]; e_h22 = P.h22[o0];
            if constexpr (!LEAN) {   // (lean schedule: the forcing fields follow under the second pass, as in the two-kernel form)
                e_c[0] = P.c101[o0]; e_c[1] = P.c102[o0]; e_c[2] = P.c111[o0]; e_c[3] = P.c112[o0];
                e_c[4] = P.c201[o0]; e_c[5] = P.c202[o0]; e_c[6] = P.c212[o0]; e_c[7] = P.c222[o0];
                if (P.has_topo) { e_dz1 = P.dz1[o0]; e_dz2 = P.dz2[o0]; }
            }
        }
        if constexpr (DIRECT_ONE) {
            if (tid < EPB * 4 * N) face_meta(tid, fin_d);
        }
        // the NEIGHBOUR side of every face point: global loads only, nothing to wait for
        // (the sums run in the extrapolation kernel's order: m ascending, per variable)
        for (int t = tid; t < 2 * EPB * 4 * N; t += BS) {
            const int side = t / (EPB * 4 * N), fi = t % (EPB * 4 * N);
            if (side == 0) continue;
            const int le = fi / (4 * N), r = fi % (4 * N), f = r / N, k = r % N;
            const Elem2 el = decode_elem2(bx * EPB + le, D.count, D.region, H, P.md_h, P.md_w);
            if (!el.valid) continue;
            const int d = f >> 1, plus = f & 1;
            T sv[3] = {T(0.0), T(0.0), T(0.0)};
            const int ne = d == 0 ? el.ei + (plus ? 1 : -1) : el.ej + (plus ? 1 : -1);
            // the line of nodal values the side is extrapolated from (null: a received halo line instead), and - when that line
            // belongs to ANOTHER TILE of this launch - the rotation into this tile's basis
            const T* qs = nullptr;
            gp<const double> wn = nullptr, hs = nullptr, rot = nullptr;
            size_t nb = 0;
            int ns = 1;
            double X = 0.0;
            if (ne >= 0 && ne < H) {   // the neighbour element of the same tile: its opposite face, from its nodal values
                const long nelem_nbr = el.e + (d == 0 ? (plus ? 1 : -1) : (plus ? H : -H));
                nb = (size_t)nelem_nbr * N2 + (d == 0 ? k * N : k);
                ns = d == 0 ? 1 : N;
                wn = plus ? P.K->em : P.K->ep;   // (from memory: the LDS copies are not there yet)
                qs = D.q;
                if (P.has_topo) hs = P.hsurf;
            } else if (D.q_all != nullptr) {
                // a tile edge whose neighbour tile is in this launch: ITS edge line, formed here from its nodal values exactly
                // as its ring pack would form it (sw_extrap_faces: the sum, the rotation into this tile's basis, the flip of the
                // line) - the evaluation is then ONE launch, the 5 us ring launch in front of a 43 us kernel is gone.
                // (selected values, not a computed index: that would put the whole parameter block into scratch; everything
                // the line needs comes from THIS tile's parameters - no chain of loads through the neighbour's)
                const bool e_hi = plus != 0;
                const int code = d == 0 ? (e_hi ? P.pull[E_E] : P.pull[E_W]) : (e_hi ? P.pull[E_N] : P.pull[E_S]);
                gp<const SwPullEdge> src = P.pull_tab + (size_t)(code >> 2);
                const int se = (code >> 2) & 3, sd = se >> 1 ? 0 : 1, splus = se & 1;   // (E_S, E_N: lines along i; E_W, E_E: along j)
                int pos = (d == 0 ? el.ej : el.ei) * N + k;
                if (code & 1) pos = H * N - 1 - pos;
                const int sa = pos / N, sk = pos - sa * N;
                const int sei = sd == 0 ? (splus ? H - 1 : 0) : sa, sej = sd == 0 ? sa : (splus ? H - 1 : 0);
                nb = (size_t)(sej * H + sei) * N2 + (sd == 0 ? sk * N : sk);
                ns = sd == 0 ? 1 : N;
                wn = splus ? P.K->ep : P.K->em;   // (the tiles of a batch share their operators: sw_batch_build checks)
                qs = D.q_all + (size_t)(code >> 4) * D.q_stride;
                if (code & 2) hs = src->hs;
                rot = src->rot;
                X = P.pull_x[(size_t)(code >> 2) * (H * N) + sa * N + sk];
            }
            if (qs != nullptr) {
#pragma unroll
                for (int m = 0; m < N; ++m) {
                    const double wm = wn[m];
                    T h = qs[nb + m * ns];
                    if (hs != nullptr) h = h + hs[nb + m * ns];
                    sv[0] += wm * h;
                    sv[1] += wm * qs[fs + nb + m * ns];
                    sv[2] += wm * qs[2 * fs + nb + m * ns];
                }
                if (rot != nullptr) rotate_contra<T>(rot, X, sv[1], sv[2]);
            } else {   // a tile edge: the received halo line
                tp<T, const T> nbr = d == 0 ? (plus ? S.halo_e : S.halo_w) + (size_t)el.ej * N + k
                                            : (plus ? S.halo_n : S.halo_s) + (size_t)el.ei * N + k;
                const size_t nstride = (size_t)H * N;
#pragma unroll
                for (int v = 0; v < 3; ++v) sv[v] = nbr[v * nstride];
            }
#pragma unroll
            for (int v = 0; v < 3; ++v) WX_FSIDE(1, le, f, v, k) = sv[v];
        }
        if (act0) {
            const int lp = le0 * C::LE + C::lidx(pt0 / N, pt0 % N);
            T h = dq0;
            if (P.has_topo) h = h + hs0;
            fld[0][lp] = h; fld[1][lp] = dq1; fld[2][lp] = dq2;
        }
        __syncthreads();
        // the OWN side, from the staged nodal values
        for (int t = tid; t < EPB * 4 * N; t += BS) {
            const int le = t / (4 * N), r = t % (4 * N), f = r / N, k = r % N;
            const Elem2 el = decode_elem2(bx * EPB + le, D.count, D.region, H, P.md_h, P.md_w);
            if (!el.valid) continue;
            const int d = f >> 1, plus = f & 1;
            T sv[3] = {T(0.0), T(0.0), T(0.0)};
            const int base = d == 0 ? C::lidx(k, 0) : C::lidx(0, k);
            const int stride = d == 0 ? 1 : C::NP;
#pragma unroll
            for (int m = 0; m < N; ++m) {
                const double wm = plus ? sEp[m] : sEm[m];
#pragma unroll
                for (int v = 0; v < 3; ++v) sv[v] += wm * fld[v][le * C::LE + base + m * stride];
            }
#pragma unroll
            for (int v = 0; v < 3; ++v) WX_FSIDE(0, le, f, v, k) = sv[v];
        }
        __syncthreads();
    }

    // ---- face stage: AUSM common flux of the 4 faces (rhs_sw.py:157-207), as a load part and a flux part: when one pass of the
    // workgroup covers every face point (n >= 3) the point loads are issued between the two, so that they are in flight
    // under the face arithmetic instead of after it (vector-memory results return in issue order: the faces come first)
    auto face_load = [&](int fi, FaceInSw& in) {
        in.valid = false;
        const int le = fi / (4 * N);
        const int r = fi % (4 * N);
        const int f = r / N, k = r % N;
        const Elem2 el = decode_elem2(bx * EPB + le, D.count, D.region, H, P.md_h, P.md_w);
        if (!el.valid) return;
        const int d = f >> 1, plus = f & 1;
        tp<T, const T> own = S.itf + ((size_t)el.e * 4 + f) * 3 * N + k;
        tp<T, const T> nbr;
        size_t nstride = N;
        long nelem_nbr = -1;   // DIRECT: the neighbour element inside the tile (-1: the face lies on the tile edge)
        size_t o_own, o_nbr;  // slots in the halo-padded interface arrays (own side, neighbour side)
        gp<const double> sgp, hddp, hodp;
        if (d == 0) {
            const int ne = el.ei + (plus ? 1 : -1);
            if (ne >= 0 && ne < H) { nbr = S.itf + ((size_t)(el.e + (plus ? 1 : -1)) * 4 + (f ^ 1)) * 3 * N + k; nelem_nbr = el.e + (plus ? 1 : -1); }
            else { nbr = (plus ? S.halo_e : S.halo_w) + (size_t)el.ej * N + k; nstride = (size_t)H * N; }
            const size_t row = (size_t)el.ej * (H + 2);
            o_own = (row + el.ei + 1) * 2 * N + plus * N + k;
            o_nbr = (row + el.ei + 1 + (plus ? 1 : -1)) * 2 * N + (1 - plus) * N + k;
            sgp = P.sgi; hddp = P.h11i; hodp = P.h21i;
        } else {
            const int ne = el.ej + (plus ? 1 : -1);
            if (ne >= 0 && ne < H) { nbr = S.itf + ((size_t)(el.e + (plus ? H : -H)) * 4 + (f ^ 1)) * 3 * N + k; nelem_nbr = el.e + (plus ? H : -H); }
            else { nbr = (plus ? S.halo_n : S.halo_s) + (size_t)el.ei * N + k; nstride = (size_t)H * N; }
            o_own = ((size_t)(el.ej + 1) * H + el.ei) * 2 * N + plus * N + k;
            o_nbr = ((size_t)(el.ej + 1 + (plus ? 1 : -1)) * H + el.ei) * 2 * N + (1 - plus) * N + k;
            sgp = P.sgj; hddp = P.h22j; hodp = P.h12j;
        }
        T qo[3], qn[3];
        if constexpr (DIRECT) {
            // (never taken: the direct form goes through face_meta + the staged sides)
        } else {
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                qo[v] = own[v * N];
                qn[v] = nbr[v * nstride];
            }
        }
        if (P.has_topo) {  // "substract topo after extrapolation" (rhs_sw.py:153-155), slot by slot
            // (a branch per direction, not a selected pointer: the compiler kept {hsi, hsj} as an ARRAY IN SCRATCH indexed by
            // d - 16 bytes stored by every thread of every workgroup at the top of the kernel, 22 MB of writes per S7
            // evaluation next to 33 MB of results, and a dependent scratch load in front of the face arithmetic)
            if (d == 0) { qo[0] = qo[0] - P.hsi[o_own]; qn[0] = qn[0] - P.hsi[o_nbr]; }
            else { qo[0] = qo[0] - P.hsj[o_own]; qn[0] = qn[0] - P.hsj[o_nbr]; }
        }
        in.sg = sgp[o_own]; in.hdd = hddp[o_own]; in.hod = hodp[o_own];
#pragma unroll
        for (int v = 0; v < 3; ++v) { in.qo[v] = qo[v]; in.qn[v] = qn[v]; }
        in.le = le; in.f = f; in.k = k; in.d = d; in.plus = plus;
        in.valid = true;
    };
    auto face_flux = [&](const FaceInSw& in) {
        if (!in.valid) return;
        const int le = in.le, f = in.f, k = in.k, d = in.d, plus = in.plus;
        const double sg = in.sg, hdd = in.hdd, hod = in.hod;
        T qo[3] = {in.qo[0], in.qo[1], in.qo[2]};
        T qn[3] = {in.qn[0], in.qn[1], in.qn[2]};
        if constexpr (DIRECT) {
#pragma unroll
            for (int v = 0; v < 3; ++v) { qo[v] = WX_FSIDE(0, le, f, v, k); qn[v] = WX_FSIDE(1, le, f, v, k); }
            if (P.has_topo) {  // "substract topo after extrapolation" (rhs_sw.py:153-155), slot by slot
                qo[0] = qo[0] - in.hs_o;
                qn[0] = qn[0] - in.hs_n;
            }
        }
        T qL[3], qR[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            qL[v] = w_sel(plus != 0, qo[v], qn[v]);
            qR[v] = w_sel(plus != 0, qn[v], qo[v]);
        }
        const T unL = w_sel(d == 0, qL[1], qL[2]), unR = w_sel(d == 0, qR[1], qR[2]);
        const T aL = w_sqrt(kGravity * qL[0] * hdd), aR = w_sqrt(kGravity * qR[0] * hdd);
        const T tL = qL[0] * aL, tR = qR[0] * aR;
        const T mL = w_sel(w_real(tL) != 0.0 || w_abs(tL) != 0.0, unL / tL, T(0.0));
        const T mR = w_sel(w_real(tR) != 0.0 || w_abs(tR) != 0.0, unR / tR, T(0.0));
        const T M = 0.25 * ((mL + 1.0) * (mL + 1.0) - (mR - 1.0) * (mR - 1.0));
        const T Mp = w_max(T(0.0), M), Mm = w_min(T(0.0), M);
        T out[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) out[v] = sg * (Mp * aL * qL[v] + Mm * aR * qR[v]);
        const double sgg = sg * (0.5 * kGravity);
        const T hL2 = qL[0] * qL[0], hR2 = qR[0] * qR[0];
        const T pddL = (sgg * hdd) * hL2, pddR = (sgg * hdd) * hR2;
        const T podL = (sgg * hod) * hL2, podR = (sgg * hod) * hR2;
        const T pn = 0.5 * ((1.0 + mL) * pddL + (1.0 - mR) * pddR);  // on the normal component
        const T po = 0.5 * ((1.0 + mL) * podL + (1.0 - mR) * podR);  // on the other one
        out[1] += w_sel(d == 0, pn, po);
        out[2] += w_sel(d == 0, po, pn);
#pragma unroll
        for (int v = 0; v < 3; ++v) fr[le][f][v][k] = out[v];
        };
    constexpr bool ONE_PASS = EPB * 4 * N <= BS && !DIRECT;
    FaceInSw fin;
    fin.valid = false;
    if constexpr (ONE_PASS) {
        if (tid < EPB * 4 * N) face_load(tid, fin);
    } else if constexpr (DIRECT_ONE) {
        face_flux(fin_d);
    } else {
        for (int fi = tid; fi < EPB * 4 * N; fi += BS) {
            FaceInSw in;
            if constexpr (DIRECT) face_meta(fi, in);
            else face_load(fi, in);
            face_flux(in);
        }
    }

    // ---- point stage
    const int le = tid / N2, pt = tid % N2;
    const Elem2 el = decode_elem2(bx * EPB + le, D.count, D.region, H, P.md_h, P.md_w);
    const bool active = (le < EPB) && el.valid;
    const int jl = pt / N, il = pt % N;
    const int lf = le < EPB ? le : 0;
    const int lb = lf * C::LE;
    const int lpt = lb + C::lidx(jl, il);
    const size_t o = (size_t)el.e * N2 + pt;

    T q0 = T(1.0), q1 = T(0.0), q2 = T(0.0);
    double sg = 1.0, h11 = 0, h12 = 0, h21 = 0, h22 = 0;
    double c101 = 0, c102 = 0, c111 = 0, c112 = 0, c201 = 0, c202 = 0, c212 = 0, c222 = 0, dz1 = 0.0, dz2 = 0.0;
    auto load_forcing_fields = [&]() {
        c101 = P.c101[o]; c102 = P.c102[o]; c111 = P.c111[o]; c112 = P.c112[o];
        c201 = P.c201[o]; c202 = P.c202[o]; c212 = P.c212[o]; c222 = P.c222[o];
        if (P.has_topo) { dz1 = P.dz1[o]; dz2 = P.dz2[o]; }
    };
    if constexpr (DIRECT) {   // (everything was loaded at the top of the kernel)
        if (active) {
            q0 = dq0; q1 = dq1; q2 = dq2;
            sg = e_sg; h11 = e_h11; h12 = e_h12; h21 = e_h21; h22 = e_h22;
            if constexpr (!LEAN) {
                c101 = e_c[0]; c102 = e_c[1]; c111 = e_c[2]; c112 = e_c[3];
                c201 = e_c[4]; c202 = e_c[5]; c212 = e_c[6]; c222 = e_c[7];
                dz1 = e_dz1; dz2 = e_dz2;
            }
        }
    } else if (active) {   // every load of the point stage, issued together (LEAN: the state and the metric tensor only)
        q0 = D.q[o]; q1 = D.q[fs + o]; q2 = D.q[2 * fs + o];
        sg = P.sg[o];
        h11 = P.h11[o]; h12 = P.h12[o]; h21 = P.h21[o]; h22 = P.h22[o];
        if constexpr (!LEAN) load_forcing_fields();
    }
    if constexpr (LEAN) __builtin_amdgcn_sched_barrier(0);
    if constexpr (ONE_PASS) face_flux(fin);   // the face values were issued first and are here first; the point loads fly on
    const T u1 = q1 / q0, u2 = q2 / q0;
    const T hsq = q0 * q0;
    T forc1 = T(0.0), forc2 = T(0.0);
    auto forcing = [&]() {
        forc1 = 2.0 * (c101 * q1 + c102 * q2) + c111 * q1 * u1 + 2.0 * c112 * q1 * u2 + kGravity * q0 * (h11 * dz1 + h12 * dz2);
        forc2 = 2.0 * (c201 * q1 + c202 * q2) + 2.0 * c212 * q1 * u2 + c222 * q2 * u2 + kGravity * q0 * (h21 * dz1 + h22 * dz2);
    };
    if constexpr (!LEAN) {
        if (active) forcing();
    }
    T acc0 = T(0.0), acc1 = T(0.0), acc2 = T(0.0);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        if constexpr (LEAN) {
            if (d == 1) {   // the forcing fields: in flight under the second pass, consumed behind it
                __builtin_amdgcn_sched_barrier(0);
                if (active) load_forcing_fields();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const T ud = w_sel(d == 0, u1, u2);
        const double ha = d == 0 ? h11 : h12, hb = d == 0 ? h21 : h22;
        if (d > 0 || DIRECT) __syncthreads();   // (DIRECT: the face stage has read the nodal values out of fld)
        if (le < EPB) {
            fld[0][lpt] = sg * w_sel(d == 0, q1, q2);
            fld[1][lpt] = sg * (q1 * ud + (0.5 * kGravity * ha) * hsq);
            fld[2][lpt] = sg * (q2 * ud + (0.5 * kGravity * hb) * hsq);
        }
        __syncthreads();
        const int base = d == 0 ? lb + C::lidx(jl, 0) : lb + C::lidx(0, il);
        const int stride = d == 0 ? 1 : C::NP;
        const int idx = d == 0 ? il : jl;
        const int fp = d == 0 ? jl : il;
        const double cm = sCm[idx], cp = sCp[idx];
        T dv[3];
        if constexpr (LEAN) {
            // one variable after the other in a ROLLED loop (the same sums in the same order): eight LDS reads in flight
            // instead of the 24 + 8 the unrolled form clusters (80 registers at its peak)
            double dm[N];
#pragma unroll
            for (int m = 0; m < N; ++m) dm[m] = sD[idx * N + m];
            dv[0] = dv[1] = dv[2] = 0.0;
#pragma unroll 1
            for (int v = 0; v < 3; ++v) {
                double a = cm * fr[lf][2 * d][v][fp] + cp * fr[lf][2 * d + 1][v][fp];
#pragma unroll
                for (int m = 0; m < N; ++m) a += dm[m] * fld[v][base + m * stride];
                if (v == 0) dv[0] = a;
                else if (v == 1) dv[1] = a;
                else dv[2] = a;
            }
        } else {
#pragma unroll
            for (int v = 0; v < 3; ++v) dv[v] = cm * fr[lf][2 * d][v][fp] + cp * fr[lf][2 * d + 1][v][fp];
#pragma unroll
            for (int m = 0; m < N; ++m) {
                const double dm = sD[idx * N + m];
#pragma unroll
                for (int v = 0; v < 3; ++v) dv[v] += dm * fld[v][base + m * stride];
            }
        }
        acc0 += dv[0];
        acc1 += dv[1];
        acc2 += dv[2];
    }
    if (!PIPE && !active) return;
    if constexpr (LEAN) {
        __builtin_amdgcn_sched_barrier(0);
        if (active) forcing();
    }
    const double inv_sg = 1.0 / sg;
    T r0 = inv_sg * (-acc0), r1 = inv_sg * (-acc1) - forc1, r2 = inv_sg * (-acc2) - forc2;
    if (active) {
        if (D.axpy) {  // integrators/tvdrk3.py:12-19 stage formed in the store
            r0 = D.cb * q0 + D.cc * r0; r1 = D.cb * q1 + D.cc * r1; r2 = D.cb * q2 + D.cc * r2;
            if (D.y != nullptr) { r0 += D.ca * D.y[o]; r1 += D.ca * D.y[fs + o]; r2 += D.ca * D.y[2 * fs + o]; }
        }
        D.rhs[o] = r0;
        D.rhs[fs + o] = r1;
        D.rhs[2 * fs + o] = r2;
    }
    // ---- stage pipeline: the output is the next stage's state - extrapolate it to the faces while it is in registers
    // (the next evaluation then needs no extrapolation launch: a second instantiation, the plain kernel is as it was)
    if (PIPE && D.prepare) {
        __syncthreads();   // the last directional pass has finished reading fld
        if (le < EPB) {
            T h = r0;
            if (active && P.has_topo) h = h + P.hsurf[o];
            fld[0][lpt] = active ? h : T(0.0);
            fld[1][lpt] = active ? r1 : T(0.0);
            fld[2][lpt] = active ? r2 : T(0.0);
        }
        __syncthreads();
        sw_extrap_faces<N, T>(P, fld, bx * EPB, D.count, D.region, sw_slot<T>(P, 1 - D.slot));
    }
#undef WX_FSIDE
}

// which instantiations take the lean schedule (and the occupancy it is made for)
template <int N, typename T, bool PIPE>
constexpr bool sw_lean() { return WX_SW_LEAN && std::is_same<T, double>::value && !PIPE && Cfg2<N>::EPB * 4 * N <= Cfg2<N>::BS && N >= 6; }
template <int N, typename T, bool PIPE>
constexpr int sw_waves() { return sw_lean<N, T, PIPE>() ? kSwLeanWaves : kSwWaves; }

// one tile per launch: parameters by value
template <int N, typename T>
__global__ __launch_bounds__(Cfg2<N>::BS) void sw_extrap_kernel(const SwParams<T> P, const SwDyn<T> D) {
    sw_extrap_body<N, T>(P, D);
}
template <int N, typename T, bool PIPE>
__global__ __launch_bounds__(Cfg2<N>::BS, (sw_waves<N, T, PIPE>())) void sw_rhs_kernel(const SwParams<T> P, const SwDyn<T> D) {
    sw_rhs_body<N, T, PIPE, false, sw_lean<N, T, PIPE>()>(P, D);
}
template <int N, typename T>
__global__ __launch_bounds__(Cfg2<N>::BS, (sw_waves<N, T, false>())) void sw_rhs_direct_kernel(const SwParams<T> P, const SwDyn<T> D) {
    sw_rhs_body<N, T, false, true, sw_lean<N, T, false>()>(P, D);
}
// the tile-edge lines alone (the ring of elements on the four tile edges): what the direct form still has to exchange
template <int N, typename T>
__device__ __forceinline__ void sw_extrap_ring_body(const SwParams<T> P, const SwDyn<T> D) {
    using C = Cfg2<N>;
    constexpr int N2 = C::N2, EPB = C::EPB;
    __shared__ T fld[3][EPB * C::LE];
    const int tid = threadIdx.x;
    const int H = P.H;
    const size_t fs = (size_t)P.nelem * N2;
    const int w = H > 2 ? H - 2 : 0, ring = H * H - w * w;
    {
        const int le = tid / N2, pt = tid % N2;
        const Elem2 el = decode_elem2(blockIdx.x * EPB + le, ring, WX_REGION_BOUNDARY, H, P.md_h, P.md_w);
        if (le < EPB && el.valid) {
            const size_t o = (size_t)el.e * N2 + pt;
            const int lp = le * C::LE + C::lidx(pt / N, pt % N);
            T h = D.q[o];
            if (P.has_topo) h = h + P.hsurf[o];
            fld[0][lp] = h;
            fld[1][lp] = D.q[fs + o];
            fld[2][lp] = D.q[2 * fs + o];
        }
    }
    __syncthreads();
    sw_extrap_faces<N, T>(P, fld, blockIdx.x * EPB, ring, WX_REGION_BOUNDARY, sw_slot<T>(P, D.slot));
}
template <int N, typename T>
__global__ __launch_bounds__(Cfg2<N>::BS) void sw_extrap_ring_kernel(const SwParams<T> P, const SwDyn<T> D) {
    sw_extrap_ring_body<N, T>(P, D);
}
template <int N, typename T>
__global__ __launch_bounds__(Cfg2<N>::BS) void sw_extrap_ring_batch_kernel(const SwParams<T>* __restrict__ PB, const T* q,
                                                                          size_t stride) {
    SwDyn<T> D{q + (size_t)blockIdx.y * stride, nullptr, 0, 0, 0, nullptr, 0.0, 0.0, 1.0, 0, 0};
    sw_extrap_ring_body<N, T>(PB[blockIdx.y], D);
}
template <int N, typename T>
__global__ __launch_bounds__(Cfg2<N>::BS, (sw_waves<N, T, false>())) void sw_rhs_direct_batch_kernel(const SwParams<T>* __restrict__ PB, const T* q, T* rhs,
                                                                         size_t stride, int count, int region, int axpy,
                                                                         const T* y, double ca, double cb, double cc, int pull) {
    SwDyn<T> D{q + (size_t)blockIdx.y * stride, rhs + (size_t)blockIdx.y * stride, count, region, axpy,
               y ? y + (size_t)blockIdx.y * stride : nullptr, ca, cb, cc, 0, 0};
    if (pull) { D.q_all = q; D.q_stride = stride; }
    sw_rhs_body<N, T, false, true, sw_lean<N, T, false>()>(PB[blockIdx.y], D);
}
// several tiles (the panels one rank owns) per launch: blockIdx.y selects the tile's static parameters
// from a device-resident table; states/results are slices of one stacked array
template <int N, typename T>
__global__ __launch_bounds__(Cfg2<N>::BS) void sw_extrap_batch_kernel(const SwParams<T>* __restrict__ PB, const T* q,
                                                                     size_t stride, int slot) {
    SwDyn<T> D{q + (size_t)blockIdx.y * stride, nullptr, 0, 0, 0, nullptr, 0.0, 0.0, 1.0, slot, 0};
    sw_extrap_body<N, T>(PB[blockIdx.y], D);
}
template <int N, typename T, bool PIPE>
__global__ __launch_bounds__(Cfg2<N>::BS, (sw_waves<N, T, PIPE>())) void sw_rhs_batch_kernel(const SwParams<T>* __restrict__ PB, const T* q, T* rhs,
                                                                  size_t stride, int count, int region, int axpy,
                                                                  const T* y, double ca, double cb, double cc, int slot,
                                                                  int prepare) {
    SwDyn<T> D{q + (size_t)blockIdx.y * stride, rhs + (size_t)blockIdx.y * stride, count, region, axpy,
               y ? y + (size_t)blockIdx.y * stride : nullptr, ca, cb, cc, slot, prepare};
    sw_rhs_body<N, T, PIPE, false, sw_lean<N, T, PIPE>()>(PB[blockIdx.y], D);
}

}  // namespace wx

// ------------------------------------------------------------------------------------------------
using namespace wx;

struct wx_sw_plan {
    int n, H, panel;
    wx_dtype dtype;
    void* itf = nullptr;
    void* itf2 = nullptr;   // interface slot 1 (wx_sw_plan_reserve: the stage pipeline)
    SwConsts* consts = nullptr;
    SwConsts hconsts;       // (the host's copy: what a batch needs to know of its tiles - flips, operators)
    SwParams<double> base;
};

// Several plans driven by one launch per phase (the panels one rank owns; S7 is launch-bound).
struct wx_sw_batch {
    int n, H, count;
    wx_dtype dtype;
    void* table = nullptr;  // device: SwParams<T>[count]
    bool pipelined = false; // the table holds both slots (wx_sw_batch_create_pipelined)
    bool pulls = false;     // every halo line of every tile is the send line of a tile of this batch (SwParams::pull is set)
    void* pull_tab = nullptr;   // device: SwPullEdge[4 * count] (pulls only)
};

namespace {

template <typename T>
SwParams<T> make_sw_params(const wx_sw_plan* pl) {
    SwParams<T> P;
    const SwParams<double>& b = pl->base;
    P.H = b.H; P.nelem = b.nelem; P.has_topo = b.has_topo; P.md_h = b.md_h; P.md_w = b.md_w;
    P.itf = static_cast<T*>(pl->itf);
    P.halo_s = P.halo_n = P.halo_w = P.halo_e = nullptr;
    P.send_s = P.send_n = P.send_w = P.send_e = nullptr;
    P.itf2 = static_cast<T*>(pl->itf2);
    P.halo2_s = P.halo2_n = P.halo2_w = P.halo2_e = nullptr;
    P.send2_s = P.send2_n = P.send2_w = P.send2_e = nullptr;
    P.sg = b.sg; P.h11 = b.h11; P.h12 = b.h12; P.h21 = b.h21; P.h22 = b.h22;
    P.c101 = b.c101; P.c102 = b.c102; P.c111 = b.c111; P.c112 = b.c112;
    P.c201 = b.c201; P.c202 = b.c202; P.c212 = b.c212; P.c222 = b.c222;
    P.sgi = b.sgi; P.sgj = b.sgj; P.h11i = b.h11i; P.h21i = b.h21i; P.h12j = b.h12j; P.h22j = b.h22j;
    P.hsurf = b.hsurf; P.dz1 = b.dz1; P.dz2 = b.dz2; P.hsi = b.hsi; P.hsj = b.hsj;
    P.bsn = b.bsn; P.bwe = b.bwe; P.K = pl->consts;
    P.pull[0] = P.pull[1] = P.pull[2] = P.pull[3] = -1;
    P.pull_tab = nullptr;
    P.pull_x = nullptr;
    return P;
}

template <typename T>
void set_edges(SwParams<T>& P, void* const send[4], const void* const halo[4], int slot = 0) {
    if (send && slot == 0) {
        P.send_s = static_cast<T*>(send[0]); P.send_n = static_cast<T*>(send[1]);
        P.send_w = static_cast<T*>(send[2]); P.send_e = static_cast<T*>(send[3]);
    }
    if (halo && slot == 0) {
        P.halo_s = static_cast<const T*>(halo[0]); P.halo_n = static_cast<const T*>(halo[1]);
        P.halo_w = static_cast<const T*>(halo[2]); P.halo_e = static_cast<const T*>(halo[3]);
    }
    if (send && slot == 1) {
        P.send2_s = static_cast<T*>(send[0]); P.send2_n = static_cast<T*>(send[1]);
        P.send2_w = static_cast<T*>(send[2]); P.send2_e = static_cast<T*>(send[3]);
    }
    if (halo && slot == 1) {
        P.halo2_s = static_cast<const T*>(halo[0]); P.halo2_n = static_cast<const T*>(halo[1]);
        P.halo2_w = static_cast<const T*>(halo[2]); P.halo2_e = static_cast<const T*>(halo[3]);
    }
}

int sw_region_count(int region, int H) {
    const int w = H > 2 ? H - 2 : 0;
    return region == WX_REGION_ALL ? H * H : (region == WX_REGION_INTERIOR ? w * w : H * H - w * w);
}

// what: 0 single extrap, 1 single rhs, 2 batch extrap, 3 batch rhs
template <int N, typename T>
wx_status sw_launch(int what, const SwParams<T>* P, const SwDyn<T>& D, const SwParams<T>* table, int nb, size_t stride,
                    hipStream_t st) {
    using C = Cfg2<N>;
    const int cnt = (what == 0 || what == 2) ? (P ? P->nelem : D.count) : D.count;
    if (cnt == 0) return WX_OK;
    const int grid = (cnt + C::EPB - 1) / C::EPB;
    const bool pipe = D.prepare != 0 || D.slot != 0;   // the stage pipeline's instantiation
    if (what >= 4) {   // the direct form: ring-only extrapolation (4, 6), one-launch RHS with its workgroups in XCD slabs (5, 7)
        if (what == 4 || what == 6) {
            const int H = (int)D.count;   // (the ring launches carry H in D.count)
            const int ww = H > 2 ? H - 2 : 0, ring = H * H - ww * ww;
            const int g = (ring + C::EPB - 1) / C::EPB;
            if (g == 0) return WX_OK;
            if (what == 4) hipLaunchKernelGGL((sw_extrap_ring_kernel<N, T>), dim3(g), dim3(C::BS), 0, st, *P, D);
            else hipLaunchKernelGGL((sw_extrap_ring_batch_kernel<N, T>), dim3(g, nb), dim3(C::BS), 0, st, table, D.q, stride);
        } else {
            const int g8 = 8 * ((grid + 7) / 8);   // a multiple of eight workgroups: the surplus finds no element
            if (what == 5) hipLaunchKernelGGL((sw_rhs_direct_kernel<N, T>), dim3(g8), dim3(C::BS), 0, st, *P, D);
            else hipLaunchKernelGGL((sw_rhs_direct_batch_kernel<N, T>), dim3(g8, nb), dim3(C::BS), 0, st, table, D.q, D.rhs, stride,
                                    D.count, D.region, D.axpy, D.y, D.ca, D.cb, D.cc, D.q_all != nullptr ? 1 : 0);
        }
        WX_HIP_TRY(hipGetLastError());
        return WX_OK;
    }
    switch (what) {
        case 0: hipLaunchKernelGGL((sw_extrap_kernel<N, T>), dim3(grid), dim3(C::BS), 0, st, *P, D); break;
        case 1:
            if (pipe) hipLaunchKernelGGL((sw_rhs_kernel<N, T, true>), dim3(grid), dim3(C::BS), 0, st, *P, D);
            else hipLaunchKernelGGL((sw_rhs_kernel<N, T, false>), dim3(grid), dim3(C::BS), 0, st, *P, D);
            break;
        case 2: hipLaunchKernelGGL((sw_extrap_batch_kernel<N, T>), dim3(grid, nb), dim3(C::BS), 0, st, table, D.q, stride, D.slot); break;
        default:
            if (pipe) hipLaunchKernelGGL((sw_rhs_batch_kernel<N, T, true>), dim3(grid, nb), dim3(C::BS), 0, st, table, D.q, D.rhs,
                                         stride, D.count, D.region, D.axpy, D.y, D.ca, D.cb, D.cc, D.slot, D.prepare);
            else hipLaunchKernelGGL((sw_rhs_batch_kernel<N, T, false>), dim3(grid, nb), dim3(C::BS), 0, st, table, D.q, D.rhs,
                                    stride, D.count, D.region, D.axpy, D.y, D.ca, D.cb, D.cc, 0, 0);
    }
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <typename T>
wx_status sw_dispatch(int n, int what, const SwParams<T>* P, const SwDyn<T>& D, const SwParams<T>* table, int nb,
                      size_t stride, hipStream_t st) {
    switch (n) {
        case 2: return sw_launch<2, T>(what, P, D, table, nb, stride, st);
        case 3: return sw_launch<3, T>(what, P, D, table, nb, stride, st);
        case 4: return sw_launch<4, T>(what, P, D, table, nb, stride, st);
        case 5: return sw_launch<5, T>(what, P, D, table, nb, stride, st);
        case 6: return sw_launch<6, T>(what, P, D, table, nb, stride, st);
        case 7: return sw_launch<7, T>(what, P, D, table, nb, stride, st);
        case 8: return sw_launch<8, T>(what, P, D, table, nb, stride, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", n);
}

// slot: the interface / edge set that holds (extrapolation: receives) the faces of q; next_send + prepare: the stage pipeline
template <typename T>
wx_status sw_run(wx_sw_plan* pl, bool extrap, const void* q, void* const send[4], const void* const halo[4], void* rhs,
                 int region, int count, hipStream_t st, int axpy = 0, const void* y = nullptr, double ca = 0.0,
                 double cb = 0.0, double cc = 1.0, int slot = 0, void* const next_send[4] = nullptr, int prepare = 0) {
    SwParams<T> P = make_sw_params<T>(pl);
    set_edges<T>(P, send, halo, slot);
    if (prepare) set_edges<T>(P, next_send, nullptr, 1 - slot);
    SwDyn<T> D{static_cast<const T*>(q), static_cast<T*>(rhs), count, region, axpy, static_cast<const T*>(y), ca, cb, cc,
               slot, prepare};
    return sw_dispatch<T>(pl->n, extrap ? 0 : 1, &P, D, nullptr, 0, 0, st);
}

// two tiles extrapolate with the same weights on the same grid (a line of one can be formed with the constants of the other)
bool sw_same_operators(const wx_sw_plan* a, const wx_sw_plan* b) {
    return a->n == b->n && a->H == b->H && memcmp(a->hconsts.em, b->hconsts.em, sizeof(a->hconsts.em)) == 0 &&
           memcmp(a->hconsts.ep, b->hconsts.ep, sizeof(a->hconsts.ep)) == 0;
}

template <typename T>
wx_status sw_batch_build(wx_sw_batch* b, wx_sw_plan* const plans[], int count, void* const send[][4],
                         const void* const halo[][4], void* const send2[][4] = nullptr, const void* const halo2[][4] = nullptr) {
    SwParams<T>* host = new (std::nothrow) SwParams<T>[count];
    if (!host) return fail(WX_ERR_NOMEM, "out of host memory");
    for (int i = 0; i < count; ++i) {
        host[i] = make_sw_params<T>(plans[i]);
        set_edges<T>(host[i], send[i], halo[i]);
        if (send2 && halo2) set_edges<T>(host[i], send2[i], halo2[i], 1);
    }
    // which tile's send line is this halo line?  (same-rank neighbours: the exchange aliases the two.)  When EVERY halo line of
    // the batch is one, the direct form needs no ring pack: its tile-edge faces read the neighbour tile's nodal values.
    // WXHIP_SW_PULL=0: keep the packed lines (the A/B and the test of the two against each other).
    const char* env = getenv("WXHIP_SW_PULL");
    bool all = !(env && env[0] == '0') && !(send2 && halo2) && count > 0 && send && halo;
    for (int i = 0; all && i < count; ++i)
        for (int e = 0; all && e < 4; ++e) {
            int found = -1;
            for (int j = 0; halo[i][e] && found < 0 && j < count; ++j)
                for (int f = 0; f < 4; ++f)
                    if (send[j][f] == halo[i][e] && sw_same_operators(plans[j], plans[i])) { found = 4 * j + f; break; }
            if (found < 0) { all = false; break; }
            const wx_sw_plan* src = plans[found >> 2];
            host[i].pull[e] = 4 * found + (src->base.has_topo ? 2 : 0) + (src->hconsts.flip[found & 3] ? 1 : 0);
        }
    hipError_t e = hipSuccess;
    if (all) {   // the tiles' edges as sources
        SwPullEdge* tab = new (std::nothrow) SwPullEdge[4 * count];
        if (!tab) { delete[] host; return fail(WX_ERR_NOMEM, "out of host memory"); }
        for (int j = 0; j < count; ++j)
            for (int f = 0; f < 4; ++f) {
                SwPullEdge& t = tab[4 * j + f];
                memcpy(t.rot, plans[j]->hconsts.rot[f], sizeof(t.rot));
                t.hs = plans[j]->base.has_topo ? plans[j]->base.hsurf.p : nullptr;
            }
        const size_t line = (size_t)plans[0]->H * plans[0]->n, tab_bytes = sizeof(SwPullEdge) * 4 * count;
        e = hipMalloc(&b->pull_tab, tab_bytes + sizeof(double) * 4 * count * line);
        if (e == hipSuccess) e = hipMemcpy(b->pull_tab, tab, tab_bytes, hipMemcpyHostToDevice);
        delete[] tab;
        double* xs = reinterpret_cast<double*>(static_cast<char*>(b->pull_tab) + tab_bytes);
        for (int j = 0; e == hipSuccess && j < count; ++j)
            for (int f = 0; e == hipSuccess && f < 4; ++f)
                e = hipMemcpy(xs + (size_t)(4 * j + f) * line, f >> 1 ? plans[j]->base.bwe.p : plans[j]->base.bsn.p,
                              sizeof(double) * line, hipMemcpyDeviceToDevice);
        for (int i = 0; i < count; ++i) {
            host[i].pull_tab = static_cast<const SwPullEdge*>(b->pull_tab);
            host[i].pull_x = xs;
        }
    } else {
        for (int i = 0; i < count; ++i) host[i].pull[0] = host[i].pull[1] = host[i].pull[2] = host[i].pull[3] = -1;
    }
    b->pulls = all;
    if (e == hipSuccess) e = hipMalloc(&b->table, sizeof(SwParams<T>) * count);
    if (e == hipSuccess) e = hipMemcpy(b->table, host, sizeof(SwParams<T>) * count, hipMemcpyHostToDevice);
    delete[] host;
    if (e != hipSuccess) return fail(WX_ERR_HIP, "wx_sw_batch_create: %s", hipGetErrorString(e));
    return WX_OK;
}

template <typename T>
wx_status sw_batch_run(wx_sw_batch* b, bool extrap, const void* q, void* rhs, size_t stride, int region, hipStream_t st,
                       int axpy = 0, const void* y = nullptr, double ca = 0.0, double cb = 0.0, double cc = 1.0, int slot = 0,
                       int prepare = 0) {
    SwDyn<T> D{static_cast<const T*>(q), static_cast<T*>(rhs), extrap ? b->H * b->H : sw_region_count(region, b->H), region,
               axpy, static_cast<const T*>(y), ca, cb, cc, slot, prepare};
    return sw_dispatch<T>(b->n, extrap ? 2 : 3, nullptr, D, static_cast<const SwParams<T>*>(b->table), b->count, stride, st);
}

}  // namespace

template <typename T>
static wx_status sw_run_direct(wx_sw_plan* pl, const void* q, const void* const halo[4], void* out, int region, int axpy,
                               const void* y, double a, double b, double c, hipStream_t st) {
    SwParams<T> P = make_sw_params<T>(pl);
    set_edges<T>(P, nullptr, halo);
    SwDyn<T> D{static_cast<const T*>(q), static_cast<T*>(out), sw_region_count(region, pl->H), region, axpy,
               static_cast<const T*>(y), a, b, c, 0, 0};
    return sw_dispatch<T>(pl->n, 5, &P, D, nullptr, 0, 0, st);
}

template <typename T>
static wx_status sw_batch_direct(wx_sw_batch* b, int what, const void* q, void* out, size_t stride, int region, int axpy,
                                 const void* y, double ca, double cb, double cc, hipStream_t st) {
    SwDyn<T> D{static_cast<const T*>(q), static_cast<T*>(out), what == 6 ? b->H : sw_region_count(region, b->H), region, axpy,
               static_cast<const T*>(y), ca, cb, cc, 0, 0};
    if (what == 7 && b->pulls) D.q_all = D.q;   // (the batch kernel takes the stacked state and its stride from its own arguments)
    return sw_dispatch<T>(b->n, what, nullptr, D, static_cast<const SwParams<T>*>(b->table), b->count, stride, st);
}

extern "C" {

wx_status wx_sw_plan_create(wx_sw_plan** out, int n, int H, wx_dtype dtype, int panel, const wx_dfr_ops* ops,
                            const wx_sw_metric* m) {
    static const int all_edges[4] = {1, 1, 1, 1};
    return wx_sw_plan_create_tile(out, n, H, dtype, panel, all_edges, ops, m);
}

wx_status wx_sw_plan_create_tile(wx_sw_plan** out, int n, int H, wx_dtype dtype, int panel, const int on_panel_edge[4],
                                 const wx_dfr_ops* ops, const wx_sw_metric* m) {
    if (!on_panel_edge) return fail(WX_ERR_INVALID, "wx_sw_plan_create_tile: null on_panel_edge");
    if (!out || !ops || !m) return fail(WX_ERR_INVALID, "wx_sw_plan_create: null argument");
    *out = nullptr;
    if (n < 2 || n > kMaxN2) return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..%d", n, kMaxN2);
    if (H < 1) return fail(WX_ERR_INVALID, "bad tile size H=%d", H);
    if (panel < 0 || panel > 5) return fail(WX_ERR_INVALID, "panel %d not in 0..5", panel);
    if (dtype != WX_F64 && dtype != WX_C128 && dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "unknown dtype %d", (int)dtype);
    if (!ops->extrap_neg || !ops->extrap_pos || !ops->diff_solpt || !ops->correction)
        return fail(WX_ERR_INVALID, "wx_dfr_ops has a null member");
    const double* req[] = {m->sqrtG, m->H_contra_11, m->H_contra_12, m->H_contra_21, m->H_contra_22,
                           m->christoffel_1_01, m->christoffel_1_02, m->christoffel_1_11, m->christoffel_1_12,
                           m->christoffel_2_01, m->christoffel_2_02, m->christoffel_2_12, m->christoffel_2_22,
                           m->sqrtG_itf_i, m->sqrtG_itf_j, m->H_contra_11_itf_i, m->H_contra_21_itf_i,
                           m->H_contra_12_itf_j, m->H_contra_22_itf_j, m->boundary_sn, m->boundary_we};
    for (const double* p : req)
        if (!p) return fail(WX_ERR_INVALID, "wx_sw_metric has a null member");
    const int ntopo = (m->hsurf != nullptr) + (m->dzdx1 != nullptr) + (m->dzdx2 != nullptr) +
                      (m->hsurf_itf_i != nullptr) + (m->hsurf_itf_j != nullptr);
    if (ntopo != 0 && ntopo != 5) return fail(WX_ERR_INVALID, "topography needs all five arrays or none");

    wx_sw_plan* pl = new (std::nothrow) wx_sw_plan();
    if (!pl) return fail(WX_ERR_NOMEM, "out of host memory");
    pl->n = n; pl->H = H; pl->panel = panel; pl->dtype = dtype;
    const size_t esz = dtype == WX_F64 ? 8 : 16;
    hipError_t e = hipMalloc(&pl->itf, (size_t)H * H * 4 * 3 * n * esz);
    if (e == hipSuccess) e = hipMalloc((void**)&pl->consts, sizeof(SwConsts));
    SwConsts hc;
    memset(&hc, 0, sizeof(hc));
    for (int i = 0; i < n; ++i) {
        hc.em[i] = ops->extrap_neg[i]; hc.ep[i] = ops->extrap_pos[i];
        hc.cm[i] = ops->correction[2 * i]; hc.cp[i] = ops->correction[2 * i + 1];
        for (int j = 0; j < n; ++j) hc.D[i * n + j] = ops->diff_solpt[i * n + j];
    }
    static const double identity[8] = {1, 0, 0, 0, 0, 1, 0, 0};
    for (int ed = 0; ed < 4; ++ed) {  // interior tile edges: no flip, no rotation
        hc.flip[ed] = on_panel_edge[ed] ? kFlip[panel][ed] : 0;
        for (int i = 0; i < 8; ++i) hc.rot[ed][i] = on_panel_edge[ed] ? kRot[panel][ed][i] : identity[i];
    }
    pl->hconsts = hc;
    if (e == hipSuccess) e = hipMemcpy(pl->consts, &hc, sizeof(hc), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (pl->itf) (void)hipFree(pl->itf);
        if (pl->consts) (void)hipFree(pl->consts);
        delete pl;
        return fail(WX_ERR_HIP, "wx_sw_plan_create: device allocation failed: %s", hipGetErrorString(e));
    }
    SwParams<double>& b = pl->base;
    b.H = H; b.nelem = H * H; b.has_topo = ntopo == 5;
    b.md_h = (H >= 2 && (unsigned long long)H * H * H < (1ull << 32)) ? (unsigned)((1ull << 32) / (unsigned)H) + 1u : 0u;
    b.md_w = (H >= 4 && (unsigned long long)H * H * H < (1ull << 32)) ? (unsigned)((1ull << 32) / (unsigned)(H - 2)) + 1u : 0u;
    b.sg = m->sqrtG; b.h11 = m->H_contra_11; b.h12 = m->H_contra_12; b.h21 = m->H_contra_21; b.h22 = m->H_contra_22;
    b.c101 = m->christoffel_1_01; b.c102 = m->christoffel_1_02; b.c111 = m->christoffel_1_11;
    b.c112 = m->christoffel_1_12; b.c201 = m->christoffel_2_01; b.c202 = m->christoffel_2_02;
    b.c212 = m->christoffel_2_12; b.c222 = m->christoffel_2_22;
    b.sgi = m->sqrtG_itf_i; b.sgj = m->sqrtG_itf_j; b.h11i = m->H_contra_11_itf_i; b.h21i = m->H_contra_21_itf_i;
    b.h12j = m->H_contra_12_itf_j; b.h22j = m->H_contra_22_itf_j;
    b.hsurf = m->hsurf; b.dz1 = m->dzdx1; b.dz2 = m->dzdx2; b.hsi = m->hsurf_itf_i; b.hsj = m->hsurf_itf_j;
    b.bsn = m->boundary_sn; b.bwe = m->boundary_we;
    *out = pl;
    return WX_OK;
}

wx_status wx_sw_plan_destroy(wx_sw_plan* pl) {
    if (!pl) return WX_OK;
    hipError_t e = hipFree(pl->itf);
    if (pl->itf2) (void)hipFree(pl->itf2);
    hipError_t e2 = hipFree(pl->consts);
    if (e == hipSuccess) e = e2;
    delete pl;
    if (e != hipSuccess) return fail(WX_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
    return WX_OK;
}

size_t wx_sw_edge_count(const wx_sw_plan* pl) { return pl ? (size_t)3 * pl->H * pl->n : 0; }
wx_dtype wx_sw_plan_dtype(const wx_sw_plan* pl) { return pl ? pl->dtype : WX_F64; }

wx_status wx_sw_extrap_pack(wx_sw_plan* pl, const void* q, void* const send[4], wx_stream stream) {
    if (!pl || !q) return fail(WX_ERR_INVALID, "wx_sw_extrap_pack: null argument");
    WX_STREAM(st, stream);
    switch (pl->dtype) {
        case WX_F64: return sw_run<double>(pl, true, q, send, nullptr, nullptr, 0, 0, st);
        case WX_C128: return sw_run<cplx>(pl, true, q, send, nullptr, nullptr, 0, 0, st);
        case WX_DUAL128: return sw_run<dual>(pl, true, q, send, nullptr, nullptr, 0, 0, st);
    }
    return fail(WX_ERR_INVALID, "bad plan dtype");
}

wx_status wx_sw_rhs(wx_sw_plan* pl, const void* q, const void* const halo[4], void* rhs, wx_region region,
                    wx_stream stream) {
    if (!pl || !q || !rhs) return fail(WX_ERR_INVALID, "wx_sw_rhs: null argument");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    if (region != WX_REGION_INTERIOR) {
        if (!halo) return fail(WX_ERR_INVALID, "wx_sw_rhs: halo is required for this region");
        for (int e = 0; e < 4; ++e)
            if (!halo[e]) return fail(WX_ERR_INVALID, "wx_sw_rhs: halo[%d] is null", e);
    }
    const int count = sw_region_count(region, pl->H);
    WX_STREAM(st, stream);
    switch (pl->dtype) {
        case WX_F64: return sw_run<double>(pl, false, q, nullptr, halo, rhs, region, count, st);
        case WX_C128: return sw_run<cplx>(pl, false, q, nullptr, halo, rhs, region, count, st);
        case WX_DUAL128: return sw_run<dual>(pl, false, q, nullptr, halo, rhs, region, count, st);
    }
    return fail(WX_ERR_INVALID, "bad plan dtype");
}

static wx_status sw_check_rhs_args(const void* pl, const void* q, const void* out, const void* const halo[4], int region) {
    if (!pl || !q || !out) return fail(WX_ERR_INVALID, "wx_sw_rhs: null argument");
    if (out == q) return fail(WX_ERR_INVALID, "wx_sw_rhs: output must not alias the state");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", region);
    if (region != WX_REGION_INTERIOR) {
        if (!halo) return fail(WX_ERR_INVALID, "wx_sw_rhs: halo is required for this region");
        for (int e = 0; e < 4; ++e)
            if (!halo[e]) return fail(WX_ERR_INVALID, "wx_sw_rhs: halo[%d] is null", e);
    }
    return WX_OK;
}

wx_status wx_sw_rhs_axpy(wx_sw_plan* pl, const void* q, const void* const halo[4], const void* y, void* out, double a,
                         double b, double c, wx_region region, wx_stream stream) {
    wx_status ok = sw_check_rhs_args(pl, q, out, halo, region);
    if (ok != WX_OK) return ok;
    const int count = sw_region_count(region, pl->H);
    WX_STREAM(st, stream);
    switch (pl->dtype) {
        case WX_F64: return sw_run<double>(pl, false, q, nullptr, halo, out, region, count, st, 1, y, a, b, c);
        case WX_C128: return sw_run<cplx>(pl, false, q, nullptr, halo, out, region, count, st, 1, y, a, b, c);
        default: return sw_run<dual>(pl, false, q, nullptr, halo, out, region, count, st, 1, y, a, b, c);
    }
}

wx_status wx_sw_batch_rhs_axpy(wx_sw_batch* bt, const void* q, const void* y, void* out, size_t panel_stride, double a,
                               double b, double c, wx_region region, wx_stream stream) {
    if (!bt || !q || !out) return fail(WX_ERR_INVALID, "wx_sw_batch_rhs_axpy: null argument");
    if (out == q) return fail(WX_ERR_INVALID, "wx_sw_batch_rhs_axpy: output must not alias the state");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    WX_STREAM(st, stream);
    switch (bt->dtype) {
        case WX_F64: return sw_batch_run<double>(bt, false, q, out, panel_stride, region, st, 1, y, a, b, c);
        case WX_C128: return sw_batch_run<cplx>(bt, false, q, out, panel_stride, region, st, 1, y, a, b, c);
        default: return sw_batch_run<dual>(bt, false, q, out, panel_stride, region, st, 1, y, a, b, c);
    }
}

wx_status wx_sw_batch_create(wx_sw_batch** out, wx_sw_plan* const plans[], int count, void* const send[][4],
                             const void* const halo[][4]) {
    if (!out || !plans || !send || !halo || count < 1) return fail(WX_ERR_INVALID, "wx_sw_batch_create: bad argument");
    *out = nullptr;
    for (int i = 0; i < count; ++i) {
        if (!plans[i]) return fail(WX_ERR_INVALID, "wx_sw_batch_create: plans[%d] is null", i);
        if (plans[i]->n != plans[0]->n || plans[i]->H != plans[0]->H || plans[i]->dtype != plans[0]->dtype)
            return fail(WX_ERR_INVALID, "wx_sw_batch_create: plans differ in n, H or dtype");
        for (int e = 0; e < 4; ++e)
            if (!send[i][e] || !halo[i][e]) return fail(WX_ERR_INVALID, "wx_sw_batch_create: null edge buffer");
    }
    wx_sw_batch* b = new (std::nothrow) wx_sw_batch();
    if (!b) return fail(WX_ERR_NOMEM, "out of host memory");
    b->n = plans[0]->n; b->H = plans[0]->H; b->count = count; b->dtype = plans[0]->dtype;
    wx_status st;
    switch (b->dtype) {
        case WX_F64: st = sw_batch_build<double>(b, plans, count, send, halo); break;
        case WX_C128: st = sw_batch_build<cplx>(b, plans, count, send, halo); break;
        default: st = sw_batch_build<dual>(b, plans, count, send, halo);
    }
    if (st != WX_OK) {
        if (b->table) (void)hipFree(b->table);
        if (b->pull_tab) (void)hipFree(b->pull_tab);
        delete b;
        return st;
    }
    *out = b;
    return WX_OK;
}

wx_status wx_sw_batch_destroy(wx_sw_batch* b) {
    if (!b) return WX_OK;
    hipError_t e = hipFree(b->table);
    if (b->pull_tab) { const hipError_t e2 = hipFree(b->pull_tab); if (e == hipSuccess) e = e2; }
    delete b;
    if (e != hipSuccess) return fail(WX_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
    return WX_OK;
}

wx_status wx_sw_batch_extrap_pack(wx_sw_batch* b, const void* q, size_t panel_stride, wx_stream stream) {
    if (!b || !q) return fail(WX_ERR_INVALID, "wx_sw_batch_extrap_pack: null argument");
    WX_STREAM(st, stream);
    switch (b->dtype) {
        case WX_F64: return sw_batch_run<double>(b, true, q, nullptr, panel_stride, 0, st);
        case WX_C128: return sw_batch_run<cplx>(b, true, q, nullptr, panel_stride, 0, st);
        default: return sw_batch_run<dual>(b, true, q, nullptr, panel_stride, 0, st);
    }
}

wx_status wx_sw_batch_rhs(wx_sw_batch* b, const void* q, void* rhs, size_t panel_stride, wx_region region,
                          wx_stream stream) {
    if (!b || !q || !rhs) return fail(WX_ERR_INVALID, "wx_sw_batch_rhs: null argument");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    WX_STREAM(st, stream);
    switch (b->dtype) {
        case WX_F64: return sw_batch_run<double>(b, false, q, rhs, panel_stride, region, st);
        case WX_C128: return sw_batch_run<cplx>(b, false, q, rhs, panel_stride, region, st);
        default: return sw_batch_run<dual>(b, false, q, rhs, panel_stride, region, st);
    }
}

// ---- the direct form: no interface buffer (one launch after the exchange of the tile-edge lines)
wx_status wx_sw_extrap_pack_ring(wx_sw_plan* pl, const void* q, void* const send[4], wx_stream stream) {
    if (!pl || !q) return fail(WX_ERR_INVALID, "wx_sw_extrap_pack_ring: null argument");
    WX_STREAM(st, stream);
    switch (pl->dtype) {
        case WX_F64: { SwParams<double> P = make_sw_params<double>(pl); set_edges<double>(P, send, nullptr);
                       SwDyn<double> D{static_cast<const double*>(q), nullptr, pl->H, 0, 0, nullptr, 0.0, 0.0, 1.0, 0, 0};
                       return sw_dispatch<double>(pl->n, 4, &P, D, nullptr, 0, 0, st); }
        case WX_C128: { SwParams<cplx> P = make_sw_params<cplx>(pl); set_edges<cplx>(P, send, nullptr);
                        SwDyn<cplx> D{static_cast<const cplx*>(q), nullptr, pl->H, 0, 0, nullptr, 0.0, 0.0, 1.0, 0, 0};
                        return sw_dispatch<cplx>(pl->n, 4, &P, D, nullptr, 0, 0, st); }
        default: { SwParams<dual> P = make_sw_params<dual>(pl); set_edges<dual>(P, send, nullptr);
                   SwDyn<dual> D{static_cast<const dual*>(q), nullptr, pl->H, 0, 0, nullptr, 0.0, 0.0, 1.0, 0, 0};
                   return sw_dispatch<dual>(pl->n, 4, &P, D, nullptr, 0, 0, st); }
    }
}

wx_status wx_sw_rhs_direct(wx_sw_plan* pl, const void* q, const void* const halo[4], const void* y, void* out, double a,
                           double b, double c, int axpy, wx_region region, wx_stream stream) {
    wx_status ok = sw_check_rhs_args(pl, q, out, halo, region);
    if (ok != WX_OK) return ok;
    WX_STREAM(st, stream);
    switch (pl->dtype) {
        case WX_F64: return sw_run_direct<double>(pl, q, halo, out, region, axpy, y, a, b, c, st);
        case WX_C128: return sw_run_direct<cplx>(pl, q, halo, out, region, axpy, y, a, b, c, st);
        default: return sw_run_direct<dual>(pl, q, halo, out, region, axpy, y, a, b, c, st);
    }
}

wx_status wx_sw_batch_extrap_pack_ring(wx_sw_batch* b, const void* q, size_t panel_stride, wx_stream stream) {
    if (!b || !q) return fail(WX_ERR_INVALID, "wx_sw_batch_extrap_pack_ring: null argument");
    WX_STREAM(st, stream);
    switch (b->dtype) {
        case WX_F64: return sw_batch_direct<double>(b, 6, q, nullptr, panel_stride, 0, 0, nullptr, 0.0, 0.0, 1.0, st);
        case WX_C128: return sw_batch_direct<cplx>(b, 6, q, nullptr, panel_stride, 0, 0, nullptr, 0.0, 0.0, 1.0, st);
        default: return sw_batch_direct<dual>(b, 6, q, nullptr, panel_stride, 0, 0, nullptr, 0.0, 0.0, 1.0, st);
    }
}

wx_status wx_sw_batch_direct_pulls(const wx_sw_batch* b, int* pulls) {
    if (!b || !pulls) return fail(WX_ERR_INVALID, "wx_sw_batch_direct_pulls: null argument");
    *pulls = b->pulls ? 1 : 0;
    return WX_OK;
}

wx_status wx_sw_batch_rhs_direct(wx_sw_batch* bt, const void* q, const void* y, void* out, size_t panel_stride, double a,
                                 double b, double c, int axpy, wx_region region, wx_stream stream) {
    if (!bt || !q || !out) return fail(WX_ERR_INVALID, "wx_sw_batch_rhs_direct: null argument");
    if (out == q) return fail(WX_ERR_INVALID, "wx_sw_batch_rhs_direct: output must not alias the state");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    WX_STREAM(st, stream);
    switch (bt->dtype) {
        case WX_F64: return sw_batch_direct<double>(bt, 7, q, out, panel_stride, region, axpy, y, a, b, c, st);
        case WX_C128: return sw_batch_direct<cplx>(bt, 7, q, out, panel_stride, region, axpy, y, a, b, c, st);
        default: return sw_batch_direct<dual>(bt, 7, q, out, panel_stride, region, axpy, y, a, b, c, st);
    }
}

// ---- stage pipeline (the shallow-water twin of wx_euler3d_stage)
wx_status wx_sw_plan_reserve(wx_sw_plan* pl, int what) {
    if (!pl) return fail(WX_ERR_INVALID, "wx_sw_plan_reserve: null plan");
    if (what & ~WX_RESERVE_STAGE) return fail(WX_ERR_INVALID, "wx_sw_plan_reserve: unknown flags %d", what);
    if ((what & WX_RESERVE_STAGE) && !pl->itf2) {
        const size_t bytes = (size_t)pl->H * pl->H * 4 * 3 * pl->n * (pl->dtype == WX_F64 ? 8 : 16);
        hipError_t e = hipMalloc(&pl->itf2, bytes);
        if (e != hipSuccess) return fail(WX_ERR_NOMEM, "hipMalloc(%zu bytes) for the second interface buffer failed: %s", bytes,
                                         hipGetErrorString(e));
    }
    return WX_OK;
}

wx_status wx_sw_extrap_pack_slot(wx_sw_plan* pl, const void* q, void* const send[4], int slot, wx_stream stream) {
    if (!pl || !q) return fail(WX_ERR_INVALID, "wx_sw_extrap_pack_slot: null argument");
    if (slot != 0 && slot != 1) return fail(WX_ERR_INVALID, "interface slot %d not in {0,1}", slot);
    if (slot == 1 && !pl->itf2) return fail(WX_ERR_INVALID, "wx_sw_extrap_pack_slot: call wx_sw_plan_reserve(plan, WX_RESERVE_STAGE) first");
    WX_STREAM(st, stream);
    switch (pl->dtype) {
        case WX_F64: return sw_run<double>(pl, true, q, send, nullptr, nullptr, 0, 0, st, 0, nullptr, 0.0, 0.0, 1.0, slot);
        case WX_C128: return sw_run<cplx>(pl, true, q, send, nullptr, nullptr, 0, 0, st, 0, nullptr, 0.0, 0.0, 1.0, slot);
        default: return sw_run<dual>(pl, true, q, send, nullptr, nullptr, 0, 0, st, 0, nullptr, 0.0, 0.0, 1.0, slot);
    }
}

wx_status wx_sw_stage(wx_sw_plan* pl, const void* q, const void* const halo[4], const void* y, void* out, double a, double b,
                      double c, wx_region region, int itf_in, void* const next_send[4], int prepare_next, wx_stream stream) {
    wx_status ok = sw_check_rhs_args(pl, q, out, halo, region);
    if (ok != WX_OK) return ok;
    if (itf_in != 0 && itf_in != 1) return fail(WX_ERR_INVALID, "interface slot %d not in {0,1}", itf_in);
    if ((prepare_next || itf_in == 1) && !pl->itf2)
        return fail(WX_ERR_INVALID, "wx_sw_stage: call wx_sw_plan_reserve(plan, WX_RESERVE_STAGE) first");
    const int count = sw_region_count(region, pl->H);
    const int prep = prepare_next ? 1 : 0;
    WX_STREAM(st, stream);
    switch (pl->dtype) {
        case WX_F64: return sw_run<double>(pl, false, q, nullptr, halo, out, region, count, st, 1, y, a, b, c, itf_in, next_send, prep);
        case WX_C128: return sw_run<cplx>(pl, false, q, nullptr, halo, out, region, count, st, 1, y, a, b, c, itf_in, next_send, prep);
        default: return sw_run<dual>(pl, false, q, nullptr, halo, out, region, count, st, 1, y, a, b, c, itf_in, next_send, prep);
    }
}

wx_status wx_sw_batch_create_pipelined(wx_sw_batch** out, wx_sw_plan* const plans[], int count, void* const send[][4],
                                       const void* const halo[][4], void* const send2[][4], const void* const halo2[][4]) {
    if (!out || !plans || !send || !halo || !send2 || !halo2 || count < 1)
        return fail(WX_ERR_INVALID, "wx_sw_batch_create_pipelined: bad argument");
    *out = nullptr;
    for (int i = 0; i < count; ++i) {
        if (!plans[i]) return fail(WX_ERR_INVALID, "wx_sw_batch_create_pipelined: plans[%d] is null", i);
        if (plans[i]->n != plans[0]->n || plans[i]->H != plans[0]->H || plans[i]->dtype != plans[0]->dtype)
            return fail(WX_ERR_INVALID, "wx_sw_batch_create_pipelined: plans differ in n, H or dtype");
        if (!plans[i]->itf2) return fail(WX_ERR_INVALID, "wx_sw_batch_create_pipelined: call wx_sw_plan_reserve on every plan first");
        for (int e = 0; e < 4; ++e)
            if (!send[i][e] || !halo[i][e] || !send2[i][e] || !halo2[i][e])
                return fail(WX_ERR_INVALID, "wx_sw_batch_create_pipelined: null edge buffer");
    }
    wx_sw_batch* b = new (std::nothrow) wx_sw_batch();
    if (!b) return fail(WX_ERR_NOMEM, "out of host memory");
    b->n = plans[0]->n; b->H = plans[0]->H; b->count = count; b->dtype = plans[0]->dtype; b->pipelined = true;
    wx_status st;
    switch (b->dtype) {
        case WX_F64: st = sw_batch_build<double>(b, plans, count, send, halo, send2, halo2); break;
        case WX_C128: st = sw_batch_build<cplx>(b, plans, count, send, halo, send2, halo2); break;
        default: st = sw_batch_build<dual>(b, plans, count, send, halo, send2, halo2);
    }
    if (st != WX_OK) {
        if (b->table) (void)hipFree(b->table);
        if (b->pull_tab) (void)hipFree(b->pull_tab);
        delete b;
        return st;
    }
    *out = b;
    return WX_OK;
}

wx_status wx_sw_batch_extrap_pack_slot(wx_sw_batch* b, const void* q, size_t panel_stride, int slot, wx_stream stream) {
    if (!b || !q) return fail(WX_ERR_INVALID, "wx_sw_batch_extrap_pack_slot: null argument");
    if (slot != 0 && slot != 1) return fail(WX_ERR_INVALID, "interface slot %d not in {0,1}", slot);
    if (slot == 1 && !b->pipelined) return fail(WX_ERR_INVALID, "wx_sw_batch_extrap_pack_slot: the batch has one slot (wx_sw_batch_create_pipelined)");
    WX_STREAM(st, stream);
    switch (b->dtype) {
        case WX_F64: return sw_batch_run<double>(b, true, q, nullptr, panel_stride, 0, st, 0, nullptr, 0.0, 0.0, 1.0, slot);
        case WX_C128: return sw_batch_run<cplx>(b, true, q, nullptr, panel_stride, 0, st, 0, nullptr, 0.0, 0.0, 1.0, slot);
        default: return sw_batch_run<dual>(b, true, q, nullptr, panel_stride, 0, st, 0, nullptr, 0.0, 0.0, 1.0, slot);
    }
}

wx_status wx_sw_batch_stage(wx_sw_batch* bt, const void* q, const void* y, void* out, size_t panel_stride, double a, double b,
                            double c, wx_region region, int itf_in, int prepare_next, wx_stream stream) {
    if (!bt || !q || !out) return fail(WX_ERR_INVALID, "wx_sw_batch_stage: null argument");
    if (out == q) return fail(WX_ERR_INVALID, "wx_sw_batch_stage: output must not alias the state");
    if (region != WX_REGION_ALL && region != WX_REGION_INTERIOR && region != WX_REGION_BOUNDARY)
        return fail(WX_ERR_INVALID, "unknown region %d", (int)region);
    if (itf_in != 0 && itf_in != 1) return fail(WX_ERR_INVALID, "interface slot %d not in {0,1}", itf_in);
    if (!bt->pipelined) return fail(WX_ERR_INVALID, "wx_sw_batch_stage: the batch has one slot (wx_sw_batch_create_pipelined)");
    const int prep = prepare_next ? 1 : 0;
    WX_STREAM(st, stream);
    switch (bt->dtype) {
        case WX_F64: return sw_batch_run<double>(bt, false, q, out, panel_stride, region, st, 1, y, a, b, c, itf_in, prep);
        case WX_C128: return sw_batch_run<cplx>(bt, false, q, out, panel_stride, region, st, 1, y, a, b, c, itf_in, prep);
        default: return sw_batch_run<dual>(bt, false, q, out, panel_stride, region, st, 1, y, a, b, c, itf_in, prep);
    }
}

}  // extern "C"
