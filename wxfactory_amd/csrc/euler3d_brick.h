// The low-order form of the 3-D Euler evaluation (n = 2, 3, 4: the orders of every shipped .ini and of the reference's own
// RHS benchmark matrix, tests/rhs_benchmark/run.sh:67-71): ONE kernel per evaluation, no interface buffer.
//
// Why: at low order the interface buffer of the two-kernel form is the traffic.  An element of n^3 points has 6 n^2 face
// points: K1 writes 5 * 6 / n values per solution point (120 B/point at n = 2 beside 40 B/point of state), K2 reads them
// back twice (own and neighbour side), and both neighbours solve every Riemann problem.  Measured at n = 2 on the
// reference's benchmark size (profiles/r06_low_order_ceiling.txt): K1 50 us + K2 168 us, of which K2's loads and stores
// alone take 124 us - the evaluation is bound by the bytes of its own intermediate, not by its arithmetic.
//
// Here a workgroup owns a BRICK of elements (4 x 4 x 2 at n = 2, 2 x 2 x 2 at n = 3, 2 x 2 x 1 at n = 4):
//   * the state of the brick goes to LDS once (log rho, rho u^i, log rho theta: rhs_dfr.py:50-71);
//   * the face states of both sides of every face INSIDE the brick are extrapolated from LDS, the Riemann problem of such a
//     face is solved ONCE and its seven results are handed to both elements (fluxes.py:326-403 writes the common flux
//     to both sides too);
//   * for a face on the brick's surface the outer state comes from the received halo (tile edge), from the wall rule
//     (ground / top, pde_euler_cubesphere.py:150-156), or is extrapolated HERE from the neighbour element's nodal values
//     read from Q - n values per face point and variable out of the L2 / Infinity Cache instead of one out of an
//     interface buffer that a kernel before this one had to write;
//   * the rest (pointwise fluxes, forcing, the three directional passes, epilogue) is the fused kernel's (euler3d_rhs.h).
// The tile-edge messages come from a pack kernel over the ring of boundary elements only (euler3d_extrap.h, PACK_ONLY).
// Arithmetic is the two-kernel form's, operation for operation (same extrapolation sums, same Rusanov expressions): the two
// forms agree to rounding (the interface metric of a face inside the brick is read from the lower element's slot for both
// sides, where the two-kernel form reads each element's own copy of the same number).
#pragma once

// diagnostic builds of the one-kernel form (wrong results; profiles/r06_brick_breakdown.txt): 1 the states beyond the brick's
// surface are copies of the own state (no loads of neighbour elements, no logarithms of their values); 2 no Riemann arithmetic;
// 3 no face stage at all; 4 no Christoffel loads
#ifndef WX_BRICK_DIAG
#define WX_BRICK_DIAG 0
#endif

namespace wx {

template <int N>
struct BrickCfg {
    static constexpr bool on = N <= 4;
    static constexpr int N2 = N * N, N3 = N * N * N;
    static constexpr int LOG_EPB = N == 2 ? 5 : (N == 3 ? 3 : (N == 4 ? 2 : 0));   // elements per brick: 32, 8, 4
    static constexpr int EPB = 1 << LOG_EPB;
    static constexpr int BS = ((EPB * N3 + 63) / 64) * 64;
    static constexpr int NP = Cfg<N>::NP;
    static constexpr int LE = N2 * NP;
    // waves per SIMD the register allocation aims at: n = 2 holds two workgroups per CU whatever it does (59 KB of LDS each)
#ifndef WX_BRICK_WAVES
#define WX_BRICK_WAVES 2
#endif
    static constexpr int WAVES = N == 2 ? 2 : WX_BRICK_WAVES;
    __host__ __device__ static constexpr int lidx(int kl, int jl, int il) { return (kl * N + jl) * NP + il; }
};

// The elements of one launch: up to four boxes of a tile (ALL: the tile; INTERIOR: the tile without its ring; BOUNDARY: the
// south and north rows, the west and east columns), every level, each box cut into bricks of its own shape
// (2^li x 2^lj x 2^lk elements, li + lj + lk = LOG_EPB: a one-element-wide strip takes long bricks).  Bricks start at the
// box origin; the last brick of a row / column / the top layer may be cut (its surplus threads idle).
struct BrickBoxes {
    int nbricks;           // bricks of one tile in this launch = plane * (bricks along k)
    int plane;             // in-plane bricks, all boxes
    unsigned md_plane;     // fast_div multiplier for `plane`
    int lk;                // log2 of the brick's vertical extent (the same for every box)
    int start1, start2, start3;   // first in-plane brick of boxes 1, 2, 3 (INT_MAX: no such box)
    int i0[4], j0[4], i1[4], j1[4];   // the boxes' element ranges [i0, i1) x [j0, j1)
    int li[4], lj[4], nbi[4];         // brick shape and bricks per row
    unsigned mdi[4];                  // fast_div multiplier for nbi
};
// (constant indices only: a run-time index into a by-value kernel argument sends the whole struct to scratch)
#define WX_BOX(G, field, box) ((box) == 0 ? (G).field[0] : ((box) == 1 ? (G).field[1] : ((box) == 2 ? (G).field[2] : (G).field[3])))

// the logarithms of this file (float64): the lean form of wx_math.h - the face stage is bound by its instruction count, and the
// library's log is 84 of them against 38.  The pack kernel of the one-kernel form takes the same function (euler3d_extrap.h,
// PACK), so a tile-edge state is the same number whether a brick extrapolates it or the neighbour tile packs it.
__device__ __forceinline__ double b_log(double x) { return lean_log(x); }
__device__ __forceinline__ dual b_log(dual x) { return lean_log(x); }

template <typename X>
__device__ __forceinline__ X* raw_ptr(X* p) { return p; }
template <typename X>
__device__ __forceinline__ X* raw_ptr(gp<X> p) { return p.raw(); }

// x / d for the small numbers of a brick's face decode (x < 2048, 1 <= d <= 33): (x * kSmallDiv[d]) >> 16, kSmallDiv[d] = 65536 / d + 1
__constant__ unsigned kSmallDiv[34] = {0, 65537, 32769, 21846, 16385, 13108, 10923, 9363, 8193, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4097,
                                       3856, 3641, 3450, 3277, 3121, 2979, 2850, 2731, 2622, 2521, 2428, 2341, 2260, 2185, 2115, 2049, 1986};
__device__ __forceinline__ int small_div(int x, unsigned m) { return (int)(((unsigned)x * m) >> 16); }

struct BrickAt {
    int i0, j0, k0, li, lj, lk;
    int vi, vj, vk;   // extent of the brick's elements that exist (the last brick of a row / column / the top layer is cut)
    bool any;
};

__device__ __forceinline__ BrickAt brick_at(const BrickBoxes& G, int L, int V) {
    BrickAt b;
    b.any = L < G.nbricks;
    if (!b.any) L = 0;
    const int kb = fast_div(L, G.plane, G.md_plane);
    const int bp = L - kb * G.plane;
    const int box = (bp >= G.start1) + (bp >= G.start2) + (bp >= G.start3);
    const int r = bp - (box == 0 ? 0 : (box == 1 ? G.start1 : (box == 2 ? G.start2 : G.start3)));
    const int nbi = WX_BOX(G, nbi, box);
    const int bj = fast_div(r, nbi, WX_BOX(G, mdi, box)), bi = r - bj * nbi;
    b.li = WX_BOX(G, li, box); b.lj = WX_BOX(G, lj, box); b.lk = G.lk;
    b.i0 = WX_BOX(G, i0, box) + (bi << b.li); b.j0 = WX_BOX(G, j0, box) + (bj << b.lj); b.k0 = kb << b.lk;
    const int ri = WX_BOX(G, i1, box) - b.i0, rj = WX_BOX(G, j1, box) - b.j0, rk = V - b.k0;
    b.vi = ri < (1 << b.li) ? ri : (1 << b.li);
    b.vj = rj < (1 << b.lj) ? rj : (1 << b.lj);
    b.vk = rk < (1 << b.lk) ? rk : (1 << b.lk);
    return b;
}

struct BElem {
    int ei, ej, ek, e;
    bool valid;
};

// the element at position (lbi, lbj, lbk) of the brick; le = its slot in the workgroup's LDS images
__device__ __forceinline__ int brick_slot(const BrickAt& b, int lbi, int lbj, int lbk) { return lbi + (lbj << b.li) + (lbk << (b.li + b.lj)); }

template <int EPB>
__device__ __forceinline__ BElem brick_elem(const BrickAt& b, int le, int H) {
    BElem r;
    const int lbi = le & ((1 << b.li) - 1), lbj = (le >> b.li) & ((1 << b.lj) - 1), lbk = le >> (b.li + b.lj);
    r.ei = b.i0 + lbi; r.ej = b.j0 + lbj; r.ek = b.k0 + lbk;
    r.valid = b.any && le < EPB && lbi < b.vi && lbj < b.vj && lbk < b.vk;
    if (!r.valid) { r.ei = b.i0; r.ej = b.j0; r.ek = b.k0; }   // (an addressable element: the brick's first)
    r.e = (r.ek * H + r.ej) * H + r.ei;
    return r;
}

// both sides of one Riemann problem (rusanov_face from the left element's point of view + the two right-side values):
// qL, qR [7]: the five face values in, pressure and log pressure filled here; lgL, lgR = log(rho theta) of the two states - the
// sum the extrapolation exponentiated (pde_euler_cubesphere.py:158-160 takes the logarithm of that exponential again: the
// same number to rounding, two logarithms per face point less).  wall: 0 none, 1 the RIGHT state is the mirror image of the left
// one (top of the model: own = left), 2 the LEFT state mirrors the right one (ground)
template <typename T>
__device__ __forceinline__ void rusanov_both(T* qL, T* qR, T lgL, T lgR, int d, int wall, double sg, double h0, double h1, double h2,
                                             bool advection_only, T* outL, T& bR, T& lpR) {
    const T gL = kGamma * (lgL + kLogRdOverP0), gR = kGamma * (lgR + kLogRdOverP0);
    qL[5] = kP0 * w_exp(gL); qR[5] = kP0 * w_exp(gR);
    qL[6] = kLogP0 + gL; qR[6] = kLogP0 + gR;
    const double hdd = d == 0 ? h0 : (d == 1 ? h1 : h2);
    const T rL = 1.0 / qL[0], rR = 1.0 / qR[0];
    T uL = w_sel(d == 0, qL[1], w_sel(d == 1, qL[2], qL[3])) * rL;
    T uR = w_sel(d == 0, qR[1], w_sel(d == 1, qR[2], qR[3])) * rR;
    if (wall == 1) uR = -uL;   // no-flow wall: odd symmetry of w (pde_euler_cubesphere.py:150-156)
    if (wall == 2) uL = -uR;
    rusanov_face<T>(qL, qR, uL, uR, rL, rR, sg, h0, h1, h2, hdd, true, advection_only, outL);
    const double sgh2 = sg * h2;
    bR = 0.5 * (sgh2 * qL[5] + sgh2 * qR[5]) / qR[5];
    lpR = qR[6];
}

// The pointers a face item selects between (four halos, six interface-metric arrays), copied OUT of the parameter block into
// scalars before any select: `d == 0 ? P.sgi : P.sgj` on a block that lives in memory (the batched launch's copy, or any block
// a by-reference lambda capture has pinned there) becomes a load through a selected ADDRESS and drags the block into scratch.
// No lambdas and no pointer aggregates in this file for that reason.
// interface metric of face point fp of face (d, plus) of element (ek, ej, ei): the element's own slot (face_load)
template <int N, typename T, bool G>
__device__ __forceinline__ void brick_face_metric(const double* sgi, const double* sgj, const double* sgk, const double* hi,
                                                  const double* hj, const double* hk, int H, int V, int ek, int ej, int ei, int d, int pl,
                                                  int fp, double& sg, double& h0, double& h1, double& h2) {
    constexpr int N2 = N * N;
    size_t o, hfs;
    if (d == 0) {
        o = (((size_t)ek * H + ej) * (H + 2) + ei + 1) * 2 * N2 + pl * N2 + fp;
        hfs = (size_t)V * H * (H + 2) * 2 * N2;
    } else if (d == 1) {
        o = (((size_t)ek * (H + 2) + ej + 1) * H + ei) * 2 * N2 + pl * N2 + fp;
        hfs = (size_t)V * (H + 2) * H * 2 * N2;
    } else {
        o = ((((size_t)ek + 1) * H + ej) * H + ei) * 2 * N2 + pl * N2 + fp;
        hfs = (size_t)(V + 2) * H * H * 2 * N2;
    }
    const pp<T, const double, G> sgp(d == 0 ? sgi : (d == 1 ? sgj : sgk));
    const pp<T, const double, G> hp = pp<T, const double, G>(d == 0 ? hi : (d == 1 ? hj : hk)) + ((size_t)d * 3 * hfs + o);
    sg = sgp[o]; h0 = hp[0]; h1 = hp[hfs]; h2 = hp[2 * hfs];
}

// face state of element slot `le` from the LDS images: s[0..4] the five values, lg = log(rho theta) (the sum before its exponential)
template <int N, typename T, typename W>
__device__ __forceinline__ void brick_extrap_lds(const T* img, int img_stride, int off, int lstride, W wv, T* s, T& lg) {
#pragma unroll
    for (int v = 0; v < 5; ++v) s[v] = T(0.0);
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const double wm = wv[m];
#pragma unroll
        for (int v = 0; v < 5; ++v) s[v] += wm * img[v * img_stride + off + m * lstride];
    }
    lg = s[4];
    s[0] = w_exp(s[0]);
    s[4] = w_exp(s[4]);
}

// nodal line of face point fp along d: LDS offset and stride, point index and stride
template <int N>
__device__ __forceinline__ void brick_line(int d, int fp, int& lbase, int& lstride, int& pbase, int& pstride) {
    using C = BrickCfg<N>;
    const int a = fp / N, b = fp - a * N;
    if (d == 0) { lbase = C::lidx(a, b, 0); lstride = 1; pbase = (a * N + b) * N; pstride = 1; }
    else if (d == 1) { lbase = C::lidx(a, 0, b); lstride = C::NP; pbase = a * C::N2 + b; pstride = N; }
    else { lbase = C::lidx(0, a, b); lstride = N * C::NP; pbase = a * N + b; pstride = C::N2; }
}

// one face point on the brick's surface
// kind: 0 a neighbour element of this tile, 1 halo (the received message), 2 wall, 3 a tile of the same batch across the tile edge
// (its nodal line: element `pes` of tile `pt2`, face point `pfp` of its edge `pe2` - see EulerParams::pull_tile)
struct Surf { int d, plus, le, fp, ei, ej, ek, kind, pt2, pe2, pes, pfp; };

// the other tiles of a batched launch, for kind 3 (null table: a single-tile launch, or no pulls)
// (the four neighbour tiles packed into ONE word, a byte per edge S, N, W, E: tile + 1 in bits 0-4, its edge in bits 5-6, its flip
// in bit 7 - picked out by a shift.  Not four members picked out by a select: a select between members of a block that lives in
// memory becomes a load through a selected address and pins the block in scratch, see the note above brick_face_metric.)
struct BrickBatchCtx {
    const void* table;  // the launch's parameter table, or null: no pulls
    long long stride;   // doubles between consecutive tiles' states
    int self;
    unsigned packed;
};
template <typename T>
__device__ __forceinline__ unsigned brick_pack_pulls(const EulerParams<T>* me) {
    unsigned w = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        w |= ((unsigned)((me->pull_tile[e] + 1) & 31) | ((unsigned)(me->pull_edge[e] & 3) << 5) | ((unsigned)(me->pull_flip[e] & 1) << 7)) << (8 * e);
    return w;
}

template <int N, typename T, bool G>
__device__ __forceinline__ Surf surf_decode(const EulerParams<T, G>& P, const BrickBatchCtx& ctx, const BrickAt& bk, int w, int s0, int s1,
                                            int s2, unsigned m_vi, unsigned m_vj, int H, int V) {
    constexpr int N2 = N * N;
    const int vi = bk.vi, vj = bk.vj, vk = bk.vk;
    Surf u;
    int x = w;
    if (x < 2 * s0) { u.d = 0; u.plus = x < s0; if (!u.plus) x -= s0; }
    else if (x < 2 * (s0 + s1)) { u.d = 1; x -= 2 * s0; u.plus = x < s1; if (!u.plus) x -= s1; }
    else { u.d = 2; x -= 2 * (s0 + s1); u.plus = x < s2; if (!u.plus) x -= s2; }
    const int s = x / N2;
    u.fp = x - s * N2;
    int lbi, lbj, lbk;
    if (u.d == 0) { lbk = small_div(s, m_vj); lbj = s - lbk * vj; lbi = u.plus ? vi - 1 : 0; }
    else if (u.d == 1) { lbk = small_div(s, m_vi); lbi = s - lbk * vi; lbj = u.plus ? vj - 1 : 0; }
    else { lbj = small_div(s, m_vi); lbi = s - lbj * vi; lbk = u.plus ? vk - 1 : 0; }
    u.le = brick_slot(bk, lbi, lbj, lbk);
    u.ei = bk.i0 + lbi; u.ej = bk.j0 + lbj; u.ek = bk.k0 + lbk;
    const int gd = u.d == 0 ? u.ei : (u.d == 1 ? u.ej : u.ek);
    const int ng = u.plus ? gd + 1 : gd - 1;
    const bool outside = ng < 0 || ng >= (u.d == 2 ? V : H);
    u.kind = (WX_BRICK_DIAG == 1) ? 2 : (outside ? (u.d == 2 ? 2 : 1) : 0);
    u.pt2 = -1; u.pe2 = 0; u.pes = 0; u.pfp = 0;
    if (u.kind == 1 && ctx.table != nullptr) {
        // the tile edge this face lies on, and who is behind it (constant indices: see WX_BOX)
        const unsigned inf = (ctx.packed >> (8 * ((u.d == 0 ? E_W : E_S) + u.plus))) & 0xffu;   // (E_S, E_N, E_W, E_E = 0, 1, 2, 3)
        const int t2 = (int)(inf & 31u) - 1;
        if (t2 >= 0) {
            const int e2 = (int)((inf >> 5) & 3u);
            const int fl = (int)(inf >> 7);
            // our slot of its message is (ek, along, a, b): it wrote that slot from its element along_s, face point (a, b_s)
            // (extrap_faces: al = flip ? H - 1 - along : along, bb = flip ? N - 1 - b : b)
            const int along = u.d == 0 ? u.ej : u.ei;
            const int a = u.fp / N, b = u.fp - a * N;
            const int along_s = fl ? H - 1 - along : along, b_s = fl ? N - 1 - b : b;
            const int ei_s = e2 == E_W ? 0 : (e2 == E_E ? H - 1 : along_s);
            const int ej_s = e2 == E_S ? 0 : (e2 == E_N ? H - 1 : along_s);
            u.kind = 3; u.pt2 = t2; u.pe2 = e2;
            u.pes = (u.ek * H + ej_s) * H + ei_s;
            u.pfp = a * N + b_s;
        }
    }
    return u;
}

// the loads of a surface item: the outer element's nodal line (kind 0: nv[m][0..4]) or the halo values (kind 1: hv[0..4]).
// (two sets of registers, each written on one path only: one set written on both paths meets in copies behind the branch, and
// a copy of a loaded value is a wait for it - the loads would not stay in flight)
template <int N, typename T, bool G>
__device__ __forceinline__ void surf_load(const EulerParams<T, G>& P, const BrickBatchCtx& ctx, const Surf& u, const T* halo_s,
                                          const T* halo_n, const T* halo_w, const T* halo_e, T (*nv)[5], T* hv, double& px) {
    constexpr int N2 = N * N, N3 = N2 * N;
    const int H = P.H, V = P.V;
    int lbase, lstride, pbase, pstride;
    brick_line<N>(u.d, u.fp, lbase, lstride, pbase, pstride);
    if (u.kind == 0 || u.kind == 3) {
        // the nodal line of the element behind the face: of this tile, or (kind 3) of the tile across the tile edge - the
        // same loads at another tile's distance (load_state adds the offset to every array of the state: q, and the shift's v)
        const size_t fs = (size_t)P.nelem * N3;
        size_t eo;
        int ps = pstride;
        if (u.kind == 0) {
            const int e = (u.ek * H + u.ej) * H + u.ei;
            eo = (size_t)(e + (u.plus ? 1 : -1) * (u.d == 0 ? 1 : (u.d == 1 ? H : H * H))) * N3 + pbase;
        } else {
            int lb2, ls2, pb2, ps2;
            brick_line<N>(u.pe2 >= E_W ? 0 : 1, u.pfp, lb2, ls2, pb2, ps2);
            eo = (size_t)((long long)(u.pt2 - ctx.self) * ctx.stride + (long long)u.pes * N3 + pb2);
            ps = ps2;
            // the sender's edge coordinate of this line (its boundary_we / boundary_sn, reached through this plan's constants)
            const int edge = (u.d == 0 ? E_W : E_S) + u.plus;
            const int along_s = u.pe2 >= E_W ? (u.pes / H) % H : u.pes % H;
            px = P.K->pull_x[edge][along_s * N + u.pfp % N];
        }
#pragma unroll
        for (int m = 0; m < N; ++m) load_state<T>(P, eo + (size_t)(m * ps), fs, nv[m][0], nv[m][1], nv[m][2], nv[m][3], nv[m][4]);
    } else if (u.kind == 1) {
        // lateral tile edge: the received message (process_topology.py:595-606), five planes V H n^2 apart
        const size_t vsh = (size_t)V * H * N2;
        pp<T, const T, G> hb(u.d == 0 ? (u.plus ? halo_e : halo_w) : (u.plus ? halo_n : halo_s));
        hb = hb + ((size_t)u.ek * H + (u.d == 0 ? u.ej : u.ei)) * N2 + u.fp;
#pragma unroll
        for (int v = 0; v < 5; ++v) hv[v] = hb[v * vsh];
    }
}

// The face stage of a brick.  img: the five LDS images of the brick's state (log rho, rho u1, rho u2, rho w, log rho theta),
// element slot `le` at le * LE.  store(le, f, fp, out[0..4], B, log p) receives the seven results of face f of element le.
// Two classes of work items, each a wave-uniform way to the two face states:
//   1. faces between two elements of the brick: both states from LDS, ONE Riemann problem, results to both elements;
//   2. faces on the brick's surface: the own state from LDS, the outer one from the neighbour element's nodal values (read
//      from Q and extrapolated here as that element's own evaluation does), from the received halo (tile edge) or from the wall
//      rule (ground / top).  The loads of a thread's first surface item are issued BEFORE it works through class 1.
// Both classes meet in ONE copy of the Riemann arithmetic: a face gives the same bits whichever class a launch's bricks put it
// in (INTERIOR + BOUNDARY == ALL bit for bit).
// rot: the wave that starts class 1 (the classes rarely fill whole rounds: the surplus rotates over the SIMDs from brick to brick).
template <int N, typename T, bool G, typename Store>
__device__ __forceinline__ void brick_face_stage(const EulerParams<T, G>& P, const BrickBatchCtx& ctx, const BrickAt& bk, const T* img,
                                                 int img_stride, int rot, Store store) {
    using C = BrickCfg<N>;
    constexpr int N2 = C::N2, BS = C::BS;
    const int tid = threadIdx.x;
    const int H = P.H, V = P.V;
    const int vi = bk.vi, vj = bk.vj, vk = bk.vk;
    const unsigned m_vi = kSmallDiv[vi], m_vj = kSmallDiv[vj], m_vi1 = kSmallDiv[vi > 1 ? vi - 1 : 1], m_vj1 = kSmallDiv[vj > 1 ? vj - 1 : 1];
    // class 1: per direction the lower element of each inner face, positions (lbi, lbj, lbk) with lb_d < v_d - 1
    const int c0 = (vi - 1) * vj * vk * N2, c1 = vi * (vj - 1) * vk * N2, c2 = vi * vj * (vk - 1) * N2;
    // class 2: per direction the plus faces of the last layer, then the minus faces of the first
    const int s0 = vj * vk * N2, s1 = vi * vk * N2, s2 = vi * vj * N2;
    const int nI = c0 + c1 + c2, nS = 2 * (s0 + s1 + s2);
    const T* const halo_e = raw_ptr(P.halo_e); const T* const halo_w = raw_ptr(P.halo_w);
    const T* const halo_n = raw_ptr(P.halo_n); const T* const halo_s = raw_ptr(P.halo_s);
    const double* const sgi = raw_ptr(P.sgi); const double* const sgj = raw_ptr(P.sgj); const double* const sgk = raw_ptr(P.sgk);
    const double* const hi = raw_ptr(P.hi); const double* const hj = raw_ptr(P.hj); const double* const hk = raw_ptr(P.hk);

    // this thread's items: k1 of class 1 (w = t1 + it BS), then k2 of class 2 (w = tid + it BS)
    const int t1 = (tid + 64 * rot) & (BS - 1);
    const int k1 = t1 < nI ? (nI - t1 + BS - 1) / BS : 0;
    const int k2 = tid < nS ? (nS - tid + BS - 1) / BS : 0;

    // ---- the first surface item: decode and LOADS, now
    T nv[N][5], hv[5];
    double g0, g1, g2, g3, px;
    // (dual numbers: twice the registers per value - the first surface item's loads are issued where they are used)
    constexpr bool PREFETCH = std::is_same<T, double>::value;
    Surf u = surf_decode<N, T, G>(P, ctx, bk, tid < nS ? tid : 0, s0, s1, s2, m_vi, m_vj, H, V);
    if (PREFETCH && k2 > 0) {
        brick_face_metric<N, T, G>(sgi, sgj, sgk, hi, hj, hk, H, V, u.ek, u.ej, u.ei, u.d, u.plus, u.fp, g0, g1, g2, g3);
        surf_load<N, T>(P, ctx, u, halo_s, halo_n, halo_w, halo_e, nv, hv, px);
    }

    for (int it = 0; it < k1 + k2; ++it) {
        T qL[7], qR[7], lgL, lgR;
        double sg, h0, h1, h2;
        int d, fp, le, nle = 0, wall = 0;
        bool both, own_is_L = true;
        if (it < k1) {
            int x = t1 + it * BS;
            if (x < c0) d = 0;
            else if (x < c0 + c1) { d = 1; x -= c0; }
            else { d = 2; x -= c0 + c1; }
            const int s = x / N2;
            fp = x - s * N2;
            // the lower element: s = (lbk * B + lbj) * A + lbi over the box (A, B, .) = (vi, vj, vk) with side d one shorter
            const int A = d == 0 ? vi - 1 : vi, B = d == 1 ? vj - 1 : vj;
            const int r = small_div(s, d == 0 ? m_vi1 : m_vi), lbi = s - r * A;
            const int lbk = small_div(r, d == 1 ? m_vj1 : m_vj), lbj = r - lbk * B;
            le = brick_slot(bk, lbi, lbj, lbk);
            nle = le + (d == 0 ? 1 : (d == 1 ? (1 << bk.li) : (1 << (bk.li + bk.lj))));
            brick_face_metric<N, T, G>(sgi, sgj, sgk, hi, hj, hk, H, V, bk.k0 + lbk, bk.j0 + lbj, bk.i0 + lbi, d, 1, fp, sg, h0, h1, h2);
            int lbase, lstride, pbase, pstride;
            brick_line<N>(d, fp, lbase, lstride, pbase, pstride);
            brick_extrap_lds<N, T>(img, img_stride, le * C::LE + lbase, lstride, P.K->ep, qL, lgL);
            brick_extrap_lds<N, T>(img, img_stride, nle * C::LE + lbase, lstride, P.K->em, qR, lgR);
            both = true;
        } else {
            if (it > k1 || !PREFETCH) {   // (a later surface item of this thread: its loads here)
                u = surf_decode<N, T, G>(P, ctx, bk, tid + (it - k1) * BS, s0, s1, s2, m_vi, m_vj, H, V);
                brick_face_metric<N, T, G>(sgi, sgj, sgk, hi, hj, hk, H, V, u.ek, u.ej, u.ei, u.d, u.plus, u.fp, g0, g1, g2, g3);
                surf_load<N, T>(P, ctx, u, halo_s, halo_n, halo_w, halo_e, nv, hv, px);
            }
            d = u.d; fp = u.fp; le = u.le;
            sg = g0; h0 = g1; h1 = g2; h2 = g3;
            int lbase, lstride, pbase, pstride;
            brick_line<N>(d, fp, lbase, lstride, pbase, pstride);
            T so[5], sn[5], lgo, lgn;
            if (u.plus) brick_extrap_lds<N, T>(img, img_stride, le * C::LE + lbase, lstride, P.K->ep, so, lgo);
            else brick_extrap_lds<N, T>(img, img_stride, le * C::LE + lbase, lstride, P.K->em, so, lgo);
            if (u.kind == 0 || u.kind == 3) {
                // the neighbour's face state from its nodal values, as its own extrapolation forms it (its face towards this brick;
                // kind 3: the outward face of the tile across the tile edge, its edge pe2: N and E are plus faces)
                const bool nplus = u.kind == 3 ? (u.pe2 == E_N || u.pe2 == E_E) : !u.plus;
#pragma unroll
                for (int v = 0; v < 5; ++v) sn[v] = T(0.0);
#pragma unroll
                for (int m = 0; m < N; ++m) {
                    const double wm = nplus ? P.K->ep[m] : P.K->em[m];
                    sn[0] += wm * b_log(nv[m][0]);
                    sn[1] += wm * nv[m][1];
                    sn[2] += wm * nv[m][2];
                    sn[3] += wm * nv[m][3];
                    sn[4] += wm * b_log(nv[m][4]);
                }
                lgn = sn[4];
                sn[0] = w_exp(sn[0]);
                sn[4] = w_exp(sn[4]);
                if (u.kind == 3) {
                    // ... then what its pack does before sending (process_topology.py:322-386): the horizontal contravariant pair
                    // rotated into THIS panel's basis with ITS table and ITS edge coordinate (EulerConsts::prot, pull_x)
                    rotate_contra<T>(P.K->prot[(u.d == 0 ? E_W : E_S) + u.plus], px, sn[1], sn[2]);
                }
            } else if (u.kind == 1) {
#pragma unroll
                for (int v = 0; v < 5; ++v) sn[v] = hv[v];
                lgn = b_log(sn[4]);
            } else {
                wall = u.plus ? 1 : 2;
#pragma unroll
                for (int v = 0; v < 5; ++v) sn[v] = so[v];
                lgn = lgo;
            }
            // left = the lower element's plus-side state, right = the upper element's minus-side state
            own_is_L = u.plus != 0;
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                qL[v] = w_sel(own_is_L, so[v], sn[v]);
                qR[v] = w_sel(own_is_L, sn[v], so[v]);
            }
            lgL = w_sel(own_is_L, lgo, lgn);
            lgR = w_sel(own_is_L, lgn, lgo);
            both = false;
        }
        T out[7], bR, lpR;
        if (WX_BRICK_DIAG == 2) {
            T sum = T(sg + h0 + h1 + h2);
#pragma unroll
            for (int v = 0; v < 5; ++v) sum += qL[v] + qR[v];
#pragma unroll
            for (int c = 0; c < 7; ++c) out[c] = sum;
            bR = sum; lpR = sum;
        } else {
            rusanov_both<T>(qL, qR, lgL, lgR, d, wall, sg, h0, h1, h2, P.advection_only != 0, out, bR, lpR);
        }
        if (both) {
            store(le, 2 * d + 1, fp, out, out[5], out[6]);
            store(nle, 2 * d, fp, out, bR, lpR);
        } else if (own_is_L) {
            store(le, 2 * d + 1, fp, out, out[5], out[6]);
        } else {
            store(le, 2 * d, fp, out, bR, lpR);
        }
    }
}

// tile-edge messages of the stage's OUTPUT (the next stage's state) from its LDS images (log rho, momenta, log rho theta):
// what extrap_faces does for the outward faces of the tile edge, on a brick
template <int N, typename T, bool G>
__device__ __forceinline__ void brick_pack_edges(const EulerParams<T, G>& P, const BrickAt& bk, const T* img, int img_stride) {
    using C = BrickCfg<N>;
    constexpr int N2 = C::N2, EPB = C::EPB, BS = C::BS;
    const int H = P.H, V = P.V;
    for (int fi = threadIdx.x; fi < EPB * 4 * N2; fi += BS) {
        const int le = fi / (4 * N2);
        const int r = fi - le * (4 * N2);
        const int f = r / N2, fp = r - f * N2;
        const BElem el = brick_elem<EPB>(bk, le, H);
        if (!el.valid) continue;
        const int d = f >> 1, plus = f & 1;
        const int a = fp / N, b = fp - a * N;
        int edge = -1, along = 0;
        double X = 0.0;
        if (d == 0 && ((plus && el.ei == H - 1) || (!plus && el.ei == 0))) {
            edge = plus ? E_E : E_W; along = el.ej; X = P.bwe[el.ej * N + b];
        } else if (d == 1 && ((plus && el.ej == H - 1) || (!plus && el.ej == 0))) {
            edge = plus ? E_N : E_S; along = el.ei; X = P.bsn[el.ei * N + b];
        }
        pp<T, T, G> sendp = edge == E_S ? P.nsend_s : (edge == E_N ? P.nsend_n : (edge == E_W ? P.nsend_w : P.nsend_e));
        if (edge < 0 || sendp == nullptr) continue;
        int base, stride;
        if (d == 0) { base = C::lidx(a, b, 0); stride = 1; }
        else { base = C::lidx(a, 0, b); stride = C::NP; }
        const auto w = plus ? P.K->ep : P.K->em;
        T s[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) s[v] = T(0.0);
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const double wm = w[m];
#pragma unroll
            for (int v = 0; v < 5; ++v) s[v] += wm * img[v * img_stride + le * C::LE + base + m * stride];
        }
        s[0] = w_exp(s[0]);
        s[4] = w_exp(s[4]);
        rotate_contra<T>(P.K->rot[edge], X, s[1], s[2]);
        int al = along, bb = b;
        if (P.K->flip[edge]) { al = H - 1 - along; bb = N - 1 - b; }
        const size_t eo = ((size_t)el.ek * H + al) * N2 + a * N + bb;
        const size_t vs = (size_t)V * H * N2;
        pp<T, T, G> out = sendp + eo;
#pragma unroll
        for (int v = 0; v < 5; ++v) out[v * vs] = s[v];
    }
}

// where the fused kernel keeps the seven results of face f of element slot le (euler_rhs_body's face-flux image)
template <int N, typename T>
struct BrickFaceStore {
    T* frs;
    __device__ __forceinline__ void operator()(int le, int f, int fp, const T* out, T bq, T lp) const {
        constexpr int N2 = N * N;
        T* q = frs + (le * 6 + f) * (7 * N2) + fp;
#pragma unroll
        for (int c = 0; c < 5; ++c) q[c * N2] = out[c];
        q[5 * N2] = bq;
        q[6 * N2] = lp;
    }
};

// ------------------------------------------------------------------------------------------------
// the fused evaluation on a brick (float64).  EPI: the stage pipeline's epilogue (optional exponential filter + NaN flag on the
// output, the tile-edge messages of the output) - a separate instantiation, the plain kernel keeps its schedule.
// ------------------------------------------------------------------------------------------------
template <int N, typename T, bool EPI, bool G>
__device__ __forceinline__ void euler_brick_body(const EulerParams<T, G>& P, const BrickBoxes& GB, const BrickBatchCtx& ctx) {
    using C = BrickCfg<N>;
    static_assert(std::is_same<T, double>::value, "the brick form of the fused kernel: float64 (the dual form: euler3d_brick_jvp.h)");
    constexpr int N2 = C::N2, N3 = C::N3, EPB = C::EPB, BS = C::BS;
    constexpr int NF = 8;   // staged fields: 4 F rows, A, B (per direction) + log p + sqrtG*rho
    constexpr int NC = 7;   // face quantities, see rusanov_face
    constexpr int FST = NC * N2;
    __shared__ T smem[NF * EPB * C::LE + EPB * 6 * FST];
    T(*fld)[EPB * C::LE] = reinterpret_cast<T(*)[EPB * C::LE]>(smem);
    T* frs = smem + NF * EPB * C::LE;
#define WX_FR(le_, f_, c_, fp_) frs[((le_) * 6 + (f_)) * FST + (c_) * N2 + (fp_)]
    __shared__ double sD[N * N], sHF[N * N], sCm[N], sCp[N];
    __shared__ double sEF[EPI ? N * N : 1];

    const int tid = threadIdx.x;
    __builtin_assume(tid < BS);
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;
    const int brick_id = xcd_slab_block(blockIdx.x, gridDim.x >> 3);
    const BrickAt bk = brick_at(GB, brick_id, V);
    if (!bk.any) return;   // (uniform over the workgroup: the launch is padded to a multiple of eight workgroups)
    for (int i = tid; i < N * N; i += BS) {
        sD[i] = P.K->D[i];
        sHF[i] = P.K->HF[i];
        if (EPI) sEF[i] = P.K->EF[i];
    }
    if (tid < N) {
        sCm[tid] = P.K->cm[tid];
        sCp[tid] = P.K->cp[tid];
    }

    const int le = tid / N3, pt = tid - le * N3;
    const BElem el = brick_elem<EPB>(bk, le, H);
    const bool active = el.valid;
    const int kl = pt / N2, jl = (pt / N) % N, il = pt % N;
    const int lb = (le < EPB ? le : 0) * C::LE;
    const int lpt = lb + C::lidx(kl, jl, il);
    const size_t o = (size_t)el.e * N3 + pt;

    // ---- every load of the point stage goes out first, in the order of use: state, Christoffel symbols, the rest of the metric.
    // The face stage issues the loads of its surface items behind them and works on LDS meanwhile.
    PointIn<T> S;
    k2_point_loads<T, false>(P, active, o, fs, S, o, fs);
    double cg[27], idzv = 0.0, dco = 0.0, dur0 = 0.0, dur1 = 0.0, dur2 = 0.0;
    {
        const bool ld = active && WX_BRICK_DIAG != 4;
        // (two batches of unconditional loads: a condition per load is a branch and a full wait per load)
        if (ld && P.rot_zero) {   // non-rotating planet: the 9 rotation symbols are identically zero
#pragma unroll
            for (int i = 0; i < 27; ++i) cg[i] = (i % 9) < 3 ? 0.0 : ldm(P.chr + (size_t)i * fs + o);
        } else if (ld) {
#pragma unroll
            for (int i = 0; i < 27; ++i) cg[i] = ldm(P.chr + (size_t)i * fs + o);
        } else {
#pragma unroll
            for (int i = 0; i < 27; ++i) cg[i] = 0.0;
        }
        if (ld) idzv = ldm(P.idz + o);
        if (ld && P.has_damp) { dco = P.dcoef[o]; dur0 = P.duref[o]; dur1 = P.duref[fs + o]; dur2 = P.duref[2 * fs + o]; }
    }
    const T q0 = S.q0, q1 = S.q1, q2 = S.q2, q3 = S.q3, q4 = S.q4;
    const double sg = S.sg;
    const T lq4 = b_log(q4);
    if (le < EPB) {
        fld[0][lpt] = b_log(q0);
        fld[1][lpt] = q1;
        fld[2][lpt] = q2;
        fld[3][lpt] = q3;
        fld[4][lpt] = lq4;
    }
    __syncthreads();

    // (the values loaded for the forcing are first USED here: without this the compiler pulls the forcing's first operations -
    // 2 c, g / dz - up to the loads, and the wait for all of them in front of the barrier)
#pragma unroll
    for (int i = 0; i < 27; ++i) asm volatile("" : "+v"(cg[i]));
    asm volatile("" : "+v"(idzv));

    // ---- pointwise quantities (the logarithm of rho theta is the one the extrapolation took)
    const T rinv = 1.0 / q0;
    const T u1 = q1 * rinv, u2 = q2 * rinv, u3 = q3 * rinv;
    const T glog = kGamma * (lq4 + kLogRdOverP0);
    const T p = kP0 * w_exp(glog);
    const T logp = kLogP0 + glog;
    if (le < EPB) {
        fld[6][lpt] = logp;
        fld[7][lpt] = sg * q0;
    }

    // ---- forcing (k2_forcing's arithmetic on the values loaded above)
    T fc0 = T(0.0), fc1 = T(0.0), fc2 = T(0.0);
    double gcoef = 0.0;
    if (active) {
        T fc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double* c = cg + i * 9;
            const double c01 = c[0], c02 = c[1], c03 = c[2], c11 = c[3], c12 = c[4], c13 = c[5], c22 = c[6], c23 = c[7], c33 = c[8];
            fc[i] = 2.0 * q0 * (c01 * u1 + c02 * u2 + c03 * u3) + c11 * (q0 * u1 * u1 + S.h00 * p) +
                    2.0 * c12 * (q0 * u1 * u2 + S.h01 * p) + 2.0 * c13 * (q0 * u1 * u3 + S.h02 * p) +
                    c22 * (q0 * u2 * u2 + S.h11 * p) + 2.0 * c23 * (q0 * u2 * u3 + S.h12 * p) +
                    c33 * (q0 * u3 * u3 + S.h22 * p);
        }
        if (P.has_damp) {
            const T dw = dco * q0;
            fc[0] += dw * (u1 - dur0);
            fc[1] += dw * (u2 - dur1);
            fc[2] += dw * (u3 - dur2);
        }
        fc0 = fc[0]; fc1 = fc[1]; fc2 = fc[2];
        gcoef = idzv * kGravity;
    }

    // ---- face stage: every Riemann problem of the brick once
    if (WX_BRICK_DIAG != 3)
    brick_face_stage<N, T>(P, ctx, bk, &fld[0][0], EPB * C::LE, brick_id & (BS / 64 - 1), BrickFaceStore<N, T>{frs});

    T acc0 = T(0.0), acc1 = sg * fc0, acc2 = sg * fc1, acc4 = T(0.0), accw = sg * fc2;
    T hf = T(0.0);

    // ---- three directional passes (euler_rhs_body's vector-pipe form)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const T ud = w_sel(d == 0, u1, w_sel(d == 1, u2, u3));
        const double hd0 = d == 0 ? S.h00 : (d == 1 ? S.h01 : S.h02);
        const double hd1 = d == 0 ? S.h01 : (d == 1 ? S.h11 : S.h12);
        const double hd2 = d == 0 ? S.h02 : (d == 1 ? S.h12 : S.h22);
        const T sgu = sg * ud;
        const T Bd = T(sg * hd2);
        __syncthreads();  // the face stage's / the previous direction's reads of the images are done
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
        const T pB = p * Bd;
#pragma unroll 1
        for (int c0 = 0; c0 < 7; c0 += kFieldBatch) {
#pragma unroll
            for (int cc = 0; cc < kFieldBatch; ++cc) {
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
    if (EPI && P.efilter) {
        // the per-step exponential filter (operators.py:114-119, 257-261) on the stage's output while it is in registers
        T t0 = active ? sg * r0 : T(0.0), t1 = active ? sg * r1 : T(0.0), t2 = active ? sg * r2 : T(0.0),
          t3 = active ? sg * r3 : T(0.0), t4 = active ? sg * r4 : T(0.0);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
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
            (r0 != r0 || r1 != r1 || r2 != r2 || r3 != r3 || r4 != r4))
            *P.nan_flag = 1;   // many writers, one value: a plain store is as good as an atomic OR (never cleared here)
    }
    if (active) {
        store_r<T>(P, o, r0);
        store_r<T>(P, fs + o, r1);
        store_r<T>(P, 2 * fs + o, r2);
        store_r<T>(P, 3 * fs + o, r3);
        store_r<T>(P, 4 * fs + o, r4);
    }
    if (EPI) {
        // the output is the next stage's state: its tile-edge messages now, while it is in registers (no pack launch)
        const bool any_send = P.nsend_s != nullptr || P.nsend_n != nullptr || P.nsend_w != nullptr || P.nsend_e != nullptr;
        const bool at_edge = bk.i0 == 0 || bk.j0 == 0 || bk.i0 + (1 << bk.li) >= H || bk.j0 + (1 << bk.lj) >= H;
        if (any_send && at_edge) {   // (uniform over the workgroup)
            __syncthreads();
            if (le < EPB) {
                fld[0][lpt] = active ? b_log(r0) : T(0.0);
                fld[1][lpt] = r1;
                fld[2][lpt] = r2;
                fld[3][lpt] = r3;
                fld[4][lpt] = active ? b_log(r4) : T(0.0);
            }
            __syncthreads();
            brick_pack_edges<N, T>(P, bk, &fld[0][0], EPB * C::LE);
        }
    }
#undef WX_FR
}

template <int N, bool EPI>
__global__ __launch_bounds__(BrickCfg<N>::BS, BrickCfg<N>::WAVES) void euler_brick_kernel(const EulerParams<double> P, const BrickBoxes GB) {
    euler_brick_body<N, double, EPI>(P, GB, BrickBatchCtx{nullptr, 0, 0, 0u});
}

template <int N>
__global__ __launch_bounds__(BrickCfg<N>::BS, BrickCfg<N>::WAVES) void euler_brick_batch_kernel(const EulerParams<double>* table,
                                                                                     const EulerBatchDyn<double> dyn,
                                                                                     const BrickBoxes GB) {
    // the block comes out of memory: typed so that every access through its pointers is a global one (wx_common.h: gp)
    EulerParams<double, true> P = *reinterpret_cast<const EulerParams<double, true>*>(table + blockIdx.y);
    const size_t off = (size_t)blockIdx.y * dyn.stride;
    batch_state<double>(P, dyn);
    P.rhs = dyn.rhs ? dyn.rhs + off : (double*)nullptr;
    P.y = dyn.y ? dyn.y + off : (const double*)nullptr;
    P.z = dyn.z ? dyn.z + off : (const double*)nullptr;
    P.region = dyn.region; P.count = dyn.count;
    P.axpy = dyn.axpy; P.ca = dyn.ca; P.cb = dyn.cb; P.cc = dyn.cc; P.cd = dyn.cd;
    // (tile-edge states from the other tiles of this launch when the batch holds every tile's neighbours: EulerParams::pull_tile)
    const BrickBatchCtx ctx{dyn.pulls ? table : nullptr, (long long)dyn.stride, (int)blockIdx.y, brick_pack_pulls(table + blockIdx.y)};
    euler_brick_body<N, double, false>(P, GB, ctx);
}

}  // namespace wx
