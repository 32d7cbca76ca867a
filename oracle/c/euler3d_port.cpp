// CPU restatement ("port") of the 3-D Euler RHS on one cubed-sphere tile: sum-factorised, element-blocked C++ with
// OpenMP.  TEST INFRASTRUCTURE - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
// load or call this; the product (wxfactory_amd/) never does.  Parity status: PINNED - tests/test_oracle_c.py checks
// it against the golden vectors produced by running the reference itself (oracle/refharness/gen_golden.py).
//
// Follows, phase by phase (reference wx_factory/...):
//   wxo_euler3d_extrapolate   rhs/rhs_dfr.py:50-71 (log-space for rho and rho*theta)
//   wxo_euler3d_rhs           pointwise fluxes            pde/pde_euler_cubesphere.py:72-124
//                             interior derivatives        rhs/rhs_dfr.py:89-104 (Kronecker operators of
//                                                         geometry/operators.py:157-183 in sum-factorised form)
//                             halo-padded interfaces, BCs rhs/rhs_dfr.py:203-268, pde_euler_cubesphere.py:150-156
//                             Rusanov fluxes              pde/fluxes.py:150-222, 326-403, 507-582
//                             corrections + assembly      rhs/rhs_dfr.py:106-139
//                             forcing                     pde_euler_cubesphere.py:12-25, 203-290; init/dcmip.py:676-757
// The tile-edge pack (rotation, flip) stays in oracle/euler3d.py::pack_edges, which accepts these face arrays.
//
// Layouts as in the reference (geometry/cubed_sphere_3d.py:187-205): q, rhs (5, V, H, H, n^3), point
// (kl n + jl) n + il; faces (5, V, H, H, 2 n^2) per direction, [0, n^2) minus side, [n^2, 2 n^2) plus side;
// halos (5, V, H, n^2) per lateral edge; interface metric halo-padded along its direction.
//
// Scalar type: float64 (wxo_euler3d_*) or complex128 (wxo_euler3d_*_c: the state, faces, halos and the result are
// interleaved (re, im) pairs; the metric stays real) with NumPy's rules where complex numbers have no natural order -
// abs() is the real modulus, maximum() compares lexicographically (real part first) - so that Im R(Q + i eps v) / eps is
// the reference's complex-step Jacobian-vector product (solvers/matvec.py:56-61) term by term.
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstring>

#ifdef _OPENMP
#include <omp.h>
#endif

// glibc ships vector variants of exp / log (libmvec, <= 4 ulp) but announces them only under -ffast-math, which an oracle
// must not use.  Announcing them by hand lets `#pragma omp simd` loops call _ZGV*_exp / _ZGV*_log while every other
// floating-point rule stays strict.  (Round 3's port spent most of its time in ~2500 scalar libm calls per element and
// streamed 4.5 GB/s per two-thread process: not a fair CPU baseline.)
extern "C" {
__attribute__((simd("notinbranch"))) double exp(double) noexcept;
__attribute__((simd("notinbranch"))) double log(double) noexcept;
}

namespace {

constexpr double kGravity = 9.80616, kP0 = 100000.0, kRd = 287.05, kCpd = 1005.46;
constexpr double kCvd = kCpd - kRd, kGamma = kCpd / kCvd;
constexpr int MAXN = 8, MAXN2 = MAXN * MAXN, MAXN3 = MAXN * MAXN * MAXN;

using cplx = std::complex<double>;

inline double s_abs(double x) { return std::fabs(x); }
inline double s_abs(const cplx& x) { return std::abs(x); }   // numpy.abs: the modulus, a real number
inline bool s_isnan(double x) { return x != x; }
inline bool s_isnan(const cplx& x) { return x.real() != x.real() || x.imag() != x.imag(); }
inline bool s_greater(double a, double b) { return a > b; }
inline bool s_greater(const cplx& a, const cplx& b) {   // numpy's lexicographic order of complex numbers
    return a.real() > b.real() || (a.real() == b.real() && a.imag() > b.imag());
}

template <typename S>
inline S pressure_of(S rho_theta) { return kP0 * std::exp(kGamma * std::log(rho_theta * (kRd / kP0))); }

// what one side of a face contributes to its Riemann problem
template <typename S>
struct Side {
    S q[5], un, p;
    double sg, h[3];
};

// out: F* of the five rows, A* (advective rho-w flux), and 1/2 (P_L + P_R) before the division by the side's pressure
template <typename S>
inline void rusanov(const Side<S>& L, const Side<S>& R, int d, bool advection_only, S* fstar, S& astar, S& pavg) {
    S eL = S(s_abs(L.un)), eR = S(s_abs(R.un));
    if (!advection_only) {
        eL += std::sqrt(L.h[d] * kGamma * L.p / L.q[0]);
        eR += std::sqrt(R.h[d] * kGamma * R.p / R.q[0]);
    }
    const S eig = (s_greater(eL, eR) || s_isnan(eL)) ? eL : eR;   // numpy.maximum propagates NaN
    const int mom[3] = {1, 2, 3};
    S fL[5], fR[5];
    for (int v = 0; v < 5; ++v) {
        fL[v] = L.sg * L.un * L.q[v];
        fR[v] = R.sg * R.un * R.q[v];
    }
    const S aL = fL[3], aR = fR[3];
    for (int i = 0; i < 3; ++i) {
        fL[mom[i]] += L.sg * L.h[i] * L.p;
        fR[mom[i]] += R.sg * R.h[i] * R.p;
    }
    for (int v = 0; v < 5; ++v) fstar[v] = 0.5 * (fL[v] + fR[v] - eig * L.sg * (R.q[v] - L.q[v]));
    astar = 0.5 * (aL + aR - eig * L.sg * (R.q[3] - L.q[3]));
    pavg = 0.5 * (L.sg * L.h[2] * L.p + R.sg * R.h[2] * R.p);
}

struct Tile {
    int n, n2, n3, H, V;
    size_t fs;    // field stride of point arrays
    size_t ffs;   // field stride of face arrays
};



// q (5, V, H, H, n^3) -> itf_d (5, V, H, H, 2 n^2), d = i, j, k
template <typename S>
int extrapolate_impl(int n, int H, int V, const double* em, const double* ep, const S* q, S* itf_i, S* itf_j, S* itf_k,
                     int nthreads) {
    if (n < 2 || n > MAXN) return 1;
    const int n2 = n * n, n3 = n2 * n;
    const size_t nelem = (size_t)V * H * H, fs = nelem * n3, ffs = nelem * 2 * n2;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (long e = 0; e < (long)nelem; ++e) {
        S a[MAXN3];
        for (int v = 0; v < 5; ++v) {
            const S* src = q + v * fs + (size_t)e * n3;
            const bool lg = (v == 0 || v == 4);
            for (int p = 0; p < n3; ++p) a[p] = lg ? std::log(src[p]) : src[p];
            S* fi = itf_i + v * ffs + (size_t)e * 2 * n2;
            S* fj = itf_j + v * ffs + (size_t)e * 2 * n2;
            S* fk = itf_k + v * ffs + (size_t)e * 2 * n2;
            for (int x = 0; x < n; ++x)
                for (int y = 0; y < n; ++y) {
                    S mi = 0, pi = 0, mj = 0, pj = 0, mk = 0, pk = 0;
                    for (int m = 0; m < n; ++m) {
                        const S vi = a[(x * n + y) * n + m];   // (kl = x, jl = y, il = m)
                        const S vj = a[(x * n + m) * n + y];   // (kl = x, jl = m, il = y)
                        const S vk = a[(m * n + x) * n + y];   // (kl = m, jl = x, il = y)
                        mi += em[m] * vi; pi += ep[m] * vi;
                        mj += em[m] * vj; pj += ep[m] * vj;
                        mk += em[m] * vk; pk += ep[m] * vk;
                    }
                    const int fp = x * n + y;
                    fi[fp] = lg ? std::exp(mi) : mi; fi[n2 + fp] = lg ? std::exp(pi) : pi;
                    fj[fp] = lg ? std::exp(mj) : mj; fj[n2 + fp] = lg ? std::exp(pj) : pj;
                    fk[fp] = lg ? std::exp(mk) : mk; fk[n2 + fp] = lg ? std::exp(pk) : pk;
                }
        }
    }
    return 0;
}

template <typename S>
int rhs_impl(int n, int H, int V, int case_number, const double* D, const double* C, const double* HF, const S* q,
             const S* itf_i, const S* itf_j, const S* itf_k, const S* halo_s, const S* halo_n, const S* halo_w,
             const S* halo_e, const double* sg, const double* h, const double* chr, const double* idz, const double* sgi,
             const double* sgj, const double* sgk, const double* hi, const double* hj, const double* hk,
             const double* dcoef, const double* duref, S* rhs, int nthreads) {
    if (n < 2 || n > MAXN) return 1;
    const bool advection_only = case_number < 13;
    const bool damp = (case_number == 21 || case_number == 22) && dcoef && duref;
    Tile T{n, n * n, n * n * n, H, V, (size_t)V * H * H * n * n * n, (size_t)V * H * H * 2 * n * n};
    const int n2 = T.n2, n3 = T.n3;
    const size_t fs = T.fs, ffs = T.ffs;
    const size_t hs = (size_t)V * H * n2;   // variable stride of a halo face
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for collapse(2) schedule(static)
    for (int ek = 0; ek < V; ++ek)
        for (int ej = 0; ej < H; ++ej)
            for (int ei = 0; ei < H; ++ei) {
                const size_t e = ((size_t)ek * H + ej) * H + ei;
                // ---- common fluxes on the six faces: [face][quantity][face point]
                //      quantities 0..4 F*, 5 A*, 6 1/2 (P_L + P_R) / p_own, 7 log p_own
                S fr[6][8][MAXN2];
                for (int f = 0; f < 6; ++f) {
                    const int d = f >> 1, plus = f & 1;
                    const S* itf = d == 0 ? itf_i : (d == 1 ? itf_j : itf_k);
                    const double *sgp, *hp;
                    size_t hfs, oL, oR;   // interface metric: field stride, offsets of the L and R slots of this face
                    const int lo = plus ? 1 : 0;   // padded index of the element left of the face is e_d + lo
                    if (d == 0) {
                        hfs = (size_t)V * H * (H + 2) * 2 * n2;
                        const size_t row = ((size_t)ek * H + ej) * (H + 2);
                        oL = (row + ei + lo) * 2 * n2 + n2; oR = (row + ei + lo + 1) * 2 * n2;
                        sgp = sgi; hp = hi + 0 * 3 * hfs;
                    } else if (d == 1) {
                        hfs = (size_t)V * (H + 2) * H * 2 * n2;
                        oL = (((size_t)ek * (H + 2) + ej + lo) * H + ei) * 2 * n2 + n2;
                        oR = (((size_t)ek * (H + 2) + ej + lo + 1) * H + ei) * 2 * n2;
                        sgp = sgj; hp = hj + 1 * 3 * hfs;
                    } else {
                        hfs = (size_t)(V + 2) * H * H * 2 * n2;
                        oL = ((((size_t)ek + lo) * H + ej) * H + ei) * 2 * n2 + n2;
                        oR = ((((size_t)ek + lo + 1) * H + ej) * H + ei) * 2 * n2;
                        sgp = sgk; hp = hk + 2 * 3 * hfs;
                    }
                    // the neighbour across the face: an element of the tile, a received halo, or (vertical ends) the wall
                    const int c = d == 0 ? ei : (d == 1 ? ej : ek), cn = c + (plus ? 1 : -1), ext = d == 2 ? V : H;
                    const bool inside = cn >= 0 && cn < ext;
                    const size_t estr = d == 0 ? 1 : (d == 1 ? (size_t)H : (size_t)H * H);
                    const S* halo = nullptr;
                    size_t ho = 0;
                    if (!inside && d == 0) { halo = plus ? halo_e : halo_w; ho = ((size_t)ek * H + ej) * n2; }
                    if (!inside && d == 1) { halo = plus ? halo_n : halo_s; ho = ((size_t)ek * H + ei) * n2; }
                    for (int fp = 0; fp < n2; ++fp) {
                        Side<S> own, nbr;
                        for (int v = 0; v < 5; ++v) {
                            own.q[v] = itf[v * ffs + e * 2 * n2 + plus * n2 + fp];
                            if (inside) nbr.q[v] = itf[v * ffs + (plus ? e + estr : e - estr) * 2 * n2 + (1 - plus) * n2 + fp];
                            else if (halo) nbr.q[v] = halo[v * hs + ho + fp];
                            else nbr.q[v] = own.q[v];   // ground / top: the ghost slot copies the state (rhs_dfr.py:257-268)
                        }
                        own.un = own.q[1 + d] / own.q[0];
                        nbr.un = nbr.q[1 + d] / nbr.q[0];
                        if (!inside && d == 2) nbr.un = -own.un;   // no-flow wall: odd w (pde_euler_cubesphere.py:150-156)
                        own.p = pressure_of(own.q[4]);
                        nbr.p = pressure_of(nbr.q[4]);
                        Side<S>& L = plus ? own : nbr;
                        Side<S>& R = plus ? nbr : own;
                        L.sg = sgp[oL + fp]; R.sg = sgp[oR + fp];
                        for (int r = 0; r < 3; ++r) { L.h[r] = hp[r * hfs + oL + fp]; R.h[r] = hp[r * hfs + oR + fp]; }
                        S fstar[5], astar, pavg;
                        rusanov(L, R, d, advection_only, fstar, astar, pavg);
                        for (int v = 0; v < 5; ++v) fr[f][v][fp] = fstar[v];
                        fr[f][5][fp] = astar;
                        fr[f][6][fp] = pavg / own.p;
                        fr[f][7][fp] = std::log(own.p);
                    }
                }
                // ---- nodal quantities
                S qq[5][MAXN3], u[3][MAXN3], p[MAXN3], lp[MAXN3];
                double sgv[MAXN3], hv[9][MAXN3];
                for (int v = 0; v < 5; ++v)
                    for (int pt = 0; pt < n3; ++pt) qq[v][pt] = q[v * fs + e * n3 + pt];
                std::memcpy(sgv, sg + e * n3, sizeof(double) * n3);
                for (int r = 0; r < 9; ++r) std::memcpy(hv[r], h + r * fs + e * n3, sizeof(double) * n3);
                for (int pt = 0; pt < n3; ++pt) {
                    for (int i = 0; i < 3; ++i) u[i][pt] = qq[1 + i][pt] / qq[0][pt];
                    p[pt] = pressure_of(qq[4][pt]);
                    lp[pt] = std::log(p[pt]);
                }
                S tot[5][MAXN3], wtot[MAXN3];
                for (int pt = 0; pt < n3; ++pt) {
                    for (int v = 0; v < 5; ++v) tot[v][pt] = 0.0;
                    wtot[pt] = 0.0;
                }
                for (int d = 0; d < 3; ++d) {
                    // fields to differentiate along d: 0..4 F^d, 5 A^d, 6 B^d, 7 log p
                    S g[8][MAXN3];
                    for (int pt = 0; pt < n3; ++pt) {
                        const S sgu = sgv[pt] * u[d][pt];
                        for (int v = 0; v < 5; ++v) g[v][pt] = sgu * qq[v][pt];
                        g[5][pt] = g[3][pt];
                        for (int i = 0; i < 3; ++i) g[1 + i][pt] += sgv[pt] * hv[3 * d + i][pt] * p[pt];
                        g[6][pt] = sgv[pt] * hv[3 * d + 2][pt];
                        g[7][pt] = lp[pt];
                    }
                    const int stride = d == 0 ? 1 : (d == 1 ? n : n2);
                    for (int kl = 0; kl < n; ++kl)
                        for (int jl = 0; jl < n; ++jl)
                            for (int il = 0; il < n; ++il) {
                                const int pt = (kl * n + jl) * n + il;
                                const int ix = d == 0 ? il : (d == 1 ? jl : kl);
                                const int base = pt - ix * stride;
                                const int fp = d == 0 ? kl * n + jl : (d == 1 ? kl * n + il : jl * n + il);
                                const double cm = C[2 * ix], cp = C[2 * ix + 1];
                                S r[8];
                                for (int c = 0; c < 8; ++c) {
                                    S a = 0.0;
                                    for (int m = 0; m < n; ++m) a += D[ix * n + m] * g[c][base + m * stride];
                                    r[c] = a + cm * fr[2 * d][c][fp] + cp * fr[2 * d + 1][c][fp];
                                }
                                for (int v = 0; v < 5; ++v) tot[v][pt] += r[v];
                                // W^d = [A@D + A*@C] + p [B@D + B*@C] + p B [log p@D + log p^@C]  (rhs_dfr.py:113-136)
                                wtot[pt] += r[5] + p[pt] * r[6] + p[pt] * g[6][pt] * r[7];
                            }
                }
                // ---- assembly, forcing
                for (int kl = 0; kl < n; ++kl)
                    for (int jl = 0; jl < n; ++jl)
                        for (int il = 0; il < n; ++il) {
                            const int pt = (kl * n + jl) * n + il;
                            const size_t o = e * n3 + pt;
                            const double isg = 1.0 / sgv[pt];
                            const S rho = qq[0][pt];
                            S R[5];
                            for (int v = 0; v < 5; ++v) R[v] = -isg * tot[v][pt];
                            R[3] = -isg * wtot[pt];
                            S force[3];
                            const S u1 = u[0][pt], u2 = u[1][pt], u3 = u[2][pt], pp = p[pt];
                            for (int i = 0; i < 3; ++i) {
                                const double* c = chr + (size_t)(i * 9) * fs + o;
                                force[i] = 2.0 * rho * (c[0] * u1 + c[fs] * u2 + c[2 * fs] * u3) +
                                           c[3 * fs] * (rho * u1 * u1 + hv[0][pt] * pp) +
                                           2.0 * c[4 * fs] * (rho * u1 * u2 + hv[1][pt] * pp) +
                                           2.0 * c[5 * fs] * (rho * u1 * u3 + hv[2][pt] * pp) +
                                           c[6 * fs] * (rho * u2 * u2 + hv[4][pt] * pp) +
                                           2.0 * c[7 * fs] * (rho * u2 * u3 + hv[5][pt] * pp) +
                                           c[8 * fs] * (rho * u3 * u3 + hv[8][pt] * pp);
                            }
                            S hfv = 0.0;   // vertical high-filter of sqrtG rho (operators.py:75-80)
                            for (int m = 0; m < n; ++m) {
                                const int pm = (m * n + jl) * n + il;
                                hfv += HF[kl * n + m] * sgv[pm] * qq[0][pm];
                            }
                            force[2] += idz[o] * kGravity * isg * hfv;
                            if (damp) {
                                const S dw = dcoef[o] * rho;
                                force[0] += dw * (u1 - duref[o]);
                                force[1] += dw * (u2 - duref[fs + o]);
                                force[2] += dw * (u3 - duref[2 * fs + o]);
                            }
                            for (int i = 0; i < 3; ++i) R[1 + i] -= force[i];
                            for (int v = 0; v < 5; ++v) rhs[v * fs + o] = advection_only ? S(0.0) : R[v];
                        }
            }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// float64 fast path: the same phases written for the vector units - structure-of-arrays scratch per element, every loop
// over face points / nodes an `omp simd` loop with unit stride, the transcendentals through libmvec, the 1-D operators
// applied along contiguous lines.  Term for term the arithmetic of rhs_impl<double> / extrapolate_impl<double> above
// (which stay: the complex instantiation, and the cross-check of this path in tests/test_oracle_c.py).
// ------------------------------------------------------------------------------------------------------------------
template <int N>
int extrapolate_fast(int H, int V, const double* em, const double* ep, const double* q, double* itf_i, double* itf_j,
                     double* itf_k, int nthreads) {
    constexpr int N2 = N * N, N3 = N2 * N;
    const size_t nelem = (size_t)V * H * H, fs = nelem * N3, ffs = nelem * 2 * N2;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (long e = 0; e < (long)nelem; ++e) {
        alignas(64) double a[N3], face[6][N2];
        for (int v = 0; v < 5; ++v) {
            const double* src = q + v * fs + (size_t)e * N3;
            const bool lg = (v == 0 || v == 4);
            if (lg) {
#pragma omp simd
                for (int p = 0; p < N3; ++p) a[p] = log(src[p]);
            } else {
#pragma omp simd
                for (int p = 0; p < N3; ++p) a[p] = src[p];
            }
            for (int x = 0; x < N; ++x) {
                // i: the line runs along the contiguous index (a dot product per face point)
                for (int y = 0; y < N; ++y) {
                    double mi = 0, pi = 0;
                    for (int m = 0; m < N; ++m) { const double vi = a[(x * N + y) * N + m]; mi += em[m] * vi; pi += ep[m] * vi; }
                    face[0][x * N + y] = mi; face[1][x * N + y] = pi;
                }
                // j, k: the face index y is the contiguous one
                double mj[N], pj[N], mk[N], pk[N];
#pragma omp simd
                for (int y = 0; y < N; ++y) { mj[y] = pj[y] = mk[y] = pk[y] = 0.0; }
                for (int m = 0; m < N; ++m) {
#pragma omp simd
                    for (int y = 0; y < N; ++y) {
                        const double vj = a[(x * N + m) * N + y], vk = a[(m * N + x) * N + y];
                        mj[y] += em[m] * vj; pj[y] += ep[m] * vj;
                        mk[y] += em[m] * vk; pk[y] += ep[m] * vk;
                    }
                }
#pragma omp simd
                for (int y = 0; y < N; ++y) {
                    face[2][x * N + y] = mj[y]; face[3][x * N + y] = pj[y];
                    face[4][x * N + y] = mk[y]; face[5][x * N + y] = pk[y];
                }
            }
            if (lg)
                for (int f = 0; f < 6; ++f) {
#pragma omp simd
                    for (int fp = 0; fp < N2; ++fp) face[f][fp] = exp(face[f][fp]);
                }
            double* dst[3] = {itf_i + v * ffs + (size_t)e * 2 * N2, itf_j + v * ffs + (size_t)e * 2 * N2,
                              itf_k + v * ffs + (size_t)e * 2 * N2};
            for (int d = 0; d < 3; ++d) {
                std::memcpy(dst[d], face[2 * d], sizeof(double) * N2);
                std::memcpy(dst[d] + N2, face[2 * d + 1], sizeof(double) * N2);
            }
        }
    }
    return 0;
}

template <int N>
int rhs_fast(int H, int V, int case_number, const double* D, const double* C, const double* HF, const double* q,
             const double* itf_i, const double* itf_j, const double* itf_k, const double* halo_s, const double* halo_n,
             const double* halo_w, const double* halo_e, const double* sg, const double* h, const double* chr,
             const double* idz, const double* sgi, const double* sgj, const double* sgk, const double* hi, const double* hj,
             const double* hk, const double* dcoef, const double* duref, double* rhs, int nthreads) {
    constexpr int N2 = N * N, N3 = N2 * N;
    const bool advection_only = case_number < 13;
    const bool damp = (case_number == 21 || case_number == 22) && dcoef && duref;
    const size_t fs = (size_t)V * H * H * N3, ffs = (size_t)V * H * H * 2 * N2;
    const size_t hs = (size_t)V * H * N2;
    double Dt[N][N], cm[N], cp[N];   // Dt[m][i] = D[i][m]
    for (int i = 0; i < N; ++i) {
        cm[i] = C[2 * i]; cp[i] = C[2 * i + 1];
        for (int m = 0; m < N; ++m) Dt[m][i] = D[i * N + m];
    }
    const double kRdP0 = kRd / kP0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for collapse(2) schedule(static)
    for (int ek = 0; ek < V; ++ek)
        for (int ej = 0; ej < H; ++ej)
            for (int ei = 0; ei < H; ++ei) {
                const size_t e = ((size_t)ek * H + ej) * H + ei;
                // ---- common fluxes on the six faces: [face][quantity][face point] (quantities as in rhs_impl)
                alignas(64) double fr[6][8][N2];
                for (int f = 0; f < 6; ++f) {
                    const int d = f >> 1, plus = f & 1;
                    const double* itf = d == 0 ? itf_i : (d == 1 ? itf_j : itf_k);
                    const double *sgp, *hp;
                    size_t hfs, oL, oR;
                    const int lo = plus ? 1 : 0;
                    if (d == 0) {
                        hfs = (size_t)V * H * (H + 2) * 2 * N2;
                        const size_t row = ((size_t)ek * H + ej) * (H + 2);
                        oL = (row + ei + lo) * 2 * N2 + N2; oR = (row + ei + lo + 1) * 2 * N2;
                        sgp = sgi; hp = hi + 0 * 3 * hfs;
                    } else if (d == 1) {
                        hfs = (size_t)V * (H + 2) * H * 2 * N2;
                        oL = (((size_t)ek * (H + 2) + ej + lo) * H + ei) * 2 * N2 + N2;
                        oR = (((size_t)ek * (H + 2) + ej + lo + 1) * H + ei) * 2 * N2;
                        sgp = sgj; hp = hj + 1 * 3 * hfs;
                    } else {
                        hfs = (size_t)(V + 2) * H * H * 2 * N2;
                        oL = ((((size_t)ek + lo) * H + ej) * H + ei) * 2 * N2 + N2;
                        oR = ((((size_t)ek + lo + 1) * H + ej) * H + ei) * 2 * N2;
                        sgp = sgk; hp = hk + 2 * 3 * hfs;
                    }
                    const int c = d == 0 ? ei : (d == 1 ? ej : ek), cn = c + (plus ? 1 : -1), ext = d == 2 ? V : H;
                    const bool inside = cn >= 0 && cn < ext;
                    const size_t estr = d == 0 ? 1 : (d == 1 ? (size_t)H : (size_t)H * H);
                    const bool wall = !inside && d == 2;
                    // the two sides as (left, right) = (lower element's plus face, upper element's minus face)
                    alignas(64) double qo[5][N2], qn[5][N2];
                    for (int v = 0; v < 5; ++v) {
                        const double* own = itf + v * ffs + e * 2 * N2 + plus * N2;
                        const double* nbr;
                        if (inside) nbr = itf + v * ffs + (plus ? e + estr : e - estr) * 2 * N2 + (1 - plus) * N2;
                        else if (d == 0) nbr = (plus ? halo_e : halo_w) + v * hs + ((size_t)ek * H + ej) * N2;
                        else if (d == 1) nbr = (plus ? halo_n : halo_s) + v * hs + ((size_t)ek * H + ei) * N2;
                        else nbr = own;   // ground / top: the ghost slot copies the state (rhs_dfr.py:257-268)
                        std::memcpy(qo[v], own, sizeof(double) * N2);
                        std::memcpy(qn[v], nbr, sizeof(double) * N2);
                    }
                    const double* sgL = sgp + oL; const double* sgR = sgp + oR;
                    const double* hL[3] = {hp + oL, hp + hfs + oL, hp + 2 * hfs + oL};
                    const double* hR[3] = {hp + oR, hp + hfs + oR, hp + 2 * hfs + oR};
                    const double* qL[5]; const double* qR[5];
                    for (int v = 0; v < 5; ++v) { qL[v] = plus ? qo[v] : qn[v]; qR[v] = plus ? qn[v] : qo[v]; }
                    alignas(64) double pL[N2], pR[N2];
#pragma omp simd
                    for (int fp = 0; fp < N2; ++fp) {
                        pL[fp] = kP0 * exp(kGamma * log(qL[4][fp] * kRdP0));
                        pR[fp] = kP0 * exp(kGamma * log(qR[4][fp] * kRdP0));
                    }
                    const double* pown = plus ? pL : pR;
                    alignas(64) double lpown[N2];
#pragma omp simd
                    for (int fp = 0; fp < N2; ++fp) lpown[fp] = log(pown[fp]);
                    // (explicit scalars and restrict pointers: with the per-variable arrays of the generic form the
                    // vectoriser gives this loop up)
                    const double* __restrict l0 = qL[0]; const double* __restrict l1 = qL[1]; const double* __restrict l2 = qL[2];
                    const double* __restrict l3 = qL[3]; const double* __restrict l4 = qL[4];
                    const double* __restrict r0 = qR[0]; const double* __restrict r1 = qR[1]; const double* __restrict r2 = qR[2];
                    const double* __restrict r3 = qR[3]; const double* __restrict r4 = qR[4];
                    const double* __restrict lN = qL[1 + d]; const double* __restrict rN = qR[1 + d];
                    const double* __restrict hL0 = hL[0]; const double* __restrict hL1 = hL[1]; const double* __restrict hL2 = hL[2];
                    const double* __restrict hR0 = hR[0]; const double* __restrict hR1 = hR[1]; const double* __restrict hR2 = hR[2];
                    const double* __restrict hLd = hL[d]; const double* __restrict hRd = hR[d];
                    double* __restrict o0 = fr[f][0]; double* __restrict o1 = fr[f][1]; double* __restrict o2 = fr[f][2];
                    double* __restrict o3 = fr[f][3]; double* __restrict o4 = fr[f][4]; double* __restrict o5 = fr[f][5];
                    double* __restrict o6 = fr[f][6]; double* __restrict o7 = fr[f][7];
                    // no-flow wall: the ghost's normal velocity is minus the own one -  unX = kXX unL + kXY unR
                    const double kLL = (wall && !plus) ? 0.0 : 1.0, kLR = (wall && !plus) ? -1.0 : 0.0;
                    const double kRR = (wall && plus) ? 0.0 : 1.0, kRL = (wall && plus) ? -1.0 : 0.0;
                    alignas(64) double unLv[N2], unRv[N2], eLv[N2], eRv[N2];
#pragma omp simd
                    for (int fp = 0; fp < N2; ++fp) {
                        const double a = lN[fp] / l0[fp], b = rN[fp] / r0[fp];
                        unLv[fp] = kLL * a + kLR * b;
                        unRv[fp] = kRR * b + kRL * a;
                        eLv[fp] = std::fabs(unLv[fp]);
                        eRv[fp] = std::fabs(unRv[fp]);
                    }
                    if (!advection_only) {
#pragma omp simd
                        for (int fp = 0; fp < N2; ++fp) {
                            eLv[fp] += std::sqrt(hLd[fp] * kGamma * pL[fp] / l0[fp]);
                            eRv[fp] += std::sqrt(hRd[fp] * kGamma * pR[fp] / r0[fp]);
                        }
                    }
#pragma omp simd
                    for (int fp = 0; fp < N2; ++fp) {
                        const double unL = unLv[fp], unR = unRv[fp], eL = eLv[fp], eR = eRv[fp];
                        const double pl = pL[fp], pr = pR[fp];
                        const double eig = ((eL > eR) | (eL != eL)) ? eL : eR;   // numpy.maximum propagates NaN
                        const double sL = sgL[fp], sR = sgR[fp];
                        const double es = eig * sL, suL = sL * unL, suR = sR * unR;
                        const double a0 = suL * l0[fp], b0 = suR * r0[fp];
                        const double a1 = suL * l1[fp], b1 = suR * r1[fp];
                        const double a2 = suL * l2[fp], b2 = suR * r2[fp];
                        const double a3 = suL * l3[fp], b3 = suR * r3[fp];
                        const double a4 = suL * l4[fp], b4 = suR * r4[fp];
                        o0[fp] = 0.5 * (a0 + b0 - es * (r0[fp] - l0[fp]));
                        o1[fp] = 0.5 * ((a1 + sL * hL0[fp] * pl) + (b1 + sR * hR0[fp] * pr) - es * (r1[fp] - l1[fp]));
                        o2[fp] = 0.5 * ((a2 + sL * hL1[fp] * pl) + (b2 + sR * hR1[fp] * pr) - es * (r2[fp] - l2[fp]));
                        o3[fp] = 0.5 * ((a3 + sL * hL2[fp] * pl) + (b3 + sR * hR2[fp] * pr) - es * (r3[fp] - l3[fp]));
                        o4[fp] = 0.5 * (a4 + b4 - es * (r4[fp] - l4[fp]));
                        o5[fp] = 0.5 * (a3 + b3 - es * (r3[fp] - l3[fp]));
                        o6[fp] = 0.5 * (sL * hL2[fp] * pl + sR * hR2[fp] * pr) / pown[fp];
                        o7[fp] = lpown[fp];
                    }
                }
                // ---- nodal quantities
                alignas(64) double qq[5][N3], u[3][N3], p[N3], lp[N3], tot[5][N3], wtot[N3], g[8][N3], r[8][N3];
                const double* sgv = sg + e * N3;
                const double* hv[9];
                for (int rr = 0; rr < 9; ++rr) hv[rr] = h + rr * fs + e * N3;
                for (int v = 0; v < 5; ++v) std::memcpy(qq[v], q + v * fs + e * N3, sizeof(double) * N3);
#pragma omp simd
                for (int pt = 0; pt < N3; ++pt) {
                    const double ir = 1.0 / qq[0][pt];
                    u[0][pt] = qq[1][pt] / qq[0][pt]; u[1][pt] = qq[2][pt] / qq[0][pt]; u[2][pt] = qq[3][pt] / qq[0][pt];
                    (void)ir;
                    p[pt] = kP0 * exp(kGamma * log(qq[4][pt] * kRdP0));
                    for (int v = 0; v < 5; ++v) tot[v][pt] = 0.0;
                    wtot[pt] = 0.0;
                }
#pragma omp simd
                for (int pt = 0; pt < N3; ++pt) lp[pt] = log(p[pt]);
                for (int d = 0; d < 3; ++d) {
#pragma omp simd
                    for (int pt = 0; pt < N3; ++pt) {
                        const double sgu = sgv[pt] * u[d][pt];
                        for (int v = 0; v < 5; ++v) g[v][pt] = sgu * qq[v][pt];
                        g[5][pt] = g[3][pt];
                        for (int i = 0; i < 3; ++i) g[1 + i][pt] += sgv[pt] * hv[3 * d + i][pt] * p[pt];
                        g[6][pt] = sgv[pt] * hv[3 * d + 2][pt];
                        g[7][pt] = lp[pt];
                    }
                    // r[c] = g[c] @ D along d + cm F*_minus + cp F*_plus, every line of the element
                    for (int c = 0; c < 8; ++c) {
                        const double* gc = g[c];
                        double* rc = r[c];
                        const double* fm = fr[2 * d][c];
                        const double* fpl = fr[2 * d + 1][c];
                        if (d == 0) {
                            for (int row = 0; row < N2; ++row) {   // (kl, jl): the line is the contiguous run of il
                                double acc[N];
#pragma omp simd
                                for (int il = 0; il < N; ++il) acc[il] = cm[il] * fm[row] + cp[il] * fpl[row];
                                for (int m = 0; m < N; ++m) {
                                    const double gm = gc[row * N + m];
#pragma omp simd
                                    for (int il = 0; il < N; ++il) acc[il] += Dt[m][il] * gm;
                                }
#pragma omp simd
                                for (int il = 0; il < N; ++il) rc[row * N + il] = acc[il];
                            }
                        } else if (d == 1) {
                            for (int kl = 0; kl < N; ++kl)
                                for (int jl = 0; jl < N; ++jl) {
                                    double acc[N];
#pragma omp simd
                                    for (int il = 0; il < N; ++il) acc[il] = cm[jl] * fm[kl * N + il] + cp[jl] * fpl[kl * N + il];
                                    for (int m = 0; m < N; ++m) {
                                        const double w = Dt[m][jl];
#pragma omp simd
                                        for (int il = 0; il < N; ++il) acc[il] += w * gc[(kl * N + m) * N + il];
                                    }
#pragma omp simd
                                    for (int il = 0; il < N; ++il) rc[(kl * N + jl) * N + il] = acc[il];
                                }
                        } else {
                            for (int kl = 0; kl < N; ++kl) {
                                double acc[N2];
#pragma omp simd
                                for (int x = 0; x < N2; ++x) acc[x] = cm[kl] * fm[x] + cp[kl] * fpl[x];
                                for (int m = 0; m < N; ++m) {
                                    const double w = Dt[m][kl];
#pragma omp simd
                                    for (int x = 0; x < N2; ++x) acc[x] += w * gc[m * N2 + x];
                                }
#pragma omp simd
                                for (int x = 0; x < N2; ++x) rc[kl * N2 + x] = acc[x];
                            }
                        }
                    }
#pragma omp simd
                    for (int pt = 0; pt < N3; ++pt) {
                        for (int v = 0; v < 5; ++v) tot[v][pt] += r[v][pt];
                        // W^d = [A@D + A*@C] + p [B@D + B*@C] + p B [log p@D + log p^@C]  (rhs_dfr.py:113-136)
                        wtot[pt] += r[5][pt] + p[pt] * r[6][pt] + p[pt] * g[6][pt] * r[7][pt];
                    }
                }
                // ---- vertical high-filter of sqrtG rho (operators.py:75-80), then assembly and forcing
                alignas(64) double hfv[N3], sr[N3];
#pragma omp simd
                for (int pt = 0; pt < N3; ++pt) sr[pt] = sgv[pt] * qq[0][pt];
                for (int kl = 0; kl < N; ++kl) {
                    double acc[N2];
#pragma omp simd
                    for (int x = 0; x < N2; ++x) acc[x] = 0.0;
                    for (int m = 0; m < N; ++m) {
                        const double w = HF[kl * N + m];
#pragma omp simd
                        for (int x = 0; x < N2; ++x) acc[x] += w * sr[m * N2 + x];
                    }
#pragma omp simd
                    for (int x = 0; x < N2; ++x) hfv[kl * N2 + x] = acc[x];
                }
                const double* idzv = idz + e * N3;
                const double* __restrict h00 = hv[0]; const double* __restrict h01 = hv[1]; const double* __restrict h02 = hv[2];
                const double* __restrict h11 = hv[4]; const double* __restrict h12 = hv[5]; const double* __restrict h22 = hv[8];
                const double* __restrict rh = qq[0]; const double* __restrict pu1 = u[0]; const double* __restrict pu2 = u[1];
                const double* __restrict pu3 = u[2];
                alignas(64) double force[3][N3];
                for (int i = 0; i < 3; ++i) {   // one momentum row at a time: nine Christoffel streams per loop
                    const double* __restrict c0 = chr + (size_t)(i * 9 + 0) * fs + e * N3; const double* __restrict c1 = c0 + fs;
                    const double* __restrict c2 = c1 + fs; const double* __restrict c3 = c2 + fs; const double* __restrict c4 = c3 + fs;
                    const double* __restrict c5 = c4 + fs; const double* __restrict c6 = c5 + fs; const double* __restrict c7 = c6 + fs;
                    const double* __restrict c8 = c7 + fs;
                    double* __restrict fo = force[i];
#pragma omp simd
                    for (int pt = 0; pt < N3; ++pt) {
                        const double rho = rh[pt], u1 = pu1[pt], u2 = pu2[pt], u3 = pu3[pt], pp = p[pt];
                        fo[pt] = 2.0 * rho * (c0[pt] * u1 + c1[pt] * u2 + c2[pt] * u3) +
                                 c3[pt] * (rho * u1 * u1 + h00[pt] * pp) +
                                 2.0 * c4[pt] * (rho * u1 * u2 + h01[pt] * pp) +
                                 2.0 * c5[pt] * (rho * u1 * u3 + h02[pt] * pp) +
                                 c6[pt] * (rho * u2 * u2 + h11[pt] * pp) +
                                 2.0 * c7[pt] * (rho * u2 * u3 + h12[pt] * pp) +
                                 c8[pt] * (rho * u3 * u3 + h22[pt] * pp);
                    }
                }
                if (damp) {
                    const double* __restrict dc = dcoef + e * N3;
                    for (int i = 0; i < 3; ++i) {
                        const double* __restrict ur = duref + (size_t)i * fs + e * N3;
                        const double* __restrict ui = u[i];
                        double* __restrict fo = force[i];
#pragma omp simd
                        for (int pt = 0; pt < N3; ++pt) fo[pt] += dc[pt] * rh[pt] * (ui[pt] - ur[pt]);
                    }
                }
                double* __restrict R0 = rhs + 0 * fs + e * N3; double* __restrict R1 = rhs + 1 * fs + e * N3;
                double* __restrict R2 = rhs + 2 * fs + e * N3; double* __restrict R3 = rhs + 3 * fs + e * N3;
                double* __restrict R4 = rhs + 4 * fs + e * N3;
                if (advection_only) {   // cases < 13: every row is zeroed (pde_euler_cubesphere.py:203-290)
                    for (int v = 0; v < 5; ++v) std::memset(rhs + v * fs + e * N3, 0, sizeof(double) * N3);
                    continue;
                }
#pragma omp simd
                for (int pt = 0; pt < N3; ++pt) {
                    const double isg = 1.0 / sgv[pt];
                    const double f2 = force[2][pt] + idzv[pt] * kGravity * isg * hfv[pt];
                    R0[pt] = -isg * tot[0][pt];
                    R1[pt] = -isg * tot[1][pt] - force[0][pt];
                    R2[pt] = -isg * tot[2][pt] - force[1][pt];
                    R3[pt] = -isg * wtot[pt] - f2;
                    R4[pt] = -isg * tot[4][pt];
                }
            }
    return 0;
}

#define WXO_DISPATCH(fn, ...)                 \
    switch (n) {                              \
        case 2: return fn<2>(__VA_ARGS__);    \
        case 3: return fn<3>(__VA_ARGS__);    \
        case 4: return fn<4>(__VA_ARGS__);    \
        case 5: return fn<5>(__VA_ARGS__);    \
        case 6: return fn<6>(__VA_ARGS__);    \
        case 7: return fn<7>(__VA_ARGS__);    \
        case 8: return fn<8>(__VA_ARGS__);    \
    }                                         \
    return 1

}  // namespace

extern "C" {

// the scalar, type-generic forms (what the complex entry points instantiate): kept callable for the cross-check
int wxo_euler3d_extrapolate_generic(int n, int H, int V, const double* em, const double* ep, const double* q, double* itf_i,
                                    double* itf_j, double* itf_k, int nthreads) {
    return extrapolate_impl<double>(n, H, V, em, ep, q, itf_i, itf_j, itf_k, nthreads);
}

int wxo_euler3d_rhs_generic(int n, int H, int V, int case_number, const double* D, const double* C, const double* HF,
                            const double* q, const double* itf_i, const double* itf_j, const double* itf_k,
                            const double* halo_s, const double* halo_n, const double* halo_w, const double* halo_e,
                            const double* sg, const double* h, const double* chr, const double* idz, const double* sgi,
                            const double* sgj, const double* sgk, const double* hi, const double* hj, const double* hk,
                            const double* dcoef, const double* duref, double* rhs, int nthreads) {
    return rhs_impl<double>(n, H, V, case_number, D, C, HF, q, itf_i, itf_j, itf_k, halo_s, halo_n, halo_w, halo_e, sg, h, chr,
                            idz, sgi, sgj, sgk, hi, hj, hk, dcoef, duref, rhs, nthreads);
}

int wxo_euler3d_extrapolate(int n, int H, int V, const double* em, const double* ep, const double* q, double* itf_i,
                            double* itf_j, double* itf_k, int nthreads) {
    WXO_DISPATCH(extrapolate_fast, H, V, em, ep, q, itf_i, itf_j, itf_k, nthreads);
}

// complex128 arrays as interleaved (re, im) doubles
int wxo_euler3d_extrapolate_c(int n, int H, int V, const double* em, const double* ep, const double* q, double* itf_i,
                              double* itf_j, double* itf_k, int nthreads) {
    return extrapolate_impl<cplx>(n, H, V, em, ep, reinterpret_cast<const cplx*>(q), reinterpret_cast<cplx*>(itf_i),
                                  reinterpret_cast<cplx*>(itf_j), reinterpret_cast<cplx*>(itf_k), nthreads);
}

int wxo_euler3d_rhs(int n, int H, int V, int case_number, const double* D, const double* C, const double* HF, const double* q,
                    const double* itf_i, const double* itf_j, const double* itf_k, const double* halo_s,
                    const double* halo_n, const double* halo_w, const double* halo_e, const double* sg, const double* h,
                    const double* chr, const double* idz, const double* sgi, const double* sgj, const double* sgk,
                    const double* hi, const double* hj, const double* hk, const double* dcoef, const double* duref,
                    double* rhs, int nthreads) {
    WXO_DISPATCH(rhs_fast, H, V, case_number, D, C, HF, q, itf_i, itf_j, itf_k, halo_s, halo_n, halo_w, halo_e, sg, h, chr, idz,
                 sgi, sgj, sgk, hi, hj, hk, dcoef, duref, rhs, nthreads);
}

int wxo_euler3d_rhs_c(int n, int H, int V, int case_number, const double* D, const double* C, const double* HF, const double* q,
                      const double* itf_i, const double* itf_j, const double* itf_k, const double* halo_s,
                      const double* halo_n, const double* halo_w, const double* halo_e, const double* sg, const double* h,
                      const double* chr, const double* idz, const double* sgi, const double* sgj, const double* sgk,
                      const double* hi, const double* hj, const double* hk, const double* dcoef, const double* duref,
                      double* rhs, int nthreads) {
    auto c = [](const double* x) { return reinterpret_cast<const cplx*>(x); };
    return rhs_impl<cplx>(n, H, V, case_number, D, C, HF, c(q), c(itf_i), c(itf_j), c(itf_k), c(halo_s), c(halo_n), c(halo_w),
                          c(halo_e), sg, h, chr, idz, sgi, sgj, sgk, hi, hj, hk, dcoef, duref, reinterpret_cast<cplx*>(rhs),
                          nthreads);
}

int wxo_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
