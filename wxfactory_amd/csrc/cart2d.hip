// 2-D Cartesian Euler (the reference's plumbing case) and the legacy `pde` module entry points.
//
// (1) HIP equivalents, one to one, of the reference's native module functions
//     (pde/interface.cpp:282-302 CPU, pde/interface.cu:433-442 CUDA):
//       pointwise_eulercartesian_2d     kernels/pointwise_flux.hpp:3-31
//       riemann_eulercartesian_ausm_2d  kernels/riemann_flux.hpp:5-80 + boundary_flux.hpp:3-24
//       forcing_euler_cubesphere_3d     kernels/forcing.hpp:6-100
//     Same argument order and array layouts, so PDEEulerCartesian (pde_euler_cartesian.py:8-48)
//     can call them unchanged through HipDevice.pde.
// (2) wx_cart2d_rhs: the whole 2-D RHS (rhs_dfr.py:8-45) in ONE launch - the case is 57 k DOF,
//     launch-latency bound, so each workgroup rebuilds its neighbours' face values from their
//     nodal values (L2 hits) instead of a second kernel.
#include "wx_common.h"
#include "wx_math.h"

#include <cstring>
#include <new>

namespace wx {

template <typename T>
__device__ __forceinline__ T pressure_pow(T rho_theta) {
    // p0 * pow(rho_theta * Rd / p0, gamma)   (riemann_flux.hpp:33-34, boundary_flux.hpp:13)
    return kP0 * w_exp(kGamma * w_log(rho_theta * kRd * (1.0 / kP0)));
}
template <>
__device__ __forceinline__ double pressure_pow<double>(double rho_theta) {
    return kP0 * pow(rho_theta * kRd * (1.0 / kP0), kGamma);
}

// ---------------------------------------------------------------- pointwise_eulercartesian_2d
template <typename T>
__global__ void cart2d_pointwise_kernel(const T* __restrict__ q, T* __restrict__ f1, T* __restrict__ f3, size_t npts) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const T rho = q[i], ru = q[npts + i], rw = q[2 * npts + i], rt = q[3 * npts + i];
    const T inv = 1.0 / rho;
    const T u = ru * inv, w = rw * inv;
    const T p = kP0 * w_exp(kGamma * w_log(kRdOverP0 * rt));
    f1[i] = ru;
    f1[npts + i] = ru * u + p;
    f1[2 * npts + i] = ru * w;
    f1[3 * npts + i] = rt * u;
    f3[i] = rw;
    f3[npts + i] = rw * u;
    f3[2 * npts + i] = rw * w + p;
    f3[3 * npts + i] = rt * w;
}

// ---------------------------------------------------------------- AUSM (riemann_flux.hpp:5-80)
template <typename T>
__device__ __forceinline__ void ausm_2d(const T* qL, const T* qR, int dir, T* f) {
    const T invL = 1.0 / qL[0], invR = 1.0 / qR[0];
    const T uL = qL[1] * invL, wL = qL[2] * invL, uR = qR[1] * invR, wR = qR[2] * invR;
    const T pL = pressure_pow<T>(qL[3]), pR = pressure_pow<T>(qR[3]);
    const T aL = w_sqrt(kGamma * pL * invL), aR = w_sqrt(kGamma * pR * invR);
    const T vL = w_sel(dir == 0, uL, wL), vR = w_sel(dir == 0, uR, wR);
    const T ML = vL / aL + 1.0, MR = vR / aR - 1.0;
    const T M = 0.25 * (ML * ML - MR * MR);
    const T Mmax = w_max(T(0.0), M) * aL, Mmin = w_min(T(0.0), M) * aR;
    const T pf = 0.5 * (ML * pL - MR * pR);
    f[0] = qL[0] * Mmax + qR[0] * Mmin;
    f[1] = w_sel(dir == 0, pf, qL[1] * Mmax + qR[1] * Mmin);
    f[2] = w_sel(dir == 0, qL[2] * Mmax + qR[2] * Mmin, pf);
    f[3] = qL[3] * Mmax + qR[3] * Mmin;
}

// one thread per (element, k) for both directions; interior faces + solid walls
// (loop structure of pde/interface.cpp:154-238)
template <typename T>
__global__ void cart2d_riemann_kernel(const T* __restrict__ qi1, const T* __restrict__ qi3, T* __restrict__ fi1,
                                      T* __restrict__ fi3, int nx, int nz, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = nz * nx * n;
    if (t >= total) return;
    const int k = t % n, ex = (t / n) % nx, ez = t / (n * nx);
    const size_t stride = (size_t)nz * nx * 2 * n;
    const size_t e0 = ((size_t)ez * nx + ex) * 2 * n;
    T qL[4], qR[4], f[4];
    // horizontal: plus side of (ez, ex) with minus side of (ez, ex+1)
    if (ex + 1 < nx) {
        const size_t il = e0 + n + k, ir = e0 + 2 * n + k;
#pragma unroll
        for (int v = 0; v < 4; ++v) { qL[v] = qi1[v * stride + il]; qR[v] = qi1[v * stride + ir]; }
        ausm_2d<T>(qL, qR, 0, f);
#pragma unroll
        for (int v = 0; v < 4; ++v) { fi1[v * stride + il] = f[v]; fi1[v * stride + ir] = f[v]; }
    }
    if (ez + 1 < nz) {
        const size_t il = e0 + n + k, ir = e0 + (size_t)nx * 2 * n + k;
#pragma unroll
        for (int v = 0; v < 4; ++v) { qL[v] = qi3[v * stride + il]; qR[v] = qi3[v * stride + ir]; }
        ausm_2d<T>(qL, qR, 1, f);
#pragma unroll
        for (int v = 0; v < 4; ++v) { fi3[v * stride + il] = f[v]; fi3[v * stride + ir] = f[v]; }
    }
    // walls: only the pressure on the normal momentum (boundary_flux.hpp:3-24)
    if (ex == 0 || ex == nx - 1) {
        for (int side = 0; side < 2; ++side) {
            if ((side == 0 && ex != 0) || (side == 1 && ex != nx - 1)) continue;
            const size_t i = e0 + side * n + k;
            const T p = pressure_pow<T>(qi1[3 * stride + i]);
            fi1[i] = T(0.0); fi1[stride + i] = p; fi1[2 * stride + i] = T(0.0); fi1[3 * stride + i] = T(0.0);
        }
    }
    if (ez == 0 || ez == nz - 1) {
        for (int side = 0; side < 2; ++side) {
            if ((side == 0 && ez != 0) || (side == 1 && ez != nz - 1)) continue;
            const size_t i = e0 + side * n + k;
            const T p = pressure_pow<T>(qi3[3 * stride + i]);
            fi3[i] = T(0.0); fi3[stride + i] = T(0.0); fi3[2 * stride + i] = p; fi3[3 * stride + i] = T(0.0);
        }
    }
}

// ---------------------------------------------------------------- forcing_euler_cubesphere_3d
template <typename T>
__global__ void euler3d_forcing_kernel(const T* __restrict__ q, const T* __restrict__ pressure,
                                       const double* __restrict__ h, const double* __restrict__ chr,
                                       T* __restrict__ forcing, size_t npts) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const T r = q[i];
    const T u = q[npts + i] / r, v = q[2 * npts + i] / r, w = q[3 * npts + i] / r;
    const T p = pressure[i];
    const double h11 = h[i], h12 = h[npts + i], h13 = h[2 * npts + i], h22 = h[4 * npts + i], h23 = h[5 * npts + i],
                 h33 = h[8 * npts + i];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double* c = chr + (size_t)(9 * a) * npts + i;
        const double c01 = c[0], c02 = c[npts], c03 = c[2 * npts], c11 = c[3 * npts], c12 = c[4 * npts],
                     c13 = c[5 * npts], c22 = c[6 * npts], c23 = c[7 * npts], c33 = c[8 * npts];
        forcing[(size_t)(1 + a) * npts + i] =
            2.0 * r * (c01 * u + c02 * v + c03 * w) + c11 * (r * u * u + h11 * p) + 2.0 * c12 * (r * u * v + h12 * p) +
            2.0 * c13 * (r * u * w + h13 * p) + c22 * (r * v * v + h22 * p) + 2.0 * c23 * (r * v * w + h23 * p) +
            c33 * (r * w * w + h33 * p);
    }
}

// ---------------------------------------------------------------- fused 2-D RHS (rhs_dfr.py:8-45)
constexpr int kMaxNc = 8;
struct CartConsts {
    double em[kMaxNc], ep[kMaxNc], cm[kMaxNc], cp[kMaxNc], D[kMaxNc * kMaxNc];
};

template <typename T>
struct CartParams {
    int nx, nz;
    double sx, sz;  // -2/dx1, -2/dx3
    const T* q;
    T* rhs;
    const CartConsts* K;
};

// One workgroup per element, N*N threads (rounded to a wave).  LDS holds the element and its four
// neighbours (5 x 4 vars x N^2); face values of both sides are extrapolated from it.
template <int N, typename T>
__global__ __launch_bounds__(64) void cart2d_rhs_kernel(const CartParams<T> P) {
    constexpr int N2 = N * N;
    constexpr int NP = (N % 2 == 0) ? N + 1 : N;
    __shared__ T el[5][4][N * NP];   // 0 self, 1 W, 2 E, 3 S(below), 4 N(above)
    __shared__ T fr[4][4][N];        // common flux on faces W,E,B,T x 4 vars
    __shared__ T fl[4][N * NP];      // flux field being differentiated
    __shared__ double sD[N * N], sEm[N], sEp[N], sCm[N], sCp[N];
    const int tid = threadIdx.x;
    const int ex = blockIdx.x % P.nx, ez = blockIdx.x / P.nx;
    const size_t fs = (size_t)P.nx * P.nz * N2;
    for (int i = tid; i < N * N; i += 64) sD[i] = P.K->D[i];
    if (tid < N) { sEm[tid] = P.K->em[tid]; sEp[tid] = P.K->ep[tid]; sCm[tid] = P.K->cm[tid]; sCp[tid] = P.K->cp[tid]; }
    for (int i = tid; i < 5 * N2; i += 64) {
        const int s = i / N2, pt = i % N2;
        const int x = ex + (s == 2) - (s == 1), z = ez + (s == 4) - (s == 3);
        if (x < 0 || x >= P.nx || z < 0 || z >= P.nz) continue;
        const size_t o = ((size_t)z * P.nx + x) * N2 + pt;
        const int lp = (pt / N) * NP + pt % N;
#pragma unroll
        for (int v = 0; v < 4; ++v) el[s][v][lp] = P.q[v * fs + o];
    }
    __syncthreads();
    // faces: thread = (face f, k)
    if (tid < 4 * N) {
        const int f = tid / N, k = tid % N;
        const int d = f >> 1, plus = f & 1;
        const int base = d == 0 ? k * NP : k, stride = d == 0 ? 1 : NP;
        const int nb = 1 + f;  // neighbour slot in el[]
        const bool wall = d == 0 ? (plus ? ex == P.nx - 1 : ex == 0) : (plus ? ez == P.nz - 1 : ez == 0);
        T qo[4], qn[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            T a = T(0.0), b = T(0.0);
#pragma unroll
            for (int m = 0; m < N; ++m) {
                a += (plus ? sEp[m] : sEm[m]) * el[0][v][base + m * stride];
                if (!wall) b += (plus ? sEm[m] : sEp[m]) * el[nb][v][base + m * stride];
            }
            qo[v] = a;
            qn[v] = b;
        }
        T out[4];
        if (wall) {
            const T p = pressure_pow<T>(qo[3]);
            out[0] = T(0.0); out[1] = w_sel(d == 0, p, T(0.0)); out[2] = w_sel(d == 0, T(0.0), p); out[3] = T(0.0);
        } else if (plus) {
            ausm_2d<T>(qo, qn, d, out);
        } else {
            ausm_2d<T>(qn, qo, d, out);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) fr[f][v][k] = out[v];
    }
    // point stage
    const bool act = tid < N2;
    const int kl = act ? tid / N : 0, il = act ? tid % N : 0;
    const int lp = kl * NP + il;
    T rho = T(1.0), ru = T(0.0), rw = T(0.0), rt = T(1.0);
    if (act) { rho = el[0][0][lp]; ru = el[0][1][lp]; rw = el[0][2][lp]; rt = el[0][3][lp]; }
    const T inv = 1.0 / rho;
    const T u = ru * inv, w = rw * inv;
    const T p = kP0 * w_exp(kGamma * w_log(kRdOverP0 * rt));
    T r[4] = {T(0.0), T(0.0), T(0.0), T(0.0)};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        __syncthreads();
        if (act) {
            if (d == 0) { fl[0][lp] = ru; fl[1][lp] = ru * u + p; fl[2][lp] = ru * w; fl[3][lp] = rt * u; }
            else { fl[0][lp] = rw; fl[1][lp] = rw * u; fl[2][lp] = rw * w + p; fl[3][lp] = rt * w; }
        }
        __syncthreads();
        const int base = d == 0 ? kl * NP : il, stride = d == 0 ? 1 : NP;
        const int idx = d == 0 ? il : kl, fp = d == 0 ? kl : il;
        const double sc = d == 0 ? P.sx : P.sz;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            T a = sCm[idx] * fr[2 * d][v][fp] + sCp[idx] * fr[2 * d + 1][v][fp];
#pragma unroll
            for (int m = 0; m < N; ++m) a += sD[idx * N + m] * fl[v][base + m * stride];
            r[v] += a * sc;
        }
    }
    if (!act) return;
    r[2] -= rho * kGravity;  // pde_euler_cartesian.py:47-48
    const size_t o = ((size_t)ez * P.nx + ex) * N2 + tid;
#pragma unroll
    for (int v = 0; v < 4; ++v) P.rhs[v * fs + o] = r[v];
}

}  // namespace wx

using namespace wx;

struct wx_cart2d_plan {
    int n, nx, nz;
    double dx1, dx3;
    wx_dtype dtype;
    CartConsts* consts = nullptr;
};

namespace {
template <int N, typename T>
wx_status cart_launch(const CartParams<T>& P, hipStream_t st) {
    hipLaunchKernelGGL((cart2d_rhs_kernel<N, T>), dim3(P.nx * P.nz), dim3(64), 0, st, P);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}
template <typename T>
wx_status cart_dispatch(int n, const CartParams<T>& P, hipStream_t st) {
    switch (n) {
        case 2: return cart_launch<2, T>(P, st);
        case 3: return cart_launch<3, T>(P, st);
        case 4: return cart_launch<4, T>(P, st);
        case 5: return cart_launch<5, T>(P, st);
        case 6: return cart_launch<6, T>(P, st);
        case 7: return cart_launch<7, T>(P, st);
        case 8: return cart_launch<8, T>(P, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", n);
}
}  // namespace

extern "C" {

wx_status wx_pointwise_eulercartesian_2d(const void* q, void* flux_x1, void* flux_x3, int num_elem_x1, int num_elem_x3,
                                         int num_solpts_tot, wx_dtype dtype, wx_stream stream) {
    if (!q || !flux_x1 || !flux_x3) return fail(WX_ERR_INVALID, "wx_pointwise_eulercartesian_2d: null argument");
    if (num_elem_x1 < 1 || num_elem_x3 < 1 || num_solpts_tot < 1) return fail(WX_ERR_INVALID, "bad sizes");
    const size_t npts = (size_t)num_elem_x1 * num_elem_x3 * num_solpts_tot;
    const int grid = (int)((npts + 255) / 256);
    WX_STREAM(st, stream);
    if (dtype == WX_F64)
        hipLaunchKernelGGL((cart2d_pointwise_kernel<double>), dim3(grid), dim3(256), 0, st, (const double*)q,
                           (double*)flux_x1, (double*)flux_x3, npts);
    else if (dtype == WX_C128)
        hipLaunchKernelGGL((cart2d_pointwise_kernel<cplx>), dim3(grid), dim3(256), 0, st, (const cplx*)q, (cplx*)flux_x1,
                           (cplx*)flux_x3, npts);
    else if (dtype == WX_DUAL128)
        hipLaunchKernelGGL((cart2d_pointwise_kernel<dual>), dim3(grid), dim3(256), 0, st, (const dual*)q, (dual*)flux_x1,
                           (dual*)flux_x3, npts);
    else
        return fail(WX_ERR_INVALID, "unknown dtype %d (the reference's CUDA dispatcher is silent here)", (int)dtype);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_riemann_eulercartesian_ausm_2d(const void* q_itf_x1, const void* q_itf_x3, void* flux_itf_x1,
                                            void* flux_itf_x3, int num_elem_x1, int num_elem_x3, int num_solpts,
                                            wx_dtype dtype, wx_stream stream) {
    if (!q_itf_x1 || !q_itf_x3 || !flux_itf_x1 || !flux_itf_x3)
        return fail(WX_ERR_INVALID, "wx_riemann_eulercartesian_ausm_2d: null argument");
    if (num_elem_x1 < 1 || num_elem_x3 < 1 || num_solpts < 1) return fail(WX_ERR_INVALID, "bad sizes");
    const int total = num_elem_x1 * num_elem_x3 * num_solpts;
    const int grid = (total + 255) / 256;
    WX_STREAM(st, stream);
    if (dtype == WX_F64)
        hipLaunchKernelGGL((cart2d_riemann_kernel<double>), dim3(grid), dim3(256), 0, st, (const double*)q_itf_x1,
                           (const double*)q_itf_x3, (double*)flux_itf_x1, (double*)flux_itf_x3, num_elem_x1, num_elem_x3,
                           num_solpts);
    else if (dtype == WX_C128)
        hipLaunchKernelGGL((cart2d_riemann_kernel<cplx>), dim3(grid), dim3(256), 0, st, (const cplx*)q_itf_x1,
                           (const cplx*)q_itf_x3, (cplx*)flux_itf_x1, (cplx*)flux_itf_x3, num_elem_x1, num_elem_x3,
                           num_solpts);
    else if (dtype == WX_DUAL128)
        hipLaunchKernelGGL((cart2d_riemann_kernel<dual>), dim3(grid), dim3(256), 0, st, (const dual*)q_itf_x1,
                           (const dual*)q_itf_x3, (dual*)flux_itf_x1, (dual*)flux_itf_x3, num_elem_x1, num_elem_x3,
                           num_solpts);
    else
        return fail(WX_ERR_INVALID, "unknown dtype %d", (int)dtype);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_forcing_euler_cubesphere_3d(const void* q, const void* pressure, const double* sqrt_g, const double* h,
                                         const double* christoffel, void* forcing, int num_elem_x1, int num_elem_x2,
                                         int num_elem_x3, int num_solpts, wx_dtype dtype, wx_stream stream) {
    (void)sqrt_g;  // accepted and unused, as in the reference kernel (kernels/forcing.hpp)
    if (!q || !pressure || !h || !christoffel || !forcing)
        return fail(WX_ERR_INVALID, "wx_forcing_euler_cubesphere_3d: null argument");
    const size_t npts = (size_t)num_elem_x1 * num_elem_x2 * num_elem_x3 * num_solpts;
    const int grid = (int)((npts + 127) / 128);
    WX_STREAM(st, stream);
    if (dtype == WX_F64)
        hipLaunchKernelGGL((euler3d_forcing_kernel<double>), dim3(grid), dim3(128), 0, st, (const double*)q,
                           (const double*)pressure, h, christoffel, (double*)forcing, npts);
    else if (dtype == WX_C128)
        hipLaunchKernelGGL((euler3d_forcing_kernel<cplx>), dim3(grid), dim3(128), 0, st, (const cplx*)q,
                           (const cplx*)pressure, h, christoffel, (cplx*)forcing, npts);
    else if (dtype == WX_DUAL128)
        hipLaunchKernelGGL((euler3d_forcing_kernel<dual>), dim3(grid), dim3(128), 0, st, (const dual*)q,
                           (const dual*)pressure, h, christoffel, (dual*)forcing, npts);
    else
        return fail(WX_ERR_INVALID, "unknown dtype %d", (int)dtype);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_cart2d_plan_create(wx_cart2d_plan** out, int n, int num_elem_x1, int num_elem_x3, double dx1, double dx3,
                                wx_dtype dtype, const wx_dfr_ops* ops) {
    if (!out || !ops) return fail(WX_ERR_INVALID, "wx_cart2d_plan_create: null argument");
    *out = nullptr;
    if (n < 2 || n > kMaxNc) return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..%d", n, kMaxNc);
    if (num_elem_x1 < 1 || num_elem_x3 < 1 || !(dx1 > 0.0) || !(dx3 > 0.0)) return fail(WX_ERR_INVALID, "bad grid");
    if (dtype != WX_F64 && dtype != WX_C128 && dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "unknown dtype %d", (int)dtype);
    if (!ops->extrap_neg || !ops->extrap_pos || !ops->diff_solpt || !ops->correction)
        return fail(WX_ERR_INVALID, "wx_dfr_ops has a null member");
    wx_cart2d_plan* pl = new (std::nothrow) wx_cart2d_plan();
    if (!pl) return fail(WX_ERR_NOMEM, "out of host memory");
    pl->n = n; pl->nx = num_elem_x1; pl->nz = num_elem_x3; pl->dx1 = dx1; pl->dx3 = dx3; pl->dtype = dtype;
    CartConsts hc;
    memset(&hc, 0, sizeof(hc));
    for (int i = 0; i < n; ++i) {
        hc.em[i] = ops->extrap_neg[i]; hc.ep[i] = ops->extrap_pos[i];
        hc.cm[i] = ops->correction[2 * i]; hc.cp[i] = ops->correction[2 * i + 1];
        for (int j = 0; j < n; ++j) hc.D[i * n + j] = ops->diff_solpt[i * n + j];
    }
    hipError_t e = hipMalloc((void**)&pl->consts, sizeof(CartConsts));
    if (e == hipSuccess) e = hipMemcpy(pl->consts, &hc, sizeof(hc), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (pl->consts) (void)hipFree(pl->consts);
        delete pl;
        return fail(WX_ERR_HIP, "wx_cart2d_plan_create: %s", hipGetErrorString(e));
    }
    *out = pl;
    return WX_OK;
}

wx_status wx_cart2d_plan_destroy(wx_cart2d_plan* pl) {
    if (!pl) return WX_OK;
    hipError_t e = hipFree(pl->consts);
    delete pl;
    if (e != hipSuccess) return fail(WX_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
    return WX_OK;
}

wx_status wx_cart2d_rhs(wx_cart2d_plan* pl, const void* q, void* rhs, wx_stream stream) {
    if (!pl || !q || !rhs) return fail(WX_ERR_INVALID, "wx_cart2d_rhs: null argument");
    WX_STREAM(st, stream);
    if (pl->dtype == WX_F64) {
        CartParams<double> P{pl->nx, pl->nz, -2.0 / pl->dx1, -2.0 / pl->dx3, (const double*)q, (double*)rhs, pl->consts};
        return cart_dispatch<double>(pl->n, P, st);
    }
    if (pl->dtype == WX_DUAL128) {
        CartParams<dual> P{pl->nx, pl->nz, -2.0 / pl->dx1, -2.0 / pl->dx3, (const dual*)q, (dual*)rhs, pl->consts};
        return cart_dispatch<dual>(pl->n, P, st);
    }
    CartParams<cplx> P{pl->nx, pl->nz, -2.0 / pl->dx1, -2.0 / pl->dx3, (const cplx*)q, (cplx*)rhs, pl->consts};
    return cart_dispatch<cplx>(pl->n, P, st);
}

}  // extern "C"
