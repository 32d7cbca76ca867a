// Per-step filters of the explicit time loop, gfx950.
//
//   expfilter_kernel   operators.apply_filter_3d (geometry/operators.py:114-119, 257-261):
//                      out = ((sqrtG * q) @ (Fx Fy Fz)) / sqrtG, the dense n^3 x n^3 operator applied as
//                      three 1-D contractions through LDS; optionally raises the NaN flag of the result
//                      in the same pass (simulation.py:399-408 would re-read the state for that)
//   nan_kernel         simulation._check_for_nan on a state that is not filtered
//   sponge_kernel      the 2-D Cartesian Rayleigh sponge  rho_w *= 1 / (1 + beta dt)  (operators.py:242-253)
//
// All three are single-pass streaming kernels (HBM-bound: 8 (2 nvar + 1) B per point for the filter).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wx_common.h"
#include "wx_math.h"
#include "wx_mfma.h"

namespace wx {

constexpr int kFilterMaxVar = 5;

template <int N>
struct FCfg {
    static constexpr int N2 = N * N, N3 = N * N * N;
    static constexpr int EPB = (N3 >= 216) ? 1 : (256 / N3);
    static constexpr int BS = ((EPB * N3 + 63) / 64) * 64;
    static constexpr int NP = (N % 2 == 0) ? N + 1 : N;  // padded row: conflict-free reads along j and k
    static constexpr int LE = N2 * NP;
    __host__ __device__ static constexpr int lidx(int kl, int jl, int il) { return (kl * N + jl) * NP + il; }
};

__device__ __forceinline__ bool w_isnan(double a) { return a != a; }
__device__ __forceinline__ bool w_isnan(cplx a) { return a.re != a.re || a.im != a.im; }

template <int N, typename T>
__global__ __launch_bounds__(FCfg<N>::BS) void expfilter_kernel(const T* __restrict__ q, T* __restrict__ out,
                                                                const double* __restrict__ sqrtG,
                                                                const double* __restrict__ filter, int nvar, size_t nelem,
                                                                int* nan_flag, size_t q_panel_stride, size_t sg_panel_stride) {
    // blockIdx.y = panel of a stacked state (strides 0 and gridDim.y = 1 for a single panel)
    q += (size_t)blockIdx.y * q_panel_stride;
    out += (size_t)blockIdx.y * q_panel_stride;
    sqrtG += (size_t)blockIdx.y * sg_panel_stride;
    using C = FCfg<N>;
    constexpr int N2 = C::N2, N3 = C::N3, EPB = C::EPB;
    __shared__ T fld[kFilterMaxVar][EPB * C::LE];
    __shared__ double sF[N * N];
    const int tid = threadIdx.x;
    for (int i = tid; i < N * N; i += C::BS) sF[i] = filter[i];
    const int le = tid / N3, pt = tid % N3;
    const size_t e = (size_t)blockIdx.x * EPB + le;
    const bool active = le < EPB && e < nelem;
    const int kl = pt / N2, jl = (pt / N) % N, il = pt % N;
    const int lb = (le < EPB ? le : 0) * C::LE;
    const int lpt = lb + C::lidx(kl, jl, il);
    const size_t o = e * N3 + pt, fs = nelem * N3;
    const double sg = active ? sqrtG[o] : 1.0;
    T v[kFilterMaxVar];
#pragma unroll
    for (int f = 0; f < kFilterMaxVar; ++f) v[f] = (active && f < nvar) ? sg * q[(size_t)f * fs + o] : T(0.0);

    // n = 8, float64: the three passes on the matrix cores (wx_mfma.h: in place, every thread stages and picks up its
    // own node, the eight waves contract one octet of lines each); the filter matrix is the operator, no face step
    constexpr bool MF = N == 8 && std::is_same<T, double>::value && WX_MFMA;
    if constexpr (MF) {
        static_assert(C::LE == kMfLE && EPB == 1, "the matrix-core pass owns one n = 8 element per workgroup");
        double* fm = reinterpret_cast<double*>(&fld[0][0]);
        const int lptm = mf_idx(kl, jl, il);
        const MfOps4 ops = mf4_load_ops(filter, nullptr, nullptr, nullptr, tid & 63);
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int f = 0; f < kFilterMaxVar; ++f) fm[f * kMfLE + lptm] = v[f];
            __syncthreads();
            if (d == 0) mf4_dir_pass<0, false, kFilterMaxVar, false>(fm, fm, ops, wave, tid & 63);
            else if (d == 1) mf4_dir_pass<1, false, kFilterMaxVar, false>(fm, fm, ops, wave, tid & 63);
            else mf4_dir_pass<2, false, kFilterMaxVar, false>(fm, fm, ops, wave, tid & 63);
            __syncthreads();
#pragma unroll
            for (int f = 0; f < kFilterMaxVar; ++f) v[f] = fm[f * kMfLE + lptm];
        }
    }
#pragma unroll
    for (int d = 0; d < (MF ? 0 : 3); ++d) {
        if (d > 0) __syncthreads();
        if (le < EPB) {
#pragma unroll
            for (int f = 0; f < kFilterMaxVar; ++f) fld[f][lpt] = v[f];
        }
        __syncthreads();
        int base, stride, idx;
        if (d == 0) { base = lb + C::lidx(kl, jl, 0); stride = 1; idx = il; }
        else if (d == 1) { base = lb + C::lidx(kl, 0, il); stride = C::NP; idx = jl; }
        else { base = lb + C::lidx(0, jl, il); stride = N * C::NP; idx = kl; }
#pragma unroll
        for (int f = 0; f < kFilterMaxVar; ++f) v[f] = T(0.0);
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const double w = sF[idx * N + m];
#pragma unroll
            for (int f = 0; f < kFilterMaxVar; ++f) v[f] += w * fld[f][base + m * stride];
        }
    }
    if (active) {
        const double inv = 1.0 / sg;  // metric.inv_sqrtG_new is exactly 1 / sqrtG_new (metric3d.py)
        bool bad = false;
#pragma unroll
        for (int f = 0; f < kFilterMaxVar; ++f) {
            if (f < nvar) {
                const T r = v[f] * inv;
                out[(size_t)f * fs + o] = r;
                bad = bad || w_isnan(r);
            }
        }
        if (bad && nan_flag) *nan_flag = 1;
    }
}

__global__ __launch_bounds__(256) void nan_kernel(const double* __restrict__ x, size_t count, int* flag) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const double a = x[i];
        bad = bad || (a != a);
    }
    if (bad) *flag = 1;
}

template <typename T>
__global__ __launch_bounds__(256) void sponge_kernel(T* __restrict__ rho_w, const double* __restrict__ beta, double dt,
                                                     size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) rho_w[i] = (1.0 / (1.0 + beta[i] * dt)) * rho_w[i];
}

template <int N, typename T>
static wx_status launch_filter(const void* q, void* out, const double* sg, const double* f, int nvar, size_t nelem,
                               int* flag, int npanels, hipStream_t st) {
    using C = FCfg<N>;
    const size_t grid = (nelem + C::EPB - 1) / C::EPB;
    const size_t sgs = npanels > 1 ? nelem * C::N3 : 0, qs = sgs * nvar;
    hipLaunchKernelGGL((expfilter_kernel<N, T>), dim3((unsigned)grid, (unsigned)npanels), dim3(C::BS), 0, st,
                       static_cast<const T*>(q), static_cast<T*>(out), sg, f, nvar, nelem, flag, qs, sgs);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <typename T>
static wx_status dispatch_filter(int n, const void* q, void* out, const double* sg, const double* f, int nvar,
                                 size_t nelem, int* flag, int npanels, hipStream_t st) {
    switch (n) {
        case 2: return launch_filter<2, T>(q, out, sg, f, nvar, nelem, flag, npanels, st);
        case 3: return launch_filter<3, T>(q, out, sg, f, nvar, nelem, flag, npanels, st);
        case 4: return launch_filter<4, T>(q, out, sg, f, nvar, nelem, flag, npanels, st);
        case 5: return launch_filter<5, T>(q, out, sg, f, nvar, nelem, flag, npanels, st);
        case 6: return launch_filter<6, T>(q, out, sg, f, nvar, nelem, flag, npanels, st);
        case 7: return launch_filter<7, T>(q, out, sg, f, nvar, nelem, flag, npanels, st);
        case 8: return launch_filter<8, T>(q, out, sg, f, nvar, nelem, flag, npanels, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", n);
}

}  // namespace wx

using namespace wx;

struct wx_expfilter {
    int n;
    double* filter;  // device, n x n row-major
};

extern "C" {

wx_status wx_expfilter_create(wx_expfilter** out, int n, const double* filter) {
    if (!out || !filter) return fail(WX_ERR_INVALID, "wx_expfilter_create: null argument");
    if (n < 2 || n > 8) return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", n);
    wx_expfilter* h = new (std::nothrow) wx_expfilter{n, nullptr};
    if (!h) return fail(WX_ERR_NOMEM, "wx_expfilter_create: out of host memory");
    hipError_t e = hipMalloc((void**)&h->filter, sizeof(double) * n * n);
    if (e == hipSuccess) e = hipMemcpy(h->filter, filter, sizeof(double) * n * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->filter) (void)hipFree(h->filter);
        delete h;
        return fail(WX_ERR_HIP, "filter upload failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return WX_OK;
}

wx_status wx_expfilter_destroy(wx_expfilter* h) {
    if (!h) return WX_OK;
    hipError_t e = hipFree(h->filter);
    delete h;
    if (e != hipSuccess) return fail(WX_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
    return WX_OK;
}

int wx_expfilter_uses_matrix_cores(const wx_expfilter* h, wx_dtype dtype) {
    if (!h) return -1;
    return (WX_MFMA && h->n == 8 && dtype == WX_F64) ? 1 : 0;
}

static wx_status expfilter_apply_impl(const wx_expfilter* h, const void* q, void* out, const double* sqrtG, int nvar,
                                      size_t nelem, int npanels, wx_dtype dtype, int* nan_flag, wx_stream stream);

wx_status wx_expfilter_apply(const wx_expfilter* h, const void* q, void* out, const double* sqrtG, int nvar, size_t nelem,
                             wx_dtype dtype, int* nan_flag, wx_stream stream) {
    return expfilter_apply_impl(h, q, out, sqrtG, nvar, nelem, 1, dtype, nan_flag, stream);
}

wx_status wx_expfilter_apply_stacked(const wx_expfilter* h, const void* q, void* out, const double* sqrtG, int nvar,
                                     size_t nelem, int npanels, wx_dtype dtype, int* nan_flag, wx_stream stream) {
    if (npanels < 1 || npanels > 65535) return fail(WX_ERR_INVALID, "npanels %d not in 1..65535", npanels);
    return expfilter_apply_impl(h, q, out, sqrtG, nvar, nelem, npanels, dtype, nan_flag, stream);
}

static wx_status expfilter_apply_impl(const wx_expfilter* h, const void* q, void* out, const double* sqrtG, int nvar,
                                      size_t nelem, int npanels, wx_dtype dtype, int* nan_flag, wx_stream stream) {
    if (!h || !q || !out || !sqrtG) return fail(WX_ERR_INVALID, "wx_expfilter_apply: null argument");
    if (nvar < 1 || nvar > kFilterMaxVar) return fail(WX_ERR_INVALID, "nvar %d not in 1..%d", nvar, kFilterMaxVar);
    if (nelem == 0) return WX_OK;
    if (nelem > 0x7fffffffu) return fail(WX_ERR_INVALID, "too many elements for one launch");
    WX_STREAM(st, stream);
    // the filter is linear with real coefficients: dual numbers filter component-wise, like complex ones
    if (dtype == WX_F64) return dispatch_filter<double>(h->n, q, out, sqrtG, h->filter, nvar, nelem, nan_flag, npanels, st);
    if (dtype == WX_C128 || dtype == WX_DUAL128)
        return dispatch_filter<cplx>(h->n, q, out, sqrtG, h->filter, nvar, nelem, nan_flag, npanels, st);
    return fail(WX_ERR_INVALID, "unknown dtype %d", (int)dtype);
}

wx_status wx_check_nan(const void* q, size_t count, wx_dtype dtype, int* flag, wx_stream stream) {
    if (!flag) return fail(WX_ERR_INVALID, "wx_check_nan: null flag");
    if (count == 0) return WX_OK;
    if (!q) return fail(WX_ERR_INVALID, "wx_check_nan: null array");
    if (dtype != WX_F64 && dtype != WX_C128 && dtype != WX_DUAL128) return fail(WX_ERR_INVALID, "unknown dtype %d", (int)dtype);
    const size_t doubles = count * (dtype == WX_F64 ? 1 : 2);
    size_t grid = (doubles + 256 * 8 - 1) / (256 * 8);
    if (grid > 256 * 16) grid = 256 * 16;
    WX_STREAM(st, stream);
    hipLaunchKernelGGL(nan_kernel, dim3((unsigned)grid), dim3(256), 0, st,
                       static_cast<const double*>(q), doubles, flag);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_cart2d_sponge(void* rho_w, const double* beta, double dt, size_t count, wx_dtype dtype, wx_stream stream) {
    if (count == 0) return WX_OK;
    if (!rho_w || !beta) return fail(WX_ERR_INVALID, "wx_cart2d_sponge: null argument");
    const unsigned grid = (unsigned)((count + 255) / 256);
    WX_STREAM(st, stream);
    if (dtype == WX_F64) hipLaunchKernelGGL(sponge_kernel<double>, dim3(grid), dim3(256), 0, st, static_cast<double*>(rho_w), beta, dt, count);
    else if (dtype == WX_C128 || dtype == WX_DUAL128)
        hipLaunchKernelGGL(sponge_kernel<cplx>, dim3(grid), dim3(256), 0, st, static_cast<cplx*>(rho_w), beta, dt, count);
    else return fail(WX_ERR_INVALID, "unknown dtype %d", (int)dtype);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

}  // extern "C"
