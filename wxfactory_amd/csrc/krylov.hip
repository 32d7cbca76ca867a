// Vector kernels of the matrix-free Krylov solvers, gfx950.
//
// At E7 a Krylov vector is 442 M doubles (3.5 GB); the orthogonalisation of solvers/fgmres.py:150-200 and
// solvers/kiops.py:170-200 is then pure HBM streaming over the basis, and expressing it with array expressions
// (h = V @ w;  w = w - h @ V) costs extra passes and temporaries of 3.5 GB each.  Two kernels do it in the minimum
// number of passes:
//   multi_dot   out[k] = <V[k], w>,  k < m      one pass over the m basis rows and w        (m + 1) n reads
//   multi_axpy  w -= sum_k h[k] V[k]            one pass over the rows, w read and written   (m + 1) n reads + n writes
// Deterministic: multi_dot reduces per workgroup into a partial buffer, a second tiny kernel sums the partials in
// a fixed order (no floating-point atomics).
#include <hip/hip_runtime.h>

#include "wx_common.h"

namespace wx {

constexpr int kDotBlocks = 2048;   // partial sums per row
constexpr int kDotThreads = 256;
constexpr int kRowsPerPass = 8;    // accumulators per thread

template <int R>
__global__ __launch_bounds__(kDotThreads) void multi_dot_kernel(const double* __restrict__ V, size_t ldv, int row0,
                                                                const double* __restrict__ w, size_t n,
                                                                double* __restrict__ partial, int m) {
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double wi = w[i];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] += V[(size_t)(row0 + r) * ldv + i] * wi;
    }
    __shared__ double red[R][kDotThreads / 64];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double v = acc[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[r][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < R) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < kDotThreads / 64; ++k) v += red[threadIdx.x][k];
        partial[(size_t)blockIdx.x * m + row0 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(64) void multi_dot_finish_kernel(const double* __restrict__ partial, int blocks, int m,
                                                              double* __restrict__ out) {
    const int k = blockIdx.x;  // one wave per row, fixed summation order
    double v = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 64) v += partial[(size_t)b * m + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (threadIdx.x == 0) out[k] = v;
}

template <int R>
__global__ __launch_bounds__(256) void multi_axpy_kernel(double* __restrict__ w, const double* __restrict__ V, size_t ldv,
                                                         int row0, const double* __restrict__ h, size_t n) {
    double c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) c[r] = h[row0 + r];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double s = w[i];
#pragma unroll
        for (int r = 0; r < R; ++r) s -= c[r] * V[(size_t)(row0 + r) * ldv + i];
        w[i] = s;
    }
}

template <int R>
static void launch_dot(const double* V, size_t ldv, int row0, const double* w, size_t n, double* partial, int m,
                       hipStream_t st) {
    hipLaunchKernelGGL((multi_dot_kernel<R>), dim3(kDotBlocks), dim3(kDotThreads), 0, st, V, ldv, row0, w, n, partial, m);
}
template <int R>
static void launch_axpy(double* w, const double* V, size_t ldv, int row0, const double* h, size_t n, hipStream_t st) {
    const size_t want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192 ? (want ? want : 1) : 8192);
    hipLaunchKernelGGL((multi_axpy_kernel<R>), dim3(grid), dim3(256), 0, st, w, V, ldv, row0, h, n);
}

}  // namespace wx

using namespace wx;

extern "C" {

size_t wx_multi_dot_workspace(int m) { return (size_t)kDotBlocks * (m > 0 ? m : 1); }

wx_status wx_multi_dot(const double* V, size_t ldv, int m, const double* w, size_t n, double* out, double* workspace,
                       wx_stream stream) {
    if (m <= 0) return WX_OK;
    if (!V || !w || !out || !workspace) return fail(WX_ERR_INVALID, "wx_multi_dot: null argument");
    if (ldv < n) return fail(WX_ERR_INVALID, "wx_multi_dot: row stride %zu shorter than the vectors (%zu)", ldv, n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int r = 0;
    for (; r + kRowsPerPass <= m; r += kRowsPerPass) launch_dot<kRowsPerPass>(V, ldv, r, w, n, workspace, m, st);
    switch (m - r) {
        case 1: launch_dot<1>(V, ldv, r, w, n, workspace, m, st); break;
        case 2: launch_dot<2>(V, ldv, r, w, n, workspace, m, st); break;
        case 3: launch_dot<3>(V, ldv, r, w, n, workspace, m, st); break;
        case 4: launch_dot<4>(V, ldv, r, w, n, workspace, m, st); break;
        case 5: launch_dot<5>(V, ldv, r, w, n, workspace, m, st); break;
        case 6: launch_dot<6>(V, ldv, r, w, n, workspace, m, st); break;
        case 7: launch_dot<7>(V, ldv, r, w, n, workspace, m, st); break;
        default: break;
    }
    hipLaunchKernelGGL(multi_dot_finish_kernel, dim3(m), dim3(64), 0, st, workspace, kDotBlocks, m, out);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_multi_axpy(double* w, const double* V, size_t ldv, int m, const double* h, size_t n, wx_stream stream) {
    if (m <= 0 || n == 0) return WX_OK;
    if (!V || !w || !h) return fail(WX_ERR_INVALID, "wx_multi_axpy: null argument");
    if (ldv < n) return fail(WX_ERR_INVALID, "wx_multi_axpy: row stride %zu shorter than the vectors (%zu)", ldv, n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int r = 0;
    for (; r + kRowsPerPass <= m; r += kRowsPerPass) launch_axpy<kRowsPerPass>(w, V, ldv, r, h, n, st);
    switch (m - r) {
        case 1: launch_axpy<1>(w, V, ldv, r, h, n, st); break;
        case 2: launch_axpy<2>(w, V, ldv, r, h, n, st); break;
        case 3: launch_axpy<3>(w, V, ldv, r, h, n, st); break;
        case 4: launch_axpy<4>(w, V, ldv, r, h, n, st); break;
        case 5: launch_axpy<5>(w, V, ldv, r, h, n, st); break;
        case 6: launch_axpy<6>(w, V, ldv, r, h, n, st); break;
        case 7: launch_axpy<7>(w, V, ldv, r, h, n, st); break;
        default: break;
    }
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

}  // extern "C"
