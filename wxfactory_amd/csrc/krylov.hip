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
constexpr int kRowsPerPass2 = 4;   // rows per pass of the two-vector kernels (2 accumulators / coefficients per row)

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

// The two kernels of the low-synchronisation Gram-Schmidt step (solvers/fgmres.py:16-73: every basis row against the
// LAST TWO rows in one fused reduction, then both rows corrected, scaled and mutually orthogonalised):
//   multi_dot2   partial sums of <V[k], a> and <V[k], b> for R rows in one pass (a, b are read once per pass)
//   pair_update  a -= sum_k ha[k] V[k];  b -= sum_k hb[k] V[k];  a *= sa;  b = (b - cross a) * sb   in one pass
template <int R>
__global__ __launch_bounds__(kDotThreads) void multi_dot2_kernel(const double* __restrict__ V, size_t ldv, int row0,
                                                                 const double* __restrict__ a, const double* __restrict__ b,
                                                                 size_t n, double* __restrict__ partial, int m) {
    double acc[2 * R];
#pragma unroll
    for (int r = 0; r < 2 * R; ++r) acc[r] = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double ai = a[i], bi = b[i];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double v = V[(size_t)(row0 + r) * ldv + i];
            acc[r] += v * ai;
            acc[R + r] += v * bi;
        }
    }
    __shared__ double red[2 * R][kDotThreads / 64];
#pragma unroll
    for (int r = 0; r < 2 * R; ++r) {
        double v = acc[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[r][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 2 * R) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < kDotThreads / 64; ++k) v += red[threadIdx.x][k];
        // partial layout: [block][2 m], first the m products with a, then the m with b
        const int r = threadIdx.x % R, which = threadIdx.x / R;
        partial[(size_t)blockIdx.x * 2 * m + which * m + row0 + r] = v;
    }
}

template <int R>
__global__ __launch_bounds__(256) void pair_update_kernel(double* __restrict__ a, double* __restrict__ b,
                                                          const double* __restrict__ V, size_t ldv, int row0,
                                                          const double* __restrict__ ha, const double* __restrict__ hb,
                                                          size_t n, int last, double sa, double cross, double sb) {
    double ca[R > 0 ? R : 1], cb[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) { ca[r] = ha[row0 + r]; cb[r] = hb[row0 + r]; }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double x = a[i], y = b[i];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double v = V[(size_t)(row0 + r) * ldv + i];
            x -= ca[r] * v;
            y -= cb[r] * v;
        }
        if (last) {   // the scalings belong after the last batch of rows
            x *= sa;
            y = (y - cross * x) * sb;
        }
        a[i] = x;
        b[i] = y;
    }
}

template <int R>
static void launch_dot2(const double* V, size_t ldv, int row0, const double* a, const double* b, size_t n, double* partial,
                        int m, hipStream_t st) {
    hipLaunchKernelGGL((multi_dot2_kernel<R>), dim3(kDotBlocks), dim3(kDotThreads), 0, st, V, ldv, row0, a, b, n, partial, m);
}
template <int R>
static void launch_pair(double* a, double* b, const double* V, size_t ldv, int row0, const double* ha, const double* hb,
                        size_t n, int last, double sa, double cross, double sb, hipStream_t st) {
    const size_t want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192 ? (want ? want : 1) : 8192);
    hipLaunchKernelGGL((pair_update_kernel<R>), dim3(grid), dim3(256), 0, st, a, b, V, ldv, row0, ha, hb, n, last, sa, cross, sb);
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

size_t wx_multi_dot_workspace(int m) { return (size_t)kDotBlocks * (m > 0 ? m : 1); }   // (wx_multi_dot2: pass 2 m)

wx_status wx_multi_dot(const double* V, size_t ldv, int m, const double* w, size_t n, double* out, double* workspace,
                       wx_stream stream) {
    if (m <= 0) return WX_OK;
    if (!V || !w || !out || !workspace) return fail(WX_ERR_INVALID, "wx_multi_dot: null argument");
    if (ldv < n) return fail(WX_ERR_INVALID, "wx_multi_dot: row stride %zu shorter than the vectors (%zu)", ldv, n);
    WX_STREAM(st, stream);
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

wx_status wx_multi_dot2(const double* V, size_t ldv, int m, const double* a, const double* b, size_t n, double* out,
                        double* workspace, wx_stream stream) {
    if (m <= 0) return WX_OK;
    if (!V || !a || !b || !out || !workspace) return fail(WX_ERR_INVALID, "wx_multi_dot2: null argument");
    if (ldv < n) return fail(WX_ERR_INVALID, "wx_multi_dot2: row stride %zu shorter than the vectors (%zu)", ldv, n);
    WX_STREAM(st, stream);
    int r = 0;
    for (; r + kRowsPerPass2 <= m; r += kRowsPerPass2) launch_dot2<kRowsPerPass2>(V, ldv, r, a, b, n, workspace, m, st);
    switch (m - r) {
        case 1: launch_dot2<1>(V, ldv, r, a, b, n, workspace, m, st); break;
        case 2: launch_dot2<2>(V, ldv, r, a, b, n, workspace, m, st); break;
        case 3: launch_dot2<3>(V, ldv, r, a, b, n, workspace, m, st); break;
        default: break;
    }
    hipLaunchKernelGGL(multi_dot_finish_kernel, dim3(2 * m), dim3(64), 0, st, workspace, kDotBlocks, 2 * m, out);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_pair_update(double* a, double* b, const double* V, size_t ldv, int m, const double* ha, const double* hb,
                         size_t n, double scale_a, double cross, double scale_b, wx_stream stream) {
    if (n == 0) return WX_OK;
    if (!a || !b || (m > 0 && (!V || !ha || !hb))) return fail(WX_ERR_INVALID, "wx_pair_update: null argument");
    if (m > 0 && ldv < n) return fail(WX_ERR_INVALID, "wx_pair_update: row stride %zu shorter than the vectors (%zu)", ldv, n);
    WX_STREAM(st, stream);
    int r = 0;
    for (; r + kRowsPerPass2 < m; r += kRowsPerPass2)   // (strictly less: the last batch carries the scalings)
        launch_pair<kRowsPerPass2>(a, b, V, ldv, r, ha, hb, n, 0, 1.0, 0.0, 1.0, st);
    switch (m - r) {
        case 0: launch_pair<0>(a, b, V, ldv, r, ha, hb, n, 1, scale_a, cross, scale_b, st); break;
        case 1: launch_pair<1>(a, b, V, ldv, r, ha, hb, n, 1, scale_a, cross, scale_b, st); break;
        case 2: launch_pair<2>(a, b, V, ldv, r, ha, hb, n, 1, scale_a, cross, scale_b, st); break;
        case 3: launch_pair<3>(a, b, V, ldv, r, ha, hb, n, 1, scale_a, cross, scale_b, st); break;
        default: launch_pair<4>(a, b, V, ldv, r, ha, hb, n, 1, scale_a, cross, scale_b, st); break;
    }
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_multi_axpy(double* w, const double* V, size_t ldv, int m, const double* h, size_t n, wx_stream stream) {
    if (m <= 0 || n == 0) return WX_OK;
    if (!V || !w || !h) return fail(WX_ERR_INVALID, "wx_multi_axpy: null argument");
    if (ldv < n) return fail(WX_ERR_INVALID, "wx_multi_axpy: row stride %zu shorter than the vectors (%zu)", ldv, n);
    WX_STREAM(st, stream);
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
