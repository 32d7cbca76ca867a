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
#include <atomic>
#include <cstdlib>

#include "wx_common.h"


namespace wx {

constexpr int kDotBlocks = 2048;   // partial sums per row, at most (the workspaces are sized for it)
// ... and no more workgroups than the vector has work for: each covers at least 256 components (one per thread).  At the sizes
// of the shipped .ini files (30-100 k values per vector) that is 120-400 partials per product instead of 2048 - the launch of
// 2048 mostly empty workgroups and the finish over their partials weighed on an FGMRES iteration there
// (profiles/r05_fgmres_ini.txt); from 0.5 M components on it is the 2048 of before
static inline int dot_blocks(size_t n) {
    const size_t want = (n + 255) / 256;
    return (int)(want < 1 ? 1 : (want < (size_t)kDotBlocks ? want : (size_t)kDotBlocks));
}
constexpr int kDotThreads = 256;
// Rows per pass.  Every pass re-reads the vector(s) the rows are applied to (and the update kernels re-write them), so a
// basis of j rows costs j + ceil(j / R) (+ writes) vector sweeps: at R = 4 the two-vector kernels of an FGMRES cycle of
// 20 moved 38 vectors per Krylov vector on average, at R = 16 they move 26 (round 3; profiles/r03_fgmres_*).
constexpr int kRowsPerPass = 16;    // accumulators per thread
constexpr int kRowsPerPass2 = 16;   // rows per pass of the two-vector kernels (2 accumulators / coefficients per row)

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
                                                         int row0, const double* __restrict__ h, size_t n, double scale) {
    double c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) c[r] = h[row0 + r];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double s = w[i];
#pragma unroll
        for (int r = 0; r < R; ++r) s -= c[r] * V[(size_t)(row0 + r) * ldv + i];
        w[i] = s * scale;   // (1.0 on every pass but the last of wx_multi_axpy_scaled: exact)
    }
}

// Augmented update of the phi-function Krylov methods (solvers/kiops.py:170-173, pmex.py:160-163) as one streaming
// kernel: V[j, :n] = aw + uflip (n x p, row-major) @ V[j-1, n:n+p];  V[j, n:] = V[j-1, n+1:], 0.
// (torch.addmv hands the n x p product to a rocBLAS gemv that runs at a fraction of the streaming rate for p of 1-4.)
__global__ __launch_bounds__(256) void aug_update_kernel(double* __restrict__ V, size_t ldv, int j, size_t n, int p,
                                                         const double* __restrict__ aw, const double* __restrict__ uflip) {
    __shared__ double aug[16];
    double* vj = V + (size_t)j * ldv;
    const double* vp = V + (size_t)(j - 1) * ldv;
    if ((int)threadIdx.x < p) aug[threadIdx.x] = vp[n + threadIdx.x];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double w = aw[i];
        for (int k = 0; k < p; ++k) w += uflip[i * p + k] * aug[k];
        vj[i] = w;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < p) vj[n + threadIdx.x] = (int)threadIdx.x + 1 < p ? aug[threadIdx.x + 1] : 0.0;
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
                                                          size_t n, int last, double sa, double cross, double sb,
                                                          const double* __restrict__ dsc) {
    if (dsc != nullptr) { sa = dsc[0]; cross = dsc[1]; sb = dsc[2]; }   // (the device pass of fgmres: scalings left by the step kernel)
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

// One Krylov vector of KIOPS finished in three short launches, for vectors so short that the generic path (solvers/
// kiops.py:170-207 as array expressions: a rocBLAS gemv for the augmented update alone takes 70 us at n = 17 k, the
// 2048-partial reductions of wx_multi_dot another 9 us each) costs far more than the arithmetic.  Row j of V
// (n + p doubles per row, row stride ldv), G = ceil((n + p) / 1024) workgroups, partial sums in a caller workspace:
//   a) V[j][:n] = aw + uflip (n x p, row-major) @ V[j-1][n:];  V[j][n:] = V[j-1][n+1:], 0;
//      partial <V[r], V[j]>, max(0, j - iop) <= r < j
//   b) h[r] = sum of the partials (every workgroup, same order);  V[j] -= sum_r h[r] V[r];  partial |V[j]|^2
//   c) hcol[r] = h[r];  hcol[j] = |V[j]|;  V[j] /= hcol[j]           (hcol = column j-1 of the Hessenberg matrix)
constexpr int kFinishThreads = 256;
constexpr int kFinishChunk = 1024;   // elements per workgroup
constexpr int kFinishMaxIop = 4;
constexpr int kFinishMaxP = 16;

__device__ __forceinline__ double wg_sum256(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();   // red is free again
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(kFinishThreads) void kiops_finish_a(double* __restrict__ V, size_t ldv, int j, size_t n, int p,
                                                                 int iop, const double* __restrict__ aw,
                                                                 const double* __restrict__ uflip, double* __restrict__ part) {
    __shared__ double red[4];
    __shared__ double aug[kFinishMaxP];
    double* vj = V + (size_t)j * ldv;
    const double* vp = V + (size_t)(j - 1) * ldv;
    if (threadIdx.x < p) aug[threadIdx.x] = vp[n + threadIdx.x];
    __syncthreads();
    const int ilow = j - iop > 0 ? j - iop : 0, nr = j - ilow;
    const size_t len = n + (size_t)p, lo = (size_t)blockIdx.x * kFinishChunk;
    double acc[kFinishMaxIop] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int it = 0; it < kFinishChunk / kFinishThreads; ++it) {
        const size_t i = lo + it * kFinishThreads + threadIdx.x;
        if (i < len) {
            double w;
            if (i < n) {
                w = aw[i];
                for (int k = 0; k < p; ++k) w += uflip[i * p + k] * aug[k];
            } else {
                const int k = (int)(i - n);
                w = k + 1 < p ? aug[k + 1] : 0.0;
            }
            vj[i] = w;
            for (int r = 0; r < nr; ++r) acc[r] += V[(size_t)(ilow + r) * ldv + i] * w;
        }
    }
    for (int r = 0; r < nr; ++r) {
        const double t = wg_sum256(acc[r], red);
        if (threadIdx.x == 0) part[(size_t)blockIdx.x * (kFinishMaxIop + 1) + r] = t;
    }
}

__global__ __launch_bounds__(kFinishThreads) void kiops_finish_b(double* __restrict__ V, size_t ldv, int j, size_t n, int p,
                                                                 int iop, double* __restrict__ part, double* __restrict__ hcol) {
    __shared__ double red[4];
    double* vj = V + (size_t)j * ldv;
    const int ilow = j - iop > 0 ? j - iop : 0, nr = j - ilow;
    double h[kFinishMaxIop] = {0.0, 0.0, 0.0, 0.0};
    // every workgroup sums all partials itself, one per thread (gridDim.x <= 256), in the same tree: same h everywhere
    for (int r = 0; r < nr; ++r)
        h[r] = wg_sum256(threadIdx.x < gridDim.x ? part[(size_t)threadIdx.x * (kFinishMaxIop + 1) + r] : 0.0, red);
    const size_t len = n + (size_t)p, lo = (size_t)blockIdx.x * kFinishChunk;
    double nn = 0.0;
#pragma unroll
    for (int it = 0; it < kFinishChunk / kFinishThreads; ++it) {
        const size_t i = lo + it * kFinishThreads + threadIdx.x;
        if (i < len) {
            double w = vj[i];
            for (int r = 0; r < nr; ++r) w -= h[r] * V[(size_t)(ilow + r) * ldv + i];
            vj[i] = w;
            nn += w * w;
        }
    }
    const double t = wg_sum256(nn, red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * (kFinishMaxIop + 1) + kFinishMaxIop] = t;
    if (blockIdx.x == 0 && (int)threadIdx.x < nr) hcol[ilow + threadIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(kFinishThreads) void kiops_finish_c(double* __restrict__ V, size_t ldv, int j, size_t n, int p,
                                                                 const double* __restrict__ part, double* __restrict__ hcol) {
    __shared__ double red[4];
    double* vj = V + (size_t)j * ldv;
    const double nrm =
        sqrt(wg_sum256(threadIdx.x < gridDim.x ? part[(size_t)threadIdx.x * (kFinishMaxIop + 1) + kFinishMaxIop] : 0.0, red));
    const size_t len = n + (size_t)p, lo = (size_t)blockIdx.x * kFinishChunk;
#pragma unroll
    for (int it = 0; it < kFinishChunk / kFinishThreads; ++it) {
        const size_t i = lo + it * kFinishThreads + threadIdx.x;
        if (i < len) vj[i] /= nrm;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) hcol[j] = nrm;
}

// The same Krylov vector for LONG vectors (E7: 442 M doubles), as three streaming kernels whose reductions leave through
// small device buffers, so that the caller can complete them (all-reduce over ranks, the p replicated augmented
// components) between the stages:
//   long_a  V[j][:n] = aw + uflip @ V[j-1][n:];  V[j][n:] = V[j-1][n+1:], 0;  partial <V[r][:n], V[j][:n]>, ilow <= r < j
//   long_b  V[j][:] -= sum_r h[r] V[r][:]  (h: device, the caller's completed products);  partial |V[j][:n]|^2
//   long_c  V[j][:] /= sqrt(*nrm2)         (nrm2: device, the caller's completed squared norm);  hcol[j] = that root
// Passes over memory per vector (p = 1, iop = 2): 5 + 4 + 2 = 11 vector sweeps, against 14 and a tall-skinny rocBLAS gemv
// for the array-expression form; deterministic two-stage reductions (multi_dot_finish_kernel sums the partials).
constexpr int kLongBlocks = 2048;

// scales (nullable): lazy normalisation - row r of the basis holds its n-long part UNSCALED, scales[r] = 1 / |V[r]| is
// applied where the row is used (its augmented components ARE kept scaled: p values); aw is then A applied to the unscaled
// row j-1.  One multiply per use instead of a read-modify-write sweep per Krylov vector (kiops_long_c).
__global__ __launch_bounds__(kFinishThreads) void kiops_long_a(double* __restrict__ V, size_t ldv, int j, size_t n, int p,
                                                               int iop, const double* __restrict__ aw,
                                                               const double* __restrict__ uflip, double* __restrict__ part,
                                                               const double* __restrict__ scales) {
    __shared__ double red[4];
    __shared__ double aug[kFinishMaxP];
    double* vj = V + (size_t)j * ldv;
    const double* vp = V + (size_t)(j - 1) * ldv;
    if (threadIdx.x < p) aug[threadIdx.x] = vp[n + threadIdx.x];
    __syncthreads();
    const int ilow = j - iop > 0 ? j - iop : 0, nr = j - ilow;
    const double sa = scales ? scales[j - 1] : 1.0;
    double acc[kFinishMaxIop] = {0.0, 0.0, 0.0, 0.0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double w;
        if (aw) {   // (null: the n-long part has been formed by the matvec's own store, wx_kiops_long_a_formed)
            w = scales ? sa * aw[i] : aw[i];
            for (int k = 0; k < p; ++k) w += uflip[i * p + k] * aug[k];
            vj[i] = w;
        } else {
            w = vj[i];
        }
        for (int r = 0; r < nr; ++r) acc[r] += V[(size_t)(ilow + r) * ldv + i] * w;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < p) vj[n + threadIdx.x] = (int)threadIdx.x + 1 < p ? aug[threadIdx.x + 1] : 0.0;
    for (int r = 0; r < kFinishMaxIop; ++r) {
        double t = wg_sum256(r < nr ? acc[r] : 0.0, red);
        if (scales && r < nr) t *= scales[ilow + r];
        if (threadIdx.x == 0) part[(size_t)blockIdx.x * kFinishMaxIop + r] = t;
    }
}

__global__ __launch_bounds__(kFinishThreads) void kiops_long_b(double* __restrict__ V, size_t ldv, int j, size_t n, int p,
                                                               int iop, const double* __restrict__ h, double* __restrict__ part,
                                                               const double* __restrict__ scales) {
    __shared__ double red[4];
    double* vj = V + (size_t)j * ldv;
    const int ilow = j - iop > 0 ? j - iop : 0, nr = j - ilow;
    double c[kFinishMaxIop] = {0.0, 0.0, 0.0, 0.0}, cs[kFinishMaxIop] = {0.0, 0.0, 0.0, 0.0};
    for (int r = 0; r < nr; ++r) {
        c[r] = h[r];
        cs[r] = scales ? h[r] * scales[ilow + r] : h[r];   // on the n-long (unscaled) part of row r
    }
    const size_t len = n + (size_t)p, stride = (size_t)gridDim.x * blockDim.x;
    double nn = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
        double w = vj[i];
        if (i < n) {
            for (int r = 0; r < nr; ++r) w -= cs[r] * V[(size_t)(ilow + r) * ldv + i];
        } else {
            for (int r = 0; r < nr; ++r) w -= c[r] * V[(size_t)(ilow + r) * ldv + i];
        }
        vj[i] = w;
        if (i < n) nn += w * w;
    }
    const double t = wg_sum256(nn, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// the products arrive as pairs of partial sums (from the matvec's own store: 172 800 pairs for the E7 sphere): summed in a fixed
// order in two steps - kFinishStage1 workgroups per product over contiguous chunks, then one workgroup over their results, which
// also applies the rows' scales and shifts the augmented components of row j  (one workgroup alone took 0.28 ms)
constexpr int kFinishStage1 = 128;
__global__ __launch_bounds__(256) void kiops_long_a_finish1(const double* __restrict__ part, size_t nblocks, double* __restrict__ tmp) {
    __shared__ double red[4];
    const int r = blockIdx.y;
    const size_t chunk = (nblocks + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * chunk;
    const size_t hi = lo + chunk < nblocks ? lo + chunk : nblocks;
    double v = 0.0;
    for (size_t b = lo + threadIdx.x; b < hi; b += 256) v += part[2 * b + r];
    const double t = wg_sum256(v, red);
    if (threadIdx.x == 0) tmp[(size_t)r * gridDim.x + blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void kiops_long_a_finish_kernel(double* __restrict__ V, size_t ldv, int j, size_t n, int p,
                                                                   int ilow, const double* __restrict__ tmp,
                                                                   double* __restrict__ dots, const double* __restrict__ scales) {
    __shared__ double red[4];
    const int r = blockIdx.x;
    const double t = wg_sum256((int)threadIdx.x < kFinishStage1 ? tmp[(size_t)r * kFinishStage1 + threadIdx.x] : 0.0, red);
    if (threadIdx.x == 0) dots[r] = scales ? t * scales[ilow + r] : t;
    if (r == 0 && (int)threadIdx.x < p) {
        const double* vp = V + (size_t)(j - 1) * ldv;
        V[(size_t)j * ldv + n + threadIdx.x] = (int)threadIdx.x + 1 < p ? vp[n + threadIdx.x + 1] : 0.0;
    }
}

// the subtraction and the norm of row j done by the NEXT product's tangent-extrapolation kernel (euler_tan_extrap_kernel<N, true>:
// one partial squared norm per workgroup): summed here in a fixed order in two steps, and the p augmented components of the
// row corrected with the same coefficients (kiops_long_b's order)
__global__ __launch_bounds__(256) void kiops_fold_sum1(const double* __restrict__ part, size_t nblocks, double* __restrict__ tmp) {
    __shared__ double red[4];
    const size_t chunk = (nblocks + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * chunk;
    const size_t hi = lo + chunk < nblocks ? lo + chunk : nblocks;
    double v = 0.0;
    for (size_t b = lo + threadIdx.x; b < hi; b += 256) v += part[b];
    const double t = wg_sum256(v, red);
    if (threadIdx.x == 0) tmp[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void kiops_fold_finish_kernel(double* __restrict__ V, size_t ldv, int j, size_t n, int p, int ilow,
                                                                 int nr, const double* __restrict__ h, const double* __restrict__ tmp,
                                                                 double* __restrict__ nrm2) {
    __shared__ double red[4];
    const double t = wg_sum256((int)threadIdx.x < kFinishStage1 ? tmp[threadIdx.x] : 0.0, red);
    if (threadIdx.x == 0) *nrm2 = t;
    if ((int)threadIdx.x < p) {
        double w = V[(size_t)j * ldv + n + threadIdx.x];
        for (int r = 0; r < nr; ++r) w -= h[r] * V[(size_t)(ilow + r) * ldv + n + threadIdx.x];
        V[(size_t)j * ldv + n + threadIdx.x] = w;
    }
}

// lazy normalisation: the augmented components of row j scaled, hcol[j] = |V[j]|, scales[j] = 1 / |V[j]|; the n-long part stays
__global__ void kiops_long_c_lazy(double* __restrict__ V, size_t ldv, int j, size_t n, int p, const double* __restrict__ nrm2,
                                  double* __restrict__ hcol, double* __restrict__ scales) {
    const double nrm = sqrt(*nrm2);
    if ((int)threadIdx.x < p) V[(size_t)j * ldv + n + threadIdx.x] /= nrm;
    if (threadIdx.x == 0) {
        hcol[j] = nrm;
        scales[j] = 1.0 / nrm;
    }
}

__global__ __launch_bounds__(kFinishThreads) void kiops_long_c(double* __restrict__ V, size_t ldv, int j, size_t len,
                                                               const double* __restrict__ nrm2, double* __restrict__ hcol) {
    double* vj = V + (size_t)j * ldv;
    const double nrm = sqrt(*nrm2);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) vj[i] /= nrm;
    if (blockIdx.x == 0 && threadIdx.x == 0) hcol[j] = nrm;
}

template <int R>
static void launch_dot2(const double* V, size_t ldv, int row0, const double* a, const double* b, size_t n, double* partial,
                        int m, hipStream_t st, int blocks = 0) {
    hipLaunchKernelGGL((multi_dot2_kernel<R>), dim3(blocks > 0 ? blocks : dot_blocks(n)), dim3(kDotThreads), 0, st, V, ldv, row0, a, b, n,
                       partial, m);
}
template <int R>
static void launch_pair(double* a, double* b, const double* V, size_t ldv, int row0, const double* ha, const double* hb,
                        size_t n, int last, double sa, double cross, double sb, hipStream_t st, const double* dsc = nullptr) {
    const size_t want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192 ? (want ? want : 1) : 8192);
    hipLaunchKernelGGL((pair_update_kernel<R>), dim3(grid), dim3(256), 0, st, a, b, V, ldv, row0, ha, hb, n, last, sa, cross, sb, dsc);
}

template <int R>
static void launch_dot(const double* V, size_t ldv, int row0, const double* w, size_t n, double* partial, int m,
                       hipStream_t st) {
    hipLaunchKernelGGL((multi_dot_kernel<R>), dim3(dot_blocks(n)), dim3(kDotThreads), 0, st, V, ldv, row0, w, n, partial, m);
}
template <int R>
static void launch_axpy(double* w, const double* V, size_t ldv, int row0, const double* h, size_t n, double scale,
                        hipStream_t st) {
    const size_t want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192 ? (want ? want : 1) : 8192);
    hipLaunchKernelGGL((multi_axpy_kernel<R>), dim3(grid), dim3(256), 0, st, w, V, ldv, row0, h, n, scale);
}

// launch the instantiation for `rem` rows (1 <= rem <= R)
template <int R>
static void dispatch_dot(int rem, const double* V, size_t ldv, int row0, const double* w, size_t n, double* partial, int m,
                         hipStream_t st) {
    if (rem == R) launch_dot<R>(V, ldv, row0, w, n, partial, m, st);
    else if constexpr (R > 1) dispatch_dot<R - 1>(rem, V, ldv, row0, w, n, partial, m, st);
}
template <int R>
static void dispatch_axpy(int rem, double* w, const double* V, size_t ldv, int row0, const double* h, size_t n, double scale,
                          hipStream_t st) {
    if (rem == R) launch_axpy<R>(w, V, ldv, row0, h, n, scale, st);
    else if constexpr (R > 1) dispatch_axpy<R - 1>(rem, w, V, ldv, row0, h, n, scale, st);
}
template <int R>
static void dispatch_dot2(int rem, const double* V, size_t ldv, int row0, const double* a, const double* b, size_t n,
                          double* partial, int m, hipStream_t st, int blocks = 0) {
    if (rem == R) launch_dot2<R>(V, ldv, row0, a, b, n, partial, m, st, blocks);
    else if constexpr (R > 1) dispatch_dot2<R - 1>(rem, V, ldv, row0, a, b, n, partial, m, st, blocks);
}
template <int R>
static void dispatch_pair(int rem, double* a, double* b, const double* V, size_t ldv, int row0, const double* ha,
                          const double* hb, size_t n, int last, double sa, double cross, double sb, hipStream_t st,
                          const double* dsc = nullptr) {
    if (rem == R) launch_pair<R>(a, b, V, ldv, row0, ha, hb, n, last, sa, cross, sb, st, dsc);
    else if constexpr (R > 0) dispatch_pair<R - 1>(rem, a, b, V, ldv, row0, ha, hb, n, last, sa, cross, sb, st, dsc);
}


// ---- one Krylov vector of PMEX (solvers/pmex.py:157-233) with no host round trip: the augmented update, the (j+1) x 2
// block of products (multi_dot2), the projector's coefficients and the norm estimate in a one-workgroup kernel, the
// correction (normalised in the same pass when the estimate stands), the vector's own norm where it does not, the
// Hessenberg column.  The host reads the columns of a whole pass afterwards, as it does for KIOPS.
constexpr int kPmexMaxM = 128;

// error-free transformations for the sum of squares the reference accumulates in the platform's extended precision
__device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) {
    s = a + b;
    const double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}

// G: [<v_k, v_{j-1}>, k <= j] then [<v_k, w>, k <= j] (2 (j+1) doubles).  LT / Linv: ld x ld row-major, persistent over the
// solve (LT strictly upper: column c = products of v_c with the older vectors; Linv = (I + LT^T)^{-1}, unit lower).
// sol[0:j] = g - LT (Linv g);  hcol[0:j] = sol;  scal[0] = factor for the correction pass (1 / norm estimate, or 1),
// scal[1] = the estimate (-1: the difference came out negative), scal[2] = 1 when the estimate stands.
// ST: the triangles the three products below walk - Linv's lower one, LT's upper one, j x j - staged in LDS by the whole
// workgroup first (j <= kPmexStage): each product is a thread's OWN sequential sum, whose every term was a dependent round trip
// to memory (20 us of a 114 us Krylov vector at the shipped .ini sizes); the sums keep their order, the results their bits.
constexpr int kPmexStage = 64;
template <bool ST>
__device__ __forceinline__ void pmex_project_body(const double* __restrict__ G, int j, double* LT, double* Linv, int ld, double tol,
                                                  double* __restrict__ sol, double* __restrict__ hcol,
                                                  double* __restrict__ scal) {
    __shared__ double g[kPmexMaxM], t[kPmexMaxM], c[kPmexMaxM];
    __shared__ double sL[ST ? kPmexStage * (kPmexStage + 1) : 1], sU[ST ? kPmexStage * (kPmexStage + 1) : 1];
    constexpr int SS = kPmexStage + 1;
    const int tid = threadIdx.x, bs = blockDim.x;
    const double* g0 = G;            // products with v_{j-1}
    const double* g1 = G + (j + 1);  // products with the new vector
    for (int k = tid; k < j; k += bs) g[k] = g1[k];
    if constexpr (ST) {   // (batches of eight loads in flight)
        for (int base = tid; base < j * j; base += 8 * bs) {
            double tl[8], tu[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * bs, r = idx / j, k = idx - r * j;
                const bool in = idx < j * j;
                // (not what this step itself writes below: row j-1 of Linv left of its diagonal, column j-1 of LT)
                tl[u] = (in && k <= r && (r < j - 1 || k == r)) ? Linv[(size_t)r * ld + k] : 0.0;
                tu[u] = (in && k > r && k < j - 1) ? LT[(size_t)r * ld + k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * bs, r = idx / j, k = idx - r * j;
                if (idx < j * j) {
                    if (k <= r) { if (r < j - 1 || k == r) sL[r * SS + k] = tl[u]; }
                    else if (k < j - 1) sU[r * SS + k] = tu[u];
                }
            }
        }
    }
    if (j > 1) {
        for (int k = tid; k < j - 1; k += bs) {
            c[k] = g0[k];
            LT[(size_t)k * ld + (j - 1)] = g0[k];
            if constexpr (ST) sU[k * SS + (j - 1)] = g0[k];
        }
        __syncthreads();
        // row j-1 of Linv: -c^T Linv[0:j-1, 0:j-1]  (Linv unit lower triangular: rows i >= k contribute to column k)
        for (int k = tid; k < j - 1; k += bs) {
            double a = 0.0;
            for (int i = k; i < j - 1; ++i) a += c[i] * (ST ? sL[i * SS + k] : Linv[(size_t)i * ld + k]);
            Linv[(size_t)(j - 1) * ld + k] = -a;
            if constexpr (ST) sL[(j - 1) * SS + k] = -a;
        }
    }
    __syncthreads();
    for (int r = tid; r < j; r += bs) {
        double a = 0.0;
        for (int k = 0; k <= r; ++k) a += (ST ? sL[r * SS + k] : Linv[(size_t)r * ld + k]) * g[k];
        t[r] = a;
    }
    __syncthreads();
    for (int r = tid; r < j; r += bs) {
        double a = 0.0;
        for (int cc = r + 1; cc < j; ++cc) a += (ST ? sU[r * SS + cc] : LT[(size_t)r * ld + cc]) * t[cc];
        const double v = g[r] - a;
        sol[r] = v;
        hcol[r] = v;
    }
    if (tid == 0) {
        double hi = 0.0, lo = 0.0;   // sum of g_k^2 in double-double
        for (int k = 0; k < j; ++k) {
            const double pk = g[k] * g[k];
            const double ek = fma(g[k], g[k], -pk);
            double sk, e2;
            two_sum(hi, pk, sk, e2);
            lo += ek + e2;
            hi = sk;
        }
        double d, e;
        two_sum(g1[j], -hi, d, e);
        const double diff = d + (e - lo);   // <w, w> - sum g_k^2
        const bool stands = !(diff < 0.0);
        const double est = stands ? sqrt(diff) : -1.0;
        scal[0] = (stands && est >= tol) ? 1.0 / est : 1.0;
        scal[1] = est;
        scal[2] = stands ? 1.0 : 0.0;
    }
}
__device__ __forceinline__ void pmex_project(const double* __restrict__ G, int j, double* LT, double* Linv, int ld, double tol,
                                             double* __restrict__ sol, double* __restrict__ hcol, double* __restrict__ scal) {
    if (j <= kPmexStage) pmex_project_body<true>(G, j, LT, Linv, ld, tol, sol, hcol, scal);   // (uniform over the launch)
    else pmex_project_body<false>(G, j, LT, Linv, ld, tol, sol, hcol, scal);
}
__global__ __launch_bounds__(256) void pmex_project_kernel(const double* __restrict__ G, int j, double* __restrict__ LT,
                                                           double* __restrict__ Linv, int ld, double tol,
                                                           double* __restrict__ sol, double* __restrict__ hcol,
                                                           double* __restrict__ scal) {
    pmex_project(G, j, LT, Linv, ld, tol, sol, hcol, scal);
}

// ---- the same vector in FOUR launches instead of nine to thirteen (one rank; the sizes of the shipped .ini files, where a
// Krylov vector is a chain of dependent 5-10 us launches): each kernel below performs exactly the floating-point operations
// of the kernels it replaces, in their order; the products are summed over fewer, larger groups (one workgroup of partials per
// 256 components instead of always 2048), so the Hessenberg columns agree with the separate launches' to rounding and PMEX's
// decisions - checked against the reference's own statistics - are the same (tests/test_pmex_gpu.py, test_sw_gpu.py)
//   pmex_aug_dot2_kernel        aug_update_kernel + every multi_dot2_kernel pass (the new row is formed on the fly, and dotted)
//   pmex_finish_project_kernel  multi_dot_finish_kernel + pmex_project_kernel
//   pmex_axpy_all_kernel        every multi_axpy_dev_kernel pass
//   pmex_finish_scale_kernel    pmex_finish_kernel + scale_if_kernel
template <int R>
__global__ __launch_bounds__(kDotThreads) void pmex_aug_dot2_kernel(double* __restrict__ V, size_t ldv, int j, size_t n, int p,
                                                                    const double* __restrict__ aw,
                                                                    const double* __restrict__ uflip,
                                                                    double* __restrict__ partial) {
    __shared__ double aug[16];
    __shared__ double red[2 * R][kDotThreads / 64];
    const int m = j + 1;
    double* vj = V + (size_t)j * ldv;
    const double* vp = V + (size_t)(j - 1) * ldv;
    if ((int)threadIdx.x < p) aug[threadIdx.x] = vp[n + threadIdx.x];
    __syncthreads();
    const size_t len = n + (size_t)p, stride = (size_t)gridDim.x * blockDim.x;
    for (int row0 = 0; row0 < m; row0 += R) {
        const int nr = m - row0 < R ? m - row0 : R;
        double acc[2 * R];
#pragma unroll
        for (int r = 0; r < 2 * R; ++r) acc[r] = 0.0;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
            const double ai = vp[i];
            double bi;
            if (i < n) {   // (aug_update_kernel's expression)
                bi = aw[i];
                for (int k = 0; k < p; ++k) bi += uflip[i * p + k] * aug[k];
            } else {
                const int t = (int)(i - n);
                bi = t + 1 < p ? aug[t + 1] : 0.0;
            }
            if (row0 == 0) vj[i] = bi;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r < nr) {
                    const double v = row0 + r == j ? bi : V[(size_t)(row0 + r) * ldv + i];
                    acc[r] += v * ai;
                    acc[R + r] += v * bi;
                }
            }
        }
        if (row0 > 0) __syncthreads();   // the previous chunk's partials have been read out of `red`
#pragma unroll
        for (int r = 0; r < 2 * R; ++r) {
            double v = acc[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0) red[r][threadIdx.x >> 6] = v;
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * R) {
            const int r = threadIdx.x % R, which = threadIdx.x / R;
            if (r < nr) {
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < kDotThreads / 64; ++k) v += red[threadIdx.x][k];
                partial[(size_t)blockIdx.x * 2 * m + which * m + row0 + r] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void pmex_finish_project_kernel(const double* __restrict__ partial, int blocks, int j,
                                                                  double* __restrict__ LT, double* __restrict__ Linv, int ld,
                                                                  double tol, double* __restrict__ sol, double* __restrict__ hcol,
                                                                  double* __restrict__ scal, double* __restrict__ G) {
    const int m2 = 2 * (j + 1), wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = wave; k < m2; k += 4) {   // (multi_dot_finish_kernel: one wave per product, the same order)
        double v = 0.0;
        for (int b = lane; b < blocks; b += 64) v += partial[(size_t)b * m2 + k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) G[k] = v;
    }
    __threadfence_block();
    __syncthreads();
    pmex_project(G, j, LT, Linv, ld, tol, sol, hcol, scal);
}

__global__ __launch_bounds__(256) void pmex_axpy_all_kernel(double* __restrict__ w, const double* __restrict__ V, size_t ldv, int j,
                                                            const double* __restrict__ h, size_t n,
                                                            const double* __restrict__ scale, double* __restrict__ part,
                                                            size_t nnorm) {
    __shared__ double red[4];
    __shared__ double cf[kPmexMaxM];
    for (int r = threadIdx.x; r < j; r += blockDim.x) cf[r] = h[r];
    __syncthreads();
    const double sc = *scale;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double nn = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double s = w[i];
#pragma unroll 8
        for (int r = 0; r < j; ++r) s -= cf[r] * V[(size_t)r * ldv + i];
        s *= sc;
        w[i] = s;
        if (i < nnorm) nn += s * s;
    }
    const double tsum = wg_sum256(nn, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tsum;
}

__global__ __launch_bounds__(256) void pmex_finish_scale_kernel(double* __restrict__ w, size_t n, const double* __restrict__ part,
                                                                int nblocks, double tol, double* __restrict__ scal,
                                                                double* __restrict__ hnorm, double* __restrict__ own) {
    __shared__ double red[4];
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) v += part[b];
    const double total = wg_sum256(v, red);   // (every workgroup sums the partials itself, in pmex_finish_kernel's order)
    const bool stands = scal[2] != 0.0;
    const double nrm = stands ? scal[1] : sqrt(total);
    const double f = (!stands && nrm >= tol) ? 1.0 / nrm : 1.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *hnorm = nrm;
        *own = stands ? 0.0 : 1.0;
        scal[3] = f;
    }
    if (f == 1.0) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) w[i] *= f;
}

// w = (w - sum_k h[k] V[row0 + k]) * (*scale or 1);  when `part` is given, the squared norm of what was written, per workgroup
// (nnorm: the squared norm covers the first nnorm components only - the rank-local n-long part when the vectors are split
// over ranks, whose p replicated augmented components enter once, after the all-reduce)
template <int R>
__global__ __launch_bounds__(256) void multi_axpy_dev_kernel(double* __restrict__ w, const double* __restrict__ V, size_t ldv,
                                                             int row0, const double* __restrict__ h, size_t n,
                                                             const double* __restrict__ scale, double* __restrict__ part,
                                                             size_t nnorm) {
    __shared__ double red[4];
    double cf[R];
#pragma unroll
    for (int r = 0; r < R; ++r) cf[r] = h[row0 + r];
    const double sc = scale ? *scale : 1.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double nn = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double s = w[i];
#pragma unroll
        for (int r = 0; r < R; ++r) s -= cf[r] * V[(size_t)(row0 + r) * ldv + i];
        s *= sc;
        w[i] = s;
        if (i < nnorm) nn += s * s;
    }
    if (part) {
        const double tsum = wg_sum256(nn, red);
        if (threadIdx.x == 0) part[blockIdx.x] = tsum;
    }
}

// hcol[j] = the vector's norm - the estimate, or the root of the summed partials where the estimate fell;  scal[3] = the
// factor still to be applied (1 when the correction pass normalised the vector already or at a breakdown);  own[0] = 1 when
// the vector's own norm was needed (pmex's `reg_comm_nrm`)
// (part == null: the squared norm is complete in *total_in already - the several-rank form, pmex_split_norm_kernel + all-reduce)
__global__ __launch_bounds__(256) void pmex_finish_kernel(const double* __restrict__ part, int nblocks, double tol,
                                                          double* __restrict__ scal, double* __restrict__ hnorm,
                                                          double* __restrict__ own, const double* __restrict__ total_in) {
    __shared__ double red[4];
    double v = 0.0;
    if (part)
        for (int b = threadIdx.x; b < nblocks; b += blockDim.x) v += part[b];
    double total = wg_sum256(v, red);
    if (!part) total = *total_in;
    if (threadIdx.x == 0) {
        const bool stands = scal[2] != 0.0;
        const double nrm = stands ? scal[1] : sqrt(total);
        *hnorm = nrm;
        *own = stands ? 0.0 : 1.0;
        scal[3] = (!stands && nrm >= tol) ? 1.0 / nrm : 1.0;
    }
}

// several ranks: the products of the rank-local n-long parts have been all-reduced into G; the p augmented components are
// replicated on every rank and enter once, here:  G[k] += <V[k, n:], V[j-1, n:]>,  G[m + k] += <V[k, n:], V[j, n:]>,  k < m = j+1
__global__ __launch_bounds__(256) void pmex_split_aug_kernel(const double* __restrict__ V, size_t ldv, int j, size_t n, int p,
                                                             double* __restrict__ G) {
    const int m = j + 1;
    for (int t = threadIdx.x; t < 2 * m; t += blockDim.x) {
        const int k = t < m ? t : t - m;
        const double* other = V + (size_t)(t < m ? j - 1 : j) * ldv + n;
        const double* vk = V + (size_t)k * ldv + n;
        double a = 0.0;
        for (int c = 0; c < p; ++c) a += vk[c] * other[c];
        G[t] += a;
    }
}
// the rank-local squared norm of the corrected vector: the sum of the correction pass's partials, ready for the all-reduce ...
__global__ __launch_bounds__(256) void pmex_split_norm_kernel(const double* __restrict__ part, int nblocks, double* __restrict__ total) {
    __shared__ double red[4];
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) v += part[b];
    const double t = wg_sum256(v, red);
    if (threadIdx.x == 0) *total = t;
}
// ... and, after it, the replicated augmented components' share
__global__ void pmex_split_norm_aug_kernel(const double* __restrict__ vj_aug, int p, double* __restrict__ total) {
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int c = 0; c < p; ++c) a += vj_aug[c] * vj_aug[c];
        *total += a;
    }
}

__global__ __launch_bounds__(256) void scale_if_kernel(double* __restrict__ w, size_t n, const double* __restrict__ factor) {
    const double f = *factor;
    if (f == 1.0) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) w[i] *= f;
}

template <int R>
static void launch_axpy_dev(double* w, const double* V, size_t ldv, int row0, const double* h, size_t n, const double* scale,
                            double* part, unsigned grid, hipStream_t st, size_t nnorm) {
    hipLaunchKernelGGL((multi_axpy_dev_kernel<R>), dim3(grid), dim3(256), 0, st, w, V, ldv, row0, h, n, scale, part, nnorm);
}
template <int R>
static void dispatch_axpy_dev(int rem, double* w, const double* V, size_t ldv, int row0, const double* h, size_t n,
                              const double* scale, double* part, unsigned grid, hipStream_t st, size_t nnorm) {
    if (rem == R) launch_axpy_dev<R>(w, V, ldv, row0, h, n, scale, part, grid, st, nnorm);
    else if constexpr (R > 1) dispatch_axpy_dev<R - 1>(rem, w, V, ldv, row0, h, n, scale, part, grid, st, nnorm);
}

constexpr unsigned kPmexAxpyBlocks = 2048;
constexpr unsigned kPmexFusedDotBlocks = 512;
constexpr size_t kPmexFusedMaxLen = 1u << 21;   // vectors up to this length are built by the four fused launches (longer ones are
                                                // bound by the sweeps, not by the launches, and keep the row-batched kernels)

}  // namespace wx

using namespace wx;

// ---- the two vector kernels of a Krylov step at LAUNCH-BOUND lengths (fgmres' device pass at the sizes of the shipped .ini files:
// 30 k - 100 k components): any number of basis rows in ONE launch each, where the row-batched kernels above take a launch per
// 16 rows - the time of such a step is the number of its launches.
//   multi_dot2_small   partial[block][2 J]: <V[r], a>, <V[r], b>, r < J over the workgroup's chunk of 256 ept components
//                      (small_products); needs n <= gridDim.x * kSmallThreads * ept, ept <= kSmallEpt
//   pair_update_small  a -= sum_r ha[r] V[r]; b -= sum_r hb[r] V[r]; a *= dsc[0]; b = (b - dsc[1] a) dsc[2]   (m <= 64 rows)

// The products of a workgroup's chunk (kSmallThreads ept components from c0; ept <= kSmallEpt): a and b of the chunk staged in LDS,
// the ROWS dealt over the sixteen waves, a wave's row components all in flight before the first product - one wave reduction per
// row and workgroup (each wave walking every row over its own lanes' components took four, six dependent cross-lane steps each),
// and 16 loads per lane in flight (four waves with 8 each moved 1 TB/s at 100 k components: 14 us).
// red[r] = <V[r], a>, red[64 + r] = <V[r], b> over the chunk, valid after the barrier.
constexpr int kSmallThreads = 1024;
constexpr int kSmallEpt = 2;
constexpr int kSmallBlocks = 128;    // at most (the one-launch step's barrier: two words per lane of the waiting wave)
constexpr int kSmallMaxLen = kSmallBlocks * kSmallThreads * kSmallEpt;

// the rows' part: sa, sb = the chunk's components of the two vectors (staged by the caller, barrier included); own_row (or -1): a
// row that IS the second vector (PMEX forms its new row in the same launch: its chunk is sb, not yet a row in memory)
__device__ __forceinline__ void small_rows(const double* __restrict__ V, size_t ldv, int J, size_t n, int ept, const double* sa,
                                           const double* sb, double* red, int own_row) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t c0 = (size_t)blockIdx.x * kSmallThreads * ept;
    for (int r = wave; r < J; r += kSmallThreads / 64) {
        const double* row = V + (size_t)r * ldv + c0;
        const bool own = r == own_row;   // (uniform over the wave)
        double pa = 0.0, pb = 0.0;
        for (int e0 = 0; e0 < ept; ++e0) {   // (a batch: 16 components per lane)
            double v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = lane + 64 * (16 * e0 + e);
                v[e] = own ? sb[k] : (c0 + k < n ? row[k] : 0.0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = lane + 64 * (16 * e0 + e);
                pa += v[e] * sa[k];
                pb += v[e] * sb[k];
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            pa += __shfl_down(pa, off, 64);
            pb += __shfl_down(pb, off, 64);
        }
        if (lane == 0) { red[r] = pa; red[64 + r] = pb; }
    }
    __syncthreads();
}

__device__ __forceinline__ void small_products(const double* __restrict__ V, size_t ldv, int J, const double* a, const double* b,
                                               size_t n, int ept, double* sa, double* sb, double* red) {
    const int tid = threadIdx.x;
    const size_t c0 = (size_t)blockIdx.x * kSmallThreads * ept;
    for (int e = 0; e < ept; ++e) {
        const int k = tid + kSmallThreads * e;
        const size_t i = c0 + k;
        sa[k] = i < n ? a[i] : 0.0;
        sb[k] = i < n ? b[i] : 0.0;
    }
    __syncthreads();
    small_rows(V, ldv, J, n, ept, sa, sb, red, -1);
}

// components per thread such that kSmallBlocks workgroups cover n
static inline int small_ept(size_t n) {
    const size_t per = (size_t)kSmallBlocks * kSmallThreads;
    const size_t e = (n + per - 1) / per;
    return (int)(e < 1 ? 1 : e);
}
static inline int small_blocks(size_t n, int ept) {
    const size_t per = (size_t)kSmallThreads * ept;
    const size_t g = (n + per - 1) / per;
    return (int)(g < 1 ? 1 : g);
}

// ---- PMEX at launch-bound lengths (one rank, up to kSmallMaxLen components, at most 64 rows): the new row formed and the 2 (j + 1)
// products taken by the scheme above (pmex_aug_dot2_kernel walks the rows in passes of 16 with 32 accumulators per thread and as
// many wave reductions per wave: 22 us of a 114 us Krylov vector at the shipped .ini sizes), at most kSmallBlocks partial sums per
// product, which pmex_finish_project_small_kernel sums with all of them in flight before the projector (pmex_project: staged).
__global__ __launch_bounds__(kSmallThreads) void pmex_aug_products_small_kernel(double* __restrict__ V, size_t ldv, int j, size_t n,
                                                                                int p, const double* __restrict__ aw,
                                                                                const double* __restrict__ uflip, int ept,
                                                                                double* __restrict__ partial) {
    __shared__ double sa[kSmallThreads * kSmallEpt], sb[kSmallThreads * kSmallEpt], red[128];
    __shared__ double aug[16];
    const int tid = threadIdx.x, m = j + 1;
    double* vj = V + (size_t)j * ldv;
    const double* vp = V + (size_t)(j - 1) * ldv;
    if (tid < p) aug[tid] = vp[n + tid];
    __syncthreads();
    const size_t len = n + (size_t)p, c0 = (size_t)blockIdx.x * kSmallThreads * ept;
    for (int e = 0; e < ept; ++e) {
        const int k = tid + kSmallThreads * e;
        const size_t i = c0 + k;
        double ai = 0.0, bi = 0.0;
        if (i < len) {
            ai = vp[i];
            if (i < n) {   // (aug_update_kernel's expression)
                bi = aw[i];
                for (int q = 0; q < p; ++q) bi += uflip[i * p + q] * aug[q];
            } else {
                const int t = (int)(i - n);
                bi = t + 1 < p ? aug[t + 1] : 0.0;
            }
            vj[i] = bi;
        }
        sa[k] = ai;
        sb[k] = bi;
    }
    __syncthreads();
    small_rows(V, ldv, m, len, ept, sa, sb, red, j);
    if (tid < 2 * m) {
        const int which = tid / m, r = tid - which * m;
        partial[(size_t)blockIdx.x * 2 * m + tid] = red[64 * which + r];
    }
}

__global__ __launch_bounds__(kSmallThreads) void pmex_finish_project_small_kernel(const double* __restrict__ partial, int blocks, int j,
                                                                                  double* LT, double* Linv, int ld, double tol,
                                                                                  double* __restrict__ sol, double* __restrict__ hcol,
                                                                                  double* __restrict__ scal, double* __restrict__ G) {
    __shared__ double grp[8][128];
    const int m2 = 2 * (j + 1);   // <= 128
    const int k = threadIdx.x & 127, g = threadIdx.x >> 7;   // product, group of workgroups (8 groups of <= 16)
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int b = g + 8 * u;
        v[u] = (k < m2 && b < blocks) ? partial[(size_t)b * m2 + k] : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
    grp[g][k] = acc;
    __syncthreads();
    if ((int)threadIdx.x < m2) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += grp[q][threadIdx.x];
        G[threadIdx.x] = t;
    }
    __threadfence_block();
    __syncthreads();
    pmex_project(G, j, LT, Linv, ld, tol, sol, hcol, scal);
}

__global__ __launch_bounds__(kSmallThreads) void multi_dot2_small_kernel(const double* __restrict__ V, size_t ldv, int J,
                                                                         const double* __restrict__ a, const double* __restrict__ b,
                                                                         size_t n, int ept, double* __restrict__ partial) {
    __shared__ double sa[kSmallThreads * kSmallEpt], sb[kSmallThreads * kSmallEpt], red[128];
    small_products(V, ldv, J, a, b, n, ept, sa, sb, red);
    if ((int)threadIdx.x < 2 * J) {
        const int k = (int)threadIdx.x < J ? (int)threadIdx.x : 64 + ((int)threadIdx.x - J);
        partial[(size_t)blockIdx.x * 2 * J + threadIdx.x] = red[k];
    }
}

__global__ __launch_bounds__(256) void pair_update_small_kernel(double* __restrict__ a, double* __restrict__ b,
                                                                const double* __restrict__ V, size_t ldv, int m,
                                                                const double* __restrict__ ha, const double* __restrict__ hb, size_t n,
                                                                const double* __restrict__ dsc) {
    __shared__ double ca[64], cb[64];
    if ((int)threadIdx.x < m) { ca[threadIdx.x] = ha[threadIdx.x]; cb[threadIdx.x] = hb[threadIdx.x]; }
    __syncthreads();
    const double sa = dsc[0], cross = dsc[1], sb = dsc[2];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double x = a[i], y = b[i];
        for (int r = 0; r < m; ++r) {
            const double v = V[(size_t)r * ldv + i];
            x -= ca[r] * v;
            y -= cb[r] * v;
        }
        x *= sa;
        a[i] = x;
        b[i] = (y - cross * x) * sb;
    }
}

// ---- fgmres' lagged one-synchronisation Gram-Schmidt step on the device (solvers/fgmres.py:16-73; the host form:
// solvers.py _LowSyncGramSchmidt.step, statement for statement).  A Krylov vector then costs no host round trip: the host
// enqueues several vectors (operator, products, this kernel, the update of the two rows) and reads the Hessenberg columns
// once per pass - the iteration at which the reference would have stopped is found from those columns, vectors built past it
// are discarded.  One wave; lane i = row i (rows <= 64).
//   G: the 2 J products <V[k], a>, <V[k], b> (a = row J-2, b = row J-1);  R, T, K: (ld x ld) row-major, device
//   coef: ha at 0, hb at ld, (1/norm, cross, 1/norm) at 2 ld;  vn[J-2] = the norm of row J-2 (the operator's next scale)
//   flag: 0, or the first step whose norm estimate fell below the host form's thresholds (_SUSPECT, _BREAKDOWN) - that step and
//   every later one of the pass leave the rows untouched (zero coefficients, unit scalings) and the host redoes them.
constexpr double kGsSuspect = 1e-12, kGsBreakdown = 1e-14;
constexpr int kGsFusedBlocks = 64;            // workgroups of the products where the step kernel sums their partials itself
constexpr int kGsFusedMaxLen = 2 * 1024 * 1024;   // ... for vectors up to this length (beyond: 2048 blocks and the finish launch)

// the step's small algebra, one wave (lane i = row i); ga, gb: the lane's two products
// Values one workgroup of the one-launch step leaves for the others (partial sums, update coefficients) travel as agent-scope
// relaxed atomics: stores written through, loads served from the coherent level - with whole fences (__threadfence: write back and
// invalidate the XCD's L2) around the barrier the launch took 85 us more, at any size.
// lane k's value in every lane, k uniform: two v_readlane (the general shuffle goes through the LDS crossbar, ~100 cycles on the
// dependent chain of the substitutions below)
__device__ __forceinline__ double lane_bcast(double v, int k) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k), __builtin_amdgcn_readlane(__double2loint(v), k));
}
__device__ __forceinline__ void gs_put(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double gs_get(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// sT[k * 64 + i] = T[k][i] (k, i < J-2) and sRt[c * 65 + i] = R[i][1 + c] (i < J-1, c < J-3): what the two dependent chains below
// read, staged by the whole workgroup (gs_stage) - from global memory each of their ~2 (J-2) steps was a round trip
// sRt[63 * 65 + i] = K[i][J-3] (the lagged products' column, i < J-2) and sRt[64 * 65 - 1] = the flag (as a double): no global load
// is left on the algebra's chain
constexpr int kGsStageT = 64 * 64, kGsStageR = 64 * 65;
__device__ __forceinline__ void gs_stage(const double* T, const double* R, const double* K, const int* flag, int J, int ld, double* sT,
                                         double* sRt) {
    const int m2 = J - 2, m3 = J - 3;
    if ((int)threadIdx.x < m2 && m3 >= 0) sRt[63 * 65 + threadIdx.x] = K[threadIdx.x * ld + (m3 > 0 ? m3 : 0)];
    if (threadIdx.x == 64) sRt[64 * 65 - 1] = (double)*flag;
    // (batches of eight loads in flight: one element per trip of the loop was a memory round trip per element)
    const int bs = blockDim.x;
    for (int base = threadIdx.x; base < m2 * m2; base += 8 * bs) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * bs, k = idx / m2, i = idx - k * m2;
            t[u] = idx < m2 * m2 ? T[k * ld + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * bs, k = idx / m2, i = idx - k * m2;
            if (idx < m2 * m2) sT[k * 64 + i] = t[u];
        }
    }
    if (m3 > 0)
        for (int base = threadIdx.x; base < (J - 1) * m3; base += 8 * bs) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * bs, i = idx / m3, c = idx - i * m3;
                t[u] = idx < (J - 1) * m3 ? R[i * ld + 1 + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * bs, i = idx / m3, c = idx - i * m3;
                if (idx < (J - 1) * m3) sRt[c * 65 + i] = t[u];
            }
        }
}

__device__ __forceinline__ void gs_step_algebra(int i, double ga, double gb, int J, double* R, double* T, double* K, int ld, double* coef,
                                                double* vn, int* flag, const double* sT, const double* sRt) {
    const int m2 = J - 2;
    const bool dead = sRt[64 * 65 - 1] != 0.0;
    const double s = i < m2 ? ga : 0.0;
    double ss = s * s, sR = s * gb;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off, 64);
        sR += __shfl_xor(sR, off, 64);
    }
    const double gaa = lane_bcast(ga, m2), gbb = lane_bcast(gb, m2);   // <a, a>, <a, b>
    const double d = gaa - ss;
    const bool bad = dead || !(d == d) || d <= kGsSuspect * gaa;        // (suspect, breakdown, NaN: the host form's branches)
    if (bad) {
        if (i == 0 && !dead) *flag = J;
        if (i < m2) { gs_put(coef + i, 0.0); gs_put(coef + ld + i, 0.0); }
        if (i == 0) { gs_put(coef + 2 * ld, 1.0); gs_put(coef + 2 * ld + 1, 0.0); gs_put(coef + 2 * ld + 2, 1.0); vn[m2] = 1.0; }
        return;
    }
    const double norm = sqrt(d);
    const double cross = (gbb - sR) / norm;
    const double colJ1 = i == m2 ? cross : gb;                          // column J-1, rows 0 .. J-2
    if (i < J - 1) R[i * ld + J - 1] = colJ1;
    if (i == 0) { R[m2 * ld + m2] = norm; vn[m2] = norm; gs_put(coef + 2 * ld, 1.0 / norm); gs_put(coef + 2 * ld + 1, cross); gs_put(coef + 2 * ld + 2, 1.0 / norm); }
    if (i < m2) { T[i * ld + m2] = s / norm; gs_put(coef + i, s); gs_put(coef + ld + i, gb); }
    if (m2 > 0) {
        // L r3 = s, L = I + strict lower part of T[:m2, :m2]^T: forward substitution, column by column
        double acc = s;
        for (int k = 0; k < m2; ++k) {
            const double rk = lane_bcast(acc, k);
            if (i > k && i < m2) acc -= sT[k * 64 + i] * rk;
        }
        // column J-2 finished (the lagged correction); kept in a register for the product below: rows < J-2, then the norm
        const double colJ2 = i < m2 ? sRt[63 * 65 + i] + acc : (i == m2 ? norm : 0.0);
        if (i < m2) R[i * ld + m2] = colJ2;
        // K[:J-1, J-2] = (R[:J-1, J-1] - R[:J-1, 1:J-1] @ r3) / norm   (columns 1 .. J-3 from memory, column J-2 from the register)
        double dot = 0.0;
        for (int c = 0; c < m2; ++c) {
            const double rc = lane_bcast(acc, c);
            if (i < J - 1) dot += (c == m2 - 1 ? colJ2 : sRt[c * 65 + i]) * rc;
        }
        if (i < J - 1) K[i * ld + m2] = (colJ1 - dot) / norm;
    } else if (i == 0) {
        K[0] = cross / norm;
    }
}

// the products' partial sums ([block][2 J]) summed by four waves over the blocks -> wave 0's lanes hold <V[i], a>, <V[i], b>
// (returns false on the other three waves)
__device__ __forceinline__ bool gs_sum_partials(const double* partial, int blocks, int J, double (*red)[128], double& ga, double& gb) {
    const int i = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double xa = 0.0, xb = 0.0;
    if (i < J && wave < 4)
        for (int blk0 = wave; blk0 < blocks; blk0 += 32) {   // (eight blocks' pairs in flight, summed in the order of the blocks)
            double ta[8], tb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int blk = blk0 + 4 * u;
                ta[u] = blk < blocks ? gs_get(partial + (size_t)blk * 2 * J + i) : 0.0;
                tb[u] = blk < blocks ? gs_get(partial + (size_t)blk * 2 * J + J + i) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                xa += ta[u];
                xb += tb[u];
            }
        }
    if (wave < 4) {   // (the one-launch step runs it with sixteen waves)
        red[wave][i] = xa;
        red[wave][64 + i] = xb;
    }
    __syncthreads();
    if (wave != 0) return false;
    ga = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    gb = (red[0][64 + i] + red[1][64 + i]) + (red[2][64 + i] + red[3][64 + i]);
    return true;
}

__global__ __launch_bounds__(256) void fgmres_gs_step_kernel(const double* __restrict__ G, const double* __restrict__ partial,
                                                             int blocks, int J, double* R, double* T, double* K, int ld, double* coef,
                                                             double* vn, int* flag) {
    __shared__ double sT[kGsStageT], sRt[kGsStageR];
    __shared__ double red[4][128];
    const int i = threadIdx.x & 63, wave = threadIdx.x >> 6;
    gs_stage(T, R, K, flag, J, ld, sT, sRt);
    double ga = 0.0, gb = 0.0;
    if (partial != nullptr) {   // (multi_dot2_kernel's partial sums summed here: no finish launch)
        if (!gs_sum_partials(partial, blocks, J, red, ga, gb)) return;
    } else {
        __syncthreads();
        if (wave != 0) return;
        ga = i < J ? G[i] : 0.0;
        gb = i < J ? G[J + i] : 0.0;
    }
    gs_step_algebra(i, ga, gb, J, R, T, K, ld, coef, vn, flag, sT, sRt);
}

// ---- a whole Gram-Schmidt step of fgmres' device pass in ONE launch at launch-bound lengths: the products of all rows with the
// two newest (multi_dot2_small), the step's algebra (by the workgroup that arrives last), the update of the two rows
// (pair_update_small) - three dependent launches of ~10 us each otherwise.  The <= kSmallBlocks workgroups are all resident
// (256 CUs); they meet at a barrier through 1 + kSmallBlocks words of the workspace, each set to (kGsMagic << 32) | sequence
// number: sync[1 + g] by workgroup g when its products are written, sync[0] by workgroup 0 when the coefficients are
// (the sequence number is the library's, one per launch: nothing has to be zeroed, an aborted launch leaves nothing behind).  A
// workgroup's components of a and b stay in its registers from the products to the update.  The wait is bounded: a workgroup that
// does not see the release within kGsSpinLimit polls sets flag = -1 and leaves its rows untouched (the host raises).
constexpr unsigned long long kGsMagic = 0x57584753ull;
constexpr unsigned kGsSpinLimit = 1u << 22;

__global__ __launch_bounds__(kSmallThreads) void fgmres_vector_small_kernel(double* __restrict__ V, size_t ldv, int J, size_t n, int ept,
                                                                  double* partial, double* R, double* T, double* K, int ld,
                                                                  double* coef, double* vn, int* flag, unsigned long long* sync,
                                                                  unsigned seq) {
    __shared__ double sT[kGsStageT], sRt[kGsStageR];
    __shared__ double sa[kSmallThreads * kSmallEpt], sb[kSmallThreads * kSmallEpt];
    __shared__ double red[4][128];
    __shared__ double ca[64], cb[64], sc[3];
    __shared__ int s_lost;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* a = V + (size_t)(J - 2) * ldv;
    double* b = V + (size_t)(J - 1) * ldv;
    // ---- products (multi_dot2_small_kernel's arithmetic); this workgroup's components of a and b stay in LDS for the update
    small_products(V, ldv, J, a, b, n, ept, sa, sb, red[0]);
    if (tid < 2 * J) {
        const int k = tid < J ? tid : 64 + (tid - J);
        gs_put(partial + (size_t)blockIdx.x * 2 * J + tid, red[0][k]);
    }
    __syncthreads();   // (waits for this workgroup's stores as well)
    // ---- arrive: a word per workgroup; workgroup 0 waits for all of them, does the step's algebra and releases the others
    // (no read-modify-write: sixty-four workgroups retrying a compare-and-swap on one word took 80 us)
    const unsigned long long released = (kGsMagic << 32) | (unsigned long long)seq;
    if (tid == 0) {
        s_lost = 0;
        __hip_atomic_store(&sync[1 + blockIdx.x], released, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0) {
        gs_stage(T, R, K, flag, J, ld, sT, sRt);
        if (wave == 0) {
            unsigned polls = 0;
            for (;;) {
                const bool here = (lane >= (int)gridDim.x ||
                                   __hip_atomic_load(&sync[1 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == released) &&
                                  (lane + 64 >= (int)gridDim.x ||
                                   __hip_atomic_load(&sync[1 + 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == released);
                if (__all(here)) break;
                if (++polls > kGsSpinLimit) {
                    if (lane == 0) s_lost = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        if (!s_lost) {   // (uniform over the workgroup)
            double ga = 0.0, gb = 0.0;
            if (gs_sum_partials(partial, (int)gridDim.x, J, red, ga, gb)) gs_step_algebra(lane, ga, gb, J, R, T, K, ld, coef, vn, flag, sT, sRt);
            __syncthreads();   // (wave 0's stores of the coefficients are complete)
            if (tid == 0) __hip_atomic_store(&sync[0], released, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (tid == 0) {
        unsigned polls = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != released) {
            if (++polls > kGsSpinLimit) { s_lost = 1; break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    if (s_lost) {   // the release never came: say so, leave the rows as they are
        if (tid == 0) atomicExch(flag, -1);
        return;
    }
    // ---- the update of this workgroup's components (pair_update_small_kernel's arithmetic) with the coefficients just left
    const int m = J - 2;
    if (tid < m) {
        ca[tid] = gs_get(coef + tid);
        cb[tid] = gs_get(coef + ld + tid);
    }
    if (tid < 3) sc[tid] = gs_get(coef + 2 * ld + tid);
    __syncthreads();
    const double scale_a = sc[0], cross = sc[1], scale_b = sc[2];
    const size_t c0 = (size_t)blockIdx.x * kSmallThreads * ept;
    double x[kSmallEpt], y[kSmallEpt];
#pragma unroll
    for (int e = 0; e < kSmallEpt; ++e) {
        x[e] = e < ept ? sa[tid + kSmallThreads * e] : 0.0;
        y[e] = e < ept ? sb[tid + kSmallThreads * e] : 0.0;
    }
#pragma unroll 8
    for (int r = 0; r < m; ++r) {
        const double* row = V + (size_t)r * ldv + c0;
        const double car = ca[r], cbr = cb[r];
#pragma unroll
        for (int e = 0; e < kSmallEpt; ++e) {
            const int k = tid + kSmallThreads * e;
            const double v = (e < ept && c0 + k < n) ? row[k] : 0.0;
            x[e] -= car * v;
            y[e] -= cbr * v;
        }
    }
#pragma unroll
    for (int e = 0; e < kSmallEpt; ++e) {
        const size_t i = c0 + tid + kSmallThreads * e;
        if (e < ept && i < n) {
            const double xs = x[e] * scale_a;
            a[i] = xs;
            b[i] = (y[e] - cross * xs) * scale_b;
        }
    }
}

extern "C" {

size_t wx_multi_dot_workspace(int m) { return (size_t)kDotBlocks * (m > 0 ? m : 1); }   // (wx_multi_dot2: pass 2 m)

wx_status wx_multi_dot(const double* V, size_t ldv, int m, const double* w, size_t n, double* out, double* workspace,
                       wx_stream stream) {
    if (m <= 0) return WX_OK;
    if (!V || !w || !out || !workspace) return fail(WX_ERR_INVALID, "wx_multi_dot: null argument");
    if (ldv < n) return fail(WX_ERR_INVALID, "wx_multi_dot: row stride %zu shorter than the vectors (%zu)", ldv, n);
    WX_STREAM(st, stream);
    for (int r = 0; r < m; r += kRowsPerPass)
        dispatch_dot<kRowsPerPass>(m - r < kRowsPerPass ? m - r : kRowsPerPass, V, ldv, r, w, n, workspace, m, st);
    hipLaunchKernelGGL(multi_dot_finish_kernel, dim3(m), dim3(64), 0, st, workspace, dot_blocks(n), m, out);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

size_t wx_kiops_finish_workspace(size_t len) { return ((len + kFinishChunk - 1) / kFinishChunk) * (kFinishMaxIop + 1); }

wx_status wx_kiops_finish(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* aw, const double* uflip,
                          double* hcol, double* workspace, wx_stream stream) {
    if (!V || !aw || !uflip || !hcol || !workspace) return fail(WX_ERR_INVALID, "wx_kiops_finish: null argument");
    if (j < 1 || p < 1 || p > kFinishMaxP || iop < 1 || iop > kFinishMaxIop || ldv < n + (size_t)p ||
        n + (size_t)p > (size_t)kFinishChunk * kFinishThreads)
        return fail(WX_ERR_INVALID, "wx_kiops_finish: bad shape (j=%d p=%d iop=%d ldv=%zu n=%zu)", j, p, iop, ldv, n);
    WX_STREAM(st, stream);
    const unsigned G = (unsigned)((n + p + kFinishChunk - 1) / kFinishChunk);
    hipLaunchKernelGGL(kiops_finish_a, dim3(G), dim3(kFinishThreads), 0, st, V, ldv, j, n, p, iop, aw, uflip, workspace);
    hipLaunchKernelGGL(kiops_finish_b, dim3(G), dim3(kFinishThreads), 0, st, V, ldv, j, n, p, iop, workspace, hcol);
    hipLaunchKernelGGL(kiops_finish_c, dim3(G), dim3(kFinishThreads), 0, st, V, ldv, j, n, p, workspace, hcol);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

size_t wx_kiops_long_workspace(void) { return (size_t)kLongBlocks * kFinishMaxIop; }

static wx_status kiops_long_check(const void* V, int j, int p, int iop, size_t ldv, size_t n, const char* who) {
    if (!V) return fail(WX_ERR_INVALID, "%s: null argument", who);
    if (j < 1 || p < 1 || p > kFinishMaxP || iop < 1 || iop > kFinishMaxIop || ldv < n + (size_t)p)
        return fail(WX_ERR_INVALID, "%s: bad shape (j=%d p=%d iop=%d ldv=%zu n=%zu)", who, j, p, iop, ldv, n);
    return WX_OK;
}

wx_status wx_kiops_long_a(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* aw, const double* uflip,
                          double* dots, double* workspace, wx_stream stream) {
    return wx_kiops_long_a_scaled(V, ldv, j, n, p, iop, aw, uflip, dots, workspace, nullptr, stream);
}

wx_status wx_kiops_long_a_scaled(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* aw, const double* uflip,
                                 double* dots, double* workspace, const double* scales, wx_stream stream) {
    wx_status s = kiops_long_check(V, j, p, iop, ldv, n, "wx_kiops_long_a");
    if (s != WX_OK) return s;
    if (!aw || !uflip || !dots || !workspace) return fail(WX_ERR_INVALID, "wx_kiops_long_a: null argument");
    WX_STREAM(st, stream);
    hipLaunchKernelGGL(kiops_long_a, dim3(kLongBlocks), dim3(kFinishThreads), 0, st, V, ldv, j, n, p, iop, aw, uflip, workspace, scales);
    const int nr = j - (j - iop > 0 ? j - iop : 0);
    hipLaunchKernelGGL(multi_dot_finish_kernel, dim3(nr), dim3(64), 0, st, workspace, kLongBlocks, kFinishMaxIop, dots);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_kiops_long_a_formed(double* V, size_t ldv, int j, size_t n, int p, int iop, double* dots, double* workspace,
                                 const double* scales, wx_stream stream) {
    wx_status s = kiops_long_check(V, j, p, iop, ldv, n, "wx_kiops_long_a_formed");
    if (s != WX_OK) return s;
    if (!dots || !workspace) return fail(WX_ERR_INVALID, "wx_kiops_long_a_formed: null argument");
    WX_STREAM(st, stream);
    hipLaunchKernelGGL(kiops_long_a, dim3(kLongBlocks), dim3(kFinishThreads), 0, st, V, ldv, j, n, p, iop,
                       static_cast<const double*>(nullptr), static_cast<const double*>(nullptr), workspace, scales);
    const int nr = j - (j - iop > 0 ? j - iop : 0);
    hipLaunchKernelGGL(multi_dot_finish_kernel, dim3(nr), dim3(64), 0, st, workspace, kLongBlocks, kFinishMaxIop, dots);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_kiops_long_a_finish(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* partials, size_t nblocks,
                                 double* dots, double* workspace, const double* scales, wx_stream stream) {
    wx_status s = kiops_long_check(V, j, p, iop, ldv, n, "wx_kiops_long_a_finish");
    if (s != WX_OK) return s;
    if (!partials || !dots || !workspace) return fail(WX_ERR_INVALID, "wx_kiops_long_a_finish: null argument");
    const int ilow = j - iop > 0 ? j - iop : 0, nr = j - ilow;
    if (nr > 2) return fail(WX_ERR_INVALID, "wx_kiops_long_a_finish: %d products (the matvec's store leaves two at most)", nr);
    static_assert(2 * kFinishStage1 <= kLongBlocks * kFinishMaxIop, "the long-vector workspace holds the first step's results");
    WX_STREAM(st, stream);
    hipLaunchKernelGGL(kiops_long_a_finish1, dim3(kFinishStage1, nr), dim3(256), 0, st, partials, nblocks, workspace);
    hipLaunchKernelGGL(kiops_long_a_finish_kernel, dim3(nr), dim3(256), 0, st, V, ldv, j, n, p, ilow, workspace, dots, scales);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_kiops_long_b_fold_finish(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* h,
                                      const double* partials, size_t nblocks, double* nrm2, double* workspace, wx_stream stream) {
    wx_status s = kiops_long_check(V, j, p, iop, ldv, n, "wx_kiops_long_b_fold_finish");
    if (s != WX_OK) return s;
    if (!h || !partials || !nrm2 || !workspace || nblocks == 0) return fail(WX_ERR_INVALID, "wx_kiops_long_b_fold_finish: null argument");
    const int ilow = j - iop > 0 ? j - iop : 0, nr = j - ilow;
    WX_STREAM(st, stream);
    hipLaunchKernelGGL(kiops_fold_sum1, dim3(kFinishStage1), dim3(256), 0, st, partials, nblocks, workspace);
    hipLaunchKernelGGL(kiops_fold_finish_kernel, dim3(1), dim3(256), 0, st, V, ldv, j, n, p, ilow, nr, h, workspace, nrm2);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_kiops_long_b(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* h, double* nrm2,
                          double* workspace, wx_stream stream) {
    return wx_kiops_long_b_scaled(V, ldv, j, n, p, iop, h, nrm2, workspace, nullptr, stream);
}

wx_status wx_kiops_long_b_scaled(double* V, size_t ldv, int j, size_t n, int p, int iop, const double* h, double* nrm2,
                                 double* workspace, const double* scales, wx_stream stream) {
    wx_status s = kiops_long_check(V, j, p, iop, ldv, n, "wx_kiops_long_b");
    if (s != WX_OK) return s;
    if (!h || !nrm2 || !workspace) return fail(WX_ERR_INVALID, "wx_kiops_long_b: null argument");
    WX_STREAM(st, stream);
    hipLaunchKernelGGL(kiops_long_b, dim3(kLongBlocks), dim3(kFinishThreads), 0, st, V, ldv, j, n, p, iop, h, workspace, scales);
    hipLaunchKernelGGL(multi_dot_finish_kernel, dim3(1), dim3(64), 0, st, workspace, kLongBlocks, 1, nrm2);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_kiops_long_c(double* V, size_t ldv, int j, size_t n, int p, const double* nrm2, double* hcol, wx_stream stream) {
    if (!V || !nrm2 || !hcol || j < 1 || p < 1 || ldv < n + (size_t)p) return fail(WX_ERR_INVALID, "wx_kiops_long_c: bad argument");
    WX_STREAM(st, stream);
    hipLaunchKernelGGL(kiops_long_c, dim3(kLongBlocks), dim3(kFinishThreads), 0, st, V, ldv, j, n + (size_t)p, nrm2, hcol);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_kiops_long_c_lazy(double* V, size_t ldv, int j, size_t n, int p, const double* nrm2, double* hcol, double* scales,
                               wx_stream stream) {
    if (!V || !nrm2 || !hcol || !scales || j < 1 || p < 1 || p > kFinishMaxP || ldv < n + (size_t)p)
        return fail(WX_ERR_INVALID, "wx_kiops_long_c_lazy: bad argument");
    WX_STREAM(st, stream);
    hipLaunchKernelGGL(kiops_long_c_lazy, dim3(1), dim3(64), 0, st, V, ldv, j, n, p, nrm2, hcol, scales);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_multi_dot2(const double* V, size_t ldv, int m, const double* a, const double* b, size_t n, double* out,
                        double* workspace, wx_stream stream) {
    if (m <= 0) return WX_OK;
    if (!V || !a || !b || !out || !workspace) return fail(WX_ERR_INVALID, "wx_multi_dot2: null argument");
    if (ldv < n) return fail(WX_ERR_INVALID, "wx_multi_dot2: row stride %zu shorter than the vectors (%zu)", ldv, n);
    WX_STREAM(st, stream);
    for (int r = 0; r < m; r += kRowsPerPass2)
        dispatch_dot2<kRowsPerPass2>(m - r < kRowsPerPass2 ? m - r : kRowsPerPass2, V, ldv, r, a, b, n, workspace, m, st);
    hipLaunchKernelGGL(multi_dot_finish_kernel, dim3(2 * m), dim3(64), 0, st, workspace, dot_blocks(n), 2 * m, out);
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
    dispatch_pair<kRowsPerPass2>(m - r, a, b, V, ldv, r, ha, hb, n, 1, scale_a, cross, scale_b, st);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

size_t wx_fgmres_workspace(int rows) {   // products' partial sums + the 2 rows products themselves
    const int r = rows > 0 ? rows : 1;
    return wx_multi_dot_workspace(2 * r) + 2 * (size_t)r + 2 + kSmallBlocks;   // (+ the barrier words of the one-launch step)
}

// One Krylov vector of fgmres' low-synchronisation Gram-Schmidt (step J: rows a = V[J-2], b = V[J-1]) without a host round trip:
// the 2 J products, their reduction over the ranks (comm nullable), the step's small algebra (fgmres_gs_step_kernel), the update
// of the two rows with the coefficients that kernel left on the device.
wx_status wx_fgmres_vector(double* V, size_t ldv, int J, size_t n, double* R, double* T, double* K, int ld, double* coef,
                           double* vn, int* flag, double* workspace, wx_comm* comm, wx_stream stream) {
    if (!V || !R || !T || !K || !coef || !vn || !flag || !workspace) return fail(WX_ERR_INVALID, "wx_fgmres_vector: null argument");
    if (J < 2 || J > 64 || J > ld || ldv < n) return fail(WX_ERR_INVALID, "wx_fgmres_vector: J = %d (2..64, <= ld = %d), row stride %zu, n = %zu", J, ld, ldv, n);
    WX_STREAM(st, stream);
    double* a = V + (size_t)(J - 2) * ldv;
    double* b = V + (size_t)(J - 1) * ldv;
    double* G = workspace + wx_multi_dot_workspace(2 * ld);
    if (comm) {
        wx_status s = wx_multi_dot2(V, ldv, J, a, b, n, G, workspace, stream);
        if (s != WX_OK) return s;
        s = wx_comm_allreduce(comm, G, (size_t)(2 * J), WX_REDUCE_SUM, stream);
        if (s != WX_OK) return s;
        hipLaunchKernelGGL(fgmres_gs_step_kernel, dim3(1), dim3(256), 0, st, G, (const double*)nullptr, 0, J, R, T, K, ld, coef, vn, flag);
    } else if (n <= (size_t)kSmallMaxLen && J <= 64) {
        // one rank, launch-bound lengths: the products of ALL rows in one launch, their partial sums summed by the step kernel, the
        // update of both rows over ALL rows in one launch: three launches per Krylov vector beside the operator's - or, on request,
        // ONE, the workgroups meeting at a barrier inside it (not under stream capture: a replay would meet its own old sequence number)
        const int ept = small_ept(n);
        const int blocks = small_blocks(n, ept);
        // (opt-in: measured equal to the three launches - the GPU runs dependent launches 0.1 us apart - and a barrier inside a launch
        // needs every workgroup resident, which a GPU shared with other processes does not promise; profiles/r06_fgmres_device_passes.txt)
        const char* sw = getenv("WXHIP_FGMRES_ONE_LAUNCH");   // (read per call: the A/B of tests/test_callers_gpu.py)
        const bool one_launch = sw && sw[0] == '1';
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (one_launch && n > 0 && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
            static std::atomic<unsigned> sequence{0};
            const unsigned seq = sequence.fetch_add(1u) + 1u;
            unsigned long long* sync = reinterpret_cast<unsigned long long*>(G + 2 * (size_t)ld);
            hipLaunchKernelGGL(fgmres_vector_small_kernel, dim3(blocks), dim3(kSmallThreads), 0, st, V, ldv, J, n, ept, workspace, R, T, K, ld, coef,
                               vn, flag, sync, seq);
            WX_HIP_TRY(hipGetLastError());
            return WX_OK;
        }
        hipLaunchKernelGGL(multi_dot2_small_kernel, dim3(blocks), dim3(kSmallThreads), 0, st, V, ldv, J, a, b, n, ept, workspace);
        hipLaunchKernelGGL(fgmres_gs_step_kernel, dim3(1), dim3(256), 0, st, (const double*)nullptr, workspace, blocks, J, R, T, K, ld,
                           coef, vn, flag);
        if (n > 0) {
            const size_t wantp = (n + 255) / 256;
            hipLaunchKernelGGL(pair_update_small_kernel, dim3((unsigned)(wantp ? wantp : 1)), dim3(256), 0, st, a, b, V, ldv, J - 2, coef,
                               coef + ld, n, coef + 2 * ld);
        }
        WX_HIP_TRY(hipGetLastError());
        return WX_OK;
    } else if (n <= (size_t)kGsFusedMaxLen) {
        // one rank, launch-bound lengths: the products over at most kGsFusedBlocks workgroups, their partial sums summed by the
        // step kernel itself (four waves over <= 64 blocks: a dozen loads per lane) - one launch less per Krylov vector
        const int want = dot_blocks(n), blocks = want < kGsFusedBlocks ? want : kGsFusedBlocks;
        for (int r = 0; r < J; r += kRowsPerPass2)
            dispatch_dot2<kRowsPerPass2>(J - r < kRowsPerPass2 ? J - r : kRowsPerPass2, V, ldv, r, a, b, n, workspace, J, st, blocks);
        hipLaunchKernelGGL(fgmres_gs_step_kernel, dim3(1), dim3(256), 0, st, (const double*)nullptr, workspace, blocks, J, R, T, K, ld,
                           coef, vn, flag);
    } else {
        const wx_status s = wx_multi_dot2(V, ldv, J, a, b, n, G, workspace, stream);
        if (s != WX_OK) return s;
        hipLaunchKernelGGL(fgmres_gs_step_kernel, dim3(1), dim3(256), 0, st, G, (const double*)nullptr, 0, J, R, T, K, ld, coef, vn, flag);
    }
    const int m = J - 2;
    if (n > 0) {
        int r = 0;
        for (; r + kRowsPerPass2 < m; r += kRowsPerPass2)   // (strictly less: the last batch carries the scalings)
            launch_pair<kRowsPerPass2>(a, b, V, ldv, r, coef, coef + ld, n, 0, 1.0, 0.0, 1.0, st);
        dispatch_pair<kRowsPerPass2>(m - r, a, b, V, ldv, r, coef, coef + ld, n, 1, 1.0, 0.0, 1.0, st, coef + 2 * ld);
    }
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_multi_axpy_scaled(double* w, const double* V, size_t ldv, int m, const double* h, size_t n, double scale,
                               wx_stream stream) {
    if (m <= 0 || n == 0) return WX_OK;
    if (!V || !w || !h) return fail(WX_ERR_INVALID, "wx_multi_axpy: null argument");
    if (ldv < n) return fail(WX_ERR_INVALID, "wx_multi_axpy: row stride %zu shorter than the vectors (%zu)", ldv, n);
    WX_STREAM(st, stream);
    for (int r = 0; r < m; r += kRowsPerPass)
        dispatch_axpy<kRowsPerPass>(m - r < kRowsPerPass ? m - r : kRowsPerPass, w, V, ldv, r, h, n,
                                    r + kRowsPerPass >= m ? scale : 1.0, st);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_multi_axpy(double* w, const double* V, size_t ldv, int m, const double* h, size_t n, wx_stream stream) {
    return wx_multi_axpy_scaled(w, V, ldv, m, h, n, 1.0, stream);
}

wx_status wx_krylov_aug_update(double* V, size_t ldv, int j, size_t n, int p, const double* aw, const double* uflip,
                               wx_stream stream) {
    if (!V || (n > 0 && (!aw || !uflip))) return fail(WX_ERR_INVALID, "wx_krylov_aug_update: null argument");   // (n == 0: an idle rank's empty operands)
    if (j < 1 || p < 1 || p > 16 || ldv < n + (size_t)p)
        return fail(WX_ERR_INVALID, "wx_krylov_aug_update: j = %d, p = %d (1..16), row stride %zu, n = %zu", j, p, ldv, n);
    WX_STREAM(st, stream);
    const size_t want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192 ? (want ? want : 1) : 8192);
    hipLaunchKernelGGL(aug_update_kernel, dim3(grid), dim3(256), 0, st, V, ldv, j, n, p, aw, uflip);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}


// doubles of scratch for wx_pmex_vector with a basis of at most mmax vectors
size_t wx_pmex_workspace(int mmax) {
    const int m = mmax > 0 ? mmax : 1;
    return (size_t)kDotBlocks * 2 * (m + 1) + 2 * (size_t)(m + 1) + (size_t)m + 8 + kPmexAxpyBlocks;
}

// split: the vectors are cut over ranks (this rank holds n of the components, the p augmented ones are replicated): the
// products and the own norm are taken over the n-long parts, all-reduced on `comm` (nullable: one rank taking the several-rank
// code path) on the caller's stream - graph nodes under capture -, the augmented components added once afterwards
// (solvers/pmex.py:150-173, 194-218)
static wx_status pmex_vector_impl(double* V, size_t ldv, int j, size_t n, int p, const double* aw, const double* uflip, double* LT,
                                  double* Linv, int ld, double tol, double* hcol, double* own, double* workspace, int mmax,
                                  bool split, wx_comm* comm, wx_stream stream, const char* who) {
    // (n == 0: a rank that owns no tile - ranks 6, 7 of an 8-GPU node - holds the p augmented components only; its n-long
    // operands are empty tensors, whose address is null.  It still takes part in both reductions and runs every kernel of the
    // augmented part: none of them touches aw / uflip when n == 0.)
    if (!V || (n > 0 && (!aw || !uflip)) || !LT || !Linv || !hcol || !own || !workspace)
        return fail(WX_ERR_INVALID, "%s: null argument", who);
    if (j < 1 || j > mmax || mmax > kPmexMaxM || ld < mmax || p < 1 || p > 16 || ldv < n + (size_t)p)
        return fail(WX_ERR_INVALID, "%s: j = %d, mmax = %d (<= %d), ld = %d, p = %d (1..16), row stride %zu, n = %zu",
                    who, j, mmax, kPmexMaxM, ld, p, ldv, n);
    WX_STREAM(st, stream);
    double* dotw = workspace;
    double* G = dotw + (size_t)kDotBlocks * 2 * (mmax + 1);
    double* sol = G + 2 * (size_t)(mmax + 1);
    double* scal = sol + mmax;
    double* part = scal + 8;
    const size_t len = n + (size_t)p;
    double* vj = V + (size_t)j * ldv;
    const char* fused_sw = getenv("WXHIP_PMEX_FUSED");   // (A/B and tests: "0" = the row-batched launches at every size)
    if (!split && !(fused_sw && fused_sw[0] == '0') && len <= kPmexFusedMaxLen) {   // launch-bound sizes: four launches (see pmex_aug_dot2_kernel)
        const size_t wantf = (len + 255) / 256;
        const unsigned gridf = (unsigned)(wantf < kPmexAxpyBlocks ? (wantf ? wantf : 1) : kPmexAxpyBlocks);
        const char* small_sw = getenv("WXHIP_PMEX_SMALL");   // (A/B and tests: "0" = the round-5 product and projector launches)
        if (!(small_sw && small_sw[0] == '0') && len <= (size_t)kSmallMaxLen && j + 1 <= 64) {
            const int ept = small_ept(len), blocks = small_blocks(len, ept);
            hipLaunchKernelGGL(pmex_aug_products_small_kernel, dim3(blocks), dim3(kSmallThreads), 0, st, V, ldv, j, n, p, aw, uflip, ept,
                               dotw);
            hipLaunchKernelGGL(pmex_finish_project_small_kernel, dim3(1), dim3(kSmallThreads), 0, st, dotw, blocks, j, LT, Linv, ld, tol,
                               sol, hcol, scal, G);
            hipLaunchKernelGGL(pmex_axpy_all_kernel, dim3(gridf), dim3(256), 0, st, vj, V, ldv, j, sol, len, scal, part, len);
            hipLaunchKernelGGL(pmex_finish_scale_kernel, dim3(gridf), dim3(256), 0, st, vj, len, part, (int)gridf, tol, scal, hcol + j,
                               own);
            WX_HIP_TRY(hipGetLastError());
            return WX_OK;
        }
        // one workgroup of partial products per 256 components and no more (the row-batched kernels always leave kDotBlocks =
        // 2048 partials per product, which ONE finishing workgroup cannot sum in a launch's time): at most 512 here
        const unsigned gridd = (unsigned)(wantf < kPmexFusedDotBlocks ? (wantf ? wantf : 1) : kPmexFusedDotBlocks);
        hipLaunchKernelGGL((pmex_aug_dot2_kernel<kRowsPerPass2>), dim3(gridd), dim3(kDotThreads), 0, st, V, ldv, j, n, p, aw,
                           uflip, dotw);
        hipLaunchKernelGGL(pmex_finish_project_kernel, dim3(1), dim3(256), 0, st, dotw, (int)gridd, j, LT, Linv, ld, tol, sol, hcol,
                           scal, G);
        hipLaunchKernelGGL(pmex_axpy_all_kernel, dim3(gridf), dim3(256), 0, st, vj, V, ldv, j, sol, len, scal, part, len);
        hipLaunchKernelGGL(pmex_finish_scale_kernel, dim3(gridf), dim3(256), 0, st, vj, len, part, (int)gridf, tol, scal, hcol + j,
                           own);
        WX_HIP_TRY(hipGetLastError());
        return WX_OK;
    }
    {
        const size_t want = (n + 255) / 256;
        const unsigned grid = (unsigned)(want < 8192 ? (want ? want : 1) : 8192);
        hipLaunchKernelGGL(aug_update_kernel, dim3(grid), dim3(256), 0, st, V, ldv, j, n, p, aw, uflip);
    }
    const int m = j + 1;
    const size_t dlen = split ? n : len;   // what the products run over
    for (int r = 0; r < m; r += kRowsPerPass2)
        dispatch_dot2<kRowsPerPass2>(m - r < kRowsPerPass2 ? m - r : kRowsPerPass2, V, ldv, r, V + (size_t)(j - 1) * ldv, vj, dlen,
                                     dotw, m, st);
    hipLaunchKernelGGL(multi_dot_finish_kernel, dim3(2 * m), dim3(64), 0, st, dotw, dot_blocks(dlen), 2 * m, G);
    if (split) {
        WX_HIP_TRY(hipGetLastError());
        if (comm) {
            const wx_status cs = wx_comm_allreduce(comm, G, (size_t)(2 * m), WX_REDUCE_SUM, stream);
            if (cs != WX_OK) return cs;
        }
        hipLaunchKernelGGL(pmex_split_aug_kernel, dim3(1), dim3(256), 0, st, V, ldv, j, n, p, G);
    }
    hipLaunchKernelGGL(pmex_project_kernel, dim3(1), dim3(256), 0, st, G, j, LT, Linv, ld, tol, sol, hcol, scal);
    const size_t want = (len + 255) / 256;
    const unsigned grid = (unsigned)(want < kPmexAxpyBlocks ? (want ? want : 1) : kPmexAxpyBlocks);
    for (int r = 0; r < j; r += kRowsPerPass) {
        const bool last = r + kRowsPerPass >= j;
        dispatch_axpy_dev<kRowsPerPass>(j - r < kRowsPerPass ? j - r : kRowsPerPass, vj, V, ldv, r, sol, len,
                                        last ? scal : nullptr, last ? part : nullptr, grid, st, split ? n : len);
    }
    if (split) {
        double* total = scal + 4;
        hipLaunchKernelGGL(pmex_split_norm_kernel, dim3(1), dim3(256), 0, st, part, (int)grid, total);
        WX_HIP_TRY(hipGetLastError());
        if (comm) {
            const wx_status cs = wx_comm_allreduce(comm, total, 1, WX_REDUCE_SUM, stream);
            if (cs != WX_OK) return cs;
        }
        hipLaunchKernelGGL(pmex_split_norm_aug_kernel, dim3(1), dim3(64), 0, st, vj + n, p, total);
        hipLaunchKernelGGL(pmex_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)nullptr, 0, tol, scal, hcol + j, own, total);
    } else {
        hipLaunchKernelGGL(pmex_finish_kernel, dim3(1), dim3(256), 0, st, part, (int)grid, tol, scal, hcol + j, own,
                           (const double*)nullptr);
    }
    hipLaunchKernelGGL(scale_if_kernel, dim3(grid), dim3(256), 0, st, vj, len, scal + 3);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

wx_status wx_pmex_vector(double* V, size_t ldv, int j, size_t n, int p, const double* aw, const double* uflip, double* LT,
                         double* Linv, int ld, double tol, double* hcol, double* own, double* workspace, int mmax,
                         wx_stream stream) {
    return pmex_vector_impl(V, ldv, j, n, p, aw, uflip, LT, Linv, ld, tol, hcol, own, workspace, mmax, false, nullptr, stream,
                            "wx_pmex_vector");
}

wx_status wx_pmex_vector_split(double* V, size_t ldv, int j, size_t n, int p, const double* aw, const double* uflip, double* LT,
                               double* Linv, int ld, double tol, double* hcol, double* own, double* workspace, int mmax,
                               wx_comm* comm, wx_stream stream) {
    return pmex_vector_impl(V, ldv, j, n, p, aw, uflip, LT, Linv, ld, tol, hcol, own, workspace, mmax, true, comm, stream,
                            "wx_pmex_vector_split");
}

// wx_pmex_vector with the complex-step matvec in front (wx_euler3d_batch_extrap_pack + wx_euler3d_batch_jvp on the previous
// vector), from ONE host call: for a rank that owns the whole sphere at launch-bound sizes, where the host side of separate
// calls costs more than the kernels (the PMEX twin of wx_euler3d_batch_kiops_vector).
wx_status wx_euler3d_batch_pmex_vector(const wx_euler3d_batch* b, const double* q, double* V, size_t ldv, int j, size_t n,
                                       int p, double eps, double scale, const double* uflip, double* LT, double* Linv, int ld,
                                       double tol, double* hcol, double* own, double* aw, double* workspace, int mmax,
                                       size_t panel_stride, wx_stream stream) {
    if (!b || !q || !V || !aw) return fail(WX_ERR_INVALID, "wx_euler3d_batch_pmex_vector: null argument");
    if (j < 1) return fail(WX_ERR_INVALID, "wx_euler3d_batch_pmex_vector: j = %d", j);
    const double* v = V + (size_t)(j - 1) * ldv;
    wx_status s = wx_euler3d_batch_extrap_pack(b, q, v, eps, panel_stride, stream);
    if (s != WX_OK) return s;
    s = wx_euler3d_batch_jvp(b, q, v, eps, aw, scale, panel_stride, WX_REGION_ALL, stream);
    if (s != WX_OK) return s;
    return wx_pmex_vector(V, ldv, j, n, p, aw, uflip, LT, Linv, ld, tol, hcol, own, workspace, mmax, stream);
}

}  // extern "C"
