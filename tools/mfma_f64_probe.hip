// Development probe (not product code): v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 on gfx950.
//   1. checks the operand / result lane maps the derivative kernels rely on, with asymmetric integer data:
//        A[row = lane & 15][k = lane >> 4],  B[k = lane >> 4][col = lane & 15],
//        D[row = (lane >> 4) + 4 * reg][col = lane & 15]
//   2. measures issue interval (independent accumulators) and dependent-accumulator latency in shader cycles,
//      one wave per SIMD and two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/_bin/mfma_f64_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double* A, const double* B, double* D) {
    // A: 16 x 4 row-major, B: 4 x 16 row-major, D: 16 x 16 row-major
    const int l = threadIdx.x;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int CHAINS, bool DEP>
__global__ void rate_kernel(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    const double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    d4 acc[CHAINS];
    for (int i = 0; i < CHAINS; ++i) acc[i] = d4{0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    // make the last results architecturally needed before the second stamp
    double s = 0;
    for (int i = 0; i < CHAINS; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    asm volatile("s_nop 0" ::"v"(s));
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + l] = s;
    if ((l & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + l / 64] = t1 - t0;
}

template <int CHAINS>
__global__ void rate4_kernel(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    const double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    double acc[CHAINS];
    for (int i = 0; i < CHAINS; ++i) acc[i] = 0.0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < CHAINS; ++i) s += acc[i];
    asm volatile("s_nop 0" ::"v"(s));
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + l] = s;
    if ((l & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + l / 64] = t1 - t0;
}

// the same flops on the vector pipe, for comparison: CHAINS independent v_fma_f64 chains
template <int CHAINS>
__global__ void valu_kernel(double* out, long long* cyc, int iters) {
    const int l = threadIdx.x;
    const double a = 1.0 + l * 1e-9, b = 1e-9 * l;
    double acc[CHAINS];
    for (int i = 0; i < CHAINS; ++i) acc[i] = i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < CHAINS; ++i) s += acc[i];
    asm volatile("s_nop 0" ::"v"(s));
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + l] = s;
    if ((l & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + l / 64] = t1 - t0;
}

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

template <typename F>
static double run(F launch, long long* dcyc, int nwaves) {
    launch();
    CK(hipDeviceSynchronize());
    launch();
    CK(hipDeviceSynchronize());
    std::vector<long long> h(nwaves);
    CK(hipMemcpy(h.data(), dcyc, sizeof(long long) * nwaves, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / nwaves;
}

int main() {
    // ---- layout
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 11 + j * 5 + (j * j) % 7;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
    printf("layout check v_mfma_f64_16x16x4_f64: %s (%d of 256 differ)\n", bad ? "MISMATCH" : "ok", bad);

    // ---- rates: 256 CUs x (4 or 8) waves
    const int iters = 4096;
    double* dout;
    long long* dcyc;
    CK(hipMalloc(&dout, 8 * 1024 * 512));
    CK(hipMalloc(&dcyc, 8 * 1024 * 8));
    for (int wpb : {4, 8}) {
        const int blocks = 256, threads = 64 * wpb, nw = blocks * wpb;
        double c;
        c = run([&] { hipLaunchKernelGGL((rate_kernel<1, true>), dim3(blocks), dim3(threads), 0, 0, dout, dcyc, iters); }, dcyc, nw);
        printf("%d waves/CU  16x16x4 f64, 1 dependent chain : %.1f cycles per MFMA per wave\n", wpb, c / iters);
        c = run([&] { hipLaunchKernelGGL((rate_kernel<4, false>), dim3(blocks), dim3(threads), 0, 0, dout, dcyc, iters); }, dcyc, nw);
        printf("%d waves/CU  16x16x4 f64, 4 independent     : %.1f cycles per MFMA per wave\n", wpb, c / iters / 4);
        c = run([&] { hipLaunchKernelGGL((rate4_kernel<1>), dim3(blocks), dim3(threads), 0, 0, dout, dcyc, iters); }, dcyc, nw);
        printf("%d waves/CU  4x4x4_4b f64, 1 dependent chain: %.1f cycles per MFMA per wave\n", wpb, c / iters);
        c = run([&] { hipLaunchKernelGGL((rate4_kernel<8>), dim3(blocks), dim3(threads), 0, 0, dout, dcyc, iters); }, dcyc, nw);
        printf("%d waves/CU  4x4x4_4b f64, 8 independent    : %.1f cycles per MFMA per wave\n", wpb, c / iters / 8);
        c = run([&] { hipLaunchKernelGGL((valu_kernel<8>), dim3(blocks), dim3(threads), 0, 0, dout, dcyc, iters); }, dcyc, nw);
        printf("%d waves/CU  v_fma_f64, 8 independent       : %.1f cycles per FMA per wave\n", wpb, c / iters / 8);
    }
    return bad != 0;
}
