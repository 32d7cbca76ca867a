// What puts 3-4 us in front of the Euler batch kernels' start when the kernel before them in the stream has ended (kernel trace of
// the FGMRES device passes: Krylov kernels follow each other 0.1 us apart, the Euler kernels start 3-4 us late)?  Candidates,
// one dummy kernel each, alternating with a small kernel: a 512-byte argument block, 100 KB of static LDS, 200 registers,
// a 64 KB code body.  Development tool: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/gap_probe tools/gap_probe.hip; run under
// rocprofv3 --kernel-trace and read the gaps with tools/trace_gaps.py.
#include <hip/hip_runtime.h>
#include <cstdio>

struct Big { double v[64]; };
// ~17 us of nothing, so that the host stays ahead of the queue and the gaps are the device's own
__device__ __forceinline__ void linger() {
    for (int i = 0; i < 5; ++i) __builtin_amdgcn_s_sleep(127);
}

__global__ void k_small(double* p) { linger(); p[threadIdx.x + blockIdx.x * blockDim.x] += 1.0; }
__global__ void k_bigarg(double* p, Big b) { linger(); p[threadIdx.x + blockIdx.x * blockDim.x] += b.v[threadIdx.x & 63]; }
__global__ void k_lds(double* p) {
    linger();
    __shared__ double s[12800];   // 100 KB
    for (int i = threadIdx.x; i < 12800; i += blockDim.x) s[i] = p[i & 255];
    __syncthreads();
    p[threadIdx.x + blockIdx.x * blockDim.x] += s[(threadIdx.x * 37) % 12800];
}
__global__ __launch_bounds__(256, 1) void k_vgpr(double* p) {
    linger();
    double a[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) a[i] = p[(threadIdx.x + i * 256) & 65535];
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += a[i] * a[(i * 7) % 96];
    p[threadIdx.x + blockIdx.x * blockDim.x] += s;
}
#define REP10(X) X X X X X X X X X X
__global__ void k_code(double* p, int sel) {   // a long straight-line body (~60 KB of instructions), only its head executed
    linger();
    double x = p[threadIdx.x + blockIdx.x * blockDim.x];
    if (sel == 12345) { REP10(REP10(REP10(REP10(x = x * 1.0000001 + 3.0;)))) }   // (a dependent chain: not folded)
    p[threadIdx.x + blockIdx.x * blockDim.x] = x + 1.0;
}

__global__ void k_grid2(double* p) { linger(); p[threadIdx.x + (blockIdx.x + blockIdx.y * gridDim.x) * blockDim.x] += 1.0; }

int main() {
    double* p;
    hipMalloc(&p, 1 << 20);
    hipMemset(p, 0, 1 << 20);
    Big b;
    for (int i = 0; i < 64; ++i) b.v[i] = i;
    hipStream_t st;
    hipStreamCreate(&st);
    for (int it = 0; it < 400; ++it) {
        hipLaunchKernelGGL(k_small, dim3(81), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_bigarg, dim3(81), dim3(256), 0, st, p, b);
        hipLaunchKernelGGL(k_small, dim3(81), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_lds, dim3(81), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_small, dim3(81), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_vgpr, dim3(81), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_small, dim3(81), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_code, dim3(81), dim3(256), 0, st, p, it);
        hipLaunchKernelGGL(k_small, dim3(81), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_grid2, dim3(16, 6), dim3(256), 0, st, p);
    }
    hipStreamSynchronize(st);
    printf("done\n");
    return 0;
}
