// Development probe: operand / result lane maps of v_mfma_f64_4x4x4_4b_f64 on gfx950, found empirically:
// A = unit at lane la, B = unit at lane lb -> which result lanes are non-zero.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void probe(int la, int lb, double* out) {
    const int l = threadIdx.x;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
    out[l] = d;
}

int main() {
    double* dout;
    hipMalloc(&dout, 64 * 8);
    std::vector<double> h(64);
    // block 0 (lanes 0..15) and one cross-block check
    printf("la lb -> result lanes (block 0 operands)\n");
    for (int la = 0; la < 16; ++la) {
        for (int lb = 0; lb < 16; ++lb) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, la, lb, dout);
            hipMemcpy(h.data(), dout, 64 * 8, hipMemcpyDeviceToHost);
            int n = 0, first = -1;
            for (int l = 0; l < 64; ++l)
                if (h[l] != 0.0) { ++n; if (first < 0) first = l; }
            if (n) printf("A@%2d B@%2d -> %d lane(s), first %2d\n", la, lb, n, first);
        }
    }
    for (int la : {16, 17, 20}) for (int lb : {16, 17, 20, 0}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, la, lb, dout);
        hipMemcpy(h.data(), dout, 64 * 8, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) if (h[l] != 0.0) printf("A@%2d B@%2d -> lane %2d\n", la, lb, l);
    }
    return 0;
}
