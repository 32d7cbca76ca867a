// libwxhip.so: error string, version, device probe.
#include "wx_common.h"
#include "wx_math.h"
#include "wx_mfma.h"

#include <cmath>
#include <new>

#ifndef WX_K2_DIAG
#define WX_K2_DIAG 0
#endif
#define WX_STR2(x) #x
#define WX_STR(x) WX_STR2(x)

namespace wx {
char* last_error_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace wx

extern "C" {

const char* wx_last_error(void) { return wx::last_error_buf(); }

const char* wx_version(void) { return "wxhip 0.1.0 gfx950"; }

// the build switches this library was compiled with (measurement provenance: bench.py refuses counter figures taken
// on another variant)
const char* wx_build_info(void) { return "WX_MFMA=" WX_STR(WX_MFMA) " WX_K2_DIAG=" WX_STR(WX_K2_DIAG); }

int wx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

}  // extern "C"

// ---- the streaming copy a roofline fraction is put beside: what this part sustains on a pure read-once / write-once
// pattern (16 bytes per lane, grid-stride, one pass) - the achievable side of "achieved vs 8 TB/s"
// U loads of 16 bytes in flight per lane before the first store; NT: non-temporal loads and stores (the data is used once)
template <int U, bool NT>
__global__ __launch_bounds__(256) void wx_stream_copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        double2 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const double2* p = src + i + k * stride;
            if (NT) { v[k].x = __builtin_nontemporal_load(&p->x); v[k].y = __builtin_nontemporal_load(&p->y); }
            else v[k] = *p;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            double2* p = dst + i + k * stride;
            if (NT) { __builtin_nontemporal_store(v[k].x, &p->x); __builtin_nontemporal_store(v[k].y, &p->y); }
            else *p = v[k];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

template <int U, bool NT>
static void launch_copy(const void* src, void* dst, size_t n16, int wg_per_cu, hipStream_t st) {
    size_t blocks = (n16 / U + 255) / 256;
    if (blocks > (size_t)256 * wg_per_cu) blocks = (size_t)256 * wg_per_cu;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL((wx_stream_copy_kernel<U, NT>), dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const double2*>(src),
                       static_cast<double2*>(dst), n16);
}

// development entry point (not in wxhip.h; tools/copybench.py): the copy with its launch shape as arguments
extern "C" wx_status wx_stream_copy_variant(const void* src, void* dst, size_t bytes, int unroll, int nt, int wg_per_cu,
                                            wx_stream stream) {
    if (!src || !dst) return wx::fail(WX_ERR_INVALID, "wx_stream_copy: null argument");
    if (bytes % 16 != 0 || ((uintptr_t)src | (uintptr_t)dst) % 16 != 0)
        return wx::fail(WX_ERR_INVALID, "wx_stream_copy: buffers and size must be multiples of 16 bytes");
    if (bytes == 0) return WX_OK;
    if (wg_per_cu < 1 || wg_per_cu > 64) return wx::fail(WX_ERR_INVALID, "wx_stream_copy: %d workgroups per CU", wg_per_cu);
    WX_STREAM(st, stream);
    const size_t n16 = bytes / 16;
    switch (unroll * 2 + (nt ? 1 : 0)) {
        case 2: launch_copy<1, false>(src, dst, n16, wg_per_cu, st); break;
        case 3: launch_copy<1, true>(src, dst, n16, wg_per_cu, st); break;
        case 4: launch_copy<2, false>(src, dst, n16, wg_per_cu, st); break;
        case 5: launch_copy<2, true>(src, dst, n16, wg_per_cu, st); break;
        case 8: launch_copy<4, false>(src, dst, n16, wg_per_cu, st); break;
        case 9: launch_copy<4, true>(src, dst, n16, wg_per_cu, st); break;
        case 16: launch_copy<8, false>(src, dst, n16, wg_per_cu, st); break;
        case 17: launch_copy<8, true>(src, dst, n16, wg_per_cu, st); break;
        default: return wx::fail(WX_ERR_INVALID, "wx_stream_copy: unroll %d not in {1, 2, 4, 8}", unroll);
    }
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

// the shape that measured fastest on MI355X (tools/copybench.py, profiles/r04_copybench.log)
constexpr int kCopyUnroll = 2, kCopyNT = 1, kCopyWgPerCu = 2;   // the best of tools/copybench.py (profiles/r04_copybench.log)
extern "C" wx_status wx_stream_copy(const void* src, void* dst, size_t bytes, wx_stream stream) {
    return wx_stream_copy_variant(src, dst, bytes, kCopyUnroll, kCopyNT, kCopyWgPerCu, stream);
}

// the read side alone: every lane sums what it streams (16 bytes per load, 4 loads in flight), one double per workgroup comes out
__global__ __launch_bounds__(256) void wx_stream_read_kernel(const double2* __restrict__ src, size_t n16, double* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const double2 *p0 = src + i, *p1 = p0 + stride, *p2 = p1 + stride, *p3 = p2 + stride;
        const double x0 = __builtin_nontemporal_load(&p0->x), y0 = __builtin_nontemporal_load(&p0->y);
        const double x1 = __builtin_nontemporal_load(&p1->x), y1 = __builtin_nontemporal_load(&p1->y);
        const double x2 = __builtin_nontemporal_load(&p2->x), y2 = __builtin_nontemporal_load(&p2->y);
        const double x3 = __builtin_nontemporal_load(&p3->x), y3 = __builtin_nontemporal_load(&p3->y);
        a0 += x0 + y0; a1 += x1 + y1; a2 += x2 + y2; a3 += x3 + y3;
    }
    for (; i < n16; i += stride) a0 += src[i].x + src[i].y;
    double a = (a0 + a1) + (a2 + a3);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

extern "C" int wx_stream_read_sink_doubles(void) { return 256 * 8; }

extern "C" wx_status wx_stream_read(const void* src, size_t bytes, double* sink, wx_stream stream) {
    if (!src || !sink) return wx::fail(WX_ERR_INVALID, "wx_stream_read: null argument");
    if (bytes % 16 != 0 || (uintptr_t)src % 16 != 0) return wx::fail(WX_ERR_INVALID, "wx_stream_read: buffer and size must be multiples of 16 bytes");
    WX_STREAM(st, stream);
    hipLaunchKernelGGL(wx_stream_read_kernel, dim3(256 * 8), dim3(256), 0, st, static_cast<const double2*>(src), bytes / 16, sink);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

// ---- the nine-stamp timing row of the reference's RHS (rhs/rhs.py:39-41, 68-118) for callers without torch
struct wx_phase_timer {
    hipEvent_t ev[9];
    bool set[9];
};

extern "C" {

wx_status wx_phase_timer_create(wx_phase_timer** out) {
    if (!out) return wx::fail(WX_ERR_INVALID, "wx_phase_timer_create: null argument");
    *out = nullptr;
    wx_phase_timer* t = new (std::nothrow) wx_phase_timer();
    if (!t) return wx::fail(WX_ERR_NOMEM, "out of host memory");
    for (int i = 0; i < 9; ++i) {
        t->set[i] = false;
        hipError_t e = hipEventCreate(&t->ev[i]);
        if (e != hipSuccess) {
            for (int k = 0; k < i; ++k) (void)hipEventDestroy(t->ev[k]);
            delete t;
            return wx::fail(WX_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
        }
    }
    *out = t;
    return WX_OK;
}

wx_status wx_phase_timer_destroy(wx_phase_timer* t) {
    if (!t) return WX_OK;
    for (int i = 0; i < 9; ++i) (void)hipEventDestroy(t->ev[i]);
    delete t;
    return WX_OK;
}

wx_status wx_phase_timer_stamp(wx_phase_timer* t, int slot, wx_stream stream) {
    if (!t || slot < 0 || slot > 8) return wx::fail(WX_ERR_INVALID, "wx_phase_timer_stamp: bad timer or slot %d", slot);
    WX_STREAM(st, stream);
    if (slot == 0)
        for (int i = 0; i < 9; ++i) t->set[i] = false;   // a new evaluation
    WX_HIP_TRY(hipEventRecord(t->ev[slot], st));
    t->set[slot] = true;
    return WX_OK;
}

wx_status wx_phase_timer_elapsed(wx_phase_timer* t, double seconds[9]) {
    if (!t || !seconds) return wx::fail(WX_ERR_INVALID, "wx_phase_timer_elapsed: null argument");
    if (!t->set[0] || !t->set[8]) return wx::fail(WX_ERR_INVALID, "wx_phase_timer_elapsed: slots 0 and 8 must be stamped");
    // every stamped event is waited for, not only the last: the slots may have been stamped on different streams
    for (int i = 0; i <= 8; ++i)
        if (t->set[i]) WX_HIP_TRY(hipEventSynchronize(t->ev[i]));
    int prev = 0;
    for (int i = 1; i <= 8; ++i) {
        seconds[i - 1] = 0.0;
        if (!t->set[i]) continue;   // this phase shares its kernel with the next stamped one
        float ms = 0.0f;
        WX_HIP_TRY(hipEventElapsedTime(&ms, t->ev[prev], t->ev[i]));
        seconds[i - 1] = ms > 0.0f ? 1e-3 * ms : 0.0;   // (stamps of two streams: a phase that ended before its predecessor waited 0)
        prev = i;
    }
    float total = 0.0f;
    WX_HIP_TRY(hipEventElapsedTime(&total, t->ev[0], t->ev[8]));
    seconds[8] = 1e-3 * total;
    return WX_OK;
}

wx_status wx_phase_timer_since_start(wx_phase_timer* t, double seconds[9]) {
    if (!t || !seconds) return wx::fail(WX_ERR_INVALID, "wx_phase_timer_since_start: null argument");
    if (!t->set[0]) return wx::fail(WX_ERR_INVALID, "wx_phase_timer_since_start: slot 0 must be stamped");
    for (int i = 0; i <= 8; ++i)
        if (t->set[i]) WX_HIP_TRY(hipEventSynchronize(t->ev[i]));
    seconds[0] = 0.0;
    for (int i = 1; i <= 8; ++i) {
        seconds[i] = -1.0;
        if (!t->set[i]) continue;
        float ms = 0.0f;
        WX_HIP_TRY(hipEventElapsedTime(&ms, t->ev[0], t->ev[i]));
        seconds[i] = 1e-3 * ms;
    }
    return WX_OK;
}

wx_status wx_stream_priority_range(int* least, int* greatest) {
    int lo = 0, hi = 0;
    WX_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (least) *least = lo;
    if (greatest) *greatest = hi;
    return WX_OK;
}

wx_status wx_stream_create(wx_stream* out, int priority_class) {
    if (!out) return wx::fail(WX_ERR_INVALID, "wx_stream_create: null argument");
    *out = nullptr;
    int lo = 0, hi = 0;   // numerically: least >= greatest
    WX_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const int prio = priority_class > 0 ? lo : (priority_class < 0 ? hi : (lo + hi) / 2);
    hipStream_t st = nullptr;
    WX_HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio));
    *out = st;
    return WX_OK;
}

wx_status wx_stream_destroy(wx_stream stream) {
    if (!stream) return WX_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipStreamSynchronize(st);
    WX_HIP_TRY(hipStreamDestroy(st));
    return WX_OK;
}

// the float64 logarithm of the one-kernel form (wx_math.h: lean_log) on an array - so that its accuracy is a tested statement
__global__ __launch_bounds__(256) void wx_lean_log_kernel(const double* __restrict__ x, double* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = wx::lean_log(x[i]);
}

wx_status wx_lean_log(const double* x, double* y, size_t n, wx_stream stream) {
    if (n == 0) return WX_OK;
    if (!x || !y) return wx::fail(WX_ERR_INVALID, "wx_lean_log: null argument");
    WX_STREAM(st, stream);
    hipLaunchKernelGGL(wx_lean_log_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

// ---- fgmres' host side (solvers/fgmres.py:75-94, 202-262): the Givens rotations of the Hessenberg columns, the residual estimate and
// the stopping test, the back substitution - O(restart^2) scalar work per cycle that the reference does in Python on the host.  With
// the Gram-Schmidt step on the device (wx_fgmres_vector) these loops were a sixth of a Rosenbrock step at the shipped .ini sizes
// (7.5 us per column in the interpreter against a 39 us Krylov vector).  Plain host C, IEEE operations in the reference's order
// (no contraction; the squares through pow as Python's ** 2 takes them): the same bits as the interpreted loop.
#pragma clang fp contract(off)
static void wx_rotg(double a, double b, double* c, double* s) {
    if (b == 0.0) { *c = 1.0; *s = 0.0; return; }
    if (a == 0.0) { *c = 0.0; *s = 1.0; return; }
    const double fa = fabs(a), fb = fabs(b);
    const double scl = fa < fb ? fa : fb;
    const double sigma = fa > fb ? copysign(1.0, a) : copysign(1.0, b);
    const double r = sigma * (scl * sqrt(pow(a / scl, 2.0) + pow(b / scl, 2.0)));
    *c = a / r;
    *s = b / r;
}

int wx_fgmres_rotate_columns(const double* R, int ld, int j0, int j1, int restart, const double* vn, double* cs, double* sn, double* g,
                             double* Hm, double tol_abs, double* rate, double* res, int* stopped) {
    if (stopped) *stopped = 0;
    if (!R || !vn || !cs || !sn || !g || !Hm || !rate || !res || !stopped || ld < 2 || j0 < 0 || j1 > ld - 2 || j1 > restart) return -1;
    double hj[258];
    if (ld > 258) return -1;
    for (int j = j0; j < j1; ++j) {
        for (int i = 0; i < j + 2; ++i) hj[i] = R[(size_t)i * ld + (j + 1)];
        for (int i = 0; i < j; ++i) {   // previous rotations
            const double t = cs[i] * hj[i] + sn[i] * hj[i + 1];
            hj[i + 1] = -sn[i] * hj[i] + cs[i] * hj[i + 1];
            hj[i] = t;
        }
        double c = 1.0, s = 0.0;
        if (hj[j + 1] != 0.0) {
            wx_rotg(hj[j], hj[j + 1], &c, &s);
            hj[j] = c * hj[j] + s * hj[j + 1];
            hj[j + 1] = 0.0;
            const double g0 = g[j], g1 = g[j + 1];
            g[j] = c * g0 + s * g1;
            g[j + 1] = -s * g0 + c * g1;
        }
        cs[j] = c;
        sn[j] = s;
        for (int i = 0; i < j + 2; ++i) Hm[(size_t)j * ld + i] = hj[i];
        if (g[j] != 0.0 && fabs(g[j + 1]) > 0.0) {   // the running decay of the residual estimate (NaN: none yet)
            const double q = fabs(g[j + 1]) / fabs(g[j]);
            *rate = (j == 0 || *rate != *rate) ? q : 0.5 * (*rate + q);
        }
        const double v_norm = vn[j + 1], norm_r = fabs(g[j + 1]);
        res[j - j0] = norm_r;
        if (j < restart - 1 || v_norm == 0.0) {
            if (norm_r < tol_abs || norm_r != norm_r || v_norm == 0.0) {   // converged, NaN, or breakdown
                *stopped = 1;
                return j - j0 + 1;
            }
        }
    }
    return j1 - j0;
}

int wx_fgmres_back_substitute(const double* Hm, int ld, int k, const double* g, double* y) {
    if (!Hm || !g || !y || k < 0 || k > ld) return -1;
    for (int i = k - 1; i >= 0; --i) {
        double acc = g[i];
        for (int l = i + 1; l < k; ++l) acc -= Hm[(size_t)l * ld + i] * y[l];
        y[i] = acc / Hm[(size_t)i * ld + i];
    }
    return 0;
}

}  // extern "C"
