// Host-side helpers of libwxhip.so: status/error plumbing shared by every entry point.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/wxhip.h"

namespace wx {

char* last_error_buf();  // thread-local, 512 bytes (wx_api.hip)

inline wx_status fail(wx_status st, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return st;
}

#define WX_HIP_TRY(expr)                                                                              \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return ::wx::fail(WX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                              \
    } while (0)

// Every entry point enqueues on a caller-supplied stream.  A stream belongs to one device, and a launch on it
// from a thread whose CURRENT device is another one fails with an invalid-handle error (a process that drives
// several GPUs, or a rank whose current device is not the plan's): make the stream's device current for the
// duration of the call.  Costs two runtime queries per call; a no-op for the null stream.
struct StreamDeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit StreamDeviceGuard(hipStream_t st) {
        hipDevice_t dev = -1;
        if (st != nullptr && hipStreamGetDevice(st, &dev) == hipSuccess && hipGetDevice(&prev) == hipSuccess &&
            (int)dev != prev)
            switched = hipSetDevice((int)dev) == hipSuccess;
    }
    ~StreamDeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    StreamDeviceGuard(const StreamDeviceGuard&) = delete;
    StreamDeviceGuard& operator=(const StreamDeviceGuard&) = delete;
};
#define WX_STREAM(st, stream)                             \
    hipStream_t st = static_cast<hipStream_t>(stream);    \
    ::wx::StreamDeviceGuard st##_device_guard(st)

// Pointers to device buffers inside parameter blocks.  A pointer that a kernel LOADS from memory (the parameter table of a
// batched launch) has no known address space, and the compiler addresses through it with FLAT instructions - which count
// against the LDS counter as well as the vector-memory one, so that every wait for an LDS read also waits for the global
// loads in flight.  (A by-value kernel argument is known to be global; the cast-and-back and __builtin_assume idioms do not
// survive to code generation with this compiler; typing the members themselves with the address space breaks every host
// assignment, because host functions are checked in the device pass too.)  gp<X> is a pointer for the host - same size,
// same layout, implicit conversions - and on the device every access through it is made in the global address space.
#if defined(__HIP_DEVICE_COMPILE__)
#define WX_GLOBAL __attribute__((address_space(1)))
#else
#define WX_GLOBAL
#endif
template <typename X>
struct gp {
    X* p;
    gp() = default;
    __host__ __device__ gp(X* q) : p(q) {}
    __host__ __device__ gp(decltype(nullptr)) : p(nullptr) {}
    template <typename Y, typename = typename std::enable_if<std::is_convertible<Y*, X*>::value && !std::is_same<Y, X>::value>::type>
    __host__ __device__ gp(gp<Y> o) : p(o.p) {}
    __host__ operator X*() const { return p; }   // (host only: device code that would fall back to a generic pointer does not compile)
    __host__ __device__ X* raw() const { return p; }
    __host__ __device__ explicit operator bool() const { return p != nullptr; }
    __host__ __device__ bool operator==(decltype(nullptr)) const { return p == nullptr; }
    __host__ __device__ bool operator!=(decltype(nullptr)) const { return p != nullptr; }
    __host__ __device__ gp operator+(long long o) const { return gp(p + o); }
    __host__ __device__ gp operator+(unsigned long long o) const { return gp(p + o); }
    __host__ __device__ gp operator+(long o) const { return gp(p + o); }
    __host__ __device__ gp operator+(unsigned long o) const { return gp(p + o); }
    __host__ __device__ gp operator+(int o) const { return gp(p + o); }
    __host__ __device__ gp operator+(unsigned o) const { return gp(p + o); }
    __host__ __device__ long long operator-(gp o) const { return p - o.p; }
    __host__ __device__ X WX_GLOBAL* g() const { return (X WX_GLOBAL*)p; }
    template <typename I>
    __host__ __device__ X WX_GLOBAL& operator[](I i) const { return g()[i]; }
    __host__ __device__ X WX_GLOBAL& operator*() const { return *g(); }
    __host__ __device__ X WX_GLOBAL* operator->() const { return g(); }
};
// ... of the state's value type T: global for float64; complex / dual values are class types, which C++ cannot copy out of
// a qualified address space without per-member loads - those keep plain (generic) pointers, their metric loads are global
template <typename T, typename X>
using tp = typename std::conditional<std::is_same<T, double>::value, gp<X>, X*>::type;
// ... and switchable per instantiation: G = true for kernels whose parameter block comes out of a device table
template <typename T, typename X, bool G>
using pp = typename std::conditional<G && std::is_same<T, double>::value, gp<X>, X*>::type;
static_assert(sizeof(gp<const double>) == sizeof(const double*), "a gp is a pointer");

}  // namespace wx
