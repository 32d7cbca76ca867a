// Launchers (grid shapes, XCD-aware element order is inside the kernels) and the column-form kernels' entry points.
#pragma once

namespace wx {

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// grid of a single-tile launch of one-element workgroups: the region's own shape, in slot order (decode_elem_grid)
static inline dim3 region_grid(int region, int H, int V) {
    const int w = H > 2 ? H - 2 : 0;
    if (region == WX_REGION_ALL) return dim3(H, H, V);
    if (region == WX_REGION_INTERIOR) return dim3(w, w, V);
    return dim3(H * H - w * w, 1, V);
}

template <int N, typename T>
static wx_status launch_extrap(const EulerParams<T>& P, hipStream_t st) {
    using C = Cfg<N>;
    if constexpr (grid3_for<N>()) {
        hipLaunchKernelGGL((euler_extrap_kernel<N, T>), region_grid(WX_REGION_ALL, P.H, P.V), dim3(C::BS), 0, st, P);
    } else {
        const int grid = (P.nelem + C::EPB - 1) / C::EPB;
        hipLaunchKernelGGL((euler_extrap_kernel<N, T>), dim3(grid), dim3(C::BS), 0, st, P);
    }
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N, bool PIPE>
__global__ __launch_bounds__(Cfg<N>::BS, kK2Waves) void euler_rhs_column_kernel(const EulerParams<double> P) {
    euler_rhs_body<N, double, PIPE, true>(P);
}

template <int N>
static wx_status launch_rhs_column(const EulerParams<double>& P, hipStream_t st) {
    using C = Cfg<N>;
    if (P.count == 0) return WX_OK;
    const int grid = (P.count + C::EPB - 1) / C::EPB;
    if (P.itf_out != nullptr) hipLaunchKernelGGL((euler_rhs_column_kernel<N, true>), dim3(8 * ((grid + 7) / 8)), dim3(C::BS), 0, st, P);
    else hipLaunchKernelGGL((euler_rhs_column_kernel<N, false>), dim3(8 * ((grid + 7) / 8)), dim3(C::BS), 0, st, P);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

static wx_status dispatch_rhs_column(int n, const EulerParams<double>& P, hipStream_t st) {
    switch (n) {
        case 2: return launch_rhs_column<2>(P, st);
        case 3: return launch_rhs_column<3>(P, st);
        case 4: return launch_rhs_column<4>(P, st);
        case 5: return launch_rhs_column<5>(P, st);
        case 6: return launch_rhs_column<6>(P, st);
        case 7: return launch_rhs_column<7>(P, st);
        case 8: return launch_rhs_column<8>(P, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", n);
}

template <int N, typename T>
static wx_status launch_rhs(const EulerParams<T>& P, hipStream_t st) {
    using C = Cfg<N>;
    if (P.count == 0) return WX_OK;
    if constexpr (grid3_for<N>()) {
        const dim3 grid = region_grid(P.region, P.H, P.V);
        if (P.itf_out != nullptr) hipLaunchKernelGGL((euler_rhs_kernel<N, T, true>), grid, dim3(C::BS), 0, st, P);
        else hipLaunchKernelGGL((euler_rhs_kernel<N, T, false>), grid, dim3(C::BS), 0, st, P);
    } else {
        const int grid = (P.count + C::EPB - 1) / C::EPB;
        if (P.itf_out != nullptr) hipLaunchKernelGGL((euler_rhs_kernel<N, T, true>), dim3(grid), dim3(C::BS), 0, st, P);
        else hipLaunchKernelGGL((euler_rhs_kernel<N, T, false>), dim3(grid), dim3(C::BS), 0, st, P);
    }
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N, typename T>
static wx_status launch_extrap_batch(const EulerParams<T>* table, const EulerBatchDyn<T>& dyn, int nelem, int ntiles,
                                     hipStream_t st) {
    using C = Cfg<N>;
    const int grid = (nelem + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_extrap_batch_kernel<N, T>), dim3(grid, ntiles), dim3(C::BS), 0, st, table, dyn);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N, typename T>
static wx_status launch_rhs_batch(const EulerParams<T>* table, const EulerBatchDyn<T>& dyn, int ntiles, hipStream_t st) {
    using C = Cfg<N>;
    if (dyn.count == 0) return WX_OK;
    const int grid = (dyn.count + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_rhs_batch_kernel<N, T>), dim3(grid, ntiles), dim3(C::BS), 0, st, table, dyn);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N>
static wx_status launch_jvp_batch(const EulerParams<dual>* table, const EulerBatchDyn<dual>& dyn, int ntiles, hipStream_t st) {
    using C = Cfg<N>;
    if (dyn.count == 0) return WX_OK;
    const int grid = (dyn.count + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_jvp_batch_kernel<N>), dim3(grid, ntiles), dim3(C::BS), 0, st, table, dyn);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N>
static wx_status launch_jvp(const EulerParams<dual>& P, hipStream_t st) {
    using C = Cfg<N>;
    if (P.count == 0) return WX_OK;
    if constexpr (grid3_for<N>()) {
        hipLaunchKernelGGL((euler_jvp_kernel<N>), region_grid(P.region, P.H, P.V), dim3(C::BS), 0, st, P);
    } else {
        const int grid = (P.count + C::EPB - 1) / C::EPB;
        hipLaunchKernelGGL((euler_jvp_kernel<N>), dim3(grid), dim3(C::BS), 0, st, P);
    }
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::BS, kJvpWaves) void euler_jvp_column_kernel(const EulerParams<dual> P) {
    if constexpr (N == 8 && WX_MFMA) euler_jvp_body_mf<true>(P);
    else euler_jvp_body<N, true>(P);
}

template <int N>
static wx_status launch_jvp_column(const EulerParams<dual>& P, hipStream_t st) {
    using C = Cfg<N>;
    if (P.count == 0) return WX_OK;
    const int grid = (P.count + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_jvp_column_kernel<N>), dim3(8 * ((grid + 7) / 8)), dim3(C::BS), 0, st, P);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

static wx_status dispatch_jvp_column(int n, const EulerParams<dual>& P, hipStream_t st) {
    switch (n) {
        case 2: return launch_jvp_column<2>(P, st);
        case 3: return launch_jvp_column<3>(P, st);
        case 4: return launch_jvp_column<4>(P, st);
        case 5: return launch_jvp_column<5>(P, st);
        case 6: return launch_jvp_column<6>(P, st);
        case 7: return launch_jvp_column<7>(P, st);
        case 8: return launch_jvp_column<8>(P, st);
    }
    return fail(WX_ERR_UNSUPPORTED, "num_solpts %d not in 2..8", n);
}

}  // namespace wx
