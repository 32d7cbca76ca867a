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

// ---- the low-order one-kernel form (euler3d_brick.h)
static inline int ceil_log2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

// the boxes of a region and their brick shapes: lk as large as the order's default and the tile's depth allow, the in-plane
// shape of each box the one that pads its area least (then the squarest, then the longest along i)
static inline BrickBoxes brick_boxes(int log_epb, int lk_default, int region, int H, int V) {
    BrickBoxes G;
    memset(&G, 0, sizeof(G));
    const int w = H > 2 ? H - 2 : 0;
    int nb = 0;
    auto box = [&](int i0, int i1, int j0, int j1) {
        if (i1 > i0 && j1 > j0) { G.i0[nb] = i0; G.i1[nb] = i1; G.j0[nb] = j0; G.j1[nb] = j1; ++nb; }
    };
    if (region == WX_REGION_ALL) box(0, H, 0, H);
    else if (region == WX_REGION_INTERIOR) box(1, 1 + w, 1, 1 + w);
    else {
        box(0, H, 0, 1);                        // south row
        if (H > 1) box(0, H, H - 1, H);         // north row
        box(0, 1, 1, 1 + w);                    // west column
        if (H > 1) box(H - 1, H, 1, 1 + w);     // east column
    }
    G.lk = lk_default < ceil_log2(V) ? lk_default : ceil_log2(V);
    const int lp = log_epb - G.lk;
    int starts[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < nb; ++b) {
        const int wi = G.i1[b] - G.i0[b], wj = G.j1[b] - G.j0[b];
        long best = -1;
        int bli = lp, blj = 0;
        for (int li = lp; li >= 0; --li) {
            const int lj = lp - li;
            const long area = (long)(((wi + (1 << li) - 1) >> li) << li) * (long)(((wj + (1 << lj) - 1) >> lj) << lj);
            const int skew = li > lj ? li - lj : lj - li;
            const long cost = area * 64 + skew;   // padded area first, squareness second, longer along i on ties (li descends)
            if (best < 0 || cost < best) { best = cost; bli = li; blj = lj; }
        }
        G.li[b] = bli; G.lj[b] = blj;
        G.nbi[b] = (wi + (1 << bli) - 1) >> bli;
        const int nbj = (wj + (1 << blj) - 1) >> blj;
        starts[b + 1] = starts[b] + G.nbi[b] * nbj;
    }
    G.plane = starts[nb];
    G.start1 = nb > 1 ? starts[1] : 0x7fffffff;
    G.start2 = nb > 2 ? starts[2] : 0x7fffffff;
    G.start3 = nb > 3 ? starts[3] : 0x7fffffff;
    const int nbk = (V + (1 << G.lk) - 1) >> G.lk;
    G.nbricks = G.plane * nbk;
    auto magic = [&](int d, long long nmax) -> unsigned {
        return (d >= 2 && nmax * (long long)d < (1ll << 32)) ? (unsigned)((1ull << 32) / (unsigned)d) + 1u : 0u;
    };
    G.md_plane = magic(G.plane, (long long)G.nbricks + 8);
    for (int b = 0; b < 4; ++b) {
        if (b >= nb) { G.nbi[b] = 1; G.i1[b] = G.i0[b] = G.j0[b] = G.j1[b] = 0; }
        G.mdi[b] = magic(G.nbi[b], G.plane + 8);
    }
    return G;
}

template <int N>
constexpr int brick_lk_default() { return N == 4 ? 0 : 1; }

template <int N>
static wx_status launch_brick(const EulerParams<double>& P, bool epi, hipStream_t st) {
    if constexpr (BrickCfg<N>::on) {
        using C = BrickCfg<N>;
        if (P.count == 0) return WX_OK;
        const BrickBoxes GB = brick_boxes(C::LOG_EPB, brick_lk_default<N>(), P.region, P.H, P.V);
        const dim3 grid(8 * ((GB.nbricks + 7) / 8));
        if (epi) hipLaunchKernelGGL((euler_brick_kernel<N, true>), grid, dim3(C::BS), 0, st, P, GB);
        else hipLaunchKernelGGL((euler_brick_kernel<N, false>), grid, dim3(C::BS), 0, st, P, GB);
        WX_HIP_TRY(hipGetLastError());
        return WX_OK;
    } else {
        return fail(WX_ERR_UNSUPPORTED, "the one-kernel form serves num_solpts 2..4, not %d", N);
    }
}

template <int N>
static wx_status launch_brick_batch(const EulerParams<double>* table, const EulerBatchDyn<double>& dyn, int H, int V, int ntiles,
                                    hipStream_t st) {
    if constexpr (BrickCfg<N>::on) {
        using C = BrickCfg<N>;
        if (dyn.count == 0) return WX_OK;
        const BrickBoxes GB = brick_boxes(C::LOG_EPB, brick_lk_default<N>(), dyn.region, H, V);
        const dim3 grid(8 * ((GB.nbricks + 7) / 8), ntiles);
        hipLaunchKernelGGL((euler_brick_batch_kernel<N>), grid, dim3(C::BS), 0, st, table, dyn, GB);
        WX_HIP_TRY(hipGetLastError());
        return WX_OK;
    } else {
        return fail(WX_ERR_UNSUPPORTED, "the one-kernel form serves num_solpts 2..4, not %d", N);
    }
}

template <int N>
static wx_status launch_jvp_brick(const EulerParams<dual>& P, hipStream_t st) {
    if constexpr (BrickCfg<N>::on) {
        using C = BrickCfg<N>;
        if (P.count == 0) return WX_OK;
        const BrickBoxes GB = brick_boxes(C::LOG_EPB, brick_lk_default<N>(), P.region, P.H, P.V);
        hipLaunchKernelGGL((euler_jvp_brick_kernel<N>), dim3(8 * ((GB.nbricks + 7) / 8)), dim3(C::BS), 0, st, P, GB);
        WX_HIP_TRY(hipGetLastError());
        return WX_OK;
    } else {
        return fail(WX_ERR_UNSUPPORTED, "the one-kernel form serves num_solpts 2..4, not %d", N);
    }
}

template <int N>
static wx_status launch_jvp_brick_batch(const EulerParams<dual>* table, const EulerBatchDyn<dual>& dyn, int H, int V, int ntiles,
                                        hipStream_t st) {
    if constexpr (BrickCfg<N>::on) {
        using C = BrickCfg<N>;
        if (dyn.count == 0) return WX_OK;
        const BrickBoxes GB = brick_boxes(C::LOG_EPB, brick_lk_default<N>(), dyn.region, H, V);
        hipLaunchKernelGGL((euler_jvp_brick_batch_kernel<N>), dim3(8 * ((GB.nbricks + 7) / 8), ntiles), dim3(C::BS), 0, st, table, dyn, GB);
        WX_HIP_TRY(hipGetLastError());
        return WX_OK;
    } else {
        return fail(WX_ERR_UNSUPPORTED, "the one-kernel form serves num_solpts 2..4, not %d", N);
    }
}

// the pack kernel of the one-kernel form: the ring's outward faces -> edge messages
template <int N, typename T>
static wx_status launch_pack(EulerParams<T> P, hipStream_t st) {
    using C = Cfg<N>;
    const int w = P.H > 2 ? P.H - 2 : 0;
    P.region = WX_REGION_BOUNDARY;
    P.count = P.V * (P.H * P.H - w * w);
    const int grid = (P.count + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_pack_kernel<N, T>), dim3(grid), dim3(C::BS), 0, st, P);
    WX_HIP_TRY(hipGetLastError());
    return WX_OK;
}

template <int N, typename T>
static wx_status launch_pack_batch(const EulerParams<T>* table, EulerBatchDyn<T> dyn, int H, int V, int ntiles, hipStream_t st) {
    using C = Cfg<N>;
    const int w = H > 2 ? H - 2 : 0;
    dyn.region = WX_REGION_BOUNDARY;
    dyn.count = V * (H * H - w * w);
    const int grid = (dyn.count + C::EPB - 1) / C::EPB;
    hipLaunchKernelGGL((euler_pack_batch_kernel<N, T>), dim3(grid, ntiles), dim3(C::BS), 0, st, table, dyn);
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
