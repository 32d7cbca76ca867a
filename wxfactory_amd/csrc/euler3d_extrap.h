// K1 of the 3-D Euler path: extrapolation of the state to the element faces + tile-edge pack (phases 1-2,
// rhs_dfr.py:50-71, 141-172; process_topology.py:269-386), the tangent-only form of the prepared JVP, the batched form.
#pragma once

namespace wx {

// Phase 1-2 on nodal values already staged in LDS (log rho, rho u1, rho u2, rho w, log rho*theta):
// one thread per face point extrapolates, exponentiates, writes the interface buffer and, on outward
// tile-edge faces, the rotated / flipped edge message.  Shared by K1 and by K2's stage-pipeline epilogue.
// PACK: the edge messages only, nothing to the interface buffer (the pack kernel of the low-order one-kernel form, euler3d_brick.h)
template <int N, typename T, bool COLM = false, bool G3 = false, bool PACK = false, bool G>
__device__ __forceinline__ void extrap_faces(const EulerParams<T, G>& P, T (*fld)[Cfg<N>::EPB * Cfg<N>::LE], int slot0,
                                             int count, int region, pp<T, T, G> itf_dst, pp<T, T, G> ss, pp<T, T, G> sn, pp<T, T, G> sw,
                                             pp<T, T, G> se) {
    using C = Cfg<N>;
    constexpr int N2 = C::N2, EPB = C::EPB, BS = C::BS;
    const int tid = threadIdx.x;
    __builtin_assume(tid < (int)Cfg<N>::BS);   // (the launch bounds: lets one-element workgroups drop their `le < EPB` guards)
    const int H = P.H, V = P.V;
    for (int fi = tid; fi < EPB * 6 * N2; fi += BS) {
        const int le = fi / (6 * N2);
        const int r = fi % (6 * N2);
        int f = r / N2;
        const int fp = r % N2;
        // a face is a whole number of waves when n^2 is a multiple of 64 (n = 8): tell the compiler, so that the
        // face's direction, strides and weights live in scalar registers
        if (N2 % 64 == 0 && BS % 64 == 0) f = __builtin_amdgcn_readfirstlane(f);
        const Elem el = COLM ? decode_elem_col(slot0 + le, count, region, H, V, P.md_v, P.md_h, P.md_w) : decode_blk<EPB, G3>(P, slot0 + le, count, region);
        if (!el.valid) continue;
        if (kNoVertFaces && f >= 4) continue;
        const int d = f >> 1, plus = f & 1;
        const int a = fp / N, b = fp % N;
        if (PACK && !((d == 0 && (plus ? el.ei == H - 1 : el.ei == 0)) || (d == 1 && (plus ? el.ej == H - 1 : el.ej == 0)))) continue;
        // point index of m-th node on the line normal to the face, and its stride
        int base, stride;
        if (d == 0) { base = C::lidx(a, b, 0); stride = 1; }           // (kl=a, jl=b, il=m)
        else if (d == 1) { base = C::lidx(a, 0, b); stride = C::NP; }  // (kl=a, jl=m, il=b)
        else { base = C::lidx(0, a, b); stride = N * C::NP; }          // (kl=m, jl=a, il=b)
        const auto w = plus ? P.K->ep : P.K->em;
        T s[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) s[v] = T(0.0);
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const double wm = w[m];
#pragma unroll
            for (int v = 0; v < 5; ++v) s[v] += wm * fld[v][le * C::LE + base + m * stride];
        }
        s[0] = w_exp(s[0]);
        s[4] = w_exp(s[4]);
        if (!PACK) {
            pp<T, T, G> dst = itf_dst + ((size_t)el.e * 6 + f) * NQ * N2 + fp;
#pragma unroll
            for (int v = 0; v < 5; ++v) dst[v * N2] = s[v];
        }

        // outward faces of the tile edge: rotate into the neighbour's basis, flip, pack
        int edge = -1, along = 0;
        double X = 0.0;
        if (d == 0 && ((plus && el.ei == H - 1) || (!plus && el.ei == 0))) {
            edge = plus ? E_E : E_W;
            along = el.ej;
            X = P.bwe[el.ej * N + b];
        } else if (d == 1 && ((plus && el.ej == H - 1) || (!plus && el.ej == 0))) {
            edge = plus ? E_N : E_S;
            along = el.ei;
            X = P.bsn[el.ei * N + b];
        }
        pp<T, T, G> sendp = edge == E_S ? ss : (edge == E_N ? sn : (edge == E_W ? sw : se));
        if (edge >= 0 && sendp != nullptr) {
            rotate_contra<T>(P.K->rot[edge], X, s[1], s[2]);
            int al = along, bb = b;
            if (P.K->flip[edge]) { al = H - 1 - along; bb = N - 1 - b; }
            const size_t eo = ((size_t)el.ek * H + al) * N2 + a * N + bb;
            const size_t vs = (size_t)V * H * N2;
            pp<T, T, G> out = sendp + eo;
#pragma unroll
            for (int v = 0; v < 5; ++v) out[v * vs] = s[v];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: extrapolation to element faces + tile-edge pack
// ------------------------------------------------------------------------------------------------
// PACK: the ring of boundary elements only (P.count = its size), edge messages only
template <int N, typename T, bool G3 = false, bool PACK = false, bool G>
__device__ __forceinline__ void euler_extrap_body(const EulerParams<T, G>& P) {
    using C = Cfg<N>;
    constexpr int N2 = C::N2, N3 = C::N3, EPB = C::EPB;
    __shared__ T fld[5][EPB * C::LE];

    const int tid = threadIdx.x;
    __builtin_assume(tid < (int)Cfg<N>::BS);   // (the launch bounds: lets one-element workgroups drop their `le < EPB` guards)
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;

    {
        const int le = tid / N3, pt = tid % N3;
        const Elem el = decode_blk<EPB, G3>(P, blockIdx.x * EPB + le, PACK ? P.count : P.nelem, PACK ? WX_REGION_BOUNDARY : WX_REGION_ALL);
        if (le < EPB && el.valid) {
            const size_t o = (size_t)el.e * N3 + pt;
            const int lp = le * C::LE + C::lidx(pt / N2, (pt / N) % N, pt % N);
            T a0, a1, a2, a3, a4;
            load_state<T>(P, o, fs, a0, a1, a2, a3, a4);
            if constexpr (PACK && !std::is_same<T, cplx>::value) {   // (the one-kernel form's logarithm: euler3d_brick.h, b_log)
                fld[0][lp] = lean_log(a0);
                fld[4][lp] = lean_log(a4);
            } else {
                fld[0][lp] = w_log(a0);
                fld[4][lp] = w_log(a4);
            }
            fld[1][lp] = a1;
            fld[2][lp] = a2;
            fld[3][lp] = a3;
        }
    }
    __syncthreads();

    extrap_faces<N, T, false, G3, PACK>(P, fld, blockIdx.x * EPB, PACK ? P.count : P.nelem, PACK ? WX_REGION_BOUNDARY : WX_REGION_ALL, P.itf,
                                        P.send_s, P.send_n, P.send_w, P.send_e);
}

template <int N, typename T>
__global__ __launch_bounds__(Cfg<N>::BS, kK1Waves) void euler_extrap_kernel(const EulerParams<T> P) {
    euler_extrap_body<N, T, grid3_for<N>()>(P);
}

// the pack kernel of the one-kernel form: the ring's outward faces -> rotated / flipped edge messages (slots, not the grid form)
template <int N, typename T>
__global__ __launch_bounds__(Cfg<N>::BS, kK1Waves) void euler_pack_kernel(const EulerParams<T> P) {
    euler_extrap_body<N, T, false, true>(P);
}

// K1 for the prepared complex-step JVP (wx_euler3d_jvp_tangent_extrap_pack): only the TANGENTS of the face states of
// (q, eps v) are wanted - the values are cached.  The same arithmetic as the dual-number instantiation above, term by
// term, on seven real planes (log rho and log rho*theta; their tangents t / q; the three momentum tangents) instead of
// five 16-byte ones: 32 KB of LDS instead of 46 (n = 8), no value parts carried for the momentum rows - 0.283 -> 0.235 ms per
// E7 panel (5.4 TB/s, the float64 K1's rate), bit-identical tangents (the prepared and unprepared products still agree to
// the last bit: tests/test_n8_kernels_gpu.py).
// FIX: the tangent is corrected in place before it is used (EulerParams: tc_*): w = t - cs0 r0 - cs1 r1 in kiops_long_b's order
// (csrc/krylov.hip), written back, its squared norm left as one partial sum per workgroup
template <int N, bool FIX>
__global__ __launch_bounds__(Cfg<N>::BS, kK1Waves) void euler_tan_extrap_kernel(const EulerParams<dual> P) {
    using C = Cfg<N>;
    constexpr int N2 = C::N2, N3 = C::N3, EPB = C::EPB, BS = C::BS;
    __shared__ double pl[7][EPB * C::LE];
    __shared__ double red[FIX ? BS / 64 : 1];
    const int tid = threadIdx.x;
    __builtin_assume(tid < (int)Cfg<N>::BS);   // (the launch bounds: lets one-element workgroups drop their `le < EPB` guards)
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;
    {
        const int le = tid / N3, pt = tid % N3;
        const Elem el = decode_blk<EPB, grid3_for<N>()>(P, blockIdx.x * EPB + le, P.nelem, WX_REGION_ALL);
        double nn = 0.0;
        if (le < EPB && el.valid) {
            const size_t o = (size_t)el.e * N3 + pt;
            const int lp = le * C::LE + C::lidx(pt / N2, (pt / N) % N, pt % N);
            const auto r = P.q_re, t = P.q_tan;
            const double e = P.jvp_eps;
            const double r0 = r[o], r4 = r[4 * fs + o];
            double tv[5] = {t[o], t[fs + o], t[2 * fs + o], t[3 * fs + o], t[4 * fs + o]};
            if constexpr (FIX) {
                const auto a0 = P.tc_r0, a1 = P.tc_r1;
                const double cs0 = P.tcs ? P.tch[0] * P.tcs[0] : P.tch[0];
                const double cs1 = a1 != nullptr ? (P.tcs ? P.tch[1] * P.tcs[1] : P.tch[1]) : 0.0;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    double w = tv[k];
                    w -= cs0 * a0[k * fs + o];
                    if (a1 != nullptr) w -= cs1 * a1[k * fs + o];
                    P.tc_out[k * fs + o] = w;
                    nn += w * w;
                    tv[k] = w;
                }
            }
            pl[0][lp] = w_log(r0);
            pl[1][lp] = (e * tv[0]) / r0;
            pl[2][lp] = e * tv[1];
            pl[3][lp] = e * tv[2];
            pl[4][lp] = e * tv[3];
            pl[5][lp] = w_log(r4);
            pl[6][lp] = (e * tv[4]) / r4;
        }
        if constexpr (FIX) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nn += __shfl_down(nn, off, 64);
            if ((tid & 63) == 0) red[tid >> 6] = nn;
        }
    }
    __syncthreads();
    if constexpr (FIX) {
        if (tid == 0) {
            double tsum = 0.0;
#pragma unroll
            for (int w = 0; w < BS / 64; ++w) tsum += red[w];
            P.tc_part[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tsum;
        }
    }
    for (int fi = tid; fi < EPB * 6 * N2; fi += BS) {
        const int le = fi / (6 * N2);
        const int r = fi % (6 * N2);
        int f = r / N2;
        const int fp = r % N2;
        if (N2 % 64 == 0 && BS % 64 == 0) f = __builtin_amdgcn_readfirstlane(f);
        const Elem el = decode_blk<EPB, grid3_for<N>()>(P, blockIdx.x * EPB + le, P.nelem, WX_REGION_ALL);
        if (!el.valid) continue;
        const int d = f >> 1, plus = f & 1;
        const int a = fp / N, b = fp % N;
        int base, stride;
        if (d == 0) { base = C::lidx(a, b, 0); stride = 1; }
        else if (d == 1) { base = C::lidx(a, 0, b); stride = C::NP; }
        else { base = C::lidx(0, a, b); stride = N * C::NP; }
        const auto w = plus ? P.K->ep : P.K->em;
        double s[7];
#pragma unroll
        for (int v = 0; v < 7; ++v) s[v] = 0.0;
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const double wm = w[m];
#pragma unroll
            for (int v = 0; v < 7; ++v) s[v] += wm * pl[v][le * C::LE + base + m * stride];
        }
        double tn[5];
        tn[0] = exp(s[0]) * s[1];
        tn[1] = s[2]; tn[2] = s[3]; tn[3] = s[4];
        tn[4] = exp(s[5]) * s[6];
        double* dt = P.ft + ((size_t)el.e * 6 + f) * 5 * N2 + fp;
#pragma unroll
        for (int v = 0; v < 5; ++v) dt[v * N2] = tn[v];
        int edge = -1, along = 0;
        double X = 0.0;
        if (d == 0 && ((plus && el.ei == H - 1) || (!plus && el.ei == 0))) {
            edge = plus ? E_E : E_W;
            along = el.ej;
            X = P.bwe[el.ej * N + b];
        } else if (d == 1 && ((plus && el.ej == H - 1) || (!plus && el.ej == 0))) {
            edge = plus ? E_N : E_S;
            along = el.ei;
            X = P.bsn[el.ei * N + b];
        }
        dual* sendp = edge == E_S ? P.send_s : (edge == E_N ? P.send_n : (edge == E_W ? P.send_w : P.send_e));
        if (edge >= 0 && sendp != nullptr) {
            rotate_contra<double>(P.K->rot[edge], X, tn[1], tn[2]);
            int al = along, bb = b;
            if (P.K->flip[edge]) { al = H - 1 - along; bb = N - 1 - b; }
            const size_t eo = ((size_t)el.ek * H + al) * N2 + a * N + bb;
            const size_t vs = (size_t)V * H * N2;
            double* out = reinterpret_cast<double*>(sendp) + eo;
#pragma unroll
            for (int v = 0; v < 5; ++v) out[v * vs] = tn[v];
        }
    }
}

// All tiles of a rank in one launch (blockIdx.y = tile): the static per-tile parameters come from a device table,
// the state is a slice of one stacked tensor.  For small tiles the evaluation is launch-bound.
template <typename T>
struct EulerBatchDyn {
    const T *q, *y, *z;
    T* rhs;
    size_t stride;  // elements of T between consecutive tiles' states
    int region, count, axpy;
    double ca, cb, cc, cd;
    // shifted state q + eps v (float64) or the dual state (q, eps v) formed on load from REAL arrays (dual):
    const double *q_re, *q_tan;  // stride_re doubles apart per tile
    double* out_tan;             // dual JVP output (real)
    size_t stride_re;
    double eps, scale;
    int jvp;
    const double* dscale;   // EulerParams::dscale (fgmres' device pass), nullable
    int pulls;              // one-kernel form: tile-edge states pulled from the other tiles of the launch (EulerParams::pull_tile)
};

template <typename T, bool G>
__device__ __forceinline__ void batch_state(EulerParams<T, G>& P, const EulerBatchDyn<T>& dyn) {
    const size_t off = (size_t)blockIdx.y * dyn.stride, offr = (size_t)blockIdx.y * dyn.stride_re;
    P.q = dyn.q ? dyn.q + off : (const T*)nullptr;
    P.q_re = dyn.q_re ? dyn.q_re + offr : (const double*)nullptr;
    P.q_tan = dyn.q_tan ? dyn.q_tan + offr : (const double*)nullptr;
    P.out_tan = dyn.out_tan ? dyn.out_tan + offr : (double*)nullptr;
    P.jvp = dyn.jvp; P.jvp_eps = dyn.eps; P.jvp_scale = dyn.scale;
    P.dscale = dyn.dscale;
}

// The parameters of tile blockIdx.y for a batched launch: the table entry goes to LDS (one 8-byte word per thread), one
// thread patches in the per-launch fields, and the body reads what it needs where it needs it.  (Round 2 copied the
// entry into registers - `EulerParams<T> P = table[blockIdx.y]` -: about 120 values live for the whole kernel, which no
// register file holds beside the kernel's own state.  The batched JVP kernels spilled 470-600 bytes per lane and took
// 2.6 x the batched RHS kernel at the reference's benchmark sizes.)
template <typename T, typename Patch>
__device__ __forceinline__ const EulerParams<T>& batch_params(EulerParams<T>& sP, const EulerParams<T>* table, Patch patch) {
    static_assert(sizeof(EulerParams<T>) % 8 == 0, "copied in 8-byte words");
    constexpr int W = sizeof(EulerParams<T>) / 8;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(table + blockIdx.y);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&sP);
    for (int i = threadIdx.x; i < W; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) patch(sP);
    __syncthreads();
    return sP;
}

template <int N, typename T>
__global__ __launch_bounds__(Cfg<N>::BS, kK1Waves) void euler_extrap_batch_kernel(const EulerParams<T>* table,
                                                                                  const EulerBatchDyn<T> dyn) {
    if constexpr (std::is_same<T, double>::value) {   // (float64: the register copy fits - 66 VGPRs, nothing spills - and is faster)
        // the block comes out of memory: typed so that every access through its pointers is a global one (wx_common.h: gp)
        EulerParams<T, true> P = *reinterpret_cast<const EulerParams<T, true>*>(table + blockIdx.y);
        batch_state<T>(P, dyn);
        euler_extrap_body<N, T>(P);
    } else {
        __shared__ EulerParams<T> sP;
        euler_extrap_body<N, T>(batch_params<T>(sP, table, [&](EulerParams<T>& P) { batch_state<T>(P, dyn); }));
    }
}

// ... and the pack kernel of the one-kernel form for all tiles (dyn.count = the ring's size)
template <int N, typename T>
__global__ __launch_bounds__(Cfg<N>::BS, kK1Waves) void euler_pack_batch_kernel(const EulerParams<T>* table,
                                                                                const EulerBatchDyn<T> dyn) {
    if constexpr (std::is_same<T, double>::value) {
        EulerParams<T, true> P = *reinterpret_cast<const EulerParams<T, true>*>(table + blockIdx.y);
        batch_state<T>(P, dyn);
        P.count = dyn.count;
        euler_extrap_body<N, T, false, true>(P);
    } else {
        __shared__ EulerParams<T> sP;
        euler_extrap_body<N, T, false, true>(batch_params<T>(sP, table, [&](EulerParams<T>& P) { batch_state<T>(P, dyn); P.count = dyn.count; }));
    }
}

}  // namespace wx
