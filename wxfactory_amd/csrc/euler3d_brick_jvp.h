// The one-kernel form (euler3d_brick.h) of the complex-step Jacobian-vector product (solvers/matvec.py:56-61) at low order:
// the dual state (q, eps v) formed on load, the face states of the brick extrapolated on chip as dual numbers (the logarithm's
// value part by the lean form, its tangent t / q), every Riemann problem of the brick solved once in dual arithmetic, only the
// tangents kept (euler3d_jvp.h: tangent-only planes, the same passes and store, KIOPS' fused store included).  Why here even
// more than for R(Q): the two-kernel form's interface buffer holds DUAL values - 240 B/point at n = 2 - written by one kernel
// and read twice by the next.
#pragma once

namespace wx {

// the seven dual results of face f of element slot le -> the JVP kernel's face images (tangents of F*; B*, log p as duals)
template <int N, int EPB>
struct BrickJvpFaceStore {
    double* frt;   // [EPB][6][5][N2]
    dual* frf;     // [EPB][6][2][N2]
    __device__ __forceinline__ void operator()(int le, int f, int fp, const dual* out, dual bq, dual lp) const {
        constexpr int N2 = N * N;
        double* t = frt + ((le * 6 + f) * 5) * N2 + fp;
#pragma unroll
        for (int c = 0; c < 5; ++c) t[c * N2] = out[c].im;
        dual* q = frf + ((le * 6 + f) * 2) * N2 + fp;
        q[0] = bq;
        q[N2] = lp;
    }
};

template <int N>
__device__ __forceinline__ void euler_jvp_brick_body(const EulerParams<dual>& P, const BrickBoxes& GB, const BrickBatchCtx& ctx) {
    using C = BrickCfg<N>;
    using T = dual;
    constexpr int N2 = C::N2, N3 = C::N3, EPB = C::EPB, BS = C::BS;
    // ten real planes: the brick's dual state for the extrapolation (five dual planes) first, then - the face stage done - the
    // tangent planes ft[0..5] and fx[0..2] of the passes (euler3d_jvp.h) in the same words
    __shared__ double pool[10 * EPB * C::LE];
    double(*ft)[EPB * C::LE] = reinterpret_cast<double(*)[EPB * C::LE]>(pool);
    double(*fx)[EPB * C::LE] = reinterpret_cast<double(*)[EPB * C::LE]>(pool + 6 * EPB * C::LE);
    T* img = reinterpret_cast<T*>(pool);
    __shared__ double frt[EPB * 6 * 5 * N2];
    __shared__ T frf[EPB * 6 * 2 * N2];
    __shared__ double sD[N * N], sHF[N * N], sCm[N], sCp[N];
#define WX_FRT(le_, f_, c_, fp_) frt[(((le_) * 6 + (f_)) * 5 + (c_)) * N2 + (fp_)]
#define WX_FRF(le_, f_, c_, fp_) frf[(((le_) * 6 + (f_)) * 2 + (c_)) * N2 + (fp_)]

    const int tid = threadIdx.x;
    __builtin_assume(tid < BS);
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;
    const int brick_id = xcd_slab_block(blockIdx.x, gridDim.x >> 3);
    const BrickAt bk = brick_at(GB, brick_id, V);
    if (!bk.any) return;   // (uniform over the workgroup: the launch is padded to a multiple of eight workgroups)
    for (int i = tid; i < N * N; i += BS) {
        sD[i] = P.K->D[i];
        sHF[i] = P.K->HF[i];
    }
    if (tid < N) {
        sCm[tid] = P.K->cm[tid];
        sCp[tid] = P.K->cp[tid];
    }

    const int le = tid / N3, pt = tid - le * N3;
    const BElem el = brick_elem<EPB>(bk, le, H);
    const bool active = el.valid;
    const int kl = pt / N2, jl = (pt / N) % N, il = pt % N;
    const int lb = (le < EPB ? le : 0) * C::LE;
    const int lpt = lb + C::lidx(kl, jl, il);
    const size_t o = (size_t)el.e * N3 + pt;

    // ---- the dual state -> registers and, in the form the extrapolation wants, -> LDS
    PointIn<T> S;
    k2_point_loads<T, false>(P, active, o, fs, S, o, fs);
    const T q0 = S.q0, q1 = S.q1, q2 = S.q2, q3 = S.q3, q4 = S.q4;
    const double sg = S.sg;
    const T lq4 = b_log(q4);
    if (le < EPB) {
        img[0 * EPB * C::LE + lpt] = b_log(q0);
        img[1 * EPB * C::LE + lpt] = q1;
        img[2 * EPB * C::LE + lpt] = q2;
        img[3 * EPB * C::LE + lpt] = q3;
        img[4 * EPB * C::LE + lpt] = lq4;
    }
    __syncthreads();

    // ---- face stage: every Riemann problem of the brick once, in dual arithmetic
    brick_face_stage<N, T>(P, ctx, bk, img, EPB * C::LE, brick_id & (BS / 64 - 1), BrickJvpFaceStore<N, EPB>{frt, frf});

    // ---- pointwise quantities
    const T rinv = 1.0 / q0;
    const T u1 = q1 * rinv, u2 = q2 * rinv, u3 = q3 * rinv;
    const T glog = kGamma * (lq4 + kLogRdOverP0);
    const T p = kP0 * w_exp(glog);
    const T lp = kLogP0 + glog;

    // ---- forcing (tangent)
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc4 = 0.0, accw = 0.0, hf = 0.0, gcoef = 0.0;
    if (active) {
        const JvpForcing F = jvp_forcing<false>(P, o, fs, sg, S.h00, S.h01, S.h02, S.h11, S.h12, S.h22, q0, u1, u2, u3, p, o, fs);
        acc1 = F.f1; acc2 = F.f2; accw = F.fw; gcoef = F.gcoef;
    }

#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const T ud = w_sel(d == 0, u1, w_sel(d == 1, u2, u3));
        const double hd0 = d == 0 ? S.h00 : (d == 1 ? S.h01 : S.h02);
        const double hd1 = d == 0 ? S.h01 : (d == 1 ? S.h11 : S.h12);
        const double hd2 = d == 0 ? S.h02 : (d == 1 ? S.h12 : S.h22);
        const T sgu = sg * ud;
        const double Bd = sg * hd2;
        __syncthreads();  // the face stage's (its reads of the state images, which these planes overwrite) / the previous direction's reads are done
        if (le < EPB) {
            ft[0][lpt] = (sgu * q0).im;
            ft[1][lpt] = (sgu * q1 + (sg * hd0) * p).im;
            ft[2][lpt] = (sgu * q2 + (sg * hd1) * p).im;
            ft[3][lpt] = (sgu * q4).im;
            ft[4][lpt] = (sgu * q3).im;
            fx[0][lpt] = Bd;
            if (d == 0) {   // (the three planes that do not change with the direction)
                fx[1][lpt] = lp.re;
                fx[2][lpt] = lp.im;
                ft[5][lpt] = sg * q0.im;
            }
        }
        __syncthreads();

        int base, stride, idx, fp;
        if (d == 0) { base = lb + C::lidx(kl, jl, 0); stride = 1; idx = il; fp = kl * N + jl; }
        else if (d == 1) { base = lb + C::lidx(kl, 0, il); stride = C::NP; idx = jl; fp = kl * N + il; }
        else { base = lb + C::lidx(0, jl, il); stride = N * C::NP; idx = kl; fp = jl * N + il; }
        double dm[N];
#pragma unroll
        for (int m = 0; m < N; ++m) dm[m] = sD[idx * N + m];
        const double cm = sCm[idx], cp = sCp[idx];
        const int lf = le < EPB ? le : 0;
#pragma unroll 1
        for (int c0 = 0; c0 < 5; c0 += kFieldBatch) {
#pragma unroll
            for (int cc = 0; cc < kFieldBatch; ++cc) {
                const int c = c0 + cc;
                if (c < 5) {
                    double a = cm * WX_FRT(lf, 2 * d, c, fp) + cp * WX_FRT(lf, 2 * d + 1, c, fp);
#pragma unroll
                    for (int m = 0; m < N; ++m) a += dm[m] * ft[c][base + m * stride];
                    if (c == 0) acc0 += a;
                    else if (c == 1) acc1 += a;
                    else if (c == 2) acc2 += a;
                    else if (c == 3) acc4 += a;
                    else accw += a;
                }
            }
        }
        // W^d = [A@D + A*@C] + p [B@D + B*@C] + p B [log p@D + log p^@C]  (rhs_dfr.py:113-136): tangent of the two products
        {
            double xs0 = 0.0, xs1 = 0.0, xs2 = 0.0;
#pragma unroll 1
            for (int w = 0; w < 3; ++w) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < N; ++m) acc += dm[m] * fx[w][base + m * stride];
                if (w == 0) xs0 = acc;
                else if (w == 1) xs1 = acc;
                else xs2 = acc;
            }
            const T a5 = cm * WX_FRF(lf, 2 * d, 0, fp) + cp * WX_FRF(lf, 2 * d + 1, 0, fp) + xs0;
            const T a6 = cm * WX_FRF(lf, 2 * d, 1, fp) + cp * WX_FRF(lf, 2 * d + 1, 1, fp) + T(xs1, xs2);
            accw += (a5 * p).im + (a6 * (p * Bd)).im;
        }
        if (d == 2) {
#pragma unroll
            for (int m = 0; m < N; ++m) hf += sHF[idx * N + m] * ft[5][base + m * stride];
        }
    }

    if (P.jz == nullptr) {
        if (active) {
            const double s = P.advection_only ? 0.0 : -P.jvp_scale / sg;
            accw += gcoef * hf;  // gravity: inv_dzdeta * g * 1/sqrtG * HF_k(sqrtG rho)
            P.out_tan[o] = s * acc0;
            P.out_tan[fs + o] = s * acc1;
            P.out_tan[2 * fs + o] = s * acc2;
            P.out_tan[3 * fs + o] = s * accw;
            P.out_tan[4 * fs + o] = s * acc4;
        }
    } else {   // (uniform over the launch)
        const double s = (!active || P.advection_only) ? 0.0 : -P.jvp_scale / sg;
        accw += gcoef * hf;
        jvp_store_axpy<BS>(P, active, o, fs, s * acc0, s * acc1, s * acc2, s * accw, s * acc4);
    }
#undef WX_FRT
#undef WX_FRF
}

template <int N>
__global__ __launch_bounds__(BrickCfg<N>::BS, BrickCfg<N>::WAVES) void euler_jvp_brick_kernel(const EulerParams<dual> P, const BrickBoxes GB) {
    euler_jvp_brick_body<N>(P, GB, BrickBatchCtx{nullptr, 0, 0, 0u});
}

template <int N>
__global__ __launch_bounds__(BrickCfg<N>::BS, BrickCfg<N>::WAVES) void euler_jvp_brick_batch_kernel(const EulerParams<dual>* table,
                                                                                                const EulerBatchDyn<dual> dyn,
                                                                                                const BrickBoxes GB) {
    __shared__ EulerParams<dual> sP;
    const EulerParams<dual>& P = batch_params<dual>(sP, table, [&](EulerParams<dual>& Q) {
        batch_state<dual>(Q, dyn);
        Q.region = dyn.region; Q.count = dyn.count;
    });
    const BrickBatchCtx ctx{dyn.pulls ? (const void*)table : nullptr, (long long)dyn.stride_re, (int)blockIdx.y,
                            brick_pack_pulls<dual>(table + blockIdx.y)};
    euler_jvp_brick_body<N>(P, GB, ctx);
}

}  // namespace wx
