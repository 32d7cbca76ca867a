// The fused kernel specialised for the complex-step Jacobian-vector product (solvers/matvec.py:56-61): dual state
// formed on load, tangent-only LDS planes; + the plan-time zero scan.
#pragma once

namespace wx {

// ------------------------------------------------------------------------------------------------
// K2-JVP: the fused phases 3-8 specialised for the complex-step Jacobian-vector product (wx_euler3d_jvp).
// Only the TANGENT of R is wanted, and the derivative contractions are linear, so of the eight fields the generic
// dual-number kernel stages through LDS as (value, tangent) pairs, six need their tangent only (the four flux
// rows, the advective rho*w flux, sqrtG*rho), one is a pure metric quantity with no tangent (B = sqrtG h^{d3})
// and one needs both (log p, multiplied by p B afterwards).  The same holds for the face quantities.  LDS per
// element 118 KB -> 70 KB and half the registers in the accumulators: TWO workgroups per CU instead of one,
// and 40 % fewer LDS bytes and contraction flops.  Arithmetic is the generic kernel's, term by term; the Riemann
// problems use the own / neighbour form (rusanov_own) and the Christoffel rows are read one at a time (rolled loop):
// with everything in flight the compiler wanted 184 VGPRs.  WXHIP_JVP_LEAN=0 (environment, read once) sends
// wx_euler3d_jvp through the generic kernel instead.
// ------------------------------------------------------------------------------------------------
// tangent of the forcing of the three momentum rows, times sqrtG (.f1, .f2, .fw), and gcoef = inv_dzdeta * g
struct JvpForcing { double f1, f2, fw, gcoef; };
template <bool CACHED>
__device__ __forceinline__ JvpForcing jvp_forcing(const EulerParams<dual>& P, size_t o, size_t fs, double sg, double h00,
                                                  double h01, double h02, double h11, double h12, double h22, dual q0,
                                                  dual u1, dual u2, dual u3, dual p, size_t om, size_t fsm) {
    JvpForcing r{0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
        const double* c = P.chr + (size_t)(i * 9) * fsm + om;
        double c01 = 0.0, c02 = 0.0, c03 = 0.0;
        if (!P.rot_zero) { c01 = ldm_if<CACHED>(c); c02 = ldm_if<CACHED>(c + fsm); c03 = ldm_if<CACHED>(c + 2 * fsm); }
        const double c11 = ldm_if<CACHED>(c + 3 * fsm), c12 = ldm_if<CACHED>(c + 4 * fsm), c13 = ldm_if<CACHED>(c + 5 * fsm),
                     c22 = ldm_if<CACHED>(c + 6 * fsm), c23 = ldm_if<CACHED>(c + 7 * fsm), c33 = ldm_if<CACHED>(c + 8 * fsm);
        dual f = 2.0 * q0 * (c01 * u1 + c02 * u2 + c03 * u3) + c11 * (q0 * u1 * u1 + h00 * p) +
                 2.0 * c12 * (q0 * u1 * u2 + h01 * p) + 2.0 * c13 * (q0 * u1 * u3 + h02 * p) +
                 c22 * (q0 * u2 * u2 + h11 * p) + 2.0 * c23 * (q0 * u2 * u3 + h12 * p) +
                 c33 * (q0 * u3 * u3 + h22 * p);
        if (P.has_damp) f += (P.dcoef[o] * q0) * ((i == 0 ? u1 : (i == 1 ? u2 : u3)) - P.duref[(size_t)i * fs + o]);
        if (i == 0) r.f1 = sg * f.im;
        else if (i == 1) r.f2 = sg * f.im;
        else r.fw = sg * f.im;
    }
    r.gcoef = ldm_if<CACHED>(P.idz + om) * kGravity;
    return r;
}

// The product's store when P.jz is set (t = jvp_scale * tangent, what the plain store writes): KIOPS' next Krylov vector
// out = *jzs * t + *jza * z  (solvers/kiops.py:170-176) in the same store, and with P.jr0 that vector's products with one or two
// basis rows on top, summed over the workgroup (wave shuffles, then one value per wave through LDS): the first streaming stage
// of the long-vector build (csrc/krylov.hip: kiops_long_a) has then nothing left to read.
template <int BS>
__device__ __forceinline__ void jvp_store_axpy(const EulerParams<dual>& P, bool active, size_t o, size_t fs, double t0, double t1,
                                          double t2, double t3, double t4) {
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0, w4 = 0.0;
    if (active) {
        const double sa = P.jzs ? *P.jzs : 1.0, zc = *P.jza;
        w0 = sa * t0 + zc * P.jz[o];
        w1 = sa * t1 + zc * P.jz[fs + o];
        w2 = sa * t2 + zc * P.jz[2 * fs + o];
        w3 = sa * t3 + zc * P.jz[3 * fs + o];
        w4 = sa * t4 + zc * P.jz[4 * fs + o];
        P.out_tan[o] = w0;
        P.out_tan[fs + o] = w1;
        P.out_tan[2 * fs + o] = w2;
        P.out_tan[3 * fs + o] = w3;
        P.out_tan[4 * fs + o] = w4;
    }
    if (P.jr0 == nullptr) return;   // (uniform over the launch: every thread of the workgroup reaches the barriers below)
    __shared__ double jred[2 * (BS / 64)];
    double d0 = 0.0, d1 = 0.0;
    if (active) {
        d0 = P.jr0[o] * w0 + P.jr0[fs + o] * w1 + P.jr0[2 * fs + o] * w2 + P.jr0[3 * fs + o] * w3 + P.jr0[4 * fs + o] * w4;
        if (P.jr1 != nullptr)
            d1 = P.jr1[o] * w0 + P.jr1[fs + o] * w1 + P.jr1[2 * fs + o] * w2 + P.jr1[3 * fs + o] * w3 + P.jr1[4 * fs + o] * w4;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        d0 += __shfl_down(d0, off, 64);
        d1 += __shfl_down(d1, off, 64);
    }
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) {
        jred[tid >> 6] = d0;
        jred[BS / 64 + (tid >> 6)] = d1;
    }
    __syncthreads();
    if (tid == 0) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int w = 0; w < BS / 64; ++w) {
            s0 += jred[w];
            s1 += jred[BS / 64 + w];
        }
        const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        P.jpart[2 * wg] = s0;
        P.jpart[2 * wg + 1] = s1;
    }
}

template <int N, bool COLM = false, bool G3 = false>
__device__ __forceinline__ void euler_jvp_body(const EulerParams<dual>& P) {
    using C = Cfg<N>;
    using T = dual;
    const int bx = COLM ? xcd_slab_block(blockIdx.x, gridDim.x >> 3) : (int)blockIdx.x;
    constexpr int N2 = C::N2, N3 = C::N3, EPB = C::EPB, BS = C::BS;
    __shared__ double ft[6][EPB * C::LE];     // tangents: F rows rho, rho u1, rho u2, rho theta; A; sqrtG*rho
    __shared__ double fx[3][EPB * C::LE];     // B = sqrtG h^{d3} (metric only); log p, value and tangent planes
    __shared__ double frt[EPB][6][5][N2];     // tangents of the face quantities 0..4 of rusanov_face
    __shared__ T frf[EPB][6][2][N2];          // B*_own, log p_own
    __shared__ double sD[N * N], sHF[N * N], sCm[N], sCp[N];

    const int tid = threadIdx.x;
    __builtin_assume(tid < (int)Cfg<N>::BS);   // (the launch bounds: lets one-element workgroups drop their `le < EPB` guards)
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;
    for (int i = tid; i < N * N; i += BS) {
        sD[i] = P.K->D[i];
        sHF[i] = P.K->HF[i];
    }
    if (tid < N) {
        sCm[tid] = P.K->cm[tid];
        sCp[tid] = P.K->cp[tid];
    }

    const int le = tid / N3, pt = tid % N3;
    const Elem el = COLM ? decode_elem_col(bx * EPB + le, P.count, P.region, H, V, P.md_v, P.md_h, P.md_w) : decode_blk<EPB, G3>(P, bx * EPB + le, P.count, P.region);
    const bool active = (le < EPB) && el.valid;
    const int kl = pt / N2, jl = (pt / N) % N, il = pt % N;
    const int lb = (le < EPB ? le : 0) * C::LE;
    const int lpt = lb + C::lidx(kl, jl, il);
    const size_t o = (size_t)el.e * N3 + pt;

    // ---- face stage
    for (int fi = tid; fi < EPB * 6 * N2; fi += BS) {
        const int fle = fi / (6 * N2);
        const int r = fi % (6 * N2);
        const int f = r / N2, fp = r % N2;
        const Elem fel = COLM ? decode_elem_col(bx * EPB + fle, P.count, P.region, H, V, P.md_v, P.md_h, P.md_w) : decode_blk<EPB, G3>(P, bx * EPB + fle, P.count, P.region);
        if (!fel.valid) continue;
        T out[7];
        face_problem<N, T, true, COLM>(P, fel, f, fp, out);
#pragma unroll
        for (int c = 0; c < 5; ++c) frt[fle][f][c][fp] = out[c].im;
        frf[fle][f][0][fp] = out[5];
        frf[fle][f][1][fp] = out[6];
    }

    PointIn<T> S;
    const size_t om = COLM ? ((size_t)el.ej * H + el.ei) * N2 + pt % N2 : o;
    const size_t fsm = COLM ? (size_t)H * H * N2 : fs;
    k2_point_loads<T, COLM>(P, active, o, fs, S, om, fsm);
    const T q0 = S.q0, q1 = S.q1, q2 = S.q2, q3 = S.q3, q4 = S.q4;
    const double sg = S.sg;
    // ---- pointwise quantities
    const T rinv = 1.0 / q0;
    const T u1 = q1 * rinv, u2 = q2 * rinv, u3 = q3 * rinv;
    const T glog = kGamma * w_log(kRdOverP0 * q4);
    const T p = kP0 * w_exp(glog);
    if (le < EPB) {
        const T lp = kLogP0 + glog;
        fx[1][lpt] = lp.re;
        fx[2][lpt] = lp.im;
        ft[5][lpt] = sg * q0.im;
    }

    // ---- forcing (tangent)
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc4 = 0.0, accw = 0.0, hf = 0.0, gcoef = 0.0;
    if (active) {
        const JvpForcing F = jvp_forcing<COLM>(P, o, fs, sg, S.h00, S.h01, S.h02, S.h11, S.h12, S.h22, q0, u1, u2, u3, p, om, fsm);
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
        __syncthreads();  // face stage / previous direction's reads are done
        if (le < EPB) {
            ft[0][lpt] = (sgu * q0).im;
            ft[1][lpt] = (sgu * q1 + (sg * hd0) * p).im;
            ft[2][lpt] = (sgu * q2 + (sg * hd1) * p).im;
            ft[3][lpt] = (sgu * q4).im;
            ft[4][lpt] = (sgu * q3).im;
            fx[0][lpt] = Bd;
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
        constexpr int FB = kFieldBatch;
#pragma unroll 1
        for (int c0 = 0; c0 < 5; c0 += FB) {
#pragma unroll
            for (int cc = 0; cc < FB; ++cc) {
                const int c = c0 + cc;
                if (c < 5) {
                    double a = cm * frt[lf][2 * d][c][fp] + cp * frt[lf][2 * d + 1][c][fp];
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
        // W^d = [A@D + A*@C] + p [B@D + B*@C] + p B [log p@D + log p^@C]  (rhs_dfr.py:113-136): tangent of the
        // two products; B@D is a metric-only number.  One plane at a time: bounds the LDS reads in flight
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
            const T a5 = cm * frf[lf][2 * d][0][fp] + cp * frf[lf][2 * d + 1][0][fp] + xs0;
            const T a6 = cm * frf[lf][2 * d][1][fp] + cp * frf[lf][2 * d + 1][1][fp] + T(xs1, xs2);
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
        jvp_store_axpy<Cfg<N>::BS>(P, active, o, fs, s * acc0, s * acc1, s * acc2, s * accw, s * acc4);
    }
}

// The same kernel with the contractions on the matrix cores (n = 8; mf4_dir_pass, the layout and the in-place scheme of
// the fused RHS kernel).  Nine real planes per direction: the five flux tangents, B (metric), log p value and tangent -
// eight take D | cm | cp with their face pairs as the third k-step - and the tangent of sqrtG rho for the vertical
// high-filter; the tangent of B* has no nodal part and keeps its two-term correction on the vector pipe.
constexpr int kJvFS = 9 * 64 + 16;   // doubles per face of the JVP kernel's face image (9 quantities)

template <bool COLM = false, bool G3 = false>
__device__ __forceinline__ void euler_jvp_body_mf(const EulerParams<dual>& P) {
    using T = dual;
    const int bx = COLM ? xcd_slab_block(blockIdx.x, gridDim.x >> 3) : (int)blockIdx.x;
    constexpr int N = 8, N2 = 64, N3 = 512;
    __shared__ double pl[9 * kMfLE];   // 0-4 flux tangents (rho, rho u1, rho u2, rho theta, A); 5 B; 6, 7 log p (value, tangent); 8 (sqrtG rho)'
    __shared__ double fq[6 * kJvFS];   // per face: 0-4 tangents of F*; 5 B*.re; 6, 7 log p_own (value, tangent); 8 B*.im
    __shared__ double sCm[N], sCp[N];
    const int tid = threadIdx.x;
    __builtin_assume(tid < (int)Cfg<N>::BS);   // (the launch bounds: lets one-element workgroups drop their `le < EPB` guards)
    const int H = P.H, V = P.V;
    const size_t fs = (size_t)P.nelem * N3;
    if (tid < N) {
        sCm[tid] = P.K->cm[tid];
        sCp[tid] = P.K->cp[tid];
    }
    const Elem el = COLM ? decode_elem_col(bx, P.count, P.region, H, V, P.md_v, P.md_h, P.md_w) : decode_blk<1, G3>(P, bx, P.count, P.region);
    const bool active = el.valid;
    const int kl = tid / N2, jl = (tid / N) % N, il = tid % N;
    const int lptm = mf_idx(kl, jl, il);
    const size_t o = (size_t)el.e * N3 + tid;

    // ---- face stage
    for (int fi = tid; fi < 6 * N2; fi += 512) {
        const int f = fi / N2, fp = fi % N2;
        if (!el.valid) continue;
        T out[7];
        face_problem<N, T, true, COLM>(P, el, f, fp, out);
        double* q = fq + f * kJvFS + fp;
#pragma unroll
        for (int c = 0; c < 5; ++c) q[c * N2] = out[c].im;
        q[5 * N2] = out[5].re; q[8 * N2] = out[5].im;
        q[6 * N2] = out[6].re; q[7 * N2] = out[6].im;
    }

    // the operator fragments of the matrix-core passes: loaded behind the face stage, the kernel's register peak (euler_rhs_body)
    const MfOps4 mops = mf4_load_ops(P.K->D, P.K->cm, P.K->cp, P.K->HF, tid & 63);

    PointIn<T> S;
    // (column form: a 32-bit offset into the slabs, which are small - one register instead of two beside `o`)
    const unsigned om32 = (unsigned)((el.ej * H + el.ei) * N2 + tid % N2);
    const size_t om = COLM ? (size_t)om32 : o;
    const size_t fsm = COLM ? (size_t)H * H * N2 : fs;
    k2_point_loads<T, COLM>(P, active, o, fs, S, om, fsm);
    const T q0 = S.q0, q1 = S.q1, q2 = S.q2, q3 = S.q3, q4 = S.q4;
    const double sg = S.sg;
    const T rinv = 1.0 / q0;
    const T u1 = q1 * rinv, u2 = q2 * rinv, u3 = q3 * rinv;
    const T glog = kGamma * w_log(kRdOverP0 * q4);
    const T p = kP0 * w_exp(glog);
    const T lp = kLogP0 + glog;

    // ---- forcing (tangent)
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc4 = 0.0, accw = 0.0, hf = 0.0, gcoef = 0.0;
    if (active) {
        const JvpForcing F = jvp_forcing<COLM>(P, o, fs, sg, S.h00, S.h01, S.h02, S.h11, S.h12, S.h22, q0, u1, u2, u3, p, om, fsm);
        acc1 = F.f1; acc2 = F.f2; accw = F.fw; gcoef = F.gcoef;
    }

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (kSkelDirs) {   // diagnostic builds: the staged data consumed, no passes
        pl[lptm] = lp.im;
        __syncthreads();
        acc0 += fq[tid & 63] + pl[lptm] + u1.im + u2.im + u3.im;
    }
#pragma unroll
    for (int d = 0; d < (kSkelDirs ? 0 : 3); ++d) {
        const T ud = w_sel(d == 0, u1, w_sel(d == 1, u2, u3));
        const double hd0 = d == 0 ? S.h00 : (d == 1 ? S.h01 : S.h02);
        const double hd1 = d == 0 ? S.h01 : (d == 1 ? S.h11 : S.h12);
        const double hd2 = d == 0 ? S.h02 : (d == 1 ? S.h12 : S.h22);
        const T sgu = sg * ud;
        const double Bd = sg * hd2;
        // each thread stages its own node of the nine planes, the eight waves contract all lines in place, each thread
        // picks its own node up again (no barrier between a thread's read and its next write)
        pl[0 * kMfLE + lptm] = (sgu * q0).im;
        pl[1 * kMfLE + lptm] = (sgu * q1 + (sg * hd0) * p).im;
        pl[2 * kMfLE + lptm] = (sgu * q2 + (sg * hd1) * p).im;
        pl[3 * kMfLE + lptm] = (sgu * q4).im;
        pl[4 * kMfLE + lptm] = (sgu * q3).im;
        pl[5 * kMfLE + lptm] = Bd;
        pl[6 * kMfLE + lptm] = lp.re;
        pl[7 * kMfLE + lptm] = lp.im;
        if (d == 2) pl[8 * kMfLE + lptm] = sg * q0.im;
        __syncthreads();
        if (d == 0) mf4_dir_pass<0, true, 8, true, kJvFS, kJvpMfFieldBatch>(pl, fq, mops, wave, tid & 63);
        else if (d == 1) mf4_dir_pass<1, true, 8, true, kJvFS, kJvpMfFieldBatch>(pl, fq, mops, wave, tid & 63);
        else mf4_dir_pass<2, true, 8, true, kJvFS, kJvpMfFieldBatch>(pl, fq, mops, wave, tid & 63);
        __syncthreads();
        acc0 += pl[0 * kMfLE + lptm];
        acc1 += pl[1 * kMfLE + lptm];
        acc2 += pl[2 * kMfLE + lptm];
        acc4 += pl[3 * kMfLE + lptm];
        accw += pl[4 * kMfLE + lptm];
        // W^d = [A@D + A*@C] + p [B@D + B*@C] + p B [log p@D + log p^@C]  (rhs_dfr.py:113-136): tangent of the two products
        const int fp = d == 0 ? kl * N + jl : (d == 1 ? kl * N + il : jl * N + il);
        const int ix = d == 0 ? il : (d == 1 ? jl : kl);
        const T a5(pl[5 * kMfLE + lptm], sCm[ix] * fq[(2 * d) * kJvFS + 8 * N2 + fp] + sCp[ix] * fq[(2 * d + 1) * kJvFS + 8 * N2 + fp]);
        const T a6(pl[6 * kMfLE + lptm], pl[7 * kMfLE + lptm]);
        accw += (a5 * p).im + (a6 * (p * Bd)).im;
        if (d == 2) hf = pl[8 * kMfLE + lptm];
    }

    if (P.jz == nullptr) {
        if (active) {
            const double sc = P.advection_only ? 0.0 : -P.jvp_scale / sg;
            accw += gcoef * hf;  // gravity: inv_dzdeta * g * 1/sqrtG * HF_k(sqrtG rho)
            P.out_tan[o] = sc * acc0;
            P.out_tan[fs + o] = sc * acc1;
            P.out_tan[2 * fs + o] = sc * acc2;
            P.out_tan[3 * fs + o] = sc * accw;
            P.out_tan[4 * fs + o] = sc * acc4;
        }
    } else {   // (uniform over the launch)
        const double sc = (!active || P.advection_only) ? 0.0 : -P.jvp_scale / sg;
        accw += gcoef * hf;
        jvp_store_axpy<Cfg<N>::BS>(P, active, o, fs, sc * acc0, sc * acc1, sc * acc2, sc * accw, sc * acc4);
    }
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::BS, kJvpWaves) void euler_jvp_kernel(const EulerParams<dual> P) {
    if constexpr (N == 8 && WX_MFMA) euler_jvp_body_mf<false, true>(P);
    else euler_jvp_body<N, false, grid3_for<N>()>(P);
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::BS, kJvpWaves) void euler_jvp_batch_kernel(const EulerParams<dual>* table,
                                                                                const EulerBatchDyn<dual> dyn) {
    __shared__ EulerParams<dual> sP;
    const EulerParams<dual>& P = batch_params<dual>(sP, table, [&](EulerParams<dual>& Q) {
        batch_state<dual>(Q, dyn);
        Q.region = dyn.region; Q.count = dyn.count;
    });
    if constexpr (N == 8 && WX_MFMA) euler_jvp_body_mf<false>(P);
    else euler_jvp_body<N, false>(P);
}

// plan-time scan of a static field: raises *flag when any value differs from (+/-) zero
__global__ __launch_bounds__(256) void any_nonzero_kernel(const double* __restrict__ x, size_t count, int* flag) {
    bool any = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        any = any || (x[i] != 0.0);
    if (any) *flag = 1;   // many writers, one value: benign
}

}  // namespace wx
