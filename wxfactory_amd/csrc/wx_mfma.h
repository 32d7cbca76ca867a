// Matrix-core (MFMA) form of the 1-D operator passes of the element kernels, n = 8, float64 (gfx950).
// Shared by the fused RHS / JVP kernels (euler3d.hip) and the exponential filter (filters.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace wx {

// ------------------------------------------------------------------------------------------------
// Derivative contractions on the matrix cores (n = 8, float64).
//
// One directional pass of the fused kernel is  out[i'] = sum_m D[i'][m] F[m] + cm[i'] F*_minus + cp[i'] F*_plus  on
// every line of 8 nodes of every staged field: an 8 x 10 operator applied to 64 lines x 7 fields.  It runs on
// v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 products per instruction, 16 cycles each, no padding - the
// operator is cut into 4 x 4 blocks (output half I, input quarter M; the face pair is a third k-step with two zero
// columns), so a field's 64 lines cost 24 instructions x 16 cycles.  (Round 2 built the pass on
// v_mfma_f64_16x16x4_f64 first: half of each 16 x 16 tile is padding for an 8-row operator, 12 instructions x 64 cycles
// per field; A/B and counters in DESIGN.md 4.1 and profiles/r02_k2_mfma_*.)  Every nodal value is read from LDS once per
// direction (eight times on the vector pipe) and the 70 f64 FMAs per point and direction leave the VALU.
// Results go back IN PLACE (a wave reads its lines before it writes them; waves touch disjoint lines), the point
// threads then pick up their own node.  LDS image: node (kl, jl, il) of a field at kl*72 + jl*8 + (il ^ jl).
// (SQ_LDS_BANK_CONFLICT reads 1.17 cycles per LDS instruction on this kernel.  That is the counter's floor for 64 lanes x
// 8 bytes, not a conflict of the image: a build without the passes - plain stores of 64 consecutive doubles only -
// reads 1.33, and an image re-ordered so that each half-wave of the result writes lands on 32 different bank pairs
// changed neither the count, to the last digit, nor the time: profiles/r03_lds_bank_conflict_ab.txt.)
// ------------------------------------------------------------------------------------------------
#ifndef WX_MFMA
#define WX_MFMA 1   // 0: the vector-pipe contractions for n = 8 too (A/B builds)
#endif
constexpr int kMfLE = 8 * 72;        // doubles per field image
#ifndef WX_MF_FS_PAD
#define WX_MF_FS_PAD 16
#endif
constexpr int kMfFS = 7 * 64 + WX_MF_FS_PAD;   // doubles per face in the face-flux image: faces 2d and 2d+1 land on different banks
__device__ __forceinline__ int mf_idx(int kl, int jl, int il) { return kl * 72 + jl * 8 + (il ^ jl); }

// Lane maps (found with tools/mfma_f64_4x4_probe.hip): lane l, k = l >> 4, block g = (l >> 2) & 3, x = l & 3:
//   A_g[i = x][k],  B_g[k][j = x],  result D_g[i][j] in lane 16 i + 4 g + j.
// Block g = I + 2 L works on output half I (nodes 4 I .. 4 I + 3) of the four lines u = 4 L .. 4 L + 3 of one
// octet of lines (u = 0..7, v fixed); wave w owns octet v = w of every field.  The lane ends up with node
// 4 I + (l >> 4) of line u = 4 L + (l & 3).
struct MfOps4 { double a0, a1, af, h0, h1; };
// Dm, HFm: 8 x 8 row-major operators; cm, cp: the two correction columns (8 each; null: no face step)
__device__ __forceinline__ MfOps4 mf4_load_ops(const double* Dm, const double* cm, const double* cp, const double* HFm, int lane) {
    const int k = lane >> 4, I = (lane >> 2) & 1, x = lane & 3, row = 4 * I + x;
    MfOps4 o;
    o.a0 = Dm[row * 8 + k]; o.a1 = Dm[row * 8 + 4 + k];
    o.af = cm == nullptr ? 0.0 : (k == 0 ? cm[row] : (k == 1 ? cp[row] : 0.0));
    o.h0 = HFm == nullptr ? 0.0 : HFm[row * 8 + k]; o.h1 = HFm == nullptr ? 0.0 : HFm[row * 8 + 4 + k];
    return o;
}

// NDF fields carry the operator D | cm | cp (image f, face quantity f, f < NDF); with HFLD the image NDF takes HF (no
// faces) when D == 2.  FS: doubles per face in the face image.  FB: fields whose operands are in flight together.
template <int D, bool CORR, int NDF = 7, bool HFLD = true, int FS = kMfFS, int FB = 8>
__device__ __forceinline__ void mf4_dir_pass(double* fld, const double* frs, const MfOps4& op, int wave, int lane) {
    const int k = lane >> 4, g = (lane >> 2) & 3, I = g & 1, u = 4 * (g >> 1) + (lane & 3), v = wave;
    int r0, r1, wo;   // operand nodes k, 4 + k of line (u, v); result node 4 I + k
    if (D == 0) { r0 = mf_idx(v, u, k); r1 = mf_idx(v, u, 4 + k); wo = mf_idx(v, u, 4 * I + k); }
    else if (D == 1) { r0 = mf_idx(v, k, u); r1 = mf_idx(v, 4 + k, u); wo = mf_idx(v, 4 * I + k, u); }
    else { r0 = mf_idx(k, v, u); r1 = mf_idx(4 + k, v, u); wo = mf_idx(4 * I + k, v, u); }
    const int fo = (2 * D + (k & 1)) * FS + 8 * v + u;
    constexpr int NFLD = NDF + ((D == 2 && HFLD) ? 1 : 0);
#pragma unroll
    for (int f0 = 0; f0 < NFLD; f0 += FB) {
        double b0[FB], b1[FB], bf[FB], acc[FB];
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            const int f = f0 + i;
            if (f < NFLD) {
                b0[i] = fld[f * kMfLE + r0];
                b1[i] = fld[f * kMfLE + r1];
                if (CORR && f < NDF) bf[i] = frs[fo + f * 64];
            }
        }
#pragma unroll
        for (int i = 0; i < FB; ++i)
            if (f0 + i < NFLD) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(f0 + i < NDF ? op.a0 : op.h0, b0[i], 0.0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < FB; ++i)
            if (f0 + i < NFLD) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(f0 + i < NDF ? op.a1 : op.h1, b1[i], acc[i], 0, 0, 0);
        if (CORR) {
#pragma unroll
            for (int i = 0; i < FB; ++i)
                if (f0 + i < NDF) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(op.af, bf[i], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < FB; ++i)
            if (f0 + i < NFLD) fld[(f0 + i) * kMfLE + wo] = acc[i];
    }
}


}  // namespace wx
