// Matrix-core (MFMA) form of the 1-D operator passes of the element kernels, n = 8, float64 (gfx950).
// Shared by the fused RHS / JVP kernels (euler3d.hip) and the exponential filter (filters.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace wx {

// ------------------------------------------------------------------------------------------------
// Derivative contractions on the matrix cores (n = 8, float64).
//
// One directional pass of the fused kernel is  out[i'] = sum_m D[i'][m] F[m] + cm[i'] F*_minus + cp[i'] F*_plus  on
// every line of 8 nodes of every staged field: an 8 x 10 operator applied to 64 lines x 7 fields.  On
// v_mfma_f64_16x16x4_f64 (A 16 x 4, B 4 x 16, one f64 per lane each; C/D 16 x 16, 4 f64 per lane) one "job" is
//   A = the operator, rows 0..7 (rows 8..15 are zero: the tile's unavoidable padding for an 8-row operator),
//   B = 16 lines of one field, 3 k-steps: nodes {2a} , nodes {2a+1}, the two common face values (slots 2, 3 zero),
// and leaves in lane (a = lane >> 4, c = lane & 15), registers 0 and 1, the results at nodes 2a and 2a + 1 of line c:
// the row order of A and the k order of A and B are permuted (row m <-> node 2 (m & 3) + (m >> 2)) so that a lane
// reads and writes two NEIGHBOURING nodes.  Every nodal value is read from LDS once per direction (eight times on
// the vector pipe) and the 70 f64 FMAs per point and direction of the contraction leave the VALU.
// Results go back IN PLACE (a job reads its 16 lines before it writes them; jobs touch disjoint lines), the point
// threads then pick up their own node.  LDS image: node (kl, jl, il) of a field at kl*72 + jl*8 + (il ^ jl):
// conflict-free for all six access patterns (point threads: plane of 64; jobs: 2 nodes x 16 lines reads,
// 1 node x 16 lines writes, along i, j and k), found by exhaustive search over paddings and XOR swizzles.
// Lines of a job: u = c & 7, v = 2 t + (c >> 3)  with (jl, kl) / (il, kl) / (il, jl) = (u, v) for d = 0 / 1 / 2,
// t = wave & 3; fields c0 = wave >> 2, c0 + 2, c0 + 4, c0 + 6.
// ------------------------------------------------------------------------------------------------
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
constexpr int kMfLE = 8 * 72;        // doubles per field image
constexpr int kMfFS = 7 * 64 + 16;   // doubles per face in the face-flux image: faces 2d and 2d+1 land on different banks
__device__ __forceinline__ int mf_idx(int kl, int jl, int il) { return kl * 72 + jl * 8 + (il ^ jl); }
// permuted row / k-slot order of the operator tile
__device__ __forceinline__ int mf_node(int m) { return ((m & 3) << 1) | (m >> 2); }

// The A operands of this lane: D|cm|cp (3 k-steps) and HF (2 k-steps), from the plan's constants.
struct MfOps { double a0, a1, a2, h0, h1; };
// Dm, HFm: 8 x 8 row-major operators; cm, cp: the two correction columns (8 each)
__device__ __forceinline__ MfOps mf_load_ops(const double* Dm, const double* cm, const double* cp, const double* HFm, int lane) {
    const int a = lane >> 4, c = lane & 15;
    MfOps o{0.0, 0.0, 0.0, 0.0, 0.0};
    if (c < 8) {
        const int row = mf_node(c);
        o.a0 = Dm[row * 8 + 2 * a]; o.a1 = Dm[row * 8 + 2 * a + 1];
        o.a2 = a == 0 ? cm[row] : (a == 1 ? cp[row] : 0.0);
        o.h0 = HFm[row * 8 + 2 * a]; o.h1 = HFm[row * 8 + 2 * a + 1];
    }
    return o;
}

// One directional pass over the staged fields of one element (one workgroup of 8 waves).  fld: field images
// (kMfLE apart), frs: face-flux image [6 faces][kMfFS] with quantity c of face point fp at c*64 + fp.
// NFLD = 7 (d < 2) or 8 (d = 2: field 7 = sqrtG rho takes the vertical high-filter HF instead of D, no faces).
template <int D, bool CORR>
__device__ __forceinline__ void mf_dir_pass(double* fld, const double* frs, const MfOps& op, int wave, int lane) {
    const int a = lane >> 4, c = lane & 15, u = c & 7;
    const int v = 2 * (wave & 3) + (c >> 3);
    const int m0 = 2 * a, m1 = 2 * a + 1;
    int i0, i1;
    if (D == 0) { i0 = mf_idx(v, u, m0); i1 = mf_idx(v, u, m1); }
    else if (D == 1) { i0 = mf_idx(v, m0, u); i1 = mf_idx(v, m1, u); }
    else { i0 = mf_idx(m0, v, u); i1 = mf_idx(m1, v, u); }
    const int fo = (2 * D + (a & 1)) * kMfFS + 16 * (wave & 3) + c;   // face point of line (u, v) = 8 v + u
    const int c0 = wave >> 2;
    constexpr int NJ = 4;
    double b0[NJ], b1[NJ], b2[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int f = c0 + 2 * j;
        if (f < 7 || D == 2) {
            b0[j] = fld[f * kMfLE + i0];
            b1[j] = fld[f * kMfLE + i1];
            if (CORR && f < 7) b2[j] = frs[fo + f * 64];
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int f = c0 + 2 * j;
        if (f < 7) {
            mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(op.a0, b0[j], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(op.a1, b1[j], acc, 0, 0, 0);
            if (CORR) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(op.a2, b2[j], acc, 0, 0, 0);
            fld[f * kMfLE + i0] = acc[0];
            fld[f * kMfLE + i1] = acc[1];
        } else if (D == 2) {
            mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(op.h0, b0[j], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(op.h1, b1[j], acc, 0, 0, 0);
            fld[f * kMfLE + i0] = acc[0];
            fld[f * kMfLE + i1] = acc[1];
        }
    }
}

// The same pass on v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 products per instruction, 16 cycles each, no
// padding - the 8 x 10 operator is cut into 4 x 4 blocks (output half I, input quarter M; the face pair is a third
// k-step with two zero columns), so a field's 64 lines cost 24 instructions x 16 cycles instead of 12 x 64.
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
