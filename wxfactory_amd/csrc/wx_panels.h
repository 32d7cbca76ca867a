// Cubed-sphere edge tables shared by the pack kernels (one tile per panel).
#pragma once

namespace wx {

enum { E_S = 0, E_N = 1, E_W = 2, E_E = 3 };

// flip / rotation tables, reference process_topology.py:126-175 (one tile per panel).
// rot[p][edge] = (m0..m7):  b1 = m0 a1 + m1 a2 + c (m2 a1 + m3 a2),  b2 = m4 a1 + m5 a2 + c (m6 a1 + m7 a2)
static const int kFlip[6][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {1, 1, 0, 0}, {0, 1, 0, 0}, {0, 1, 1, 0}, {1, 0, 0, 1}};
#define WX_W0 {1, 0, 0, 0, 0, 1, 1, 0}
#define WX_E0 {1, 0, 0, 0, 0, 1, -1, 0}
static const double kRot[6][4][8] = {
    {{1, 0, 0, 1, 0, 1, 0, 0}, {1, 0, 0, -1, 0, 1, 0, 0}, WX_W0, WX_E0},
    {{0, 1, 0, 0, -1, 0, 0, -1}, {0, -1, 0, 0, 1, 0, 0, -1}, WX_W0, WX_E0},
    {{-1, 0, 0, -1, 0, -1, 0, 0}, {-1, 0, 0, 1, 0, -1, 0, 0}, WX_W0, WX_E0},
    {{0, -1, 0, 0, 1, 0, 0, 1}, {0, 1, 0, 0, -1, 0, 0, 1}, WX_W0, WX_E0},
    {{1, 0, 0, 1, 0, 1, 0, 0}, {-1, 0, 0, 1, 0, -1, 0, 0}, {0, -1, -1, 0, 1, 0, 0, 0}, {0, 1, -1, 0, -1, 0, 0, 0}},
    {{-1, 0, 0, -1, 0, -1, 0, 0}, {1, 0, 0, -1, 0, 1, 0, 0}, {0, 1, 1, 0, -1, 0, 0, 0}, {0, -1, 1, 0, 1, 0, 0, 0}},
};


// Rotation of the horizontal contravariant pair into the neighbour panel's basis, c = 2X/(1+X^2)
template <typename T>
__device__ __forceinline__ void rotate_contra(const double* m, double X, T& a1, T& a2) {
    const double c = 2.0 * X / (1.0 + X * X);
    const T b1 = m[0] * a1 + m[1] * a2 + c * (m[2] * a1 + m[3] * a2);
    const T b2 = m[4] * a1 + m[5] * a2 + c * (m[6] * a1 + m[7] * a2);
    a1 = b1;
    a2 = b2;
}

}  // namespace wx
