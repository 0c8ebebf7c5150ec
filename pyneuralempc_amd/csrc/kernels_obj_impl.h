// Objective of the quadratic / linear family (objective/jax.py:28-41 stand-in), shared by the assembly kernels
// (kernels_post.hip) and the fused fixed-shape evaluation kernel (kernels_coopfx_impl.h).
#pragma once

#include "nempc_internal.h"

namespace nempc {

// one wave per problem; lanes stride over the horizon
template <typename T>
__device__ __forceinline__ void objective_body(int b, int lane, int H, int nx, int nu, const ObjOffsets& o,
                                               const T* __restrict__ P, const T* __restrict__ Z,
                                               T* __restrict__ f, T* __restrict__ grad) {
    const int n = H * (nx + nu);
    const T* z = Z + (size_t)b * n;
    const T *Rm = P + o.R, *Rs = P + o.Rs;
    const T *xref = P + o.xref, *uref = P + o.uref, *cx = P + o.cx, *cu = P + o.cu;
    double acc = 0.0;
    for (int t = lane; t < H; t += 64) {
        // the last step may carry its own state weight (terminal cost)
        const T* Q = t == H - 1 ? P + o.QT : P + o.Q;
        const T* Qs = t == H - 1 ? P + o.QTs : P + o.Qs;
        const T* x = z + t * nx;
        const T* u = z + H * nx + t * nu;
        for (int i = 0; i < nx; ++i) {
            const T dxi = x[i] - xref[t * nx + i];
            T qd = T(0), qsd = T(0);
            for (int j = 0; j < nx; ++j) {
                const T dxj = x[j] - xref[t * nx + j];
                qd = fma(Q[i * nx + j], dxj, qd);
                qsd = fma(Qs[i * nx + j], dxj, qsd);
            }
            acc += (double)(dxi * qd + cx[t * nx + i] * x[i]);
            if (grad) grad[(size_t)b * n + t * nx + i] = qsd + cx[t * nx + i];
        }
        for (int i = 0; i < nu; ++i) {
            const T dui = u[i] - uref[t * nu + i];
            T rd = T(0), rsd = T(0);
            for (int j = 0; j < nu; ++j) {
                const T duj = u[j] - uref[t * nu + j];
                rd = fma(Rm[i * nu + j], duj, rd);
                rsd = fma(Rs[i * nu + j], duj, rsd);
            }
            acc += (double)(dui * rd + cu[t * nu + i] * u[i]);
            if (grad) grad[(size_t)b * n + H * nx + t * nu + i] = rsd + cu[t * nu + i];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (f && lane == 0) f[b] = (T)acc;
}


}  // namespace nempc
