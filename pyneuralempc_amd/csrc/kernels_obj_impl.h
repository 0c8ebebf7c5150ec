// Objective of the quadratic / linear family (objective/jax.py:28-41 stand-in), shared by the assembly kernels
// (kernels_post.hip) and the fused fixed-shape evaluation kernel (kernels_coopfx_impl.h).
#pragma once

#include "nempc_internal.h"

namespace nempc {

// Sum over the wave, valid in lane 0, in the pairing order of the shuffle tree (offsets 32, 16, 8, 4, 2, 1; lane i adds
// lane i + offset) -- on the vector unit alone: gfx950's row swaps for the two wide steps, DPP row shifts for the four
// narrow ones.  The ds_bpermute tree it replaces is six dependent LDS round trips.
template <int N>
__device__ __forceinline__ double row_shl_add(double s) {
    const int lo = __double2loint(s), hi = __double2hiint(s);
    // row_shl:N -- lane i reads lane i + N of its 16-lane row; lanes whose source is outside the row read 0
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, 0x100 + N, 0xf, 0xf, false);
    const int h2 = __builtin_amdgcn_update_dpp(0, hi, 0x100 + N, 0xf, 0xf, false);
    return s + __hiloint2double(h2, l2);
}
__device__ __forceinline__ double wave_sum_lane0(double s) {
    {
        auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(s), (unsigned)__double2loint(s), false, false);
        auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(s), (unsigned)__double2hiint(s), false, false);
        s = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);     // lane i < 32: s_i + s_{i+32}
    }
    {
        auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(s), (unsigned)__double2loint(s), false, false);
        auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(s), (unsigned)__double2hiint(s), false, false);
        s = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);     // lane i < 16: + lane i + 16
    }
    s = row_shl_add<8>(s);
    s = row_shl_add<4>(s);
    s = row_shl_add<2>(s);
    s = row_shl_add<1>(s);
    return s;
}

// Maximum / minimum over the wave, valid in lane 0, same tree: lanes without a partner inside their row keep their own
// value (idempotent operations need no neutral element).
template <int N, bool MAX>
__device__ __forceinline__ double row_shl_ext(double s) {
    const int lo = __double2loint(s), hi = __double2hiint(s);
    const int l2 = __builtin_amdgcn_update_dpp(lo, lo, 0x100 + N, 0xf, 0xf, false);
    const int h2 = __builtin_amdgcn_update_dpp(hi, hi, 0x100 + N, 0xf, 0xf, false);
    const double o = __hiloint2double(h2, l2);
    return MAX ? fmax(s, o) : fmin(s, o);
}
template <bool MAX>
__device__ __forceinline__ double wave_ext_lane0(double s) {
    {
        auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(s), (unsigned)__double2loint(s), false, false);
        auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(s), (unsigned)__double2hiint(s), false, false);
        const double x = __hiloint2double((int)h[0], (int)l[0]), y = __hiloint2double((int)h[1], (int)l[1]);
        s = MAX ? fmax(x, y) : fmin(x, y);
    }
    {
        auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(s), (unsigned)__double2loint(s), false, false);
        auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(s), (unsigned)__double2hiint(s), false, false);
        const double x = __hiloint2double((int)h[0], (int)l[0]), y = __hiloint2double((int)h[1], (int)l[1]);
        s = MAX ? fmax(x, y) : fmin(x, y);
    }
    s = row_shl_ext<8, MAX>(s);
    s = row_shl_ext<4, MAX>(s);
    s = row_shl_ext<2, MAX>(s);
    s = row_shl_ext<1, MAX>(s);
    return s;
}
__device__ __forceinline__ double wave_max_lane0(double s) { return wave_ext_lane0<true>(s); }
__device__ __forceinline__ double wave_min_lane0(double s) { return wave_ext_lane0<false>(s); }
// lane 0's value in every lane (a scalar register; all lanes of the wave must be active)
__device__ __forceinline__ double wave_bcast_lane0(double s) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(s)), __builtin_amdgcn_readfirstlane(__double2loint(s)));
}

// one wave per problem; lanes stride over the horizon.  z: the problem's variables (global memory, or a copy in LDS)
// (returns the objective value, valid in lane 0; writes the gradient row of problem b when grad is given)
template <typename T>
__device__ __forceinline__ double objective_row_value(int b, int lane, int H, int nx, int nu, const ObjOffsets& o,
                                                      const T* __restrict__ P, const T* __restrict__ z,
                                                      T* __restrict__ grad) {
    const int n = H * (nx + nu);
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
    return wave_sum_lane0(acc);
}
template <typename T>
__device__ __forceinline__ void objective_row(int b, int lane, int H, int nx, int nu, const ObjOffsets& o,
                                              const T* __restrict__ P, const T* __restrict__ z,
                                              T* __restrict__ f, T* __restrict__ grad) {
    const double acc = objective_row_value<T>(b, lane, H, nx, nu, o, P, z, grad);
    if (f && lane == 0) f[b] = (T)acc;
}

template <typename T>
__device__ __forceinline__ void objective_body(int b, int lane, int H, int nx, int nu, const ObjOffsets& o,
                                               const T* __restrict__ P, const T* __restrict__ Z,
                                               T* __restrict__ f, T* __restrict__ grad) {
    objective_row<T>(b, lane, H, nx, nu, o, P, Z + (size_t)b * (H * (nx + nu)), f, grad);
}

}  // namespace nempc
