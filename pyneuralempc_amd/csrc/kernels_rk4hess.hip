// RK4 Lagrangian blocks with the network work on the matrix cores.
//
// Reference: RK4Integrator.hessian (integrator/rk4.py:181-285) propagates a second-order chain rule
//     h_{k+1} = R_k^T Hf(xi_k) R_k + c DT sum_j df_i/dxi_j h_{k,j},   R_k = I + c DT [dk_k ; 0]
// per output and only for nx+nu = 3 (hard-coded np.eye(3,3)); contracted with the multipliers
// (optimizer/ipopt.py:79-80) that is
//     d2(lam . Phi) = sum_s R_s^T ( sum_i nu_s[i] Hf_i(xi_s) ) R_s
// with stage multipliers from the adjoint recursion nu_3 = DT/6 lam, nu_{s-1} = DT/6 w_{s-1} lam + c_s J_s[:, :nx]^T nu_s
// (same formulation as rowhess_valu_kernel in kernels_valu.hip, which stays the generic path and the A/B check).
//
// Pipeline (four launches, all on the caller's stream):
//   1. rows_mfma_kernel with stage records: per (row, stage) [xi_s | J_s | dk_{s-1}]           (matrix cores)
//   2. rk4_nu_kernel: the adjoint recursion, thread per row                                       (tiny)
//   3. rowhess_mfma_kernel in direct mode over the 4*R (row, stage) pairs: Htilde_s = sum_i nu_s[i] Hf_i(xi_s)
//      by forward-over-reverse                                                                    (matrix cores)
//   4. rk4_congruence_kernel: blk = sum_s R_s^T Htilde_s R_s, wave per row out of LDS, exactly symmetric
#include "nempc_internal.h"

namespace nempc {

namespace {

template <typename T>
__global__ __launch_bounds__(256) void rk4_nu_kernel(size_t R, int H, int nx, int nin, int m, T DT,
                                                     const T* __restrict__ lam, const T* __restrict__ stage,
                                                     int stride, T* __restrict__ nu) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const size_t b = r / H, t = r - b * H;
    const T* lrow = lam + b * m + t * nx;
    const T s6 = DT / T(6);
    T* nr = nu + r * 4 * nx;
    for (int i = 0; i < nx; ++i) nr[3 * nx + i] = s6 * lrow[i];
    for (int s = 3; s >= 1; --s) {
        const T c = (s == 3) ? DT : T(0.5) * DT;
        const T wprev = (s - 1 == 0) ? T(1) : T(2);
        const T* J = stage + (r * 4 + s) * (size_t)stride + nin;   // (nx, nin) row-major
        for (int j = 0; j < nx; ++j) {
            T v = T(0);
            for (int i = 0; i < nx; ++i) v = fma(J[i * nin + j], nr[s * nx + i], v);
            nr[(s - 1) * nx + j] = s6 * wprev * lrow[j] + c * v;
        }
    }
}

// one wave per (problem, step) row; LDS per wave: Ht[4][nin*nin] | C[3][nx*nin] with C_s = c_s dk_{s-1} | M[3][nin*nin] | out[nin*nin]
template <typename T>
__global__ __launch_bounds__(256) void rk4_congruence_kernel(size_t R, int nx, int nin, T DT,
                                                             const T* __restrict__ stage, int stride,
                                                             const T* __restrict__ ht, T* __restrict__ blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t r = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= R) return;   // whole wave leaves together; no block-level barrier below
    const int nn = nin * nin, jn = nx * nin;
    T* sH = reinterpret_cast<T*>(lds_raw) + (size_t)wave * (8 * nn + 3 * jn);
    T* sC = sH + 4 * nn;
    for (int e = lane; e < 4 * nn; e += 64) sH[e] = ht[r * 4 * nn + e];
    for (int e = lane; e < 3 * jn; e += 64) {
        const int s = e / jn + 1, rem = e - (s - 1) * jn;
        const T c = (s == 3) ? DT : T(0.5) * DT;
        sC[e] = c * stage[(r * 4 + s) * (size_t)stride + nin + jn + rem];
    }
    // this wave's LDS writes become visible to its own lanes (no other wave touches this slice)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // R^T Ht R with R = I + [C ; 0], summed over the stages, in two products through LDS (entry by entry it is
    // O(nx^2) multiply-adds and twice as many LDS reads per entry and stage, and both triangles are needed):
    //   M_s = Ht_s R_s:        M_s[p][q] = Ht_s[p][q] + sum_e Ht_s[p][e] C_s[e][q]
    //   out = Ht_0 + sum_s R_s^T M_s:  out[p][q] = Ht_0[p][q] + sum_s (M_s[p][q] + sum_e C_s[e][p] M_s[e][q])
    T* sM = sC + 3 * jn;
    T* sO = sM + 3 * nn;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int idx = lane; idx < 3 * nn; idx += 64) {
        const int s = idx / nn, pq = idx - s * nn, p = pq / nin, q = pq - p * nin;
        const T* Hs = sH + (s + 1) * nn;
        const T* C = sC + s * jn;
        T acc = Hs[pq];
        for (int e = 0; e < nx; ++e) acc = fma(Hs[p * nin + e], C[e * nin + q], acc);
        sM[idx] = acc;
    }
    wave_sync();
    for (int pq = lane; pq < nn; pq += 64) {
        const int p = pq / nin, q = pq - p * nin;
        T v = sH[pq];
        for (int s = 0; s < 3; ++s) {
            const T* M = sM + s * nn;
            const T* C = sC + s * jn;
            T acc = M[pq];
            for (int e = 0; e < nx; ++e) acc = fma(C[e * nin + p], M[e * nin + q], acc);
            v += acc;
        }
        sO[pq] = v;
    }
    wave_sync();
    for (int pq = lane; pq < nn; pq += 64) {
        const int p = pq / nin, q = pq - p * nin;
        // the two triangles round differently: average them so the block is exactly symmetric
        blocks[r * nn + pq] = (p == q) ? sO[pq] : T(0.5) * (sO[pq] + sO[q * nin + p]);
    }
}

int dev_alloc_rk4(void** p, size_t bytes) {
    if (*p) return NEMPC_OK;
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) {
        *p = nullptr;
        set_error(std::string("hipMalloc(rk4 hessian workspace): ") + hipGetErrorString(e));
        return NEMPC_ENOMEM;
    }
    return NEMPC_OK;
}

template <typename T>
int run_small_kernels(Handle& h, size_t R, const void* lambda, void* blocks, int stride, bool congruence, hipStream_t s) {
    const int nx = h.cfg.nx, nin = h.nin;
    if (!congruence) {
        hipLaunchKernelGGL(rk4_nu_kernel<T>, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, R, h.cfg.H, nx, nin, h.m,
                           (T)h.cfg.DT, (const T*)lambda, (const T*)h.d_rk4_stage, stride, (T*)h.d_rk4_nu);
    } else {
        const size_t lds = 4 * (size_t)(8 * nin * nin + 3 * nx * nin) * sizeof(T);
        NEMPC_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(rk4_congruence_kernel<T>), lds));
        hipLaunchKernelGGL(rk4_congruence_kernel<T>, dim3((unsigned)((R + 3) / 4)), dim3(256), lds, s, R, nx, nin,
                           (T)h.cfg.DT, (const T*)h.d_rk4_stage, stride, (const T*)h.d_rk4_ht, (T*)blocks);
    }
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

}  // namespace

void rk4hess_free(Handle& h) {
    for (void** p : {&h.d_rk4_stage, &h.d_rk4_nu, &h.d_rk4_ht}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}

int launch_rowhess_rk4_mfma(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks,
                            hipStream_t s, void* g_out, void* tiles_out) {
    const int nx = h.cfg.nx, nin = h.nin;
    const int stride = nin + 2 * nx * nin;
    const size_t Rcap = (size_t)h.cfg.max_batch * h.cfg.H, R = (size_t)B * h.cfg.H;
    int rc;
    if ((rc = dev_alloc_rk4(&h.d_rk4_stage, Rcap * 4 * stride * h.esz))) return rc;
    if ((rc = dev_alloc_rk4(&h.d_rk4_nu, Rcap * 4 * nx * h.esz))) return rc;
    if ((rc = dev_alloc_rk4(&h.d_rk4_ht, Rcap * 4 * nin * nin * h.esz))) return rc;
    // 1. stage records; the launch's defects and tiles go to the caller (the batched solver wants them anyway) or to
    //    the handle's scratch outputs
    if ((rc = launch_rows_mfma_stages(h, B, Z, X0, g_out ? g_out : h.d_g_ws, tiles_out ? tiles_out : h.d_tiles_ws,
                                      h.d_rk4_stage, stride, s)))
        return rc;
    const bool f64 = h.cfg.dtype == NEMPC_F64;
    // 2. stage multipliers
    if ((rc = f64 ? run_small_kernels<double>(h, R, lambda, blocks, stride, false, s)
                  : run_small_kernels<float>(h, R, lambda, blocks, stride, false, s)))
        return rc;
    // 3. contracted network Hessians at the four stage inputs
    if ((rc = launch_rowhess_mfma_direct(h, B, Z, X0, lambda, h.d_rk4_ht, h.d_rk4_stage, stride, h.d_rk4_nu, 4, s)))
        return rc;
    h.last_hess_kernel += 10;      // (the network kernel of step 3, inside the pipeline)
    // 4. congruence sum
    return f64 ? run_small_kernels<double>(h, R, lambda, blocks, stride, true, s)
               : run_small_kernels<float>(h, R, lambda, blocks, stride, true, s);
}

// The same pipeline for networks on the layer-at-a-time GEMM path (kernels_layered.hip): steps 1 and 3 are its launches
int launch_rowhess_rk4_layered(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks, hipStream_t s,
                               void* g_out, void* tiles_out) {
    const int nx = h.cfg.nx, nin = h.nin;
    const int stride = nin + 2 * nx * nin;
    const size_t Rcap = (size_t)h.cfg.max_batch * h.cfg.H, R = (size_t)B * h.cfg.H;
    int rc;
    if ((rc = dev_alloc_rk4(&h.d_rk4_stage, Rcap * 4 * stride * h.esz))) return rc;
    if ((rc = dev_alloc_rk4(&h.d_rk4_nu, Rcap * 4 * nx * h.esz))) return rc;
    if ((rc = dev_alloc_rk4(&h.d_rk4_ht, Rcap * 4 * nin * nin * h.esz))) return rc;
    if ((rc = launch_rows_layered_stages(h, B, Z, X0, g_out ? g_out : h.d_g_ws, tiles_out ? tiles_out : h.d_tiles_ws, h.d_rk4_stage,
                                         stride, s)))
        return rc;
    const bool f64 = h.cfg.dtype == NEMPC_F64;
    if ((rc = f64 ? run_small_kernels<double>(h, R, lambda, blocks, stride, false, s)
                  : run_small_kernels<float>(h, R, lambda, blocks, stride, false, s)))
        return rc;
    if ((rc = launch_rowhess_layered_direct(h, (long long)R * 4, h.d_rk4_stage, stride, h.d_rk4_nu, h.d_rk4_ht, s))) return rc;
    h.last_hess_kernel += 10;
    return f64 ? run_small_kernels<double>(h, R, lambda, blocks, stride, true, s)
               : run_small_kernels<float>(h, R, lambda, blocks, stride, true, s);
}

}  // namespace nempc
