// Matrix-core kernel for the per-row Lagrangian blocks of the Hessian callback (DISCRET / UNITY):
//     Hblk[p][d] = d^2 (lambda . f) / d xi_p d xi_d          (w*(nx+nu) square per (problem, step) row)
// i.e. the lambda-contracted form of Model.hessian (model/tensorflow.py:77-109) that
// IpoptProblem.hessian sums with the multipliers (optimizer/ipopt.py:66-86).
//
// Method: forward-over-reverse.  One wave owns a tile of 16 rows (same operand layout as
// kernels_mfma_impl.h: features on the MFMA M dimension, rows on N, activations never leave registers).
//   base sweep    a_l (forward), delta_l = d(lambda.f)/d a_l (reverse); kept as
//                 S1_l = s'(z_l) = d1(a_l)  and  E_l = delta_l r2(a_l)   (activations.h; tanh: 1 - a^2 and -2 delta a;
//                 so that d(delta_l * S1_l) = S1_l * d delta_l + E_l * d a_l)
//   per input p   tangent forward  da_0 = S1_0 * W_0[p,:],  da_l = S1_l * (W_l^T da_{l-1})
//                 tangent reverse  dcz_{L} = E_L * da_L,  dcz_{l-1} = S1_{l-1} * (W_l dcz_l) + E_{l-1} * da_{l-1}
//                 column           H[:, p] = W_0 dcz_0        (skinny MFMA, like the Jacobian's last step)
// MFMA count per tile: base (forward + reverse) + nin * (2 hidden sweeps + 1 skinny), all K >= 16
// contractions on the matrix cores; the VALU only does the elementwise products.
// One wave per SIMD (the register budget is ~300 of the 512 unified VGPRs for fp64 2x64).
#pragma once

#include "kernels_mfma_impl.h"

namespace nempc {

struct HessParams {
    MfmaParams base;     // blob, offsets, dims, Z, X0 (g/tiles unused)
    const void* lambda;  // (B, m)
    void* blocks;        // (B, H, nin, nin)
    int p0tab, wLb, ksx; // extra blob tables: first-layer rows, output-layer fragments for W_L lambda, ceil(nx/4)
    // direct mode (RK4 pipeline, kernels_rk4hess.hip): the kernel's rows are (row, stage) pairs whose network input
    // and multipliers are given explicitly instead of being gathered from Z / lambda
    const void* xi_direct;   // record r at xi_direct + r * xi_stride, first nin values = xi
    int xi_stride;
    const void* lam_direct;  // (rows, nx) stage multipliers nu_s
    int vdiv;                // rows per (problem, step) row: the extra inputs of row r are those of r / vdiv
    // fused assembly (cooperative kernel, plain models, tril values only): the kernel writes
    // hvals[b][e] = sigma_b * objc[e] + block element straight from its block buffer in LDS -- smap[t*nin*nin + pq] is
    // the tril entry a block element lands in (-1: none); the n_orph entries no block reaches (objective only) follow at
    // smap[H*nin*nin ...] and are written by the problem's last row
    void* hvals;
    const void* sigma;
    const int32_t* smap;
    const void* objc;
    int nnz, n_orph;
    // blocks AND the rows' first-order evaluation from one launch (fixed-shape kernel only; the batched solver's trial
    // point): defects (B, m) and compact tiles (B, H, nx, nin)
    void* ev_g;
    void* ev_tiles;
};

template <typename T, int WP, int NH, bool WLDS, int ACT>
__global__ __launch_bounds__(256) void rowhess_mfma_kernel(HessParams hp) {
    using Ops = MfmaOps<T>;
    using A = ActL<T, ACT>;
    using V4 = typename Ops::V4;
    constexpr int MT = WP / 16;
    const MfmaParams& p = hp.base;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const T* __restrict__ gblob = static_cast<const T*>(p.blob);
    const T* wsrc;
    T* scratch;
    if (WLDS) {
        copy_blob_to_lds<T>(gblob, lds, p.off.total, threadIdx.x, blockDim.x);
        __syncthreads();
        wsrc = lds;
        scratch = lds + ((p.off.total + 1) & ~1) + wave * p.scratch_per_wave;
    } else {
        wsrc = gblob;
        scratch = lds + wave * p.scratch_per_wave;
    }
    const int nx = p.nx, nin = p.nin, H = p.H;
    const int n = p.gk.n;
    const size_t R = (size_t)p.B * H * (hp.xi_direct ? hp.vdiv : 1);
    const T* __restrict__ Z = static_cast<const T*>(p.Z);
    const T* __restrict__ X0 = static_cast<const T*>(p.X0);
    const T* __restrict__ lam = static_cast<const T*>(hp.lambda);
    T* __restrict__ blocks = static_cast<T*>(hp.blocks);

    // per-wave scratch: xi0[16][nin] | lam[16][nx] | Hs[16][nin][nin]
    T* s_xi0 = scratch;
    T* s_lam = s_xi0 + 16 * nin;
    T* s_H = s_lam + 16 * nx;
    T* s_ex = s_H + 16 * nin * nin;   // [16][ne] extra inputs

    for (int tile = blockIdx.x * nwaves + wave; tile < p.ntiles; tile += gridDim.x * nwaves) {
        const size_t row0 = (size_t)tile * 16;
        for (int e = lane; e < 16 * (nin + nx); e += 64) {
            const int cc = e & 15, d = e >> 4;   // d over nin inputs then nx multipliers
            const size_t r = row0 + cc;
            T v = T(0);
            if (r < R) {
                if (hp.xi_direct) {
                    if (d < nin) v = static_cast<const T*>(hp.xi_direct)[r * (size_t)hp.xi_stride + d];
                    else v = static_cast<const T*>(hp.lam_direct)[r * nx + (d - nin)];
                } else {
                    const int b = (int)((unsigned)r / (unsigned)H), t = (int)((unsigned)r - (unsigned)b * (unsigned)H);
                    if (d < nin) v = gather_input<T>(p.gk, Z + (size_t)b * n, X0, b, t, d);
                    else v = lam[(size_t)b * p.m + t * nx + (d - nin)];
                }
            }
            if (d < nin) s_xi0[cc * nin + d] = v;
            else s_lam[cc * nx + (d - nin)] = v;
        }
        for (int e = lane; e < 16 * p.ne; e += 64) {
            const int cc = e / p.ne, j = e - cc * p.ne;
            const size_t r = row0 + cc;
            s_ex[e] = (r < R) ? static_cast<const T*>(p.extra)[(hp.xi_direct ? r / hp.vdiv : r) * p.ne + j] : T(0);
        }
        wave_sync();

        // ---- forward values
        V4 S1[NH][MT];  // holds a_l first, then 1 - a_l^2
        {
            T xin[kMaxKs];
#pragma unroll
            for (int ks = 0; ks < kMaxKs; ++ks) {
                const int d = 4 * ks + q;
                xin[ks] = (ks < p.ks && d < nin) ? s_xi0[c * nin + d]
                                                  : ((ks < p.ks && d < nin + p.ne) ? s_ex[c * p.ne + (d - nin)] : T(0));
            }
            const T* bias = wsrc + p.off.bias[0];
#pragma unroll
            for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                for (int r = 0; r < 4; ++r) S1[0][mo][r] = bias[(mo * 4 + r) * 4 + q];
            const T* w = wsrc + p.off.w0f;
#pragma unroll
            for (int ks = 0; ks < kMaxKs; ++ks) {
                if (ks < p.ks) {
#pragma unroll
                    for (int mo = 0; mo < MT; ++mo) S1[0][mo] = Ops::mma(w[(ks * MT + mo) * 64 + lane], xin[ks], S1[0][mo]);
                }
            }
#pragma unroll
            for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                for (int r = 0; r < 4; ++r) S1[0][mo][r] = A::f(S1[0][mo][r], p.acts, 0);
        }
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            const T* bias = wsrc + p.off.bias[l];
#pragma unroll
            for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                for (int r = 0; r < 4; ++r) S1[l][mo][r] = bias[(mo * 4 + r) * 4 + q];
            layer_mma<T, MT, MT, WLDS>(wsrc + p.off.wf[l], lane, S1[l - 1], S1[l]);
#pragma unroll
            for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                for (int r = 0; r < 4; ++r) S1[l][mo][r] = A::f(S1[l][mo][r], p.acts, l);
        }

        // ---- base reverse sweep: delta_l, then S1_l = 1 - a_l^2 and E_l = -2 delta_l a_l
        V4 E[NH][MT];
        {
            V4 dl[MT];
            T lamB[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int d = 4 * ks + q;
                lamB[ks] = (ks < hp.ksx && d < nx) ? s_lam[c * nx + d] : T(0);
            }
            const T* w = wsrc + hp.wLb;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                dl[mt] = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    if (ks < hp.ksx) dl[mt] = Ops::mma(w[(ks * MT + mt) * 64 + lane], lamB[ks], dl[mt]);
            }
#pragma unroll
            for (int l = NH - 1; l >= 0; --l) {
                V4 cz[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const V4 a = S1[l][mt];
                    V4 s1, e;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s1[r] = A::d1(a[r], p.acts, l);
                        e[r] = dl[mt][r] * A::r2(a[r], p.acts, l);
                    }
                    E[l][mt] = e;
                    S1[l][mt] = s1;
                    cz[mt] = dl[mt] * s1;
                }
                if (l > 0) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) dl[mt] = V4{T(0), T(0), T(0), T(0)};
                    layer_mma<T, MT, MT, WLDS>(wsrc + p.off.wb[l], lane, cz, dl);
                }
            }
        }

        // ---- one tangent sweep per input direction
        for (int pd = 0; pd < nin; ++pd) {
            V4 da[NH][MT];
            {
                const T* tab = wsrc + hp.p0tab + pd * MT * 16;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) da[0][mt][r] = S1[0][mt][r] * tab[(mt * 4 + r) * 4 + q];
            }
#pragma unroll
            for (int l = 1; l < NH; ++l) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) da[l][mt] = V4{T(0), T(0), T(0), T(0)};
                layer_mma<T, MT, MT, WLDS>(wsrc + p.off.wf[l], lane, da[l - 1], da[l]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) da[l][mt] = da[l][mt] * S1[l][mt];
            }
            V4 dcz[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) dcz[mt] = E[NH - 1][mt] * da[NH - 1][mt];
#pragma unroll
            for (int l = NH - 1; l >= 1; --l) {
                V4 ddl[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) ddl[mt] = V4{T(0), T(0), T(0), T(0)};
                layer_mma<T, MT, MT, WLDS>(wsrc + p.off.wb[l], lane, dcz, ddl);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) dcz[mt] = S1[l - 1][mt] * ddl[mt] + E[l - 1][mt] * da[l - 1][mt];
            }
            if (p.mb == 1) {
                V4 hcol[1] = {V4{T(0), T(0), T(0), T(0)}};
                layer_mma<T, MT, 1, WLDS>(wsrc + p.off.w0b, lane, dcz, hcol);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = Ops::row(q, r);
                    if (d < nin) s_H[(c * nin + pd) * nin + d] = hcol[0][r];
                }
            } else {   // 17..32 network inputs
                V4 hcol[2] = {V4{T(0), T(0), T(0), T(0)}, V4{T(0), T(0), T(0), T(0)}};
                layer_mma<T, MT, 2, WLDS>(wsrc + p.off.w0b, lane, dcz, hcol);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int d = 16 * mb + Ops::row(q, r);
                        if (d < nin) s_H[(c * nin + pd) * nin + d] = hcol[mb][r];
                    }
            }
        }
        wave_sync();

        // ---- output: exactly symmetric blocks (lower triangle mirrored), 16 rows contiguous in memory
        const int bsz = nin * nin;
        for (int e = lane; e < 16 * bsz; e += 64) {
            const int cc = e / bsz, rem = e - cc * bsz;
            const int a1 = rem / nin, a2 = rem - a1 * nin;
            const int hi = a1 > a2 ? a1 : a2, lo = a1 > a2 ? a2 : a1;
            if (row0 + cc < R) blocks[row0 * bsz + e] = s_H[(cc * nin + hi) * nin + lo];
        }
        wave_sync();
    }
}

template <typename T, int ACT>
int launch_rowhess_mfma_act(const Handle& h, HessParams hp, hipStream_t s);
template <typename T>
int launch_rowhess_mfma_typed(const Handle& h, HessParams hp, hipStream_t s);

}  // namespace nempc
