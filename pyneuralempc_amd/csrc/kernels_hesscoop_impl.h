// CU-cooperative matrix-core kernel for the per-row Lagrangian blocks (DISCRET / UNITY rows, and the (row, stage)
// pairs of the RK4 pipeline in direct mode):
//     Hblk[p][d] = d^2 (lambda . f) / d xi_p d xi_d
// Same mathematics as rowhess_mfma_kernel (kernels_hess_impl.h: forward-over-reverse, one tangent sweep per input
// direction) with the work split of rows_coop_kernel: a workgroup of MT = WP/16 waves owns a tile of 16 rows, wave w
// owns feature block w of every layer, its slices of the hidden-to-hidden weights live in registers for the whole
// launch, and activations / cotangents / tangents cross waves through a double-buffered LDS exchange area.
// TG tangent directions are swept together: TG independent MFMA chains per wave and TG-fold fewer exchange barriers.
// Register budget per wave stays under 256 VGPRs, so two workgroups (width <= 64) or two waves per SIMD (width 128)
// interleave on every SIMD -- the wave-per-tile kernel needs > 256 registers and runs one wave per SIMD.
#pragma once

#include "kernels_coop_impl.h"
#include "kernels_hess_impl.h"

namespace nempc {

struct HessCoopLayout {  // element offsets inside dynamic LDS
    int w0f;             // layer-0 fragments                               ks * MT * 64
    int tail;            // p0tab | wLb | seed | bias_l | biasL             off.total - off.p0tab
    int x, xhalf;        // exchange buffer: 2 halves of TG * MT * 256
    int part;            // K-split partials of the Hessian columns         TG * MT * NR * 64
    int scratch;         // xi0[16][nin] | lam[16][nx] | ex[16][ne] | H[16][nin][nin]
    int total;
};

template <typename T, int WP, int NH, int TG, int ACT>
__global__ __launch_bounds__((WP / 16) * 64, 2) void rowhess_coop_kernel(HessParams hp, HessCoopLayout lay) {
    using Ops = MfmaOps<T>;
    using A = ActL<T, ACT>;
    using V4 = typename Ops::V4;
    constexpr int MT = WP / 16;
    constexpr int NTHREADS = MT * 64;
    const MfmaParams& p = hp.base;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const T* __restrict__ gblob = static_cast<const T*>(p.blob);
    const int nx = p.nx, nin = p.nin, ne = p.ne, H = p.H, n = p.gk.n;
    const int NR = coop_nr<T>(nin);
    const size_t R = (size_t)p.B * H * (hp.xi_direct ? hp.vdiv : 1);
    const T* __restrict__ Z = static_cast<const T*>(p.Z);
    const T* __restrict__ X0 = static_cast<const T*>(p.X0);
    const T* __restrict__ lam = static_cast<const T*>(hp.lambda);
    T* __restrict__ blocks = static_cast<T*>(hp.blocks);

    // ---- tables -> LDS, this wave's weight slices -> registers
    T* s_w0f = lds + lay.w0f;
    T* s_tail = lds + lay.tail;
    copy_blob_to_lds<T>(gblob + p.off.w0f, s_w0f, p.ks * MT * 64, tid, NTHREADS);
    copy_blob_to_lds<T>(gblob + p.off.p0tab, s_tail, p.off.total - p.off.p0tab, tid, NTHREADS);
    const T* s_p0tab = s_tail;
    const T* s_wLb = s_tail + (p.off.wLb - p.off.p0tab);
    const T* s_bias = s_tail + (p.off.bias[0] - p.off.p0tab);   // bias_l at + l * MT * 16
    CoopWeights<T, WP, NH> W;
#pragma unroll
    for (int l = 1; l < NH; ++l)
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) {
            W.wf[l - 1][i] = gblob[p.off.wf[l] + (i * MT + w) * 64 + lane];
            W.wb[l - 1][i] = gblob[p.off.wb[l] + (i * MT + w) * 64 + lane];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) W.w0b[r] = gblob[p.off.w0b + (w * 4 + r) * 64 + lane];

    T* XB = lds + lay.x;
    T* PART = lds + lay.part;
    T* s_xi0 = lds + lay.scratch;
    T* s_lam = s_xi0 + 16 * nin;
    T* s_ex = s_lam + 16 * nx;
    T* s_H = s_ex + 16 * ne;
    int xsel = 0;

    // contiguous, balanced range of tiles for this workgroup
    const int t_begin = blockIdx.x * p.tiles_per_wg + ((int)blockIdx.x < p.tiles_rem ? (int)blockIdx.x : p.tiles_rem);
    const int t_end = t_begin + p.tiles_per_wg + ((int)blockIdx.x < p.tiles_rem ? 1 : 0);

    for (int tile = t_begin; tile < t_end; ++tile) {
        const size_t row0 = (size_t)tile * 16;
        // ---- stage inputs: window inputs, multipliers, extras (columns x rows, flat over the workgroup)
        for (int e = tid; e < 16 * (nin + nx + ne); e += NTHREADS) {
            const int cc = e & 15, d = e >> 4;
            const size_t r = row0 + cc;
            T v = T(0);
            if (r < R) {
                if (d < nin + nx) {
                    if (hp.xi_direct) {
                        v = d < nin ? static_cast<const T*>(hp.xi_direct)[r * (size_t)hp.xi_stride + d]
                                    : static_cast<const T*>(hp.lam_direct)[r * nx + (d - nin)];
                    } else {
                        const int b = (int)((unsigned)r / (unsigned)H), t = (int)((unsigned)r - (unsigned)b * (unsigned)H);
                        v = d < nin ? gather_input<T>(p.gk, Z + (size_t)b * n, X0, b, t, d)
                                    : lam[(size_t)b * p.m + t * nx + (d - nin)];
                    }
                } else {
                    v = static_cast<const T*>(p.extra)[(hp.xi_direct ? r / hp.vdiv : r) * ne + (d - nin - nx)];
                }
            }
            if (d < nin) s_xi0[cc * nin + d] = v;
            else if (d < nin + nx) s_lam[cc * nx + (d - nin)] = v;
            else s_ex[cc * ne + (d - nin - nx)] = v;
        }
        lds_barrier();

        // ---- forward values of this wave's feature block: S1[l] holds a_l, later s'(z_l) (tanh: 1 - a_l^2)
        V4 S1[NH], E[NH];
        {
            const T* bias = s_bias + w * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) S1[0][r] = bias[r * 4 + q];
            for (int ks = 0; ks < p.ks; ++ks) {
                const int d = 4 * ks + q;
                T v = T(0);
                if (d < nin) v = s_xi0[c * nin + d];
                else if (d < nin + ne) v = s_ex[c * ne + (d - nin)];
                S1[0] = Ops::mma(s_w0f[(ks * MT + w) * 64 + lane], v, S1[0]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) S1[0][r] = A::f(S1[0][r], p.acts, 0);
        }
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            T* X = XB + (xsel & 1) * lay.xhalf;
            ++xsel;
#pragma unroll
            for (int r = 0; r < 4; ++r) X[(w * 4 + r) * 64 + lane] = S1[l - 1][r];
            lds_barrier();
            const T* bias = s_bias + l * MT * 16 + w * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) S1[l][r] = bias[r * 4 + q];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) S1[l] = Ops::mma(W.wf[l - 1][mt * 4 + r], X[(mt * 4 + r) * 64 + lane], S1[l]);
#pragma unroll
            for (int r = 0; r < 4; ++r) S1[l][r] = A::f(S1[l][r], p.acts, l);
        }

        // ---- base reverse sweep: delta_l = d(lambda.f)/d a_l; keep S1_l = s'(z_l) and E_l = delta_l r2(a_l) (tanh: 1 - a_l^2, -2 delta_l a_l)
        {
            V4 dl = V4{T(0), T(0), T(0), T(0)};
            for (int ks = 0; ks < hp.ksx; ++ks) {
                const int d = 4 * ks + q;
                const T lb = d < nx ? s_lam[c * nx + d] : T(0);
                dl = Ops::mma(s_wLb[(ks * MT + w) * 64 + lane], lb, dl);
            }
#pragma unroll
            for (int l = NH - 1; l >= 0; --l) {
                const V4 a = S1[l];
                V4 s1, e;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1[r] = A::d1(a[r], p.acts, l);
                    e[r] = dl[r] * A::r2(a[r], p.acts, l);
                }
                E[l] = e;
                S1[l] = s1;
                if (l > 0) {
                    const V4 cz = dl * s1;
                    T* X = XB + (xsel & 1) * lay.xhalf;
                    ++xsel;
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[(w * 4 + r) * 64 + lane] = cz[r];
                    lds_barrier();
                    dl = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            dl = Ops::mma(W.wb[l - 1][mt * 4 + r], X[(mt * 4 + r) * 64 + lane], dl);
                }
            }
        }

        // ---- tangent sweeps, TG input directions at a time (a leftover direction is swept twice, stored once)
        for (int pd0 = 0; pd0 < nin; pd0 += TG) {
            V4 da[TG][NH];
#pragma unroll
            for (int g = 0; g < TG; ++g) {
                const int pd = pd0 + g < nin ? pd0 + g : nin - 1;
                const T* tab = s_p0tab + pd * MT * 16 + w * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) da[g][0][r] = S1[0][r] * tab[r * 4 + q];
            }
#pragma unroll
            for (int l = 1; l < NH; ++l) {
                T* X = XB + (xsel & 1) * lay.xhalf;
                ++xsel;
#pragma unroll
                for (int g = 0; g < TG; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[((g * MT + w) * 4 + r) * 64 + lane] = da[g][l - 1][r];
                lds_barrier();
#pragma unroll
                for (int g = 0; g < TG; ++g) da[g][l] = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int g = 0; g < TG; ++g)
                            da[g][l] = Ops::mma(W.wf[l - 1][mt * 4 + r], X[((g * MT + mt) * 4 + r) * 64 + lane], da[g][l]);
#pragma unroll
                for (int g = 0; g < TG; ++g) da[g][l] = da[g][l] * S1[l];
            }
            V4 dcz[TG];
#pragma unroll
            for (int g = 0; g < TG; ++g) dcz[g] = E[NH - 1] * da[g][NH - 1];
#pragma unroll
            for (int l = NH - 1; l >= 1; --l) {
                T* X = XB + (xsel & 1) * lay.xhalf;
                ++xsel;
#pragma unroll
                for (int g = 0; g < TG; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[((g * MT + w) * 4 + r) * 64 + lane] = dcz[g][r];
                lds_barrier();
                V4 ddl[TG];
#pragma unroll
                for (int g = 0; g < TG; ++g) ddl[g] = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int g = 0; g < TG; ++g)
                            ddl[g] = Ops::mma(W.wb[l - 1][mt * 4 + r], X[((g * MT + mt) * 4 + r) * 64 + lane], ddl[g]);
#pragma unroll
                for (int g = 0; g < TG; ++g) dcz[g] = S1[l - 1] * ddl[g] + E[l - 1] * da[g][l - 1];
            }
            // Hessian columns H[:, pd] = W_0 dcz_0: K-split partial over this wave's block, reduced below
#pragma unroll
            for (int g = 0; g < TG; ++g) {
                V4 pj = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int r = 0; r < 4; ++r) pj = Ops::mma(W.w0b[r], dcz[g][r], pj);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < NR) PART[((g * MT + w) * NR + r) * 64 + lane] = pj[r];
            }
            lds_barrier();
            for (int item = tid; item < TG * 16 * nin; item += NTHREADS) {
                const int g = item / (16 * nin), rem = item - g * 16 * nin;
                const int d = rem >> 4, cc = rem & 15;
                if (pd0 + g < nin) {
                    // accumulator layout: row index d sits at (q', r') with d = row(q', r')
                    const int qq = sizeof(T) == 8 ? (d & 3) : (d >> 2), rr = sizeof(T) == 8 ? (d >> 2) : (d & 3);
                    T acc = T(0);
                    for (int ww = 0; ww < MT; ++ww) acc += PART[((g * MT + ww) * NR + rr) * 64 + qq * 16 + cc];
                    s_H[(cc * nin + (pd0 + g)) * nin + d] = acc;
                }
            }
            lds_barrier();
        }

        // ---- output: exactly symmetric blocks (lower triangle mirrored), the tile's 16 rows contiguous in memory
        const int bsz = nin * nin;
        if (hp.hvals) {
            // fused assembly (nempc_hess asked for the tril values only): assemble_hess_kernel's arithmetic, here
            T* hv = static_cast<T*>(hp.hvals);
            const T* sg = static_cast<const T*>(hp.sigma);
            const T* oc = static_cast<const T*>(hp.objc);
            const int Hh = p.H;
            for (int e = tid; e < 16 * bsz; e += NTHREADS) {
                const int cc = e / bsz, rem = e - cc * bsz;
                const size_t r = (size_t)row0 + cc;
                if (r >= (size_t)R) continue;
                const int a1 = rem / nin, a2 = rem - a1 * nin;
                const int hi = a1 > a2 ? a1 : a2, lo = a1 > a2 ? a2 : a1;
                const size_t b = r / Hh;
                const int t = (int)(r - b * Hh);
                const int ent = hp.smap[t * bsz + rem];
                if (ent >= 0) {
                    T v = sg[b] * oc[ent];
                    v += s_H[(cc * nin + hi) * nin + lo];
                    hv[b * hp.nnz + ent] = v;
                }
                if (t == Hh - 1 && rem < hp.n_orph) {
                    const int oe = hp.smap[Hh * bsz + rem];
                    hv[b * hp.nnz + oe] = sg[b] * oc[oe];
                }
            }
        } else
        for (int e = tid; e < 16 * bsz; e += NTHREADS) {
            const int cc = e / bsz, rem = e - cc * bsz;
            const int a1 = rem / nin, a2 = rem - a1 * nin;
            const int hi = a1 > a2 ? a1 : a2, lo = a1 > a2 ? a2 : a1;
            if (row0 + cc < R) blocks[row0 * bsz + e] = s_H[(cc * nin + hi) * nin + lo];
        }
        lds_barrier();
    }
}

}  // namespace nempc
