// Fixed-shape cooperative kernel for the per-row Lagrangian blocks (gfx950): plain Discret / Unity models of a compiled
// shape (nx, nu as template parameters; BASELINE configs[1] / configs[4]: 2 states, 1 control, MLP 2x64).
//     Hblk[p][q] = d^2 (lambda . f) / d xi_p d xi_q                     (IpoptProblem.hessian, optimizer/ipopt.py:66-86,
//                                                                        contracting Model.hessian, tensorflow.py:77-109)
//
// A different factorisation from rowhess_coop_kernel (kernels_hesscoop_impl.h), which runs forward-over-reverse: one
// tangent-forward AND one tangent-reverse sweep per input direction, 2 nin hidden-layer products on top of the base
// sweeps.  Here the second-order chain rule is summed layer by layer instead,
//     Hblk = sum_l  P_l^T diag(delta_l * s''(z_l)) P_l,        P_l = d z_l / d xi   (pre-activation tangents),
//                                                               delta_l = d (lambda . f) / d a_l (ONE reverse sweep),
// so the matrix cores only carry the base forward sweep, the base reverse sweep and nin tangent-FORWARD sweeps --
// 1 + (NH-1)(2 + nin) 16-step products per tile and wave: 81 MFMAs at 2/1, 2x64 against 142 -- and the contraction
// over the hidden units, which has per-row operands and so is no shared-weight product, runs on the vector unit where
// a lane already holds its four features of every operand: nin(nin+1)/2 pairs x 4 features x NH layers fused
// multiply-adds (48 at 2/1), fewer than the elementwise work of the tangent-reverse sweeps it replaces.  P_0 is the
// first layer's weight rows (constant: its pair products are a table), s'' = r2(a) d1(a) from the layer's output
// (activations.h).  The work split is rows_coopfx_kernel's: a workgroup of MT = WP/16 waves owns a pass of NT tiles,
// wave w owns feature block w of every layer with its hidden-to-hidden slices in registers, activations / cotangents /
// tangents cross waves through the double-buffered LDS exchange area, the skinny layer (W_L lambda) runs on the
// vector unit, the quad sums use the row swaps, the next pass's inputs are fetched under the current one.
// Outputs: the full symmetric blocks (B, H, nin, nin), or -- nempc_hess asked for the tril values only -- the assembled
// hvals (sigma * objective constant + block element through the scatter map), as the generic kernel.
#pragma once

#include "kernels_coopfx_impl.h"
#include "kernels_hess_impl.h"

namespace nempc {

template <typename T, int WP, int NH, int NT, int NX, int NU, int TG = NX + NU, bool EV = false>
struct HfxLayout {   // element offsets inside dynamic LDS, all compile-time (TG: tangent directions per exchange)
    static constexpr int MT = WP / 16;
    static constexpr int NIN = NX + NU;
    static constexpr int KS = (NIN + 3) / 4;
    static constexpr int NPAIR = NIN * (NIN + 1) / 2;
    static constexpr int NVT = NPAIR + (EV ? NX * NIN + NX : 0);      // K-split quantities per row: pairs [+ tile + outputs]
    static constexpr int up16(int v) { return (v + 15) & ~15; }
    // small tables, copied flat from off.fx_small: [w0f | seed | bias_0..NH-1 | biasL | p0tab]
    static constexpr int W0F = 0;
    static constexpr int SEED = W0F + KS * MT * 64;
    static constexpr int BIAS = SEED + NX * MT * 16;
    static constexpr int BIASL = BIAS + NH * MT * 16;
    static constexpr int P0 = BIASL + 16;                            // first-layer rows per lane: [d][MT*16]
    static constexpr int SMALL_END = P0 + NIN * MT * 16;
    static constexpr int PP = up16(SMALL_END);                       // pair products P0[p] * P0[q] per feature: [pq][MT*16]
    static constexpr int XH = NT * TG * MT * 256;                    // exchange buffer: two halves of NT * TG sets
    static constexpr int X = up16(PP + NPAIR * MT * 16);
    static constexpr int PART = X + 2 * XH;                          // K-split partials [w][j][pq][16 rows]
    static constexpr int PART_SZ = up16(MT * NT * NVT * 16);
    static constexpr int IN_TILE = 16 * (NIN + NX);                  // per tile xi[16][NIN] then lambda[16][NX]
    static constexpr int IN = PART + PART_SZ;
    static constexpr int IN_SZ = up16(NT * IN_TILE);
    static constexpr int TOTAL = IN + 2 * IN_SZ;
};

struct HfxArgs {   // host-prepared
    const void* Z;
    const void* X0;
    const void* lambda;     // (B, m)
    const void* small;      // blob + off.fx_small
    const void* wslice;     // blob + off.coop_slices
    int tiles_per_wg, tiles_rem;
    unsigned R;             // rows = B*H
    unsigned invH;          // ceil(2^32 / H), 0 for H == 1
    int H, n, m;
    int nload;              // 16-byte loads per lane of a wave's packed slice (the kernel takes the hidden-layer part)
    // direct mode (RK4 pipeline, kernels_rk4hess.hip): the rows are (row, stage) pairs whose network input and
    // multipliers are given explicitly -- record r at xi_direct + r * xi_stride (first nin values), lam_direct (rows, nx)
    const void* xi_direct;
    const void* lam_direct;
    int xi_stride;
    void* blocks;           // (B, H, nin, nin) [direct mode: (rows, nin, nin)] or null
    // fused tril assembly (see HessParams)
    void* hvals;
    const void* sigma;
    const int32_t* smap;
    const void* objc;
    int nnz, n_orph;
    // EV instantiation (the batched solver): the first-order evaluation of the same rows leaves with the blocks -- defects
    // g (B, m) [the first H*NX rows of a problem] and compact tiles (B, H, NX, NIN), as the row kernels write them
    void* g_out;
    void* tiles_out;
    int ident;              // 1: Discret (Phi = x + f), 0: Unity
};

template <typename T, int NT, int NTHREADS, int NCOL>
struct HfxStage {
    static constexpr int ITEMS = (NCOL * NT * 16 + NTHREADS - 1) / NTHREADS;
    T v[ITEMS];
};

// inputs of a pass: item = (column, row); columns = NIN network inputs then the NX multipliers of the row's defects
template <typename T, int WP, int NT, int NX, int NU>
__device__ __forceinline__ void hfx_stage_load(const HfxArgs& a, int t0, int tid, HfxStage<T, NT, (WP / 16) * 64, NX + NU + NX>& sr) {
    constexpr int NIN = NX + NU, NCOL = NIN + NX, ROWS = NT * 16, NTHREADS = (WP / 16) * 64;
    const T* __restrict__ Z = static_cast<const T*>(a.Z);
    const T* __restrict__ X0 = static_cast<const T*>(a.X0);
    const T* __restrict__ lam = static_cast<const T*>(a.lambda);
#pragma unroll
    for (int it = 0; it < HfxStage<T, NT, NTHREADS, NCOL>::ITEMS; ++it) {
        const int item = tid + it * NTHREADS;
        const int col = item / ROWS, idx = item - col * ROWS;      // compile-time divisor
        T v = T(0);
        const unsigned r = (unsigned)t0 * 16u + (unsigned)idx;
        if (col < NCOL && r < a.R) {
            if (a.xi_direct) {
                v = col < NIN ? static_cast<const T*>(a.xi_direct)[(size_t)r * a.xi_stride + col]
                              : static_cast<const T*>(a.lam_direct)[(size_t)r * NX + (col - NIN)];
            } else {
                const unsigned b = a.invH ? __umulhi(r, a.invH) : r;
                const int t = (int)(r - b * (unsigned)a.H);
                const T* z = Z + (size_t)b * a.n;
                if (col < NX) v = (t == 0) ? X0[(size_t)b * NX + col] : z[(t - 1) * NX + col];
                else if (col < NIN) v = z[a.H * NX + t * NU + (col - NX)];
                else v = lam[(size_t)b * a.m + t * NX + (col - NIN)];
            }
        }
        sr.v[it] = v;
    }
}

template <typename T, int WP, int NH, int NTL, int NT, int NX, int NU, int TG>
__device__ __forceinline__ void hfx_stage_store(T* in, int tid, const HfxStage<T, NT, (WP / 16) * 64, NX + NU + NX>& sr) {
    constexpr int NIN = NX + NU, NCOL = NIN + NX, ROWS = NT * 16, NTHREADS = (WP / 16) * 64;
    using L = HfxLayout<T, WP, NH, NTL, NX, NU, TG>;
#pragma unroll
    for (int it = 0; it < HfxStage<T, NT, NTHREADS, NCOL>::ITEMS; ++it) {
        const int item = tid + it * NTHREADS;
        const int col = item / ROWS, idx = item - col * ROWS;
        if (col < NCOL) {
            T* tile = in + (idx >> 4) * L::IN_TILE;
            if (col < NIN) tile[(idx & 15) * NIN + col] = sr.v[it];
            else tile[16 * NIN + (idx & 15) * NX + (col - NIN)] = sr.v[it];
        }
    }
}

// One pass over NTc (<= NT) tiles starting at tile t0, inputs in `in`.
// RWB: the backward slices (W_l as an A operand) are fetched at the start of every pass instead of living in registers
// for the whole kernel -- they are read by the base reverse sweep only, and at widths where a pass is tens of thousands
// of cycles (3x128) their registers are what the tangent / contraction phase needs.
// EV: the pass also leaves the rows' first-order evaluation (HfxArgs::g_out / tiles_out): the outputs W_L a + b_L from the
// activations the forward sweep holds, the tile W_L (s' * P_{NH-1}) from the last layer's tangents -- NX * (NIN + 1) more
// K-split quantities per row next to the NPAIR pair sums, no further matrix product.
template <typename T, int WP, int NH, int NT, int NX, int NU, int NTc, int ACT, int TG, bool RWB, bool EV = false>
__device__ __forceinline__ void hfx_pass(const HfxArgs& a, T* lds, const T (&wf)[NH > 1 ? NH - 1 : 1][(WP / 16) * 4],
                                         const T (&wb_res)[NH > 1 ? NH - 1 : 1][(WP / 16) * 4], const T* in, int t0, int tid,
                                         int& xsel, const HfxStage<T, NT, (WP / 16) * 64, NX + NU + NX>& nxt, bool has_next,
                                         T* in_next) {
    using Ops = MfmaOps<T>;
    using A = Act<T, ACT>;
    static_assert(ACT != NEMPC_ACT_RUNTIME, "the compiled-shape kernels take a compile-time activation");
    using V4 = typename Ops::V4;
    using L = HfxLayout<T, WP, NH, NT, NX, NU, TG, EV>;
    constexpr int MT = WP / 16, NTHREADS = MT * 64, NIN = NX + NU, KS = L::KS, NPAIR = L::NPAIR, NVT = L::NVT;
    static_assert(!EV || NH > 1, "the evaluation outputs read the last hidden layer's tangents");
    const int lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    T ev[NTc][EV ? NX * NIN + NX : 1];      // EV: tile partials [k][p], then output partials [k]

    T wb[NH > 1 ? NH - 1 : 1][MT * 4];
    if constexpr (RWB) {
        constexpr int VEC = 16 / (int)sizeof(T);
        typedef T vecT __attribute__((ext_vector_type(VEC)));
        const vecT* __restrict__ ws = static_cast<const vecT*>(a.wslice) + (size_t)w * a.nload * 64 + lane;
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            constexpr int PER = MT * 4 / VEC;                         // 16-byte vectors per fragment set
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const vecT v = ws[((l - 1) * 2 * PER + PER + k) * 64];
#pragma unroll
                for (int e = 0; e < VEC; ++e) wb[l - 1][k * VEC + e] = v[e];
            }
        }
    } else {
#pragma unroll
        for (int l = 1; l < NH; ++l)
#pragma unroll
            for (int i = 0; i < MT * 4; ++i) wb[l - 1][i] = wb_res[l - 1][i];
    }
    V4 s[NH][NTc];          // a_l first, then s'(z_l)
    // ---- layer 0, this wave's feature block
    {
        const T* bias = lds + L::BIAS + w * 16;
        V4 b0;
#pragma unroll
        for (int r = 0; r < 4; ++r) b0[r] = bias[r * 4 + q];
#pragma unroll
        for (int j = 0; j < NTc; ++j) s[0][j] = b0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const T wfrag = lds[L::W0F + (ks * MT + w) * 64 + lane];
            const int d = 4 * ks + q;
#pragma unroll
            for (int j = 0; j < NTc; ++j) {
                const T v = d < NIN ? in[j * L::IN_TILE + c * NIN + d] : T(0);
                s[0][j] = Ops::mma(wfrag, v, s[0][j]);
            }
        }
#pragma unroll
        for (int j = 0; j < NTc; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[0][j][r] = A::f(s[0][j][r]);
    }
    // ---- hidden-to-hidden layers through the double-buffered exchange area
#pragma unroll
    for (int l = 1; l < NH; ++l) {
        T* X = lds + L::X + (xsel & 1) * L::XH;
        ++xsel;
#pragma unroll
        for (int j = 0; j < NTc; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[((j * MT + w) * 4 + r) * 64 + lane] = s[l - 1][j][r];
        lds_barrier();
        const T* bias = lds + L::BIAS + l * MT * 16 + w * 16;
        V4 b0;
#pragma unroll
        for (int r = 0; r < 4; ++r) b0[r] = bias[r * 4 + q];
#pragma unroll
        for (int j = 0; j < NTc; ++j) s[l][j] = b0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < NTc; ++j)
                    s[l][j] = Ops::mma(wf[l - 1][mt * 4 + r], X[((j * MT + mt) * 4 + r) * 64 + lane], s[l][j]);
#pragma unroll
        for (int j = 0; j < NTc; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[l][j][r] = A::f(s[l][j][r]);
    }
    // ---- base reverse sweep: delta_l = d(lambda . f)/d a_l; the top one is W_L lambda, on the vector unit (NX terms)
    V4 wgt[NH][NTc];        // delta_l * s''(z_l)
    {
        V4 dl[NTc];
#pragma unroll
        for (int j = 0; j < NTc; ++j) dl[j] = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const T* seed = lds + L::SEED + k * MT * 16 + w * 16;
            V4 wl;
#pragma unroll
            for (int r = 0; r < 4; ++r) wl[r] = seed[r * 4 + q];
#pragma unroll
            for (int j = 0; j < NTc; ++j) {
                const T lm = in[j * L::IN_TILE + 16 * NIN + c * NX + k];
#pragma unroll
                for (int r = 0; r < 4; ++r) dl[j][r] = fma(wl[r], lm, dl[j][r]);
                if constexpr (EV) {      // output k over this lane's features (s still holds the activations)
                    T v = wl[0] * s[NH - 1][j][0];
#pragma unroll
                    for (int r = 1; r < 4; ++r) v = fma(wl[r], s[NH - 1][j][r], v);
                    ev[j][NX * NIN + k] = v;
                }
            }
        }
#pragma unroll
        for (int l = NH - 1; l >= 0; --l) {
#pragma unroll
            for (int j = 0; j < NTc; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const T av = s[l][j][r];
                    const T s1 = A::d1(av);
                    wgt[l][j][r] = dl[j][r] * (A::r2(av) * s1);
                    s[l][j][r] = s1;
                }
            if (l > 0) {
                T* X = lds + L::X + (xsel & 1) * L::XH;
                ++xsel;
#pragma unroll
                for (int j = 0; j < NTc; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[((j * MT + w) * 4 + r) * 64 + lane] = dl[j][r] * s[l][j][r];
                lds_barrier();
#pragma unroll
                for (int j = 0; j < NTc; ++j) dl[j] = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < NTc; ++j)
                            dl[j] = Ops::mma(wb[l - 1][mt * 4 + r], X[((j * MT + mt) * 4 + r) * 64 + lane], dl[j]);
            }
        }
    }
    // ---- contraction, layer 0: P_0 = the first layer's rows (constant), their pair products are a table
    T acc[NTc][NPAIR];
#pragma unroll
    for (int pq = 0; pq < NPAIR; ++pq) {
        const T* pp = lds + L::PP + pq * MT * 16 + w * 16;
        V4 ppv;
#pragma unroll
        for (int r = 0; r < 4; ++r) ppv[r] = pp[r * 4 + q];
#pragma unroll
        for (int j = 0; j < NTc; ++j) {
            T v = wgt[0][j][0] * ppv[0];
#pragma unroll
            for (int r = 1; r < 4; ++r) v = fma(wgt[0][j][r], ppv[r], v);
            acc[j][pq] = v;
        }
    }
    // ---- tangent-forward sweeps of all NIN directions together, contracted layer by layer
    if constexpr (NH > 1) {
        V4 tg[NIN][NTc];    // D_{l-1} = s'(z_{l-1}) * P_{l-1}: what the next layer's product reads
#pragma unroll
        for (int p = 0; p < NIN; ++p) {
            const T* p0 = lds + L::P0 + p * MT * 16 + w * 16;
            V4 w0;
#pragma unroll
            for (int r = 0; r < 4; ++r) w0[r] = p0[r * 4 + q];
#pragma unroll
            for (int j = 0; j < NTc; ++j) tg[p][j] = s[0][j] * w0;
        }
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            V4 P[NIN][NTc];
            // TG directions per exchange (all of them when they fit the exchange area)
#pragma unroll
            for (int g0 = 0; g0 < NIN; g0 += TG) {
                T* X = lds + L::X + (xsel & 1) * L::XH;
                ++xsel;
#pragma unroll
                for (int g = 0; g < TG; ++g)
#pragma unroll
                    for (int j = 0; j < NTc; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (g0 + g < NIN) X[(((g * NTc + j) * MT + w) * 4 + r) * 64 + lane] = tg[g0 + g < NIN ? g0 + g : 0][j][r];
                lds_barrier();
#pragma unroll
                for (int g = 0; g < TG; ++g)
#pragma unroll
                    for (int j = 0; j < NTc; ++j)
                        if (g0 + g < NIN) P[g0 + g < NIN ? g0 + g : 0][j] = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int g = 0; g < TG; ++g)
#pragma unroll
                            for (int j = 0; j < NTc; ++j)
                                if (g0 + g < NIN)
                                    P[g0 + g < NIN ? g0 + g : 0][j] = Ops::mma(wf[l - 1][mt * 4 + r],
                                                                               X[(((g * NTc + j) * MT + mt) * 4 + r) * 64 + lane],
                                                                               P[g0 + g < NIN ? g0 + g : 0][j]);
            }
            // acc[pq] += sum_r (delta s'')_r P[p]_r P[q]_r   over this lane's four features
#pragma unroll
            for (int j = 0; j < NTc; ++j) {
                int pq = 0;
#pragma unroll
                for (int p = 0; p < NIN; ++p) {
                    const V4 wp = wgt[l][j] * P[p][j];
#pragma unroll
                    for (int qq = 0; qq <= p; ++qq, ++pq)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[j][pq] = fma(wp[r], P[qq][j][r], acc[j][pq]);
                }
            }
            if (l < NH - 1) {
#pragma unroll
                for (int p = 0; p < NIN; ++p)
#pragma unroll
                    for (int j = 0; j < NTc; ++j) tg[p][j] = s[l][j] * P[p][j];
            }
            if constexpr (EV) {
                if (l == NH - 1) {       // tile[k][p] = sum_features W_L[k] * s'(z) * P[p]
#pragma unroll
                    for (int k = 0; k < NX; ++k) {
                        const T* seed = lds + L::SEED + k * MT * 16 + w * 16;
                        V4 wl;
#pragma unroll
                        for (int r = 0; r < 4; ++r) wl[r] = seed[r * 4 + q];
#pragma unroll
                        for (int j = 0; j < NTc; ++j) {
                            const V4 ws1 = wl * s[l][j];
#pragma unroll
                            for (int p = 0; p < NIN; ++p) {
                                T v = ws1[0] * P[p][j][0];
#pragma unroll
                                for (int r = 1; r < 4; ++r) v = fma(ws1[r], P[p][j][r], v);
                                ev[j][k * NIN + p] = v;
                            }
                        }
                    }
                }
            }
        }
    }
    // ---- quad sums -> this wave's partials [j][pq][16 rows]
    {
        T sv[NTc * NVT];
#pragma unroll
        for (int j = 0; j < NTc; ++j) {
#pragma unroll
            for (int pq = 0; pq < NPAIR; ++pq) sv[j * NVT + pq] = acc[j][pq];
            if constexpr (EV)
#pragma unroll
                for (int e = 0; e < NVT - NPAIR; ++e) sv[j * NVT + NPAIR + e] = ev[j][e];
        }
        fx_rowsums_store<T, NTc * NVT>(sv, lds + L::PART + w * (NT * NVT) * 16, lane);
    }
    lds_barrier();
    if (has_next) hfx_stage_store<T, WP, NH, NT, NT, NX, NU, TG>(in_next, tid, nxt);

    // ---- outputs: the sum over the MT waves is taken here, in wave order
    constexpr int BSZ = NIN * NIN;
    const T* PART = lds + L::PART;
    T* const blocks = static_cast<T*>(a.blocks);
    T* const hv = static_cast<T*>(a.hvals);
#pragma unroll
    for (int it = 0; it < (NTc * 16 * BSZ + NTHREADS - 1) / NTHREADS; ++it) {
        const int item = tid + it * NTHREADS;
        const int idx = item / BSZ, e = item - idx * BSZ;        // compile-time divisors
        const unsigned r = (unsigned)t0 * 16u + (unsigned)idx;
        if (item < NTc * 16 * BSZ && r < a.R) {
            const int j = idx >> 4, cc = idx & 15;
            const int a1 = e / NIN, a2 = e - a1 * NIN;
            const int hi = a1 > a2 ? a1 : a2, lo = a1 > a2 ? a2 : a1;
            const int pq = hi * (hi + 1) / 2 + lo;
            T v = T(0);
#pragma unroll
            for (int ww = 0; ww < MT; ++ww) v += PART[(ww * (NT * NVT) + j * NVT + pq) * 16 + cc];
            if (hv) {
                // fused assembly (nempc_hess asked for the tril values only): assemble_hess_kernel's arithmetic, here
                const T* sg = static_cast<const T*>(a.sigma);
                const T* oc = static_cast<const T*>(a.objc);
                const unsigned b = a.invH ? __umulhi(r, a.invH) : r;
                const int t = (int)(r - b * (unsigned)a.H);
                const int ent = a.smap[t * BSZ + e];
                if (ent >= 0) hv[(size_t)b * a.nnz + ent] = sg[b] * oc[ent] + v;
                if (t == a.H - 1 && e < a.n_orph) {
                    const int oe = a.smap[a.H * BSZ + e];
                    hv[(size_t)b * a.nnz + oe] = sg[b] * oc[oe];
                }
            } else {
                blocks[(size_t)t0 * (16 * BSZ) + item] = v;
            }
        }
    }
    if constexpr (EV) {
        // the rows' evaluation: tile entries (+ the identity of Discret), then the defects Phi - x_t
        constexpr int ESZ = NX * NIN + NX;
        T* const tl = static_cast<T*>(a.tiles_out);
        T* const go = static_cast<T*>(a.g_out);
        const T* __restrict__ Zg = static_cast<const T*>(a.Z);
#pragma unroll
        for (int it = 0; it < (NTc * 16 * ESZ + NTHREADS - 1) / NTHREADS; ++it) {
            const int item = tid + it * NTHREADS;
            const int idx = item / ESZ, e = item - idx * ESZ;
            const unsigned r = (unsigned)t0 * 16u + (unsigned)idx;
            if (item < NTc * 16 * ESZ && r < a.R) {
                const int j = idx >> 4, cc = idx & 15;
                T v = T(0);
#pragma unroll
                for (int ww = 0; ww < MT; ++ww) v += PART[(ww * (NT * NVT) + j * NVT + NPAIR + e) * 16 + cc];
                if (e < NX * NIN) {
                    const int k = e / NIN, pcol = e - k * NIN;
                    if (a.ident && pcol == k) v += T(1);
                    tl[(size_t)r * (NX * NIN) + e] = v;
                } else {
                    const int k = e - NX * NIN;
                    const unsigned b = a.invH ? __umulhi(r, a.invH) : r;
                    const int t = (int)(r - b * (unsigned)a.H);
                    v += lds[L::BIASL + (sizeof(T) == 8 ? k : (k & 3) * 4 + (k >> 2))];     // (packed in the output tile's lane order)
                    if (a.ident) v += in[j * L::IN_TILE + cc * NIN + k];
                    go[(size_t)b * a.m + t * NX + k] = v - Zg[(size_t)b * a.n + t * NX + k];
                }
            }
        }
    }
    lds_barrier();
}

template <typename T, int WP, int NH, int NT, int NX, int NU, int ACT, int TG = NX + NU, bool RWB = false, bool EV = false>
__global__ __launch_bounds__((WP / 16) * 64, 2) void rowhess_coopfx_kernel(HfxArgs a) {
    using L = HfxLayout<T, WP, NH, NT, NX, NU, TG, EV>;
    constexpr int MT = WP / 16;
    constexpr int NTHREADS = MT * 64;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int NIN = NX + NU, NCOL = NIN + NX;
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;

    const int t_begin = blockIdx.x * a.tiles_per_wg + ((int)blockIdx.x < a.tiles_rem ? (int)blockIdx.x : a.tiles_rem);
    const int t_end = t_begin + a.tiles_per_wg + ((int)blockIdx.x < a.tiles_rem ? 1 : 0);

    // ---- every global load of the prologue is issued before anything waits: inputs, small tables, weight slices
    HfxStage<T, NT, NTHREADS, NCOL> sr;
    int t0 = t_begin;
    hfx_stage_load<T, WP, NT, NX, NU>(a, t0, tid, sr);
    constexpr int SMALL_VECS = (L::SMALL_END + VEC - 1) / VEC;
    constexpr int SMALL_PER_THREAD = (SMALL_VECS + NTHREADS - 1) / NTHREADS;
    vecT sm[SMALL_PER_THREAD];
    {
        const vecT* __restrict__ gs = static_cast<const vecT*>(a.small);
#pragma unroll
        for (int u = 0; u < SMALL_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
            if (idx < SMALL_VECS) sm[u] = gs[idx];
        }
    }
    // this wave's hidden-to-hidden slices (the leading part of its packed slice: wf, wb per hidden layer)
    constexpr int NFRAG = (NH - 1) * 2 * MT * 4;
    constexpr int NLOAD = (NFRAG + VEC - 1) / VEC;
    vecT wv[NLOAD > 0 ? NLOAD : 1];
    {
        const vecT* __restrict__ ws = static_cast<const vecT*>(a.wslice) + (size_t)w * a.nload * 64 + lane;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) wv[k] = ws[k * 64];
    }
    {
        vecT* ls = reinterpret_cast<vecT*>(lds + L::W0F);
#pragma unroll
        for (int u = 0; u < SMALL_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
            if (idx < SMALL_VECS) ls[idx] = sm[u];
        }
    }
    T wf[NH > 1 ? NH - 1 : 1][MT * 4], wb[NH > 1 ? NH - 1 : 1][MT * 4];
    {
        int f = 0;
#pragma unroll
        for (int l = 1; l < NH; ++l) {
#pragma unroll
            for (int i = 0; i < MT * 4; ++i, ++f) wf[l - 1][i] = wv[f / VEC][f % VEC];
#pragma unroll
            for (int i = 0; i < MT * 4; ++i, ++f) wb[l - 1][i] = RWB ? T(0) : wv[f / VEC][f % VEC];
        }
    }
    T* const in_base = lds + L::IN;
    hfx_stage_store<T, WP, NH, NT, NT, NX, NU, TG>(in_base, tid, sr);
    lds_barrier();
    // pair products of the first layer's rows, per feature: PP[pq][i] = P0[p][i] * P0[q][i]
    for (int e = tid; e < L::NPAIR * MT * 16; e += NTHREADS) {
        const int pq = e / (MT * 16), i = e - pq * (MT * 16);
        int p = 0;
        while ((p + 1) * (p + 2) / 2 <= pq) ++p;
        const int qq = pq - p * (p + 1) / 2;
        lds[L::PP + e] = lds[L::P0 + p * MT * 16 + i] * lds[L::P0 + qq * MT * 16 + i];
    }

    int parity = 0, xsel = 0;
    while (t0 < t_end) {
        const int t_cur = t0;
        const int n_cur = t_end - t0 < NT ? t_end - t0 : NT;
        t0 += n_cur;
        const bool more = t0 < t_end;
        if (more) hfx_stage_load<T, WP, NT, NX, NU>(a, t0, tid, sr);        // next pass's inputs, under this pass
        lds_barrier();
        const T* in = in_base + parity * L::IN_SZ;
        T* const in_next = in_base + (parity ^ 1) * L::IN_SZ;
        if (n_cur == 1) hfx_pass<T, WP, NH, NT, NX, NU, 1, ACT, TG, RWB, EV>(a, lds, wf, wb, in, t_cur, tid, xsel, sr, more, in_next);
        if constexpr (NT >= 2) { if (n_cur == 2) hfx_pass<T, WP, NH, NT, NX, NU, 2, ACT, TG, RWB, EV>(a, lds, wf, wb, in, t_cur, tid, xsel, sr, more, in_next); }
        parity ^= 1;
    }
}

}  // namespace nempc
