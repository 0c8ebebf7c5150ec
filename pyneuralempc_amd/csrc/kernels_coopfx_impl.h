// Fixed-shape cooperative row kernel (gfx950): the CU-cooperative matrix-core kernel of kernels_coop_impl.h with the
// problem shape (nx, nu) as template parameters, for plain models (window 1, no extra inputs) under the Discret / Unity
// transcriptions -- BASELINE configs[1] and configs[4] (2 states, 1 control, MLP 2x64).
//
// Why a second instantiation family.  Per-wave stamps of the generic kernel (tools/diag_stamps.py) put ~55 % of a pass
// in work that is not arithmetic: index arithmetic on runtime dims, scalar-register spills (100 SGPRs, 88 v_readlane per
// pass), a separate reduction phase with its two barriers and a 4.5k-cycle epilogue for three stores per thread.  Two
// waves share a SIMD; when both are in such phases the double-precision pipe idles (measured busy share 52 %).  With the
// dims fixed every index folds to a constant, the LDS carve-up is a compile-time table, the K-split partials are summed
// by the epilogue itself (no reduction phase), and row -> (problem, step) is two multiply-highs instead of an LDS table.
//
// Same math, same operand-layout trick and the same packed weights as the generic kernel (see kernels_mfma_impl.h and
// kernels_coop_impl.h); results agree with it to rounding (the K-split partial sums are added in the same order).
#pragma once

#include "kernels_coop_impl.h"
#include "kernels_obj_impl.h"

namespace nempc {

template <typename T, int WP, int NH, int TPW, int NX, int NU>
struct FxLayout {   // element offsets inside dynamic LDS, all compile-time
    static constexpr int MT = WP / 16;
    static constexpr int NIN = NX + NU;
    static constexpr int KS = (NIN + 3) / 4;
    static constexpr int NR = sizeof(T) == 8 ? (NIN + 3) / 4 : 4;   // accumulator registers holding the NIN input rows
    static constexpr int NRO = sizeof(T) == 8 ? (NX + 3) / 4 : 4;   // ... the NX output rows
    static constexpr int JROW = NX * NIN;
    // small tables, copied flat from off.fx_small: [w0f | seed | bias_0..NH-1 | biasL | p0tab]
    static constexpr int W0F = 0;
    static constexpr int SEED = W0F + KS * MT * 64;
    static constexpr int BIAS = SEED + NX * MT * 16;
    static constexpr int BIASL = BIAS + NH * MT * 16;
    static constexpr int P0 = BIASL + 16;                            // first-layer rows per lane: [d][MT*16]
    static constexpr int SMALL_END = P0 + NIN * MT * 16;
    // exchange buffer: two halves of TPW activation sets (one cotangent per sweep)
    static constexpr int XH = TPW * MT * 256;
    static constexpr int X = (SMALL_END + 15) & ~15;
    // fused evaluation, epilogue: the exchange area is reused as [row buffer RB_CAP | tiles of the pass TS_SZ]
    static constexpr int TS_SZ = (TPW * 16 * JROW + 15) & ~15;
    static constexpr int RB_CAP = 2 * XH - TS_SZ;
    // K-split partials, one value per (tile, wave, quantity, row): network outputs PF[j][w][k][16], Jacobian rows
    // PJ[k][j][w][d][16]
    static constexpr int PART = X + 2 * XH;
    static constexpr int PF_SZ = TPW * MT * NX * 16;
    static constexpr int PART_SZ = PF_SZ + NX * TPW * MT * NIN * 16;
    // inputs, double-buffered: per tile xi[16][NIN] then xt[16][NX]
    static constexpr int IN_TILE = 16 * (NIN + NX);
    static constexpr int IN = PART + PART_SZ;
    static constexpr int IN_SZ = (TPW * IN_TILE + 15) & ~15;
    static constexpr int TOTAL = IN + 2 * IN_SZ;
};

struct FxArgs {   // host-prepared; the fields the first loads need come first
    const void* Z;
    const void* X0;
    const void* small;      // blob + off.fx_small
    const void* wslice;     // blob + off.coop_slices
    int tiles_per_wg, tiles_rem;
    unsigned R;             // rows = B*H
    unsigned invH;          // ceil(2^32 / H), 0 for H == 1
    int H, n, m;
    int ident;              // 1: Discret (Phi = x + f), 0: Unity
    int box;
    int small_vecs;
    void* g;
    void* tiles;            // may be null in the fused evaluation (nobody asked for the compact tiles)
    // ---- fused evaluation (FUSE instantiation): the dense Jacobian rows and the objective of the workgroup's
    //      problems leave from the same launch; the separate assembly launch and its boundary disappear
    void* jac;              // (B, m, n) dense, 16-byte aligned, n a multiple of the vector width
    void* f;                // (B) or null
    void* grad;             // (B, n) or null
    const void* P;          // objective table (Handle::d_obj), copied to LDS behind the layout
    int p_elems;
    int rb_rows;            // dense rows per chunk of the LDS row buffer (FxLayout::RB_CAP / n, at least 1)
    unsigned inv_nvec;      // ceil(2^32 / (n / VEC)): flat vector index -> row by multiply-high
    ObjOffsets oo;
};

template <typename T, int WP, int NH, int TPW, int NX, int NU>
struct FxCtx {
    using L = FxLayout<T, WP, NH, TPW, NX, NU>;
    T* lds;
    const T* __restrict__ Z;
    const T* __restrict__ X0;
    T* __restrict__ gout;
    T* __restrict__ tiles;
    unsigned R, invH;
    int H, n, m, ident, box;
    T* __restrict__ jac;
    int rb_rows;            // dense rows the LDS row buffer holds (RB_CAP / n)
    unsigned inv_nvec;      // ceil(2^32 / (n / VEC)), 0 for one vector per row
};

// inputs of a pass: item = (column, row); columns = NIN network inputs then the NX current states x_t
template <typename T, int NT, int NTHREADS, int NCOL>
struct FxStage {
    static constexpr int ROWS = NT * 16;
    static constexpr int ITEMS = (NCOL * ROWS + NTHREADS - 1) / NTHREADS;
    T v[ITEMS];
};

template <typename T, int WP, int NH, int TPW, int NX, int NU, int NT>
__device__ __forceinline__ void fx_stage_load(const FxCtx<T, WP, NH, TPW, NX, NU>& cx, int t0, int tid,
                                              FxStage<T, NT, (WP / 16) * 64, NX + NU + NX>& sr) {
    constexpr int NIN = NX + NU, NCOL = NIN + NX, ROWS = NT * 16, NTHREADS = (WP / 16) * 64;
#pragma unroll
    for (int it = 0; it < FxStage<T, NT, NTHREADS, NCOL>::ITEMS; ++it) {
        const int item = tid + it * NTHREADS;
        const int col = item / ROWS, idx = item - col * ROWS;      // compile-time divisor
        T v = T(0);
        const unsigned r = (unsigned)t0 * 16u + (unsigned)idx;
        if (col < NCOL && r < cx.R) {
            const unsigned b = cx.invH ? __umulhi(r, cx.invH) : r;
            const int t = (int)(r - b * (unsigned)cx.H);
            const T* z = cx.Z + (size_t)b * cx.n;
            if (col < NX) v = (t == 0) ? cx.X0[(size_t)b * NX + col] : z[(t - 1) * NX + col];
            else if (col < NIN) v = z[cx.H * NX + t * NU + (col - NX)];
            else v = z[t * NX + (col - NIN)];
        }
        sr.v[it] = v;
    }
}

template <typename T, int WP, int NH, int TPW, int NX, int NU, int NT>
__device__ __forceinline__ void fx_stage_store(T* in, int tid, const FxStage<T, NT, (WP / 16) * 64, NX + NU + NX>& sr) {
    constexpr int NIN = NX + NU, NCOL = NIN + NX, ROWS = NT * 16, NTHREADS = (WP / 16) * 64;
    using L = FxLayout<T, WP, NH, TPW, NX, NU>;
#pragma unroll
    for (int it = 0; it < FxStage<T, NT, NTHREADS, NCOL>::ITEMS; ++it) {
        const int item = tid + it * NTHREADS;
        const int col = item / ROWS, idx = item - col * ROWS;
        if (col < NCOL) {
            T* tile = in + (idx >> 4) * L::IN_TILE;
            if (col < NIN) tile[(idx & 15) * NIN + col] = sr.v[it];
            else tile[16 * NIN + (idx & 15) * NX + (col - NIN)] = sr.v[it];
        }
    }
}

// One pass over NT tiles starting at tile t0, inputs in `in`.
// Sum of a per-lane value over the four 16-lane rows of the wave (the K index q of the accumulator layout), left in every
// lane: gfx950's v_permlane16_swap / v_permlane32_swap exchange rows inside the vector unit (no LDS round trip); with both
// operands equal, swap + add folds rows {0,1},{2,3} and then the two halves.  Fixed order ((r0+r1)+(r2+r3)).
__device__ __forceinline__ double fx_qsum(double s) {
    unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
    auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    s = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);
    lo = (unsigned)__double2loint(s); hi = (unsigned)__double2hiint(s);
    auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
}
__device__ __forceinline__ float fx_qsum(float s) {
    unsigned v = __float_as_uint(s);
    auto a16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    s = __uint_as_float(a16[0]) + __uint_as_float(a16[1]);
    v = __float_as_uint(s);
    auto a32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return __uint_as_float(a32[0]) + __uint_as_float(a32[1]);
}

// `nxt` / `in_next` (when has_next): the NEXT pass's inputs, already in registers; they go to the other input buffer
// BEFORE this pass's global stores are issued -- vmcnt counts stores too and retires in order, so a wait for those loads
// placed after the stores would sit out the stores' acknowledgement (with the dense rows fused in: the whole HBM time).
template <typename T, int WP, int NH, int TPW, int NX, int NU, int NT, bool FUSE>
__device__ __forceinline__ void fx_pass(const FxCtx<T, WP, NH, TPW, NX, NU>& cx, const CoopWeights<T, WP, NH>& W,
                                        const T* in, int t0, int tid, int& xsel,
                                        const FxStage<T, TPW, (WP / 16) * 64, NX + NU + NX>& nxt, bool has_next, T* in_next) {
    using Ops = MfmaOps<T>;
    using V4 = typename Ops::V4;
    using L = FxLayout<T, WP, NH, TPW, NX, NU>;
    constexpr int MT = WP / 16, NTHREADS = MT * 64, NIN = NX + NU, KS = L::KS, JROW = L::JROW;
    const int lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    T* const lds = cx.lds;
    T* const PART = lds + L::PART;

    V4 a[NH][NT];
    // ---- layer 0, this wave's feature block
    {
        const T* bias = lds + L::BIAS + w * 16;
        V4 b0;
#pragma unroll
        for (int r = 0; r < 4; ++r) b0[r] = bias[r * 4 + q];
#pragma unroll
        for (int j = 0; j < NT; ++j) a[0][j] = b0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const T wfrag = lds[L::W0F + (ks * MT + w) * 64 + lane];
            const int d = 4 * ks + q;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const T v = d < NIN ? in[j * L::IN_TILE + c * NIN + d] : T(0);
                a[0][j] = Ops::mma(wfrag, v, a[0][j]);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[0][j][r] = Ops::tanh_(a[0][j][r]);
    }
    // ---- hidden-to-hidden layers through the double-buffered exchange area (see kernels_coop_impl.h)
#pragma unroll
    for (int l = 1; l < NH; ++l) {
        T* X = lds + L::X + (xsel & 1) * L::XH;
        ++xsel;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[((j * MT + w) * 4 + r) * 64 + lane] = a[l - 1][j][r];
        lds_barrier();
        const T* bias = lds + L::BIAS + l * MT * 16 + w * 16;
        V4 b0;
#pragma unroll
        for (int r = 0; r < 4; ++r) b0[r] = bias[r * 4 + q];
#pragma unroll
        for (int j = 0; j < NT; ++j) a[l][j] = b0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    a[l][j] = Ops::mma(W.wf[l - 1][mt * 4 + r], X[((j * MT + mt) * 4 + r) * 64 + lane], a[l][j]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[l][j][r] = Ops::tanh_(a[l][j][r]);
    }
    // ---- network output: K-split partial over this wave's 16 hidden units.  The two skinny layers (NX outputs here,
    //      NIN inputs at the end of the reverse sweep) used to be MFMAs that compute 16 output rows for the 2-3 that
    //      exist -- 12 of a tile-wave's 61 matrix instructions, on the pipe that bounds the kernel.  On the vector unit
    //      they are 4 FMAs per value and lane plus the row sum above: a fifth of the double-precision pipe time.
    V4 wl[NX];      // W_L[f(q,r)][k] for this lane's four features (the reverse sweep's seed is the same numbers)
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const T* seed = lds + L::SEED + k * MT * 16 + w * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) wl[k][r] = seed[r * 4 + q];
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            T sv = a[NH - 1][j][0] * wl[k][0];
#pragma unroll
            for (int r = 1; r < 4; ++r) sv = fma(a[NH - 1][j][r], wl[k][r], sv);
            sv = fx_qsum(sv);
            if (q == 0) PART[((j * MT + w) * NX + k) * 16 + c] = sv;
        }
#pragma unroll
    for (int l = 0; l < NH; ++l)
#pragma unroll
        for (int j = 0; j < NT; ++j) a[l][j] = T(1) - a[l][j] * a[l][j];

    // ---- reverse sweep, one cotangent (network output) at a time
    T* const PJ = PART + L::PF_SZ;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        V4 cv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) cv[j] = wl[k] * a[NH - 1][j];
#pragma unroll
        for (int l = NH - 1; l >= 1; --l) {
            T* X = lds + L::X + (xsel & 1) * L::XH;
            ++xsel;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[((j * MT + w) * 4 + r) * 64 + lane] = cv[j][r];
            lds_barrier();
            V4 cn[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) cn[j] = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        cn[j] = Ops::mma(W.wb[l - 1][mt * 4 + r], X[((j * MT + mt) * 4 + r) * 64 + lane], cn[j]);
#pragma unroll
            for (int j = 0; j < NT; ++j) cv[j] = cn[j] * a[l - 1][j];
        }
        // last reverse step onto the NIN inputs, on the vector unit: J[k][d] partial = sum_r cv_r * W0[d][f(q,r)]
#pragma unroll
        for (int d = 0; d < NIN; ++d) {
            const T* p0 = lds + L::P0 + d * MT * 16 + w * 16;
            V4 w0;
#pragma unroll
            for (int r = 0; r < 4; ++r) w0[r] = p0[r * 4 + q];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                T sv = cv[j][0] * w0[0];
#pragma unroll
                for (int r = 1; r < 4; ++r) sv = fma(cv[j][r], w0[r], sv);
                sv = fx_qsum(sv);
                if (q == 0) PJ[(((k * TPW + j) * MT + w) * NIN + d) * 16 + c] = sv;
            }
        }
    }
    lds_barrier();
    if (has_next) fx_stage_store<T, WP, NH, TPW, NX, NU, TPW>(in_next, tid, nxt);

    // ---- outputs straight from the partials (no reduction phase): the sum over the MT waves is taken here, in wave
    //      order like the generic kernel's reduction
    // tiles: the pass's NT*16 rows are contiguous in memory -> lanes run over the flat element index (coalesced).
    // The fused evaluation also keeps them in LDS (TS, at the end of the now idle exchange area) for the dense rows below.
    T* const TS = lds + L::X + L::RB_CAP;
#pragma unroll
    for (int it = 0; it < (NT * 16 * JROW + NTHREADS - 1) / NTHREADS; ++it) {
        const int item = tid + it * NTHREADS;
        const int idx = item / JROW, kd = item - idx * JROW;     // compile-time divisors
        const int k = kd / NIN, d = kd - k * NIN;
        const unsigned r = (unsigned)t0 * 16u + (unsigned)idx;
        if (item < NT * 16 * JROW && r < cx.R) {
            const int j = idx >> 4, cc = idx & 15;
            T v = T(0);
#pragma unroll
            for (int ww = 0; ww < MT; ++ww) v += PJ[(((k * TPW + j) * MT + ww) * NIN + d) * 16 + cc];
            if (cx.ident && d == k) v += T(1);
            if (!FUSE || cx.tiles) cx.tiles[(size_t)t0 * (16 * JROW) + item] = v;
            if (FUSE) TS[item] = v;
        }
    }
    // defects: lanes run over (row, state) with the state fastest -> contiguous inside a problem
#pragma unroll
    for (int it = 0; it < (NT * 16 * NX + NTHREADS - 1) / NTHREADS; ++it) {
        const int item = tid + it * NTHREADS;
        const int idx = item / NX, i = item - idx * NX;
        const unsigned r = (unsigned)t0 * 16u + (unsigned)idx;
        if (item < NT * 16 * NX && r < cx.R) {
            const int j = idx >> 4, cc = idx & 15;
            const int qq = sizeof(T) == 8 ? (i & 3) : (i >> 2), rr = sizeof(T) == 8 ? (i >> 2) : (i & 3);
            T f = lds[L::BIASL + rr * 4 + qq];
#pragma unroll
            for (int ww = 0; ww < MT; ++ww) f += PART[((j * MT + ww) * NX + i) * 16 + cc];
            const T* tin = in + j * L::IN_TILE;
            const T xp = tin[cc * NIN + i];
            const T xt = tin[16 * NIN + cc * NX + i];
            const T phi = (cx.ident ? xp : T(0)) + f;
            const unsigned b = cx.invH ? __umulhi(r, cx.invH) : r;
            const int t = (int)(r - b * (unsigned)cx.H);
            T* gp = cx.gout + (size_t)b * cx.m + t * NX + i;
            gp[0] = phi - xt;
            if (cx.box) gp[(size_t)cx.H * NX] = xt;
        }
    }
    if constexpr (FUSE) {
        // ---- dense Jacobian rows of the pass (integrator/discret.py:38-56, unity.py:38-56; ipopt.py:88-96): row (t, i)
        //      of problem b holds -1 at x_t[i], the tile's state block at x_{t-1} (t >= 1), its control block at u_t,
        //      zeros elsewhere; box rows (+1 selectors) follow the defect rows of each problem.
        //      Every vector instruction costs the matrix pipe four cycles (tools/ubench_dpops.hip), so the rows are NOT
        //      computed per output vector (classify the column, compare with the row's step, select: ~66 vector
        //      instructions per store, 1.3 M per launch = 2.3 us at B=1024).  They are assembled in LDS -- the idle
        //      exchange area, zero-filled, then the 4 non-zeros per row dropped in by one lane per row -- and streamed out
        //      as a flat copy: one ds_read_b128 and one 1 KB-per-wave store per vector, addressing on the scalar unit.
        constexpr int VEC = 16 / (int)sizeof(T);
        typedef T vecT __attribute__((ext_vector_type(VEC)));
        constexpr int DR = NT * 16 * NX;                                 // dense rows of a full pass
        T* const RB = lds + L::X;
        vecT* const RBv = reinterpret_cast<vecT*>(RB);
        const int n = cx.n, nvec = n / VEC;
        const unsigned r0 = (unsigned)t0 * 16u;
        const int drv = r0 + NT * 16u <= cx.R ? DR : (int)(cx.R - r0) * NX;     // rows of the pass that exist
        const int rpc = drv < cx.rb_rows ? drv : cx.rb_rows;            // rows per chunk of the LDS row buffer
        const unsigned b0 = cx.invH ? __umulhi(r0, cx.invH) : r0;       // first problem of the pass (uniform)
        const vecT zero = {};
#ifdef NEMPC_EXP_NODENSE      // timing experiment only
        const int nkind = 0;
#else
        const int nkind = cx.box ? 2 : 1;
#endif
        for (int kind = 0; kind < nkind; ++kind) {
            for (int c0 = 0; c0 < drv; c0 += rpc) {
                const int nr = drv - c0 < rpc ? drv - c0 : rpc;
                lds_barrier();                 // TS complete; the previous chunk has left the buffer
                // each wave owns a run of rows: zero them, then one lane per row drops the non-zeros in (the LDS
                // executes one wave's operations in order, so no barrier between the two)
                const int rpw = (nr + MT - 1) / MT;
                const int lo = w * rpw, hi = lo + rpw < nr ? lo + rpw : nr;
                for (int v = lo * nvec + lane; v < hi * nvec; v += 64) RBv[v] = zero;
                asm volatile("" ::: "memory");
                if (lo + lane < hi) {
                    const int lr = c0 + lo + lane;
                    const int lrow = lr / NX, i = lr - lrow * NX;
                    const unsigned r = r0 + (unsigned)lrow;
                    const unsigned b = cx.invH ? __umulhi(r, cx.invH) : r;
                    const int t = (int)(r - b * (unsigned)cx.H);
                    T* row = RB + (lo + lane) * n;
                    if (kind == 0) {
                        const T* ts = TS + lrow * JROW + i * NIN;
                        if (t >= 1) {
#pragma unroll
                            for (int jj = 0; jj < NX; ++jj) row[(t - 1) * NX + jj] = ts[jj];
                        }
                        row[t * NX + i] = T(-1);
#pragma unroll
                        for (int jj = 0; jj < NU; ++jj) row[cx.H * NX + t * NU + jj] = ts[NX + jj];
                    } else {
                        row[t * NX + i] = T(1);
                    }
                }
                lds_barrier();
                const int nvc = nr * nvec;
                // write-through stores (sc0 sc1): the rows go out to memory as they are issued instead of sitting dirty
                // in L2 until the end-of-kernel write-back (C2, B=1024: whole evaluation 21.9 -> 20.3 us)
                if (!cx.box) {
                    // without box rows m = H*NX: dense row (r, i) is row r*NX + i of one (R*NX, n) matrix, the chunk is
                    // one contiguous block of memory
                    const char* base = reinterpret_cast<const char*>(cx.jac) +
                                       ((size_t)r0 * NX + (size_t)c0) * (size_t)n * sizeof(T);
                    // reads of a batch are all in flight before the first store waits for its data
                    constexpr int UB = 4;
                    for (int f0 = tid; f0 < nvc; f0 += UB * NTHREADS) {
                        vecT v[UB];
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const int fv = f0 + u * NTHREADS;
                            if (fv < nvc) v[u] = RBv[fv];
                        }
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const int fv = f0 + u * NTHREADS;
#ifdef NEMPC_EXP_NODENSE_STORE   // timing experiment only
                            if (v[u][0] == T(123.456))
#endif
                            if (fv < nvc) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(fv * 16), "v"(v[u]), "s"(base));
                        }
                    }
                } else {
                    const char* base = reinterpret_cast<const char*>(cx.jac) + (size_t)b0 * cx.m * (size_t)n * sizeof(T);
                    constexpr int UB = 4;
                    for (int f0 = tid; f0 < nvc; f0 += UB * NTHREADS) {
                        vecT v[UB];
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const int fv = f0 + u * NTHREADS;
                            if (fv < nvc) v[u] = RBv[fv];
                        }
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const int fv = f0 + u * NTHREADS;
                            const int lrr = cx.inv_nvec ? (int)__umulhi((unsigned)fv, cx.inv_nvec) : fv;
                            const int cv = fv - lrr * nvec;
                            const int lr = c0 + lrr;
                            const int lrow = lr / NX, i = lr - lrow * NX;
                            const unsigned r = r0 + (unsigned)lrow;
                            const unsigned b = cx.invH ? __umulhi(r, cx.invH) : r;
                            const int t = (int)(r - b * (unsigned)cx.H);
                            const int rowidx = (int)(b - b0) * cx.m + (kind ? cx.H * NX : 0) + t * NX + i;
                            const int voff = (rowidx * n + cv * VEC) * (int)sizeof(T);
                            if (fv < nvc) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(voff), "v"(v[u]), "s"(base));
                        }
                    }
                }
            }
        }
    }
    lds_barrier();
}

// FUSE: the whole hessian-free evaluation in this launch -- g, [tiles,] dense jac, f, grad
template <typename T, int WP, int NH, int TPW, int NX, int NU, bool FUSE = false>
__global__ __launch_bounds__((WP / 16) * 64, 2) void rows_coopfx_kernel(
    // the 16 dwords the first vector loads depend on, as plain arguments: with -amdgpu-kernarg-preload-count=16 they are
    // in scalar registers when the wave starts instead of behind a scalar-load round trip (the struct repeats them)
    const void* pZ, const void* pX0, const void* psmall, const void* pwslice, int ptiles_per_wg, int ptiles_rem,
    unsigned pR, unsigned pinvH, int pH, int pn, int pm, int pident, FxArgs a) {
    a.Z = pZ; a.X0 = pX0; a.small = psmall; a.wslice = pwslice; a.tiles_per_wg = ptiles_per_wg; a.tiles_rem = ptiles_rem;
    a.R = pR; a.invH = pinvH; a.H = pH; a.n = pn; a.m = pm; a.ident = pident;
    using L = FxLayout<T, WP, NH, TPW, NX, NU>;
    constexpr int MT = WP / 16;
    constexpr int NTHREADS = MT * 64;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int NCOL = NX + NU + NX;
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;

    FxCtx<T, WP, NH, TPW, NX, NU> cx;
    cx.lds = lds;
    cx.Z = static_cast<const T*>(a.Z);
    cx.X0 = static_cast<const T*>(a.X0);
    cx.R = a.R; cx.invH = a.invH; cx.H = a.H; cx.n = a.n;

    const int t_begin = blockIdx.x * a.tiles_per_wg + ((int)blockIdx.x < a.tiles_rem ? (int)blockIdx.x : a.tiles_rem);
    const int t_end = t_begin + a.tiles_per_wg + ((int)blockIdx.x < a.tiles_rem ? 1 : 0);

    // ---- every global load of the prologue is issued before anything waits: inputs, small tables, weight slices.
    //      The staging registers always cover a full TPW-tile pass; a shorter pass just leaves rows unused.
    FxStage<T, TPW, NTHREADS, NCOL> sr;
    int t0 = t_begin;
    fx_stage_load<T, WP, NH, TPW, NX, NU, TPW>(cx, t0, tid, sr);
    // fused evaluation: the objective of this workgroup's problems (those whose first row lies in its tile range) is one
    // problem per wave, one step per lane.  Its inputs are fetched HERE, with the first loads, and it is evaluated while
    // the weight slices stream in; done at the end of the kernel it added its own global round trip to every
    // workgroup's tail (1.3 us of the launch).  Horizons beyond 64 steps and problems beyond the first MT take the tail path.
    unsigned ob_lo = 0, ob_hi = 0;
    bool ob_pro = false;
    T ox[NX], ou[NU], oxr[NX], our[NU], ocx[NX], ocu[NU], oQ[NX * NX], oQs[NX * NX], oR[NU * NU], oRs[NU * NU];
#ifdef NEMPC_EXP_NOOBJ   // timing experiment only
    a.f = nullptr; a.grad = nullptr;
#endif
    if constexpr (FUSE) {
        if (a.f || a.grad) {
            const unsigned r_lo = (unsigned)t_begin * 16u;
            unsigned r_hi = (unsigned)t_end * 16u;
            if (r_hi > a.R) r_hi = a.R;
            const unsigned Hh = (unsigned)a.H;
            ob_lo = (r_lo + Hh - 1) / Hh;
            ob_hi = (r_hi + Hh - 1) / Hh;                 // problems ob_lo .. ob_hi - 1
            ob_pro = a.H <= 64;
            const unsigned bw = ob_lo + (unsigned)w;
            if (ob_pro && bw < ob_hi) {
                const T* __restrict__ P = static_cast<const T*>(a.P);
                const int t = lane < a.H ? lane : 0;
                const T* z = cx.Z + (size_t)bw * a.n;
                const bool last = t == a.H - 1;
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    ox[i] = z[t * NX + i];
                    oxr[i] = P[a.oo.xref + t * NX + i];
                    ocx[i] = P[a.oo.cx + t * NX + i];
                }
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    ou[i] = z[a.H * NX + t * NU + i];
                    our[i] = P[a.oo.uref + t * NU + i];
                    ocu[i] = P[a.oo.cu + t * NU + i];
                }
#pragma unroll
                for (int i = 0; i < NX * NX; ++i) {
                    oQ[i] = P[(last ? a.oo.QT : a.oo.Q) + i];
                    oQs[i] = P[(last ? a.oo.QTs : a.oo.Qs) + i];
                }
#pragma unroll
                for (int i = 0; i < NU * NU; ++i) {
                    oR[i] = P[a.oo.R + i];
                    oRs[i] = P[a.oo.Rs + i];
                }
            }
        }
    }
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
    // fused evaluation: the objective table rides along (-> LDS behind the layout, read at the very end); tables
    // beyond PV_PER_THREAD * NTHREADS elements (horizons past ~80 steps) are completed just before they are used
    constexpr int PV_PER_THREAD = 2;
    T pv[PV_PER_THREAD];
    if constexpr (FUSE) {
        const T* __restrict__ gp = static_cast<const T*>(a.P);
#pragma unroll
        for (int u = 0; u < PV_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
            pv[u] = idx < a.p_elems ? gp[idx] : T(0);
        }
    }
    constexpr int NFRAG = (NH - 1) * 2 * MT * 4 + 8;
    constexpr int NLOAD = (NFRAG + VEC - 1) / VEC;
    vecT wv[NLOAD];
    {
        const vecT* __restrict__ ws = static_cast<const vecT*>(a.wslice) + (size_t)w * NLOAD * 64 + lane;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) wv[k] = ws[k * 64];
    }
    cx.gout = static_cast<T*>(a.g);
    cx.tiles = static_cast<T*>(a.tiles);
    cx.jac = static_cast<T*>(a.jac);
    cx.m = a.m; cx.ident = a.ident; cx.box = a.box;
    cx.rb_rows = a.rb_rows; cx.inv_nvec = a.inv_nvec;
    {
        vecT* ls = reinterpret_cast<vecT*>(lds + L::W0F);
#pragma unroll
        for (int u = 0; u < SMALL_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
            if (idx < SMALL_VECS) ls[idx] = sm[u];
        }
    }
    if constexpr (FUSE) {
#pragma unroll
        for (int u = 0; u < PV_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
            if (idx < a.p_elems) lds[L::TOTAL + idx] = pv[u];
        }
    }
    if constexpr (FUSE) {
        const unsigned bw = ob_lo + (unsigned)w;
        if (ob_pro && bw < ob_hi) {
            // same operations in the same order as objective_body (kernels_obj_impl.h): same bits
            double acc = 0.0;
            if (lane < a.H) {
                T* grad = static_cast<T*>(a.grad);
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const T dxi = ox[i] - oxr[i];
                    T qd = T(0), qsd = T(0);
#pragma unroll
                    for (int j = 0; j < NX; ++j) {
                        const T dxj = ox[j] - oxr[j];
                        qd = fma(oQ[i * NX + j], dxj, qd);
                        qsd = fma(oQs[i * NX + j], dxj, qsd);
                    }
                    acc += (double)(dxi * qd + ocx[i] * ox[i]);
                    if (grad) grad[(size_t)bw * a.n + lane * NX + i] = qsd + ocx[i];
                }
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    const T dui = ou[i] - our[i];
                    T rd = T(0), rsd = T(0);
#pragma unroll
                    for (int j = 0; j < NU; ++j) {
                        const T duj = ou[j] - our[j];
                        rd = fma(oR[i * NU + j], duj, rd);
                        rsd = fma(oRs[i * NU + j], duj, rsd);
                    }
                    acc += (double)(dui * rd + ocu[i] * ou[i]);
                    if (grad) grad[(size_t)bw * a.n + a.H * NX + lane * NU + i] = rsd + ocu[i];
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
            if (a.f && lane == 0) static_cast<T*>(a.f)[bw] = (T)acc;
        }
    }
    CoopWeights<T, WP, NH> W;
    {
        int f = 0;
#pragma unroll
        for (int l = 1; l < NH; ++l) {
#pragma unroll
            for (int i = 0; i < MT * 4; ++i, ++f) W.wf[l - 1][i] = wv[f / VEC][f % VEC];
#pragma unroll
            for (int i = 0; i < MT * 4; ++i, ++f) W.wb[l - 1][i] = wv[f / VEC][f % VEC];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r, ++f) W.wL[r] = wv[f / VEC][f % VEC];
#pragma unroll
        for (int r = 0; r < 4; ++r, ++f) W.w0b[r] = wv[f / VEC][f % VEC];
    }

    int parity = 0, xsel = 0;
    T* const in_base = lds + L::IN;
    fx_stage_store<T, WP, NH, TPW, NX, NU, TPW>(in_base, tid, sr);
    while (t0 < t_end) {
        const int t_cur = t0;
        const int n_cur = t_end - t0 < TPW ? t_end - t0 : TPW;
        t0 += n_cur;
        const bool more = t0 < t_end;
        if (more) fx_stage_load<T, WP, NH, TPW, NX, NU, TPW>(cx, t0, tid, sr);   // next pass's inputs, under this pass
        lds_barrier();
        const T* in = in_base + parity * L::IN_SZ;
        T* const in_next = in_base + (parity ^ 1) * L::IN_SZ;
        if (n_cur == 1) fx_pass<T, WP, NH, TPW, NX, NU, 1, FUSE>(cx, W, in, t_cur, tid, xsel, sr, more, in_next);
        if constexpr (TPW >= 2) { if (n_cur == 2) fx_pass<T, WP, NH, TPW, NX, NU, 2, FUSE>(cx, W, in, t_cur, tid, xsel, sr, more, in_next); }
        if constexpr (TPW >= 3) { if (n_cur == 3) fx_pass<T, WP, NH, TPW, NX, NU, 3, FUSE>(cx, W, in, t_cur, tid, xsel, sr, more, in_next); }
        parity ^= 1;
    }
    if constexpr (FUSE) {
        // ---- objective of the problems whose first row lies in this workgroup's tile range, one problem per wave
        //      at a time (same routine, hence the same bits, as the assembly kernels' objective blocks)
        if (a.f || a.grad) {
            if (a.p_elems > PV_PER_THREAD * NTHREADS) {
                const T* __restrict__ gp = static_cast<const T*>(a.P);
                for (int i = tid + PV_PER_THREAD * NTHREADS; i < a.p_elems; i += NTHREADS) lds[L::TOTAL + i] = gp[i];
                __syncthreads();
            }
            for (unsigned b = ob_lo + (unsigned)w + (ob_pro ? MT : 0); b < ob_hi; b += MT)
                objective_body<T>((int)b, lane, a.H, NX, NU, a.oo, lds + L::TOTAL, cx.Z, static_cast<T*>(a.f),
                                  static_cast<T*>(a.grad));
        }
    }
}

}  // namespace nempc
