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

// A/B switches of the fused dense output (tools/build_variant.py -D...)
#ifndef NEMPC_FX_ZERO_EARLY
#define NEMPC_FX_ZERO_EARLY 0      // 1: background zeros before the pass's first barrier instead of behind layer 0
#endif

#ifdef NEMPC_STAMPS_NO_FX          // (tools/diag_stamps_c3.py: stamps in the run-time-dims kernel only)
#undef COOP_WGSTAMP
#undef COOP_WGSTAMP_REAL
#define COOP_WGSTAMP(p, i) do { } while (0)
#define COOP_WGSTAMP_REAL(p, i) do { } while (0)
#endif
// diagnostic builds only (tools/diag_stamps.py): the per-workgroup timeline has 13 event slots; -DNEMPC_STAMPS_PRO spends
// them on the prologue instead of the pass
#ifdef NEMPC_STAMPS_PRO
#define FX_STAMP_PASS(p, i) do { } while (0)
#define FX_STAMP_PRO(p, i) COOP_WGSTAMP(p, i)
#else
#define FX_STAMP_PASS(p, i) COOP_WGSTAMP(p, i)
#define FX_STAMP_PRO(p, i) do { } while (0)
#endif

// (Raised wave priority -- s_setprio -- over the matrix-instruction blocks was measured in round 4: 15.89 / 15.93 us against
// 15.92 / 16.03 without; no effect, removed.)

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
    // K-split partials, one value per (wave, tile, quantity, row): network outputs PF[w][j][k][16], Jacobian rows
    // PJ[k][w][j][d][16] (value-major inside a wave's block: fx_rowsums_store writes four values per store)
    static constexpr int PART = X + 2 * XH;
    static constexpr int PF_SZ = TPW * MT * NX * 16;
    static constexpr int PART_SZ = PF_SZ + NX * TPW * MT * NIN * 16;
    // inputs, double-buffered: per tile xi[16][NIN] then xt[16][NX]
    static constexpr int IN_TILE = 16 * (NIN + NX);
    static constexpr int IN = PART + PART_SZ;
    static constexpr int IN_SZ = (TPW * IN_TILE + 15) & ~15;
    static constexpr int TOTAL = IN + 2 * IN_SZ;
};

constexpr int FX_ZCOPY = 512;   // elements of Z (the workgroup's problems) parked in LDS for the objective: 2 per thread

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
    int zp_max;             // problems whose variables fit the LDS copy (FX_ZCOPY / n; 0 when the table itself is too long)
    long long* dbg;         // diagnostic builds only
    ObjOffsets oo;
    // ---- Gauss-Newton Hessian callback from this launch (GN instantiation; nempc_hess_gn asked for the tril values only):
    //      hvals[b][e] = sigma_b * objc[e] + sum_i w_i T_ip T_iq through the scatter form of the Hessian map (see HessParams)
    void* gn_hvals;
    const void* gn_sigma;
    const void* gn_w;       // (B, H*nx) weights per defect row, null = ones
    const int32_t* gn_smap;
    const void* gn_objc;
    int gn_nnz, gn_n_orph;
    // ---- sparse contract from this launch: the band-pattern values (B, nnz) in nempc_jac_structure order (row-major over the
    //      dense matrix: per defect row (t, i) the state block at x_{t-1} (t >= 1), -1 at x_t[i], the control block at u_t;
    //      then the box rows' +1 selectors), written by the lane that holds the row -- no tile round trip, no assembly launch
    void* jac_sp;
    int sp_nnz;
};

// The argument block proper, read where it is needed.  Kernel arguments are fetched by scalar loads that the compiler
// issues at the top of the kernel, and with ~100 scalar registers live it then waits for them at once to spill them
// into vector lanes: a cold round trip (0.3-0.5 us) in front of the very first vector load.  What the prologue needs
// travels in the preloaded leading arguments; the rest -- output pointers, m, box, ... -- is only used by a pass's
// epilogue and is read THERE through the kernarg segment pointer (the empty asm pins the earliest point).
constexpr int FX_ARGS_KERNARG_OFFSET = 64;      // five pointers, four dwords and the dense-matrix pointer precede the FxArgs argument
typedef const FxArgs __attribute__((address_space(4)))* FxArgsK;
__device__ __forceinline__ FxArgsK fx_late_args() {
    const char __attribute__((address_space(4)))* kp =
        (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return (FxArgsK)(kp + FX_ARGS_KERNARG_OFFSET);
}

template <typename T, int WP, int NH, int TPW, int NX, int NU>
struct FxCtx {
    using L = FxLayout<T, WP, NH, TPW, NX, NU>;
    T* lds;
    const T* __restrict__ Z;
    const T* __restrict__ X0;
    unsigned R, invH;
    int H, n;
    bool rev;               // reverse sweeps wanted (tiles or dense rows asked for); false: defects [+ objective] only
    T* jac;                 // fused evaluation: the dense Jacobian (null: not asked for)
    bool box;               // box rows follow each problem's defect rows
    long long* dbg;         // diagnostic builds only (-DNEMPC_STAMPS, tools/diag_stamps.py): per-workgroup timeline
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

// Several row sums at once.  v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the
// second; fed two DIFFERENT values a, b it leaves [a0 b0 a2 b2] and [a1 b1 a3 b3], whose sum holds a0+a1 and a2+a3 in rows
// 0 / 2 and b0+b1, b2+b3 in rows 1 / 3: one swap pair and one add take two values through a stage (fx_qsum spends that
// on one).  v_permlane32_swap then folds two such registers, and row q of the result is the full sum of value q -- four
// values for 9 instructions instead of 24, each summed in fx_qsum's order ((r0 + r1) + (r2 + r3)).
__device__ __forceinline__ double fx_swap16_add(double a, double b) {
    auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double fx_swap32_add(double a, double b) {
    auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ float fx_swap16_add(float a, float b) {
    auto v = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(v[0]) + __uint_as_float(v[1]);
}
__device__ __forceinline__ float fx_swap32_add(float a, float b) {
    auto v = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(v[0]) + __uint_as_float(v[1]);
}
// Row sums of NV per-lane values -> dst[v * 16 + c] (value-major, 16 row entries each): groups of four leave through one
// lane-linear store (lane = q*16 + c holds value 4g + q), the remainder through the narrower forms.
template <typename T, int NV>
__device__ __forceinline__ void fx_rowsums_store(const T (&s)[NV], T* dst, int lane) {
    const int q = lane >> 4;
#pragma unroll
    for (int g = 0; g < NV / 4; ++g)
        dst[g * 64 + lane] = fx_swap32_add(fx_swap16_add(s[4 * g], s[4 * g + 1]), fx_swap16_add(s[4 * g + 2], s[4 * g + 3]));
    constexpr int G = NV / 4, REM = NV % 4;
    if constexpr (REM == 1) {
        const T v = fx_qsum(s[4 * G]);
        if (q == 0) dst[G * 64 + lane] = v;
    } else if constexpr (REM == 2) {
        const T p = fx_swap16_add(s[4 * G], s[4 * G + 1]);
        const T v = fx_swap32_add(p, p);                      // rows [a b a b]
        if (q < 2) dst[G * 64 + lane] = v;
    } else if constexpr (REM == 3) {
        const T v = fx_swap32_add(fx_swap16_add(s[4 * G], s[4 * G + 1]), fx_swap16_add(s[4 * G + 2], s[4 * G + 2]));
        if (q < 3) dst[G * 64 + lane] = v;
    }
}

// Fused evaluation, dense Jacobian (integrator/discret.py:38-56, unity.py:38-56; ipopt.py:88-96), part one: the
// BACKGROUND.  Of a dense row's n entries all but 1 + NX + NU are structural zeros -- 98 % of the matrix at C2, 97 % at
// C5 -- and WHICH entries are zero does not depend on the iterate.  So the zeros of a pass's rows are streamed at the
// START of the pass, from registers, as one flat run of 16-byte write-through stores per contiguous block: they cost a
// store instruction each (no LDS, no address arithmetic beyond an add), need nothing the pass computes, and reach HBM
// while the matrix pipe works -- the memory system is idle then.  Part two (the pass's epilogue) overwrites the few
// non-zeros once every zero store of the workgroup has been acknowledged (vmcnt(0) on every wave, then the barrier):
// same-address writes from one workgroup go through one L2 channel in the order they were issued, and the earlier one
// has arrived.  Before, the rows were assembled in LDS after the sweeps and copied out from there: the whole 20 MB
// (246 MB at C5) left in the launch's tail, the last workgroup's copy-out exposed (2.3 us of a 17 us launch), with a
// zero-fill, a row buffer, two barriers per chunk and ~400 vector instructions per wave on the pipe the matrix
// instructions need.
//   no box rows: m = H*NX, dense row (r, i) is row r*NX + i of ONE (R*NX, n) matrix -- the pass's rows are one block.
//   box rows:    per problem a defect block and a box block (+1 selectors, written in part two as well); the run is
//                cut where the problem changes (scalar arithmetic, at most a few pieces).
template <typename T, int NX, int NTHREADS>
__device__ __forceinline__ void fx_zero_rows(T* o_jac, unsigned r0, int nrows, int n, int H, unsigned invH, bool box, int tid) {
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    const vecT zero = {};
    const int nvec = n / VEC;
#if defined(NEMPC_EXP_NODENSE) || defined(NEMPC_EXP_NOZERO)      // timing experiments only
    const int drv = 0;
#else
    const int drv = nrows * NX;                     // dense rows of the pass that exist
#endif
    // a flat run of nv 16-byte vectors from `base`: whole rounds of NTHREADS stores with the lane's offset fixed and the
    // (scalar) base stepping -- no vector instruction in the loop but the store -- then the partial round
    auto run = [&](const char* base, int nv) {
        const int full = nv / NTHREADS;
        const int voff = tid * 16;
        for (int k = 0; k < full; ++k) {
#ifdef NEMPC_STAMPS      // (diagnostic builds: the stamps' asm statements cost the compiler its proof that `base` is uniform)
            base = fx_uniform_ptr(base);
#endif
            asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(zero), "s"(base) : "memory");
            base += NTHREADS * 16;
        }
#ifdef NEMPC_STAMPS
        base = fx_uniform_ptr(base);
#endif
        if (tid < nv - full * NTHREADS)
            asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(voff), "v"(zero), "s"(base) : "memory");
        // (store-data hazard: `zero`'s registers may be re-used right behind a store -- the wait states ride in the store's own
        // asm statement: a separate `s_nop` statement does not keep the scheduler from placing an instruction in between)
    };
    if (!box) {
        run(fx_uniform_ptr(reinterpret_cast<const char*>(o_jac) + (size_t)r0 * NX * (size_t)n * sizeof(T)), drv * nvec);
    } else {
        const int HNX = H * NX;
        for (int s0 = 0; s0 < drv;) {
            const unsigned r = r0 + (unsigned)(s0 / NX);
            const unsigned b = invH ? __umulhi(r, invH) : r;
            const int k = (int)(r - b * (unsigned)H) * NX;              // first dense row of the piece within its block
            const int len = drv - s0 < HNX - k ? drv - s0 : HNX - k;    // rows up to the end of the problem / pass
#pragma unroll
            for (int kind = 0; kind < 2; ++kind)
                run(fx_uniform_ptr(reinterpret_cast<const char*>(o_jac) +
                                   ((size_t)b * (2 * HNX) + (size_t)(kind * HNX + k)) * (size_t)n * sizeof(T)), len * nvec);
            s0 += len;
        }
    }
}

// `nxt` / `in_next` (when has_next): the NEXT pass's inputs, already in registers; they go to the other input buffer
// BEFORE this pass's global stores are issued -- vmcnt counts stores too and retires in order, so a wait for those loads
// placed after the stores would sit out the stores' acknowledgement (with the dense rows fused in: the whole HBM time).
template <typename T, int WP, int NH, int TPW, int NX, int NU, int NT, bool FUSE, int ACT, bool GN = false>
__device__ __forceinline__ void fx_pass(const FxCtx<T, WP, NH, TPW, NX, NU>& cx, const CoopWeights<T, WP, NH>& W,
                                        const T* in, int t0, int tid, int& xsel,
                                        const FxStage<T, TPW, (WP / 16) * 64, NX + NU + NX>& nxt, bool has_next, T* in_next) {
    using Ops = MfmaOps<T>;
    using A = Act<T, ACT>;
    static_assert(ACT != NEMPC_ACT_RUNTIME, "the compiled-shape kernels take a compile-time activation");
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
            for (int r = 0; r < 4; ++r) a[0][j][r] = A::f(a[0][j][r]);
    }
    FX_STAMP_PRO(cx.dbg, 10);
#if !NEMPC_FX_ZERO_EARLY
    if constexpr (FUSE) {
        // background zeros of this pass's dense rows (fx_zero_rows): issued here, behind layer 0 -- the first thing a
        // pass computes needs no store slot, and in the first pass the prologue's loads are ahead of them in the queue
        if (cx.jac) {
            const unsigned r0 = (unsigned)t0 * 16u;
            const int nrows = r0 + (unsigned)NT * 16u <= cx.R ? NT * 16 : (int)(cx.R - r0);
            fx_zero_rows<T, NX, NTHREADS>(cx.jac, r0, nrows, cx.n, cx.H, cx.invH, cx.box, tid);
        }
    }
#endif
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
            for (int r = 0; r < 4; ++r) a[l][j][r] = A::f(a[l][j][r]);
    }
    FX_STAMP_PRO(cx.dbg, 11);
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
    {
        T sv[NT * NX];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                T v = a[NH - 1][j][0] * wl[k][0];
#pragma unroll
                for (int r = 1; r < 4; ++r) v = fma(a[NH - 1][j][r], wl[k][r], v);
                sv[j * NX + k] = v;
            }
        fx_rowsums_store<T, NT * NX>(sv, PART + w * (TPW * NX) * 16, lane);
    }
    FX_STAMP_PASS(cx.dbg, 5);
    T* const PJ = PART + L::PF_SZ;
    // a defect-only evaluation (a line-search trial of the batched solver: g and f, no derivatives) ends the arithmetic here
    if (cx.rev) {
#pragma unroll
    for (int l = 0; l < NH; ++l)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[l][j][r] = A::d1(a[l][j][r]);

    // ---- reverse sweep, one cotangent (network output) at a time
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
        {
            T sv[NT * NIN];
#pragma unroll
            for (int d = 0; d < NIN; ++d) {
                const T* p0 = lds + L::P0 + d * MT * 16 + w * 16;
                V4 w0;
#pragma unroll
                for (int r = 0; r < 4; ++r) w0[r] = p0[r * 4 + q];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    T v = cv[j][0] * w0[0];
#pragma unroll
                    for (int r = 1; r < 4; ++r) v = fma(cv[j][r], w0[r], v);
                    sv[j * NIN + d] = v;
                }
            }
            fx_rowsums_store<T, NT * NIN>(sv, PJ + (k * MT + w) * (TPW * NIN) * 16, lane);
        }
    }
    }   // cx.rev
    FX_STAMP_PASS(cx.dbg, 6);
    FX_STAMP_PRO(cx.dbg, 12);
    // what only the epilogue needs, fetched now: the barrier below covers the scalar loads
    const FxArgsK ka = fx_late_args();
    T* const o_tiles = static_cast<T*>(ka->tiles);
    T* const o_g = static_cast<T*>(ka->g);
    const int a_ident = ka->ident, a_box = ka->box, a_m = ka->m;
    if constexpr (FUSE) {
        // (the background zeros of this pass's dense rows: every wave waits for its own stores' acknowledgements, the
        // barrier then covers the workgroup; they were issued a whole pass ago)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
    if (has_next) fx_stage_store<T, WP, NH, TPW, NX, NU, TPW>(in_next, tid, nxt);
    FX_STAMP_PASS(cx.dbg, 7);

    if constexpr (GN) {
        // ---- Gauss-Newton callback: the blocks sum_i w_i T_i^T T_i and their tril assembly, here -- no tile round trip
        //      through memory, no assembly launch.  Lanes run over (row, state) with the state fastest: a lane sums ITS
        //      row of the tile (state i) from the K-split partials, forms w_i T_ip T_iq for the pairs, and the NX lanes
        //      of a tile row add up by a quad permute (NX = 2); the lane of state 0 writes the block's entries through
        //      the scatter map (assemble_hess_gn_kernel's arithmetic: sigma * objective constant + block element).
        static_assert(NX == 2, "the fused Gauss-Newton assembly sums the states of a row by one quad permute");
        constexpr int NPAIR = NIN * (NIN + 1) / 2, BSZ = NIN * NIN;
        T* const hv = static_cast<T*>(ka->gn_hvals);
        const T* const sg = static_cast<const T*>(ka->gn_sigma);
        const T* const gw = static_cast<const T*>(ka->gn_w);
        const T* const oc = static_cast<const T*>(ka->gn_objc);
        const int32_t* const smap = ka->gn_smap;
        const int nnz = ka->gn_nnz, n_orph = ka->gn_n_orph;
#pragma unroll
        for (int it = 0; it < (NT * 16 * NX + NTHREADS - 1) / NTHREADS; ++it) {
            const int item = tid + it * NTHREADS;
            const int idx = item / NX, i = item - idx * NX;
            const unsigned r = (unsigned)t0 * 16u + (unsigned)idx;
            const bool valid = item < NT * 16 * NX && r < cx.R;
            const int j = (idx >> 4) < NT ? (idx >> 4) : 0, cc = idx & 15;
            const unsigned b = cx.invH ? __umulhi(r, cx.invH) : r;
            const int t = (int)(r - b * (unsigned)cx.H);
            T c[NPAIR];
            {
                T ts[NIN];
#pragma unroll
                for (int d = 0; d < NIN; ++d) {
                    T v = T(0);
#pragma unroll
                    for (int ww = 0; ww < MT; ++ww) v += PJ[((i * MT + ww) * (TPW * NIN) + j * NIN + d) * 16 + cc];
                    if (a_ident && d == i) v += T(1);
                    ts[d] = valid ? v : T(0);
                }
                const T wi = (valid && gw) ? gw[(size_t)b * (cx.H * NX) + t * NX + i] : T(1);
                int pq = 0;
#pragma unroll
                for (int p = 0; p < NIN; ++p) {
                    const T wp = wi * ts[p];
#pragma unroll
                    for (int qq = 0; qq <= p; ++qq, ++pq) c[pq] = wp * ts[qq];
                }
            }
            // + the other state's lane (lanes 2k, 2k+1 hold states 0, 1 of one tile row): state 0 first, then state 1
#pragma unroll
            for (int pq = 0; pq < NPAIR; ++pq) {
                const int lo = __double2loint((double)c[pq]), hi = __double2hiint((double)c[pq]);
                const int l2 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false);     // quad_perm [1, 0, 3, 2]
                const int h2 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false);
                const T other = (T)__hiloint2double(h2, l2);
                c[pq] = i == 0 ? c[pq] + other : other + c[pq];
            }
            if (valid && i == 0) {
                const T sb = sg[b];
#pragma unroll
                for (int e = 0; e < BSZ; ++e) {
                    const int a1 = e / NIN, a2 = e - a1 * NIN;
                    const int hi2 = a1 > a2 ? a1 : a2, lo2 = a1 > a2 ? a2 : a1;
                    const int ent = smap[t * BSZ + e];
                    if (ent >= 0) hv[(size_t)b * nnz + ent] = sb * oc[ent] + c[hi2 * (hi2 + 1) / 2 + lo2];
                }
                if (t == cx.H - 1)
                    for (int e = 0; e < n_orph; ++e) {
                        const int oe = smap[cx.H * BSZ + e];
                        hv[(size_t)b * nnz + oe] = sb * oc[oe];
                    }
            }
        }
        lds_barrier();
        return;
    }

    // ---- outputs straight from the partials (no reduction phase): the sum over the MT waves is taken here, in wave
    //      order like the generic kernel's reduction
    // tiles: the pass's NT*16 rows are contiguous in memory -> lanes run over the flat element index (coalesced).
    // (The fused evaluation with the dense matrix sums them where it places them into the dense rows, below.)
    if (cx.rev && o_tiles && (!FUSE || !ka->jac))
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
            for (int ww = 0; ww < MT; ++ww) v += PJ[((k * MT + ww) * (TPW * NIN) + j * NIN + d) * 16 + cc];
            if (a_ident && d == k) v += T(1);
            o_tiles[(size_t)t0 * (16 * JROW) + item] = v;
        }
    }
    // ---- defects, and -- fused evaluation -- the non-zeros of the dense Jacobian rows, in ONE loop: lanes run over
    //      (row, state) with the state fastest, which is both the defect entry g[(t, i)] and the dense row (t, i).
    //      Dense rows, part two: over the background streamed at the start of the pass (fx_zero_rows; every zero store of
    //      the workgroup was acknowledged before the barrier above) row (t, i) of problem b gets -1 at x_t[i], the tile's
    //      state block at x_{t-1} (t >= 1), its control block at u_t; box rows (+1 selectors) follow the defect rows of
    //      each problem.  The lane sums the row's tile entries from the K-split partials in wave order (the unfused
    //      kernel's loop above) and stores them.
    T* o_jac = nullptr;
    T* o_sp = nullptr;
    int sp_nnz = 0;
    if constexpr (FUSE) {
#if !(defined(NEMPC_EXP_NODENSE) || defined(NEMPC_EXP_NONZ))      // (timing experiments only)
        o_jac = static_cast<T*>(ka->jac);
#endif
        o_sp = static_cast<T*>(ka->jac_sp);
        sp_nnz = ka->sp_nnz;
    }
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
            for (int ww = 0; ww < MT; ++ww) f += PART[(ww * (TPW * NX) + j * NX + i) * 16 + cc];
            const T* tin = in + j * L::IN_TILE;
            const T xp = tin[cc * NIN + i];
            const T xt = tin[16 * NIN + cc * NX + i];
            const T phi = (a_ident ? xp : T(0)) + f;
            const unsigned b = cx.invH ? __umulhi(r, cx.invH) : r;
            const int t = (int)(r - b * (unsigned)cx.H);
            T* gp = o_g + (size_t)b * a_m + t * NX + i;
            gp[0] = phi - xt;
            if (a_box) gp[(size_t)cx.H * NX] = xt;
            if constexpr (FUSE) {
                if (o_jac || o_sp) {
                    const int n = cx.n;
                    T ts[NIN];
#pragma unroll
                    for (int d = 0; d < NIN; ++d) {
                        T v = T(0);
#pragma unroll
                        for (int ww = 0; ww < MT; ++ww) v += PJ[((i * MT + ww) * (TPW * NIN) + j * NIN + d) * 16 + cc];
                        if (a_ident && d == i) v += T(1);
                        ts[d] = v;
                    }
                    if (o_jac) {
                    if (o_tiles) {
#pragma unroll
                        for (int d = 0; d < NIN; ++d) o_tiles[(size_t)r * JROW + i * NIN + d] = ts[d];
                    }
                    T* const row = o_jac + ((size_t)b * a_m + (size_t)(t * NX + i)) * (size_t)n;
                    if (t >= 1) {
                        if constexpr (NX == 2 && sizeof(T) == 8) {
                            fx_store_wt2(row + (t - 1) * NX, ts[0], ts[1]);      // n even: the block is 16-byte aligned
                        } else {
#pragma unroll
                            for (int jj = 0; jj < NX; ++jj) fx_store_wt(row + (t - 1) * NX + jj, ts[jj]);
                        }
                    }
                    fx_store_wt(row + t * NX + i, T(-1));
#pragma unroll
                    for (int jj = 0; jj < NU; ++jj) fx_store_wt(row + cx.H * NX + t * NU + jj, ts[NX + jj]);
                    if (a_box) fx_store_wt(row + (size_t)cx.H * NX * (size_t)n + t * NX + i, T(1));
                    }
                    if (o_sp) {
                        // the band values of dense row (t, i), in the row's column order: [state block | -1 | control block]
                        // (t = 0 has no state block: x0 is data); consecutive lanes write consecutive pieces of the problem's
                        // value vector.  2/1 in double: a piece is 32 bytes (16 at t = 0) at a 16-byte aligned offset.
                        constexpr int ROW0 = 1 + NU, ROWT = NX + 1 + NU;
                        T* const sp = o_sp + (size_t)b * sp_nnz;
                        if (t >= 1) {
                            T* const pc = sp + NX * ROW0 + ((t - 1) * NX + i) * ROWT;
                            if constexpr (NX == 2 && NU == 1 && sizeof(T) == 8) {
                                fx_store_wt2(pc, ts[0], ts[1]);
                                fx_store_wt2(pc + 2, T(-1), ts[2]);
                            } else {
#pragma unroll
                                for (int jj = 0; jj < NX; ++jj) pc[jj] = ts[jj];
                                pc[NX] = T(-1);
#pragma unroll
                                for (int jj = 0; jj < NU; ++jj) pc[NX + 1 + jj] = ts[NX + jj];
                            }
                        } else {
                            T* const pc = sp + i * ROW0;
                            if constexpr (NX == 2 && NU == 1 && sizeof(T) == 8) {
                                fx_store_wt2(pc, T(-1), ts[2]);
                            } else {
                                pc[0] = T(-1);
#pragma unroll
                                for (int jj = 0; jj < NU; ++jj) pc[1 + jj] = ts[NX + jj];
                            }
                        }
                        if (a_box) sp[NX * ROW0 + (cx.H - 1) * NX * ROWT + t * NX + i] = T(1);
                    }
                }
            }
        }
    }
    FX_STAMP_PASS(cx.dbg, 8);
    FX_STAMP_PASS(cx.dbg, 11);
    lds_barrier();
    FX_STAMP_PASS(cx.dbg, 12);
}

// FUSE: the whole hessian-free evaluation in this launch -- g, [tiles,] dense jac, f, grad
// GN: the Gauss-Newton Hessian callback from this launch (tril values only; no g / tiles / dense outputs)
template <typename T, int WP, int NH, int TPW, int NX, int NU, bool FUSE, int ACT, bool GN = false>
__global__ __launch_bounds__((WP / 16) * 64, 2) void rows_coopfx_kernel(
    // what the prologue's loads depend on, as plain arguments: with -amdgpu-kernarg-preload-count the leading 14 dwords are
    // in scalar registers when the wave starts instead of behind a scalar-load round trip (the struct carries the rest;
    // n and the objective table's offsets follow from H and the compiled shape)
    const void* pZ, const void* pX0, const void* psmall, const void* pwslice, const void* pP, unsigned ppack, unsigned pR,
    unsigned pinvH, int pH, void* pjac, FxArgs a) {
    // `a` itself is never read here (see fx_late_args); a local block holds what the preloaded arguments say
    struct {
        const void *Z, *X0, *small, *wslice, *P;
        int tiles_per_wg, tiles_rem, zp_max, H, n, p_elems;
        unsigned R, invH;
        ObjOffsets oo;
        long long* dbg;
    } pa;
    pa.Z = pZ; pa.X0 = pX0; pa.small = psmall; pa.wslice = pwslice; pa.P = pP;
    pa.tiles_per_wg = (int)(ppack & 0xffu); pa.tiles_rem = (int)((ppack >> 8) & 0x3ffu);
    pa.zp_max = (int)((ppack >> 18) & 0x3ffu);
#ifdef NEMPC_EXP_NOOBJ   // timing experiment only
    const bool want_obj = false;
#else
    const bool want_obj = ((ppack >> 28) & 3u) != 0;     // f or grad asked for
#endif
    const bool want_rev = (ppack >> 30) & 1u;            // tiles or dense rows asked for
    const bool has_box = (ppack >> 31) & 1u;             // box rows follow the defect rows (m = 2 H NX)
    pa.R = pR; pa.invH = pinvH; pa.H = pH; pa.n = pH * (NX + NU);
    pa.oo = obj_offsets(pH, NX, NU);
    pa.p_elems = FUSE ? pa.oo.total : 0;
#ifdef NEMPC_STAMPS
    pa.dbg = a.dbg;
#else
    pa.dbg = nullptr;
    (void)a;
#endif
    COOP_WGSTAMP(pa.dbg, 0);
#if defined(NEMPC_STAMPS) && !defined(NEMPC_STAMPS_NO_FX)
    if (pa.dbg && threadIdx.x == 0 && blockIdx.x < 4096)
        pa.dbg[1024 + blockIdx.x * 16 + 15] = ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
#endif
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
    cx.Z = static_cast<const T*>(pa.Z);
    cx.X0 = static_cast<const T*>(pa.X0);
    cx.R = pa.R; cx.invH = pa.invH; cx.H = pa.H; cx.n = pa.n; cx.rev = want_rev;
    cx.jac = FUSE ? static_cast<T*>(pjac) : nullptr;
    cx.box = has_box;

    const int t_begin = blockIdx.x * pa.tiles_per_wg + ((int)blockIdx.x < pa.tiles_rem ? (int)blockIdx.x : pa.tiles_rem);
    const int t_end = t_begin + pa.tiles_per_wg + ((int)blockIdx.x < pa.tiles_rem ? 1 : 0);

    // ---- every global load of the prologue is issued before anything waits: inputs, small tables, weight slices.
    //      The staging registers always cover a full TPW-tile pass; a shorter pass just leaves rows unused.
    FxStage<T, TPW, NTHREADS, NCOL> sr;
    int t0 = t_begin;
    fx_stage_load<T, WP, NH, TPW, NX, NU, TPW>(cx, t0, tid, sr);
    FX_STAMP_PRO(pa.dbg, 1);
    constexpr int SMALL_VECS = (L::SMALL_END + VEC - 1) / VEC;
    constexpr int SMALL_PER_THREAD = (SMALL_VECS + NTHREADS - 1) / NTHREADS;
    vecT sm[SMALL_PER_THREAD];
    {
        const vecT* __restrict__ gs = static_cast<const vecT*>(pa.small);
#pragma unroll
        for (int u = 0; u < SMALL_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
            if (idx < SMALL_VECS) sm[u] = gs[idx];
        }
    }
    FX_STAMP_PRO(pa.dbg, 2);
    // fused evaluation: the objective.  A problem belongs to the workgroup whose tile range holds its first row.  WHO
    // evaluates it: the two workgroups of a CU do not interleave -- the one dispatched first (the first half of the grid)
    // runs its passes at nearly full speed and is done ~1.5 us before the other, which fills gaps and finishes last
    // (tools/diag_stamps.py).  The objective is a serial chain on a single wave per problem (0.3 us); in the workgroup
    // that finishes last it is on the launch's critical path wherever it stands (at its start it delayed the whole
    // pass: 1.3 us of a 16.4 us launch, measured by leaving it out).  So the first-half workgroups, which have the slack,
    // evaluate at their END the problems of BOTH members of a pair (i, i + ceil(grid / 2)); the second half evaluates
    // none.  Both ranges' variables -- two contiguous pieces of Z -- are fetched in the prologue and parked in LDS next
    // to the objective's table (evaluated from global memory it cost a round trip of its own wherever it stood).
    // Everything these loads need is among the preloaded arguments, so they go out without waiting for the argument block.
    // (the ranges are recomputed where they are used, at the end of the kernel: kept live across the passes they cost
    // scalar registers the pass code then spills)
    auto obj_range = [&](int side, unsigned& lo, unsigned& hi, int& nlds) {
        const unsigned half_lo = gridDim.x / 2u, half_hi = gridDim.x - half_lo;
        const unsigned Hh = (unsigned)pa.H;
        const unsigned wg = side == 0 ? blockIdx.x : blockIdx.x + half_hi;       // the partner's index
        const bool have = blockIdx.x < half_hi && (side == 0 || blockIdx.x < half_lo);
        const unsigned tb = wg * (unsigned)pa.tiles_per_wg + (wg < (unsigned)pa.tiles_rem ? wg : (unsigned)pa.tiles_rem);
        const unsigned te = tb + (unsigned)pa.tiles_per_wg + (wg < (unsigned)pa.tiles_rem ? 1u : 0u);
        unsigned r_lo = tb * 16u, r_hi = te * 16u;
        if (r_lo > pa.R) r_lo = pa.R;
        if (r_hi > pa.R) r_hi = pa.R;
        lo = pa.invH ? __umulhi(r_lo + Hh - 1, pa.invH) : r_lo;
        hi = have ? (pa.invH ? __umulhi(r_hi + Hh - 1, pa.invH) : r_hi) : lo;       // problems lo .. hi - 1
        nlds = (int)(hi - lo) < pa.zp_max ? (int)(hi - lo) : pa.zp_max;             // ... whose variables fit the LDS copy
    };
    constexpr int ZV_PER_THREAD = 2;
    T zv[2][ZV_PER_THREAD];
    int zcount[2] = {0, 0};             // elements of Z parked per side
    if constexpr (FUSE) {
        if (want_obj) {
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                unsigned lo, hi;
                int nl;
                obj_range(side, lo, hi, nl);
                zcount[side] = nl * pa.n;
                const T* __restrict__ zsrc = cx.Z + (size_t)lo * pa.n;
#pragma unroll
                for (int u = 0; u < ZV_PER_THREAD; ++u) {
                    const int idx = tid + u * NTHREADS;
                    zv[side][u] = idx < zcount[side] ? zsrc[idx] : T(0);
                }
            }
        }
    }
    // fused evaluation: the objective table rides along (-> LDS behind the layout, read at the very end); tables
    // beyond PV_PER_THREAD * NTHREADS elements (horizons past ~80 steps) are completed just before they are used
    constexpr int PV_PER_THREAD = 2;
    T pv[PV_PER_THREAD];
    if constexpr (FUSE) {
        const T* __restrict__ gp = static_cast<const T*>(pa.P);
#pragma unroll
        for (int u = 0; u < PV_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
            pv[u] = idx < pa.p_elems ? gp[idx] : T(0);
        }
    }
    FX_STAMP_PRO(pa.dbg, 3);
    constexpr int NFRAG = (NH - 1) * 2 * MT * 4 + 8;
    constexpr int NLOAD = (NFRAG + VEC - 1) / VEC;
    vecT wv[NLOAD];
    {
        const vecT* __restrict__ ws = static_cast<const vecT*>(pa.wslice) + (size_t)w * NLOAD * 64 + lane;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) wv[k] = ws[k * 64];
    }
    FX_STAMP_PASS(pa.dbg, 1);
    FX_STAMP_PRO(pa.dbg, 4);
    cx.dbg = pa.dbg;
    {
        vecT* ls = reinterpret_cast<vecT*>(lds + L::W0F);
#pragma unroll
        for (int u = 0; u < SMALL_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
            if (idx < SMALL_VECS) ls[idx] = sm[u];
        }
    }
    FX_STAMP_PRO(pa.dbg, 5);
    if constexpr (FUSE) {
#pragma unroll
        for (int u = 0; u < PV_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
            if (idx < pa.p_elems) lds[L::TOTAL + idx] = pv[u];
        }
    }
    const int p_pad = (pa.p_elems + 15) & ~15;
    if constexpr (FUSE) {
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int u = 0; u < ZV_PER_THREAD; ++u) {
                const int idx = tid + u * NTHREADS;
                if (idx < zcount[side]) lds[L::TOTAL + p_pad + side * FX_ZCOPY + idx] = zv[side][u];
            }
    }
    FX_STAMP_PASS(pa.dbg, 2);
    FX_STAMP_PRO(pa.dbg, 6);
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

    FX_STAMP_PASS(pa.dbg, 3);
    int parity = 0, xsel = 0;
    T* const in_base = lds + L::IN;
    fx_stage_store<T, WP, NH, TPW, NX, NU, TPW>(in_base, tid, sr);
    FX_STAMP_PRO(pa.dbg, 7);
    while (t0 < t_end) {
        const int t_cur = t0;
        const int n_cur = t_end - t0 < TPW ? t_end - t0 : TPW;
        t0 += n_cur;
        const bool more = t0 < t_end;
        if (more) fx_stage_load<T, WP, NH, TPW, NX, NU, TPW>(cx, t0, tid, sr);   // next pass's inputs, under this pass
#if NEMPC_FX_ZERO_EARLY
        if constexpr (FUSE) {
            if (cx.jac) {
                const unsigned r0 = (unsigned)t_cur * 16u;
                const int nrows = r0 + (unsigned)n_cur * 16u <= pa.R ? n_cur * 16 : (int)(pa.R - r0);
                fx_zero_rows<T, NX, NTHREADS>(cx.jac, r0, nrows, pa.n, pa.H, pa.invH, has_box, tid);
            }
        }
#endif
        lds_barrier();
        FX_STAMP_PASS(pa.dbg, 4);
        FX_STAMP_PRO(pa.dbg, 8);
        FX_STAMP_PRO(pa.dbg, 9);
        const T* in = in_base + parity * L::IN_SZ;
        T* const in_next = in_base + (parity ^ 1) * L::IN_SZ;
        if (n_cur == 1) fx_pass<T, WP, NH, TPW, NX, NU, 1, FUSE, ACT, GN>(cx, W, in, t_cur, tid, xsel, sr, more, in_next);
        if constexpr (TPW >= 2) { if (n_cur == 2) fx_pass<T, WP, NH, TPW, NX, NU, 2, FUSE, ACT, GN>(cx, W, in, t_cur, tid, xsel, sr, more, in_next); }
        if constexpr (TPW >= 3) { if (n_cur == 3) fx_pass<T, WP, NH, TPW, NX, NU, 3, FUSE, ACT, GN>(cx, W, in, t_cur, tid, xsel, sr, more, in_next); }
        parity ^= 1;
    }
    if constexpr (FUSE) {
        if (want_obj && blockIdx.x < gridDim.x - gridDim.x / 2u) {
            const FxArgsK ka = fx_late_args();
            T* const o_f = static_cast<T*>(ka->f);
            T* const o_grad = static_cast<T*>(ka->grad);
            unsigned ob_lo[2], ob_hi[2];
            int ob_n[2];
            obj_range(0, ob_lo[0], ob_hi[0], ob_n[0]);
            obj_range(1, ob_lo[1], ob_hi[1], ob_n[1]);
            // the LDS-resident problems of both ranges, one problem per wave (flat over the two ranges so that the
            // waves share them evenly)
            const int ntot = ob_n[0] + ob_n[1];
            for (int k = w; k < ntot; k += MT) {
                const int side = k < ob_n[0] ? 0 : 1, kk = k - (side ? ob_n[0] : 0);
                objective_row<T>((int)ob_lo[side] + kk, lane, pa.H, NX, NU, pa.oo, lds + L::TOTAL,
                                 lds + L::TOTAL + p_pad + side * FX_ZCOPY + kk * pa.n, o_f, o_grad);
            }
            // ---- problems beyond the LDS copy (long horizons, many problems per workgroup): from global memory, one
            //      problem per wave at a time (same routine, hence the same bits)
            if (ob_lo[0] + (unsigned)ob_n[0] < ob_hi[0] || ob_lo[1] + (unsigned)ob_n[1] < ob_hi[1]) {
                if (pa.p_elems > PV_PER_THREAD * NTHREADS) {
                    const T* __restrict__ gp = static_cast<const T*>(pa.P);
                    for (int i = tid + PV_PER_THREAD * NTHREADS; i < pa.p_elems; i += NTHREADS) lds[L::TOTAL + i] = gp[i];
                    __syncthreads();
                }
#pragma unroll
                for (int side = 0; side < 2; ++side)
                    for (unsigned b = ob_lo[side] + (unsigned)ob_n[side] + (unsigned)w; b < ob_hi[side]; b += MT)
                        objective_body<T>((int)b, lane, pa.H, NX, NU, pa.oo, lds + L::TOTAL, cx.Z, o_f, o_grad);
            }
        }
    }
    COOP_WGSTAMP(pa.dbg, 14);
    COOP_WGSTAMP_REAL(pa.dbg, 13);
}

}  // namespace nempc
