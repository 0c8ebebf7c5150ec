// CU-cooperative matrix-core row kernel (gfx950), register-resident weight slices.
//
// Why a second MFMA kernel: at the headline size (B*H = 20480 rows = 1280 tiles of 16 rows) the
// wave-per-tile kernel gives each of the chip's 1024 SIMDs 1.25 tiles on average but 2 in the worst
// case, and a lone wave serialises its own MFMA, tanh, staging and store phases (~40k cycles per
// tile of which 15.6k are MFMA).  Here a workgroup of MT = WP/16 waves (one per SIMD for WP = 64)
// shares every tile it owns: wave w computes feature block w (16 hidden units) of every layer for
// all of the workgroup's tiles.  Consequences:
//   * a wave only ever needs ITS OWN slice of every weight matrix (MT*4 fragments per hidden layer
//     and direction): the slice lives in registers for the whole kernel (82 VGPRs for fp64 2x64),
//     is fetched once from L2, and no LDS is spent on weights -- so two workgroups fit per CU and
//     the two waves that share a SIMD come from different workgroups, drift apart, and one's MFMAs
//     run under the other's tanh / LDS / store phases;
//   * the matrix work of a tile is split evenly over the SIMDs whatever the tile count;
//   * each weight fragment feeds one MFMA per tile of the pass.
//
//   hidden layer l   every wave publishes its block of a_{l-1} (4 accumulator registers = a B
//                    operand) to the LDS exchange buffer X, barrier, then accumulates its block of
//                    W_l^T a_{l-1} over all K from X (lane-linear ds_read_b64, conflict free).
//   skinny layers    (network output, and the last reverse step onto the nx+nu inputs) are K-split:
//                    wave w contracts its own block and writes a partial; the partials are summed in
//                    the per-stage reduction.
//   reverse sweep    identical structure with W instead of W^T, one cotangent per network output.
//   epilogue         defects and compact tiles; the dense (m,n) Jacobian is streamed by the assembly kernel
//                    (kernels_post.hip).  A fused in-kernel dense store pass was built and measured: its
//                    latency-bound gather/store loop cost 15-28k cycles per pass (B=1024: 35.1 us fused vs
//                    34.3 us with the separate 6 us HBM-rate assembly launch; B=16384: 391 vs 289 us) -- rejected.
//
// Same packed blob, same math and same outputs as kernels_mfma_impl.h; see there for the operand
// layout trick that keeps activations in registers without transposes.
#pragma once

#include "kernels_mfma_impl.h"

namespace nempc {

#ifdef NEMPC_STAMPS
#define COOP_STAMP(idx)                                                                   \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        unsigned long long _t;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (cx.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0) cx.dbg[(threadIdx.x >> 6) * 64 + (idx)] = (long long)_t; \
    } while (0)
#else
#define COOP_STAMP(idx) \
    do {                \
    } while (0)
#endif

struct CoopLayout {  // element offsets inside dynamic LDS
    int w0f;         // layer-0 fragments          ks * MT * 64
    int tail;        // seed | bias_l | biasL      (off.total - off.seed)
    int x;           // exchange buffer            2 halves of TPW * MT * 256
    int xhalf;
    int part;        // partials                   (1+nx) * TPW * MT * NR * 64
    int scratch;     // per-tile scratch           TPW * scratch_per_tile
    int rowinfo;     // (b, t) per tile row as int2  TPW * 16 * 2 ints (stored in T-sized slots)
    int total;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0): in this kernel
// that would stall the first barrier on the weight-slice loads (first needed a layer later) and every
// pass's last barrier on the acknowledgement of its global stores.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Cotangents swept together in the reverse pass: two when their accumulators are no bigger than those of a 2-tile
// fp64 pass (fp64: single-tile passes; fp32: passes of <= 2 tiles).  Two independent MFMA chains per wave and half the
// exchange barriers -- what the latency-bound cases (single-tile passes, 8-wave workgroups) lack.
template <typename T>
__host__ __device__ constexpr int coop_kg(int NT) {
    return NT * (int)sizeof(T) <= 8 ? 2 : 1;   // three at a time for fp32 single-tile passes measured no better (576 vs 568 us at C3)
}
// exchange-buffer slots (16-row activation sets) a pass may publish at once
template <typename T>
__host__ __device__ constexpr int coop_slots(int TPW) {
    int m = 0;
    for (int nt = 1; nt <= TPW; ++nt) m = nt * coop_kg<T>(nt) > m ? nt * coop_kg<T>(nt) : m;
    return m;
}

template <typename T>
__host__ __device__ inline int coop_nr(int nin) {
    return sizeof(T) == 8 ? (nin + 3) / 4 : 4;
}

template <typename T, int WP, int NH>
struct CoopWeights {  // this wave's slices, one element per lane per fragment
    static constexpr int MT = WP / 16;
    T wf[NH > 1 ? NH - 1 : 1][MT * 4];
    T wb[NH > 1 ? NH - 1 : 1][MT * 4];
    T wL[4];
    T w0b[4];
};

template <typename T>
struct CoopCtx {
    const T* w0f;   // LDS
    const T* seed;  // LDS tail: seed at 0, bias[l] at bias_off[l], biasL at biasL_off
    int bias_off[3], biasL_off;
    T* X;
    T* PART;
    T* SCR;
    int* RI;
    const T* Z;
    const T* X0;
    T* gout;
    T* tiles;
    int nx, nu, nin, H, n, m, NR, jsz, spt, nstages, ks, kind, box, xt_off, ex_off, ne, inv_nin;
    const T* extra;
    unsigned inv32_jrow, inv32_nx;
    int xhalf;                      // elements per half of the double-buffered exchange area  // ceil(2^32 / d) for d >= 2: item / d == umulhi(item, inv32) while item * d < 2^32
    size_t R;
    bool rk4;
    T DT;
    RowGather gk;                   // where a row's window inputs come from (nempc_internal.h)
    T* stage_out;                   // RK4 Hessian pipeline: per (row, stage) record [xi_s | J_s | dk_{s-1}], else null
    int stage_stride;
    long long* dbg;
};

// Inputs of a pass, fetched into registers first (so the loads can be issued ahead of the weight-slice
// loads: VMEM returns in order) and written to LDS later.  Items = (column, row): columns are the nx+nu
// network inputs followed by the nx current states x_t; rows are the pass's tile rows.
template <typename T, int MT, int TPW>
struct StageRegs {
    static constexpr int ITEMS = 2;   // covers (nx+nu)+nx <= 2*MT*64/(TPW*16) columns; wider problems stage directly
    T v[ITEMS];
    int b[ITEMS], t[ITEMS];
};

template <typename T, int MT, int TPW>
__device__ __forceinline__ void stage_load(const CoopCtx<T>& cx, int t0, int nrows, int tid, StageRegs<T, MT, TPW>& sr) {
    constexpr int ROWS = TPW * 16, NTHREADS = MT * 64;
    const int ncol = cx.nin + cx.nx + cx.ne;
#pragma unroll
    for (int it = 0; it < StageRegs<T, MT, TPW>::ITEMS; ++it) {
        const int item = tid + it * NTHREADS;
        const int col = item / ROWS, idx = item - col * ROWS;
        T v = T(0);
        int b = -1, t = 0;
        const size_t r = (size_t)t0 * 16 + idx;
        if (col < ncol && idx < nrows && r < cx.R) {
            b = (int)((unsigned)r / (unsigned)cx.H);
            t = (int)((unsigned)r - (unsigned)b * (unsigned)cx.H);
            const T* z = cx.Z + (size_t)b * cx.n;
            // plain models only (rolling windows stage directly, see the kernel body): keeps this early-issued
            // path, whose registers stay live across the weight-slice loads, as small as it was
            if (col < cx.nx) v = (t == 0) ? cx.X0[(size_t)b * cx.nx + col] : z[(t - 1) * cx.nx + col];
            else if (col < cx.nin) v = z[cx.H * cx.nx + t * cx.nu + (col - cx.nx)];
            else if (col < cx.nin + cx.nx) v = z[t * cx.nx + (col - cx.nin)];
            else v = cx.extra[r * cx.ne + (col - cx.nin - cx.nx)];
        }
        sr.v[it] = v; sr.b[it] = b; sr.t[it] = t;
    }
}

template <typename T, int MT, int TPW>
__device__ __forceinline__ void stage_store(const CoopCtx<T>& cx, int nrows, int tid, const StageRegs<T, MT, TPW>& sr) {
    constexpr int ROWS = TPW * 16, NTHREADS = MT * 64;
    const int ncol = cx.nin + cx.nx + cx.ne;
#pragma unroll
    for (int it = 0; it < StageRegs<T, MT, TPW>::ITEMS; ++it) {
        const int item = tid + it * NTHREADS;
        const int col = item / ROWS, idx = item - col * ROWS;
        if (col < ncol && idx < nrows) {
            T* tile = cx.SCR + (idx >> 4) * cx.spt;
            if (col < cx.nin) tile[(idx & 15) * cx.nin + col] = sr.v[it];
            else if (col < cx.nin + cx.nx) tile[cx.xt_off + (idx & 15) * cx.nx + (col - cx.nin)] = sr.v[it];
            else tile[cx.ex_off + (idx & 15) * cx.ne + (col - cx.nin - cx.nx)] = sr.v[it];
            if (col == 0) { cx.RI[2 * idx] = sr.b[it]; cx.RI[2 * idx + 1] = sr.t[it]; }
        }
    }
}

// Fallback for problems with more input columns than StageRegs holds: load and store in one go.
template <typename T, int MT, int TPW>
__device__ __forceinline__ void stage_direct(const CoopCtx<T>& cx, int t0, int nrows, int tid) {
    constexpr int ROWS = TPW * 16, NTHREADS = MT * 64;
    const int ncol = cx.nin + cx.nx + cx.ne;
    for (int item = tid; item < ncol * ROWS; item += NTHREADS) {
        const int col = item / ROWS, idx = item - col * ROWS;
        if (idx >= nrows) continue;
        T v = T(0);
        int b = -1, t = 0;
        const size_t r = (size_t)t0 * 16 + idx;
        if (r < cx.R) {
            b = (int)((unsigned)r / (unsigned)cx.H);
            t = (int)((unsigned)r - (unsigned)b * (unsigned)cx.H);
            const T* z = cx.Z + (size_t)b * cx.n;
            if (col < cx.nin) v = gather_input<T>(cx.gk, z, cx.X0, b, t, col);   // [x_{t-1} | u_t] or the rolling window
            else if (col < cx.nin + cx.nx) v = z[t * cx.nx + (col - cx.nin)];
            else v = cx.extra[r * cx.ne + (col - cx.nin - cx.nx)];
        }
        T* tile = cx.SCR + (idx >> 4) * cx.spt;
        if (col < cx.nin) tile[(idx & 15) * cx.nin + col] = v;
        else if (col < cx.nin + cx.nx) tile[cx.xt_off + (idx & 15) * cx.nx + (col - cx.nin)] = v;
        else tile[cx.ex_off + (idx & 15) * cx.ne + (col - cx.nin - cx.nx)] = v;
        if (col == 0) { cx.RI[2 * idx] = b; cx.RI[2 * idx + 1] = t; }
    }
}

// One pass over NT (compile-time) tiles starting at tile t0.  NT is a template parameter on purpose:
// with a runtime tile count every per-tile MFMA sat in its own basic block and hipcc copied the whole
// accumulator set through AGPRs at each join (6,500 v_accvgpr_* moves, 10x slower reverse sweep).
template <typename T, int WP, int NH, int NT, bool SR>
__device__ __forceinline__ void coop_pass(const CoopCtx<T>& cx, const CoopWeights<T, WP, NH>& W, int t0, int tid) {
    using Ops = MfmaOps<T>;
    using V4 = typename Ops::V4;
    constexpr int MT = WP / 16;
    constexpr int NTHREADS = MT * 64;
    const int lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    T* X = cx.X;
    int xsel = 0;
    T* PART = cx.PART;
    T* SCR = cx.SCR;
    const T* __restrict__ Z = cx.Z;
    const T* __restrict__ X0 = cx.X0;
    const int nx = cx.nx, nu = cx.nu, nin = cx.nin, H = cx.H, n = cx.n, NR = cx.NR, jsz = cx.jsz, spt = cx.spt;
    const size_t R = cx.R;
    const bool rk4 = cx.rk4;
    const T DT = cx.DT;

    int* RI = cx.RI;
    COOP_STAMP(3);

    for (int stage = 0; stage < cx.nstages; ++stage) {
        const T cdt = (stage == 0) ? T(0) : ((stage == 3) ? DT : T(0.5) * DT);
        V4 a[NH][NT];
        // ---- layer 0, this wave's feature block
        {
            const T* bias = cx.seed + cx.bias_off[0] + w * 16;
            V4 b0;
#pragma unroll
            for (int r = 0; r < 4; ++r) b0[r] = bias[r * 4 + q];
#pragma unroll
            for (int j = 0; j < NT; ++j) a[0][j] = b0;
            for (int ks = 0; ks < cx.ks; ++ks) {
                const T wfrag = cx.w0f[(ks * MT + w) * 64 + lane];
                const int d = 4 * ks + q;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const T* s_xi0 = SCR + j * spt;
                    T v = T(0);
                    if (d < nin) {
                        v = s_xi0[c * nin + d];
                        if (stage > 0 && d < nx) v = fma(cdt, s_xi0[16 * nin + c * nx + d], v);
                        // every wave forms the same stage input; wave 0 records it for the Hessian pipeline
                        if (SR && w == 0 && RI[2 * (j * 16 + c)] >= 0)
                            cx.stage_out[(((size_t)t0 + j) * 16 + c) * 4 * cx.stage_stride + (size_t)stage * cx.stage_stride + d] = v;
                    } else if (d < nin + cx.ne) {
                        v = s_xi0[cx.ex_off + c * cx.ne + (d - nin)];
                    }
                    a[0][j] = Ops::mma(wfrag, v, a[0][j]);
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[0][j][r] = Ops::tanh_(a[0][j][r]);
        }
        COOP_STAMP(4);
        // ---- hidden-to-hidden layers.  The exchange buffer has two halves used alternately: a wave may
        // publish exchange e+1 while a slower wave still reads exchange e; the barrier of e+1 then fences
        // the readers of e before anyone writes e+2 into the same half.
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            X = cx.X + (xsel & 1) * cx.xhalf;
            ++xsel;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[((j * MT + w) * 4 + r) * 64 + lane] = a[l - 1][j][r];
            lds_barrier();
            const T* bias = cx.seed + cx.bias_off[l] + w * 16;
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
        COOP_STAMP(5);
        // ---- network output: K-split partial over this wave's block
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            V4 pf = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int r = 0; r < 4; ++r) pf = Ops::mma(W.wL[r], a[NH - 1][j][r], pf);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < NR) PART[(((0 * NT + j) * MT + w) * NR + r) * 64 + lane] = pf[r];
        }
#pragma unroll
        for (int l = 0; l < NH; ++l)
#pragma unroll
            for (int j = 0; j < NT; ++j) a[l][j] = T(1) - a[l][j] * a[l][j];

        COOP_STAMP(6);
        // ---- reverse sweep: KG cotangents (network outputs) at a time; an odd leftover is swept twice, its second copy
        //      is not stored
        constexpr int KG = coop_kg<T>(NT);
        for (int k0 = 0; k0 < nx; k0 += KG) {
            V4 cv[KG][NT];
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const int k = k0 + g < nx ? k0 + g : nx - 1;
                const T* seed = cx.seed + k * MT * 16 + w * 16;
                V4 sd;
#pragma unroll
                for (int r = 0; r < 4; ++r) sd[r] = seed[r * 4 + q];
#pragma unroll
                for (int j = 0; j < NT; ++j) cv[g][j] = sd * a[NH - 1][j];
            }
#pragma unroll
            for (int l = NH - 1; l >= 1; --l) {
                X = cx.X + (xsel & 1) * cx.xhalf;
                ++xsel;
#pragma unroll
                for (int g = 0; g < KG; ++g)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[(((g * NT + j) * MT + w) * 4 + r) * 64 + lane] = cv[g][j][r];
                lds_barrier();
                V4 cn[KG][NT];
#pragma unroll
                for (int g = 0; g < KG; ++g)
#pragma unroll
                    for (int j = 0; j < NT; ++j) cn[g][j] = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int g = 0; g < KG; ++g)
#pragma unroll
                            for (int j = 0; j < NT; ++j)
                                cn[g][j] = Ops::mma(W.wb[l - 1][mt * 4 + r],
                                                    X[(((g * NT + j) * MT + mt) * 4 + r) * 64 + lane], cn[g][j]);
#pragma unroll
                for (int g = 0; g < KG; ++g)
#pragma unroll
                    for (int j = 0; j < NT; ++j) cv[g][j] = cn[g][j] * a[l - 1][j];
            }
#pragma unroll
            for (int g = 0; g < KG; ++g)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    V4 pj = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) pj = Ops::mma(W.w0b[r], cv[g][j][r], pj);
                    if (k0 + g < nx) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r < NR) PART[((((1 + k0 + g) * NT + j) * MT + w) * NR + r) * 64 + lane] = pj[r];
                    }
                }
        }
        COOP_STAMP(7);
        lds_barrier();
        COOP_STAMP(8);

        // ---- reduce the K-split partials: f -> s_k[cc][o], J -> s_J[cc][k][d]; items = (column, row), flat
        {
            constexpr int ROWS = NT * 16;
            const int ncol = nx + nx * nin;
            for (int item = tid; item < ncol * ROWS; item += NTHREADS) {
                const int col = item / ROWS, idx = item - col * ROWS;   // ROWS is a compile-time constant
                const int j = idx >> 4, cc = idx & 15;
                const bool isf = col < nx;
                const int kd = isf ? 0 : col - nx;
                const int k = (kd * cx.inv_nin) >> 16;
                const int dsel = isf ? col : kd - k * nin;               // output index (f) / input index (J)
                const int slot = isf ? 0 : 1 + k;
                const int qq = sizeof(T) == 8 ? (dsel & 3) : (dsel >> 2), rr = sizeof(T) == 8 ? (dsel >> 2) : (dsel & 3);
                T v = isf ? cx.seed[cx.biasL_off + rr * 4 + qq] : T(0);
#pragma unroll
                for (int ww = 0; ww < MT; ++ww) v += PART[(((slot * NT + j) * MT + ww) * NR + rr) * 64 + qq * 16 + cc];
                if (isf) SCR[j * spt + 16 * nin + cc * nx + col] = v;
                else SCR[j * spt + 16 * nin + 2 * 16 * nx + cc * nx * nin + kd] = v;
            }
        }
        lds_barrier();

        // stage record for the Hessian pipeline: this stage's Jacobian and the chain Jacobian it was entered with
        if (SR) {
            const int jn = nx * nin;
            for (int e = tid; e < NT * jsz; e += NTHREADS) {
                const int j = e / jsz, e2 = e - j * jsz;
                const int cc = e2 / jn, rem2 = e2 - cc * jn;
                if (RI[2 * (j * 16 + cc)] >= 0) {
                    const T* sj = SCR + j * spt + 16 * nin + 2 * 16 * nx;
                    T* rec = cx.stage_out + ((((size_t)t0 + j) * 16 + cc) * 4 + stage) * cx.stage_stride + nin;
                    rec[rem2] = sj[e2];
                    rec[jn + rem2] = stage > 0 ? sj[jsz + e2] : T(0);
                }
            }
        }
        // ---- RK4 chain rule on the per-tile scratch (rk4.py:147-159)
        if (rk4) {
            if (stage == 0) {
                for (int e = tid; e < NT * jsz; e += NTHREADS) {
                    const int j = e / jsz, e2 = e - j * jsz;
                    T* sj = SCR + j * spt + 16 * nin + 2 * 16 * nx;
                    const T v = sj[e2];
                    sj[jsz + e2] = v;
                    sj[2 * jsz + e2] = v;
                }
                for (int e = tid; e < NT * 16 * nx; e += NTHREADS) {
                    const int j = e / (16 * nx), e2 = e - j * 16 * nx;
                    T* sk = SCR + j * spt + 16 * nin;
                    sk[16 * nx + e2] = sk[e2];
                }
            } else {
                const T wgt = (stage == 3) ? T(1) : T(2);
                for (int e = tid; e < NT * jsz; e += NTHREADS) {
                    const int j = e / jsz, e2 = e - j * jsz;
                    const int cc = e2 / (nx * nin), rem2 = e2 - cc * nx * nin;
                    const int i = rem2 / nin, d = rem2 - i * nin;
                    const T* sj = SCR + j * spt + 16 * nin + 2 * 16 * nx;
                    T v = T(0);
                    for (int e3 = 0; e3 < nx; ++e3)
                        v = fma(sj[(cc * nx + i) * nin + e3], sj[jsz + (cc * nx + e3) * nin + d], v);
                    SCR[j * spt + 16 * nin + 2 * 16 * nx + 3 * jsz + e2] = fma(cdt, v, sj[e2]);
                }
                lds_barrier();
                for (int e = tid; e < NT * jsz; e += NTHREADS) {
                    const int j = e / jsz, e2 = e - j * jsz;
                    T* sj = SCR + j * spt + 16 * nin + 2 * 16 * nx;
                    const T v = sj[3 * jsz + e2];
                    sj[jsz + e2] = v;
                    sj[2 * jsz + e2] = fma(wgt, v, sj[2 * jsz + e2]);
                }
                for (int e = tid; e < NT * 16 * nx; e += NTHREADS) {
                    const int j = e / (16 * nx), e2 = e - j * 16 * nx;
                    T* sk = SCR + j * spt + 16 * nin;
                    sk[16 * nx + e2] = fma(wgt, sk[e2], sk[16 * nx + e2]);
                }
            }
            lds_barrier();
        }
    }

    COOP_STAMP(9);
    // ---- outputs: compact tiles (16 rows contiguous in memory) and defects
    const T s6 = DT / T(6);
    // tiles: the pass's NT*16 rows are contiguous in memory -> lanes run over the flat element index (coalesced)
    const int jrow = nx * nin;
    for (int item = tid; item < NT * jsz; item += NTHREADS) {
        const int idx = jrow == 1 ? item : (int)__umulhi((unsigned)item, cx.inv32_jrow), kd = item - idx * jrow;
        if (RI[2 * idx] >= 0) {
            const int i = (kd * cx.inv_nin) >> 16, d = kd - i * nin;
            const T* sj = SCR + (idx >> 4) * spt + 16 * nin + 2 * 16 * nx + (idx & 15) * jrow;
            const T ident = (d == cx.gk.xcur + i && (rk4 || cx.kind == NEMPC_DISCRET)) ? T(1) : T(0);
            cx.tiles[(size_t)t0 * 16 * jrow + item] = (rk4 ? s6 * sj[2 * jsz + kd] : sj[kd]) + ident;
        }
    }
    // defects: lanes run over (row, state) with the state fastest -> contiguous inside a problem
    for (int item = tid; item < NT * 16 * nx; item += NTHREADS) {
        const int idx = nx == 1 ? item : (int)__umulhi((unsigned)item, cx.inv32_nx), i = item - idx * nx;
        const int b = RI[2 * idx], t = RI[2 * idx + 1];
        if (b >= 0) {
            const int cc = idx & 15;
            const T* s_xi0 = SCR + (idx >> 4) * spt;
            const T* sk = s_xi0 + 16 * nin;
            const T xp = s_xi0[cc * nin + cx.gk.xcur + i];
            T phi;
            if (rk4) phi = xp + s6 * sk[16 * nx + cc * nx + i];
            else phi = (cx.kind == NEMPC_DISCRET ? xp : T(0)) + sk[cc * nx + i];
            const T xt = s_xi0[cx.xt_off + cc * nx + i];
            cx.gout[(size_t)b * cx.m + t * nx + i] = phi - xt;
            if (cx.box) cx.gout[(size_t)b * cx.m + (size_t)H * nx + t * nx + i] = xt;
        }
    }
    lds_barrier();
    COOP_STAMP(11);
}

// SR: also write the per-(row, stage) records of the RK4 Hessian pipeline (its own instantiation: the extra stores and
// their address arithmetic cost the plain kernel 0.3 - 0.9 us when they are only branched around)
template <typename T, int WP, int NH, int TPW, bool SR = false>
__global__ __launch_bounds__((WP / 16) * 64, 2) void rows_coop_kernel(MfmaParams p, CoopLayout lay) {
    constexpr int MT = WP / 16;
    constexpr int NTHREADS = MT * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const T* __restrict__ gblob = static_cast<const T*>(p.blob);
    NEMPC_STAMP(0);

    CoopCtx<T> cx;
    cx.w0f = lds + lay.w0f;
    cx.seed = lds + lay.tail;
    for (int l = 0; l < 3; ++l) cx.bias_off[l] = p.off.bias[l] - p.off.seed;
    cx.biasL_off = p.off.biasL - p.off.seed;
    cx.X = lds + lay.x;
    cx.xhalf = lay.xhalf;
    cx.PART = lds + lay.part;
    cx.SCR = lds + lay.scratch;
    cx.RI = reinterpret_cast<int*>(lds + lay.rowinfo);
    cx.Z = static_cast<const T*>(p.Z);
    cx.X0 = static_cast<const T*>(p.X0);
    cx.gout = static_cast<T*>(p.g);
    cx.tiles = static_cast<T*>(p.tiles);
    cx.nx = p.nx; cx.nu = p.nu; cx.nin = p.nin; cx.H = p.H; cx.n = p.gk.n; cx.m = p.m;
    cx.gk = p.gk;
    cx.stage_out = static_cast<T*>(p.stage_out); cx.stage_stride = p.stage_stride;
    cx.NR = coop_nr<T>(p.nin);
    cx.jsz = 16 * p.nx * p.nin;
    cx.spt = p.scratch_per_wave;
    cx.xt_off = 16 * p.nin + 2 * 16 * p.nx + (p.kind == NEMPC_RK4 ? 4 : 1) * cx.jsz;  // after the wave-tile kernel's carve-up
    cx.ex_off = cx.xt_off + 16 * p.nx;                    // [16][ne] extra inputs
    cx.ne = p.ne;
    cx.extra = static_cast<const T*>(p.extra);
    cx.inv32_jrow = (unsigned)((0x100000000ull + (unsigned)(p.nx * p.nin) - 1) / (unsigned)(p.nx * p.nin));
    cx.inv32_nx = (unsigned)((0x100000000ull + (unsigned)p.nx - 1) / (unsigned)p.nx);
    cx.inv_nin = (65536 + p.nin - 1) / p.nin;              // kd / nin == (kd * inv_nin) >> 16 for kd < 256
    cx.rk4 = p.kind == NEMPC_RK4;
    cx.nstages = cx.rk4 ? 4 : 1;
    cx.ks = p.ks; cx.kind = p.kind; cx.box = p.box;
    cx.R = (size_t)p.B * p.H;
    cx.DT = (T)p.DT;
    cx.dbg = p.dbg;

    // contiguous, balanced range of tiles for this workgroup (quotient / remainder computed on the host)
    const int t_begin = blockIdx.x * p.tiles_per_wg + ((int)blockIdx.x < p.tiles_rem ? (int)blockIdx.x : p.tiles_rem);
    const int t_end = t_begin + p.tiles_per_wg + ((int)blockIdx.x < p.tiles_rem ? 1 : 0);

    // first pass's inputs: loads issued BEFORE the weight-slice loads, consumed (LDS stores) after them
    StageRegs<T, MT, TPW> sr;
    int t0 = t_begin;
    int nact = t_end - t0 < TPW ? t_end - t0 : TPW;
    const bool early = cx.gk.w == 1 && (cx.nin + cx.nx + cx.ne) * TPW * 16 <= StageRegs<T, MT, TPW>::ITEMS * NTHREADS;
    if (early) stage_load<T, MT, TPW>(cx, t0, nact * 16, tid, sr);
    // small tables -> LDS ...
    copy_blob_to_lds<T>(gblob + p.off.w0f, lds + lay.w0f, p.ks * MT * 64, tid, NTHREADS);
    copy_blob_to_lds<T>(gblob + p.off.seed, lds + lay.tail, p.off.total - p.off.seed, tid, NTHREADS);
    // ... then this wave's weight slices -> registers (kept for every pass); first needed by hidden layer 1,
    // so their latency hides under layer 0 of the first pass
    CoopWeights<T, WP, NH> W;
#pragma unroll
    for (int l = 1; l < NH; ++l)
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) {
#ifdef NEMPC_EXP_NOWEIGHTS   // timing experiment only: how much of the prologue is the weight fetch
            W.wf[l - 1][i] = T(1e-3) * T(lane + i);
            W.wb[l - 1][i] = T(1e-3) * T(lane - i);
#else
            W.wf[l - 1][i] = gblob[p.off.wf[l] + (i * MT + w) * 64 + lane];
            W.wb[l - 1][i] = gblob[p.off.wb[l] + (i * MT + w) * 64 + lane];
#endif
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        W.wL[r] = gblob[p.off.wLf + (w * 4 + r) * 64 + lane];
        W.w0b[r] = gblob[p.off.w0b + (w * 4 + r) * 64 + lane];
    }
    NEMPC_STAMP(1);

    while (t0 < t_end) {
        if (early) stage_store<T, MT, TPW>(cx, nact * 16, tid, sr);
        else stage_direct<T, MT, TPW>(cx, t0, nact * 16, tid);
        lds_barrier();
        if (nact == 1) coop_pass<T, WP, NH, 1, SR>(cx, W, t0, tid);
        if constexpr (TPW >= 2) { if (nact == 2) coop_pass<T, WP, NH, 2, SR>(cx, W, t0, tid); }
        if constexpr (TPW >= 3) { if (nact == 3) coop_pass<T, WP, NH, 3, SR>(cx, W, t0, tid); }
        if constexpr (TPW >= 4) { if (nact == 4) coop_pass<T, WP, NH, 4, SR>(cx, W, t0, tid); }
        t0 += nact;
#ifdef NEMPC_STAMPS
        cx.dbg = nullptr;   // diagnostic build: keep the FIRST pass's stamps
#endif
        if (t0 < t_end) {
            nact = t_end - t0 < TPW ? t_end - t0 : TPW;
            if (early) stage_load<T, MT, TPW>(cx, t0, nact * 16, tid, sr);
        }
    }
    NEMPC_STAMP(12);
}

}  // namespace nempc
