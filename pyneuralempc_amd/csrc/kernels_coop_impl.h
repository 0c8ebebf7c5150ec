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
#include "kernels_obj_impl.h"

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
// per-workgroup timeline (wave 0, lane 0 of EVERY workgroup): word idx of the workgroup's 16-word record = shader
// clock (s_memtime: local and cheap; s_memrealtime goes out to the chip's timestamp unit and costs the wave ~1 us per
// read, which is why it is taken once, at exit, into word 13 to put the workgroups on a common axis);
// word 15 = XCC id << 32 | HW_ID
#define COOP_WGSTAMP(dbgp, idx)                                                              \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if ((dbgp) && threadIdx.x == 0 && blockIdx.x < 4096)                                 \
            (dbgp)[1024 + blockIdx.x * 16 + (idx)] = (long long)__builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#define COOP_WGSTAMP_REAL(dbgp, idx)                                                         \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if ((dbgp) && threadIdx.x == 0 && blockIdx.x < 4096)                                 \
            (dbgp)[1024 + blockIdx.x * 16 + (idx)] = (long long)__builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#else
#define COOP_STAMP(idx) \
    do {                \
    } while (0)
#define COOP_WGSTAMP(dbgp, idx) \
    do {                        \
    } while (0)
#define COOP_WGSTAMP_REAL(dbgp, idx) \
    do {                             \
    } while (0)
#endif

#ifdef NEMPC_STAMPS
#define NEMPC_STAMP_A(idx)                                                                 \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        unsigned long long _t;                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (a.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0) a.dbg[(threadIdx.x >> 6) * 64 + (idx)] = (long long)_t; \
    } while (0)
#else
#define NEMPC_STAMP_A(idx) \
    do {                   \
    } while (0)
#endif

// A/B switches of the prologue / pass-boundary experiments (tools/build_variant.py -D...)
#ifndef NEMPC_COOP_PROLOGUE_BARRIER
#define NEMPC_COOP_PROLOGUE_BARRIER 0
#endif
#ifndef NEMPC_COOP_MIDSTAGE
#define NEMPC_COOP_MIDSTAGE 1
#endif

// A pointer the whole wave agrees on, pinned to scalar registers (the "s" operand of the stores below must not be left to
// the compiler's uniformity analysis)
__device__ __forceinline__ const char* fx_uniform_ptr(const char* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// A/B switch of the fused dense output (tools/build_variant.py -D...)
#ifndef NEMPC_FX_NZ_PLAIN
#define NEMPC_FX_NZ_PLAIN 0        // 1: the non-zero entries as plain stores (dirty in L2 until the end-of-kernel write-back)
#endif
#if NEMPC_FX_NZ_PLAIN
#define NEMPC_FX_WT ""
#else
#define NEMPC_FX_WT " sc0 sc1"
#endif
// Write-through stores (sc0 sc1): the data leaves for memory as it is issued instead of sitting dirty in L2 until the
// end-of-kernel write-back (C2, B=1024: whole evaluation 21.9 -> 20.3 us when the dense rows were first fused in).
__device__ __forceinline__ void fx_store_wt(double* p, double v) {
    asm volatile("global_store_dwordx2 %0, %1, off" NEMPC_FX_WT ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void fx_store_wt(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off" NEMPC_FX_WT ::"v"(p), "v"(v) : "memory");
}
// (gfx9 store-data hazard: a vector instruction that overwrites the data registers of a store of more than 64 bits needs two
// wait states behind it.  The compiler inserts them for its own stores, but it cannot see into an asm statement -- the
// sparse contract's second store had its pointer computed into the first one's data registers in the very next
// instruction, and lanes 12..15 of every 16 stored the pointer.  Every 16-byte store written as asm carries its own
// `s_nop 1`; _isa.py checks the assembled text for the pattern.)
__device__ __forceinline__ void fx_store_wt2(double* p, double v0, double v1) {      // p 16-byte aligned
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 v = {v0, v1};
    asm volatile("global_store_dwordx4 %0, %1, off" NEMPC_FX_WT "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// Fused dense Jacobian, part one: the BACKGROUND (see fx_zero_rows in kernels_coopfx_impl.h, the compiled-shape twin of
// this routine, for why and for the ordering argument).  The structural zeros of a pass's dense rows are streamed from
// registers at the start of the pass as flat runs of 16-byte write-through stores; the pass's outputs overwrite the few
// non-zeros.  The host launches this form only when a problem's block of rows is a whole number of 16-byte vectors
// (nx * n * sizeof(T) divisible by 16) and 16-byte aligned.
template <typename T, int NTHREADS>
__device__ __forceinline__ void coop_zero_rows(T* o_jac, unsigned r0, int nrows, int nx, int n, int H, unsigned invH, bool box, int tid) {
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    const vecT zero = {};
    const size_t row_bytes = (size_t)n * sizeof(T);
    const int drv = nrows * nx;                     // dense rows of the pass that exist
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
        run(fx_uniform_ptr(reinterpret_cast<const char*>(o_jac) + (size_t)r0 * nx * row_bytes), (int)((size_t)drv * row_bytes / 16));
    } else {
        const int HNX = H * nx;
        for (int s0 = 0; s0 < drv;) {
            const unsigned r = r0 + (unsigned)(s0 / nx);
            const unsigned b = invH ? __umulhi(r, invH) : r;
            const int k = (int)(r - b * (unsigned)H) * nx;              // first dense row of the piece within its block
            const int len = drv - s0 < HNX - k ? drv - s0 : HNX - k;    // rows up to the end of the problem / pass
            for (int kind = 0; kind < 2; ++kind)
                run(fx_uniform_ptr(reinterpret_cast<const char*>(o_jac) + ((size_t)b * (2 * HNX) + (size_t)(kind * HNX + k)) * row_bytes),
                    (int)((size_t)len * row_bytes / 16));
            s0 += len;
        }
    }
}

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
#ifdef NEMPC_EXP_KG1
    return 1;
#endif
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
    int bias_off[NEMPC_MFMA_MAX_HIDDEN], biasL_off;
    T* X;
    T* PART;
    T* SCR;
    int* RI;
    const T* Z;
    const T* X0;
    T* gout;
    T* tiles;
    T* jac;                         // fused dense Jacobian (B, m, n) or null (plain models only)
    T* sp;                          // band-pattern values (B, sp_nnz) in nempc_jac_structure order or null (plain models only)
    int sp_nnz;
    int nx, nu, nin, H, n, m, NR, jsz, spt, nstages, ks, kind, box, xt_off, ex_off, ne, inv_nin;
    const T* extra;
    unsigned inv32_jrow, inv32_nx;
    int xhalf;                      // elements per half of the double-buffered exchange area  // ceil(2^32 / d) for d >= 2: item / d == umulhi(item, inv32) while item * d < 2^32
    unsigned invH;                  // ceil(2^32 / H) (0 for H == 1): row / H == umulhi(row, invH)
    size_t R;
    bool rk4;
    T DT;
    RowGather gk;                   // where a row's window inputs come from (nempc_internal.h)
    T* stage_out;                   // RK4 Hessian pipeline: per (row, stage) record [xi_s | J_s | dk_{s-1}], else null
    int stage_stride;
    long long* dbg;
    ActSpec acts;                   // hidden layers' activations (NEMPC_ACT_RUNTIME instantiations)
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
            b = cx.invH ? (int)__umulhi((unsigned)r, cx.invH) : (int)r;
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
__device__ __forceinline__ void stage_store(const CoopCtx<T>& cx, T* SCRdst, int* RIdst, int nrows, int tid,
                                            const StageRegs<T, MT, TPW>& sr) {
    constexpr int ROWS = TPW * 16, NTHREADS = MT * 64;
    const int ncol = cx.nin + cx.nx + cx.ne;
#pragma unroll
    for (int it = 0; it < StageRegs<T, MT, TPW>::ITEMS; ++it) {
        const int item = tid + it * NTHREADS;
        const int col = item / ROWS, idx = item - col * ROWS;
        if (col < ncol && idx < nrows) {
            T* tile = SCRdst + (idx >> 4) * cx.spt;
            if (col < cx.nin) tile[(idx & 15) * cx.nin + col] = sr.v[it];
            else if (col < cx.nin + cx.nx) tile[cx.xt_off + (idx & 15) * cx.nx + (col - cx.nin)] = sr.v[it];
            else tile[cx.ex_off + (idx & 15) * cx.ne + (col - cx.nin - cx.nx)] = sr.v[it];
            if (col == 0) { RIdst[2 * idx] = sr.b[it]; RIdst[2 * idx + 1] = sr.t[it]; }
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
            b = cx.invH ? (int)__umulhi((unsigned)r, cx.invH) : (int)r;
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
// `nxt` (valid when has_nxt): the NEXT pass's inputs, already fetched into registers; they are written to the other scratch
// buffer (SCRnext / RInext) just before this pass's first global store.  Placed there for the vmcnt counter: stores
// count in it too and retire in order, so a wait for those loads issued after the epilogue's (runtime-many) stores
// degenerates to vmcnt(0) and sat out the stores' acknowledgement, 1.4 us per pass boundary.
// NXc / NUc > 0: the problem's dims as compile-time constants (plain models, window 1): every index division of the
// reduction, chain-rule and output phases folds, their inner loops unroll -- the same source, a second instantiation
// for a shape that is worth it (BASELINE configs[2]: 6 states, 3 controls).
template <typename T, int WP, int NH, int NT, bool SR, int TPW, int ACT, int NXc = 0, int NUc = 0>
__device__ __forceinline__ void coop_pass(const CoopCtx<T>& cx, const CoopWeights<T, WP, NH>& W, int t0, int tid,
                                          const StageRegs<T, WP / 16, TPW>& nxt, bool has_nxt, T* SCRnext, int* RInext,
                                          int nrows_next) {
    using Ops = MfmaOps<T>;
    using A = ActL<T, ACT>;
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
    constexpr bool FX = NXc > 0;
    const int nx = FX ? NXc : cx.nx, nu = FX ? NUc : cx.nu, nin = FX ? NXc + NUc : cx.nin;
    const int H = cx.H, n = cx.n, spt = cx.spt;
    const int NR = FX ? (sizeof(T) == 8 ? (NXc + NUc + 3) / 4 : 4) : cx.NR;
    const int jsz = FX ? 16 * NXc * (NXc + NUc) : cx.jsz;
    auto div_nin = [&](int kd) { return FX ? kd / nin : (kd * cx.inv_nin) >> 16; };      // kd < 256
    const size_t R = cx.R;
    const bool rk4 = cx.rk4;
    const T DT = cx.DT;

    int* RI = cx.RI;
    COOP_STAMP(48);

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
                for (int r = 0; r < 4; ++r) a[0][j][r] = A::f(a[0][j][r], cx.acts, 0);
        }
        COOP_STAMP(12 * stage + 4);
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
                for (int r = 0; r < 4; ++r) a[l][j][r] = A::f(a[l][j][r], cx.acts, l);
        }
        COOP_STAMP(12 * stage + 5);
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
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[l][j][r] = A::d1(a[l][j][r], cx.acts, l);

        COOP_STAMP(12 * stage + 6);
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
        COOP_STAMP(12 * stage + 7);
        lds_barrier();
        COOP_STAMP(12 * stage + 8);
        // ---- reduce the K-split partials: f -> s_k[cc][o], J -> s_J[cc][k][d]; items = (column, row), flat
        {
            constexpr int ROWS = NT * 16;
            const int ncol = nx + nx * nin;
            for (int item = tid; item < ncol * ROWS; item += NTHREADS) {
                const int col = item / ROWS, idx = item - col * ROWS;   // ROWS is a compile-time constant
                const int j = idx >> 4, cc = idx & 15;
                const bool isf = col < nx;
                const int kd = isf ? 0 : col - nx;
                const int k = div_nin(kd);
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
        COOP_STAMP(12 * stage + 10);

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
#pragma unroll
                    for (int e3 = 0; e3 < (FX ? NXc : nx); ++e3)
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
        COOP_STAMP(12 * stage + 11);
    }

    COOP_STAMP(49);
    // fused dense Jacobian (plain models): the background zeros of the pass's rows were streamed at its start
    // (coop_zero_rows); every wave waits for its own stores' acknowledgements, the barrier covers the workgroup, then the
    // non-zeros go over them with the compact outputs below (see fx_zero_rows in kernels_coopfx_impl.h for the ordering)
    T* const jac = cx.jac;
    if (jac) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
    if (has_nxt) stage_store<T, WP / 16, TPW>(cx, SCRnext, RInext, nrows_next, tid, nxt);
    // ---- outputs: compact tiles (16 rows contiguous in memory) and defects
    const T s6 = DT / T(6);
    // tiles: the pass's NT*16 rows are contiguous in memory -> lanes run over the flat element index (coalesced)
    const int jrow = nx * nin;
    for (int item = tid; item < NT * jsz; item += NTHREADS) {
        const int idx = FX ? item / jrow : (jrow == 1 ? item : (int)__umulhi((unsigned)item, cx.inv32_jrow)), kd = item - idx * jrow;
        if (RI[2 * idx] >= 0) {
            const int i = div_nin(kd), d = kd - i * nin;
            const T* sj = SCR + (idx >> 4) * spt + 16 * nin + 2 * 16 * nx + (idx & 15) * jrow;
            const T ident = (d == cx.gk.xcur + i && (rk4 || cx.kind == NEMPC_DISCRET)) ? T(1) : T(0);
            const T v = (rk4 ? s6 * sj[2 * jsz + kd] : sj[kd]) + ident;
            if (cx.tiles) cx.tiles[(size_t)t0 * 16 * jrow + item] = v;
            if (jac) {
                // dense row (t, i) of problem b: the state block at x_{t-1} (t >= 1; x0 has no column), the control block at u_t
                const int b = RI[2 * idx], t = RI[2 * idx + 1];
                const int col = d < nx ? (t - 1) * nx + d : H * nx + t * nu + (d - nx);
                if (d >= nx || t >= 1) fx_store_wt(jac + ((size_t)b * cx.m + (size_t)(t * nx + i)) * (size_t)n + col, v);
            }
            if (cx.sp) {
                // sparse contract: entry (t, i, d) of the band pattern -- per dense row [state block (t >= 1) | -1 | control
                // block], rows in order, row 0's blocks without the state part (nempc_jac_structure)
                const int b = RI[2 * idx], t = RI[2 * idx + 1];
                const int row0 = 1 + nu, rowt = nx + 1 + nu;
                const int base = t >= 1 ? nx * row0 + ((t - 1) * nx + i) * rowt : i * row0;
                if (d >= nx) cx.sp[(size_t)b * cx.sp_nnz + base + (t >= 1 ? nx : 0) + 1 + (d - nx)] = v;
                else if (t >= 1) cx.sp[(size_t)b * cx.sp_nnz + base + d] = v;
            }
        }
    }
    // defects: lanes run over (row, state) with the state fastest -> contiguous inside a problem
    for (int item = tid; item < NT * 16 * nx; item += NTHREADS) {
        const int idx = FX ? item / nx : (nx == 1 ? item : (int)__umulhi((unsigned)item, cx.inv32_nx)), i = item - idx * nx;
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
            if (jac) {
                T* const row = jac + ((size_t)b * cx.m + (size_t)(t * nx + i)) * (size_t)n;
                fx_store_wt(row + t * nx + i, T(-1));
                if (cx.box) fx_store_wt(row + (size_t)H * nx * (size_t)n + t * nx + i, T(1));
            }
            if (cx.sp) {
                const int row0 = 1 + nu, rowt = nx + 1 + nu;
                T* const spb = cx.sp + (size_t)b * cx.sp_nnz;
                spb[t >= 1 ? nx * row0 + ((t - 1) * nx + i) * rowt + nx : i * row0] = T(-1);
                if (cx.box) spb[nx * row0 + (H - 1) * nx * rowt + t * nx + i] = T(1);
            }
        }
    }
    lds_barrier();
    COOP_STAMP(50);
}

// Everything the kernel needs, prepared on the host (try_launch_coop): the kernel itself does no setup arithmetic.
// The first version derived its constants in the kernel -- three 64-bit scalar divisions, a dozen dependent
// s_load round trips into a 100-SGPR budget -- and spent 1.7 us between wave start and its first vector load
// (tools/diag_stamps.py); now the fields the first loads need come first and the rest arrives in their shadow.
struct CoopArgs {
    // ---- needed by the first vector loads
    const void* Z;
    const void* X0;
    const void* small;      // blob + off.coop_small: [w0f | seed | bias_l | biasL]
    const void* wslice;     // blob + off.coop_slices: per-wave register slices, 16-byte lane vectors
    const void* extra;
    int tiles_per_wg, tiles_rem;
    unsigned R;             // rows = B*H
    unsigned invH;          // ceil(2^32 / H), 0 for H == 1: r / H == umulhi(r, invH) while r*H < 2^32 (checked by the host)
    int H, n, nx, nu, nin, ne;
    int small_vecs;         // 16-byte vectors in `small`
    int nload;              // 16-byte loads per lane of a wave's slice
    int early;              // first-pass inputs through registers (plain models whose columns fit StageRegs)
    // ---- LDS carve-up (element offsets) and table offsets inside the LDS copy of `small`
    int l_w0f, l_tail, l_x, l_xhalf, l_part, l_scratch, l_scratch2, l_rowinfo, l_rowinfo2;   // *2: the second buffer
    int bias_off[NEMPC_MFMA_MAX_HIDDEN], biasL_off;
    // ---- the rest of CoopCtx
    void* g;
    void* tiles;
    void* jac;              // dense Jacobian written by this launch (plain models; null: the assembly kernel does it)
    void* sp;               // sparse contract: the band values written by this launch (plain models), row-major pattern order
    int sp_nnz;
    void* stage_out;
    long long* dbg;
    int m, NR, jsz, spt, nstages, ks, kind, box, xt_off, ex_off, inv_nin, rk4, stage_stride;
    unsigned inv32_jrow, inv32_nx;
    double DT;
    RowGather gk;
    // the objective of the batch from this launch too (dense evaluations of plain models, launch_rows_mfma_dense): f (B) and
    // grad (B, n), either may be null; P = the handle's objective table; every workgroup takes a slice of the problems
    void* obj_f;
    void* obj_grad;
    const void* obj_P;
    ObjOffsets oo;
    int B;
    ActSpec acts;           // hidden layers' activations (NEMPC_ACT_RUNTIME instantiations)
};

// SR: also write the per-(row, stage) records of the RK4 Hessian pipeline (its own instantiation: the extra stores and
// their address arithmetic cost the plain kernel 0.3 - 0.9 us when they are only branched around)
// OCC: waves per SIMD the register allocation must allow (workgroups per CU = OCC * 4 / MT)
template <typename T, int WP, int NH, int TPW, bool SR, int OCC, int ACT, int NXc = 0, int NUc = 0>
__global__ __launch_bounds__((WP / 16) * 64, OCC) void rows_coop_kernel(CoopArgs a) {
    constexpr int MT = WP / 16;
    constexpr int NTHREADS = MT * 64;
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    NEMPC_STAMP_A(0);
    COOP_WGSTAMP(a.dbg, 0);
#ifdef NEMPC_STAMPS
    if (a.dbg && threadIdx.x == 0 && blockIdx.x < 4096)
        a.dbg[1024 + blockIdx.x * 16 + 15] = ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
#endif

    CoopCtx<T> cx;
    cx.Z = static_cast<const T*>(a.Z);
    cx.X0 = static_cast<const T*>(a.X0);
    cx.extra = static_cast<const T*>(a.extra);
    cx.nx = a.nx; cx.nu = a.nu; cx.nin = a.nin; cx.H = a.H; cx.n = a.n; cx.ne = a.ne;
    cx.R = a.R; cx.invH = a.invH;

    // contiguous, balanced range of tiles for this workgroup (quotient / remainder computed on the host)
    const int t_begin = blockIdx.x * a.tiles_per_wg + ((int)blockIdx.x < a.tiles_rem ? (int)blockIdx.x : a.tiles_rem);
    const int t_end = t_begin + a.tiles_per_wg + ((int)blockIdx.x < a.tiles_rem ? 1 : 0);

    // ---- every global load of the prologue is issued before anything waits (VMEM returns in order):
    //      1. the first pass's inputs, 2. the small tables, 3. this wave's weight slices.
    StageRegs<T, MT, TPW> sr;
    int t0 = t_begin;
    int nact = t_end - t0 < TPW ? t_end - t0 : TPW;
    const bool early = a.early != 0;
#ifdef NEMPC_EXP_NOSTAGE   // timing experiment only
    for (int it = 0; it < StageRegs<T, MT, TPW>::ITEMS; ++it) { sr.v[it] = T(0.01) * T(tid & 15); sr.b[it] = tid & 7; sr.t[it] = tid & 3; }
#else
    if (early) stage_load<T, MT, TPW>(cx, t0, nact * 16, tid, sr);
#endif
    COOP_WGSTAMP(a.dbg, 1);
    constexpr int SMALL_PER_THREAD = 4;     // covers 4 * NTHREADS 16-byte vectors; larger tables loop below
    vecT sm[SMALL_PER_THREAD];
    {
        const vecT* __restrict__ gs = static_cast<const vecT*>(a.small);
#pragma unroll
        for (int u = 0; u < SMALL_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
#ifdef NEMPC_EXP_NOSMALL   // timing experiment only
            for (int e = 0; e < VEC; ++e) sm[u][e] = T(1e-3) * T(idx + e);
#else
            if (idx < a.small_vecs) sm[u] = gs[idx];
#endif
        }
    }
    COOP_WGSTAMP(a.dbg, 2);
    // The CU's texture path moves 64 B/clk and serves requests in arrival order: the two workgroups' slices are
    // 160 KB = 1.1 us of it.  Hold the slice loads back until every wave of the workgroup has queued its inputs and
    // tables, or the late waves' inputs wait behind the early waves' slices (first pass 1.9 us later, measured).
#if NEMPC_COOP_PROLOGUE_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    // this wave's weight slices -> registers (kept for every pass): nload fully coalesced 1-KB loads off one base;
    // first needed by hidden layer 1, so their latency hides under layer 0 of the first pass
    constexpr int NFRAG = (NH - 1) * 2 * MT * 4 + 8;
    constexpr int NLOAD = (NFRAG + VEC - 1) / VEC;
    vecT wv[NLOAD];
    {
        const vecT* __restrict__ ws = static_cast<const vecT*>(a.wslice) + (size_t)w * NLOAD * 64 + lane;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
#ifdef NEMPC_EXP_NOWEIGHTS   // timing experiment only: how much of the prologue is the weight fetch
            for (int e = 0; e < VEC; ++e) wv[k][e] = T(1e-3) * T(lane + k + e);
#else
            wv[k] = ws[k * 64];
#endif
        }
    }
    NEMPC_STAMP_A(1);
    COOP_WGSTAMP(a.dbg, 3);

    // ---- the rest of the context (scalar loads in the shadow of the vector loads above)
    cx.w0f = lds + a.l_w0f;
    cx.seed = lds + a.l_tail;
    for (int l = 0; l < NEMPC_MFMA_MAX_HIDDEN; ++l) cx.bias_off[l] = a.bias_off[l];
    if constexpr (ACT == NEMPC_ACT_RUNTIME) cx.acts = a.acts;
    cx.biasL_off = a.biasL_off;
    cx.X = lds + a.l_x;
    cx.xhalf = a.l_xhalf;
    cx.PART = lds + a.l_part;
    // scratch and row info are double-buffered: pass k works in buffer k & 1 while the inputs of pass k + 1 land in
    // the other one
    T* const scr_base = lds + a.l_scratch;
    int* const ri_base = reinterpret_cast<int*>(lds + a.l_rowinfo);
    const int scr_sz = a.l_scratch2 - a.l_scratch, ri_sz = (a.l_rowinfo2 - a.l_rowinfo) * (int)(sizeof(T) / sizeof(int));
    cx.SCR = scr_base;
    cx.RI = ri_base;
    cx.gout = static_cast<T*>(a.g);
    cx.tiles = static_cast<T*>(a.tiles);
    cx.jac = static_cast<T*>(a.jac);
    cx.sp = static_cast<T*>(a.sp); cx.sp_nnz = a.sp_nnz;
    cx.m = a.m;
    cx.gk = a.gk;
    cx.stage_out = static_cast<T*>(a.stage_out); cx.stage_stride = a.stage_stride;
    cx.NR = a.NR; cx.jsz = a.jsz; cx.spt = a.spt; cx.xt_off = a.xt_off; cx.ex_off = a.ex_off;
    cx.inv32_jrow = a.inv32_jrow; cx.inv32_nx = a.inv32_nx; cx.inv_nin = a.inv_nin;
    cx.rk4 = a.rk4 != 0; cx.nstages = a.nstages; cx.ks = a.ks; cx.kind = a.kind; cx.box = a.box;
    cx.DT = (T)a.DT;
    cx.dbg = a.dbg;

    // small tables -> LDS (waits for the inputs and the tables only; the weight slices are still in flight)
    {
        vecT* ls = reinterpret_cast<vecT*>(lds + a.l_w0f);
#pragma unroll
        for (int u = 0; u < SMALL_PER_THREAD; ++u) {
            const int idx = tid + u * NTHREADS;
            if (idx < a.small_vecs) ls[idx] = sm[u];
        }
        const vecT* __restrict__ gs = static_cast<const vecT*>(a.small);
        for (int idx = tid + SMALL_PER_THREAD * NTHREADS; idx < a.small_vecs; idx += NTHREADS) ls[idx] = gs[idx];
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
#ifdef NEMPC_STAMPS
    int npass = 0;
#endif

    int parity = 0;
    if (early) stage_store<T, MT, TPW>(cx, cx.SCR, cx.RI, nact * 16, tid, sr);
    while (t0 < t_end) {
        const int t_cur = t0, n_cur = nact;
        if (!early) stage_direct<T, MT, TPW>(cx, t_cur, n_cur * 16, tid);
        // the NEXT pass's inputs are fetched now, under this pass (their registers are free again): without it every
        // pass boundary exposed a full global round trip (2 us at B=1024 with both workgroups of the CU loading)
        t0 += n_cur;
        const bool pf = NEMPC_COOP_MIDSTAGE && early && t0 < t_end;
        if (t0 < t_end) {
            nact = t_end - t0 < TPW ? t_end - t0 : TPW;
            if (early) stage_load<T, MT, TPW>(cx, t0, nact * 16, tid, sr);
        }
        if (cx.jac) {
            // background zeros of this pass's dense rows: out now, overwritten by the pass's outputs
            const unsigned r0 = (unsigned)t_cur * 16u;
            const int nrows = (size_t)r0 + (size_t)n_cur * 16 <= cx.R ? n_cur * 16 : (int)(cx.R - r0);
            coop_zero_rows<T, NTHREADS>(cx.jac, r0, nrows, cx.nx, cx.n, cx.H, cx.invH, cx.box != 0, tid);
        }
        lds_barrier();
#ifdef NEMPC_STAMPS
        if (npass < 4) COOP_WGSTAMP(a.dbg, 4 + 2 * npass);
#endif
        T* const scr_n = scr_base + (parity ^ 1) * scr_sz;
        int* const ri_n = ri_base + (parity ^ 1) * ri_sz;
        if (n_cur == 1) coop_pass<T, WP, NH, 1, SR, TPW, ACT, NXc, NUc>(cx, W, t_cur, tid, sr, pf, scr_n, ri_n, nact * 16);
        if constexpr (TPW >= 2) { if (n_cur == 2) coop_pass<T, WP, NH, 2, SR, TPW, ACT, NXc, NUc>(cx, W, t_cur, tid, sr, pf, scr_n, ri_n, nact * 16); }
        if constexpr (TPW >= 3) { if (n_cur == 3) coop_pass<T, WP, NH, 3, SR, TPW, ACT, NXc, NUc>(cx, W, t_cur, tid, sr, pf, scr_n, ri_n, nact * 16); }
        if constexpr (TPW >= 4) { if (n_cur == 4) coop_pass<T, WP, NH, 4, SR, TPW, ACT, NXc, NUc>(cx, W, t_cur, tid, sr, pf, scr_n, ri_n, nact * 16); }
        if (!NEMPC_COOP_MIDSTAGE && early && t0 < t_end) stage_store<T, MT, TPW>(cx, scr_n, ri_n, nact * 16, tid, sr);
        parity ^= 1;
        cx.SCR = scr_n;
        cx.RI = ri_n;
#ifdef NEMPC_STAMPS
        cx.dbg = nullptr;   // diagnostic build: keep the FIRST pass's stamps
        if (npass < 4) COOP_WGSTAMP(a.dbg, 5 + 2 * npass);
        ++npass;
#endif
    }
    // The objective of a slice of the batch, a problem per wave (objective_body: the routine of the objective kernel, hence
    // its bits) -- after the passes, where a workgroup that has finished its tiles would otherwise idle until the launch
    // ends; a launch of its own cost 23 us behind a 550 us row launch at configs[2] and a third of the evaluation at small
    // shapes
    if (a.obj_f || a.obj_grad) {
        const T* __restrict__ Pg = static_cast<const T*>(a.obj_P);
        for (int b = (int)blockIdx.x * MT + w; b < a.B; b += (int)gridDim.x * MT)
            objective_body<T>(b, lane, a.H, a.nx, a.nu, a.oo, Pg, cx.Z, static_cast<T*>(a.obj_f), static_cast<T*>(a.obj_grad));
    }
    NEMPC_STAMP_A(12);
    COOP_WGSTAMP(a.dbg, 14);
    COOP_WGSTAMP_REAL(a.dbg, 13);
}

}  // namespace nempc
