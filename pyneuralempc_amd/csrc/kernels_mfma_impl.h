// Matrix-core row kernel (gfx950): one wavefront owns a tile of 16 (problem, step) rows.
//
// Formulation ("features on M, rows on N"): every dense layer is computed transposed,
//     Z^T (features x 16 rows) = W^T (features x K) . A^T (K x 16 rows)
// with v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32.  The weight fragment is the MFMA A operand,
// the activations are the B operand.  Because the C/D accumulator layout of these instructions puts
// the 16 batch rows on lane&15 and the feature index on (lane>>4, register), an accumulator register
// IS a valid B operand of the next layer's k-step: activations never leave registers and never get
// transposed; only the (host pre-packed) weight fragments are permuted to match.  The reverse sweep
// (one cotangent per network output) uses the same trick with W instead of W^T.
//
//   forward   a_l = s(W_l^T a_{l-1} + b_l),  f = W_L^T a_{L-1} + b_L       (s = Act<T, ACT>, activations.h: tanh by default)
//   reverse   c = W_L[:,k] s'(z);  c <- (W_l c) s'(z_{l-1}) ... ;  J[k,:] = W_1 c   (s' from a: tanh 1-a^2)
//   (value-equivalent to what tf.GradientTape.jacobian yields per row, model/tensorflow.py:53-75)
//
// The per-row integrator algebra (rk4.py:147-159 chain rule, discret.py:52 identity) runs on a
// small per-wave LDS scratch after each stage; the tile's outputs leave through coalesced stores.
//
// Packed weight blob: see mfma_pack_weights() in kernels_mfma.hip (lane-linear fragments, so a
// fragment read is one conflict-free ds_read / one fully coalesced global load).
#pragma once

#include "activations.h"
#include "nempc_internal.h"

namespace nempc {

constexpr int kMaxKs = 8;   // first-layer k-steps: network inputs (window + extras) up to 32

struct MfmaOffsets {  // element offsets into the packed blob
    int w0f, wLf, w0b, seed, biasL;
    int p0tab, wLb;  // Hessian kernel tables: first-layer rows W_0[p,:], output-layer fragments for W_L lambda
    int wf[NEMPC_MFMA_MAX_HIDDEN], wb[NEMPC_MFMA_MAX_HIDDEN], bias[NEMPC_MFMA_MAX_HIDDEN];
    int total;
    // cooperative row kernel: the same numbers once more, laid out for its prologue (kernels_coop_impl.h) --
    //   coop_small   [w0f | seed | bias_l | biasL] contiguous (one flat copy to LDS),
    //   coop_slices  per wave: its register-resident fragments (wf, wb per hidden layer, wL, w0b) as 16-byte
    //                lane vectors, load k of lane = element ((wave * coop_nload + k) * 64 + lane) * VEC
    int coop_small, coop_small_elems, coop_slices, coop_nload;
    //   fx_small     [w0f | seed | bias_l | biasL | p0tab]: the fixed-shape kernel's tables (p0tab = first-layer rows per
    //                lane, for its vector-unit skinny layers)
    int fx_small, fx_small_elems;
    int grand_total;
};

struct MfmaParams {
    const void* blob;
    MfmaOffsets off;
    int nx, nu, nin, ks;  // ks = padded-input k-steps (ceil((nin+ne)/4)) <= kMaxKs
    int mb;               // 16-row blocks of the input dimension in the last reverse step: ceil(nin/16), 1 or 2
    int ne;               // extra network inputs per row (tvp, p); no Jacobian columns
    const void* extra;    // (B,H,ne) or null
    int kind;
    double DT;
    int B, H, m, box;
    int num_cus;          // compute units of the handle's device: the launch geometry is sized from it
    const void* Z;
    const void* X0;
    void* g;
    void* tiles;
    int ntiles;
    int tiles_per_wg, tiles_rem;  // cooperative kernel: ntiles = grid * tiles_per_wg + tiles_rem
    int scratch_per_wave;  // elements
    RowGather gk;          // input gather (rolling windows); nin above is the tile width w*(nx+nu)
    void* stage_out;       // RK4 Hessian pipeline only: per (row, stage) record [xi_s (nin) | J_s (nx*nin) | dk_{s-1} (nx*nin)]
    int stage_stride;      // elements per record = nin + 2*nx*nin
    long long* dbg;        // diagnostic builds only (-DNEMPC_STAMPS): per-wave phase stamps of workgroup 0
    // fused evaluation (fixed-shape kernel only): dense Jacobian and objective from the same launch
    bool fuse_obj;         // one-launch evaluation asked for (launch_eval_fused); fuse_jac may be null (no dense matrix)
    void* fuse_jac;
    void* fuse_sparse;     // band-pattern Jacobian values (B, nnz) in nempc_jac_structure order from the row launch (plain models)
    int sp_nnz;
    void* fuse_f;
    void* fuse_grad;
    const void* obj;
    ObjOffsets oo;
    // Gauss-Newton Hessian callback from the row launch (fixed-shape kernel only, launch_hess_gn_fused)
    void* gn_hvals;
    const void* gn_sigma;
    const void* gn_w;
    const int32_t* gn_smap;
    const void* gn_objc;
    int gn_nnz, gn_n_orph;
    ActSpec acts;          // hidden layers' activations (read by the NEMPC_ACT_RUNTIME instantiations only)
};

// In-kernel stamps exist only in the diagnostic library built by tools/diag_stamps.py; the shipped
// kernels contain none.
#ifdef NEMPC_STAMPS
#define NEMPC_STAMP(idx)                                                                   \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        unsigned long long _t;                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (p.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0) p.dbg[(threadIdx.x >> 6) * 64 + (idx)] = (long long)_t; \
    } while (0)
#else
#define NEMPC_STAMP(idx) \
    do {                 \
    } while (0)
#endif

template <typename T>
struct MfmaOps;

template <>
struct MfmaOps<double> {
    typedef double V4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ V4 mma(double a, double b, V4 c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __host__ __device__ __forceinline__ int row(int q, int r) { return q + 4 * r; }
};

template <>
struct MfmaOps<float> {
    typedef float V4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ V4 mma(float a, float b, V4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __host__ __device__ __forceinline__ int row(int q, int r) { return 4 * q + r; }
};

// same-wave LDS hand-off between lanes: DS ops of one wave execute in order; keep the compiler from
// moving accesses across the point
__device__ __forceinline__ void wave_sync() {
#ifdef NEMPC_TILE_STRONG_SYNC
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Workgroup-cooperative copy of the packed blob into LDS: 16-byte loads, eight in flight per lane.
// (A scalar element-per-iteration loop serialises ~40 dependent L2 round trips per launch, which
// measured as ~15 us of the first version's 41 us.)  n_elems is a multiple of 16.
template <typename T>
__device__ __forceinline__ void copy_blob_to_lds(const T* __restrict__ g, T* l, int n_elems, int tid, int nthreads) {
    typedef T vec __attribute__((ext_vector_type(16 / sizeof(T))));
    const vec* __restrict__ gs = reinterpret_cast<const vec*>(g);
    vec* ls = reinterpret_cast<vec*>(l);
    const int nvec = n_elems / (int)(16 / sizeof(T));
    for (int i = tid; i < nvec; i += nthreads * 8) {
        vec tmp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = i + u * nthreads;
            if (idx < nvec) tmp[u] = gs[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = i + u * nthreads;
            if (idx < nvec) ls[idx] = tmp[u];
        }
    }
}

// acc[mo] += sum over k-steps (mt, r) of  W-fragment[(mt*4+r)*MO + mo] x bop[mt][r]
// Fragments are lane-linear (64 elements each).  LDS-resident weights: plain loop, the compiler
// schedules the ds_reads.  Global (L2) weights: explicit one-step-ahead prefetch fenced by
// sched_barrier so that hipcc does not hoist hundreds of fragment loads and spill.
template <typename T, int MT_IN, int MO, bool WLDS>
__device__ __forceinline__ void layer_mma(const T* __restrict__ w, int lane,
                                          const typename MfmaOps<T>::V4 (&bop)[MT_IN],
                                          typename MfmaOps<T>::V4 (&acc)[MO]) {
    using Ops = MfmaOps<T>;
#ifdef NEMPC_TILE_NO_PREFETCH
    constexpr bool SIMPLE = true;
#else
    constexpr bool SIMPLE = WLDS;
#endif
    if constexpr (SIMPLE) {
#pragma unroll
        for (int mt = 0; mt < MT_IN; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mo = 0; mo < MO; ++mo)
                    acc[mo] = Ops::mma(w[((mt * 4 + r) * MO + mo) * 64 + lane], bop[mt][r], acc[mo]);
    } else {
        // the fragments do not depend on the enclosing cotangent / stage loops: without an opaque
        // offset LICM hoists every one of them out of those loops and the kernel spills
#ifndef NEMPC_TILE_NO_OPAQUE
        int opaque = 0;
        asm volatile("" : "+s"(opaque));
        w += opaque;
#endif
        T wcur[MO], wnxt[MO];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) wcur[mo] = w[mo * 64 + lane];
#pragma unroll
        for (int it = 0; it < MT_IN * 4; ++it) {
            if (it + 1 < MT_IN * 4) {
#pragma unroll
                for (int mo = 0; mo < MO; ++mo) wnxt[mo] = w[((it + 1) * MO + mo) * 64 + lane];
            }
#pragma unroll
            for (int mo = 0; mo < MO; ++mo) acc[mo] = Ops::mma(wcur[mo], bop[it >> 2][it & 3], acc[mo]);
#ifndef NEMPC_TILE_NO_SCHEDBARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int mo = 0; mo < MO; ++mo) wcur[mo] = wnxt[mo];
        }
    }
}

template <typename T, int WP, int NH, bool WLDS, int MAXWAVES, int ACT>
__global__ __launch_bounds__(MAXWAVES * 64) void rows_mfma_kernel(MfmaParams p) {
    using Ops = MfmaOps<T>;
    using A = ActL<T, ACT>;
    using V4 = typename Ops::V4;
    constexpr int MT = WP / 16;
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
    const int n = p.gk.n, xcur = p.gk.xcur;
    const size_t R = (size_t)p.B * H;
    const T* __restrict__ Z = static_cast<const T*>(p.Z);
    const T* __restrict__ X0 = static_cast<const T*>(p.X0);
    T* __restrict__ gout = static_cast<T*>(p.g);
    T* __restrict__ tiles = static_cast<T*>(p.tiles);
    const bool rk4 = p.kind == NEMPC_RK4;
    const bool want_jac = p.tiles != nullptr;   // defect-only launches (line-search trials of the batched solver)
    const int nstages = rk4 ? 4 : 1;
    const T DT = (T)p.DT;

    // per-wave scratch carve-up (elements): xi0[16][nin] kcur[16][nx] acck[16][nx] J/dk/accdk/dkn [16][nx][nin]
    const int jsz = 16 * nx * nin;
    T* s_xi0 = scratch;
    T* s_k = s_xi0 + 16 * nin;
    T* s_acck = s_k + 16 * nx;
    T* s_J = s_acck + 16 * nx;
    T* s_dk = s_J + jsz;
    T* s_accdk = s_dk + jsz;
    T* s_dkn = s_accdk + jsz;
    T* s_ex = s_J + (rk4 ? 4 : 1) * jsz + 16 * nx;   // [16][ne] extra inputs (after the cooperative kernel's x_t slot);
                                                     // the dk / accdk / dkn arrays exist for RK4 only

    for (int tile = blockIdx.x * nwaves + wave; tile < p.ntiles; tile += gridDim.x * nwaves) {
        const size_t row0 = (size_t)tile * 16;

        // ---- stage the tile's inputs xi0[c][d] = [x_{t-1} ; u_t]  (discret.py:22, ipopt.py:20-28), or the rolling
        //      window of states / controls (tensorflow.py:112-130)
        for (int e = lane; e < 16 * nin; e += 64) {
            const int cc = e / nin, d = e - cc * nin;
            const size_t r = row0 + cc;
            T v = T(0);
            if (r < R) {
                const int b = (int)(r / H), t = (int)(r - (size_t)b * H);
                v = gather_input<T>(p.gk, Z + (size_t)b * n, X0, b, t, d);
            }
            s_xi0[e] = v;
        }
        for (int e = lane; e < 16 * p.ne; e += 64) {
            const int cc = e / p.ne, j = e - cc * p.ne;
            const size_t r = row0 + cc;
            s_ex[e] = (r < R) ? static_cast<const T*>(p.extra)[r * p.ne + j] : T(0);
        }
        wave_sync();

        for (int stage = 0; stage < nstages; ++stage) {
            const T cdt = (stage == 0) ? T(0) : ((stage == 3) ? DT : T(0.5) * DT);
            // ---- B operand of the first layer: xin[ks] = xi[4ks+q] of row c
            T xin[kMaxKs];
#pragma unroll
            for (int ks = 0; ks < kMaxKs; ++ks) {
                const int d = 4 * ks + q;
                T v = T(0);
                if (ks < p.ks && d < nin) {
                    v = s_xi0[c * nin + d];
                    if (stage > 0 && d < nx) v = fma(cdt, s_k[c * nx + d], v);
                } else if (ks < p.ks && d < nin + p.ne) {
                    v = s_ex[c * p.ne + (d - nin)];
                }
                xin[ks] = v;
                if (p.stage_out && ks < p.ks && d < nin && row0 + c < R)
                    static_cast<T*>(p.stage_out)[((row0 + c) * 4 + stage) * (size_t)p.stage_stride + d] = v;
            }
            wave_sync();  // s_k is overwritten below

            // ---- forward
            V4 a[NH][MT];
            {
                const T* bias = wsrc + p.off.bias[0];
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[0][mo][r] = bias[(mo * 4 + r) * 4 + q];
                const T* w = wsrc + p.off.w0f;
#pragma unroll
                for (int ks = 0; ks < kMaxKs; ++ks) {
                    if (ks < p.ks) {
#pragma unroll
                        for (int mo = 0; mo < MT; ++mo)
                            a[0][mo] = Ops::mma(w[(ks * MT + mo) * 64 + lane], xin[ks], a[0][mo]);
                    }
                }
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[0][mo][r] = A::f(a[0][mo][r], p.acts, 0);
            }
#pragma unroll
            for (int l = 1; l < NH; ++l) {
                const T* bias = wsrc + p.off.bias[l];
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[l][mo][r] = bias[(mo * 4 + r) * 4 + q];
                layer_mma<T, MT, MT, WLDS>(wsrc + p.off.wf[l], lane, a[l - 1], a[l]);
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[l][mo][r] = A::f(a[l][mo][r], p.acts, l);
            }
            {
                V4 fo;
                const T* bias = wsrc + p.off.biasL;
#pragma unroll
                for (int r = 0; r < 4; ++r) fo[r] = bias[r * 4 + q];
                V4 fo1[1] = {fo};
                layer_mma<T, MT, 1, WLDS>(wsrc + p.off.wLf, lane, a[NH - 1], fo1);
                fo = fo1[0];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = Ops::row(q, r);
                    if (o < nx) s_k[c * nx + o] = fo[r];
                }
            }
            // s'(z) from a (tanh: 1 - a^2) once, reused by every cotangent
            if (want_jac) {
#pragma unroll
                for (int l = 0; l < NH; ++l)
#pragma unroll
                    for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[l][mo][r] = A::d1(a[l][mo][r], p.acts, l);
            }

            // ---- reverse sweep, one cotangent per network output (skipped by defect-only launches: tiles == null)
            for (int k = 0; k < (want_jac ? nx : 0); ++k) {
                V4 cv[MT];
                const T* seed = wsrc + p.off.seed + k * MT * 16;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cv[mt][r] = seed[(mt * 4 + r) * 4 + q] * a[NH - 1][mt][r];
#pragma unroll
                for (int l = NH - 1; l >= 1; --l) {
                    V4 cn[MT];
#pragma unroll
                    for (int mo = 0; mo < MT; ++mo) cn[mo] = V4{T(0), T(0), T(0), T(0)};
                    layer_mma<T, MT, MT, WLDS>(wsrc + p.off.wb[l], lane, cv, cn);
#pragma unroll
                    for (int mo = 0; mo < MT; ++mo) cv[mo] = cn[mo] * a[l - 1][mo];
                }
                if (p.mb == 1) {
                    V4 jk1[1] = {V4{T(0), T(0), T(0), T(0)}};
                    layer_mma<T, MT, 1, WLDS>(wsrc + p.off.w0b, lane, cv, jk1);
                    const V4 jk = jk1[0];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int d = Ops::row(q, r);
                        if (d < nin) s_J[(c * nx + k) * nin + d] = jk[r];
                    }
                } else {   // 17..32 network inputs: two 16-row blocks of the input dimension
                    V4 jk2[2] = {V4{T(0), T(0), T(0), T(0)}, V4{T(0), T(0), T(0), T(0)}};
                    layer_mma<T, MT, 2, WLDS>(wsrc + p.off.w0b, lane, cv, jk2);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int d = 16 * mb + Ops::row(q, r);
                            if (d < nin) s_J[(c * nx + k) * nin + d] = jk2[mb][r];
                        }
                }
            }
            wave_sync();

            // stage record for the Hessian pipeline: this stage's Jacobian and the chain Jacobian it was entered with
            if (p.stage_out && want_jac) {
                T* so = static_cast<T*>(p.stage_out);
                const int jn = nx * nin;
                for (int e = lane; e < jsz; e += 64) {
                    const int cc = e / jn, rem = e - cc * jn;
                    if (row0 + cc < R) {
                        T* rec = so + ((row0 + cc) * 4 + stage) * (size_t)p.stage_stride + nin;
                        rec[rem] = s_J[e];
                        rec[jn + rem] = stage > 0 ? s_dk[e] : T(0);
                    }
                }
            }
            // ---- RK4 chain rule on the scratch (rk4.py:147-159)
            if (rk4) {
                if (stage == 0) {
                    for (int e = lane; e < (want_jac ? jsz : 0); e += 64) {
                        const T v = s_J[e];
                        s_dk[e] = v;
                        s_accdk[e] = v;
                    }
                    for (int e = lane; e < 16 * nx; e += 64) s_acck[e] = s_k[e];
                } else {
                    const T wgt = (stage == 3) ? T(1) : T(2);
                    for (int e = lane; e < (want_jac ? jsz : 0); e += 64) {
                        const int cc = e / (nx * nin), rem = e - cc * nx * nin;
                        const int i = rem / nin, d = rem - i * nin;
                        T v = T(0);
                        for (int e2 = 0; e2 < nx; ++e2)
                            v = fma(s_J[(cc * nx + i) * nin + e2], s_dk[(cc * nx + e2) * nin + d], v);
                        s_dkn[e] = fma(cdt, v, s_J[e]);
                    }
                    wave_sync();
                    for (int e = lane; e < (want_jac ? jsz : 0); e += 64) {
                        const T v = s_dkn[e];
                        s_dk[e] = v;
                        s_accdk[e] = fma(wgt, v, s_accdk[e]);
                    }
                    for (int e = lane; e < 16 * nx; e += 64) s_acck[e] = fma(wgt, s_k[e], s_acck[e]);
                }
                wave_sync();
            }
        }

        // ---- outputs: tiles (16 rows contiguous in memory) and defects
        const T s6 = DT / T(6);
        for (int e = lane; e < (want_jac ? jsz : 0); e += 64) {
            const int cc = e / (nx * nin), rem = e - cc * nx * nin;
            const int i = rem / nin, d = rem - i * nin;
            if (row0 + cc < R) {
                T v;
                if (rk4) v = s6 * s_accdk[e] + (d == i ? T(1) : T(0));
                else v = s_J[e] + ((p.kind == NEMPC_DISCRET && d == xcur + i) ? T(1) : T(0));
                tiles[row0 * nx * nin + e] = v;
            }
        }
        for (int e = lane; e < 16 * nx; e += 64) {
            const int cc = e / nx, i = e - cc * nx;
            const size_t r = row0 + cc;
            if (r < R) {
                const int b = (int)(r / H), t = (int)(r - (size_t)b * H);
                const T xp = s_xi0[cc * nin + xcur + i];
                T phi;
                if (rk4) phi = xp + s6 * s_acck[e];
                else phi = (p.kind == NEMPC_DISCRET ? xp : T(0)) + s_k[e];
                const T xt = Z[(size_t)b * n + t * nx + i];
                gout[(size_t)b * p.m + t * nx + i] = phi - xt;
                if (p.box) gout[(size_t)b * p.m + (size_t)H * nx + t * nx + i] = xt;
            }
        }
        wave_sync();
    }
}

template <typename T, int WP, int NH, bool WLDS, int MAXWAVES, int ACT>
int launch_one(const MfmaParams& p, int waves, int grid, size_t lds_bytes, hipStream_t s) {
    auto kern = rows_mfma_kernel<T, WP, NH, WLDS, MAXWAVES, ACT>;
    NEMPC_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds_bytes, s, p);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

// picks the instantiation for (WP, NH, weights-in-LDS): one translation unit per (dtype, hidden activation)
// (kernels_mfma_typed.inc); launch_rows_mfma_typed dispatches on the handle's activation (kernels_mfma.hip)
template <typename T, int ACT>
int launch_rows_mfma_act(const Handle& h, MfmaParams p, hipStream_t s);
template <typename T>
int launch_rows_mfma_typed(const Handle& h, MfmaParams p, hipStream_t s);

}  // namespace nempc
