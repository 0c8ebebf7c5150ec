// CU-cooperative matrix-core row kernel (gfx950) for networks whose packed weights fit in LDS.
//
// Why a second MFMA kernel: at the headline size (B*H = 20480 rows = 1280 tiles of 16 rows) the
// wave-per-tile kernel gives each of the chip's 1024 SIMDs 1.25 tiles on average but 2 in the worst
// case, and a lone wave serialises its own MFMA and tanh phases.  Here a workgroup of MT = WP/16
// waves (one per SIMD for WP = 64) shares every tile it owns: wave w computes feature block w
// (16 hidden units) of every layer for ALL of the workgroup's tiles, so the matrix work of a CU is
// split evenly over its SIMDs whatever the tile count, each weight fragment read from LDS feeds one
// MFMA per tile, and tanh work is split the same way.
//
//   hidden layer l   every wave publishes its block of a_{l-1} (4 accumulator registers = a B
//                    operand) to the LDS exchange buffer X, barrier, then accumulates its block of
//                    W_l^T a_{l-1} over all K from X (lane-linear ds_read_b64, conflict free).
//   skinny layers    (network output, and the last reverse step onto the nx+nu inputs) are K-split:
//                    wave w contracts its own block and writes a partial; the partials are summed in
//                    the per-stage reduction.
//   reverse sweep    identical structure with W instead of W^T, one cotangent per network output.
//
// Same packed blob, same math and same outputs as kernels_mfma_impl.h; see there for the operand
// layout trick that keeps activations in registers without transposes.
#pragma once

#include "kernels_mfma_impl.h"

namespace nempc {

struct CoopLayout {  // element offsets inside dynamic LDS, after the blob
    int x;           // exchange buffer      TPW * MT * 256
    int part;        // partials             (1+nx) * TPW * MT * NR * 64
    int scratch;     // per-tile scratch     TPW * scratch_per_tile
    int total;
};

template <typename T>
__host__ __device__ inline int coop_nr(int nin) {
    return sizeof(T) == 8 ? (nin + 3) / 4 : 4;
}


template <typename T>
struct CoopCtx {
    const T* wsrc;
    T* X;
    T* PART;
    T* SCR;
    const T* Z;
    const T* X0;
    T* gout;
    T* tiles;
    int nx, nu, nin, H, n, NR, jsz, spt, nstages;
    size_t R;
    bool rk4;
    T DT;
};

// One pass over NT (compile-time) tiles starting at tile t0.  NT is a template parameter on purpose:
// with a runtime tile count every per-tile MFMA sat in its own basic block and hipcc copied the whole
// accumulator set through AGPRs at each join (6,500 v_accvgpr_* moves, 10x slower reverse sweep).
template <typename T, int WP, int NH, int TPW, int NT>
__device__ __forceinline__ void coop_pass(const MfmaParams& p, const CoopCtx<T>& cx, int t0, int tid) {
    using Ops = MfmaOps<T>;
    using V4 = typename Ops::V4;
    constexpr int MT = WP / 16;
    constexpr int NTHREADS = MT * 64;
    constexpr int nact = NT;
    const int lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const T* wsrc = cx.wsrc;
    T* X = cx.X;
    T* PART = cx.PART;
    T* SCR = cx.SCR;
    const T* __restrict__ Z = cx.Z;
    const T* __restrict__ X0 = cx.X0;
    T* __restrict__ gout = cx.gout;
    T* __restrict__ tiles = cx.tiles;
    const int nx = cx.nx, nu = cx.nu, nin = cx.nin, H = cx.H, n = cx.n, NR = cx.NR, jsz = cx.jsz, spt = cx.spt;
    const int nstages = cx.nstages;
    const size_t R = cx.R;
    const bool rk4 = cx.rk4;
    const T DT = cx.DT;
        // ---- stage inputs xi0[j][cc][d] for the pass's tiles
        for (int e = tid; e < nact * 16 * nin; e += NTHREADS) {
            const int j = e / (16 * nin), e2 = e - j * 16 * nin;
            const int cc = e2 / nin, d = e2 - cc * nin;
            const size_t r = (size_t)(t0 + j) * 16 + cc;
            T v = T(0);
            if (r < R) {
                const int b = (int)(r / H), t = (int)(r - (size_t)b * H);
                const T* z = Z + (size_t)b * n;
                if (d < nx) v = (t == 0) ? X0[(size_t)b * nx + d] : z[(t - 1) * nx + d];
                else v = z[H * nx + t * nu + (d - nx)];
            }
            SCR[j * spt + e2] = v;
        }
        __syncthreads();
        NEMPC_STAMP(2);

        for (int stage = 0; stage < nstages; ++stage) {
            const T cdt = (stage == 0) ? T(0) : ((stage == 3) ? DT : T(0.5) * DT);
            V4 a[NH][NT];
            // ---- layer 0, this wave's feature block
            {
                const T* bias = wsrc + p.off.bias[0] + w * 16;
                V4 b0;
#pragma unroll
                for (int r = 0; r < 4; ++r) b0[r] = bias[r * 4 + q];
#pragma unroll
                for (int j = 0; j < NT; ++j) a[0][j] = b0;
                for (int ks = 0; ks < p.ks; ++ks) {
                    {
                        const T wfrag = wsrc[p.off.w0f + (ks * MT + w) * 64 + lane];
                        const int d = 4 * ks + q;
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            {
                                const T* s_xi0 = SCR + j * spt;
                                T v = T(0);
                                if (d < nin) {
                                    v = s_xi0[c * nin + d];
                                    if (stage > 0 && d < nx) v = fma(cdt, s_xi0[16 * nin + c * nx + d], v);
                                }
                                a[0][j] = Ops::mma(wfrag, v, a[0][j]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[0][j][r] = Ops::tanh_(a[0][j][r]);
            }
            NEMPC_STAMP(3);
            // ---- hidden-to-hidden layers
#pragma unroll
            for (int l = 1; l < NH; ++l) {
                __syncthreads();  // X free (previous readers done; also orders the s_k reads above)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[((j * MT + w) * 4 + r) * 64 + lane] = a[l - 1][j][r];
                __syncthreads();
                const T* bias = wsrc + p.off.bias[l] + w * 16;
                V4 b0;
#pragma unroll
                for (int r = 0; r < 4; ++r) b0[r] = bias[r * 4 + q];
#pragma unroll
                for (int j = 0; j < NT; ++j) a[l][j] = b0;
                const T* wl = wsrc + p.off.wf[l];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const T wfrag = wl[((mt * 4 + r) * MT + w) * 64 + lane];
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            a[l][j] = Ops::mma(wfrag, X[((j * MT + mt) * 4 + r) * 64 + lane], a[l][j]);
                    }
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[l][j][r] = Ops::tanh_(a[l][j][r]);
            }
            NEMPC_STAMP(4);
            // ---- network output: K-split partial over this wave's block
            {
                T wfr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) wfr[r] = wsrc[p.off.wLf + (w * 4 + r) * 64 + lane];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    {
                        V4 pf = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                        for (int r = 0; r < 4; ++r) pf = Ops::mma(wfr[r], a[NH - 1][j][r], pf);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r < NR) PART[(((0 * TPW + j) * MT + w) * NR + r) * 64 + lane] = pf[r];
                    }
                }
            }
#pragma unroll
            for (int l = 0; l < NH; ++l)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    a[l][j] = T(1) - a[l][j] * a[l][j];

            NEMPC_STAMP(5);
            // ---- reverse sweep
            for (int k = 0; k < nx; ++k) {
                V4 cv[NT];
                {
                    const T* seed = wsrc + p.off.seed + k * MT * 16 + w * 16;
                    V4 sd;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sd[r] = seed[r * 4 + q];
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        cv[j] = sd * a[NH - 1][j];
                }
#pragma unroll
                for (int l = NH - 1; l >= 1; --l) {
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) X[((j * MT + w) * 4 + r) * 64 + lane] = cv[j][r];
                    __syncthreads();
                    V4 cn[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) cn[j] = V4{T(0), T(0), T(0), T(0)};
                    const T* wl = wsrc + p.off.wb[l];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const T wfrag = wl[((mt * 4 + r) * MT + w) * 64 + lane];
#pragma unroll
                            for (int j = 0; j < NT; ++j)
                                cn[j] = Ops::mma(wfrag, X[((j * MT + mt) * 4 + r) * 64 + lane], cn[j]);
                        }
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        cv[j] = cn[j] * a[l - 1][j];
                }
                T wfr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) wfr[r] = wsrc[p.off.w0b + (w * 4 + r) * 64 + lane];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    {
                        V4 pj = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
                        for (int r = 0; r < 4; ++r) pj = Ops::mma(wfr[r], cv[j][r], pj);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r < NR) PART[((((1 + k) * TPW + j) * MT + w) * NR + r) * 64 + lane] = pj[r];
                    }
                }
            }
            NEMPC_STAMP(6);
            __syncthreads();
            NEMPC_STAMP(7);

            // ---- reduce the K-split partials: f -> s_k[cc][o], J -> s_J[cc][k][d]
            for (int e = tid; e < nact * 16 * nx; e += NTHREADS) {
                const int j = e / (16 * nx), e2 = e - j * 16 * nx;
                const int cc = e2 / nx, o = e2 - cc * nx;
                const int qq = sizeof(T) == 8 ? (o & 3) : (o >> 2), rr = sizeof(T) == 8 ? (o >> 2) : (o & 3);
                T v = wsrc[p.off.biasL + rr * 4 + qq];
                for (int ww = 0; ww < MT; ++ww) v += PART[(((0 * TPW + j) * MT + ww) * NR + rr) * 64 + qq * 16 + cc];
                SCR[j * spt + 16 * nin + e2] = v;
            }
            for (int e = tid; e < nact * jsz; e += NTHREADS) {
                const int j = e / jsz, e2 = e - j * jsz;
                const int cc = e2 / (nx * nin), rem2 = e2 - cc * nx * nin;
                const int k = rem2 / nin, d = rem2 - k * nin;
                const int qq = sizeof(T) == 8 ? (d & 3) : (d >> 2), rr = sizeof(T) == 8 ? (d >> 2) : (d & 3);
                T v = T(0);
                for (int ww = 0; ww < MT; ++ww)
                    v += PART[((((1 + k) * TPW + j) * MT + ww) * NR + rr) * 64 + qq * 16 + cc];
                SCR[j * spt + 16 * nin + 2 * 16 * nx + e2] = v;
            }
            __syncthreads();
            NEMPC_STAMP(8);

            // ---- RK4 chain rule on the per-tile scratch (rk4.py:147-159)
            if (rk4) {
                if (stage == 0) {
                    for (int e = tid; e < nact * jsz; e += NTHREADS) {
                        const int j = e / jsz, e2 = e - j * jsz;
                        T* sj = SCR + j * spt + 16 * nin + 2 * 16 * nx;
                        const T v = sj[e2];
                        sj[jsz + e2] = v;
                        sj[2 * jsz + e2] = v;
                    }
                    for (int e = tid; e < nact * 16 * nx; e += NTHREADS) {
                        const int j = e / (16 * nx), e2 = e - j * 16 * nx;
                        T* sk = SCR + j * spt + 16 * nin;
                        sk[16 * nx + e2] = sk[e2];
                    }
                } else {
                    const T wgt = (stage == 3) ? T(1) : T(2);
                    for (int e = tid; e < nact * jsz; e += NTHREADS) {
                        const int j = e / jsz, e2 = e - j * jsz;
                        const int cc = e2 / (nx * nin), rem2 = e2 - cc * nx * nin;
                        const int i = rem2 / nin, d = rem2 - i * nin;
                        const T* sj = SCR + j * spt + 16 * nin + 2 * 16 * nx;
                        T v = T(0);
                        for (int e3 = 0; e3 < nx; ++e3)
                            v = fma(sj[(cc * nx + i) * nin + e3], sj[jsz + (cc * nx + e3) * nin + d], v);
                        SCR[j * spt + 16 * nin + 2 * 16 * nx + 3 * jsz + e2] = fma(cdt, v, sj[e2]);
                    }
                    __syncthreads();
                    for (int e = tid; e < nact * jsz; e += NTHREADS) {
                        const int j = e / jsz, e2 = e - j * jsz;
                        T* sj = SCR + j * spt + 16 * nin + 2 * 16 * nx;
                        const T v = sj[3 * jsz + e2];
                        sj[jsz + e2] = v;
                        sj[2 * jsz + e2] = fma(wgt, v, sj[2 * jsz + e2]);
                    }
                    for (int e = tid; e < nact * 16 * nx; e += NTHREADS) {
                        const int j = e / (16 * nx), e2 = e - j * 16 * nx;
                        T* sk = SCR + j * spt + 16 * nin;
                        sk[16 * nx + e2] = fma(wgt, sk[e2], sk[16 * nx + e2]);
                    }
                }
                __syncthreads();
            }
        }

        // ---- outputs
        const T s6 = DT / T(6);
        for (int e = tid; e < nact * jsz; e += NTHREADS) {
            const int j = e / jsz, e2 = e - j * jsz;
            const int cc = e2 / (nx * nin), rem2 = e2 - cc * nx * nin;
            const int i = rem2 / nin, d = rem2 - i * nin;
            const size_t row0 = (size_t)(t0 + j) * 16;
            if (row0 + cc < R) {
                const T* sj = SCR + j * spt + 16 * nin + 2 * 16 * nx;
                T v;
                if (rk4) v = s6 * sj[2 * jsz + e2] + (d == i ? T(1) : T(0));
                else v = sj[e2] + ((p.kind == NEMPC_DISCRET && d == i) ? T(1) : T(0));
                tiles[row0 * nx * nin + e2] = v;
            }
        }
        for (int e = tid; e < nact * 16 * nx; e += NTHREADS) {
            const int j = e / (16 * nx), e2 = e - j * 16 * nx;
            const int cc = e2 / nx, i = e2 - cc * nx;
            const size_t r = (size_t)(t0 + j) * 16 + cc;
            if (r < R) {
                const int b = (int)(r / H), t = (int)(r - (size_t)b * H);
                const T* s_xi0 = SCR + j * spt;
                const T* sk = s_xi0 + 16 * nin;
                const T xp = s_xi0[cc * nin + i];
                T phi;
                if (rk4) phi = xp + s6 * sk[16 * nx + e2];
                else phi = (p.kind == NEMPC_DISCRET ? xp : T(0)) + sk[e2];
                const T xt = Z[(size_t)b * n + t * nx + i];
                gout[(size_t)b * p.m + t * nx + i] = phi - xt;
                if (p.box) gout[(size_t)b * p.m + (size_t)H * nx + t * nx + i] = xt;
            }
        }
        __syncthreads();
        NEMPC_STAMP(9);
}

template <typename T, int WP, int NH, int TPW>
__global__ __launch_bounds__((WP / 16) * 64) void rows_coop_kernel(MfmaParams p, CoopLayout lay) {
    using Ops = MfmaOps<T>;
    using V4 = typename Ops::V4;
    constexpr int MT = WP / 16;
    constexpr int NTHREADS = MT * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const T* __restrict__ gblob = static_cast<const T*>(p.blob);
    NEMPC_STAMP(0);
    copy_blob_to_lds<T>(gblob, lds, p.off.total, tid, NTHREADS);
    const T* wsrc = lds;
    T* base = lds + ((p.off.total + 1) & ~1);
    T* X = base + lay.x;
    T* PART = base + lay.part;
    T* SCR = base + lay.scratch;

    const int nx = p.nx, nu = p.nu, nin = p.nin, H = p.H;
    const int n = H * nin;
    const size_t R = (size_t)p.B * H;
    const T* __restrict__ Z = static_cast<const T*>(p.Z);
    const T* __restrict__ X0 = static_cast<const T*>(p.X0);
    T* __restrict__ gout = static_cast<T*>(p.g);
    T* __restrict__ tiles = static_cast<T*>(p.tiles);
    const bool rk4 = p.kind == NEMPC_RK4;
    const int nstages = rk4 ? 4 : 1;
    const T DT = (T)p.DT;
    const int NR = coop_nr<T>(nin);
    const int jsz = 16 * nx * nin;
    const int spt = p.scratch_per_wave;  // per-tile scratch elements (same carve-up as the wave-tile kernel)

    // contiguous, balanced range of tiles for this workgroup
    const int per = p.ntiles / gridDim.x, rem = p.ntiles % gridDim.x;
    const int t_begin = blockIdx.x * per + (blockIdx.x < rem ? blockIdx.x : rem);
    const int t_end = t_begin + per + (blockIdx.x < rem ? 1 : 0);
    __syncthreads();

    NEMPC_STAMP(1);
    CoopCtx<T> cx;
    cx.wsrc = wsrc; cx.X = X; cx.PART = PART; cx.SCR = SCR; cx.Z = Z; cx.X0 = X0; cx.gout = gout; cx.tiles = tiles;
    cx.nx = nx; cx.nu = nu; cx.nin = nin; cx.H = H; cx.n = n; cx.NR = NR; cx.jsz = jsz; cx.spt = spt;
    cx.nstages = nstages; cx.R = R; cx.rk4 = rk4; cx.DT = DT;
    int t0 = t_begin;
    while (t0 < t_end) {
        const int left = t_end - t0;
        const int passes_left = (left + TPW - 1) / TPW;
        const int nact = (left + passes_left - 1) / passes_left;  // even pass sizes, <= TPW
        if (nact == 1) coop_pass<T, WP, NH, TPW, 1>(p, cx, t0, tid);
        if constexpr (TPW >= 2) { if (nact == 2) coop_pass<T, WP, NH, TPW, 2>(p, cx, t0, tid); }
        if constexpr (TPW >= 3) { if (nact == 3) coop_pass<T, WP, NH, TPW, 3>(p, cx, t0, tid); }
        if constexpr (TPW >= 4) { if (nact == 4) coop_pass<T, WP, NH, TPW, 4>(p, cx, t0, tid); }
        t0 += nact;
    }
}

}  // namespace nempc
