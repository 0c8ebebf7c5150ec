// Objective + assembly kernels (HBM-bound streaming work; no matrix cores here).
//
//  objective_kernel      f, grad f of the quadratic/linear family   (objective/jax.py:28-41 stand-in)
//  assemble_dense_kernel tiles -> dense (B,m,n) row-major Jacobian  (integrator/discret.py:38-56,
//                        rk4.py:120-176, optimizer/ipopt.py:88-96): every output element is a constant
//                        (0, -1, +1) or one tile element; a per-handle int32 map built once on the host
//                        says which, so the kernel is a pure gather + 16-byte coalesced store stream.
//  assemble_sparse_kernel same map restricted to the structural non-zeros.
//  assemble_hess_*       per-row Lagrangian blocks -> tril values / dense (n,n)  (ipopt.py:66-86)
#include "nempc_internal.h"
#include "kernels_obj_impl.h"

namespace nempc {

namespace {

template <typename T>
__device__ __forceinline__ T map_value(int32_t code, const T* __restrict__ tile) {
    if (code >= 0) return tile[code];
    return code == MAP_ZERO ? T(0) : (code == MAP_MINUS_ONE ? T(-1) : T(1));
}

template <typename T>
__global__ __launch_bounds__(64) void objective_kernel(int B, int H, int nx, int nu, ObjOffsets o,
                                                       const T* __restrict__ P, const T* __restrict__ Z,
                                                       T* __restrict__ f, T* __restrict__ grad) {
    if ((int)blockIdx.x < B) objective_body<T>(blockIdx.x, threadIdx.x, H, nx, nu, o, P, Z, f, grad);
}

// grid (ceil(m*n / (256 * kAsmVPT * 16/sizeof(T))), B)
// each lane produces VPT x 16 bytes (2 doubles / 4 floats per store), the VPT stores of a lane 256 lanes apart so
// that every store instruction of a wave is contiguous; the map / tile loads of all VPT vectors are issued
// before the first store (memory-level parallelism instead of one dependent L2 round trip per store)
constexpr int kAsmVPT = 4;
template <typename T>
__device__ __forceinline__ void assemble_dense_body(int b, int mn, int tile_elems, const int32_t* __restrict__ map,
                                                    const T* __restrict__ tiles, T* __restrict__ jac) {
    constexpr int NV = 16 / (int)sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(NV)));
    const T* tile = tiles + (size_t)b * tile_elems;
    T* out = jac + (size_t)b * mn;
    const int base = NV * (blockIdx.x * (256 * kAsmVPT) + threadIdx.x);
    int32_t code[kAsmVPT][NV];
#pragma unroll
    for (int u = 0; u < kAsmVPT; ++u) {
        const int e = base + u * 256 * NV;
#pragma unroll
        for (int k = 0; k < NV; ++k) code[u][k] = (e + k < mn) ? map[e + k] : MAP_ZERO;
    }
    T v[kAsmVPT][NV];
#pragma unroll
    for (int u = 0; u < kAsmVPT; ++u)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[u][k] = map_value<T>(code[u][k], tile);
#pragma unroll
    for (int u = 0; u < kAsmVPT; ++u) {
        const int e = base + u * 256 * NV;
        if (e + NV <= mn && (reinterpret_cast<uintptr_t>(out + e) & 15) == 0) {
            vecT vv;
#pragma unroll
            for (int k = 0; k < NV; ++k) vv[k] = v[u][k];
            *reinterpret_cast<vecT*>(out + e) = vv;
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (e + k < mn) out[e + k] = v[u][k];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void assemble_dense_kernel(int mn, int tile_elems, const int32_t* __restrict__ map,
                                                             const T* __restrict__ tiles, T* __restrict__ jac) {
    assemble_dense_body<T>(blockIdx.y, mn, tile_elems, map, tiles, jac);
}

// objective + dense assembly in ONE launch (they are independent; a second launch costs ~1.5 us of
// boundary plus the objective's own ~4.7 us of latency-bound time at B=1024): blocks x < nb_asm stream
// the Jacobian of problem y, the extra block x == nb_asm evaluates the objective of problem y on wave 0.
template <typename T>
__global__ __launch_bounds__(256) void post_kernel(int nb_asm, int mn, int tile_elems, const int32_t* __restrict__ map,
                                                   const T* __restrict__ tiles, T* __restrict__ jac, int H, int nx,
                                                   int nu, ObjOffsets o, const T* __restrict__ P,
                                                   const T* __restrict__ Z, T* __restrict__ f, T* __restrict__ grad) {
    if ((int)blockIdx.x < nb_asm) {
        assemble_dense_body<T>(blockIdx.y, mn, tile_elems, map, tiles, jac);
    } else if (threadIdx.x < 64) {
        objective_body<T>(blockIdx.y, threadIdx.x, H, nx, nu, o, P, Z, f, grad);
    }
}

// Flat variant: the (B, m*n) output is ONE stream of 16-byte vectors cut into full blocks (a per-problem grid leaves
// the last block of every problem mostly idle: m*n = 2400 is 1.17 blocks).  Needs m*n % (16/sizeof(T)) == 0 so that a
// vector never straddles two problems.  Problem index of a vector: one exact division per block, then a
// multiply-high by a host-checked reciprocal for the in-block offset.  Blocks >= nb_asm: objective, one problem per wave.
template <typename T>
__global__ __launch_bounds__(256) void post_flat_kernel(int nb_asm, int B, int mn, unsigned magic, int tile_elems,
                                                        const int32_t* __restrict__ map, const T* __restrict__ tiles,
                                                        T* __restrict__ jac, int H, int nx, int nu, ObjOffsets o,
                                                        const T* __restrict__ P, const T* __restrict__ Z,
                                                        T* __restrict__ f, T* __restrict__ grad) {
    constexpr int NV = 16 / (int)sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(NV)));
    typedef int32_t vecI __attribute__((ext_vector_type(NV)));
    if ((int)blockIdx.x >= nb_asm) {
        const int b = ((int)blockIdx.x - nb_asm) * 4 + (threadIdx.x >> 6);
        if (b < B && (f || grad)) objective_body<T>(b, threadIdx.x & 63, H, nx, nu, o, P, Z, f, grad);
        return;
    }
    const unsigned long long e_blk = (unsigned long long)blockIdx.x * (256 * kAsmVPT * NV);
    const unsigned b_blk = (unsigned)(e_blk / (unsigned)mn);
    const unsigned r_blk = (unsigned)(e_blk - (unsigned long long)b_blk * (unsigned)mn);
    vecI code[kAsmVPT];
    unsigned bb[kAsmVPT], ee[kAsmVPT];
#pragma unroll
    for (int u = 0; u < kAsmVPT; ++u) {
        const unsigned numer = r_blk + (unsigned)(u * 256 + threadIdx.x) * NV;
        const unsigned db = __umulhi(numer, magic);
        ee[u] = numer - db * (unsigned)mn;
        bb[u] = b_blk + db;
        if (bb[u] < (unsigned)B) code[u] = *reinterpret_cast<const vecI*>(map + ee[u]);
    }
    vecT v[kAsmVPT];
#pragma unroll
    for (int u = 0; u < kAsmVPT; ++u) {
        if (bb[u] < (unsigned)B) {
            const T* tile = tiles + (size_t)bb[u] * tile_elems;
#pragma unroll
            for (int k = 0; k < NV; ++k) v[u][k] = map_value<T>(code[u][k], tile);
        }
    }
#pragma unroll
    for (int u = 0; u < kAsmVPT; ++u)
        if (bb[u] < (unsigned)B) *reinterpret_cast<vecT*>(jac + (size_t)bb[u] * mn + ee[u]) = v[u];
}

// host side: can the flat kernel handle (B, mn)?  magic = ceil(2^32 / mn) is exact for numerators < 2^32 / (magic*mn - 2^32)
inline bool flat_ok(const Handle& h, int B, const void* jac, unsigned* magic) {
    const unsigned mn = (unsigned)h.m * (unsigned)h.n;
    const unsigned NV = 16 / (unsigned)h.esz;
    if (mn % NV != 0 || (reinterpret_cast<uintptr_t>(jac) & 15) != 0) return false;
    if ((unsigned long long)B * mn >= (1ull << 32)) return false;
    const unsigned long long mg = ((1ull << 32) + mn - 1) / mn;
    const unsigned long long err = mg * mn - (1ull << 32);          // < mn
    const unsigned long long max_numer = (unsigned long long)mn + 256ull * kAsmVPT * NV;
    if (mg >= (1ull << 32) || err * max_numer >= (1ull << 32)) return false;
    *magic = (unsigned)mg;
    return true;
}

// sparse contract in one launch: band-pattern values for all problems as one flat stream (blocks < nb_asm), objective
// on the blocks after them (one problem per wave).  b = e / nnz by an exact 64-bit multiply-high.
template <typename T>
__global__ __launch_bounds__(256) void post_sparse_kernel(int nb_asm, int B, int nnz, unsigned long long magic64,
                                                          int tile_elems, const int32_t* __restrict__ map,
                                                          const T* __restrict__ tiles, T* __restrict__ vals, int H, int nx,
                                                          int nu, ObjOffsets o, const T* __restrict__ P,
                                                          const T* __restrict__ Z, T* __restrict__ f, T* __restrict__ grad) {
    if ((int)blockIdx.x >= nb_asm) {
        const int b = ((int)blockIdx.x - nb_asm) * 4 + (threadIdx.x >> 6);
        if (b < B) objective_body<T>(b, threadIdx.x & 63, H, nx, nu, o, P, Z, f, grad);
        return;
    }
    const unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (unsigned long long)B * nnz) return;
    const unsigned b = (unsigned)__umul64hi(e, magic64);      // e / nnz, exact for e * nnz < 2^64
    const unsigned k = (unsigned)(e - (unsigned long long)b * nnz);
    vals[e] = map_value<T>(map[k], tiles + (size_t)b * tile_elems);
}

template <typename T>
__global__ __launch_bounds__(256) void assemble_sparse_kernel(int nnz, int tile_elems, const int32_t* __restrict__ map,
                                                              const T* __restrict__ tiles, T* __restrict__ vals) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nnz) vals[(size_t)b * nnz + e] = map_value<T>(map[e], tiles + (size_t)b * tile_elems);
}

// hvals[b][e] = sigma_b * objc[e] + sum_k (map[e][k] >= 0 ? blocks[b][map[e][k]] : 0);  up to w per-row blocks reach
// one entry with a rolling window of w steps (the reference sums them with its projection matrices,
// model/tensorflow.py:312-330); w = 1 for plain models
template <typename T>
__global__ __launch_bounds__(256) void assemble_hess_kernel(int cnt, int w, int blk_elems, const int32_t* __restrict__ map,
                                                            const T* __restrict__ objc, const T* __restrict__ blocks,
                                                            const T* __restrict__ sigma, T* __restrict__ out) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < cnt) {
        T v = sigma[b] * objc[e];
        for (int k = 0; k < w; ++k) {
            const int32_t c = map[(size_t)e * w + k];
            if (c >= 0) v += blocks[(size_t)b * blk_elems + c];
        }
        out[(size_t)b * cnt + e] = v;
    }
}

// The Gauss-Newton form of the same assembly without the intermediate blocks: a block element is formed where it is
// used, from the row kernel's tiles, with gn_blocks_kernel's arithmetic (product first, then the weighted sum over the
// nx outputs of the row) -- one launch and one (B, H, nin, nin) round trip through memory less per callback
template <typename T>
__global__ __launch_bounds__(256) void assemble_hess_gn_kernel(int cnt, int w, int rows, int nx, int nin,
                                                               const int32_t* __restrict__ map, const T* __restrict__ objc,
                                                               const T* __restrict__ tiles, const T* __restrict__ wgt,
                                                               const T* __restrict__ sigma, T* __restrict__ out) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cnt) return;
    const int nn = nin * nin;
    T v = sigma[b] * objc[e];
    for (int k = 0; k < w; ++k) {
        const int32_t c = map[(size_t)e * w + k];
        if (c < 0) continue;
        const int r = c / nn, pq = c - r * nn, p = pq / nin, q = pq - p * nin;
        const T* t = tiles + ((size_t)b * rows + r) * (size_t)(nx * nin);
        const T* wr = wgt ? wgt + ((size_t)b * rows + r) * nx : nullptr;
        T s = T(0);
        for (int i = 0; i < nx; ++i) {
            const T tt = t[i * nin + p] * t[i * nin + q];
            s = fma(wr ? wr[i] : T(1), tt, s);
        }
        v += s;
    }
    out[(size_t)b * cnt + e] = v;
}

}  // namespace

int launch_objective(Handle& h, int B, const void* Z, void* f, void* grad, hipStream_t s) {
    ObjOffsets o = obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu);
    if (h.cfg.dtype == NEMPC_F64)
        hipLaunchKernelGGL(objective_kernel<double>, dim3(B), dim3(64), 0, s, B, h.cfg.H, h.cfg.nx, h.cfg.nu, o,
                           (const double*)h.d_obj, (const double*)Z, (double*)f, (double*)grad);
    else
        hipLaunchKernelGGL(objective_kernel<float>, dim3(B), dim3(64), 0, s, B, h.cfg.H, h.cfg.nx, h.cfg.nu, o,
                           (const float*)h.d_obj, (const float*)Z, (float*)f, (float*)grad);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

template <typename T>
int launch_post_flat(Handle& h, int B, unsigned magic, const void* tiles, void* jac, const void* Z, void* f, void* grad,
                     hipStream_t s) {
    const int mn = h.m * h.n;
    const int te = h.cfg.H * h.cfg.nx * h.nin;
    const int NV = 16 / (int)sizeof(T);
    const long long total = (long long)B * mn;
    const int nb_asm = (int)((total + 256LL * kAsmVPT * NV - 1) / (256LL * kAsmVPT * NV));
    const int nb_obj = (f || grad) ? (B + 3) / 4 : 0;
    ObjOffsets o = obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu);
    hipLaunchKernelGGL(post_flat_kernel<T>, dim3((unsigned)(nb_asm + nb_obj)), dim3(256), 0, s, nb_asm, B, mn, magic, te,
                       h.d_dense_map, (const T*)tiles, (T*)jac, h.cfg.H, h.cfg.nx, h.cfg.nu, o, (const T*)h.d_obj,
                       (const T*)Z, (T*)f, (T*)grad);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

int launch_assemble_dense(Handle& h, int B, const void* tiles, void* jac, hipStream_t s) {
    unsigned magic = 0;
    if (flat_ok(h, B, jac, &magic))
        return h.cfg.dtype == NEMPC_F64 ? launch_post_flat<double>(h, B, magic, tiles, jac, nullptr, nullptr, nullptr, s)
                                        : launch_post_flat<float>(h, B, magic, tiles, jac, nullptr, nullptr, nullptr, s);
    const int mn = h.m * h.n;
    const int te = h.cfg.H * h.cfg.nx * h.nin;
    const int per_block = 256 * kAsmVPT * (16 / (int)h.esz);
    const dim3 block(256), grid((unsigned)((mn + per_block - 1) / per_block), (unsigned)B);
    if (h.cfg.dtype == NEMPC_F64)
        hipLaunchKernelGGL(assemble_dense_kernel<double>, grid, block, 0, s, mn, te, h.d_dense_map,
                           (const double*)tiles, (double*)jac);
    else
        hipLaunchKernelGGL(assemble_dense_kernel<float>, grid, block, 0, s, mn, te, h.d_dense_map,
                           (const float*)tiles, (float*)jac);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

int launch_post(Handle& h, int B, const void* tiles, void* jac, const void* Z, void* f, void* grad, hipStream_t s) {
    unsigned magic = 0;
    if (flat_ok(h, B, jac, &magic))
        return h.cfg.dtype == NEMPC_F64 ? launch_post_flat<double>(h, B, magic, tiles, jac, Z, f, grad, s)
                                        : launch_post_flat<float>(h, B, magic, tiles, jac, Z, f, grad, s);
    const int mn = h.m * h.n;
    const int te = h.cfg.H * h.cfg.nx * h.nin;
    const int per_block = 256 * kAsmVPT * (16 / (int)h.esz);
    const int nb = (mn + per_block - 1) / per_block;
    ObjOffsets o = obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu);
    const dim3 block(256), grid((unsigned)(nb + 1), (unsigned)B);
    if (h.cfg.dtype == NEMPC_F64)
        hipLaunchKernelGGL(post_kernel<double>, grid, block, 0, s, nb, mn, te, h.d_dense_map, (const double*)tiles,
                           (double*)jac, h.cfg.H, h.cfg.nx, h.cfg.nu, o, (const double*)h.d_obj, (const double*)Z,
                           (double*)f, (double*)grad);
    else
        hipLaunchKernelGGL(post_kernel<float>, grid, block, 0, s, nb, mn, te, h.d_dense_map, (const float*)tiles,
                           (float*)jac, h.cfg.H, h.cfg.nx, h.cfg.nu, o, (const float*)h.d_obj, (const float*)Z,
                           (float*)f, (float*)grad);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

int launch_post_sparse(Handle& h, int B, const void* tiles, void* vals, const void* Z, void* f, void* grad,
                       hipStream_t s) {
    const int nnz = (int)h.jac_rows.size();
    const int te = h.cfg.H * h.cfg.nx * h.nin;
    const unsigned long long total = (unsigned long long)B * nnz;
    const int nb_asm = (int)((total + 255) / 256), nb_obj = (B + 3) / 4;
    // ceil(2^64 / nnz): floor((2^64 - 1) / nnz) + 1 (nnz >= 2 always: every row has its -1 and a control column)
    const unsigned long long magic = ~0ull / (unsigned long long)nnz + 1;
    ObjOffsets o = obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu);
    if (h.cfg.dtype == NEMPC_F64)
        hipLaunchKernelGGL(post_sparse_kernel<double>, dim3((unsigned)(nb_asm + nb_obj)), dim3(256), 0, s, nb_asm, B, nnz,
                           magic, te, h.d_sparse_map, (const double*)tiles, (double*)vals, h.cfg.H, h.cfg.nx, h.cfg.nu, o,
                           (const double*)h.d_obj, (const double*)Z, (double*)f, (double*)grad);
    else
        hipLaunchKernelGGL(post_sparse_kernel<float>, dim3((unsigned)(nb_asm + nb_obj)), dim3(256), 0, s, nb_asm, B, nnz,
                           magic, te, h.d_sparse_map, (const float*)tiles, (float*)vals, h.cfg.H, h.cfg.nx, h.cfg.nu, o,
                           (const float*)h.d_obj, (const float*)Z, (float*)f, (float*)grad);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

int launch_assemble_sparse(Handle& h, int B, const void* tiles, void* vals, hipStream_t s) {
    const int nnz = (int)h.jac_rows.size();
    const int te = h.cfg.H * h.cfg.nx * h.nin;
    const dim3 block(256), grid((unsigned)((nnz + 255) / 256), (unsigned)B);
    if (h.cfg.dtype == NEMPC_F64)
        hipLaunchKernelGGL(assemble_sparse_kernel<double>, grid, block, 0, s, nnz, te, h.d_sparse_map,
                           (const double*)tiles, (double*)vals);
    else
        hipLaunchKernelGGL(assemble_sparse_kernel<float>, grid, block, 0, s, nnz, te, h.d_sparse_map,
                           (const float*)tiles, (float*)vals);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

// Gauss-Newton blocks: blocks[r][p][q] = sum_k w[r][k] * T[r][k][p] * T[r][k][q] from the row kernel's tiles T (R, nx, nin);
// w == null means unit weights.  The product T_p * T_q is formed first, so the block is symmetric to the last bit.
template <typename T>
__global__ __launch_bounds__(256) void gn_blocks_kernel(size_t total, int nx, int nin, const T* __restrict__ tiles,
                                                        const T* __restrict__ w, T* __restrict__ blocks) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int nn = nin * nin;
    const size_t r = idx / nn;
    const int pq = (int)(idx - r * nn), p = pq / nin, q = pq - p * nin;
    const T* t = tiles + r * (size_t)(nx * nin);
    T s = T(0);
    for (int k = 0; k < nx; ++k) {
        const T tt = t[k * nin + p] * t[k * nin + q];
        s = fma(w ? w[r * nx + k] : T(1), tt, s);
    }
    blocks[idx] = s;
}

int launch_gn_blocks(Handle& h, int B, const void* tiles, const void* w, void* blocks, hipStream_t s) {
    const size_t total = (size_t)B * h.cfg.H * h.nin * h.nin;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (h.cfg.dtype == NEMPC_F64)
        hipLaunchKernelGGL(gn_blocks_kernel<double>, grid, block, 0, s, total, h.cfg.nx, h.nin, (const double*)tiles,
                           (const double*)w, (double*)blocks);
    else
        hipLaunchKernelGGL(gn_blocks_kernel<float>, grid, block, 0, s, total, h.cfg.nx, h.nin, (const float*)tiles,
                           (const float*)w, (float*)blocks);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

// d_hess_map layout (w codes per entry): [0,nnz) tril map | [nnz, nnz+n*n) dense map ; objc follows the same split in d_hess_objc
int launch_assemble_hess(Handle& h, int B, const void* blocks, const void* sigma, void* hvals, void* hdense,
                         hipStream_t s) {
    const int nnz = (int)h.hess_rows.size();
    const int nn = h.n * h.n;
    const int be = h.cfg.H * h.nin * h.nin;
    const dim3 block(256);
    const size_t esz = h.esz;
    const char* objc = (const char*)h.d_obj + (size_t)obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu).total * esz;
    if (hvals) {
        const dim3 grid((unsigned)((nnz + 255) / 256), (unsigned)B);
        if (h.cfg.dtype == NEMPC_F64)
            hipLaunchKernelGGL(assemble_hess_kernel<double>, grid, block, 0, s, nnz, h.w, be, h.d_hess_map,
                               (const double*)objc, (const double*)blocks, (const double*)sigma, (double*)hvals);
        else
            hipLaunchKernelGGL(assemble_hess_kernel<float>, grid, block, 0, s, nnz, h.w, be, h.d_hess_map,
                               (const float*)objc, (const float*)blocks, (const float*)sigma, (float*)hvals);
    }
    if (hdense) {
        const dim3 grid((unsigned)((nn + 255) / 256), (unsigned)B);
        if (h.cfg.dtype == NEMPC_F64)
            hipLaunchKernelGGL(assemble_hess_kernel<double>, grid, block, 0, s, nn, h.w, be, h.d_hess_map + (size_t)nnz * h.w,
                               (const double*)(objc + (size_t)nnz * esz), (const double*)blocks,
                               (const double*)sigma, (double*)hdense);
        else
            hipLaunchKernelGGL(assemble_hess_kernel<float>, grid, block, 0, s, nn, h.w, be, h.d_hess_map + (size_t)nnz * h.w,
                               (const float*)(objc + (size_t)nnz * esz), (const float*)blocks, (const float*)sigma,
                               (float*)hdense);
    }
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

int launch_assemble_hess_gn(Handle& h, int B, const void* tiles, const void* w, const void* sigma, void* hvals, void* hdense,
                            hipStream_t s) {
    const int nnz = (int)h.hess_rows.size();
    const int nn = h.n * h.n;
    const dim3 block(256);
    const size_t esz = h.esz;
    const char* objc = (const char*)h.d_obj + (size_t)obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu).total * esz;
    auto go = [&](int cnt, const int32_t* map, const char* oc, void* out) {
        const dim3 grid((unsigned)((cnt + 255) / 256), (unsigned)B);
        if (h.cfg.dtype == NEMPC_F64)
            hipLaunchKernelGGL(assemble_hess_gn_kernel<double>, grid, block, 0, s, cnt, h.w, h.cfg.H, h.cfg.nx, h.nin, map,
                               (const double*)oc, (const double*)tiles, (const double*)w, (const double*)sigma, (double*)out);
        else
            hipLaunchKernelGGL(assemble_hess_gn_kernel<float>, grid, block, 0, s, cnt, h.w, h.cfg.H, h.cfg.nx, h.nin, map,
                               (const float*)oc, (const float*)tiles, (const float*)w, (const float*)sigma, (float*)out);
    };
    if (hvals) go(nnz, h.d_hess_map, objc, hvals);
    if (hdense) go(nn, h.d_hess_map + (size_t)nnz * h.w, objc + (size_t)nnz * esz, hdense);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

}  // namespace nempc
