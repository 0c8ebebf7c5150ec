// Layer-at-a-time matrix-core path (gfx950) for networks the register-resident kernels do not take: hidden widths up to
// 1024, up to NEMPC_MAX_LAYERS dense layers, any activation per layer (the output layer included).
//
// The reference wraps ANY feed-forward Keras model (model/tensorflow.py:8-29,49-51) and differentiates it per row
// (tensorflow.py:53-75).  The fused row kernels of this library keep a network's weight slices in registers, which ends at
// three hidden layers of width 128 with one activation; everything else used to run on the thread-per-row kernel
// (rows_valu_kernel: ~50x slower).  For those networks a dense layer over all B*H rows is a GEMM large enough to stand on
// its own -- (B*H) x width x width -- so the network is walked one layer per launch:
//
//   forward   X_l = s_l(X_{l-1} W_l + b_l),  D_l = s_l'(z_l)                 one GEMM per layer, activation in the epilogue
//   reverse   G_{L-2} = (W_{L-1} e_k s_{L-1}') . D_{L-2}                    seed: all nx cotangents side by side
//             G_{l-1} = (G_l W_l^T) . D_{l-1}                               one GEMM per layer over nx * (B*H) columns
//             J       = G_0 W_0^T                                           skinny: onto the nin inputs
//   integrator algebra (discret.py:27,52-56 / unity.py:29 / rk4.py:69-80,147-159) per row, then the same g / compact-tile
//   outputs as the row kernels; the dense / sparse / objective launches of nempc_eval follow unchanged.
//
// Layout: every activation matrix is stored FEATURE-MAJOR, X^T[feature][row] with a row stride Rp (multiple of 64): the
// 16x16x4 matrix instructions compute Z^T (features x rows) = W^T (features x k) . X^T (k x rows), so a result register is
// 16 consecutive rows of one feature -- a coalesced store -- and the next layer's operand tile is a plain 2-D sub-block of
// X^T: no transposition anywhere, and the weights are used as nempc_set_weights left them (W row-major (in, out) forward,
// W^T row-major (out, in) reverse: d_W / d_Wt of the generic kernel).
//
// GEMM kernel: 256 threads = 4 waves own a 64 (features) x 64 (rows) block, each wave 32 x 32 = 2 x 2 accumulator tiles;
// K in chunks of 16 through double-buffered LDS (global loads of chunk c+1 in flight under the matrix instructions of
// chunk c).  Per chunk and wave: 16 matrix instructions (1024 cycles in fp64), 16 ds_read, 8 global loads.  A v_mfma_f64
// holds the vector pipe for its 64 cycles, so the bound is the matrix pipe; arbitrary M, N, K (edge tiles are
// zero-filled on load and masked on store).
#include <cstdlib>
#include <type_traits>

#include "activations.h"
#include "kernels_mfma_impl.h"
#include "nempc_internal.h"

namespace nempc {

namespace {

constexpr int LG_PAD = 16;     // elements of padding per LDS tile row (0 and 8 measured slower, round 4)
constexpr int LG_WPE = 4;      // waves per SIMD the GEMM kernel's register allocation must allow
constexpr int LG_BM = 64, LG_BN = 64, LG_BK = 16;

enum { LG_FORWARD = 0, LG_REVERSE = 1 };

template <typename T>
__device__ __forceinline__ T lg_act_f(int code, T x, T par) {
    if (code == NEMPC_ACT_TANH) return Act<T, NEMPC_ACT_TANH>::f(x);     // the 24-slot tanh of the row kernels
    return act_f<T>(code, x, par);
}

// a layer's output, s'(z) and (want_e) s''(z) from its pre-activation: from the output for the monotone activations (the
// bits of the register-resident kernels), from z itself for swish / gelu
template <typename T>
__device__ __forceinline__ void lg_act_all(int code, T z, T par, bool want_e, T& a, T& d1, T& e) {
    if (act_zbased(code)) {
        act_from_z<T>(code, z, a, d1, e);
    } else {
        a = lg_act_f<T>(code, z, par);
        d1 = act_d1<T>(code, a, par);
        e = want_e ? act_r2<T>(code, a, par) * d1 : T(0);
    }
}
// ... of NV pre-activations at once with the (wave-uniform) switch over the code taken once for the common activations (see
// lg_dval_n); the formulas -- and the bits -- are lg_act_all's
template <typename T, int NV>
__device__ __forceinline__ void lg_act_all_n(int code, T par, bool want_e, const T (&z)[NV], T (&a)[NV], T (&d1)[NV], T (&e)[NV]) {
#define LG_ACT_CASE(CODE)                                                                                     \
    case CODE:                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NV; ++i) lg_act_all<T>(CODE, z[i], par, want_e, a[i], d1[i], e[i]); \
        break;
    switch (code) {
        LG_ACT_CASE(NEMPC_ACT_TANH)
        LG_ACT_CASE(NEMPC_ACT_RELU)
        LG_ACT_CASE(NEMPC_ACT_SIGMOID)
        LG_ACT_CASE(NEMPC_ACT_LINEAR)
        default:
#pragma unroll
            for (int i = 0; i < NV; ++i) lg_act_all<T>(code, z[i], par, want_e, a[i], d1[i], e[i]);
            break;
    }
#undef LG_ACT_CASE
}

// Derivatives from the stored ACTIVATION (round 5).  For the activations whose s' is a cheap function of the output (tanh, relu, sigmoid, elu, leaky_relu, selu -- not softplus, whose s' costs an exp, and not the ones written from the
// pre-activation) a layer stores a = s(z) only; whoever needs s'(z) or s''(z) later reads a and forms d1(a) / r2(a) d1(a) in
// its own loader / epilogue.  A forward product then writes one matrix instead of two (rows) or three (Hessian sweeps):
// 2 x 256 at B*H = 20480 in fp64, 42 MB per layer and matrix.
__host__ __device__ inline bool lg_d_from_a(int act) {
    return act == NEMPC_ACT_TANH || act == NEMPC_ACT_RELU || act == NEMPC_ACT_SIGMOID || act == NEMPC_ACT_ELU ||
           act == NEMPC_ACT_LEAKY_RELU || act == NEMPC_ACT_SELU;
}
// what a value read through a "derivative" pointer stands for: code 0 -- the derivative itself (stored as such); else the
// layer's activation, turned into s' (use 0) or s'' (use 1)
// (its own switch over the cheap codes -- the formulas of act_d1 / act_r2, hence their bits -- so that softplus' expm1 is not
// compiled into every GEMM epilogue; NEMPC_ACT_LINEAR is code 0 and needs none: s' = 1 is what a linear layer stores)
template <typename T>
__device__ __forceinline__ T lg_dval(int code, T par, int use, T v) {
    T d1, r2;
    switch (code) {
        case NEMPC_ACT_TANH: d1 = fma(-v, v, T(1)); r2 = T(-2) * v; break;
        case NEMPC_ACT_RELU: d1 = v > T(0) ? T(1) : T(0); r2 = T(0); break;
        case NEMPC_ACT_SIGMOID: d1 = v * (T(1) - v); r2 = T(1) - T(2) * v; break;
        case NEMPC_ACT_ELU: d1 = v > T(0) ? T(1) : v + par; r2 = v > T(0) ? T(0) : (v != v ? v : T(1)); break;
        case NEMPC_ACT_LEAKY_RELU: d1 = v > T(0) ? T(1) : (v != v ? v : par); r2 = T(0); break;
        case NEMPC_ACT_SELU:
            d1 = v > T(0) ? T(NEMPC_SELU_LAMBDA) : v + T(NEMPC_SELU_LAMBDA * NEMPC_SELU_ALPHA);
            r2 = v > T(0) ? T(0) : (v != v ? v : T(1));
            break;
        default: return v;          // the derivative itself was stored
    }
    return use ? r2 * d1 : d1;
}
// ... of NV values at once, the (wave-uniform) switches taken ONCE: per element they are a chain of scalar branches around every
// value -- and while the other workgroups of the CU hold the vector pipe with 64-cycle matrix instructions, every instruction
// of an epilogue costs its wave a pipe slot of that length (tools/diag_stamps_layered.py).  The formulas are lg_dval's.
template <typename T, int NV>
__device__ __forceinline__ void lg_dval_n(int code, T par, int use, T (&v)[NV]) {
#define LG_DVAL_CASE(CODE)                                                                  \
    case CODE:                                                                              \
        if (use) {                                                                          \
            _Pragma("unroll") for (int i = 0; i < NV; ++i) v[i] = lg_dval<T>(CODE, par, 1, v[i]); \
        } else {                                                                            \
            _Pragma("unroll") for (int i = 0; i < NV; ++i) v[i] = lg_dval<T>(CODE, par, 0, v[i]); \
        }                                                                                   \
        break;
    switch (code) {
        LG_DVAL_CASE(NEMPC_ACT_TANH)
        LG_DVAL_CASE(NEMPC_ACT_RELU)
        LG_DVAL_CASE(NEMPC_ACT_SIGMOID)
        LG_DVAL_CASE(NEMPC_ACT_ELU)
        LG_DVAL_CASE(NEMPC_ACT_LEAKY_RELU)
        LG_DVAL_CASE(NEMPC_ACT_SELU)
        default: break;          // the derivatives themselves were stored
    }
#undef LG_DVAL_CASE
}

struct GemmArgs {
    const void* A;      // A^T: (K, M) element (k, m) at A[k * lda + m]
    const void* Bw;     // (K, N) row-major, element (k, n) at Bw[k * ldb + n]
    void* C;            // C^T: (N, M), element (n, m) at C[n * ldc + m]
    void* D;            // forward: s'(z) out, (N, M) like C;  reverse: s'(z) in, (N, Rmod) -- column m reads m % Rmod
    void* E;            // second-order sweeps (null otherwise).  forward: s''(z) out, like D;  reverse: s''(z) in, like D
    void* C2;           // reverse, with E: C2 = acc . E (the curvature weights of the layer), like C
    void* Craw;         // reverse: acc itself (the pre-activation tangents), like C;  C may then be null
    const void* bias;   // forward only, (N)
    long long lda, ldc, ldd;
    int ldb, M, N, K, mode, act;
    int nblk;           // feature blocks (ceil(N / BN)), set by the launcher
    unsigned nblk_magic;    // floor(2^32 / nblk) + 1: slot / nblk = umulhi(slot, magic) for slot < 2^32 / nblk (set by the launcher)
    int ncot, rbc;          // cotangents side by side in M and row blocks per cotangent (launcher; 1, - : plain order)
    unsigned ncot_magic;
    double actp;        // alpha of an elu / leaky_relu layer
    long long Rmod;     // reverse: rows per cotangent block (a multiple of LG_BM, so a block never straddles two)
    // SEED (the first reverse product forms its operand on the fly instead of reading a seed matrix from memory):
    //   A^T[j][k Rp + r] = (W_last[j][k] s_L'(z_L)[k][r]) D_{L-2}^T[j][r];  A = D_{L-2}^T with lda = Rp
    const void* seedW;  // W_{L-1} (width, nx) row-major
    const void* seedDl; // s_L'(z_L)^T (nx, Rmod)
    int seed_nx;
    // CONTRACT (the product's result is contracted with the next, skinny, matrix in the epilogue instead of being written):
    //   Jp[nb][d][m] = sum over the features n of block nb of Wc[n][d] E[n][m],  d < nd <= 32
    //   LG_CONTRACT_REVERSE  the last reverse product: E = C . D_0, Wc = W_0^T (out, in), nd = nin: the Jacobian's partial sums
    //   LG_CONTRACT_FORWARD  the last hidden layer: E = s(C + b) (only s' is stored), Wc = W_{L-1} (in, out), nd = nx: the
    //                        network output's partial sums (bias and activation: layered_outfinish_kernel)
    //   LG_CONTRACT_HPAIR    the Hessian's tangent products with the columns INTERLEAVED (tile = 16 rows x RM = nin inputs:
    //                        column m = (r / 16) 16 nin + p 16 + r % 16): a lane holds P[n][p][row] of all inputs p, so the
    //                        layer's curvature term  sum_n w[n][row] P[n][p][row] P[n][q][row]  (p >= q) is formed from the
    //                        accumulators -- the tangents are not written (Craw = null), no contraction launch reads them back;
    //                        w0t = w_l^T (N, ldd), Jp[nb][pair][row] = the feature block's partial sums, Rmod = valid rows
    const void* w0t;
    int ldw0, nin;
    void* Jp;
    long long ldj, jp_stride;
    // derivatives from the stored activation (lg_d_from_a):
    int dact;           // != 0: D (and E) point at the layer's ACTIVATIONS; the epilogue forms s' / s'' from them
    double dactp;
    int duse;           // CONTRACT_REVERSE with dact: 0 the multiplier is s', 1 it is s'' (layer 0's curvature weights)
    int sact;           // SEED: != 0: the loader's operand A is an activation matrix, s' is formed on the way to LDS
    double sactp;
    int store_a;        // forward: the D slot receives the activation itself (C / E are then not written by CONTRACT_FORWARD)
    long long* dbg;     // diagnostic builds only (-DNEMPC_STAMPS, tools/diag_stamps_layered.py): per-workgroup timeline
};

#ifdef NEMPC_STAMPS
// per-workgroup timeline of ONE of the products of an evaluation (NEMPC_LG_STAMP = 10 SEED + CONTRACT picks it; the last launch
// of that form wins): word idx of the workgroup's 16-word record = shader clock; 13 = the chip's real-time counter at exit,
// 15 = XCC id << 32 | HW_ID (kernels_coop_impl.h, COOP_WGSTAMP)
long long* g_lg_dbg = nullptr;
#define LG_WGSTAMP(idx)                                                                        \
    do {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (a.dbg && threadIdx.x == 0 && blockIdx.x < 4096)                                    \
            a.dbg[1024 + blockIdx.x * 16 + (idx)] = (long long)__builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#define LG_WGSTAMP_EXIT()                                                                      \
    do {                                                                                       \
        LG_WGSTAMP(14);                                                                        \
        if (a.dbg && threadIdx.x == 0 && blockIdx.x < 4096) {                                  \
            a.dbg[1024 + blockIdx.x * 16 + 13] = (long long)__builtin_amdgcn_s_memrealtime();  \
            a.dbg[1024 + blockIdx.x * 16 + 15] = ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); \
        }                                                                                      \
    } while (0)
// time of wave 0 inside the main loop by segment (sums over the chunks, shader clocks): words 5.. of the record
#define LG_SEG_DECL() unsigned long long lg_seg[5] = {0, 0, 0, 0, 0}, lg_tprev = __builtin_amdgcn_s_memtime()
#define LG_SEG(i)                                                       \
    do {                                                                \
        __builtin_amdgcn_sched_barrier(0);                              \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();     \
        lg_seg[i] += _t - lg_tprev;                                     \
        lg_tprev = _t;                                                  \
        __builtin_amdgcn_sched_barrier(0);                              \
    } while (0)
#define LG_SEG_FLUSH()                                                                       \
    do {                                                                                     \
        if (a.dbg && threadIdx.x == 0 && blockIdx.x < 4096)                                  \
            for (int _i = 0; _i < 5; ++_i) a.dbg[1024 + blockIdx.x * 16 + 5 + _i] = (long long)lg_seg[_i]; \
    } while (0)
#else
#define LG_SEG_DECL() \
    do {              \
    } while (0)
#define LG_SEG(i) \
    do {          \
    } while (0)
#define LG_SEG_FLUSH() \
    do {               \
    } while (0)
#define LG_WGSTAMP(idx) \
    do {                \
    } while (0)
#define LG_WGSTAMP_EXIT() \
    do {                  \
    } while (0)
#endif

enum { LG_CONTRACT_NONE = 0, LG_CONTRACT_REVERSE = 1, LG_CONTRACT_FORWARD = 2, LG_CONTRACT_HPAIR = 3 };

// FT = 16-feature tiles per wave: a workgroup owns BN = 64 FT features x 64 rows.  Measured (tools/layered_bench.py, round 4,
// with one chunk of load lead): FT = 1 is the fastest everywhere -- 2 x 256 at B*H = 20480: 398 / 505 / 569 us for FT = 1 / 2 /
// 4 in fp64, 244 / 264 / 304 us in fp32; 4 x 512 RK4 6/3 at B*H = 30720: 34.0 / 41.1 / 44.1 ms (0.51 / 0.42 / 0.39 of the fp64
// matrix peak).  Wider blocks read the activations fewer times but run at two waves per SIMD with coarse launch tails; with
// the XCD-aware block order below the narrow block gets its re-reads from L2 anyway.  Only FT = 1 is instantiated (the wider shapes do not fit the
// two-chunk load lead below into 128 registers).
template <int FT, int RM = 4>
struct LgShape {
    static constexpr int BM = 16 * RM;          // rows (columns of the transposed product) per workgroup: RM 16-row tiles per wave
    static constexpr int BN = 64 * FT;
    static constexpr int BK = FT == 4 ? 8 : 16;
    static constexpr int LDW = BN + LG_PAD;       // (padding: the four k-rows of a fragment read land on different banks;
    static constexpr int LDA = BM + LG_PAD;    //  +8 with four workgroups per CU measured 6 % slower)
    static constexpr int TILE = BK * (LDW + LDA);      // elements per buffer
};

// Operand loads go through buffer descriptors: scalar base + 32-bit lane offset in one instruction (with plain pointers the
// compiler kept a 64-bit pointer per load in vector registers and stepped all of them every chunk), and the range check
// gives the zero fill for free -- a chunk's descriptor covers exactly the rows of K it has, so rows beyond K and whole
// chunks beyond the last one load zeros without a branch (and without a memory access).
__device__ __forceinline__ double lg_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, double) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(double, (u2)__builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, 0, 0));
}
__device__ __forceinline__ float lg_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, float) {
    return __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}
// A chunk's descriptor, on the SCALAR unit end to end.  Every input is wave-uniform; written as a C clamp, min(max(rows, 0),
// rows_max) was selected as v_med3_i32 (there is no scalar med3), its product as v_mul_lo_u32, and both came back through
// v_readfirstlane -- five vector instructions per chunk and operand, each of which waits for a slot between the other
// workgroups' 64-cycle matrix instructions: a third of the main loop's time (tools/diag_stamps_layered.py, "issue loads").
__device__ __forceinline__ int lg_sclamp(int v, int hi) {       // min(max(v, 0), hi), scalar
    int r;
    asm("s_max_i32 %0, %1, 0\n\ts_min_i32 %0, %0, %2" : "=&s"(r) : "s"(v), "s"(hi) : "scc");
    return r;
}
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lg_rows_rsrc(const T* base, int rows, int rows_max, int ld_bytes) {
    const int bytes = lg_sclamp(rows, rows_max) * ld_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, bytes, 0x00020000);
}
// a pointer the whole wave agrees on, in scalar registers (once, in front of the loop: what is derived from it stays scalar)
template <typename T>
__device__ __forceinline__ const T* lg_uniform(const T* p) {
    const unsigned long long pv = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pv), hi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}

// (Round 5 tried RESIDENT workgroups -- as many as the chip holds, each walking blocks blockIdx.x, blockIdx.x + gridDim.x, ... -- to
// save the dispatch of a new workgroup and its cold argument loads, 2 - 4 us of a block's 45 (tools/diag_stamps_layered.py).  With the
// block body inlined into the loop the register allocation of the fp64 forms collapses (300 - 900 B of scratch, spills inside the
// main loop: 2 x 256 evaluations 238 -> 385 us with one block per workgroup, 337 us resident -- so residency itself is worth 12 %);
// with the argument block re-read through an opaque kernarg pointer every iteration it still spills (120 - 930 B); as a
// __noinline__ function per block (arguments made uniform on entry) the main loop is clean but the evaluation takes 289 - 300 us.
// Not kept; profiles/r05_layered_resident.txt.  Wave priorities (`s_setprio`) either way -- a block's start and end ahead of the
// other blocks' main loops, or behind them -- are slower than the oldest-first default: 2 x 256 fp64 197 -> 209 / 205 us.)
template <typename T, int FT, bool SEED = false, int CONTRACT = LG_CONTRACT_NONE, int RM = 4>
__global__ __launch_bounds__(256, RM == 2 ? (CONTRACT ? 5 : 6) : LG_WPE) void layered_gemm_kernel(GemmArgs a) {

    constexpr int BM = 16 * RM;
    constexpr bool IL = CONTRACT == LG_CONTRACT_HPAIR;       // interleaved columns: tile = 16 rows x RM inputs
    static_assert(!(SEED || CONTRACT) || FT == 1, "the fused forms exist for the 64-feature block only");
    using Ops = MfmaOps<T>;
    using V4 = typename Ops::V4;
    using S = LgShape<FT, RM>;
    constexpr int BN = S::BN, BK = S::BK, LDW = S::LDW, LDA = S::LDA;
    extern __shared__ __attribute__((aligned(16))) unsigned char lg_lds_raw[];
    T* const lds = reinterpret_cast<T*>(lg_lds_raw);
    auto Ws = [&](int buf, int k, int x) -> T& { return lds[buf * S::TILE + k * LDW + x]; };
    auto As = [&](int buf, int k, int x) -> T& { return lds[buf * S::TILE + BK * LDW + k * LDA + x]; };
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    // workgroup -> block, XCD-aware: the dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs, each with
    // an L2 of its own.  XCD x takes the row blocks = x (mod 8), and runs the NB feature blocks of one row block back to
    // back: the row operand (the activations) then comes from HBM once and from that XCD's L2 for the other NB - 1 feature
    // blocks.  (Row blocks fastest, as a plain 2-D grid has it, re-read the activations from memory once per feature block:
    // 3.4 TB/s for a 256 x 256 layer at B*H = 20480 -- the kernel was bandwidth-bound at 0.47 of the matrix peak.)
    // (The start of a workgroup runs beside three others that hold the vector pipe with 64-cycle matrix instructions: every
    // vector instruction up here costs it such a slot.  So the block decomposition stays on the scalar unit -- the division by
    // the number of feature blocks is a multiply-high with the launcher's reciprocal, the cotangent of a block a short
    // subtraction loop -- where the compiler's integer divisions went through the vector unit's reciprocal.)
    const int NB = a.nblk;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int sq = NB == 1 ? slot : (int)__builtin_amdgcn_readfirstlane((int)__umulhi((unsigned)slot, a.nblk_magic));       // slot / NB
    const int nb = slot - sq * NB;
    // reverse products over several cotangents (column m = cotangent * Rmod + row): the cotangents of ONE row block run back to
    // back on the same XCD -- they read the same columns of the derivative matrix, which then comes from memory once and from
    // that XCD's L2 for the other cotangents (cotangent-major, the second cotangent's pass came 1,280 workgroups later:
    // L2 hit rate 0.74, 190 MB from memory for an 84 MB operand)
    int mb;
    if (a.ncot > 1) {
        const int sg = (int)__builtin_amdgcn_readfirstlane((int)__umulhi((unsigned)sq, a.ncot_magic));      // sq / ncot
        const int cot = sq - sg * a.ncot, mbl = sg * 8 + xcd;
        if (mbl >= a.rbc) return;
        mb = cot * a.rbc + mbl;
    } else {
        mb = sq * 8 + xcd;
    }
    const long long m0 = (long long)mb * BM;
    if (m0 >= a.M) return;
    LG_WGSTAMP(0);
    const int n0 = nb * BN;
    const T* __restrict__ A = static_cast<const T*>(a.A);
    const T* __restrict__ Bw = static_cast<const T*>(a.Bw);
    const int K = a.K, N = a.N;
    const long long M = a.M;

    // loader: a wave-uniform base that steps by a chunk on the scalar unit plus per-thread 32-bit element offsets that never
    // change -- no vector arithmetic per load (a v_mfma_f64 holds the vector pipe for its 64 cycles).  Columns beyond N / M
    // are clamped onto the last one (their results are never stored); only the LAST chunk, where k may run past K, is masked.
    constexpr int NW = BK * BN / 256, NA = BK * BM / 256;       // elements per thread and chunk
    // (unsigned BYTE offsets inside a chunk; largest: 15 rows of 16 x 65536 elements of 8 bytes, 126 MB)
    unsigned offW[NW], offA[NA];
    {
        // element u of a thread sits 256 u / BN rows further down the chunk: one offset and scalar steps (64- and 32-wide tiles)
        static_assert(256 % BN == 0, "a thread's elements of the weight tile are whole rows apart");
        const int kk = tid / BN, x = tid % BN;
        const unsigned o0 = (unsigned)(kk * a.ldb + (n0 + x < N ? x : N - 1 - n0)) * (unsigned)sizeof(T);
        const unsigned step = (unsigned)((256 / BN) * a.ldb) * (unsigned)sizeof(T);
#pragma unroll
        for (int u = 0; u < NW; ++u) offW[u] = o0 + (unsigned)u * step;
    }
    if constexpr (256 % BM == 0 && !(IL && SEED)) {
        const int kk = tid / BM, x = tid % BM;
        const unsigned o0 = (unsigned)((long long)kk * a.lda + (m0 + x < M ? x : M - 1 - m0)) * (unsigned)sizeof(T);
        const unsigned step = (unsigned)((long long)(256 / BM) * a.lda) * (unsigned)sizeof(T);
#pragma unroll
        for (int u = 0; u < NA; ++u) offA[u] = o0 + (unsigned)u * step;
    } else {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int e = tid + 256 * u, kk = e / BM, x = e % BM;
            if constexpr (IL && SEED) {
                // the operand is D_0^T (or a_0^T): column x of the tile reads row 16 mb + x % 16, whatever its input x / 16
                const long long row = (long long)mb * 16 + (x & 15);
                offA[u] = (unsigned)((long long)kk * a.lda + (row < a.Rmod ? row : a.Rmod - 1)) * (unsigned)sizeof(T);
            } else {
                offA[u] = (unsigned)((long long)kk * a.lda + (m0 + x < M ? x : M - 1 - m0)) * (unsigned)sizeof(T);
            }
        }
    }
    const T* __restrict__ Wb = Bw + n0;
    // SEED: the operand's column m = (cotangent m / Rmod, row m % Rmod) reads column m % Rmod of D_{L-2}^T; a block of 64
    // columns lies inside one cotangent (Rmod is a multiple of 64)
    long long mrow0 = 0;
    int mcot = 0;           // m0 = mcot Rmod + mrow0 (at most nx - 1, resp. nin - 1, subtractions)
    if (!IL && (SEED || a.mode == LG_REVERSE)) {
        mrow0 = m0;
        while (mrow0 >= a.Rmod) { mrow0 -= a.Rmod; ++mcot; }
    }
    const T* __restrict__ Ab = A + (IL ? (SEED ? 0 : m0) : (SEED ? mrow0 : m0));
    // A chunk on its way from memory to LDS.  Two of them: the loads of chunk c + 2 are issued at the start of chunk c and
    // written to LDS at the end of chunk c + 1 -- two chunks of matrix instructions (~3 us with four waves on the SIMD) to
    // cover a memory round trip under load.  With a single set (one chunk of lead) the waves of the 256 x 256 reverse product
    // sat in s_waitcnt for 56 % of their cycles (profiles/r04_layered_gemm_pipe.txt).  Measured, whole evaluations: 4 x 512
    // RK4 28.4 -> 25.0 ms in fp64 and 16.6 -> 12.1 ms in fp32, 2 x 256 267 -> 252 us and 157 -> 133 us.
    struct ChunkRegs {
        T rw[NW], ra[NA], sw[NA];
    };
    // SEED: thread (w, x) loads rows w, w + 4, ... of every chunk, always of column x: s_L'(z_L) of its column is fetched once,
    // W_last[j][cotangent] is a scalar load per row and chunk (w is wave-uniform)
    T seed_dl = T(0);
    int seed_cot = 0;
    // row of the chunk this thread's u-th element sits in: tid / BM + (256 / BM) u -- wave-uniform for the 64-row block (a
    // scalar load of W_last then), two rows per wave for the 32-row block
    const int krow0 = RM == 4 ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid / BM;
    if constexpr (SEED && !IL) {
        const int x = tid % BM;
        seed_cot = mcot;
        // (no s_L': the tangent sweep of the Hessian, whose seed is W_0^T . D_0)
        seed_dl = a.seedDl ? static_cast<const T*>(a.seedDl)[(size_t)seed_cot * a.Rmod + mrow0 + (m0 + x < M ? x : M - 1 - m0)] : T(1);
    }
    const T* const Wbu = lg_uniform(Wb);
    const T* const Abu = lg_uniform(Ab);
    const int ldb_bytes = __builtin_amdgcn_readfirstlane(a.ldb * (int)sizeof(T));
    const int lda_bytes = __builtin_amdgcn_readfirstlane((int)(a.lda * (long long)sizeof(T)));
    const int Ku = __builtin_amdgcn_readfirstlane(K);
    auto load_chunk = [&](int ch, ChunkRegs& cr) {
        const __amdgpu_buffer_rsrc_t rw = lg_rows_rsrc<T>(Wbu + (size_t)ch * BK * a.ldb, Ku - ch * BK, BK, ldb_bytes);
        const __amdgpu_buffer_rsrc_t ra = lg_rows_rsrc<T>(Abu + (size_t)ch * BK * a.lda, Ku - ch * BK, BK, lda_bytes);
#pragma unroll
        for (int u = 0; u < NW; ++u) cr.rw[u] = lg_buf_load(rw, offW[u], T(0));
#pragma unroll
        for (int u = 0; u < NA; ++u) cr.ra[u] = lg_buf_load(ra, offA[u], T(0));
        if constexpr (SEED) {
            const T* __restrict__ sw = static_cast<const T*>(a.seedW);
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                if constexpr (IL) {
                    // (48-column tiles: neither the chunk row nor the input of a thread's u-th element is fixed)
                    const int e = tid + 256 * u, k = ch * BK + e / BM, pin = (e % BM) >> 4;
                    cr.sw[u] = sw[(size_t)(k < K ? k : K - 1) * a.seed_nx + pin];
                } else {
                    const int k = ch * BK + krow0 + (256 / BM) * u;          // (beyond K: any row -- the operand it scales loaded as zero)
                    cr.sw[u] = sw[(size_t)(k < K ? k : K - 1) * a.seed_nx + seed_cot];
                }
            }
        }
    };
    auto store_chunk = [&](int buf, const ChunkRegs& cr) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int e = tid + 256 * u;
            Ws(buf, e / BN, e % BN) = cr.rw[u];
        }
        if constexpr (SEED) {
            // (the seed kernel's order of operations: (W_last s_L') D.  One switch over the activation per chunk, and the LDS
            // writes INSIDE its arms: merged behind it, the arms' results cost a register copy each)
            auto put = [&](auto dfun) {
#pragma unroll
                for (int u = 0; u < NA; ++u) {
                    const int e = tid + 256 * u;
                    As(buf, e / BM, e % BM) = (IL ? cr.sw[u] : cr.sw[u] * seed_dl) * dfun(cr.ra[u]);
                }
            };
            const T sp = (T)a.sactp;
            switch (a.sact) {
                case NEMPC_ACT_TANH: put([&](T v) { return lg_dval<T>(NEMPC_ACT_TANH, sp, 0, v); }); break;
                case NEMPC_ACT_RELU: put([&](T v) { return lg_dval<T>(NEMPC_ACT_RELU, sp, 0, v); }); break;
                case NEMPC_ACT_SIGMOID: put([&](T v) { return lg_dval<T>(NEMPC_ACT_SIGMOID, sp, 0, v); }); break;
                case NEMPC_ACT_ELU: put([&](T v) { return lg_dval<T>(NEMPC_ACT_ELU, sp, 0, v); }); break;
                case NEMPC_ACT_LEAKY_RELU: put([&](T v) { return lg_dval<T>(NEMPC_ACT_LEAKY_RELU, sp, 0, v); }); break;
                case NEMPC_ACT_SELU: put([&](T v) { return lg_dval<T>(NEMPC_ACT_SELU, sp, 0, v); }); break;
                default: put([&](T v) { return v; }); break;          // the derivative itself was stored
            }
        } else {
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int e = tid + 256 * u;
                As(buf, e / BM, e % BM) = cr.ra[u];
            }
        }
    };

    V4 acc[FT][RM];
#pragma unroll
    for (int fn = 0; fn < FT; ++fn)
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) acc[fn][rm] = V4{T(0), T(0), T(0), T(0)};

    const int nchunks = (K + BK - 1) / BK;
    const int fb = w * 16 * FT;                 // this wave's features inside the block
    auto mma_chunk = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            T af[FT], bf[RM];
#pragma unroll
            for (int fn = 0; fn < FT; ++fn) af[fn] = Ws(buf, 4 * ks + q, fb + 16 * fn + c);
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) bf[rm] = As(buf, 4 * ks + q, 16 * rm + c);
#pragma unroll
            for (int fn = 0; fn < FT; ++fn)
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) acc[fn][rm] = Ops::mma(af[fn], bf[rm], acc[fn][rm]);
        }
    };
    auto chunk_barrier = [&]() { __syncthreads(); };
    ChunkRegs c0, c1;           // c0: even chunks, c1: odd chunks
    load_chunk(0, c0);
    load_chunk(1, c1);
    LG_WGSTAMP(1);
    store_chunk(0, c0);
    __syncthreads();
    LG_WGSTAMP(2);
    // pairs of chunks (no exit in the middle of the body: with one, the accumulators were copied between two register sets
    // every pass and every copy waited out the matrix pipe); an odd last chunk follows the loop
    // pairs of chunks, no branch in the body: the compiler's wait-count bookkeeping stays exact -- a wait for the OLDER set only
    // (behind a branch it falls back to vmcnt(0) and the lead is gone); loads beyond the last chunk return zeros.  An odd
    // last chunk follows the loop.
    int ch = 0;
    LG_SEG_DECL();
    for (; ch + 1 < nchunks; ch += 2) {
        load_chunk(ch + 2, c0);
        __builtin_amdgcn_sched_barrier(0);      // (issued HERE: left alone, the scheduler sinks the loads under the matrix instructions and half the lead is gone)
        LG_SEG(0);
        mma_chunk(0);
        LG_SEG(1);
        store_chunk(1, c1);
        LG_SEG(2);
        chunk_barrier();
        LG_SEG(3);
        load_chunk(ch + 3, c1);
        __builtin_amdgcn_sched_barrier(0);
        LG_SEG(0);
        mma_chunk(1);
        LG_SEG(1);
        store_chunk(0, c0);
        LG_SEG(2);
        chunk_barrier();
        LG_SEG(3);
    }
    if (ch < nchunks) mma_chunk(0);
    LG_SEG_FLUSH();
    LG_WGSTAMP(3);
    // Where a 256 x 256 reverse product (B*H = 20480, fp64; 68 us at the matrix peak) spends its 148 us, by leaving parts out
    // (profiles/r04_layered_gemm_limiter.txt; the experiment's switches are gone from the source): no epilogue 110 us, no global loads 118, no LDS reads 143, no barrier 149, none of
    // loads / LDS / barrier 125.  The epilogue's dependent round trip for s'(z) at the end of every workgroup is the largest
    // piece; requesting those values under the last chunk's matrix instructions costs 32 more registers (occupancy 3 instead
    // of 4-5) and measured no better overall (355 vs 360 us for the whole 2 x 256 evaluation, 34.5 vs 32.4 ms at 4 x 512).

    // ---- epilogue: register r of lane (c, q) is feature row(q, r) of the 16 x 16 tile, row c
    T* __restrict__ C = static_cast<T*>(a.C);
    T* __restrict__ D = static_cast<T*>(a.D);
    const T* __restrict__ bias = static_cast<const T*>(a.bias);
    // reverse: the derivative's column of m.  A block of 64 columns never straddles two cotangent blocks (Rmod is a multiple
    // of 64), so one division per workgroup places it
    const long long mD0 = a.mode == LG_REVERSE ? mrow0 - m0 : 0;
    if constexpr (IL) {
        // ---- tangents P_l[n][p][row] in the accumulators (tile rm = input p, lane c = row 16 mb + c): the next layer's operand
        //      D_l . P_l leaves in the same interleaved layout (when there is a next layer), the layer's curvature term
        //      sum_n w_l[n][row] P[n][p] P[n][q] is summed over this lane's four features, the four feature groups of the wave
        //      (two exchanges across the lane groups), the four waves (LDS, wave order) -- and over the feature blocks by
        //      layered_hfinish_kernel, in block order: the summation order is fixed
        constexpr int NP = RM * (RM + 1) / 2;
        const long long row = (long long)mb * 16 + c;
        const bool rok = row < a.Rmod;
        const T* __restrict__ Wl = static_cast<const T*>(a.w0t);
        T part[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = T(0);
        T wv4[4], dv4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + fb + Ops::row(q, r);
            const size_t at = (n < N && rok) ? (size_t)n * a.ldd + row : (size_t)n0 * a.ldd + (size_t)mb * 16;
            wv4[r] = Wl[at];
            dv4[r] = C ? D[at] : T(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (C) lg_dval_n<T, 4>(a.dact, (T)a.dactp, 0, dv4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + fb + Ops::row(q, r);
            const bool ok = n < N && rok;
            const T wv = ok ? wv4[r] : T(0);
            if (C) {
                const T dv = ok ? dv4[r] : T(0);
#pragma unroll
                for (int rm = 0; rm < RM; ++rm)
                    if (n < N) C[(size_t)n * a.ldc + m0 + 16 * rm + c] = acc[0][rm][r] * dv;
            }
            int i = 0;
#pragma unroll
            for (int pp = 0; pp < RM; ++pp) {
                const T wp = wv * acc[0][pp][r];
#pragma unroll
                for (int qq = 0; qq <= pp; ++qq, ++i) part[i] = fma(wp, acc[0][qq][r], part[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            part[i] += __shfl_xor(part[i], 16);
            part[i] += __shfl_xor(part[i], 32);
        }
        __syncthreads();            // (the operand buffers are about to be reused: every wave is done reading them)
        if (q == 0) {
#pragma unroll
            for (int i = 0; i < NP; ++i) lds[(w * NP + i) * 16 + c] = part[i];
        }
        __syncthreads();
        if (tid < NP * 16) {
            const int i = tid >> 4, cc = tid & 15;
            const long long rw = (long long)mb * 16 + cc;
            if (rw < a.Rmod) {
                const T v = ((lds[(0 * NP + i) * 16 + cc] + lds[(1 * NP + i) * 16 + cc]) + lds[(2 * NP + i) * 16 + cc]) + lds[(3 * NP + i) * 16 + cc];
                static_cast<T*>(a.Jp)[(size_t)nb * a.jp_stride + (size_t)i * a.ldj + rw] = v;
            }
        }
        LG_WGSTAMP_EXIT();
        return;
    } else
    if constexpr (CONTRACT != LG_CONTRACT_NONE) {
        // E (G_0 = acc . D_0, or the layer's activations) stays in registers: register r of lane (c, q) holds feature
        // row(q, r), column c -- the four q of a register are a 4-deep k slab of features, i.e. the accumulator IS a B
        // operand (kernels_mfma_impl.h), and sum_n Wc[n][d] E[n][m] is four more matrix instructions per column tile with
        // Wc's fragment as A operand.  The four waves' sums (16 features each) meet in LDS in wave order; feature blocks
        // meet in layered_jreduce_kernel / layered_outfinish_kernel in block order: the summation order is fixed.
        constexpr int LDP = RM == 4 ? (LG_PAD == 0 ? 64 : (sizeof(T) == 8 ? 80 : 68))       // (f64: q's rows 128 B apart mod 256; f32: 64 B)
                                    : BM + 8;
        static_assert((size_t)4 * 16 * LDP <= (size_t)2 * S::TILE, "partial tiles fit the operand buffers");
        const T* __restrict__ W0 = static_cast<const T*>(a.w0t);
        T* __restrict__ Jp = static_cast<T*>(a.Jp) + (size_t)nb * a.jp_stride;
        T gd[RM][4];
        // the skinny matrix's fragment of the first output tile travels with the epilogue's other loads
        T wf0[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + fb + Ops::row(q, r);
            wf0[r] = (n < N && c < a.nin) ? W0[(size_t)n * a.ldw0 + c] : T(0);
        }
        if constexpr (CONTRACT == LG_CONTRACT_REVERSE) {
            // every load of the epilogue in flight before the first value is used (left to the compiler each of the 16 was a
            // round trip of its own -- 8 of a workgroup's 46 us, tools/diag_stamps_layered.py); out-of-range entries read
            // the block's first element and are masked below
            T dv[RM * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + fb + Ops::row(q, r);
                // (one address per feature row, the column tiles at constant offsets; a column beyond M stays inside the row's
                // storage -- Rmod is a multiple of the block -- and its product is masked below)
                const T* __restrict__ dp = D + (size_t)(n < N ? n : n0) * a.ldd + (m0 + mD0 + c);
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) dv[rm * 4 + r] = dp[16 * rm];
            }
            __builtin_amdgcn_sched_barrier(0);
            lg_dval_n<T, RM * 4>(a.dact, (T)a.dactp, a.duse, dv);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + fb + Ops::row(q, r);
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    const long long m = m0 + 16 * rm + c;
                    gd[rm][r] = (n < N && m < M) ? acc[0][rm][r] * dv[rm * 4 + r] : T(0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + fb + Ops::row(q, r);
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) {
                const long long m = m0 + 16 * rm + c;
                if constexpr (CONTRACT == LG_CONTRACT_REVERSE) {
                } else {
                    (void)n; (void)m;
                }
            }
        }
        if constexpr (CONTRACT == LG_CONTRACT_FORWARD) {
            T bn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + fb + Ops::row(q, r);
                bn[r] = bias[n < N ? n : n0];
            }
            __builtin_amdgcn_sched_barrier(0);
            T z[RM * 4], x[RM * 4], d1[RM * 4], e[RM * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) z[rm * 4 + r] = acc[0][rm][r] + bn[r];
            lg_act_all_n<T, RM * 4>(a.act, (T)a.actp, a.E != nullptr, z, x, d1, e);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + fb + Ops::row(q, r);
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    const long long m = m0 + 16 * rm + c;
                    const bool ok = n < N && m < M;
                    if (ok) {
                        D[(size_t)n * a.ldd + m] = a.store_a ? x[rm * 4 + r] : d1[rm * 4 + r];
                        if (a.E) static_cast<T*>(a.E)[(size_t)n * a.ldd + m] = e[rm * 4 + r];
                    }
                    gd[rm][r] = ok ? x[rm * 4 + r] : T(0);
                }
            }
        }
        LG_WGSTAMP(4);
        const int ndt = (a.nin + 15) / 16;
        for (int dt = 0; dt < ndt; ++dt) {
            V4 P[RM];
#pragma unroll
            for (int rm = 0; rm < RM; ++rm) P[rm] = V4{T(0), T(0), T(0), T(0)};
            T wf[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wf[r] = wf0[r];
            if (dt > 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + fb + Ops::row(q, r), d = 16 * dt + c;
                    wf[r] = (n < N && d < a.nin) ? W0[(size_t)n * a.ldw0 + d] : T(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) P[rm] = Ops::mma(wf[r], gd[rm][r], P[rm]);
            // unconditional: the partial tiles go into the operand buffers, which the slower waves of the workgroup may still
            // be reading -- an odd last chunk (`if (ch < nchunks) mma_chunk(0)`: every K <= 16 product and every width with an
            // odd number of 16-deep chunks) has no barrier behind it; dt > 0: the previous pass's reads of the partial tiles
            __syncthreads();
#pragma unroll
            for (int rm = 0; rm < RM; ++rm)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[(w * 16 + Ops::row(q, r)) * LDP + 16 * rm + c] = P[rm][r];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < RM; ++u) {
                const int e = tid + 256 * u, dd = e / BM, col = e % BM;
                const int d = 16 * dt + dd;
                const long long m = m0 + col;
                if (d < a.nin && m < M) {
                    const T v = ((lds[dd * LDP + col] + lds[(16 + dd) * LDP + col]) + lds[(32 + dd) * LDP + col]) + lds[(48 + dd) * LDP + col];
                    Jp[(size_t)d * a.ldj + m] = v;
                }
            }
        }
        LG_WGSTAMP_EXIT();
        return;
    }
    // (loads first, one switch over the activation per workgroup, then the stores: see lg_dval_n)
#pragma unroll
    for (int fn = 0; fn < FT; ++fn) {
        if (a.mode == LG_FORWARD) {
            T bn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + fb + 16 * fn + Ops::row(q, r);
                bn[r] = bias[n < N ? n : n0];
            }
            __builtin_amdgcn_sched_barrier(0);
            T z[RM * 4], x[RM * 4], d1[RM * 4], e[RM * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) z[rm * 4 + r] = acc[fn][rm][r] + bn[r];
            lg_act_all_n<T, RM * 4>(a.act, (T)a.actp, a.E != nullptr, z, x, d1, e);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + fb + 16 * fn + Ops::row(q, r);
                if (n >= N) continue;
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    const long long m = m0 + 16 * rm + c;
                    if (m >= M) continue;
                    C[(size_t)n * a.ldc + m] = x[rm * 4 + r];
                    if (D) D[(size_t)n * a.ldd + m] = d1[rm * 4 + r];          // (null: a layer whose derivatives follow from C itself)
                    if (a.E) static_cast<T*>(a.E)[(size_t)n * a.ldd + m] = e[rm * 4 + r];
                }
            }
        } else {
            T dv[RM * 4], ev[RM * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + fb + 16 * fn + Ops::row(q, r);
                const size_t at = (size_t)(n < N ? n : n0) * a.ldd + (m0 + mD0 + c);       // (see the contraction form above)
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    dv[rm * 4 + r] = C ? D[at + 16 * rm] : T(0);
                    ev[rm * 4 + r] = a.C2 ? static_cast<const T*>(a.E)[at + 16 * rm] : T(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (C) lg_dval_n<T, RM * 4>(a.dact, (T)a.dactp, 0, dv);
            if (a.C2) lg_dval_n<T, RM * 4>(a.dact, (T)a.dactp, 1, ev);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + fb + 16 * fn + Ops::row(q, r);
                if (n >= N) continue;
#pragma unroll
                for (int rm = 0; rm < RM; ++rm) {
                    const long long m = m0 + 16 * rm + c;
                    if (m >= M) continue;
                    const T v = acc[fn][rm][r];
                    if (C) C[(size_t)n * a.ldc + m] = v * dv[rm * 4 + r];
                    if (a.Craw) static_cast<T*>(a.Craw)[(size_t)n * a.ldc + m] = v;
                    if (a.C2) static_cast<T*>(a.C2)[(size_t)n * a.ldc + m] = v * ev[rm * 4 + r];
                }
            }
        }
    }
    LG_WGSTAMP_EXIT();
}

// ---- the small launches around the GEMMs (thread per row / per element; all feature-major, coalesced across rows) ----

// xi^T[d][r] = input d of row r0 + r (window inputs, then the extra inputs); RK4 stages add c DT k_{s-1} to the state part
template <typename T>
__global__ void layered_gather_kernel(RowGather gk, int nin, int ne, const T* __restrict__ extra, const T* __restrict__ Z,
                                      const T* __restrict__ X0, long long r0, int R, long long Rp, T* __restrict__ xi,
                                      const T* __restrict__ kprev, T cdt) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const long long gr = r0 + r;
    const int b = (int)(gr / gk.H), t = (int)(gr - (long long)b * gk.H);
    const T* z = Z + (size_t)b * gk.n;
    for (int d = 0; d < nin; ++d) {
        T v = gather_input<T>(gk, z, X0, b, t, d);
        if (kprev && d >= gk.xcur && d < gk.xcur + gk.nx) v = fma(cdt, kprev[(size_t)(d - gk.xcur) * Rp + r], v);
        xi[(size_t)d * Rp + r] = v;
    }
    for (int j = 0; j < ne; ++j) xi[(size_t)(nin + j) * Rp + r] = extra[(size_t)gr * ne + j];
}

// Gather + layer 0 as ONE vector-unit launch (networks with few inputs, K = nin + ne <= 8: the matrix pipe has nothing to do in a
// 3-deep product).  A block is 64 rows x 64 features: a lane holds its row's K inputs in registers, wave w walks features
// 16 w .. 16 w + 15 with the weight column and the bias read from LDS (one address for the whole wave: a broadcast), the activation
// leaves as a coalesced 512-byte store per feature.  Replaces the gather launch and a one-chunk launch of the GEMM kernel (5.4 +
// 21 us at 256 features x 20480 rows in fp64: that launch is bound by its 5,120 workgroups' fixed costs, and by 84 MB of stores
// when s' is stored next to the activation).  A first version of this idea (a wave per run of features, the weights as a chain of
// scalar loads per feature, s' stored) measured slower and was dropped (profiles/r05_layered_first.txt); this one stores the
// activation only where the layer's s' can be formed from it.
// (GENERIC = false: tanh / relu / sigmoid only -- the other activations' arithmetic in the same kernel sets its register count.
//  KP = 4 | 8: the input count padded, so that a feature's K + 1 LDS reads are issued together, without a branch per input.)
constexpr int LG_FIRST_KMAX = 8;
template <typename T, bool GENERIC, int KP>
__global__ __launch_bounds__(256, 4) void layered_first_kernel(RowGather gk, int nin, int ne, const T* __restrict__ extra, const T* __restrict__ Z,
                                                               const T* __restrict__ X0, long long r0, int R, long long Rp,
                                                               const T* __restrict__ W0, int N, const T* __restrict__ b0, int act, T actp,
                                                               T* __restrict__ A, T* __restrict__ D, T* __restrict__ E) {
    __shared__ T wl[KP + 1][64];            // rows 0 .. K-1: the weights (zero above), row KP: the bias
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n0 = blockIdx.y * 64, K = nin + ne;
    for (int i = tid; i < (KP + 1) * 64; i += 256) {
        const int k = i >> 6, x = i & 63, n = n0 + x;
        wl[k][x] = n < N ? (k < K ? W0[(size_t)k * N + n] : (k == KP ? b0[n] : T(0))) : T(0);
    }
    const int r = blockIdx.x * 64 + lane;
    const bool live = r < R;
    const long long gr = r0 + (live ? r : R - 1);
    const int b = (int)(gr / gk.H), t = (int)(gr - (long long)b * gk.H);
    const T* z = Z + (size_t)b * gk.n;
    // every input's address first, then the loads together
    const T* xp[KP];
#pragma unroll
    for (int d = 0; d < KP; ++d)
        xp[d] = d < nin ? gather_input_ptr<T>(gk, z, X0, b, t, d) : (d < K ? extra + ((size_t)gr * ne + (d - nin)) : Z);
    __builtin_amdgcn_sched_barrier(0);
    T x[KP];
#pragma unroll
    for (int d = 0; d < KP; ++d) x[d] = *xp[d];
#pragma unroll
    for (int d = 0; d < KP; ++d)
        if (d >= K) x[d] = T(0);
    __syncthreads();
    const bool want_e = E != nullptr;
    auto run = [&](auto actf) {
        // (two features in flight per wave: fully unrolled the sixteen activations took 152 registers -- three waves per SIMD, the
        // launch in two rounds)
#pragma clang loop unroll_count(2)
        for (int f = 0; f < 16; ++f) {
            const int col = 16 * w + f, n = n0 + col;
            T wv[KP + 1];
#pragma unroll
            for (int d = 0; d <= KP; ++d) wv[d] = wl[d][col];
            T zz = T(0);            // (the product first, the bias last: the order of the GEMM kernel's epilogue)
#pragma unroll
            for (int d = 0; d < KP; ++d) zz = fma(x[d], wv[d], zz);
            T a, d1, e;
            actf(zz + wv[KP], a, d1, e);
            if (live && n < N) {
                A[(size_t)n * Rp + r] = a;
                if (D) D[(size_t)n * Rp + r] = d1;
                if (E) E[(size_t)n * Rp + r] = e;
            }
        }
    };
#define LG_FIRST_CASE(CODE) \
    case CODE: run([&](T zv, T& a, T& d1, T& e) { lg_act_all<T>(CODE, zv, actp, want_e, a, d1, e); }); break;
    if constexpr (GENERIC) {
        run([&](T zv, T& a, T& d1, T& e) { lg_act_all<T>(act, zv, actp, want_e, a, d1, e); });
    } else {
        switch (act) {
            LG_FIRST_CASE(NEMPC_ACT_TANH)
            LG_FIRST_CASE(NEMPC_ACT_RELU)
            default: run([&](T zv, T& a, T& d1, T& e) { lg_act_all<T>(NEMPC_ACT_SIGMOID, zv, actp, want_e, a, d1, e); }); break;
        }
    }
#undef LG_FIRST_CASE
}
template <typename T, typename... Args>
void launch_first(int act, int K, dim3 grid, hipStream_t s, Args... args) {
    const bool lean = act == NEMPC_ACT_TANH || act == NEMPC_ACT_RELU || act == NEMPC_ACT_SIGMOID;
    if (K <= 4) {
        if (lean) hipLaunchKernelGGL((layered_first_kernel<T, false, 4>), grid, dim3(256), 0, s, args...);
        else hipLaunchKernelGGL((layered_first_kernel<T, true, 4>), grid, dim3(256), 0, s, args...);
    } else {
        if (lean) hipLaunchKernelGGL((layered_first_kernel<T, false, 8>), grid, dim3(256), 0, s, args...);
        else hipLaunchKernelGGL((layered_first_kernel<T, true, 8>), grid, dim3(256), 0, s, args...);
    }
}
// NEMPC_LAYERED_FIRST=0: gather launch + one-chunk GEMM launch for layer 0, as before (A/B; tested against the default)
bool lg_first_on() {
    static const bool on = [] { const char* e = getenv("NEMPC_LAYERED_FIRST"); return !(e && atoi(e) == 0); }();
    return on;
}

// N <= NMAX outputs per column on the vector unit: out^T[n][m] = epi(sum_k A^T[k][m] Bw[k][n]).  mode 0: the network's
// output layer (bias, activation; f and s'(z_L) stored), mode 2: plain (the last reverse step onto the inputs).
// A block is 64 columns x 4 waves; wave w sums k = w, w + 4, ... with sixteen loads in flight per lane, the four partial
// sums meet in LDS.  (A thread per column walking all of K alone was a chain of K dependent-latency loads: 141 us for
// K = 256; eight loads in flight and 32 accumulators for 3 outputs: 60 us, 1.4 TB/s; this form: ~30 us.)  NMAX = 4 | 16 | 32
// keeps the accumulator count -- and with it the occupancy -- at what the layer needs.
template <typename T, int NMAX>
__global__ __launch_bounds__(256) void layered_skinny_kernel(const T* __restrict__ A, long long lda, const T* __restrict__ Bw, int ldb,
                                                             int K, int N, long long M, T* __restrict__ out, long long ldo,
                                                             const T* __restrict__ bias, int mode, int act, T* __restrict__ dout,
                                                             T actp, T* __restrict__ eout) {
    __shared__ T red[3][NMAX][64];
    constexpr int UN = 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long m = (long long)blockIdx.x * 64 + lane;
    const bool live = m < M;
    const long long mc = live ? m : M - 1;          // (clamped: the loads stay in range, the result is not stored)
    T acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = T(0);
    for (int k0 = w; k0 < K; k0 += 4 * UN) {
        T x[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = k0 + 4 * u;
            x[u] = k < K ? A[(size_t)k * lda + mc] : T(0);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = k0 + 4 * u;
            if (k < K) {
                const T* wrow = Bw + (size_t)k * ldb;
#pragma unroll
                for (int n = 0; n < NMAX; ++n)
                    if (n < N) acc[n] = fma(x[u], wrow[n], acc[n]);
            }
        }
    }
    if (w > 0) {
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
            if (n < N) red[w - 1][n][lane] = acc[n];
    }
    __syncthreads();
    if (w == 0 && live) {
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
            if (n < N) {
                const T v = ((acc[n] + red[0][n][lane]) + red[1][n][lane]) + red[2][n][lane];
                if (mode == 0) {
                    T x, d1, e;
                    lg_act_all<T>(act, v + bias[n], actp, eout != nullptr, x, d1, e);
                    out[(size_t)n * ldo + m] = x;
                    dout[(size_t)n * ldo + m] = d1;
                    if (eout) eout[(size_t)n * ldo + m] = e;
                } else {
                    out[(size_t)n * ldo + m] = v;
                }
            }
    }
}

template <typename T>
int skinny(hipStream_t s, const T* A, long long lda, const T* Bw, int ldb, int K, int N, long long M, T* out, long long ldo,
           const T* bias, int mode, int act, T* dout, T actp, T* eout = nullptr) {
    const dim3 grid((unsigned)((M + 63) / 64)), block(256);
    if (N <= 4) hipLaunchKernelGGL((layered_skinny_kernel<T, 4>), grid, block, 0, s, A, lda, Bw, ldb, K, N, M, out, ldo, bias, mode, act, dout, actp, eout);
    else if (N <= 16) hipLaunchKernelGGL((layered_skinny_kernel<T, 16>), grid, block, 0, s, A, lda, Bw, ldb, K, N, M, out, ldo, bias, mode, act, dout, actp, eout);
    else hipLaunchKernelGGL((layered_skinny_kernel<T, 32>), grid, block, 0, s, A, lda, Bw, ldb, K, N, M, out, ldo, bias, mode, act, dout, actp, eout);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

// seed of the reverse sweep: G^T[j][k Rp + r] = W_last[j][k] s_L'(z_L)[k][r] D_{L-2}^T[j][r]  (all nx cotangents side by side)
template <typename T>
__global__ void layered_seed_kernel(const T* __restrict__ Wlast, int wdt, int nx, const T* __restrict__ dL, const T* __restrict__ Dh,
                                    int R, long long Rp, T* __restrict__ G, int dact, T dactp) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (r >= R || j >= wdt) return;
    const T d = lg_dval<T>(dact, dactp, 0, Dh[(size_t)j * Rp + r]);
    for (int k = 0; k < nx; ++k) G[(size_t)j * (nx * Rp) + (size_t)k * Rp + r] = Wlast[(size_t)j * nx + k] * dL[(size_t)k * Rp + r] * d;
}

// RK4 stage bookkeeping per row (rk4.py:69-80,147-159): k_s = f, dk_s = J_s + c DT J_s[:, :nx] dk_{s-1}; weighted sums
template <typename T>
__global__ void layered_rk4_kernel(int stage, int nx, int nin, T cdt, T wgt, const T* __restrict__ f, const T* __restrict__ J,
                                   int R, long long Rp, T* __restrict__ kprev, T* __restrict__ acck, T* __restrict__ dk,
                                   T* __restrict__ dkn, T* __restrict__ accdk) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    // J^T[d][k Rp + r] = dPhi_k / dxi_d of row r
    for (int i = 0; i < nx; ++i)
        for (int d = 0; d < nin; ++d) {
            T v = J[(size_t)d * (nx * Rp) + (size_t)i * Rp + r];
            if (stage > 0) {
                T s = T(0);
                for (int e = 0; e < nx; ++e)
                    s = fma(J[(size_t)e * (nx * Rp) + (size_t)i * Rp + r], dk[(size_t)(e * nin + d) * Rp + r], s);
                v = v + cdt * s;
            }
            dkn[(size_t)(i * nin + d) * Rp + r] = v;
        }
    for (int i = 0; i < nx; ++i) {
        const T kv = f[(size_t)i * Rp + r];
        kprev[(size_t)i * Rp + r] = kv;
        acck[(size_t)i * Rp + r] = stage == 0 ? kv : fma(wgt, kv, acck[(size_t)i * Rp + r]);
        for (int d = 0; d < nin; ++d) {
            const T v = dkn[(size_t)(i * nin + d) * Rp + r];
            dk[(size_t)(i * nin + d) * Rp + r] = v;
            accdk[(size_t)(i * nin + d) * Rp + r] = stage == 0 ? v : fma(wgt, v, accdk[(size_t)(i * nin + d) * Rp + r]);
        }
    }
}

// RK4 Hessian pipeline (kernels_rk4hess.hip): the record of (row, stage) = [xi_s | J_s (nx, nin) | dk_{s-1} (nx, nin)], written
// between the stage's reverse sweep and its bookkeeping (dk still holds dk_{s-1})
template <typename T>
__global__ void layered_stage_record_kernel(int stage, int nx, int nin, long long r0, int R, long long Rp, const T* __restrict__ xi,
                                            const T* __restrict__ J, const T* __restrict__ dk, T* __restrict__ out, int stride) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    T* rec = out + ((size_t)(r0 + r) * 4 + stage) * stride;
    for (int d = 0; d < nin; ++d) rec[d] = xi[(size_t)d * Rp + r];
    for (int i = 0; i < nx; ++i)
        for (int d = 0; d < nin; ++d) {
            rec[nin + i * nin + d] = J[(size_t)d * (nx * Rp) + (size_t)i * Rp + r];
            rec[nin + nx * nin + i * nin + d] = stage > 0 ? dk[(size_t)(i * nin + d) * Rp + r] : T(0);
        }
}

// sum of `nb` partial sums `stride` elements apart, in block order, up to four loads in flight
template <typename T>
__device__ __forceinline__ T lg_blocksum(const T* __restrict__ p, int nb, long long stride) {
    T v = p[0];
    int bb = 1;
    for (; bb + 3 < nb; bb += 4) {
        const T a0 = p[(size_t)bb * stride], a1 = p[(size_t)(bb + 1) * stride], a2 = p[(size_t)(bb + 2) * stride],
                a3 = p[(size_t)(bb + 3) * stride];
        v += a0; v += a1; v += a2; v += a3;
    }
    if (bb + 2 < nb) {
        const T a0 = p[(size_t)bb * stride], a1 = p[(size_t)(bb + 1) * stride], a2 = p[(size_t)(bb + 2) * stride];
        v += a0; v += a1; v += a2;
    } else if (bb + 1 < nb) {
        const T a0 = p[(size_t)bb * stride], a1 = p[(size_t)(bb + 1) * stride];
        v += a0; v += a1;
    } else if (bb < nb) {
        v += p[(size_t)bb * stride];
    }
    return v;
}

// defects, box rows and compact tiles of the chunk's rows (same formulas as rows_valu_kernel)
template <typename T>
__global__ void layered_finish_kernel(RowGather gk, int kind, T DT, int nin, const T* __restrict__ Z, const T* __restrict__ X0,
                                      long long r0, int R, long long Rp, const T* __restrict__ f, const T* __restrict__ J,
                                      const T* __restrict__ acck, const T* __restrict__ accdk, T* __restrict__ g, int m, int box,
                                      T* __restrict__ tiles, int jblk, long long jstride, int fblk, long long fstride,
                                      const T* __restrict__ fbias) {
    // (fblk > 0: f is still the feature blocks' partial sums of a LINEAR output layer -- added here in block order, bias last,
    //  as layered_outfinish_kernel would have: its launch is gone for such networks under Discret / Unity)
    // A thread per (row, state i) -- blockIdx.y = i -- with the loads of a partial-sum run issued four at a time: as a thread per
    // row walking i, d and the blocks in nested run-time loops this launch was a chain of ~ 30 dependent loads on 80 workgroups
    // (10 us at B*H = 20480 with four feature blocks).
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int i = blockIdx.y;
    const long long gr = r0 + r;
    const int nx = gk.nx, H = gk.H;
    const int b = (int)(gr / H), t = (int)(gr - (long long)b * H);
    const T* z = Z + (size_t)b * gk.n;
    T* gout = g + (size_t)b * m + (size_t)t * nx;
    T* tile = tiles + (size_t)gr * nx * nin;
    const T s6 = DT / T(6);
    {
        const T xp = (t == 0) ? X0[(size_t)b * nx + i] : z[(t - 1) * nx + i];
        const T xt = z[t * nx + i];
        T phi;
        if (kind == NEMPC_RK4) phi = xp + s6 * acck[(size_t)i * Rp + r];
        else {
            T fv = fblk ? lg_blocksum<T>(f + (size_t)i * Rp + r, fblk, fstride) + fbias[i] : f[(size_t)i * Rp + r];
            phi = (kind == NEMPC_DISCRET ? xp : T(0)) + fv;
        }
        gout[i] = phi - xt;
        if (box) gout[(size_t)H * nx + i] = xt;
        for (int d = 0; d < nin; ++d) {
            T v;
            if (kind == NEMPC_RK4) v = s6 * accdk[(size_t)(i * nin + d) * Rp + r] + (d == i ? T(1) : T(0));
            else {
                // (jblk > 1: J is still the feature blocks' partial sums -- added here in block order, as layered_jreduce_kernel
                //  would have: the reduction launch of its own is gone for Discret / Unity)
                const size_t ji = (size_t)d * (nx * Rp) + (size_t)i * Rp + r;
                v = lg_blocksum<T>(J + ji, jblk, jstride) + ((kind == NEMPC_DISCRET && d == gk.xcur + i) ? T(1) : T(0));
            }
            tile[i * nin + d] = v;
        }
    }
}

struct LayeredWs {      // element offsets into the chunk workspace (times nothing: already multiplied by Rp)
    size_t xi, x0, x1, d[NEMPC_MAX_LAYERS], f, dl, g0, g1, j, kprev, acck, dk, dkn, accdk, total;
};

LayeredWs layered_offsets(const Handle& h, size_t Rp) {
    LayeredWs o{};
    const int nx = h.cfg.nx, nin = h.nin;
    size_t p = 0;
    o.xi = p; p += (size_t)(nin + h.ne) * Rp;
    o.x0 = p; p += (size_t)h.maxw * Rp;
    o.x1 = p; p += (size_t)h.maxw * Rp;
    for (int l = 0; l < h.nl - 1; ++l) { o.d[l] = p; p += (size_t)h.dout[l] * Rp; }
    o.f = p; p += (size_t)nx * Rp;
    o.dl = p; p += (size_t)nx * Rp;
    o.g0 = p; p += (size_t)h.maxw * nx * Rp;
    o.g1 = p; p += (size_t)h.maxw * nx * Rp;
    o.j = p; p += (size_t)nin * nx * Rp;
    if (h.cfg.integrator == NEMPC_RK4) {
        o.kprev = p; p += (size_t)nx * Rp;
        o.acck = p; p += (size_t)nx * Rp;
        o.dk = p; p += (size_t)nx * nin * Rp;
        o.dkn = p; p += (size_t)nx * nin * Rp;
        o.accdk = p; p += (size_t)nx * nin * Rp;
    }
    o.total = p;
    return o;
}

// network output from the feature blocks' partial sums (LG_CONTRACT_FORWARD), in block order: f = s_L(sum + b), s_L'(z_L)
template <typename T>
__global__ void layered_outfinish_kernel(const T* __restrict__ P, int nblk, long long stride, int nx, int R, long long Rp,
                                         const T* __restrict__ bias, int act, T actp, T* __restrict__ f, T* __restrict__ dl,
                                         T* __restrict__ el) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    for (int o = 0; o < nx; ++o) {
        T v = P[(size_t)o * Rp + r];
        for (int b = 1; b < nblk; ++b) v += P[(size_t)b * stride + (size_t)o * Rp + r];
        // (from the output for the monotone activations, from the pre-activation for swish / gelu ...: lg_act_all)
        T x, d1, e;
        lg_act_all<T>(act, v + bias[o], actp, el != nullptr, x, d1, e);
        f[(size_t)o * Rp + r] = x;
        dl[(size_t)o * Rp + r] = d1;
        if (el) el[(size_t)o * Rp + r] = e;           // s_L''(z_L): the Hessian sweeps' output-layer curvature
    }
}

// J^T = sum over the feature blocks' partial sums, in block order (LG_CONTRACT_REVERSE with more than one block)
template <typename T>
__global__ void layered_jreduce_kernel(const T* __restrict__ Jp, int nblk, long long stride, T* __restrict__ J, long long count) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        T v = Jp[i];
        for (int b = 1; b < nblk; ++b) v += Jp[(size_t)b * stride + i];
        J[i] = v;
    }
}

template <typename T, int FT, bool SEED = false, int CONTRACT = LG_CONTRACT_NONE, int RM = 4>
int gemm_ft(hipStream_t s, const GemmArgs& a) {
    using S = LgShape<FT, RM>;
    const size_t bytes = (size_t)2 * S::TILE * sizeof(T);
    auto kern = layered_gemm_kernel<T, FT, SEED, CONTRACT, RM>;
    NEMPC_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), bytes));
    GemmArgs b = a;
    b.dbg = nullptr;
#ifdef NEMPC_STAMPS
    {
        static const int sel = [] { const char* e = getenv("NEMPC_LG_STAMP"); return e ? atoi(e) : 11; }();
        if (sel == 10 * (int)SEED + CONTRACT) b.dbg = g_lg_dbg;
    }
#endif
    b.nblk = (a.N + S::BN - 1) / S::BN;
    b.nblk_magic = b.nblk == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)b.nblk) + 1u;
    const long long mblk = ((long long)a.M + S::BM - 1) / S::BM;
    long long groups = (mblk + 7) / 8;
    b.ncot = 1; b.rbc = 0; b.ncot_magic = 0;
    constexpr bool il = CONTRACT == LG_CONTRACT_HPAIR;
    static const bool cot_order = [] { const char* e = getenv("NEMPC_LG_COT_ORDER"); return !(e && atoi(e) == 0); }();
    if (cot_order && !il && (SEED || a.mode == LG_REVERSE) && a.Rmod > 0 && a.Rmod % S::BM == 0 && a.M % a.Rmod == 0 && a.M / a.Rmod > 1) {
        b.ncot = (int)(a.M / a.Rmod);
        b.rbc = (int)(a.Rmod / S::BM);
        b.ncot_magic = (unsigned)(0x100000000ull / (unsigned)b.ncot) + 1u;
        groups = (long long)((b.rbc + 7) / 8) * b.ncot;
    }
    const dim3 grid((unsigned)(8 * b.nblk * groups));      // (row blocks padded to the 8 XCDs; the surplus exits at once)
    hipLaunchKernelGGL(kern, grid, dim3(256), bytes, s, b);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

// fp64 FORWARD products (activation + two or three stores per element in the epilogue) whose 64-row tiling gives the launch
// less than about two rounds of workgroups run on 32-row blocks: twice the workgroups, five or six waves per SIMD instead of
// four, so that one workgroup's epilogue runs under the others' matrix instructions.  Measured at 2 x 256, B*H = 20480
// (tools/lg_rm_ab.sh): plain forward products 74 -> 53 us, the last hidden layer with the output contraction 82 -> 76 us.
// Not the reverse products (a load and a store per element: 5 % slower that way) and not fp32 (six to eight waves per SIMD
// already; the contraction form measured 9 % slower).  NEMPC_LG_RM = 2 | 4 forces one form (A/B, tests).
template <typename T>
bool lg_rows32(int num_cus, long long M, int N) {
    static const int rm_env = [] { const char* e = getenv("NEMPC_LG_RM"); return e ? atoi(e) : 0; }();
    if (rm_env == 2) return true;
    if (rm_env == 4 || sizeof(T) != 8) return false;
    const long long tiles64 = ((M + 63) / 64) * (long long)((N + 63) / 64);
    return tiles64 <= (long long)2 * num_cus * 4;
}
template <typename T, int CONTRACT>
int gemm_forward(int num_cus, hipStream_t s, const GemmArgs& a) {
    // (80-row blocks -- B*H = 20480 rows in exactly one round of workgroups -- measured slower, round 5: 2 x 256 fp32 149 -> 156 us,
    //  fp64 273 -> 290 with 80 - 204 B of scratch: one round means every prologue and every epilogue of the launch is exposed)
    return lg_rows32<T>(num_cus, a.M, a.N) ? gemm_ft<T, 1, false, CONTRACT, 2>(s, a) : gemm_ft<T, 1, false, CONTRACT, 4>(s, a);
}

template <typename T>
int gemm(int num_cus, hipStream_t s, int mode, int act, const T* A, long long lda, const T* Bw, int ldb, T* C, long long ldc, T* D,
         long long ldd, const T* bias, long long M, int N, int K, long long Rmod, double actp = 0.0, int dact = 0, double dactp = 0.0) {
    GemmArgs a{};
    a.actp = actp;
    a.dact = dact; a.dactp = dactp;
    a.A = A; a.Bw = Bw; a.C = C; a.D = D; a.bias = bias;
    a.lda = lda; a.ldc = ldc; a.ldd = ldd; a.ldb = ldb;
    a.M = (int)M; a.N = N; a.K = K; a.mode = mode; a.act = act; a.Rmod = Rmod;
    // 64-feature blocks (FT = 1) measured fastest at every width (LgShape above); wider blocks are not instantiated
    if (mode == LG_FORWARD) return gemm_forward<T, LG_CONTRACT_NONE>(num_cus, s, a);
    return gemm_ft<T, 1>(s, a);
}

// The fused forms of the reverse sweep (64-feature blocks only): NEMPC_LAYERED_FUSE=0 walks it with the seed kernel, plain
// products and the skinny last step instead (A/B knob)
bool layered_fuse() {
    static const bool on = [] {
        const char* e = getenv("NEMPC_LAYERED_FUSE");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

// NEMPC_LAYERED_DFA: 0 every layer stores s' (and s'') next to its activation, as in round 4; 1 (default) networks up to width 384
// form them from the activation in the Hessian sweeps and, with three or more hidden layers, in the rows path; 2 everywhere
// (A/B; each tested against the default)
int lg_dfa_on() {
    static const int on = [] {
        const char* e = getenv("NEMPC_LAYERED_DFA");
        return e ? atoi(e) : 1;
    }();
    return on;
}

// NEMPC_LAYERED_HFOLD=0: the Hessian's tangents are written and contracted by layered_hcontract_kernel, as in round 4 (A/B,
// tested against the default)
bool lg_hfold_on() {
    static const bool on = [] {
        const char* e = getenv("NEMPC_LAYERED_HFOLD");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

// tangent product with the layer's curvature term in its epilogue (interleaved columns, tile = 16 rows x nin inputs)
template <typename T>
int gemm_hpair(hipStream_t s, const GemmArgs& a, bool seed, int nin) {
    if (nin == 2) return seed ? gemm_ft<T, 1, true, LG_CONTRACT_HPAIR, 2>(s, a) : gemm_ft<T, 1, false, LG_CONTRACT_HPAIR, 2>(s, a);
    if (nin == 3) return seed ? gemm_ft<T, 1, true, LG_CONTRACT_HPAIR, 3>(s, a) : gemm_ft<T, 1, false, LG_CONTRACT_HPAIR, 3>(s, a);
    return seed ? gemm_ft<T, 1, true, LG_CONTRACT_HPAIR, 4>(s, a) : gemm_ft<T, 1, false, LG_CONTRACT_HPAIR, 4>(s, a);
}

template <typename T>
int gemm_reverse_fused(hipStream_t s, const GemmArgs& a, bool seed, bool last) {
    // (A/B, NEMPC_LG_RM_REV=2: the seed + contraction product on 32-row blocks measured 7 - 11 % slower -- 2 x 256, B*H = 20480:
    // 257 -> 274 us per evaluation in fp64, 136 -> 151 in fp32 -- so the reverse products keep the 64-row block)
    static const int rm_env = [] { const char* e = getenv("NEMPC_LG_RM_REV"); return e ? atoi(e) : 0; }();
    if (rm_env == 2 && seed && last) return gemm_ft<T, 1, true, LG_CONTRACT_REVERSE, 2>(s, a);
    if (seed && last) return gemm_ft<T, 1, true, LG_CONTRACT_REVERSE>(s, a);
    if (seed) return gemm_ft<T, 1, true, LG_CONTRACT_NONE>(s, a);
    if (last) return gemm_ft<T, 1, false, LG_CONTRACT_REVERSE>(s, a);
    return gemm_ft<T, 1>(s, a);
}

template <typename T>
int run_layered(Handle& h, int B, const void* Zv, const void* X0v, void* gv, void* tilesv, hipStream_t s, void* stage_out = nullptr,
                int stage_stride = 0) {
    const T* Z = static_cast<const T*>(Zv);
    const T* X0 = static_cast<const T*>(X0v);
    T* g = static_cast<T*>(gv);
    T* tiles = static_cast<T*>(tilesv);
    const int nx = h.cfg.nx, nin = h.nin, ne = h.ne, nl = h.nl, H = h.cfg.H;
    const long long rows = (long long)B * H;
    const long long Rc = h.layered_chunk_rows;           // multiple of 64
    T* ws = static_cast<T*>(h.d_layered_ws);
    const RowGather gk = h.gather();
    const bool rk4 = h.cfg.integrator == NEMPC_RK4;
    const int nstages = rk4 ? 4 : 1;
    const T DT = (T)h.cfg.DT;
    int rc;
    for (long long r0 = 0; r0 < rows; r0 += Rc) {
        const int R = (int)(rows - r0 < Rc ? rows - r0 : Rc);
        // row stride of every matrix of this chunk: its row count rounded to whole GEMM blocks (a short batch on a handle
        // sized for a large one does not pay for the columns it does not have)
        const long long Rm = ((long long)R + LG_BM - 1) / LG_BM * LG_BM;
        const long long Rp = Rm;
        const LayeredWs o = layered_offsets(h, (size_t)Rp);
        const dim3 rb(256), rg((unsigned)((R + 255) / 256));
        int jblk = 1;                       // the Jacobian as the finish kernel finds it: jblk feature blocks of partial sums
        const T* jsrc = ws + o.j;
        long long jstride = 0;
        const T* fsrc = nullptr;            // the network output likewise (lin_skip: partial sums of a linear output layer)
        int fblk = 0;
        long long fstride = 0;
        for (int st = 0; st < nstages; ++st) {
            const T cdt = st == 0 ? T(0) : (st == 3 ? DT : T(0.5) * DT);
            const bool fuse_out = layered_fuse();
            static const bool outskip = [] { const char* e = getenv("NEMPC_LAYERED_OUTSKIP"); return !(e && atoi(e) == 0); }();      // (A/B, tested)
            const bool lin_skip = outskip && fuse_out && nl >= 3 && !rk4 && h.act[nl - 1] == NEMPC_ACT_LINEAR;
            // gather + layer 0 in one vector-unit launch (layered_first_kernel): few inputs, two hidden layers or more (layer 0 is
            // not the layer the output contraction leaves from), Discret / Unity (the RK4 stages' inputs carry c DT k_{s-1} and
            // their records want xi)
            const bool first = lg_first_on() && !rk4 && nl - 1 >= 2 && nin + ne <= LG_FIRST_KMAX;
            const bool first_dfa = first && lg_dfa_on() != 0 && lg_d_from_a(h.act[0]);
            if (!first) {
                hipLaunchKernelGGL(layered_gather_kernel<T>, rg, rb, 0, s, gk, nin, ne, static_cast<const T*>(h.d_extra), Z, X0, r0, R, Rp,
                                   ws + o.xi, st > 0 ? ws + o.kprev : nullptr, cdt);
                NEMPC_HIP(hipGetLastError());
            }
            // ---- forward: hidden layers 0 .. nl-2 (GEMM), output layer nl-1 (skinny)
            const T* in = ws + o.xi;
            // layers that store their activation only (their s' is formed from it where it is needed: lg_d_from_a); the
            // activation then lives in the layer's own slot (o.d[l]) instead of the two alternating ones
            // (measured, tools/layered_bench.py with NEMPC_LAYERED_DFA = 0 | 1 on every launch: the ROWS path gains nothing --
            // 2 x 256 fp64 267 -> 269 us, 3 x 256 476 -> 467, 4 x 512 RK4 25.3 -> 25.4 ms: its layer-0 product is latency-bound,
            // not store-bound, and the last hidden layer stores one matrix either way -- so it keeps s' stored; the Hessian
            // sweeps, which store three matrices per layer, take it: NEMPC_LAYERED_DFA=2 forces it here too, for the A/B)
            // Round 5, measured again with primed clocks (tools/layered_ab.py, us per evaluation in fp64: stored s' / activations only
            // for all but the last hidden layer / for every layer): 2 x 256 213 / 215 / 216, 3 x 256 387 / 373 / 364.  The products
            // between hidden layers are the ones that gain (one matrix stored instead of two); layer 0 is latency-bound either
            // way.  So: 1 (default) every layer of a network with three or more hidden layers, up to width 384; 2 every layer of
            // every network; 0 none.
            auto dfa = [&](int l) {
                if (l == 0 && first_dfa) return true;       // (layered_first_kernel is store-bound: one matrix instead of two)
                return lg_d_from_a(h.act[l]) && (lg_dfa_on() == 2 || (lg_dfa_on() == 1 && h.maxw <= 384 && nl - 1 >= 3));
            };
            for (int l = 0; l < nl - 1; ++l) {
                T* out = dfa(l) ? ws + o.d[l] : ws + ((l & 1) ? o.x1 : o.x0);
                if (l == 0 && first) {
                    launch_first<T>(h.act[0], nin + ne, dim3((unsigned)((R + 63) / 64), (unsigned)((h.dout[0] + 63) / 64)), s,
                                    gk, nin, ne, static_cast<const T*>(h.d_extra), Z, X0, r0, R, Rp, static_cast<const T*>(h.d_W[0]),
                                    h.dout[0], static_cast<const T*>(h.d_b[0]), h.act[0], (T)h.actp[0], out,
                                    dfa(0) ? static_cast<T*>(nullptr) : ws + o.d[0], static_cast<T*>(nullptr));
                    NEMPC_HIP(hipGetLastError());
                    in = out;
                    continue;
                }
                if (l == nl - 2 && fuse_out) {
                    // the last hidden layer: its activations go straight into the output layer's contraction (only s' is
                    // stored); partial sums per feature block in the cotangent buffer, which the reverse sweep fills later
                    GemmArgs a{};
                    a.mode = LG_FORWARD; a.act = h.act[l]; a.actp = h.actp[l];
                    a.A = in; a.lda = Rp; a.Bw = h.d_W[l]; a.ldb = h.dout[l];
                    a.D = ws + o.d[l]; a.ldd = Rp; a.bias = h.d_b[l];
                    a.store_a = dfa(l) ? 1 : 0;
                    a.M = R; a.N = h.dout[l]; a.K = h.din[l];
                    a.w0t = h.d_W[nl - 1]; a.ldw0 = nx; a.nin = nx;
                    a.Jp = ws + o.g0; a.ldj = Rp; a.jp_stride = (long long)nx * Rp;
                    if (lin_skip) {
                        // a LINEAR output layer, Discret / Unity: s_L' = 1 (the seed loader takes a null pointer for that) and
                        // f = sum of the partial sums + bias is formed by layered_finish_kernel -- no output step.  The partial
                        // sums wait in the activation buffer this product does not read (the cotangent buffers are overwritten by
                        // the reverse sweep before the finish kernel runs).
                        T* fp = (in == ws + o.x1) ? ws + o.x0 : ws + o.x1;
                        a.Jp = fp;
                        fsrc = fp; fblk = (h.dout[l] + 63) / 64; fstride = a.jp_stride;
                    }
                    if ((rc = gemm_forward<T, LG_CONTRACT_FORWARD>(h.num_cus, s, a))) return rc;
                    if (!lin_skip) {
                        hipLaunchKernelGGL(layered_outfinish_kernel<T>, rg, rb, 0, s, ws + o.g0, (h.dout[l] + 63) / 64, a.jp_stride, nx, R, Rp,
                                           static_cast<const T*>(h.d_b[nl - 1]), h.act[nl - 1], (T)h.actp[nl - 1], ws + o.f, ws + o.dl,
                                           static_cast<T*>(nullptr));
                        NEMPC_HIP(hipGetLastError());
                    }
                    break;
                }
                if ((rc = gemm<T>(h.num_cus, s, LG_FORWARD, h.act[l], in, Rp, static_cast<const T*>(h.d_W[l]), h.dout[l], out, Rp,
                                  dfa(l) ? static_cast<T*>(nullptr) : ws + o.d[l], Rp, static_cast<const T*>(h.d_b[l]), R, h.dout[l],
                                  h.din[l], 0, h.actp[l])))
                    return rc;
                in = out;
            }
            if (!fuse_out &&
                (rc = skinny<T>(s, in, Rp, static_cast<const T*>(h.d_W[nl - 1]), nx, h.din[nl - 1], nx, (long long)R, ws + o.f, Rp,
                                static_cast<const T*>(h.d_b[nl - 1]), 0, h.act[nl - 1], ws + o.dl, (T)h.actp[nl - 1])))
                return rc;
            // ---- reverse, all nx cotangents side by side: column k Rp + r is (cotangent k, row r)
            const long long ldg = (long long)nx * Rp;
            if (nl >= 3 && layered_fuse()) {
                // two hidden layers or more: the first product forms the seed in its loader, the last one contracts with
                // W_0 in its epilogue -- neither the seed matrix nor G_0 goes through memory (2 x 256, B*H = 20480, fp64:
                // 84 MB each way, twice)
                const long long Mr = (long long)(nx - 1) * Rp + Rm;
                T* G = nullptr;
                for (int l = nl - 3; l >= 0; --l) {
                    const bool first = l == nl - 3, last = l == 0;
                    T* Gn = (G == ws + o.g0) ? ws + o.g1 : ws + o.g0;
                    GemmArgs a{};
                    a.mode = LG_REVERSE;
                    a.Bw = h.d_Wt[l + 1]; a.ldb = h.dout[l];
                    a.M = (int)Mr; a.N = h.dout[l]; a.K = h.dout[l + 1]; a.Rmod = Rp;
                    a.D = ws + o.d[l]; a.ldd = Rp;
                    if (dfa(l)) { a.dact = h.act[l]; a.dactp = h.actp[l]; }
                    if (first) {
                        a.A = ws + o.d[nl - 2]; a.lda = Rp;
                        a.seedW = h.d_W[nl - 1]; a.seedDl = lin_skip ? static_cast<const T*>(nullptr) : ws + o.dl; a.seed_nx = nx;
                        if (dfa(nl - 2)) { a.sact = h.act[nl - 2]; a.sactp = h.actp[nl - 2]; }
                    } else {
                        a.A = G; a.lda = ldg;
                    }
                    const int nblk = (h.dout[l] + 63) / 64;
                    if (last) {
                        a.w0t = h.d_Wt[0]; a.ldw0 = h.din[0]; a.nin = nin;
                        a.Jp = nblk == 1 ? ws + o.j : Gn;
                        a.ldj = ldg; a.jp_stride = (long long)nin * ldg;
                    } else {
                        a.C = Gn; a.ldc = ldg;
                    }
                    if ((rc = gemm_reverse_fused<T>(s, a, first, last))) return rc;
                    jblk = 1; jsrc = ws + o.j; jstride = 0;
                    if (last && nblk > 1 && !rk4) {        // Discret / Unity: layered_finish_kernel adds the blocks itself
                        jblk = nblk; jsrc = Gn; jstride = a.jp_stride;
                    } else if (last && nblk > 1) {
                        const long long count = (long long)nin * ldg;
                        hipLaunchKernelGGL(layered_jreduce_kernel<T>, dim3((unsigned)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096)),
                                           dim3(256), 0, s, Gn, nblk, a.jp_stride, ws + o.j, count);
                        NEMPC_HIP(hipGetLastError());
                    }
                    G = Gn;
                }
            } else {
            T* G = ws + o.g0;
            hipLaunchKernelGGL(layered_seed_kernel<T>, dim3(rg.x, (unsigned)h.dout[nl - 2]), rb, 0, s, static_cast<const T*>(h.d_W[nl - 1]),
                               h.dout[nl - 2], nx, ws + o.dl, ws + o.d[nl - 2], R, Rp, G, dfa(nl - 2) ? h.act[nl - 2] : 0, (T)h.actp[nl - 2]);
            NEMPC_HIP(hipGetLastError());
            for (int l = nl - 3; l >= 0; --l) {
                // G_l = (W_{l+1} G_{l+1}) . D_l : K = dout[l+1], N = dout[l], operand W_{l+1}^T row-major (out, in) = d_Wt[l+1]
                T* Gn = (G == ws + o.g0) ? ws + o.g1 : ws + o.g0;
                // the nx blocks of Rp columns are covered as one run of columns; block k's columns beyond Rm are never read
                if ((rc = gemm<T>(h.num_cus, s, LG_REVERSE, 0, G, ldg, static_cast<const T*>(h.d_Wt[l + 1]), h.dout[l], Gn, ldg, ws + o.d[l], Rp,
                                  nullptr, (long long)(nx - 1) * Rp + Rm, h.dout[l], h.dout[l + 1], Rp, 0.0, dfa(l) ? h.act[l] : 0, h.actp[l])))
                    return rc;
                G = Gn;
            }
            // J^T[d][k Rp + r] = sum_o W_0[d][o] G_0[o][.]: operand W_0^T (out, in) = d_Wt[0], only the nin decision inputs
            {
                const long long Mj = (long long)(nx - 1) * Rp + R;
                if ((rc = skinny<T>(s, G, ldg, static_cast<const T*>(h.d_Wt[0]), h.din[0], h.dout[0], nin, Mj, ws + o.j, ldg,
                                    static_cast<const T*>(nullptr), 2, 0, static_cast<T*>(nullptr), T(0))))
                    return rc;
            }
            }
            if (rk4 && stage_out) {
                hipLaunchKernelGGL(layered_stage_record_kernel<T>, rg, rb, 0, s, st, nx, nin, r0, R, Rp, ws + o.xi, ws + o.j, ws + o.dk,
                                   static_cast<T*>(stage_out), stage_stride);
                NEMPC_HIP(hipGetLastError());
            }
            if (rk4) {
                hipLaunchKernelGGL(layered_rk4_kernel<T>, rg, rb, 0, s, st, nx, nin, cdt, (st == 0 || st == 3) ? T(1) : T(2), ws + o.f,
                                   ws + o.j, R, Rp, ws + o.kprev, ws + o.acck, ws + o.dk, ws + o.dkn, ws + o.accdk);
                NEMPC_HIP(hipGetLastError());
            }
        }
        hipLaunchKernelGGL(layered_finish_kernel<T>, dim3(rg.x, (unsigned)nx), rb, 0, s, gk, h.cfg.integrator, DT, nin, Z, X0, r0, R, Rp, fsrc ? fsrc : ws + o.f, jsrc,
                           rk4 ? ws + o.acck : nullptr, rk4 ? ws + o.accdk : nullptr, g, h.m, h.box ? 1 : 0, tiles, jblk, jstride, fblk, fstride,
                           static_cast<const T*>(h.d_b[nl - 1]));
        NEMPC_HIP(hipGetLastError());
    }
    return NEMPC_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Contracted network Hessian  sum_k mult_k d2 f_k / d xi^2  of every row on the same GEMM kernel (model/tensorflow.py:77-109
// takes tf.hessians per output; contracted with the multipliers as optimizer/ipopt.py:79-80 does).  The layer-wise form of
// rowhess_valu_kernel / net_hessian_contracted (kernels_valu.hip):
//
//     H = sum_l P_l^T diag(w_l) P_l,    P_l = d z_l / d xi  (width_l x nin, pre-activation tangents),
//                                       w_l = q_l . s_l''(z_l),   q_l = d(mult . f) / d a_l
//
//   forward            as the rows path, every layer also stores s''(z) = r2(a) s'(z)                    (GEMMs, R columns)
//   reverse            ONE cotangent (the multipliers): q_{l-1} = W_l (q_l . D_l); the epilogue also writes w_l = q_l . E_l
//                                                                                                         (GEMMs, R columns)
//   tangents           P_0 = W_0^T (constant), P_l = W_l^T (D_{l-1} . P_{l-1}): all nin directions side by side; the first
//                      product forms D_0 . W_0^T in its loader (the SEED form)                         (GEMMs, nin R columns)
//   contraction        per layer, H[p][q][r] += sum_j w_l[j][r] P_l[j][p, r] P_l[j][q, r]: streams P_l once per block pair of
//                      inputs, thread per row, four waves split the features                                (vector unit)
// (2 + nin) GEMM sweeps instead of the 2 + 2 nin of forward-over-reverse.  RK4 models keep the generic kernel for now.

struct LayeredHws {      // element offsets into the Hessian chunk workspace
    size_t xi, x0, x1, d[NEMPC_MAX_LAYERS], e[NEMPC_MAX_LAYERS], cw[NEMPC_MAX_LAYERS], f, dl, cl, wl, q0, q1, P, a0, a1, pl, hacc, l0p, total;
};

LayeredHws layered_hess_offsets(const Handle& h, size_t Rp) {
    LayeredHws o{};
    const int nx = h.cfg.nx, nin = h.nin;
    size_t p = 0;
    o.xi = p; p += (size_t)(nin + h.ne) * Rp;
    o.x0 = p; p += (size_t)h.maxw * Rp;
    o.x1 = p; p += (size_t)h.maxw * Rp;
    for (int l = 0; l < h.nl - 1; ++l) {
        o.d[l] = p; p += (size_t)h.dout[l] * Rp;
        o.e[l] = p; p += (size_t)h.dout[l] * Rp;
        o.cw[l] = p; p += (size_t)h.dout[l] * Rp;
    }
    o.f = p; p += (size_t)nx * Rp;
    o.dl = p; p += (size_t)nx * Rp;
    o.cl = p; p += (size_t)nx * Rp;
    o.wl = p; p += (size_t)nx * Rp;
    o.q0 = p; p += (size_t)h.maxw * Rp;
    o.q1 = p; p += (size_t)h.maxw * Rp;
    o.P = p; p += (size_t)h.maxw * nin * Rp;
    o.a0 = p; p += (size_t)h.maxw * nin * Rp;
    o.a1 = p; p += (size_t)h.maxw * nin * Rp;
    o.pl = p; p += (size_t)nx * nin * Rp;
    o.hacc = p; p += (size_t)nin * nin * Rp;
    // curvature terms per feature block as pair-major partial sums (<= 32 pairs): layer 0's (CONTRACT_REVERSE of the product
    // that forms q_0) and, with nin <= 4, every other hidden layer's (CONTRACT_HPAIR of its tangent product)
    o.l0p = p; p += (size_t)((h.maxw + 63) / 64) * 32 * Rp * (size_t)(nin <= 4 ? h.nl - 1 : 1);
    o.total = p;
    return o;
}

// multipliers of the chunk's rows, feature-major, times the output layer's derivatives: cl = mult . s_L' (the cotangent on
// z_L), wl = mult . s_L'' (its curvature weights; zero for a linear output layer; s_L'' itself is in wl on entry)
// (direct: the rows are the (row, stage) pairs of the RK4 pipeline, multipliers nu[(pair)][nx] row-major)
template <typename T>
__global__ void layered_hmult_kernel(int H, int nx, int m, const T* __restrict__ lam, int direct, long long r0, int R, long long Rp,
                                     const T* __restrict__ dl, int act, T* __restrict__ cl, T* __restrict__ wl) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const long long gr = r0 + r;
    const long long b = gr / H, t = gr - b * H;
    const T* lrow = direct ? lam + (size_t)gr * nx : lam + (size_t)b * m + (size_t)t * nx;
    for (int k = 0; k < nx; ++k) {
        const T mu = lrow[k];
        if (act == NEMPC_ACT_LINEAR) {
            cl[(size_t)k * Rp + r] = mu;
            wl[(size_t)k * Rp + r] = T(0);
        } else {
            cl[(size_t)k * Rp + r] = mu * dl[(size_t)k * Rp + r];
            wl[(size_t)k * Rp + r] = mu * wl[(size_t)k * Rp + r];          // (s_L''(z_L) on entry: the forward sweep's output step)
        }
    }
}

// the last hidden layer's cotangent: q[j][r] = sum_k W_last[j][k] cl[k][r];  w = q . E (curvature weights), delta = q . D
template <typename T>
__global__ void layered_hseed_kernel(const T* __restrict__ Wlast, int wdt, int nx, const T* __restrict__ cl, const T* __restrict__ Dh,
                                     const T* __restrict__ Eh, int R, long long Rp, T* __restrict__ delta, T* __restrict__ w,
                                     int dact, T dactp) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (r >= R || j >= wdt) return;
    T q = T(0);
    for (int k = 0; k < nx; ++k) q = fma(Wlast[(size_t)j * nx + k], cl[(size_t)k * Rp + r], q);
    w[(size_t)j * Rp + r] = q * lg_dval<T>(dact, dactp, 1, Eh[(size_t)j * Rp + r]);
    delta[(size_t)j * Rp + r] = q * lg_dval<T>(dact, dactp, 0, Dh[(size_t)j * Rp + r]);
}

// H[p][q][r] (+)= sum_j w[j][r] P[j][p Rp + r] P[j][q Rp + r] for the inputs p in block pb, q in block qb (PB each, q <= p
// kept).  A block is 64 rows x NW waves (4 .. 16, chosen by the host so that the launch has a few waves per SIMD: with 4
// at B*H = 20480 there were 1.25, 20 KB of loads in flight per CU and 1.3 - 1.6 TB/s); wave v sums the features j = v,
// v + NW, ...; the partial sums meet in LDS in a fixed tree.
// W0 != null: layer 0, whose tangents are the constants P[j][p] = W0[p][j] (W_0 row-major (in, out)).
template <typename T, int PB>
__global__ __launch_bounds__(PB == 4 ? 1024 : 512) void layered_hcontract_kernel(const T* __restrict__ P, long long ldp, const T* __restrict__ W0, int ldw0,
                                                                const T* __restrict__ w, int K, int nin, int R, long long Rp,
                                                                T* __restrict__ Hacc, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hc_lds_raw[];
    T* const red = reinterpret_cast<T*>(hc_lds_raw);       // [NW / 2][PB * PB][64]
    const int lane = threadIdx.x & 63, v = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int r = blockIdx.x * 64 + lane;
    const bool live = r < R;
    const int rc = live ? r : R - 1;
    // block pair (pb, qb) with qb <= pb out of blockIdx.y
    int pb = 0, rem = blockIdx.y;
    while (rem > pb) { rem -= pb + 1; ++pb; }
    const int qb = rem;
    const int p0 = pb * PB, q0 = qb * PB;
    T acc[PB][PB];
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) acc[i][j] = T(0);
    // UN features' loads in flight per lane before their multiply-adds (one feature at a time this loop was a chain of
    // dependent-latency round trips: 76 us per layer at 2 x 256, B*H = 20480, where the bytes are worth 15 - 40)
    constexpr int UN = PB == 4 ? (sizeof(T) == 8 ? 4 : 8) : (sizeof(T) == 8 ? 2 : 4);
    const bool diag = pb == qb;
    for (int j0 = v; j0 < K; j0 += NW * UN) {
        T wj[UN], tp[UN][PB], tq[UN][PB];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + NW * u;
            const bool in = j < K;
            const int jc = in ? j : K - 1;
            wj[u] = in ? w[(size_t)jc * Rp + rc] : T(0);
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int p = p0 + i, q = q0 + i;
                if (W0) {
                    tp[u][i] = p < nin ? W0[(size_t)p * ldw0 + jc] : T(0);
                    tq[u][i] = q < nin ? W0[(size_t)q * ldw0 + jc] : T(0);
                } else {
                    tp[u][i] = p < nin ? P[(size_t)jc * ldp + (size_t)p * Rp + rc] : T(0);
                    tq[u][i] = diag ? tp[u][i] : (q < nin ? P[(size_t)jc * ldp + (size_t)q * Rp + rc] : T(0));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const T pp = wj[u] * tp[u][i];
#pragma unroll
                for (int jj = 0; jj < PB; ++jj) acc[i][jj] = fma(pp, tq[u][jj], acc[i][jj]);
            }
    }
    // fixed tree: waves [h, 2h) hand their sums to waves [0, h), h = NW/2, NW/4, ... 1
    for (int hh = NW >> 1; hh >= 1; hh >>= 1) {
        if (v >= hh && v < 2 * hh) {
#pragma unroll
            for (int i = 0; i < PB; ++i)
#pragma unroll
                for (int jj = 0; jj < PB; ++jj) red[((v - hh) * PB * PB + i * PB + jj) * 64 + lane] = acc[i][jj];
        }
        __syncthreads();
        if (v < hh) {
#pragma unroll
            for (int i = 0; i < PB; ++i)
#pragma unroll
                for (int jj = 0; jj < PB; ++jj) acc[i][jj] += red[(v * PB * PB + i * PB + jj) * 64 + lane];
        }
        __syncthreads();
    }
    if (v == 0 && live) {
#pragma unroll
        for (int i = 0; i < PB; ++i)
#pragma unroll
            for (int jj = 0; jj < PB; ++jj) {
                const int p = p0 + i, q = q0 + jj;
                if (p < nin && q <= p) {
                    T* dst = Hacc + (size_t)(p * nin + q) * Rp + r;
                    *dst = accumulate ? *dst + acc[i][jj] : acc[i][jj];
                }
            }
    }
}

// direct mode: xi^T[d][r] from the stage records (row r0 + r = (row, stage) pair), the extra inputs of the pair's row
template <typename T>
__global__ void layered_hgather_direct_kernel(const T* __restrict__ stage, int stride, int nin, int ne, const T* __restrict__ extra,
                                              long long r0, int R, long long Rp, T* __restrict__ xi) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const long long gr = r0 + r;
    const T* rec = stage + (size_t)gr * stride;
    for (int d = 0; d < nin; ++d) xi[(size_t)d * Rp + r] = rec[d];
    for (int j = 0; j < ne; ++j) xi[(size_t)(nin + j) * Rp + r] = extra[(size_t)(gr >> 2) * ne + j];
}

// Ppair[n][p (p + 1) / 2 + q] = W_0[p][n] W_0[q][n], p >= q: with it layer 0's curvature term  sum_n w_0[n][r] W_0[p][n] W_0[q][n]
// is a contraction of w_0 = q_0 . E_0 over the features -- the CONTRACT epilogue of the product that forms q_0
template <typename T>
__global__ void layered_pairs_kernel(const T* __restrict__ W0, int ldw0, int dout0, int nin, T* __restrict__ P) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= dout0) return;
    const int np = nin * (nin + 1) / 2;
    for (int p = 0; p < nin; ++p)
        for (int q = 0; q <= p; ++q) P[(size_t)n * np + p * (p + 1) / 2 + q] = W0[(size_t)p * ldw0 + n] * W0[(size_t)q * ldw0 + n];
}

// blocks[(row)][p][q] row-major, both triangles, from the lower triangle of the accumulators (null: none) plus layer 0's term
// as the feature blocks' partial sums (L0: nblk0 blocks, stride0 apart, pair-major; null: it is in the accumulators)
template <typename T>
__global__ void layered_hfinish_kernel(int nin, long long r0, int R, long long Rp, const T* __restrict__ Hacc, const T* __restrict__ L0,
                                       int nblk0, long long stride0, T* __restrict__ blocks) {
    // (a thread per (row, pair) -- blockIdx.y = p (p + 1) / 2 + q: as a thread per row this was a chain of pairs x blocks dependent
    //  loads on 80 workgroups, 13 us at B*H = 20480)
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    int p = 0, q = blockIdx.y;
    while (q > p) { ++p; q -= p; }
    T* blk = blocks + (size_t)(r0 + r) * nin * nin;
    T v = Hacc ? Hacc[(size_t)(p * nin + q) * Rp + r] : T(0);
    if (L0) v += lg_blocksum<T>(L0 + (size_t)blockIdx.y * Rp + r, nblk0, stride0);
    blk[p * nin + q] = v;
    blk[q * nin + p] = v;
}

template <typename T>
int hcontract(hipStream_t s, const T* P, long long ldp, const T* W0, int ldw0, const T* w, int K, int nin, int R, long long Rp, T* Hacc,
              bool accumulate) {
    const int PBs = nin <= 4 ? 4 : 8;
    const int nb = (nin + PBs - 1) / PBs;
    const dim3 grid((unsigned)((R + 63) / 64), (unsigned)(nb * (nb + 1) / 2));
    // waves per block: as many as keep the whole launch resident at once (4096 waves at four per SIMD -- a second, partly
    // filled round of blocks doubled the time), within 4 .. 16 (8 for the 8 x 8 accumulator form: LDS) and the feature count
    int nw = 4;
    const int nwmax = PBs == 4 ? 16 : 8;
    while (nw * 2 <= nwmax && (long long)grid.x * grid.y * nw * 2 <= 4096 && nw * 2 <= K) nw *= 2;
    const size_t lds = (size_t)(nw / 2) * PBs * PBs * 64 * sizeof(T);
    const dim3 block((unsigned)(nw * 64));
    if (PBs == 4) {
        NEMPC_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(layered_hcontract_kernel<T, 4>), lds));
        hipLaunchKernelGGL((layered_hcontract_kernel<T, 4>), grid, block, lds, s, P, ldp, W0, ldw0, w, K, nin, R, Rp, Hacc, accumulate ? 1 : 0);
    } else {
        NEMPC_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(layered_hcontract_kernel<T, 8>), lds));
        hipLaunchKernelGGL((layered_hcontract_kernel<T, 8>), grid, block, lds, s, P, ldp, W0, ldw0, w, K, nin, R, Rp, Hacc, accumulate ? 1 : 0);
    }
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

// stage != null: direct mode -- `nrows` rows whose inputs are the stage records' xi and whose multipliers are lamv[(row)][nx]
template <typename T>
int run_layered_hess(Handle& h, int B, const void* Zv, const void* X0v, const void* lamv, void* blocksv, hipStream_t s,
                     const void* stagev = nullptr, int stage_stride = 0, long long nrows = 0) {
    const T* Z = static_cast<const T*>(Zv);
    const T* X0 = static_cast<const T*>(X0v);
    const T* lam = static_cast<const T*>(lamv);
    const T* stage = static_cast<const T*>(stagev);
    T* blocks = static_cast<T*>(blocksv);
    const int nx = h.cfg.nx, nin = h.nin, ne = h.ne, nl = h.nl, H = h.cfg.H;
    const long long rows = stage ? nrows : (long long)B * H;
    const long long Rc = h.layered_hess_chunk_rows;
    T* ws = static_cast<T*>(h.d_layered_hws);
    const RowGather gk = h.gather();
    const bool lin_out = h.act[nl - 1] == NEMPC_ACT_LINEAR;
    int rc;
    // layer 0's curvature term from the epilogue of the product that forms q_0: two hidden layers or more (there is such a
    // product), at most 32 input pairs (nin <= 7)
    const int npair = nin * (nin + 1) / 2;
    const bool fuse_l0 = layered_fuse() && nl >= 3 && npair <= 32;
    if (fuse_l0 && !h.layered_pairs_valid) {
        if (!h.d_layered_pairs) NEMPC_HIP(hipMalloc(&h.d_layered_pairs, (size_t)h.maxw * 32 * sizeof(double)));
        hipLaunchKernelGGL(layered_pairs_kernel<T>, dim3((unsigned)((h.dout[0] + 255) / 256)), dim3(256), 0, s, static_cast<const T*>(h.d_W[0]),
                           h.dout[0], h.dout[0], nin, static_cast<T*>(h.d_layered_pairs));
        NEMPC_HIP(hipGetLastError());
        h.layered_pairs_valid = true;
    }
    for (long long r0 = 0; r0 < rows; r0 += Rc) {
        const int R = (int)(rows - r0 < Rc ? rows - r0 : Rc);
        const long long Rp = ((long long)R + LG_BM - 1) / LG_BM * LG_BM;
        const LayeredHws o = layered_hess_offsets(h, (size_t)Rp);
        const dim3 rb(256), rg((unsigned)((R + 255) / 256));
        // ---- forward, every layer's s' and s'' kept
        const bool first = lg_first_on() && !stage && nl - 1 >= 2 && nin + ne <= LG_FIRST_KMAX;        // (see run_layered)
        static const bool outskip = [] { const char* e = getenv("NEMPC_LAYERED_OUTSKIP"); return !(e && atoi(e) == 0); }();
        const bool lin_noout = outskip && lin_out;
        if (stage)
            hipLaunchKernelGGL(layered_hgather_direct_kernel<T>, rg, rb, 0, s, stage, stage_stride, nin, ne, static_cast<const T*>(h.d_extra),
                               r0, R, Rp, ws + o.xi);
        else if (!first)
            hipLaunchKernelGGL(layered_gather_kernel<T>, rg, rb, 0, s, gk, nin, ne, static_cast<const T*>(h.d_extra), Z, X0, r0, R, Rp,
                               ws + o.xi, static_cast<const T*>(nullptr), T(0));
        NEMPC_HIP(hipGetLastError());
        const T* in = ws + o.xi;
        // Measured (profiles/r05_layered_dfa.txt): 2 x 256 fp64 503 -> 461 us, 3 x 256 898 -> 851 us; 4 x 512 RK4 83.9 -> 85.2 ms
        // (compute-bound products: the stores were free, the extra vector work in loader and epilogue is not) -- hence the
        // width rule
        auto dfa = [&](int l) { return lg_dfa_on() && (h.maxw <= 384 || lg_dfa_on() == 2) && lg_d_from_a(h.act[l]); };
        for (int l = 0; l < nl - 1; ++l) {
            T* out = dfa(l) ? ws + o.d[l] : ws + ((l & 1) ? o.x1 : o.x0);
            // the output layer in the last hidden layer's epilogue -- unless it is LINEAR: the Hessian sweeps then need nothing of
            // it (s_L' = 1, s_L'' = 0, and f is not an output of this callback): no contraction, no output step
            const bool contract = l == nl - 2 && layered_fuse() && !lin_noout;
            if (l == 0 && first) {
                launch_first<T>(h.act[0], nin + ne, dim3((unsigned)((R + 63) / 64), (unsigned)((h.dout[0] + 63) / 64)), s, gk,
                                nin, ne, static_cast<const T*>(h.d_extra), Z, X0, r0, R, Rp, static_cast<const T*>(h.d_W[0]), h.dout[0],
                                static_cast<const T*>(h.d_b[0]), h.act[0], (T)h.actp[0], out,
                                dfa(0) ? static_cast<T*>(nullptr) : ws + o.d[0], dfa(0) ? static_cast<T*>(nullptr) : ws + o.e[0]);
                NEMPC_HIP(hipGetLastError());
                in = out;
                continue;
            }
            GemmArgs a{};
            a.mode = LG_FORWARD; a.act = h.act[l]; a.actp = h.actp[l];
            a.A = in; a.lda = Rp; a.Bw = h.d_W[l]; a.ldb = h.dout[l]; a.bias = h.d_b[l];
            a.ldd = Rp;
            if (dfa(l)) {
                // the activation alone: in the layer's slot, through C (plain product) or through D (contracted product)
                a.D = contract ? ws + o.d[l] : static_cast<T*>(nullptr); a.E = nullptr; a.store_a = contract ? 1 : 0;
            } else {
                a.D = ws + o.d[l]; a.E = ws + o.e[l];
            }
            a.M = R; a.N = h.dout[l]; a.K = h.din[l];
            if (contract) {
                // (partial sums per feature block in the tangent buffer, free until the tangent sweep)
                a.w0t = h.d_W[nl - 1]; a.ldw0 = nx; a.nin = nx;
                a.Jp = ws + o.P; a.ldj = Rp; a.jp_stride = (long long)nx * Rp;
                if ((rc = gemm_forward<T, LG_CONTRACT_FORWARD>(h.num_cus, s, a))) return rc;
                hipLaunchKernelGGL(layered_outfinish_kernel<T>, rg, rb, 0, s, ws + o.P, (h.dout[l] + 63) / 64, a.jp_stride, nx, R, Rp,
                                   static_cast<const T*>(h.d_b[nl - 1]), h.act[nl - 1], (T)h.actp[nl - 1], ws + o.f, ws + o.dl, ws + o.wl);
                NEMPC_HIP(hipGetLastError());
                in = nullptr;
                break;
            }
            a.C = out; a.ldc = Rp;
            if ((rc = gemm_forward<T, LG_CONTRACT_NONE>(h.num_cus, s, a))) return rc;
            in = out;
        }
        if (in && !lin_noout &&
            (rc = skinny<T>(s, in, Rp, static_cast<const T*>(h.d_W[nl - 1]), nx, h.din[nl - 1], nx, (long long)R, ws + o.f, Rp,
                            static_cast<const T*>(h.d_b[nl - 1]), 0, h.act[nl - 1], ws + o.dl, (T)h.actp[nl - 1], ws + o.wl)))
            return rc;
        // ---- reverse with the multipliers as the one cotangent: curvature weights w_l of every hidden layer
        hipLaunchKernelGGL(layered_hmult_kernel<T>, rg, rb, 0, s, H, nx, h.m, lam, stage ? 1 : 0, r0, R, Rp, ws + o.dl, h.act[nl - 1],
                           ws + o.cl, ws + o.wl);
        NEMPC_HIP(hipGetLastError());
        T* dq = ws + o.q0;
        hipLaunchKernelGGL(layered_hseed_kernel<T>, dim3(rg.x, (unsigned)h.dout[nl - 2]), rb, 0, s, static_cast<const T*>(h.d_W[nl - 1]),
                           h.dout[nl - 2], nx, ws + o.cl, ws + o.d[nl - 2], dfa(nl - 2) ? ws + o.d[nl - 2] : ws + o.e[nl - 2], R, Rp, dq,
                           ws + o.cw[nl - 2], dfa(nl - 2) ? h.act[nl - 2] : 0, (T)h.actp[nl - 2]);
        NEMPC_HIP(hipGetLastError());
        for (int l = nl - 3; l >= 0; --l) {
            T* dn = (dq == ws + o.q0) ? ws + o.q1 : ws + o.q0;
            GemmArgs a{};
            a.mode = LG_REVERSE;
            a.A = dq; a.lda = Rp; a.Bw = h.d_Wt[l + 1]; a.ldb = h.dout[l];
            a.M = (int)Rp; a.N = h.dout[l]; a.K = h.dout[l + 1]; a.Rmod = Rp;
            a.D = ws + o.d[l]; a.E = dfa(l) ? ws + o.d[l] : ws + o.e[l]; a.ldd = Rp;
            if (dfa(l)) { a.dact = h.act[l]; a.dactp = h.actp[l]; }
            if (l == 0 && fuse_l0) {
                // layer 0's curvature term in this product's epilogue: w_0 = q_0 . E_0 contracted with the pair products
                a.D = dfa(0) ? ws + o.d[0] : ws + o.e[0];
                a.duse = 1;
                a.w0t = h.d_layered_pairs; a.ldw0 = npair; a.nin = npair;
                a.Jp = ws + o.l0p; a.ldj = Rp; a.jp_stride = (long long)npair * Rp;
                if ((rc = gemm_ft<T, 1, false, LG_CONTRACT_REVERSE>(s, a))) return rc;
                break;
            }
            a.C = l > 0 ? dn : nullptr; a.C2 = ws + o.cw[l]; a.ldc = Rp;
            // (two loads and two stores per element in the epilogue and only B*H columns: the small-launch rule of the
            // forward products applies)
            if ((rc = gemm_forward<T, LG_CONTRACT_NONE>(h.num_cus, s, a))) return rc;
            dq = dn;
        }
        // ---- layer 0: constant tangents W_0^T
        T* Hacc = ws + o.hacc;
        if (!fuse_l0 &&
            (rc = hcontract<T>(s, nullptr, 0, static_cast<const T*>(h.d_W[0]), h.dout[0], ws + o.cw[0], h.dout[0], nin, R, Rp, Hacc, false)))
            return rc;
        // ---- tangents of all nin inputs side by side (column p Rp + r), contracted layer by layer
        // Up to four inputs (round 5): the columns are INTERLEAVED -- a tile is 16 rows x nin inputs -- so that a lane of the
        // product holds the tangents of every input of its row, and the layer's curvature term leaves from the epilogue
        // (LG_CONTRACT_HPAIR): the tangents (126 MB at 2 x 256, B*H = 20480, fp64) are neither written nor read back, and the
        // contraction launch is gone.  Otherwise: the tangents go through memory to layered_hcontract_kernel.
        const bool fold = fuse_l0 && lin_out && nin >= 2 && nin <= 4 && lg_hfold_on();
        const long long ldt = fold ? ((long long)R + 15) / 16 * 16 * nin : (long long)nin * Rp;
        const long long Mt = fold ? ldt : (long long)(nin - 1) * Rp + Rp;
        const T* ta = nullptr;
        int pblocks = fuse_l0 ? (h.dout[0] + 63) / 64 : 0;        // feature blocks of partial sums behind o.l0p so far
        bool hacc_used = !fuse_l0;
        for (int l = 1; l < nl - 1; ++l) {
            T* tn = (ta == ws + o.a0) ? ws + o.a1 : ws + o.a0;
            const bool need_a = l < nl - 2 || !lin_out;     // D_l . P_l feeds the next layer (or the output layer's curvature)
            GemmArgs a{};
            a.mode = LG_REVERSE;
            a.Bw = h.d_W[l]; a.ldb = h.dout[l];
            a.M = (int)Mt; a.N = h.dout[l]; a.K = h.din[l]; a.Rmod = fold ? (long long)R : Rp;
            a.D = ws + o.d[l]; a.ldd = Rp;
            if (dfa(l)) { a.dact = h.act[l]; a.dactp = h.actp[l]; }
            a.C = need_a ? tn : nullptr; a.Craw = fold ? static_cast<T*>(nullptr) : ws + o.P; a.ldc = ldt;
            if (l == 1) {
                a.A = ws + o.d[0]; a.lda = Rp;
                a.seedW = h.d_Wt[0]; a.seed_nx = h.din[0]; a.seedDl = nullptr;
                if (dfa(0)) { a.sact = h.act[0]; a.sactp = h.actp[0]; }
            } else {
                a.A = ta; a.lda = ldt;
            }
            if (fold) {
                a.w0t = ws + o.cw[l];
                a.Jp = ws + o.l0p + (size_t)pblocks * npair * Rp; a.ldj = Rp; a.jp_stride = (long long)npair * Rp;
                if ((rc = gemm_hpair<T>(s, a, l == 1, nin))) return rc;
                pblocks += (h.dout[l] + 63) / 64;
            } else {
                if (l == 1) { if ((rc = gemm_ft<T, 1, true, LG_CONTRACT_NONE>(s, a))) return rc; }
                else if ((rc = gemm_ft<T, 1>(s, a))) return rc;
                if ((rc = hcontract<T>(s, ws + o.P, ldt, nullptr, 0, ws + o.cw[l], h.dout[l], nin, R, Rp, Hacc, hacc_used))) return rc;
                hacc_used = true;
            }
            ta = tn;
        }
        if (!lin_out) {
            // the output layer's own curvature: P_L = W_L^T (D_{L-2} . P_{L-2}), weights mult . s_L''
            const long long Mj = (long long)(nin - 1) * Rp + R;
            if ((rc = skinny<T>(s, ta, ldt, static_cast<const T*>(h.d_W[nl - 1]), nx, h.din[nl - 1], nx, Mj, ws + o.pl, ldt,
                                static_cast<const T*>(nullptr), 2, 0, static_cast<T*>(nullptr), T(0))))
                return rc;
            if ((rc = hcontract<T>(s, ws + o.pl, ldt, nullptr, 0, ws + o.wl, nx, nin, R, Rp, Hacc, hacc_used))) return rc;
            hacc_used = true;
        }
        hipLaunchKernelGGL(layered_hfinish_kernel<T>, dim3(rg.x, (unsigned)npair), rb, 0, s, nin, r0, R, Rp, hacc_used ? Hacc : static_cast<T*>(nullptr),
                           fuse_l0 ? ws + o.l0p : static_cast<T*>(nullptr), pblocks, (long long)npair * Rp, blocks);
        NEMPC_HIP(hipGetLastError());
    }
    return NEMPC_OK;
}

}  // namespace

// Which networks take this path: at least one hidden layer, decision + extra inputs within the skinny kernel's 32
// accumulators, nx within 16, plain or rolling-window models (the gather handles both); everything the register-resident
// matrix-core kernels (mfma_supported) do not take.
bool layered_supported(const Handle& h) {
    if (h.nl < 2 || h.nl > NEMPC_MAX_LAYERS) return false;
    if (h.nin > 32 || h.cfg.nx > 16 || h.maxw > 1024) return false;
    return true;
}

// chunk workspace: rows per chunk so that the whole workspace stays near 6 GB (of 288), between 4096 and 65536 rows -- the
// larger the products, the smaller the share of their launch tails (4 x 512, 6/3, B*H = 30720 in fp64 is one chunk of 2.3 GB)
int layered_prepare(Handle& h) {
    const size_t cap = (size_t)h.cfg.max_batch * h.cfg.H;
    const LayeredWs per = layered_offsets(h, 1);
    size_t rc_rows = ((size_t)6144 << 20) / (per.total * h.esz);
    if (rc_rows > 65536) rc_rows = 65536;
    if (rc_rows < 4096) rc_rows = 4096;
    if (const char* e = getenv("NEMPC_LAYERED_CHUNK_ROWS")) {     // (tests of the chunk loop)
        const long long v = atoll(e);
        if (v > 0) rc_rows = (size_t)v;
    }
    if (rc_rows > cap) rc_rows = cap;
    rc_rows = (rc_rows + LG_BM - 1) / LG_BM * LG_BM;
    if (h.d_layered_ws && h.layered_chunk_rows == (long long)rc_rows) return NEMPC_OK;
    if (h.d_layered_ws) (void)hipFree(h.d_layered_ws);
    h.d_layered_ws = nullptr;
    h.layered_chunk_rows = (long long)rc_rows;
    const size_t bytes = layered_offsets(h, rc_rows).total * h.esz;
    hipError_t e = hipMalloc(&h.d_layered_ws, bytes);
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc (layered workspace): ") + hipGetErrorString(e));
        return NEMPC_ENOMEM;
    }
    return NEMPC_OK;
}

void layered_free(Handle& h) {
    if (h.d_layered_ws) (void)hipFree(h.d_layered_ws);
    h.d_layered_ws = nullptr;
    if (h.d_layered_hws) (void)hipFree(h.d_layered_hws);
    h.d_layered_hws = nullptr;
    if (h.d_layered_pairs) (void)hipFree(h.d_layered_pairs);
    h.d_layered_pairs = nullptr;
}

// Hessian chunk workspace, sized like the rows one (about 6 GB at most, 4096 .. 65536 rows)
int layered_hess_prepare(Handle& h) {
    const size_t cap = (size_t)h.cfg.max_batch * h.cfg.H;
    const LayeredHws per = layered_hess_offsets(h, 1);
    size_t rc_rows = ((size_t)6144 << 20) / (per.total * h.esz);
    if (rc_rows > 65536) rc_rows = 65536;
    if (rc_rows < 4096) rc_rows = 4096;
    if (const char* e = getenv("NEMPC_LAYERED_CHUNK_ROWS")) {     // (tests of the chunk loop)
        const long long v = atoll(e);
        if (v > 0) rc_rows = (size_t)v;
    }
    if (rc_rows > cap) rc_rows = cap;
    rc_rows = (rc_rows + LG_BM - 1) / LG_BM * LG_BM;
    if (h.d_layered_hws && h.layered_hess_chunk_rows == (long long)rc_rows) return NEMPC_OK;
    if (h.d_layered_hws) (void)hipFree(h.d_layered_hws);
    h.d_layered_hws = nullptr;
    h.layered_hess_chunk_rows = (long long)rc_rows;
    const size_t bytes = layered_hess_offsets(h, rc_rows).total * h.esz;
    hipError_t e = hipMalloc(&h.d_layered_hws, bytes);
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc (layered Hessian workspace): ") + hipGetErrorString(e));
        return NEMPC_ENOMEM;
    }
    return NEMPC_OK;
}

static bool layered_hess_usable(const Handle& h);

// nempc_create / nempc_reserve: both chunk workspaces and the first-layer pair table of the Hessian, so that no callback
// allocates (the `*_prepare` calls in the launchers below then find everything in place and return at once)
int layered_reserve(Handle& h) {
    int rc = h.layered ? layered_prepare(h) : NEMPC_OK;      // (layered_hess handles: the Hessian workspace only)
    if (rc) return rc;
    if (layered_hess_usable(h)) {
        if ((rc = layered_hess_prepare(h))) return rc;
        if (!h.d_layered_pairs) {
            hipError_t e = hipMalloc(&h.d_layered_pairs, (size_t)h.maxw * 32 * sizeof(double));
            if (e != hipSuccess) {
                h.d_layered_pairs = nullptr;
                set_error(std::string("hipMalloc (layered pair table): ") + hipGetErrorString(e));
                return NEMPC_ENOMEM;
            }
        }
    }
    return NEMPC_OK;
}

// Lagrangian blocks on the GEMM path: Discret / Unity directly, RK4 through the stage pipeline of kernels_rk4hess.hip.
// NEMPC_EUNSUPPORTED: a nonlinear output layer behind a single hidden layer, NEMPC_LAYERED_HESS=0 (A/B knob).
static bool layered_hess_usable(const Handle& h) {
    static const bool off = [] { const char* e = getenv("NEMPC_LAYERED_HESS"); return e && atoi(e) == 0; }();
    if (off || !(h.layered || h.layered_hess)) return false;
    return h.nl >= 2 && !(h.nl == 2 && h.act[h.nl - 1] != NEMPC_ACT_LINEAR);
}

int launch_rowhess_layered(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks, hipStream_t s) {
#ifdef NEMPC_STAMPS
    g_lg_dbg = h.d_dbg;
#endif
    if (!layered_hess_usable(h)) return NEMPC_EUNSUPPORTED;
    if (h.cfg.integrator == NEMPC_RK4) return launch_rowhess_rk4_layered(h, B, Z, X0, lambda, blocks, s, nullptr, nullptr);
    int rc = layered_hess_prepare(h);
    if (rc) return rc;
    h.last_hess_kernel = 5;
    return h.cfg.dtype == NEMPC_F64 ? run_layered_hess<double>(h, B, Z, X0, lambda, blocks, s)
                                    : run_layered_hess<float>(h, B, Z, X0, lambda, blocks, s);
}

// direct mode (RK4 pipeline, step 3): contracted network Hessians of `nrows` (row, stage) pairs at the records' inputs
int launch_rowhess_layered_direct(Handle& h, long long nrows, const void* stage, int stride, const void* nu, void* out, hipStream_t s) {
    if (!layered_hess_usable(h)) return NEMPC_EUNSUPPORTED;
    int rc = layered_hess_prepare(h);
    if (rc) return rc;
    h.last_hess_kernel = 5;
    return h.cfg.dtype == NEMPC_F64 ? run_layered_hess<double>(h, 0, nullptr, nullptr, nu, out, s, stage, stride, nrows)
                                    : run_layered_hess<float>(h, 0, nullptr, nullptr, nu, out, s, stage, stride, nrows);
}

// rows with the stage records of the RK4 Hessian pipeline (step 1)
int launch_rows_layered_stages(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, void* stage_out, int stage_stride,
                               hipStream_t s) {
    int rc = layered_prepare(h);
    if (rc) return rc;
    h.last_row_kernel = 8;
    return h.cfg.dtype == NEMPC_F64 ? run_layered<double>(h, B, Z, X0, g, tiles, s, stage_out, stage_stride)
                                    : run_layered<float>(h, B, Z, X0, g, tiles, s, stage_out, stage_stride);
}

int launch_rows_layered(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, hipStream_t s) {
#ifdef NEMPC_STAMPS
    g_lg_dbg = h.d_dbg;
#endif
    int rc = layered_prepare(h);
    if (rc) return rc;
    h.last_row_kernel = 8;
    return h.cfg.dtype == NEMPC_F64 ? run_layered<double>(h, B, Z, X0, g, tiles, s) : run_layered<float>(h, B, Z, X0, g, tiles, s);
}

}  // namespace nempc
