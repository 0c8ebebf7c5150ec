// Internal declarations shared by the translation units of libnempc.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "nempc.h"

namespace nempc {

// codes of the dense / sparse assembly maps: >= 0 is an index into one problem's tile array
// (H, nx, w*(nx+nu)); negative codes are constants
constexpr int32_t MAP_ZERO = -1;
constexpr int32_t MAP_MINUS_ONE = -2;
constexpr int32_t MAP_PLUS_ONE = -3;

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);

#define NEMPC_HIP(call)                                            \
    do {                                                           \
        hipError_t _e = (call);                                    \
        if (_e != hipSuccess) return ::nempc::hip_fail(_e, #call); \
    } while (0)

// How one (problem, step) row reads its network inputs: plain models read [x_{t-1} | u_t]; rolling-window models
// (model/tensorflow.py:112-130 rolling_input) read the last w states and controls, reaching into x0 and the bound
// history before the horizon.  Tile / block column d indexes the window in the network's input order.
struct RowGather {
    int H, nx, nu, n;       // n = H*(nx+nu)
    int w, rev;             // window length, newest-first flag
    int xcur;               // tile column of the current state x_{t-1}, component 0 (where DISCRET adds its identity)
    unsigned inv_nx, inv_nu; // ceil(2^32 / nx), ceil(2^32 / nu) (0 for a divisor of 1): d / nx == umulhi(d, inv_nx), exact while d * nx < 2^32
    const void* hx;         // (B, w-1, nx) states before x0, oldest first
    const void* hu;         // (B, w-1, nu) controls before u_0
};

// value of window column d (< w*(nx+nu)) of row (b,t); z = this problem's decision vector
template <typename T>
__device__ __forceinline__ T gather_input(const RowGather& gk, const T* __restrict__ z, const T* __restrict__ X0,
                                          int b, int t, int d) {
    const int nx = gk.nx, nu = gk.nu;
    if (gk.w == 1) {
        if (d < nx) return t == 0 ? X0[(size_t)b * nx + d] : z[(t - 1) * nx + d];
        return z[gk.H * nx + t * nu + (d - nx)];
    }
    const int wx = gk.w * nx, back = gk.w - 1;
    if (d < wx) {
        const int j = nx == 1 ? d : (int)__umulhi((unsigned)d, gk.inv_nx), c = d - j * nx;
        const int tau = t + (gk.rev ? -j : j - back);   // index into [x0 ; states]: 0 is x0
        if (tau >= 1) return z[(tau - 1) * nx + c];
        if (tau == 0) return X0[(size_t)b * nx + c];
        return static_cast<const T*>(gk.hx)[((size_t)b * back + (back + tau)) * nx + c];
    }
    d -= wx;
    const int j = nu == 1 ? d : (int)__umulhi((unsigned)d, gk.inv_nu), c = d - j * nu;
    const int tau = t + (gk.rev ? -j : j - back);
    if (tau >= 0) return z[gk.H * nx + tau * nu + c];
    return static_cast<const T*>(gk.hu)[((size_t)b * back + (back + tau)) * nu + c];
}

// ... and where that value lives (the same case analysis, as an address: a caller with several columns to fetch forms all the
// addresses first and then has all the loads in flight together)
template <typename T>
__device__ __forceinline__ const T* gather_input_ptr(const RowGather& gk, const T* __restrict__ z, const T* __restrict__ X0,
                                                     int b, int t, int d) {
    const int nx = gk.nx, nu = gk.nu;
    if (gk.w == 1) {
        if (d < nx) return t == 0 ? X0 + ((size_t)b * nx + d) : z + ((t - 1) * nx + d);
        return z + (gk.H * nx + t * nu + (d - nx));
    }
    const int wx = gk.w * nx, back = gk.w - 1;
    if (d < wx) {
        const int j = nx == 1 ? d : (int)__umulhi((unsigned)d, gk.inv_nx), c = d - j * nx;
        const int tau = t + (gk.rev ? -j : j - back);   // index into [x0 ; states]: 0 is x0
        if (tau >= 1) return z + ((tau - 1) * nx + c);
        if (tau == 0) return X0 + ((size_t)b * nx + c);
        return static_cast<const T*>(gk.hx) + (((size_t)b * back + (back + tau)) * nx + c);
    }
    d -= wx;
    const int j = nu == 1 ? d : (int)__umulhi((unsigned)d, gk.inv_nu), c = d - j * nu;
    const int tau = t + (gk.rev ? -j : j - back);
    if (tau >= 0) return z + (gk.H * nx + tau * nu + c);
    return static_cast<const T*>(gk.hu) + (((size_t)b * back + (back + tau)) * nu + c);
}

// MFMA-packed network (built by nempc_set_weights when the MFMA row kernel is selected)
struct MfmaNet {
    int wp = 0;        // padded hidden width (multiple of 16): 32 | 64 | 128
    int nh = 0;        // hidden layers
    int kin = 0;       // padded input width (multiple of 4)
    int ldw = 0;       // leading dimension (elements) of every packed matrix
    void* blob = nullptr;  // device, dtype T; see kernels_mfma.hip for the layout
    size_t blob_elems = 0;
};

// internal return code of the matrix-core row launchers: "this launch belongs on the generic kernel" (translated by
// launch_rows_mfma_stages; never leaves the library)
constexpr int NEMPC_INTERNAL_USE_VALU = -9001;

struct ObjHost {   // host copy of the objective parameters (doubles), re-uploaded whenever one of them changes
    std::vector<double> Q, R, xref, uref, cx, cu;
};

// Launch geometry of the cooperative kernels: `per_cu` co-resident workgroups on each of the device's `num_cus`
// compute units (fewer when there are fewer tiles), each owning a contiguous run of tiles_per_wg (+1 for the first
// tiles_rem workgroups) 16-row tiles.  num_cus comes from hipDeviceProp_t::multiProcessorCount of the handle's device
// (256 on an MI355X in SPX mode; a partitioned mode or another gfx950 part reports its own count).
struct GridPlan {
    int grid, tiles_per_wg, tiles_rem;
};
inline GridPlan plan_grid(int ntiles, int num_cus, int per_cu) {
    long long g = (long long)(num_cus < 1 ? 1 : num_cus) * (per_cu < 1 ? 1 : per_cu);
    if (g > ntiles) g = ntiles;
    if (g < 1) g = 1;
    return GridPlan{(int)g, ntiles / (int)g, ntiles % (int)g};
}

struct Handle {
    nempc_config cfg{};
    int num_cus = 256;           // compute units of cfg.device (nempc_create)
    int n = 0, m = 0, nl = 0;
    int nin = 0;                 // tile width = decision inputs one row's network reads: w*(nx+nu)
    int ne = 0;                  // extra network inputs (tvp + p); the network's input width is nin + ne
    const void* d_extra = nullptr;  // bound by nempc_bind_extra, (B,H,ne)
    int extra_B = 0;                // problems the bound extras cover
    int hist_B = 0;                 // problems the bound history covers
    int w = 1, rev = 0;          // rolling window (1 = plain model), newest-first flag
    const void* d_hist_x = nullptr;  // bound by nempc_bind_history
    const void* d_hist_u = nullptr;
    RowGather gather() const {
        RowGather gk;
        gk.H = cfg.H; gk.nx = cfg.nx; gk.nu = cfg.nu; gk.n = n; gk.w = w; gk.rev = rev;
        gk.xcur = rev ? 0 : (w - 1) * cfg.nx;
        gk.inv_nx = cfg.nx == 1 ? 0u : (unsigned)(((1ull << 32) + cfg.nx - 1) / (unsigned)cfg.nx);
        gk.inv_nu = cfg.nu == 1 ? 0u : (unsigned)(((1ull << 32) + cfg.nu - 1) / (unsigned)cfg.nu);
        gk.hx = d_hist_x; gk.hu = d_hist_u;
        return gk;
    }
    int din[NEMPC_MAX_LAYERS]{}, dout[NEMPC_MAX_LAYERS]{};
    int act[NEMPC_MAX_LAYERS]{};   // NEMPC_ACT_* per layer (cfg.activations)
    double actp[NEMPC_MAX_LAYERS]{};   // alpha of elu / leaky_relu layers (cfg.act_param)
    int mfma_act = -1;             // the one hidden activation when the matrix-core kernels can take the network
                                   // (same non-linear activation on every hidden layer, linear output layer), else -1
    int maxw = 0;
    bool box = false;
    bool have_weights = false;
    bool have_objective = false;
    int variant = NEMPC_KERNEL_VALU;
    size_t esz = 8;

    // network, dtype T on device. W[l] (in,out) row-major; Wt[l] (out,in) row-major
    void* d_W[NEMPC_MAX_LAYERS]{};
    void* d_Wt[NEMPC_MAX_LAYERS]{};
    void* d_b[NEMPC_MAX_LAYERS]{};
    MfmaNet mfma;

    // objective, dtype T: Q, Qs=Q+Q^T, R, Rs, xref, uref, cx, cu, QT, QTs (one allocation) + Hessian constants
    void* d_obj = nullptr;
    size_t obj_elems = 0;        // elements of d_obj (a change of values overwrites it in place)
    bool hess_maps_built = false;   // the Hessian structure / maps depend on the shape only: built once per handle
    ObjHost obj_host;            // last nempc_set_objective arguments (defaults filled in)
    std::vector<double> obj_QT;  // terminal state weight (host copy; empty = same as Q), nempc_set_terminal_weight
    // bounds (host)
    std::vector<double> box_lo, box_hi;

    // structure (host) + assembly maps (device)
    std::vector<int32_t> jac_rows, jac_cols, hess_rows, hess_cols;
    int32_t* d_dense_map = nullptr;   // (m*n)
    int32_t* d_sparse_map = nullptr;  // (nnz_jac)
    int32_t* d_hess_map = nullptr;    // (nnz_hess + n*n, w) per-row block elements summed into each entry; -1 = none
    int32_t* d_hess_smap = nullptr;   // plain models: (H*nin*nin) block element -> tril entry (-1 none), then the entries no
    int hess_n_orph = -1;             //   block reaches (hess_n_orph of them; -1: no fused assembly for this handle)

    // workspaces sized for max_batch
    void* d_tiles_ws = nullptr;  // (Bmax,H,nx,nin) when the caller does not ask for tiles
    void* d_g_ws = nullptr;      // (Bmax,m) when the caller does not ask for g
    void* d_valu_ws = nullptr;   // scratch of the generic row kernel
    size_t valu_ws_elems = 0;
    void* d_hess_ws = nullptr;   // (Bmax,H,nin,nin) per-row Lagrangian blocks
    void* d_rk4_stage = nullptr; // RK4 Hessian pipeline (allocated on first use): (Bmax*H, 4, nin + 2*nx*nin) stage records,
    void* d_rk4_nu = nullptr;    //   (Bmax*H, 4, nx) stage multipliers,
    void* d_rk4_ht = nullptr;    //   (Bmax*H, 4, nin, nin) contracted stage Hessians
    long long* d_dbg = nullptr;  // diagnostic builds only
    bool layered = false;        // rows through the layer-at-a-time GEMM pipeline (kernels_layered.hip) instead of rows_valu_kernel
    void* d_layered_ws = nullptr;   // its chunk workspace (allocated on first use)
    long long layered_chunk_rows = 0;
    void* d_layered_hws = nullptr;  // chunk workspace of its Hessian sweeps (allocated on first use)
    long long layered_hess_chunk_rows = 0;
    bool layered_hess = false;          // a register-resident (MFMA) handle whose Lagrangian blocks come from the layered sweeps (mfma_hess_on_layered)
    void* d_layered_pairs = nullptr;    // (dout[0], nin (nin + 1) / 2): W_0[p][n] W_0[q][n], p >= q (layer 0's curvature term)
    bool layered_pairs_valid = false;
    void* solver_ws = nullptr;   // solver.hip
    void* comm = nullptr;        // comm.hip: RCCL communicator of the u0 all-gather
    mutable int last_row_kernel = 0;  // 1 valu, 2 coop, 3 wave-tile
    mutable int last_hess_kernel = 0; // 1 valu, 2 coop, 3 wave-tile, 4 fixed shape; + 10: inside the RK4 pipeline (nempc_last_hess_kernel)
};

struct ObjOffsets {  // element offsets into Handle::d_obj
    int Q, Qs, R, Rs, xref, uref, cx, cu, QT, QTs, total;   // QT: weight of the last step (terminal cost), QTs = QT + QT^T
};
__host__ __device__ inline ObjOffsets obj_offsets(int H, int nx, int nu) {
    ObjOffsets o;
    int p = 0;
    o.Q = p; p += nx * nx;
    o.Qs = p; p += nx * nx;
    o.R = p; p += nu * nu;
    o.Rs = p; p += nu * nu;
    o.xref = p; p += H * nx;
    o.uref = p; p += H * nu;
    o.cx = p; p += H * nx;
    o.cu = p; p += H * nu;
    o.QT = p; p += nx * nx;
    o.QTs = p; p += nx * nx;
    o.total = p;
    return o;
}

// Dynamic LDS above 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize, which is a per-device property of the
// kernel: remember the largest size configured per (device, kernel) so the attribute call stays off the hot path.
// Returns hipSuccess when nothing had to be done.
hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes);

// ---- kernels_valu.hip : generic thread-per-row kernel
size_t valu_workspace_elems(const Handle& h);
int ensure_valu_ws(Handle& h);          // generic kernels' scratch, allocated on first need (kernels_valu.hip)
int layered_reserve(Handle& h);         // layered path: both chunk workspaces + pair table (kernels_layered.hip)
int launch_rows_valu(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, hipStream_t s);
int launch_rowhess_valu(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks,
                        hipStream_t s);

// ---- kernels_layered.hip : layer-at-a-time matrix-core path for wide / deep / mixed-activation networks
bool layered_supported(const Handle& h);
int launch_rows_layered(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, hipStream_t s);
int launch_rowhess_layered(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks, hipStream_t s);
int launch_rowhess_layered_direct(Handle& h, long long nrows, const void* stage, int stride, const void* nu, void* out, hipStream_t s);
int launch_rows_layered_stages(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, void* stage_out, int stage_stride,
                               hipStream_t s);
// kernels_rk4hess.hip: the RK4 pipeline with the layered launches as steps 1 and 3
int launch_rowhess_rk4_layered(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks, hipStream_t s,
                               void* g_out, void* tiles_out);
void layered_free(Handle& h);

// ---- kernels_mfma.hip : matrix-core row kernel
bool mfma_supported(const Handle& h);
bool mfma_slower_than_layered(const Handle& h);    // AUTO prefers the layered path (kernels_mfma.hip)
bool mfma_hess_on_layered(const Handle& h);        // AUTO: rows on the register-resident kernels, Lagrangian blocks on the layered sweeps
int mfma_pack_weights(Handle& h, const double* const* W, const double* const* b);
int launch_rows_mfma(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, hipStream_t s);
int launch_eval_fused(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, void* jac, void* f,
                      void* grad, hipStream_t s, void* sparse = nullptr);
int launch_rows_mfma_sparse(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, void* sparse, void* f,
                            void* grad, hipStream_t s);
int launch_rows_mfma_dense(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, void* jac, void* f,
                           void* grad, hipStream_t s);
int launch_hess_gn_fused(Handle& h, int B, const void* Z, const void* X0, const void* w, const void* sigma, void* hvals,
                         hipStream_t s);
void mfma_free(Handle& h);
int launch_rowhess_mfma(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks,
                        hipStream_t s);
int launch_rowhess_mfma_hvals(Handle& h, int B, const void* Z, const void* X0, const void* lambda, const void* sigma,
                              void* hvals, hipStream_t s);
int launch_rows_mfma_stages(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, void* stage_out,
                            int stage_stride, hipStream_t s);
int launch_rowhess_mfma_direct(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks,
                               const void* xi_direct, int xi_stride, const void* lam_direct, int vdiv, hipStream_t s,
                               void* fuse_hvals = nullptr, const void* fuse_sigma = nullptr, void* ev_g = nullptr,
                               void* ev_tiles = nullptr);
int launch_rowhess_eval_mfma(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks, void* g,
                             void* tiles, hipStream_t s);

// ---- kernels_rk4hess.hip : RK4 Lagrangian blocks on the matrix cores (stage records -> stage multipliers ->
//      contracted network Hessians of the four stage inputs -> congruence sum)
int launch_rowhess_rk4_mfma(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks,
                            hipStream_t s, void* g_out = nullptr, void* tiles_out = nullptr);
void rk4hess_free(Handle& h);

// ---- kernels_post.hip : objective, dense / sparse assembly, hessian assembly
int launch_objective(Handle& h, int B, const void* Z, void* f, void* grad, hipStream_t s);
int launch_assemble_hess_gn(Handle& h, int B, const void* tiles, const void* w, const void* sigma, void* hvals, void* hdense,
                            hipStream_t s);
int launch_assemble_dense(Handle& h, int B, const void* tiles, void* jac, hipStream_t s);
int launch_assemble_sparse(Handle& h, int B, const void* tiles, void* vals, hipStream_t s);
int launch_post(Handle& h, int B, const void* tiles, void* jac, const void* Z, void* f, void* grad, hipStream_t s);
int launch_post_sparse(Handle& h, int B, const void* tiles, void* vals, const void* Z, void* f, void* grad,
                       hipStream_t s);
int launch_assemble_hess(Handle& h, int B, const void* blocks, const void* sigma, void* hvals, void* hdense,
                         hipStream_t s);
int launch_gn_blocks(Handle& h, int B, const void* tiles, const void* w, void* blocks, hipStream_t s);

// ---- solver.hip : batched Gauss-Newton SQP
int solver_run(Handle& h, int B, const void* X0, void* Z, const double* lb, const double* ub,
               const nempc_solver_opts& o, int32_t* status_dev, int32_t* iters_host, hipStream_t s);
void solver_free(Handle& h);

// ---- comm.hip : RCCL all-gather of the first controls
void comm_free(Handle& h);

}  // namespace nempc
