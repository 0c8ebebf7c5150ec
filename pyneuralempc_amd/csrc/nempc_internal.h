// Internal declarations shared by the translation units of libnempc.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "nempc.h"

namespace nempc {

// codes of the dense / sparse assembly maps: >= 0 is an index into one problem's tile array
// (H, nx, nx+nu); negative codes are constants
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

// MFMA-packed network (built by nempc_set_weights when the MFMA row kernel is selected)
struct MfmaNet {
    int wp = 0;        // padded hidden width (multiple of 16): 32 | 64 | 128
    int nh = 0;        // hidden layers
    int kin = 0;       // padded input width (multiple of 4)
    int ldw = 0;       // leading dimension (elements) of every packed matrix
    void* blob = nullptr;  // device, dtype T; see kernels_mfma.hip for the layout
    size_t blob_elems = 0;
};

struct Handle {
    nempc_config cfg{};
    int n = 0, m = 0, nin = 0, nl = 0;
    int ne = 0;                  // extra network inputs (tvp + p); the network's input width is nin + ne
    const void* d_extra = nullptr;  // bound by nempc_bind_extra, (B,H,ne)
    int din[NEMPC_MAX_LAYERS]{}, dout[NEMPC_MAX_LAYERS]{};
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

    // objective, dtype T: Q, Qs=Q+Q^T, R, Rs, xref, uref, cx, cu (one allocation)
    void* d_obj = nullptr;
    // bounds (host)
    std::vector<double> box_lo, box_hi;

    // structure (host) + assembly maps (device)
    std::vector<int32_t> jac_rows, jac_cols, hess_rows, hess_cols;
    int32_t* d_dense_map = nullptr;   // (m*n)
    int32_t* d_sparse_map = nullptr;  // (nnz_jac)
    int32_t* d_hess_map = nullptr;    // (nnz_hess) index into the dense (n,n) scratch, see kernels_hess.hip

    // workspaces sized for max_batch
    void* d_tiles_ws = nullptr;  // (Bmax,H,nx,nin) when the caller does not ask for tiles
    void* d_g_ws = nullptr;      // (Bmax,m) when the caller does not ask for g
    void* d_valu_ws = nullptr;   // scratch of the generic row kernel
    size_t valu_ws_elems = 0;
    void* d_hess_ws = nullptr;   // (Bmax,H,nin,nin) per-row Lagrangian blocks
    long long* d_dbg = nullptr;  // diagnostic builds only
    void* solver_ws = nullptr;   // solver.hip
    mutable int last_row_kernel = 0;  // 1 valu, 2 coop, 3 wave-tile
};

struct ObjOffsets {  // element offsets into Handle::d_obj
    int Q, Qs, R, Rs, xref, uref, cx, cu, total;
};
inline ObjOffsets obj_offsets(int H, int nx, int nu) {
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
    o.total = p;
    return o;
}

// ---- kernels_valu.hip : generic thread-per-row kernel
size_t valu_workspace_elems(const Handle& h);
int launch_rows_valu(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, hipStream_t s);
int launch_rowhess_valu(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks,
                        hipStream_t s);

// ---- kernels_mfma.hip : matrix-core row kernel
bool mfma_supported(const Handle& h);
int mfma_pack_weights(Handle& h, const double* const* W, const double* const* b);
int launch_rows_mfma(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, hipStream_t s);
void mfma_free(Handle& h);
int launch_rowhess_mfma(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks,
                        hipStream_t s);

// ---- kernels_post.hip : objective, dense / sparse assembly, hessian assembly
int launch_objective(Handle& h, int B, const void* Z, void* f, void* grad, hipStream_t s);
int launch_assemble_dense(Handle& h, int B, const void* tiles, void* jac, hipStream_t s);
int launch_assemble_sparse(Handle& h, int B, const void* tiles, void* vals, hipStream_t s);
int launch_post(Handle& h, int B, const void* tiles, void* jac, const void* Z, void* f, void* grad, hipStream_t s);
int launch_assemble_hess(Handle& h, int B, const void* blocks, const void* sigma, void* hvals, void* hdense,
                         hipStream_t s);

// ---- solver.hip : batched Gauss-Newton SQP
int solver_run(Handle& h, int B, const void* X0, void* Z, const double* lb, const double* ub,
               const nempc_solver_opts& o, int32_t* status_dev, int32_t* iters_host, hipStream_t s);
void solver_free(Handle& h);

}  // namespace nempc
