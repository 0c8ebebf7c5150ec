// C ABI of libnempc.so: handle lifetime, parameter upload, structure export, launch sequencing.
// See include/nempc.h for the contract and the reference call sites each entry point stands for.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>

#include "nempc_internal.h"
#include "activations.h"

namespace nempc {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 65536) return hipSuccess;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = done[{dev, kernel}];
    if (bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

int hip_fail(hipError_t e, const char* what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return NEMPC_EHIP;
}

namespace {

int fail(int code, const std::string& msg) {
    set_error(msg);
    return code;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        if (prev == dev) prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <typename T>
int upload_as(const std::vector<double>& src, void* dst) {
    std::vector<T> tmp(src.size());
    for (size_t i = 0; i < src.size(); ++i) tmp[i] = (T)src[i];
    NEMPC_HIP(hipMemcpy(dst, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
    return NEMPC_OK;
}

int upload(const Handle& h, const std::vector<double>& src, void* dst) {
    if (src.empty()) return NEMPC_OK;
    return h.cfg.dtype == NEMPC_F64 ? upload_as<double>(src, dst) : upload_as<float>(src, dst);
}

int dev_alloc(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc: ") + hipGetErrorString(e));
        return NEMPC_ENOMEM;
    }
    return NEMPC_OK;
}

void dev_free(void*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

// decision variable (index into z) that tile / block column d of step t reads, or -1 when that window slot is
// data (x0 or the bound history).  Column order = the network's input order: w state rows then w control rows,
// oldest first unless rev (model/tensorflow.py:112-130).  For w = 1: [x_{t-1} | u_t].
int window_var(const Handle& h, int t, int d) {
    const int H = h.cfg.H, nx = h.cfg.nx, nu = h.cfg.nu, back = h.w - 1, wx = h.w * nx;
    if (d < wx) {
        const int j = d / nx, c = d % nx;
        const int tau = t + (h.rev ? -j : j - back);   // index into [x0 ; states]
        return tau >= 1 ? (tau - 1) * nx + c : -1;
    }
    d -= wx;
    const int j = d / nu, c = d % nu;
    const int tau = t + (h.rev ? -j : j - back);
    return tau >= 0 ? H * nx + tau * nu + c : -1;
}

int rebuild_structure(Handle& h) {
    const int H = h.cfg.H, nx = h.cfg.nx, nu = h.cfg.nu, nin = h.nin, n = h.n;
    h.m = H * nx + (h.box ? H * nx : 0);
    const int m = h.m;
    std::vector<int32_t> dense((size_t)m * n, MAP_ZERO), sparse;
    h.jac_rows.clear();
    h.jac_cols.clear();
    auto put = [&](int r, int c, int32_t code) {
        dense[(size_t)r * n + c] = code;
        h.jac_rows.push_back(r);
        h.jac_cols.push_back(c);
        sparse.push_back(code);
    };
    std::vector<std::pair<int, int32_t>> ent;
    for (int t = 0; t < H; ++t)
        for (int i = 0; i < nx; ++i) {
            const int r = t * nx + i;
            const int base = t * nx * nin + i * nin;
            ent.clear();
            for (int d = 0; d < nin; ++d) {
                const int v = window_var(h, t, d);
                if (v >= 0) ent.emplace_back(v, base + d);
            }
            ent.emplace_back(t * nx + i, MAP_MINUS_ONE);   // -x_t; the window only reaches state blocks < t
            std::sort(ent.begin(), ent.end());
            for (const auto& e : ent) put(r, e.first, e.second);
        }
    if (h.box)
        for (int k = 0; k < H * nx; ++k) put(H * nx + k, k, MAP_PLUS_ONE);

    void* p = h.d_dense_map;
    dev_free(p);
    p = h.d_sparse_map;
    dev_free(p);
    p = h.d_g_ws;
    dev_free(p);
    int rc;
    if ((rc = dev_alloc((void**)&h.d_dense_map, dense.size() * sizeof(int32_t)))) return rc;
    if ((rc = dev_alloc((void**)&h.d_sparse_map, sparse.size() * sizeof(int32_t)))) return rc;
    if ((rc = dev_alloc(&h.d_g_ws, (size_t)h.cfg.max_batch * m * h.esz))) return rc;
    NEMPC_HIP(hipMemcpy(h.d_dense_map, dense.data(), dense.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    NEMPC_HIP(hipMemcpy(h.d_sparse_map, sparse.data(), sparse.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return NEMPC_OK;
}

// Objective upload, in two parts.  The Hessian maps (tril structure, gather / scatter maps) depend on the problem's
// SHAPE only and are built once per handle; the parameter block [Q | Qs | R | Rs | xref | uref | cx | cu | QT | QTs |
// objective constants of every tril entry and of the dense (n,n) matrix] depends on the VALUES and is what a tracking
// MPC changes every step (a moving xref / uref): that path overwrites d_obj in place -- no map rebuild, no hipFree
// (a device synchronisation), no re-allocation.
int upload_objective(Handle& h, const ObjHost& o_in) {
    h.obj_host = o_in;
    const ObjHost& o = h.obj_host;
    const int H = h.cfg.H, nx = h.cfg.nx, nu = h.cfg.nu, n = h.n;
    ObjOffsets off = obj_offsets(H, nx, nu);
    std::vector<double> Qs(nx * nx), Rs(nu * nu), QT(h.obj_QT.empty() ? o.Q : h.obj_QT), QTs(nx * nx);
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < nx; ++j) {
            Qs[i * nx + j] = o.Q[i * nx + j] + o.Q[j * nx + i];
            QTs[i * nx + j] = QT[i * nx + j] + QT[j * nx + i];
        }
    for (int i = 0; i < nu; ++i)
        for (int j = 0; j < nu; ++j) Rs[i * nu + j] = o.R[i * nu + j] + o.R[j * nu + i];

    auto obj_const = [&](int r, int c, bool* structural) -> double {
        *structural = false;
        const bool rx = r < H * nx, cx = c < H * nx;
        if (rx && cx && r / nx == c / nx) {
            *structural = true;
            return (r / nx == H - 1 ? QTs : Qs)[(r % nx) * nx + (c % nx)];
        }
        if (!rx && !cx && (r - H * nx) / nu == (c - H * nx) / nu) {
            *structural = true;
            return Rs[((r - H * nx) % nu) * nu + ((c - H * nx) % nu)];
        }
        return 0.0;
    };

    int rc;
    if (!h.hess_maps_built) {
        // Hessian structure (tril, row-major) + maps.  Entry (r,c) sums the block elements of every step whose window
        // holds both variables (<= w of them; exactly one for plain models, integrator/discret.py:61-81) on top of the
        // constant objective term.
        const int nin = h.nin, w = h.w;
        std::vector<std::vector<int32_t>> codes((size_t)n * n);
        std::vector<int> wv(nin);
        for (int t = 0; t < H; ++t) {
            for (int d = 0; d < nin; ++d) wv[d] = window_var(h, t, d);
            for (int pp = 0; pp < nin; ++pp)
                for (int qq = 0; qq < nin; ++qq)
                    if (wv[pp] >= 0 && wv[qq] >= 0) codes[(size_t)wv[pp] * n + wv[qq]].push_back(t * nin * nin + pp * nin + qq);
        }
        h.hess_rows.clear();
        h.hess_cols.clear();
        std::vector<int32_t> hmap;
        auto emit = [&](int r, int c) {
            const auto& cs = codes[(size_t)r * n + c];
            for (int k = 0; k < w; ++k) hmap.push_back(k < (int)cs.size() ? cs[k] : -1);
        };
        for (int r = 0; r < n; ++r)
            for (int c = 0; c <= r; ++c) {
                bool st;
                (void)obj_const(r, c, &st);
                if (st || !codes[(size_t)r * n + c].empty()) {
                    emit(r, c);
                    h.hess_rows.push_back(r);
                    h.hess_cols.push_back(c);
                }
            }
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) emit(r, c);

        void* p = h.d_hess_map;
        dev_free(p);
        h.d_hess_map = nullptr;
        if ((rc = dev_alloc((void**)&h.d_hess_map, hmap.size() * sizeof(int32_t)))) return rc;
        NEMPC_HIP(hipMemcpy(h.d_hess_map, hmap.data(), hmap.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        // scatter form of the tril map for the Hessian kernel's fused assembly (plain models: one block element per entry)
        p = h.d_hess_smap;
        dev_free(p);
        h.d_hess_smap = nullptr;
        h.hess_n_orph = -1;
        if (w == 1) {
            const int nnz_t = (int)h.hess_rows.size(), be = H * nin * nin;
            std::vector<int32_t> smap((size_t)be, -1), orph;
            for (int e = 0; e < nnz_t; ++e) {
                if (hmap[e] >= 0) smap[hmap[e]] = e;
                else orph.push_back(e);
            }
            if ((int)orph.size() <= nin * nin) {
                smap.insert(smap.end(), orph.begin(), orph.end());
                if ((rc = dev_alloc((void**)&h.d_hess_smap, smap.size() * sizeof(int32_t)))) return rc;
                NEMPC_HIP(hipMemcpy(h.d_hess_smap, smap.data(), smap.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                h.hess_n_orph = (int)orph.size();
            }
        }
        h.hess_maps_built = true;
    }

    // parameter block: the family's parameters, then the objective constant of every tril entry and of the dense matrix
    const size_t nnz_t = h.hess_rows.size();
    std::vector<double> all((size_t)off.total + nnz_t + (size_t)n * n);
    std::copy(o.Q.begin(), o.Q.end(), all.begin() + off.Q);
    std::copy(Qs.begin(), Qs.end(), all.begin() + off.Qs);
    std::copy(QT.begin(), QT.end(), all.begin() + off.QT);
    std::copy(QTs.begin(), QTs.end(), all.begin() + off.QTs);
    std::copy(o.R.begin(), o.R.end(), all.begin() + off.R);
    std::copy(Rs.begin(), Rs.end(), all.begin() + off.Rs);
    std::copy(o.xref.begin(), o.xref.end(), all.begin() + off.xref);
    std::copy(o.uref.begin(), o.uref.end(), all.begin() + off.uref);
    std::copy(o.cx.begin(), o.cx.end(), all.begin() + off.cx);
    std::copy(o.cu.begin(), o.cu.end(), all.begin() + off.cu);
    {
        bool st;
        double* oc = all.data() + off.total;
        for (size_t e = 0; e < nnz_t; ++e) oc[e] = obj_const(h.hess_rows[e], h.hess_cols[e], &st);
        oc += nnz_t;
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) oc[(size_t)r * n + c] = obj_const(r, c, &st);
    }
    if (!h.d_obj || h.obj_elems != all.size()) {
        dev_free(h.d_obj);
        h.obj_elems = 0;
        if ((rc = dev_alloc(&h.d_obj, all.size() * h.esz))) return rc;
        h.obj_elems = all.size();
    }
    if ((rc = upload(h, all, h.d_obj))) return rc;
    h.have_objective = true;
    return NEMPC_OK;
}

void destroy_impl(Handle* h) {
    for (int l = 0; l < NEMPC_MAX_LAYERS; ++l) {
        dev_free(h->d_W[l]);
        dev_free(h->d_Wt[l]);
        dev_free(h->d_b[l]);
    }
    mfma_free(*h);
    layered_free(*h);
    rk4hess_free(*h);
    solver_free(*h);
    comm_free(*h);
    dev_free(h->d_obj);
    void* p = h->d_dense_map; dev_free(p); h->d_dense_map = nullptr;
    p = h->d_sparse_map; dev_free(p); h->d_sparse_map = nullptr;
    p = h->d_hess_map; dev_free(p); h->d_hess_map = nullptr;
    p = h->d_hess_smap; dev_free(p); h->d_hess_smap = nullptr;
    dev_free(h->d_tiles_ws);
    dev_free(h->d_g_ws);
    dev_free(h->d_valu_ws);
    dev_free(h->d_hess_ws);
    delete h;
}

}  // namespace
}  // namespace nempc

using namespace nempc;

extern "C" {

int nempc_abi_version(void) { return NEMPC_ABI_VERSION; }

const char* nempc_last_error(void) { return g_last_error.c_str(); }

int nempc_create(const nempc_config* cfg, nempc_handle* out) {
    if (!cfg || !out) return fail(NEMPC_EINVAL, "nempc_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != NEMPC_ABI_VERSION) return fail(NEMPC_EINVAL, "nempc_create: abi_version mismatch");
    if (cfg->dtype != NEMPC_F64 && cfg->dtype != NEMPC_F32) return fail(NEMPC_EINVAL, "nempc_create: bad dtype");
    if (cfg->integrator < NEMPC_DISCRET || cfg->integrator > NEMPC_RK4)
        return fail(NEMPC_EINVAL, "nempc_create: bad integrator kind");
    if (cfg->H < 1 || cfg->nx < 1 || cfg->nu < 1) return fail(NEMPC_EINVAL, "nempc_create: H, nx, nu must be >= 1");
    if (cfg->n_layers < 1 || cfg->n_layers > NEMPC_MAX_LAYERS)
        return fail(NEMPC_EINVAL, "nempc_create: n_layers out of range");
    if (cfg->widths[cfg->n_layers - 1] != cfg->nx)
        return fail(NEMPC_EINVAL, "nempc_create: last layer width must equal nx (model output = state dim)");
    for (int l = 0; l < cfg->n_layers; ++l) {
        if (cfg->widths[l] < 1) return fail(NEMPC_EINVAL, "nempc_create: layer width must be >= 1");
        if (cfg->activations[l] < 0 || cfg->activations[l] >= NEMPC_ACT_COUNT)
            return fail(NEMPC_EINVAL, "nempc_create: unknown activation code (NEMPC_ACT_*)");
        if (cfg->activations[l] == NEMPC_ACT_ELU && !(cfg->act_param[l] > 0.0))
            return fail(NEMPC_EINVAL, "nempc_create: elu needs act_param (alpha) > 0");
        if (cfg->activations[l] == NEMPC_ACT_LEAKY_RELU && !(cfg->act_param[l] >= 0.0))
            return fail(NEMPC_EINVAL, "nempc_create: leaky_relu needs act_param (alpha) >= 0");
    }
    if (cfg->max_batch < 1) return fail(NEMPC_EINVAL, "nempc_create: max_batch must be >= 1");
    if (cfg->n_extra < 0) return fail(NEMPC_EINVAL, "nempc_create: n_extra must be >= 0");
    if (cfg->kernel < NEMPC_KERNEL_AUTO || cfg->kernel > NEMPC_KERNEL_LAYERED)
        return fail(NEMPC_EINVAL, "nempc_create: bad kernel selector");
    if (cfg->integrator == NEMPC_RK4 && !(cfg->DT > 0.0)) return fail(NEMPC_EINVAL, "nempc_create: RK4 needs DT > 0");
    if (cfg->rolling_window < 1) return fail(NEMPC_EINVAL, "nempc_create: rolling_window must be >= 1");
    if (cfg->rolling_window > 1 && cfg->integrator == NEMPC_RK4)
        return fail(NEMPC_EUNSUPPORTED, "nempc_create: rolling-window models need the DISCRET or UNITY integrator");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(NEMPC_EHIP, "nempc_create: no HIP device visible (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(NEMPC_EINVAL, "nempc_create: device ordinal out of range");
    DeviceGuard dg(cfg->device);
    if (!dg.ok) return fail(NEMPC_EHIP, "nempc_create: hipSetDevice failed");

    Handle* h = new (std::nothrow) Handle();
    if (!h) return fail(NEMPC_ENOMEM, "nempc_create: out of host memory");
    h->cfg = *cfg;
    h->esz = cfg->dtype == NEMPC_F64 ? 8 : 4;
    h->w = cfg->rolling_window;
    h->rev = cfg->rolling_reverse ? 1 : 0;
    h->nin = h->w * (cfg->nx + cfg->nu);
    h->ne = cfg->n_extra;
    h->n = cfg->H * (cfg->nx + cfg->nu);
    h->nl = cfg->n_layers;
    h->maxw = 1;
    for (int l = 0; l < h->nl; ++l) {
        h->din[l] = l == 0 ? h->nin + h->ne : cfg->widths[l - 1];
        h->dout[l] = cfg->widths[l];
        if (l < h->nl - 1 && h->dout[l] > h->maxw) h->maxw = h->dout[l];
        h->act[l] = cfg->activations[l];
        h->actp[l] = cfg->act_param[l];
    }
    // matrix-core kernels: one non-linear activation on every hidden layer, linear output
    h->mfma_act = -1;
    if (h->nl >= 2 && h->act[h->nl - 1] == NEMPC_ACT_LINEAR && h->act[0] != NEMPC_ACT_LINEAR) {
        h->mfma_act = h->act[0];
        for (int l = 1; l < h->nl - 1; ++l)
            if (h->act[l] != h->act[0]) h->mfma_act = -1;
        // (the register-resident kernels are instantiated for tanh, relu, sigmoid, softplus and elu with alpha = 1)
        if (h->mfma_act > NEMPC_ACT_ELU) h->mfma_act = -1;
        for (int l = 0; l < h->nl - 1; ++l)
            if (h->act[l] == NEMPC_ACT_ELU && h->actp[l] != 1.0) h->mfma_act = -1;
    }
    // ... or any per-layer mix of the activations written from the layer's output (linear .. selu; elu / leaky_relu with
    // any alpha): the same kernels with the hidden layers' codes as launch arguments (NEMPC_ACT_RUNTIME instantiations)
    if (h->mfma_act < 0 && h->nl >= 2 && h->act[h->nl - 1] == NEMPC_ACT_LINEAR) {
        bool ok = true;
        for (int l = 0; l < h->nl - 1; ++l) ok = ok && h->act[l] >= NEMPC_ACT_LINEAR && h->act[l] < NEMPC_ACT_FIRST_ZBASED;
        static const bool rt_off = [] { const char* e = getenv("NEMPC_MFMA_RUNTIME_ACT"); return e && atoi(e) == 0; }();    // (A/B: the layered path instead)
        if (ok && !rt_off) h->mfma_act = NEMPC_ACT_RUNTIME;
    }
    {
        // compute units of the device: every "fill the chip" launch geometry is sized from this, never from a literal.
        // NEMPC_NUM_CUS overrides it (tests of the launch planning on the one device a box has).
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
            h->num_cus = prop.multiProcessorCount;
        if (const char* e = getenv("NEMPC_NUM_CUS")) {
            const int v = atoi(e);
            if (v > 0) h->num_cus = v;
        }
    }
    h->variant = NEMPC_KERNEL_VALU;
    // swish / gelu / softsign / mish / exponential / relu6 are written from the pre-activation: the layered path (any layer,
    // the output layer included: its output step has z) and the generic kernel (which keeps s', s'' of every unit from its
    // forward pass) take them; the register-resident matrix-core kernels do not (mfma_act stays -1 for such a network)
    if (cfg->kernel == NEMPC_KERNEL_MFMA || cfg->kernel == NEMPC_KERNEL_MFMA_TILE) {
        if (!mfma_supported(*h)) {
            delete h;
            return fail(NEMPC_EUNSUPPORTED, "nempc_create: MFMA row kernel does not cover these layer dims / activations");
        }
        h->variant = cfg->kernel;
    } else if (cfg->kernel == NEMPC_KERNEL_AUTO && mfma_supported(*h) && !mfma_slower_than_layered(*h)) {
        h->variant = NEMPC_KERNEL_MFMA;
        h->layered_hess = mfma_hess_on_layered(*h);
    } else if (cfg->kernel == NEMPC_KERNEL_LAYERED || cfg->kernel == NEMPC_KERNEL_AUTO) {
        // wide / deep / mixed-activation networks: rows through the layer-at-a-time GEMM pipeline (kernels_layered.hip).
        // Internally the handle stays on the generic variant -- Hessian blocks and everything else the pipeline does not
        // produce come from the generic kernels -- with launch_rows_valu handing the rows over
        if (layered_supported(*h)) {
            h->layered = true;
        } else if (cfg->kernel == NEMPC_KERNEL_LAYERED) {
            delete h;
            return fail(NEMPC_EUNSUPPORTED, "nempc_create: the layered matrix-core path needs >= 1 hidden layer, hidden widths <= 1024, "
                                            "w*(nx+nu) <= 32 and nx <= 16");
        }
    }

    int rc = NEMPC_OK;
    const size_t Bm = (size_t)cfg->max_batch;
    do {
        if ((rc = dev_alloc(&h->d_tiles_ws, Bm * cfg->H * cfg->nx * h->nin * h->esz))) break;
        // workspaces are sized and allocated HERE (and in nempc_reserve), not inside the first callback: an allocation is a
        // synchronising call, is not stream-capture safe, and an out-of-memory belongs to create, not to the middle of a solve
        if (h->layered) {
            if ((rc = layered_reserve(*h))) break;
        } else {
            if ((rc = ensure_valu_ws(*h))) break;
            if (h->layered_hess && (rc = layered_reserve(*h))) break;
        }
        if ((rc = dev_alloc(&h->d_hess_ws, Bm * cfg->H * h->nin * h->nin * h->esz))) break;
        if ((rc = rebuild_structure(*h))) break;
        ObjHost o;
        const int nx = cfg->nx, nu = cfg->nu, H = cfg->H;
        o.Q.assign(nx * nx, 0.0);
        o.R.assign(nu * nu, 0.0);
        for (int i = 0; i < nx; ++i) o.Q[i * nx + i] = 1.0;
        for (int i = 0; i < nu; ++i) o.R[i * nu + i] = 0.1;
        o.xref.assign(H * nx, 0.0); o.cx.assign(H * nx, 0.0);
        o.uref.assign(H * nu, 0.0); o.cu.assign(H * nu, 0.0);
        if ((rc = upload_objective(*h, o))) break;
    } while (0);
    if (rc) {
        std::string keep = g_last_error;
        destroy_impl(h);
        set_error(keep);
        return rc;
    }
    *out = reinterpret_cast<nempc_handle>(h);
    return NEMPC_OK;
}

int nempc_destroy(nempc_handle hh) {
    if (!hh) return NEMPC_OK;
    Handle* h = reinterpret_cast<Handle*>(hh);
    DeviceGuard dg(h->cfg.device);
    destroy_impl(h);
    return NEMPC_OK;
}

int nempc_reserve(nempc_handle hh, int32_t max_batch) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_reserve: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    if (max_batch < 1) return fail(NEMPC_EINVAL, "nempc_reserve: max_batch must be >= 1");
    if (max_batch <= h.cfg.max_batch) return NEMPC_OK;
    DeviceGuard dg(h.cfg.device);
    if (!dg.ok) return fail(NEMPC_EHIP, "nempc_reserve: hipSetDevice failed");
    // an evaluation may still be running out of the old workspaces on some stream
    NEMPC_HIP(hipDeviceSynchronize());
    h.cfg.max_batch = max_batch;
    const size_t Bm = (size_t)max_batch;
    rk4hess_free(h);      // sized lazily from max_batch on their next use
    solver_free(h);
    dev_free(h.d_tiles_ws); dev_free(h.d_valu_ws); dev_free(h.d_hess_ws); dev_free(h.d_g_ws);
    int rc = NEMPC_OK;
    do {
        if ((rc = dev_alloc(&h.d_tiles_ws, Bm * h.cfg.H * h.cfg.nx * h.nin * h.esz))) break;
        if (h.layered) {
            if ((rc = layered_reserve(h))) break;
        } else {
            if ((rc = ensure_valu_ws(h))) break;
            if (h.layered_hess && (rc = layered_reserve(h))) break;
        }
        if ((rc = dev_alloc(&h.d_hess_ws, Bm * h.cfg.H * h.nin * h.nin * h.esz))) break;
        if ((rc = dev_alloc(&h.d_g_ws, Bm * h.m * h.esz))) break;
    } while (0);
    if (rc) h.cfg.max_batch = 0;   // workspaces are gone: every evaluation is refused until a reserve succeeds
    return rc;
}

int nempc_set_weights(nempc_handle hh, const double* const* W, const double* const* b) {
    if (!hh || !W || !b) return fail(NEMPC_EINVAL, "nempc_set_weights: null argument");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    DeviceGuard dg(h.cfg.device);
    if (!dg.ok) return fail(NEMPC_EHIP, "nempc_set_weights: hipSetDevice failed");
    for (int l = 0; l < h.nl; ++l) {
        if (!W[l] || !b[l]) return fail(NEMPC_EINVAL, "nempc_set_weights: null layer pointer");
        const int win = h.din[l], wout = h.dout[l];
        for (size_t i = 0; i < (size_t)win * wout; ++i)
            if (!std::isfinite(W[l][i])) return fail(NEMPC_EINVAL, "nempc_set_weights: non-finite weight");
        std::vector<double> w(W[l], W[l] + (size_t)win * wout), wt((size_t)win * wout), bb(b[l], b[l] + wout);
        for (int i = 0; i < win; ++i)
            for (int j = 0; j < wout; ++j) wt[(size_t)j * win + i] = w[(size_t)i * wout + j];
        int rc;
        dev_free(h.d_W[l]); dev_free(h.d_Wt[l]); dev_free(h.d_b[l]);
        if ((rc = dev_alloc(&h.d_W[l], w.size() * h.esz))) return rc;
        if ((rc = dev_alloc(&h.d_Wt[l], wt.size() * h.esz))) return rc;
        if ((rc = dev_alloc(&h.d_b[l], bb.size() * h.esz))) return rc;
        if ((rc = upload(h, w, h.d_W[l]))) return rc;
        if ((rc = upload(h, wt, h.d_Wt[l]))) return rc;
        if ((rc = upload(h, bb, h.d_b[l]))) return rc;
    }
    if (h.variant != NEMPC_KERNEL_VALU) {
        int rc = mfma_pack_weights(h, W, b);
        if (rc) return rc;
    }
    h.have_weights = true;
    h.layered_pairs_valid = false;      // (the layered Hessian's table of first-layer weight products follows the weights)
    return NEMPC_OK;
}

int nempc_set_objective(nempc_handle hh, const double* Q, const double* R, const double* xref, const double* uref,
                        const double* cx, const double* cu) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_set_objective: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    DeviceGuard dg(h.cfg.device);
    if (!dg.ok) return fail(NEMPC_EHIP, "nempc_set_objective: hipSetDevice failed");
    const int nx = h.cfg.nx, nu = h.cfg.nu, H = h.cfg.H;
    ObjHost o;
    o.Q.assign(nx * nx, 0.0);
    o.R.assign(nu * nu, 0.0);
    if (Q) o.Q.assign(Q, Q + nx * nx); else for (int i = 0; i < nx; ++i) o.Q[i * nx + i] = 1.0;
    if (R) o.R.assign(R, R + nu * nu); else for (int i = 0; i < nu; ++i) o.R[i * nu + i] = 0.1;
    o.xref.assign(H * nx, 0.0); o.cx.assign(H * nx, 0.0);
    o.uref.assign(H * nu, 0.0); o.cu.assign(H * nu, 0.0);
    if (xref) o.xref.assign(xref, xref + H * nx);
    if (uref) o.uref.assign(uref, uref + H * nu);
    if (cx) o.cx.assign(cx, cx + H * nx);
    if (cu) o.cu.assign(cu, cu + H * nu);
    return upload_objective(h, o);
}

int nempc_set_terminal_weight(nempc_handle hh, const double* QT) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_set_terminal_weight: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    DeviceGuard dg(h.cfg.device);
    if (!dg.ok) return fail(NEMPC_EHIP, "nempc_set_terminal_weight: hipSetDevice failed");
    const int nx = h.cfg.nx;
    if (QT) h.obj_QT.assign(QT, QT + nx * nx); else h.obj_QT.clear();
    const ObjHost keep = h.obj_host;
    return upload_objective(h, keep);
}

int nempc_bind_extra(nempc_handle hh, const void* E, int32_t B) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_bind_extra: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    if (h.ne == 0 && E) return fail(NEMPC_EINVAL, "nempc_bind_extra: the handle was created with n_extra = 0");
    if (E && B < 1) return fail(NEMPC_EINVAL, "nempc_bind_extra: B (problems the tensor covers) must be >= 1");
    h.d_extra = E;
    h.extra_B = E ? B : 0;
    return NEMPC_OK;
}

int nempc_bind_history(nempc_handle hh, const void* hist_x, const void* hist_u, int32_t B) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_bind_history: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    if (h.w == 1 && (hist_x || hist_u))
        return fail(NEMPC_EINVAL, "nempc_bind_history: the handle was created with rolling_window = 1");
    if (h.w > 1 && (!hist_x != !hist_u)) return fail(NEMPC_EINVAL, "nempc_bind_history: bind both histories or neither");
    if (hist_x && B < 1) return fail(NEMPC_EINVAL, "nempc_bind_history: B (problems the tensors cover) must be >= 1");
    h.d_hist_x = hist_x;
    h.d_hist_u = hist_u;
    h.hist_B = hist_x ? B : 0;
    return NEMPC_OK;
}

int nempc_set_box_rows(nempc_handle hh, int enabled, const double* lo, const double* hi) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_set_box_rows: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    DeviceGuard dg(h.cfg.device);
    if (!dg.ok) return fail(NEMPC_EHIP, "nempc_set_box_rows: hipSetDevice failed");
    if (enabled && (!lo || !hi)) return fail(NEMPC_EINVAL, "nempc_set_box_rows: lo/hi required when enabled");
    h.box = enabled != 0;
    h.box_lo.clear();
    h.box_hi.clear();
    if (h.box) {
        h.box_lo.assign(lo, lo + h.cfg.nx);
        h.box_hi.assign(hi, hi + h.cfg.nx);
    }
    return rebuild_structure(h);
}

int nempc_dims(nempc_handle hh, int32_t* n, int32_t* m, int32_t* nnz_jac, int32_t* nnz_hess) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_dims: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    if (n) *n = h.n;
    if (m) *m = h.m;
    if (nnz_jac) *nnz_jac = (int32_t)h.jac_rows.size();
    if (nnz_hess) *nnz_hess = (int32_t)h.hess_rows.size();
    return NEMPC_OK;
}

int nempc_constraint_bounds(nempc_handle hh, double* cl, double* cu) {
    if (!hh || !cl || !cu) return fail(NEMPC_EINVAL, "nempc_constraint_bounds: null argument");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    const int hx = h.cfg.H * h.cfg.nx;
    for (int i = 0; i < hx; ++i) cl[i] = cu[i] = 0.0;
    if (h.box)
        for (int t = 0; t < h.cfg.H; ++t)
            for (int i = 0; i < h.cfg.nx; ++i) {
                cl[hx + t * h.cfg.nx + i] = h.box_lo[i];
                cu[hx + t * h.cfg.nx + i] = h.box_hi[i];
            }
    return NEMPC_OK;
}

int nempc_jac_structure(nempc_handle hh, int32_t* rows, int32_t* cols) {
    if (!hh || !rows || !cols) return fail(NEMPC_EINVAL, "nempc_jac_structure: null argument");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    std::memcpy(rows, h.jac_rows.data(), h.jac_rows.size() * sizeof(int32_t));
    std::memcpy(cols, h.jac_cols.data(), h.jac_cols.size() * sizeof(int32_t));
    return NEMPC_OK;
}

int nempc_hess_structure(nempc_handle hh, int32_t* rows, int32_t* cols) {
    if (!hh || !rows || !cols) return fail(NEMPC_EINVAL, "nempc_hess_structure: null argument");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    std::memcpy(rows, h.hess_rows.data(), h.hess_rows.size() * sizeof(int32_t));
    std::memcpy(cols, h.hess_cols.data(), h.hess_cols.size() * sizeof(int32_t));
    return NEMPC_OK;
}

int nempc_eval(nempc_handle hh, int32_t B, const void* Z, const void* X0, void* f, void* grad, void* g,
               void* jac_dense, void* jac_tiles, void* jac_sparse, void* stream) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_eval: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    if (B < 0 || B > h.cfg.max_batch) return fail(NEMPC_EINVAL, "nempc_eval: B outside [0, max_batch]");
    if (B == 0) return NEMPC_OK;
    if (!Z) return fail(NEMPC_EINVAL, "nempc_eval: Z is null");
    const bool need_rows = g || jac_dense || jac_tiles || jac_sparse;
    if (need_rows && !X0) return fail(NEMPC_EINVAL, "nempc_eval: X0 is null");
    if (need_rows && !h.have_weights) return fail(NEMPC_ESTATE, "nempc_eval: call nempc_set_weights first");
    if (need_rows && h.ne > 0 && !h.d_extra) return fail(NEMPC_ESTATE, "nempc_eval: n_extra > 0 but nempc_bind_extra was not called");
    if (need_rows && h.w > 1 && !h.d_hist_x)
        return fail(NEMPC_ESTATE, "nempc_eval: rolling_window > 1 but nempc_bind_history was not called");
    if (need_rows && ((h.ne > 0 && B > h.extra_B) || (h.w > 1 && B > h.hist_B)))
        return fail(NEMPC_EINVAL, "nempc_eval: B exceeds the batch the bound extras / history cover");
    DeviceGuard dg(h.cfg.device);
    if (!dg.ok) return fail(NEMPC_EHIP, "nempc_eval: hipSetDevice failed");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc;
    if (need_rows) {
        // defects only (IpoptProblem.constraints on its own): the matrix-core kernel skips its reverse sweeps
        const bool need_tiles = jac_dense || jac_tiles || jac_sparse || h.variant == NEMPC_KERNEL_VALU;
        void* tiles = jac_tiles ? jac_tiles : (need_tiles ? h.d_tiles_ws : nullptr);
        void* gout = g ? g : h.d_g_ws;
        // compiled shape: rows, objective and (dense contract) the dense assembly in ONE launch; without a derivative
        // output (g with f / grad: what a line-search trial needs) the same launch runs its forward passes only
        if (!jac_sparse && (jac_dense || f || grad) && h.variant == NEMPC_KERNEL_MFMA) {
            rc = launch_eval_fused(h, B, Z, X0, gout, jac_tiles, jac_dense, f, grad, s);
            if (rc != NEMPC_EUNSUPPORTED) return rc;
        }
        // sparse contract (what a real Ipopt run wants: SURVEY 8f-2; the reference hands cyipopt the dense (m, n),
        // optimizer/ipopt.py:88-96): rows, band values in nempc_jac_structure order and the objective from ONE launch -- of
        // the fixed-shape kernel on a compiled shape, of the cooperative kernel on every other plain-model shape it takes.
        // No tile round trip through memory, no assembly launch.
        if (jac_sparse && !jac_dense && h.variant == NEMPC_KERNEL_MFMA) {
            rc = launch_eval_fused(h, B, Z, X0, gout, jac_tiles, nullptr, f, grad, s, jac_sparse);
            if (rc != NEMPC_EUNSUPPORTED) return rc;
            rc = launch_rows_mfma_sparse(h, B, Z, X0, gout, jac_tiles, jac_sparse, f, grad, s);
            if (rc != NEMPC_EUNSUPPORTED) return rc;
        }
        // dense contract on any shape the cooperative kernel takes (plain models): the dense rows AND the objective leave
        // from the row launch itself -- no assembly launch, no tile round trip through memory, no objective launch
        if (jac_dense && !jac_sparse && h.variant == NEMPC_KERNEL_MFMA) {
            rc = launch_rows_mfma_dense(h, B, Z, X0, gout, jac_tiles, jac_dense, f, grad, s);
            if (rc != NEMPC_EUNSUPPORTED) return rc;
        }
        rc = h.variant != NEMPC_KERNEL_VALU ? launch_rows_mfma(h, B, Z, X0, gout, tiles, s)
                                            : launch_rows_valu(h, B, Z, X0, gout, tiles, s);
        if (rc) return rc;
        // sparse contract with the objective and without the dense matrix: one fused launch
        if (jac_sparse && !jac_dense && (f || grad)) return launch_post_sparse(h, B, tiles, jac_sparse, Z, f, grad, s);
        if (jac_sparse && (rc = launch_assemble_sparse(h, B, tiles, jac_sparse, s))) return rc;
        if (jac_dense && (f || grad)) return launch_post(h, B, tiles, jac_dense, Z, f, grad, s);
        if (jac_dense && (rc = launch_assemble_dense(h, B, tiles, jac_dense, s))) return rc;
    }
    if ((f || grad) && (rc = launch_objective(h, B, Z, f, grad, s))) return rc;
    return NEMPC_OK;
}

int nempc_hess(nempc_handle hh, int32_t B, const void* Z, const void* X0, const void* lambda, const void* sigma,
               void* hvals, void* hdense, void* hblocks, void* stream) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_hess: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    if (B < 0 || B > h.cfg.max_batch) return fail(NEMPC_EINVAL, "nempc_hess: B outside [0, max_batch]");
    if (B == 0) return NEMPC_OK;
    if (!Z || !X0 || !lambda || !sigma) return fail(NEMPC_EINVAL, "nempc_hess: null input");
    if (!h.have_weights) return fail(NEMPC_ESTATE, "nempc_hess: call nempc_set_weights first");
    if (h.ne > 0 && !h.d_extra) return fail(NEMPC_ESTATE, "nempc_hess: n_extra > 0 but nempc_bind_extra was not called");
    if (h.w > 1 && !h.d_hist_x)
        return fail(NEMPC_ESTATE, "nempc_hess: rolling_window > 1 but nempc_bind_history was not called");
    if ((h.ne > 0 && B > h.extra_B) || (h.w > 1 && B > h.hist_B))
        return fail(NEMPC_EINVAL, "nempc_hess: B exceeds the batch the bound extras / history cover");
    DeviceGuard dg(h.cfg.device);
    if (!dg.ok) return fail(NEMPC_EHIP, "nempc_hess: hipSetDevice failed");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc;
    // tril values only, on a cooperative-kernel shape: the Hessian kernel assembles them itself (one launch)
    if (hvals && !hdense && !hblocks && h.variant != NEMPC_KERNEL_VALU && h.cfg.integrator != NEMPC_RK4) {
        rc = launch_rowhess_mfma_hvals(h, B, Z, X0, lambda, sigma, hvals, s);
        if (rc != NEMPC_EUNSUPPORTED) return rc;
    }
    void* blocks = hblocks ? hblocks : h.d_hess_ws;
    if (h.variant == NEMPC_KERNEL_VALU) rc = launch_rowhess_valu(h, B, Z, X0, lambda, blocks, s);
    else if (h.cfg.integrator == NEMPC_RK4) {
        rc = launch_rowhess_rk4_mfma(h, B, Z, X0, lambda, blocks, s);
        // (a shape whose stage records would come from a wave-per-tile instantiation that is not used: generic kernel)
        if (rc == NEMPC_EUNSUPPORTED) rc = launch_rowhess_valu(h, B, Z, X0, lambda, blocks, s);
    } else rc = launch_rowhess_mfma(h, B, Z, X0, lambda, blocks, s);
    if (rc) return rc;
    if (!hvals && !hdense) return NEMPC_OK;
    return launch_assemble_hess(h, B, blocks, sigma, hvals, hdense, s);
}

int nempc_hess_gn(nempc_handle hh, int32_t B, const void* Z, const void* X0, const void* w, const void* sigma,
                  void* hvals, void* hdense, void* hblocks, void* stream) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_hess_gn: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    if (B < 0 || B > h.cfg.max_batch) return fail(NEMPC_EINVAL, "nempc_hess_gn: B outside [0, max_batch]");
    if (B == 0) return NEMPC_OK;
    if (!Z || !X0 || !sigma) return fail(NEMPC_EINVAL, "nempc_hess_gn: null input");
    if (!h.have_weights) return fail(NEMPC_ESTATE, "nempc_hess_gn: call nempc_set_weights first");
    if (h.ne > 0 && !h.d_extra) return fail(NEMPC_ESTATE, "nempc_hess_gn: n_extra > 0 but nempc_bind_extra was not called");
    if (h.w > 1 && !h.d_hist_x)
        return fail(NEMPC_ESTATE, "nempc_hess_gn: rolling_window > 1 but nempc_bind_history was not called");
    if ((h.ne > 0 && B > h.extra_B) || (h.w > 1 && B > h.hist_B))
        return fail(NEMPC_EINVAL, "nempc_hess_gn: B exceeds the batch the bound extras / history cover");
    DeviceGuard dg(h.cfg.device);
    if (!dg.ok) return fail(NEMPC_EHIP, "nempc_hess_gn: hipSetDevice failed");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // compiled shape, tril values only: blocks and assembly in the row launch's epilogue (one launch)
    if (hvals && !hdense && !hblocks && h.variant == NEMPC_KERNEL_MFMA) {
        const int rc1 = launch_hess_gn_fused(h, B, Z, X0, w, sigma, hvals, s);
        if (rc1 != NEMPC_EUNSUPPORTED) return rc1;
    }
    // first-order model only: the row kernel's tiles (no second-order sweep), then the same assembly as nempc_hess
    int rc = h.variant != NEMPC_KERNEL_VALU ? launch_rows_mfma(h, B, Z, X0, h.d_g_ws, h.d_tiles_ws, s)
                                            : launch_rows_valu(h, B, Z, X0, h.d_g_ws, h.d_tiles_ws, s);
    if (rc) return rc;
    // the caller does not ask for the per-row blocks: the assembly forms their elements where it uses them
    if (!hblocks) return (hvals || hdense) ? launch_assemble_hess_gn(h, B, h.d_tiles_ws, w, sigma, hvals, hdense, s) : NEMPC_OK;
    if ((rc = launch_gn_blocks(h, B, h.d_tiles_ws, w, hblocks, s))) return rc;
    if (!hvals && !hdense) return NEMPC_OK;
    return launch_assemble_hess(h, B, hblocks, sigma, hvals, hdense, s);
}

int nempc_solve(nempc_handle hh, int32_t B, const void* X0, void* Z, const double* lb, const double* ub,
                const nempc_solver_opts* opts, int32_t* status, int32_t* iters, void* stream) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_solve: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    if (B < 0 || B > h.cfg.max_batch) return fail(NEMPC_EINVAL, "nempc_solve: B outside [0, max_batch]");
    if (B == 0) { if (iters) *iters = 0; return NEMPC_OK; }
    if (!X0 || !Z || !opts || !status) return fail(NEMPC_EINVAL, "nempc_solve: null argument");
    if (!h.have_weights) return fail(NEMPC_ESTATE, "nempc_solve: call nempc_set_weights first");
    if (h.ne > 0 && !h.d_extra) return fail(NEMPC_ESTATE, "nempc_solve: n_extra > 0 but nempc_bind_extra was not called");
    if (h.ne > 0 && B > h.extra_B) return fail(NEMPC_EINVAL, "nempc_solve: B exceeds the batch the bound extras cover");
    if (h.w > 1 && !h.d_hist_x)
        return fail(NEMPC_ESTATE, "nempc_solve: rolling_window > 1 but nempc_bind_history was not called");
    if (h.w > 1 && B > h.hist_B) return fail(NEMPC_EINVAL, "nempc_solve: B exceeds the batch the bound history covers");
    if (opts->max_iter < 1 || opts->max_linesearch < 1 || !(opts->mu_factor > 0.0 && opts->mu_factor < 1.0) ||
        !(opts->mu_init > 0.0) || !(opts->mu_min > 0.0) || opts->lq_kernel < 0 || opts->lq_kernel > 3)
        return fail(NEMPC_EINVAL, "nempc_solve: bad options");
    DeviceGuard dg(h.cfg.device);
    if (!dg.ok) return fail(NEMPC_EHIP, "nempc_solve: hipSetDevice failed");
    return solver_run(h, B, X0, Z, lb, ub, *opts, status, iters, reinterpret_cast<hipStream_t>(stream));
}

int nempc_sync(nempc_handle hh, void* stream) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_sync: null handle");
    Handle& h = *reinterpret_cast<Handle*>(hh);
    DeviceGuard dg(h.cfg.device);
    NEMPC_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    return NEMPC_OK;
}

#ifdef NEMPC_STAMPS
// diagnostic library only (tools/diag_stamps.py): allocate / read the stamp buffer: 16 waves x 64 phase stamps of
// workgroup 0, then 16 words per workgroup (timeline of every workgroup, 4096 workgroups at most)
#define NEMPC_DBG_WORDS (1024 + 4096 * 16)
int nempc_debug_stamps(nempc_handle hh, long long* host_out) {
    Handle& h = *reinterpret_cast<Handle*>(hh);
    DeviceGuard dg(h.cfg.device);
    if (!h.d_dbg) {
        NEMPC_HIP(hipMalloc((void**)&h.d_dbg, sizeof(long long) * NEMPC_DBG_WORDS));
        NEMPC_HIP(hipMemset(h.d_dbg, 0, sizeof(long long) * NEMPC_DBG_WORDS));
        return NEMPC_OK;
    }
    NEMPC_HIP(hipDeviceSynchronize());
    NEMPC_HIP(hipMemcpy(host_out, h.d_dbg, sizeof(long long) * NEMPC_DBG_WORDS, hipMemcpyDeviceToHost));
    return NEMPC_OK;
}
#endif

int nempc_plan_grid(int32_t ntiles, int32_t num_cus, int32_t per_cu, int32_t* grid, int32_t* tiles_per_wg,
                    int32_t* tiles_rem) {
    if (ntiles < 1 || num_cus < 1 || per_cu < 1 || !grid || !tiles_per_wg || !tiles_rem)
        return fail(NEMPC_EINVAL, "nempc_plan_grid: bad argument");
    const GridPlan g = plan_grid(ntiles, num_cus, per_cu);
    *grid = g.grid; *tiles_per_wg = g.tiles_per_wg; *tiles_rem = g.tiles_rem;
    return NEMPC_OK;
}

int nempc_num_cus(nempc_handle hh) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_num_cus: null handle");
    return reinterpret_cast<Handle*>(hh)->num_cus;
}

int nempc_last_hess_kernel(nempc_handle hh) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_last_hess_kernel: null handle");
    return reinterpret_cast<Handle*>(hh)->last_hess_kernel;
}

int nempc_last_row_kernel(nempc_handle hh) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_last_row_kernel: null handle");
    return reinterpret_cast<Handle*>(hh)->last_row_kernel;
}

int nempc_kernel_variant(nempc_handle hh) {
    if (!hh) return fail(NEMPC_EINVAL, "nempc_kernel_variant: null handle");
    const Handle& h = *reinterpret_cast<Handle*>(hh);
    return h.layered ? NEMPC_KERNEL_LAYERED : h.variant;
}

}  // extern "C"
