// Host side of the matrix-core row kernel: shape gate, weight packing, launch geometry.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "kernels_hess_impl.h"

namespace nempc {

namespace {

int padded_width(const Handle& h) {
    int w = 0;
    for (int l = 0; l < h.nl - 1; ++l) w = std::max(w, h.dout[l]);
    if (w <= 32) return 32;
    if (w <= 64) return 64;
    if (w <= 128) return 128;
    return 0;
}

MfmaOffsets make_offsets(int wp, int nh, int ks, int nx, int nin, int esz) {
    const int MT = wp / 16;
    MfmaOffsets o{};
    int p = 0;
    o.w0f = p; p += ks * MT * 64;
    for (int l = 1; l < nh; ++l) { o.wf[l] = p; p += MT * MT * 4 * 64; }
    o.wLf = p; p += MT * 4 * 64;
    for (int l = 1; l < nh; ++l) { o.wb[l] = p; p += MT * MT * 4 * 64; }
    o.w0b = p; p += MT * 4 * 64 * ((nin + 15) / 16);   // one fragment set per 16-row block of the input dimension
    o.p0tab = p; p += nin * MT * 16;
    o.wLb = p; p += ((nx + 3) / 4) * MT * 64;
    o.seed = p; p += nx * MT * 16;
    for (int l = 0; l < nh; ++l) { o.bias[l] = p; p += MT * 16; }
    o.biasL = p; p += 16;
    o.total = p;
    // cooperative-kernel copies (16-byte aligned: every size above is a multiple of 16 elements)
    const int vec = 16 / esz;
    const int nfrag = (nh - 1) * 2 * MT * 4 + 8;
    o.coop_small = p;
    o.coop_small_elems = ks * MT * 64 + (o.total - o.seed);
    p += (o.coop_small_elems + 15) & ~15;
    o.coop_nload = (nfrag + vec - 1) / vec;
    o.coop_slices = p;
    p += MT * o.coop_nload * 64 * vec;
    o.fx_small = p;
    o.fx_small_elems = o.coop_small_elems + nin * MT * 16;
    p += (o.fx_small_elems + 15) & ~15;
    o.grand_total = p;
    return o;
}

int scratch_elems(const Handle& h) {
    const int nx = h.cfg.nx, nin = h.nin;
    // xi0 | k | acck | J [| dk | accdk | dkn : RK4 chain only] | x_t slot (cooperative kernel) | extra inputs
    const int njs = h.cfg.integrator == NEMPC_RK4 ? 4 : 1;
    return 16 * nin + 2 * 16 * nx + njs * 16 * nx * nin + 16 * nx + 16 * h.ne;
}

}  // namespace

bool mfma_supported(const Handle& h) {
    const int nh = h.nl - 1;
    if (nh < 1 || nh > NEMPC_MFMA_MAX_HIDDEN || h.mfma_act < 0) return false;
    if (nh == 4 && padded_width(h) > 64) return false;             // a fourth hidden layer: up to width 64
    // network outputs on one 16-row block; inputs (window + extras) on up to kMaxKs k-steps, the window itself on up
    // to two 16-row blocks of the last reverse step (wave-per-tile kernels; the cooperative ones take <= 16)
    if (h.cfg.nx > 16 || h.nin + h.ne > 4 * kMaxKs || h.nin > 32) return false;
    return padded_width(h) != 0;
}

// Shapes the register-resident kernels take but the layered path runs faster (AUTO then picks the layered path; asking for
// NEMPC_KERNEL_MFMA by name still gets them): fp64 networks with run-time activation codes whose weight slices do not fit
// the cooperative kernel's registers (3 x 128, a fourth layer at width 64) -- the wave-per-tile kernel with the activation
// switch inlined spills (800 B of scratch per lane) and measured 304 us against the layered path's 187 (3 x 128 relu / tanh /
// sigmoid, B = 1024, H = 20; tools/narrow_bench.py).
bool mfma_slower_than_layered(const Handle& h) {
    if (h.mfma_act != NEMPC_ACT_RUNTIME || h.cfg.dtype != NEMPC_F64 || !layered_supported(h)) return false;
    const int nh = h.nl - 1, MT = padded_width(h) / 16;
    return ((nh - 1) * 2 * MT * 4 + 8) * 2 > 144;          // coop_fits_registers<double, WP, NH>() of kernels_mfma_typed.inc
}

// Shapes whose ROWS are fastest on the register-resident kernels but whose Lagrangian blocks are not (fp64, padded width 128,
// Discret / Unity; the RK4 pipeline keeps its own network kernel; tools/narrow_bench.py, B = 1024, H = 20, 2/1, us per callback):
//   three hidden layers, one activation: the weight slices do not fit the cooperative Hessian kernel, the wave-per-tile one
//     streams 1.5 MB of weights per tile: 565 against 250 for the layered sweeps (the rows: 154 against 150 - 190)
//   two hidden layers, run-time activation codes: the cooperative Hessian kernel with the activation switches inlined, 182
//     against 161 (the rows: 101 against 105)
bool mfma_hess_on_layered(const Handle& h) {
    // (AUTO only: asking for NEMPC_KERNEL_MFMA by name keeps the register-resident Hessian kernels -- that is the A/B)
    if (h.cfg.dtype != NEMPC_F64 || h.cfg.integrator == NEMPC_RK4 || !layered_supported(h)) return false;
    return padded_width(h) == 128 && (h.nl - 1 == 3 || (h.nl - 1 == 2 && h.mfma_act == NEMPC_ACT_RUNTIME));
}

void mfma_free(Handle& h) {
    if (h.mfma.blob) (void)hipFree(h.mfma.blob);
    h.mfma.blob = nullptr;
}

// Pack every layer into MFMA A-operand fragments (lane-linear: element [frag*64 + lane]).
// lane = kq*16 + i16 supplies A[M index i16][K index kq]; the K index of k-step (mt, r) stands for
// feature 16*mt + row(kq, r) where row() is the accumulator register->row map of the dtype
// (f64: kq + 4r, f32: 4kq + r), so that accumulator register r of tile mt is the matching B operand.
int mfma_pack_weights(Handle& h, const double* const* W, const double* const* b) {
    const int wp = padded_width(h), nh = h.nl - 1, MT = wp / 16;
    const int nx = h.cfg.nx, nin = h.nin, ks = (nin + h.ne + 3) / 4, L = h.nl - 1;
    const bool f64 = h.cfg.dtype == NEMPC_F64;
    auto row = [&](int q, int r) { return f64 ? MfmaOps<double>::row(q, r) : MfmaOps<float>::row(q, r); };
    const MfmaOffsets o = make_offsets(wp, nh, ks, nx, nin, (int)h.esz);
    const int MB = (nin + 15) / 16;
    std::vector<double> blob((size_t)o.grand_total, 0.0);
    auto Wat = [&](int l, int i, int j) -> double {
        return (i < h.din[l] && j < h.dout[l]) ? W[l][(size_t)i * h.dout[l] + j] : 0.0;
    };
    for (int lane = 0; lane < 64; ++lane) {
        const int i16 = lane & 15, kq = lane >> 4;
        for (int k = 0; k < ks; ++k)
            for (int mo = 0; mo < MT; ++mo) blob[o.w0f + (k * MT + mo) * 64 + lane] = Wat(0, 4 * k + kq, 16 * mo + i16);
        for (int mt = 0; mt < MT; ++mt)
            for (int r = 0; r < 4; ++r) {
                const int kf = 16 * mt + row(kq, r);
                for (int l = 1; l < nh; ++l)
                    for (int mo = 0; mo < MT; ++mo) {
                        blob[o.wf[l] + ((mt * 4 + r) * MT + mo) * 64 + lane] = Wat(l, kf, 16 * mo + i16);
                        blob[o.wb[l] + ((mt * 4 + r) * MT + mo) * 64 + lane] = Wat(l, 16 * mo + i16, kf);
                    }
                blob[o.wLf + (mt * 4 + r) * 64 + lane] = (i16 < nx) ? Wat(L, kf, i16) : 0.0;
                for (int mb = 0; mb < MB; ++mb)
                    blob[o.w0b + ((mt * 4 + r) * MB + mb) * 64 + lane] = (16 * mb + i16 < nin) ? Wat(0, 16 * mb + i16, kf) : 0.0;
            }
    }
    for (int lane = 0; lane < 64; ++lane) {   // W_L as an A operand with M = hidden unit, K = network output
        const int i16 = lane & 15, kq = lane >> 4;
        for (int k = 0; k < (nx + 3) / 4; ++k)
            for (int mt = 0; mt < MT; ++mt)
                blob[o.wLb + (k * MT + mt) * 64 + lane] = (4 * k + kq < nx) ? Wat(L, 16 * mt + i16, 4 * k + kq) : 0.0;
    }
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 4; ++r) {
            for (int mt = 0; mt < MT; ++mt) {
                const int f = 16 * mt + row(q, r);
                for (int pp = 0; pp < nin; ++pp) blob[o.p0tab + pp * MT * 16 + (mt * 4 + r) * 4 + q] = Wat(0, pp, f);
                for (int k = 0; k < nx; ++k) blob[o.seed + k * MT * 16 + (mt * 4 + r) * 4 + q] = Wat(L, f, k);
                for (int l = 0; l < nh; ++l) blob[o.bias[l] + (mt * 4 + r) * 4 + q] = (f < h.dout[l]) ? b[l][f] : 0.0;
            }
            const int oo = row(q, r);
            blob[o.biasL + r * 4 + q] = (oo < nx) ? b[L][oo] : 0.0;
        }

    // cooperative-kernel copies (only meaningful when the last reverse step has one 16-row block, MB == 1)
    {
        const int vec = 16 / (int)h.esz;
        for (int i = 0; i < ks * MT * 64; ++i) blob[o.coop_small + i] = blob[o.w0f + i];
        for (int i = 0; i < o.total - o.seed; ++i) blob[o.coop_small + ks * MT * 64 + i] = blob[o.seed + i];
        for (int w = 0; w < MT; ++w)
            for (int lane = 0; lane < 64; ++lane) {
                int fidx = 0;
                auto put = [&](double v) {
                    blob[o.coop_slices + ((size_t)(w * o.coop_nload + fidx / vec) * 64 + lane) * vec + fidx % vec] = v;
                    ++fidx;
                };
                for (int l = 1; l < nh; ++l) {
                    for (int i = 0; i < MT * 4; ++i) put(blob[o.wf[l] + (i * MT + w) * 64 + lane]);
                    for (int i = 0; i < MT * 4; ++i) put(blob[o.wb[l] + (i * MT + w) * 64 + lane]);
                }
                for (int r = 0; r < 4; ++r) put(blob[o.wLf + (w * 4 + r) * 64 + lane]);
                for (int r = 0; r < 4; ++r) put(blob[o.w0b + ((w * 4 + r) * MB) * 64 + lane]);
            }
    }

    for (int i = 0; i < o.coop_small_elems; ++i) blob[o.fx_small + i] = blob[o.coop_small + i];
    for (int i = 0; i < nin * MT * 16; ++i) blob[o.fx_small + o.coop_small_elems + i] = blob[o.p0tab + i];

    mfma_free(h);
    hipError_t e = hipMalloc(&h.mfma.blob, blob.size() * h.esz);
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc(mfma blob): ") + hipGetErrorString(e));
        return NEMPC_ENOMEM;
    }
    if (f64) {
        NEMPC_HIP(hipMemcpy(h.mfma.blob, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
    } else {
        std::vector<float> tmp(blob.begin(), blob.end());
        NEMPC_HIP(hipMemcpy(h.mfma.blob, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
    }
    h.mfma.wp = wp;
    h.mfma.nh = nh;
    h.mfma.kin = 4 * ks;
    h.mfma.blob_elems = blob.size();
    return NEMPC_OK;
}

// one translation unit per (dtype, hidden activation) defines these (kernels_mfma_typed.inc)
#define NEMPC_DECL_ACT(T, A)                                                                       \
    template <> int launch_rows_mfma_act<T, A>(const Handle& h, MfmaParams p, hipStream_t s);     \
    template <> int launch_rowhess_mfma_act<T, A>(const Handle& h, HessParams hp, hipStream_t s);
#define NEMPC_DECL_ACTS(T)                                                                         \
    NEMPC_DECL_ACT(T, NEMPC_ACT_TANH) NEMPC_DECL_ACT(T, NEMPC_ACT_RELU) NEMPC_DECL_ACT(T, NEMPC_ACT_SIGMOID) \
    NEMPC_DECL_ACT(T, NEMPC_ACT_SOFTPLUS) NEMPC_DECL_ACT(T, NEMPC_ACT_ELU) NEMPC_DECL_ACT(T, NEMPC_ACT_RUNTIME)
NEMPC_DECL_ACTS(double)
NEMPC_DECL_ACTS(float)
#undef NEMPC_DECL_ACTS
#undef NEMPC_DECL_ACT

template <typename T>
int launch_rows_mfma_typed(const Handle& h, MfmaParams p, hipStream_t s) {
    switch (h.mfma_act) {
        case NEMPC_ACT_TANH: return launch_rows_mfma_act<T, NEMPC_ACT_TANH>(h, p, s);
        case NEMPC_ACT_RELU: return launch_rows_mfma_act<T, NEMPC_ACT_RELU>(h, p, s);
        case NEMPC_ACT_SIGMOID: return launch_rows_mfma_act<T, NEMPC_ACT_SIGMOID>(h, p, s);
        case NEMPC_ACT_SOFTPLUS: return launch_rows_mfma_act<T, NEMPC_ACT_SOFTPLUS>(h, p, s);
        case NEMPC_ACT_ELU: return launch_rows_mfma_act<T, NEMPC_ACT_ELU>(h, p, s);
        case NEMPC_ACT_RUNTIME: return launch_rows_mfma_act<T, NEMPC_ACT_RUNTIME>(h, p, s);      // per-layer codes in p.acts
    }
    set_error("launch_rows_mfma: the matrix-core kernels do not take this network's activations");
    return NEMPC_EUNSUPPORTED;
}

template <typename T>
int launch_rowhess_mfma_typed(const Handle& h, HessParams hp, hipStream_t s) {
    switch (h.mfma_act) {
        case NEMPC_ACT_TANH: return launch_rowhess_mfma_act<T, NEMPC_ACT_TANH>(h, hp, s);
        case NEMPC_ACT_RELU: return launch_rowhess_mfma_act<T, NEMPC_ACT_RELU>(h, hp, s);
        case NEMPC_ACT_SIGMOID: return launch_rowhess_mfma_act<T, NEMPC_ACT_SIGMOID>(h, hp, s);
        case NEMPC_ACT_SOFTPLUS: return launch_rowhess_mfma_act<T, NEMPC_ACT_SOFTPLUS>(h, hp, s);
        case NEMPC_ACT_ELU: return launch_rowhess_mfma_act<T, NEMPC_ACT_ELU>(h, hp, s);
        case NEMPC_ACT_RUNTIME: return launch_rowhess_mfma_act<T, NEMPC_ACT_RUNTIME>(h, hp, s);
    }
    set_error("launch_rowhess_mfma: the matrix-core kernels do not take this network's activations");
    return NEMPC_EUNSUPPORTED;
}

int launch_rows_mfma(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, hipStream_t s) {
    return launch_rows_mfma_stages(h, B, Z, X0, g, tiles, nullptr, 0, s);
}

static MfmaParams base_params(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles) {
    MfmaParams p{};
    p.blob = h.mfma.blob;
    p.nx = h.cfg.nx; p.nu = h.cfg.nu; p.nin = h.nin; p.ks = (h.nin + h.ne + 3) / 4; p.mb = (h.nin + 15) / 16;
    p.ne = h.ne; p.extra = h.d_extra;
    p.off = make_offsets(h.mfma.wp, h.mfma.nh, p.ks, p.nx, p.nin, (int)h.esz);
    for (int l = 0; l < NEMPC_MFMA_MAX_HIDDEN; ++l) {
        p.acts.code[l] = l < h.nl - 1 ? h.act[l] : NEMPC_ACT_LINEAR;
        p.acts.par[l] = l < h.nl - 1 ? h.actp[l] : 0.0;
    }
    p.kind = h.cfg.integrator; p.DT = h.cfg.DT;
    p.B = B; p.H = h.cfg.H; p.m = h.m; p.box = h.box ? 1 : 0; p.num_cus = h.num_cus;
    p.Z = Z; p.X0 = X0; p.g = g; p.tiles = tiles;
    p.gk = h.gather();
    p.ntiles = (int)(((size_t)B * h.cfg.H + 15) / 16);
    p.scratch_per_wave = (scratch_elems(h) + 1) & ~1;
    p.dbg = h.d_dbg;
    return p;
}

// Whole hessian-free evaluation (g, dense jac, f, grad [, tiles]) in ONE launch where a fixed-shape kernel covers the
// problem; NEMPC_EUNSUPPORTED (no error message, nothing launched) tells nempc_eval to take the two-launch path.
int launch_eval_fused(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, void* jac, void* f,
                      void* grad, hipStream_t s, void* sparse) {
    if (!h.mfma.blob || !h.d_obj || (!jac && !f && !grad && !sparse)) return NEMPC_EUNSUPPORTED;
    if (sparse && (h.w != 1 || jac)) return NEMPC_EUNSUPPORTED;
    MfmaParams p = base_params(h, B, Z, X0, g, tiles);
    p.fuse_obj = true;
    p.fuse_jac = jac; p.fuse_f = f; p.fuse_grad = grad;
    p.fuse_sparse = sparse; p.sp_nnz = (int)h.jac_rows.size();
    p.obj = h.d_obj;
    p.oo = obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu);
    const int rc = h.cfg.dtype == NEMPC_F64 ? launch_rows_mfma_typed<double>(h, p, s) : launch_rows_mfma_typed<float>(h, p, s);
    if (rc == NEMPC_OK && sparse) h.last_row_kernel = 6;     // the fixed-shape kernel, band values included
    return rc;
}

// Rows, compact tiles (optional), the band-pattern Jacobian values and the objective from one launch of the cooperative
// kernel (any shape it takes, plain models): the sparse contract without the tile round trip and the assembly launch.
// NEMPC_EUNSUPPORTED (nothing launched, no error message) sends nempc_eval to the row + assembly launches.
int launch_rows_mfma_sparse(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, void* sparse, void* f,
                            void* grad, hipStream_t s) {
    static const int on = [] { const char* e = getenv("NEMPC_COOP_SPARSE"); return e ? atoi(e) : 1; }();
    if (!on || !h.mfma.blob || !sparse || h.w != 1 || ((f || grad) && !h.d_obj)) return NEMPC_EUNSUPPORTED;
    MfmaParams p = base_params(h, B, Z, X0, g, tiles);
    p.fuse_sparse = sparse; p.sp_nnz = (int)h.jac_rows.size();
    if (f || grad) {
        p.fuse_f = f; p.fuse_grad = grad; p.obj = h.d_obj;
        p.oo = obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu);
    }
    const int rc = h.cfg.dtype == NEMPC_F64 ? launch_rows_mfma_typed<double>(h, p, s) : launch_rows_mfma_typed<float>(h, p, s);
    if (rc == NEMPC_OK) h.last_row_kernel = 7;        // the cooperative kernel, band values included
    return rc;
}

// Gauss-Newton Hessian callback (tril values) in ONE launch of the fixed-shape kernel: the blocks sum_i w_i T_i^T T_i are
// formed and assembled in the row kernel's epilogue.  NEMPC_EUNSUPPORTED (nothing launched) for every other shape, for
// rolling windows, and when the Hessian map has no scatter form.
int launch_hess_gn_fused(Handle& h, int B, const void* Z, const void* X0, const void* w, const void* sigma, void* hvals,
                         hipStream_t s) {
    static const int on = [] { const char* e = getenv("NEMPC_GN_FUSED"); return e ? atoi(e) : 1; }();
    if (!on || !h.mfma.blob || !hvals || h.w != 1 || !h.d_hess_smap || h.hess_n_orph < 0 || h.cfg.dtype != NEMPC_F64)
        return NEMPC_EUNSUPPORTED;
    MfmaParams p = base_params(h, B, Z, X0, h.d_g_ws, nullptr);
    p.gn_hvals = hvals; p.gn_sigma = sigma; p.gn_w = w; p.gn_smap = h.d_hess_smap;
    p.gn_objc = (const char*)h.d_obj + (size_t)obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu).total * h.esz;
    p.gn_nnz = (int)h.hess_rows.size(); p.gn_n_orph = h.hess_n_orph;
    return launch_rows_mfma_typed<double>(h, p, s);
}

// Rows, compact tiles (optional) and the DENSE Jacobian from one launch of the cooperative kernel (any shape it takes,
// plain models): background zeros streamed at the start of each pass, non-zeros written with the pass's outputs.
// NEMPC_EUNSUPPORTED (nothing launched, no error message) sends nempc_eval to the row + assembly launches.
int launch_rows_mfma_dense(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, void* jac, void* f,
                           void* grad, hipStream_t s) {
    static const int on = [] { const char* e = getenv("NEMPC_COOP_DENSE"); return e ? atoi(e) : 1; }();
    if (!on || !h.mfma.blob || !jac || ((f || grad) && !h.d_obj)) return NEMPC_EUNSUPPORTED;
    MfmaParams p = base_params(h, B, Z, X0, g, tiles);
    p.fuse_jac = jac;
    if (f || grad) {        // the objective from the same launch (its workgroups take slices of the batch after their passes)
        p.fuse_f = f; p.fuse_grad = grad; p.obj = h.d_obj;
        p.oo = obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu);
    }
    const int rc = h.cfg.dtype == NEMPC_F64 ? launch_rows_mfma_typed<double>(h, p, s) : launch_rows_mfma_typed<float>(h, p, s);
    if (rc == NEMPC_OK) h.last_row_kernel = 5;        // the cooperative kernel, dense rows included
    return rc;
}

int launch_rows_mfma_stages(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, void* stage_out,
                            int stage_stride, hipStream_t s) {
    if (!h.mfma.blob) {
        set_error("launch_rows_mfma: weights not packed");
        return NEMPC_ESTATE;
    }
    MfmaParams p{};
    p.blob = h.mfma.blob;
    p.nx = h.cfg.nx; p.nu = h.cfg.nu; p.nin = h.nin; p.ks = (h.nin + h.ne + 3) / 4; p.mb = (h.nin + 15) / 16;
    p.ne = h.ne; p.extra = h.d_extra;
    p.off = make_offsets(h.mfma.wp, h.mfma.nh, p.ks, p.nx, p.nin, (int)h.esz);
    for (int l = 0; l < NEMPC_MFMA_MAX_HIDDEN; ++l) {
        p.acts.code[l] = l < h.nl - 1 ? h.act[l] : NEMPC_ACT_LINEAR;
        p.acts.par[l] = l < h.nl - 1 ? h.actp[l] : 0.0;
    }
    p.kind = h.cfg.integrator; p.DT = h.cfg.DT;
    p.B = B; p.H = h.cfg.H; p.m = h.m; p.box = h.box ? 1 : 0; p.num_cus = h.num_cus;
    p.Z = Z; p.X0 = X0; p.g = g; p.tiles = tiles;
    p.gk = h.gather();
    p.stage_out = stage_out; p.stage_stride = stage_stride;
    p.ntiles = (int)(((size_t)B * h.cfg.H + 15) / 16);
    p.scratch_per_wave = (scratch_elems(h) + 1) & ~1;
    p.dbg = h.d_dbg;
    const int rc = h.cfg.dtype == NEMPC_F64 ? launch_rows_mfma_typed<double>(h, p, s) : launch_rows_mfma_typed<float>(h, p, s);
    if (rc != NEMPC_INTERNAL_USE_VALU) return rc;
    // an instantiation of the wave-per-tile kernel that is not used (launch_shape): the generic kernel computes the rows.
    // It has no stage records: the RK4 Hessian pipeline's callers take their generic path on NEMPC_EUNSUPPORTED
    if (stage_out) return NEMPC_EUNSUPPORTED;
    if (!tiles && h.variant == NEMPC_KERNEL_MFMA) {
        // a defect-only launch (they take the wave-per-tile kernel): the cooperative kernel with its tiles into the workspace
        // is still far ahead of the generic kernel where it serves the shape
        p.tiles = h.d_tiles_ws;
        const int rc2 = h.cfg.dtype == NEMPC_F64 ? launch_rows_mfma_typed<double>(h, p, s) : launch_rows_mfma_typed<float>(h, p, s);
        if (rc2 != NEMPC_INTERNAL_USE_VALU) return rc2;
    }
    return launch_rows_valu(h, B, Z, X0, g, tiles ? tiles : h.d_tiles_ws, s);
}

int launch_rowhess_mfma(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks,
                        hipStream_t s) {
    if (h.layered_hess) {       // (mfma_hess_on_layered: this shape's blocks are faster on the layer-at-a-time sweeps)
        const int rc = launch_rowhess_layered(h, B, Z, X0, lambda, blocks, s);
        if (rc != NEMPC_EUNSUPPORTED) return rc;
    }
    return launch_rowhess_mfma_direct(h, B, Z, X0, lambda, blocks, nullptr, 0, nullptr, 1, s);
}

// blocks and assembly in one launch: the tril values of the Lagrangian Hessian written by the cooperative Hessian kernel
// itself.  NEMPC_EUNSUPPORTED (nothing launched) when the shape runs on the wave-per-tile kernel or the model has a
// rolling window (an entry then sums block elements of several rows).
int launch_rowhess_mfma_hvals(Handle& h, int B, const void* Z, const void* X0, const void* lambda, const void* sigma,
                              void* hvals, hipStream_t s) {
    if (h.w != 1 || !h.d_hess_smap || h.hess_n_orph < 0 || h.layered_hess) return NEMPC_EUNSUPPORTED;
    return launch_rowhess_mfma_direct(h, B, Z, X0, lambda, nullptr, nullptr, 0, nullptr, 1, s, hvals, sigma);
}

// Lagrangian blocks and the first-order evaluation (defects, compact tiles) of the same rows in ONE launch of the
// fixed-shape Hessian kernel -- the batched solver's trial point.  NEMPC_EUNSUPPORTED (nothing launched) for other shapes.
int launch_rowhess_eval_mfma(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks, void* g,
                             void* tiles, hipStream_t s) {
    static const int on = [] { const char* e = getenv("NEMPC_HFX_EVAL"); return e ? atoi(e) : 1; }();
    if (!on || !h.mfma.blob || h.w != 1 || h.ne != 0 || !blocks || !g || !tiles) return NEMPC_EUNSUPPORTED;
    return launch_rowhess_mfma_direct(h, B, Z, X0, lambda, blocks, nullptr, 0, nullptr, 1, s, nullptr, nullptr, g, tiles);
}

int launch_rowhess_mfma_direct(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks,
                               const void* xi_direct, int xi_stride, const void* lam_direct, int vdiv, hipStream_t s,
                               void* fuse_hvals, const void* fuse_sigma, void* ev_g, void* ev_tiles) {
    if (!h.mfma.blob) {
        set_error("launch_rowhess_mfma: weights not packed");
        return NEMPC_ESTATE;
    }
    HessParams hp{};
    MfmaParams& p = hp.base;
    p.blob = h.mfma.blob;
    p.nx = h.cfg.nx; p.nu = h.cfg.nu; p.nin = h.nin; p.ks = (h.nin + h.ne + 3) / 4; p.mb = (h.nin + 15) / 16;
    p.ne = h.ne; p.extra = h.d_extra;
    p.off = make_offsets(h.mfma.wp, h.mfma.nh, p.ks, p.nx, p.nin, (int)h.esz);
    for (int l = 0; l < NEMPC_MFMA_MAX_HIDDEN; ++l) {
        p.acts.code[l] = l < h.nl - 1 ? h.act[l] : NEMPC_ACT_LINEAR;
        p.acts.par[l] = l < h.nl - 1 ? h.actp[l] : 0.0;
    }
    p.kind = h.cfg.integrator; p.DT = h.cfg.DT;
    p.B = B; p.H = h.cfg.H; p.m = h.m; p.box = h.box ? 1 : 0; p.num_cus = h.num_cus;
    p.Z = Z; p.X0 = X0; p.g = nullptr; p.tiles = nullptr;
    p.gk = h.gather();
    p.ntiles = (int)(((size_t)B * h.cfg.H * (xi_direct ? vdiv : 1) + 15) / 16);
    hp.xi_direct = xi_direct; hp.xi_stride = xi_stride; hp.lam_direct = lam_direct; hp.vdiv = vdiv;
    p.scratch_per_wave = (16 * h.nin + 16 * h.cfg.nx + 16 * h.nin * h.nin + 16 * h.ne + 1) & ~1;
    p.dbg = nullptr;
    hp.lambda = lambda; hp.blocks = blocks;
    hp.ev_g = ev_g; hp.ev_tiles = ev_tiles;
    if (fuse_hvals) {       // the kernel assembles the tril values itself (launch_rowhess_mfma_hvals)
        hp.hvals = fuse_hvals; hp.sigma = fuse_sigma; hp.smap = h.d_hess_smap;
        hp.objc = (const char*)h.d_obj + (size_t)obj_offsets(h.cfg.H, h.cfg.nx, h.cfg.nu).total * h.esz;
        hp.nnz = (int)h.hess_rows.size(); hp.n_orph = h.hess_n_orph;
    }
    hp.p0tab = p.off.p0tab; hp.wLb = p.off.wLb; hp.ksx = (h.cfg.nx + 3) / 4;
    return h.cfg.dtype == NEMPC_F64 ? launch_rowhess_mfma_typed<double>(h, hp, s)
                                    : launch_rowhess_mfma_typed<float>(h, hp, s);
}

}  // namespace nempc
