// Generic row kernel: one thread per (problem b, step t) row, any layer dims, fp64 or fp32.
//
// This is the shape-agnostic path (and the A/B baseline for the matrix-core kernel in
// kernels_mfma.hip).  Each lane walks the network for its own row; per-row intermediates live in a
// global workspace laid out [slot][row] so that every access is coalesced across the wave, and the
// weights are wave-uniform (scalar loads).  Outputs are register-blocked 8 wide.
//
// Math per row (reference call sites integrator/discret.py:22-30,48-56,
// unity.py:24-32, rk4.py:66-80,137-159):
//   xi = [x_{t-1} ; u_t] (or the rolling window of both),  f(xi) = MLP (any per-layer activation of activations.h, the
//   output layer included),  J = df/dxi by one reverse sweep per output,
//   DISCRET Phi = x + f, dPhi = J + [I 0];  UNITY Phi = f;  RK4 per rk4.py with
//   dk_{i+1} = J_{i+1} + c_i DT J_{i+1}[:, :nx] dk_i   (== J_{i+1} (I + c_i DT [dk_i; 0])).
#include "activations.h"
#include "nempc_internal.h"

namespace nempc {

namespace {

struct NetDev {
    int nl, nin, nx, nu, maxw, ne;   // nin = w*(nx+nu) decision inputs (tile width); the network reads nin + ne values per row
    const void* extra;               // (B,H,ne) or null
    RowGather gk;                    // where the nin window inputs of a row come from
    int din[NEMPC_MAX_LAYERS], dout[NEMPC_MAX_LAYERS];
    int act[NEMPC_MAX_LAYERS];       // NEMPC_ACT_* per layer
    double actp[NEMPC_MAX_LAYERS];   // alpha of elu / leaky_relu layers
    const void* W[NEMPC_MAX_LAYERS];
    const void* Wt[NEMPC_MAX_LAYERS];
    const void* b[NEMPC_MAX_LAYERS];
};

struct WsOff {  // slot offsets (multiply by Rcap)
    int xi, act, cot, fout, jst, kcur, dk, acck, accdk, dkn, P, cL, rk_xin, rk_J, rk_dk, rk_nu, htmp, htmp2, d1s, d2s, total;
};

WsOff ws_offsets(const Handle& h) {
    WsOff o;
    int nhid = h.nl - 1, p = 0;
    o.xi = p; p += h.nin + h.ne;
    o.act = p; p += nhid * h.maxw;
    o.cot = p; p += 2 * h.maxw;
    o.fout = p; p += h.cfg.nx;
    o.jst = p; p += h.cfg.nx * h.nin;
    o.kcur = p; p += h.cfg.nx;
    o.dk = p; p += h.cfg.nx * h.nin;
    o.acck = p; p += h.cfg.nx;
    o.accdk = p; p += h.cfg.nx * h.nin;
    o.dkn = p; p += h.cfg.nx * h.nin;
    o.P = p; p += (nhid * h.maxw + h.cfg.nx) * h.nin;   // pre-activation tangents: hidden layers, then the output layer
    o.cL = p; p += h.cfg.nx;                            // cotangent w.r.t. the output layer's pre-activation
    // RK4 Lagrangian Hessian: per stage input, Jacobian, previous chain Jacobian, stage multiplier; two nin^2 temporaries
    o.rk_xin = p; p += 4 * h.nin;
    o.rk_J = p; p += 4 * h.cfg.nx * h.nin;
    o.rk_dk = p; p += 4 * h.cfg.nx * h.nin;
    o.rk_nu = p; p += 4 * h.cfg.nx;
    o.htmp = p; p += h.nin * h.nin;
    o.htmp2 = p; p += h.nin * h.nin;
    // s'(z) and s''(z) of every unit (hidden layers, then the output layer), written by the forward pass: from the output for
    // the monotone activations, from the pre-activation for swish / gelu / softsign / mish / exponential / relu6
    o.d1s = p; p += nhid * h.maxw + h.cfg.nx;
    o.d2s = p; p += nhid * h.maxw + h.cfg.nx;
    o.total = p;
    return o;
}

// forward through the network for this lane's row; hidden activations -> ws act, output -> ws fout
template <typename T>
__device__ void net_forward(const NetDev& net, T* ws, const WsOff& o, size_t R, size_t r) {
    for (int l = 0; l < net.nl; ++l) {
        const T* W = (const T*)net.W[l];
        const T* bias = (const T*)net.b[l];
        const int win = net.din[l], wout = net.dout[l];
        const T* in = (l == 0) ? ws + (size_t)o.xi * R : ws + (size_t)(o.act + (l - 1) * net.maxw) * R;
        const bool last = (l == net.nl - 1);
        T* out = last ? ws + (size_t)o.fout * R : ws + (size_t)(o.act + l * net.maxw) * R;
        for (int jb = 0; jb < wout; jb += 8) {
            T acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = (jb + q < wout) ? bias[jb + q] : T(0);
            if (jb + 8 <= wout) {
                for (int i = 0; i < win; ++i) {
                    const T a = in[(size_t)i * R + r];
                    const T* w = W + (size_t)i * wout + jb;
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] = fma(a, w[q], acc[q]);
                }
            } else {
                for (int i = 0; i < win; ++i) {
                    const T a = in[(size_t)i * R + r];
                    const T* w = W + (size_t)i * wout + jb;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (jb + q < wout) acc[q] = fma(a, w[q], acc[q]);
                }
            }
            // unit index in the derivative slots: hidden layer l at l * maxw, the output layer behind the hidden ones
            const int u0 = (last ? (net.nl - 1) * net.maxw : l * net.maxw) + jb;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (jb + q < wout) {
                    T a, d1, d2;
                    if (act_zbased(net.act[l])) {
                        act_from_z<T>(net.act[l], acc[q], a, d1, d2);
                    } else {
                        a = act_f<T>(net.act[l], acc[q], (T)net.actp[l]);
                        d1 = act_d1<T>(net.act[l], a, (T)net.actp[l]);
                        d2 = act_r2<T>(net.act[l], a, (T)net.actp[l]) * d1;
                    }
                    out[(size_t)(jb + q) * R + r] = a;
                    ws[(size_t)(o.d1s + u0 + q) * R + r] = d1;
                    ws[(size_t)(o.d2s + u0 + q) * R + r] = d2;
                }
        }
    }
}

// reverse sweep for output k: J[k][d] -> ws jst.  mult != nullptr is unused here (see hessian).
template <typename T>
__device__ void net_jacobian_row(const NetDev& net, T* ws, const WsOff& o, size_t R, size_t r, int k) {
    const int nl = net.nl;
    T* jrow = ws + (size_t)(o.jst + k * net.nin) * R;
    // the output layer's own activation (1 for the usual linear output)
    const T* d1s = ws + (size_t)o.d1s * R;      // s'(z) of every unit (net_forward)
    const T dLk = d1s[(size_t)((nl - 1) * net.maxw + k) * R + r];
    if (nl == 1) {
        const T* W0 = (const T*)net.W[0];
        for (int d = 0; d < net.nin; ++d) jrow[(size_t)d * R + r] = W0[(size_t)d * net.dout[0] + k] * dLk;
        return;
    }
    // seed at the last hidden layer: cot[j] = W_L[j][k] s_L'(z_L[k]) s'(z_{L-1}[j])
    int cur = 0;
    {
        const T* WL = (const T*)net.W[nl - 1];
        const int w = net.din[nl - 1];
        T* c = ws + (size_t)(o.cot + cur * net.maxw) * R;
        const bool lin_out = net.act[nl - 1] == NEMPC_ACT_LINEAR;
        for (int j = 0; j < w; ++j) {
            T wv = WL[(size_t)j * net.dout[nl - 1] + k];
            if (!lin_out) wv *= dLk;
            c[(size_t)j * R + r] = wv * d1s[(size_t)((nl - 2) * net.maxw + j) * R + r];
        }
    }
    // hidden layers nl-2 .. 1 : cot_in[i] = (sum_j W_l[i][j] cot[j]) s'(z_{l-1}[i])
    for (int l = nl - 2; l >= 0; --l) {
        const T* Wt = (const T*)net.Wt[l];  // (out, in) row-major: Wt[j][i] = W[i][j]
        const int win = net.din[l], wout = net.dout[l];
        const T* c = ws + (size_t)(o.cot + cur * net.maxw) * R;
        const bool first = (l == 0);
        T* cn = first ? jrow : ws + (size_t)(o.cot + (cur ^ 1) * net.maxw) * R;
        for (int ib = 0; ib < win; ib += 8) {
            T acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = T(0);
            for (int j = 0; j < wout; ++j) {
                const T cv = c[(size_t)j * R + r];
                const T* w = Wt + (size_t)j * win + ib;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (ib + q < win) acc[q] = fma(cv, w[q], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (ib + q < win && (!first || ib + q < net.nin)) {   // extra inputs (tvp, p) get no Jacobian column
                    T v = acc[q];
                    if (!first) v *= d1s[(size_t)((l - 1) * net.maxw + ib + q) * R + r];
                    cn[(size_t)(ib + q) * R + r] = v;
                }
            }
        }
        cur ^= 1;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_valu_kernel(NetDev net, WsOff o, int kind, T DT, int B, int H,
                                                        size_t Rcap, const T* __restrict__ Z,
                                                        const T* __restrict__ X0, T* __restrict__ g, int m,
                                                        int box, T* __restrict__ tiles, T* __restrict__ ws) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (size_t)B * H) return;
    const int b = (int)(r / H), t = (int)(r % H);
    const int nx = net.nx, nin = net.nin;
    const int n = net.gk.n;
    const size_t R = Rcap;
    const T* z = Z + (size_t)b * n;

    T* xi = ws + (size_t)o.xi * R;
    for (int d = 0; d < nin; ++d) xi[(size_t)d * R + r] = gather_input<T>(net.gk, z, X0, b, t, d);
    for (int j = 0; j < net.ne; ++j) xi[(size_t)(nin + j) * R + r] = ((const T*)net.extra)[((size_t)b * H + t) * net.ne + j];

    T* fout = ws + (size_t)o.fout * R;
    T* jst = ws + (size_t)o.jst * R;
    T* gout = g + (size_t)b * m + (size_t)t * nx;
    T* tile = tiles + ((size_t)b * H + t) * nx * nin;

    net_forward<T>(net, ws, o, R, r);
    for (int k = 0; k < nx; ++k) net_jacobian_row<T>(net, ws, o, R, r, k);

    if (kind != NEMPC_RK4) {
        for (int i = 0; i < nx; ++i) {
            const T xp = (t == 0) ? X0[(size_t)b * nx + i] : z[(t - 1) * nx + i];
            const T phi = (kind == NEMPC_DISCRET ? xp : T(0)) + fout[(size_t)i * R + r];
            gout[i] = phi - z[t * nx + i];
            for (int d = 0; d < nin; ++d) {
                T v = jst[(size_t)(i * nin + d) * R + r];
                if (kind == NEMPC_DISCRET && d == net.gk.xcur + i) v += T(1);
                tile[i * nin + d] = v;
            }
        }
    } else {
        T* kcur = ws + (size_t)o.kcur * R;
        T* dk = ws + (size_t)o.dk * R;
        T* acck = ws + (size_t)o.acck * R;
        T* accdk = ws + (size_t)o.accdk * R;
        T* dkn = ws + (size_t)o.dkn * R;
        for (int i = 0; i < nx; ++i) {
            const T kv = fout[(size_t)i * R + r];
            kcur[(size_t)i * R + r] = kv;
            acck[(size_t)i * R + r] = kv;
            for (int d = 0; d < nin; ++d) {
                const T jv = jst[(size_t)(i * nin + d) * R + r];
                dk[(size_t)(i * nin + d) * R + r] = jv;
                accdk[(size_t)(i * nin + d) * R + r] = jv;
            }
        }
        for (int s = 0; s < 3; ++s) {
            const T c = (s == 2 ? T(1) : T(0.5)) * DT;
            const T wgt = (s == 2 ? T(1) : T(2));
            for (int i = 0; i < nx; ++i) {
                const T xp = (t == 0) ? X0[(size_t)b * nx + i] : z[(t - 1) * nx + i];
                xi[(size_t)i * R + r] = xp + c * kcur[(size_t)i * R + r];
            }
            net_forward<T>(net, ws, o, R, r);
            for (int k = 0; k < nx; ++k) net_jacobian_row<T>(net, ws, o, R, r, k);
            for (int i = 0; i < nx; ++i) {
                for (int d = 0; d < nin; ++d) {
                    T v = T(0);
                    for (int e = 0; e < nx; ++e)
                        v = fma(jst[(size_t)(i * nin + e) * R + r], dk[(size_t)(e * nin + d) * R + r], v);
                    dkn[(size_t)(i * nin + d) * R + r] = jst[(size_t)(i * nin + d) * R + r] + c * v;
                }
            }
            for (int i = 0; i < nx; ++i) {
                const T kv = fout[(size_t)i * R + r];
                kcur[(size_t)i * R + r] = kv;
                acck[(size_t)i * R + r] += wgt * kv;
                for (int d = 0; d < nin; ++d) {
                    const T v = dkn[(size_t)(i * nin + d) * R + r];
                    dk[(size_t)(i * nin + d) * R + r] = v;
                    accdk[(size_t)(i * nin + d) * R + r] += wgt * v;
                }
            }
        }
        const T s6 = DT / T(6);
        for (int i = 0; i < nx; ++i) {
            const T xp = (t == 0) ? X0[(size_t)b * nx + i] : z[(t - 1) * nx + i];
            gout[i] = (xp + s6 * acck[(size_t)i * R + r]) - z[t * nx + i];
            for (int d = 0; d < nin; ++d) {
                T v = s6 * accdk[(size_t)(i * nin + d) * R + r];
                if (d == i) v += T(1);
                tile[i * nin + d] = v;
            }
        }
    }
    if (box)
        for (int i = 0; i < nx; ++i) g[(size_t)b * m + (size_t)H * nx + t * nx + i] = z[t * nx + i];
}

// Contracted network Hessian at the input currently held in ws.xi:
//     hout[p][q] = sum_k mult_k d2 f_k / dxi_p dxi_q  =  sum_l P_l^T diag(delta_l * s''(z_l)) P_l
// with P_l = W_l^T D_{l-1} the pre-activation tangents (forward mode, nin directions),
// delta_l = d(mult . f)/d a_l (one reverse sweep) and s'' = r2(a) d1(a) (activations.h; tanh: -2 a (1 - a^2)); the sum
// runs over the hidden layers and, when it has an activation of its own, the output layer.
// Value-equivalent to model/tensorflow.py:77-109 contracted as in optimizer/ipopt.py:79-80.
// mult[k * mstride], hout[(p*nin+q) * hstride] (full symmetric block, lower triangle mirrored).
template <typename T>
__device__ void net_hessian_contracted(const NetDev& net, T* ws, const WsOff& o, size_t R, size_t r, const T* mult,
                                       size_t mstride, T* hout, size_t hstride) {
    const int nx = net.nx, nin = net.nin, nl = net.nl;
    for (int p = 0; p < nin * nin; ++p) hout[(size_t)p * hstride] = T(0);
    const bool lin_out = net.act[nl - 1] == NEMPC_ACT_LINEAR;
    if (nl == 1 && lin_out) return;  // linear network: no curvature
    net_forward<T>(net, ws, o, R, r);

    // forward tangents: P_l[i][p] (pre-activation), stored per layer that has an activation; D_l = s'(z_l) P_l
    for (int l = 0; l < (lin_out ? nl - 1 : nl); ++l) {
        const T* W = (const T*)net.W[l];
        const int win = net.din[l], wout = net.dout[l];
        T* P = ws + (size_t)(o.P + l * net.maxw * nin) * R;      // (the output layer's block follows the hidden ones)
        const T* Pprev = (l == 0) ? nullptr : ws + (size_t)(o.P + (l - 1) * net.maxw * nin) * R;
        const T* d1prev = (l == 0) ? nullptr : ws + (size_t)(o.d1s + (l - 1) * net.maxw) * R;
        for (int p = 0; p < nin; ++p) {
            for (int j = 0; j < wout; ++j) {
                T acc = T(0);
                if (l == 0) {
                    acc = W[(size_t)p * wout + j];
                } else {
                    for (int i = 0; i < win; ++i)
                        acc = fma(W[(size_t)i * wout + j], d1prev[(size_t)i * R + r] * Pprev[(size_t)(i * nin + p) * R + r], acc);
                }
                P[(size_t)(j * nin + p) * R + r] = acc;
            }
        }
    }
    // reverse sweep for delta_l = d(mult . f)/d a_l, accumulate the block on the way down
    int cur = 0;
    T* cL = ws + (size_t)o.cL * R;          // d(mult . f)/d z_L: mult itself for a linear output layer
    if (lin_out) {
        for (int k = 0; k < nx; ++k) cL[(size_t)k * R + r] = mult[(size_t)k * mstride];
    } else {
        const T* P = ws + (size_t)(o.P + (nl - 1) * net.maxw * nin) * R;
        for (int k = 0; k < nx; ++k) {
            const T s1 = ws[(size_t)(o.d1s + (nl - 1) * net.maxw + k) * R + r];
            const T wgt = mult[(size_t)k * mstride] * ws[(size_t)(o.d2s + (nl - 1) * net.maxw + k) * R + r];
            for (int p = 0; p < nin; ++p) {
                const T pp = wgt * P[(size_t)(k * nin + p) * R + r];
                for (int q = 0; q <= p; ++q)
                    hout[(size_t)(p * nin + q) * hstride] =
                        fma(pp, P[(size_t)(k * nin + q) * R + r], hout[(size_t)(p * nin + q) * hstride]);
            }
            cL[(size_t)k * R + r] = mult[(size_t)k * mstride] * s1;
        }
    }
    if (nl > 1) {
        const T* WL = (const T*)net.W[nl - 1];
        const int w = net.din[nl - 1];
        T* c = ws + (size_t)(o.cot + cur * net.maxw) * R;
        for (int j = 0; j < w; ++j) {
            T acc = T(0);
            for (int k = 0; k < nx; ++k) acc = fma(WL[(size_t)j * nx + k], cL[(size_t)k * R + r], acc);
            c[(size_t)j * R + r] = acc;
        }
    }
    for (int l = nl - 2; l >= 0; --l) {
        const int wout = net.dout[l];
        const T* P = ws + (size_t)(o.P + l * net.maxw * nin) * R;
        T* c = ws + (size_t)(o.cot + cur * net.maxw) * R;  // delta_l (wrt a_l)
        for (int j = 0; j < wout; ++j) {
            const T s1 = ws[(size_t)(o.d1s + l * net.maxw + j) * R + r];
            const T wgt = c[(size_t)j * R + r] * ws[(size_t)(o.d2s + l * net.maxw + j) * R + r];
            for (int p = 0; p < nin; ++p) {
                const T pp = wgt * P[(size_t)(j * nin + p) * R + r];
                for (int q = 0; q <= p; ++q)
                    hout[(size_t)(p * nin + q) * hstride] =
                        fma(pp, P[(size_t)(j * nin + q) * R + r], hout[(size_t)(p * nin + q) * hstride]);
            }
            c[(size_t)j * R + r] *= s1;  // now cot wrt z_l
        }
        if (l > 0) {
            const T* Wt = (const T*)net.Wt[l];
            const int win = net.din[l];
            T* cn = ws + (size_t)(o.cot + (cur ^ 1) * net.maxw) * R;
            for (int i = 0; i < win; ++i) {
                T acc = T(0);
                for (int j = 0; j < wout; ++j) acc = fma(Wt[(size_t)j * win + i], c[(size_t)j * R + r], acc);
                cn[(size_t)i * R + r] = acc;
            }
            cur ^= 1;
        }
    }
    for (int p = 0; p < nin; ++p)
        for (int q = p + 1; q < nin; ++q) hout[(size_t)(p * nin + q) * hstride] = hout[(size_t)(q * nin + p) * hstride];
}

// Per-row Lagrangian block  Hblk = sum_k lam_k d2 Phi_k / d[x_{t-1}|u_t]^2.
//   DISCRET / UNITY: Phi = [x +] f, so d2Phi = d2f at xi.
//   RK4 (value-equivalent to rk4.py:181-285, which the reference hard-wires to nx+nu = 3): with stage inputs
//   xi_s = xi + c_s E k_{s-1}, chain Jacobians dk_s = J_s R_s, R_s = I + c_s [dk_{s-1}; 0] (rk4.py:246-263),
//       d2(lam.Phi) = sum_s R_s^T ( sum_i nu_s[i] Hf_i(xi_s) ) R_s,
//   stage multipliers by the adjoint recursion nu_3 = DT/6 lam, nu_{s-1} = DT/6 w_{s-1} lam + c_s J_s[:, :nx]^T nu_s.
template <typename T>
__global__ __launch_bounds__(256) void rowhess_valu_kernel(NetDev net, WsOff o, int kind, T DT, int B, int H,
                                                           size_t Rcap, const T* __restrict__ Z,
                                                           const T* __restrict__ X0, const T* __restrict__ lam, int m,
                                                           T* __restrict__ blocks, T* __restrict__ ws) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (size_t)B * H) return;
    const int b = (int)(r / H), t = (int)(r % H);
    const int nx = net.nx, nin = net.nin;
    const int n = net.gk.n;
    const size_t R = Rcap;
    const T* z = Z + (size_t)b * n;
    T* blk = blocks + ((size_t)b * H + t) * nin * nin;
    const T* lrow = lam + (size_t)b * m + (size_t)t * nx;

    T* xi = ws + (size_t)o.xi * R;
    for (int d = 0; d < nin; ++d) xi[(size_t)d * R + r] = gather_input<T>(net.gk, z, X0, b, t, d);
    for (int j = 0; j < net.ne; ++j) xi[(size_t)(nin + j) * R + r] = ((const T*)net.extra)[((size_t)b * H + t) * net.ne + j];
    if (kind != NEMPC_RK4) {
        net_hessian_contracted<T>(net, ws, o, R, r, lrow, 1, blk, 1);
        return;
    }

    T* fout = ws + (size_t)o.fout * R;
    T* jst = ws + (size_t)o.jst * R;
    T* dk = ws + (size_t)o.dk * R;
    T* dkn = ws + (size_t)o.dkn * R;
    T* sxin = ws + (size_t)o.rk_xin * R;
    T* sJ = ws + (size_t)o.rk_J * R;
    T* sdk = ws + (size_t)o.rk_dk * R;
    T* snu = ws + (size_t)o.rk_nu * R;
    const int jn = nx * nin;
    // ---- forward over the four stages, keeping xi_s, J_s and dk_{s-1}
    for (int s = 0; s < 4; ++s) {
        const T c = (s == 0) ? T(0) : ((s == 3) ? DT : T(0.5) * DT);
        if (s > 0)
            for (int i = 0; i < nx; ++i) {
                const T xp = (t == 0) ? X0[(size_t)b * nx + i] : z[(t - 1) * nx + i];
                xi[(size_t)i * R + r] = xp + c * fout[(size_t)i * R + r];   // fout still holds k_{s-1}
            }
        for (int d = 0; d < nin; ++d) sxin[(size_t)(s * nin + d) * R + r] = xi[(size_t)d * R + r];
        net_forward<T>(net, ws, o, R, r);
        for (int k = 0; k < nx; ++k) net_jacobian_row<T>(net, ws, o, R, r, k);
        for (int e = 0; e < jn; ++e) {
            sJ[(size_t)(s * jn + e) * R + r] = jst[(size_t)e * R + r];
            sdk[(size_t)(s * jn + e) * R + r] = (s == 0) ? T(0) : dk[(size_t)e * R + r];   // dk_{s-1}
        }
        if (s == 0) {
            for (int e = 0; e < jn; ++e) dk[(size_t)e * R + r] = jst[(size_t)e * R + r];
        } else {
            for (int i = 0; i < nx; ++i)
                for (int d = 0; d < nin; ++d) {
                    T v = T(0);
                    for (int e = 0; e < nx; ++e)
                        v = fma(jst[(size_t)(i * nin + e) * R + r], dk[(size_t)(e * nin + d) * R + r], v);
                    dkn[(size_t)(i * nin + d) * R + r] = jst[(size_t)(i * nin + d) * R + r] + c * v;
                }
            for (int e = 0; e < jn; ++e) dk[(size_t)e * R + r] = dkn[(size_t)e * R + r];
        }
    }
    // ---- adjoint recursion for the stage multipliers
    const T s6 = DT / T(6);
    for (int i = 0; i < nx; ++i) snu[(size_t)(3 * nx + i) * R + r] = s6 * lrow[i];
    for (int s = 3; s >= 1; --s) {
        const T c = (s == 3) ? DT : T(0.5) * DT;
        const T wprev = (s - 1 == 0) ? T(1) : T(2);
        for (int j = 0; j < nx; ++j) {
            T v = T(0);
            for (int i = 0; i < nx; ++i)
                v = fma(sJ[(size_t)(s * jn + i * nin + j) * R + r], snu[(size_t)(s * nx + i) * R + r], v);
            snu[(size_t)((s - 1) * nx + j) * R + r] = s6 * wprev * lrow[j] + c * v;
        }
    }
    // ---- blk = sum_s R_s^T Htilde_s R_s
    T* ht = ws + (size_t)o.htmp * R;
    T* ht2 = ws + (size_t)o.htmp2 * R;
    for (int p = 0; p < nin * nin; ++p) blk[p] = T(0);
    for (int s = 0; s < 4; ++s) {
        const T c = (s == 0) ? T(0) : ((s == 3) ? DT : T(0.5) * DT);
        for (int d = 0; d < nin; ++d) xi[(size_t)d * R + r] = sxin[(size_t)(s * nin + d) * R + r];
        net_hessian_contracted<T>(net, ws, o, R, r, snu + (size_t)(s * nx) * R + r, R, ht + r, R);
        if (s == 0) {
            for (int p = 0; p < nin * nin; ++p) blk[p] += ht[(size_t)p * R + r];
            continue;
        }
        // ht2 = Htilde R ;  R[a][q] = delta_aq + (a < nx ? c dk_{s-1}[a][q] : 0)
        for (int a = 0; a < nin; ++a)
            for (int q = 0; q < nin; ++q) {
                T v = ht[(size_t)(a * nin + q) * R + r];
                for (int e = 0; e < nx; ++e)
                    v = fma(ht[(size_t)(a * nin + e) * R + r], c * sdk[(size_t)(s * jn + e * nin + q) * R + r], v);
                ht2[(size_t)(a * nin + q) * R + r] = v;
            }
        for (int p = 0; p < nin; ++p)
            for (int q = 0; q < nin; ++q) {
                T v = ht2[(size_t)(p * nin + q) * R + r];
                for (int e = 0; e < nx; ++e)
                    v = fma(c * sdk[(size_t)(s * jn + e * nin + p) * R + r], ht2[(size_t)(e * nin + q) * R + r], v);
                blk[p * nin + q] += v;
            }
    }
    // exact symmetry (the two triangles are rounded differently by the congruence products)
    for (int p = 0; p < nin; ++p)
        for (int q = p + 1; q < nin; ++q) {
            const T v = T(0.5) * (blk[p * nin + q] + blk[q * nin + p]);
            blk[p * nin + q] = v;
            blk[q * nin + p] = v;
        }
}

NetDev make_netdev(const Handle& h) {
    NetDev nd{};
    nd.nl = h.nl; nd.nin = h.nin; nd.nx = h.cfg.nx; nd.nu = h.cfg.nu; nd.maxw = h.maxw;
    nd.ne = h.ne; nd.extra = h.d_extra;
    nd.gk = h.gather();
    for (int l = 0; l < h.nl; ++l) {
        nd.din[l] = h.din[l]; nd.dout[l] = h.dout[l]; nd.act[l] = h.act[l]; nd.actp[l] = h.actp[l];
        nd.W[l] = h.d_W[l]; nd.Wt[l] = h.d_Wt[l]; nd.b[l] = h.d_b[l];
    }
    return nd;
}

}  // namespace

size_t valu_workspace_elems(const Handle& h) {
    return (size_t)ws_offsets(h).total * (size_t)h.cfg.max_batch * (size_t)h.cfg.H;
}

// The generic kernels' scratch (nhid * maxw * (nin + 1) elements per row: GBs for wide networks).  Handles on the generic
// variant get it in nempc_create / nempc_reserve; a handle on the layered path never touches it except in the one
// fallback below (Lagrangian blocks of a single hidden layer under a non-linear output layer, NEMPC_LAYERED_HESS=0) and
// allocates it there, on first use.
int ensure_valu_ws(Handle& h) {
    if (h.d_valu_ws) return NEMPC_OK;
    h.valu_ws_elems = valu_workspace_elems(h);
    hipError_t e = hipMalloc(&h.d_valu_ws, h.valu_ws_elems * h.esz ? h.valu_ws_elems * h.esz : 16);
    if (e != hipSuccess) {
        h.d_valu_ws = nullptr;
        set_error(std::string("hipMalloc (generic kernel workspace): ") + hipGetErrorString(e));
        return NEMPC_ENOMEM;
    }
    return NEMPC_OK;
}

int launch_rows_valu(Handle& h, int B, const void* Z, const void* X0, void* g, void* tiles, hipStream_t s) {
    // networks outside the register-resident kernels' shapes: the rows come from the layer-at-a-time GEMM pipeline
    if (h.layered) return launch_rows_layered(h, B, Z, X0, g, tiles, s);
    h.last_row_kernel = 1;
    const size_t rows = (size_t)B * h.cfg.H;
    const size_t Rcap = (size_t)h.cfg.max_batch * h.cfg.H;
    const dim3 block(256), grid((unsigned)((rows + 255) / 256));
    NetDev nd = make_netdev(h);
    WsOff o = ws_offsets(h);
    if (h.cfg.dtype == NEMPC_F64)
        hipLaunchKernelGGL(rows_valu_kernel<double>, grid, block, 0, s, nd, o, h.cfg.integrator, h.cfg.DT, B,
                           h.cfg.H, Rcap, (const double*)Z, (const double*)X0, (double*)g, h.m, (int)h.box,
                           (double*)tiles, (double*)h.d_valu_ws);
    else
        hipLaunchKernelGGL(rows_valu_kernel<float>, grid, block, 0, s, nd, o, h.cfg.integrator, (float)h.cfg.DT, B,
                           h.cfg.H, Rcap, (const float*)Z, (const float*)X0, (float*)g, h.m, (int)h.box,
                           (float*)tiles, (float*)h.d_valu_ws);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

int launch_rowhess_valu(Handle& h, int B, const void* Z, const void* X0, const void* lambda, void* blocks,
                        hipStream_t s) {
    if (h.layered) {        // Discret / Unity blocks of wide / deep networks: the GEMM path (kernels_layered.hip)
        const int rc = launch_rowhess_layered(h, B, Z, X0, lambda, blocks, s);
        if (rc != NEMPC_EUNSUPPORTED) return rc;
    }
    if (int rc = ensure_valu_ws(h)) return rc;
    const size_t rows = (size_t)B * h.cfg.H;
    const size_t Rcap = (size_t)h.cfg.max_batch * h.cfg.H;
    const dim3 block(256), grid((unsigned)((rows + 255) / 256));
    NetDev nd = make_netdev(h);
    WsOff o = ws_offsets(h);
    h.last_hess_kernel = 1;
    if (h.cfg.dtype == NEMPC_F64)
        hipLaunchKernelGGL(rowhess_valu_kernel<double>, grid, block, 0, s, nd, o, h.cfg.integrator, h.cfg.DT, B, h.cfg.H,
                           Rcap, (const double*)Z, (const double*)X0, (const double*)lambda, h.m, (double*)blocks,
                           (double*)h.d_valu_ws);
    else
        hipLaunchKernelGGL(rowhess_valu_kernel<float>, grid, block, 0, s, nd, o, h.cfg.integrator, (float)h.cfg.DT, B,
                           h.cfg.H, Rcap, (const float*)Z, (const float*)X0, (const float*)lambda, h.m, (float*)blocks,
                           (float*)h.d_valu_ws);
    NEMPC_HIP(hipGetLastError());
    return NEMPC_OK;
}

}  // namespace nempc
