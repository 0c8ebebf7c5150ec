/*
 * CPU oracle in C (OpenMP) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Batched fp64 restatement of the hessian-free callbacks (f, grad f, g, dense jac g) of
 * pyNeuralEMPC for an MLP (per-layer activation from the family of nempc_oracle.py: linear, tanh, relu, sigmoid,
 * softplus, elu) under Discret / Unity / RK4, used only
 *   - by tests/ (cross-checked against oracle/nempc_oracle.py, which is pinned to the
 *     reference-generated golden vectors), and
 *   - as the `cpu_baseline` leg of bench.py (kind "port", all host cores).
 * The product (pyneuralempc_amd) never links or calls this file.
 *
 * Reference call sites restated: integrator/discret.py:13-58, unity.py:15-58, rk4.py:57-178,
 * optimizer/ipopt.py:20-52,88-96; network derivative = analytic chain rule standing in for
 * model/tensorflow.py:53-75 (see the header of nempc_oracle.py for how that piece is pinned).
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXL 8

typedef struct {
    int H, nx, nu, kind; /* 0 discret, 1 unity, 2 rk4 */
    double DT;
    int nl;
    int din[MAXL], dout[MAXL];
    const double* W[MAXL]; /* (in,out) row-major */
    const double* b[MAXL];
    const double *Q, *R, *xref, *uref, *cx, *cu; /* (nx,nx) (nu,nu) (H,nx) (H,nu) (H,nx) (H,nu) */
    int box;
    int act[MAXL]; /* per layer: 0 linear, 1 tanh, 2 relu, 3 sigmoid, 4 softplus, 5 elu, 6 leaky_relu, 7 selu, 8 swish, 9 gelu, 10 softsign, 11 mish, 12 exponential, 13 relu6 (nempc_oracle.py ACT_IDS) */
    double actp[MAXL]; /* alpha of elu / leaky_relu */
} oracle_problem;

#define SELU_LAMBDA 1.0507009873554804934193349852946
#define SELU_ALPHA 1.6732632423543772848170429916717

/* activation and its derivative written in terms of the layer's output a = s(z) (table in nempc_oracle.py) */
static double act_f(int code, double z, double par) {
    switch (code) {
        case 1: return tanh(z);
        case 2: return z < 0.0 ? 0.0 : z;
        case 3: return 1.0 / (1.0 + exp(-z));
        case 4: return (z > 0.0 ? z : 0.0) + log1p(exp(-fabs(z)));
        case 5: return z > 0.0 ? z : par * expm1(z);
        case 6: return z > 0.0 ? z : par * z;
        case 7: return SELU_LAMBDA * (z > 0.0 ? z : SELU_ALPHA * expm1(z));
        case 8: return z / (1.0 + exp(-z));
        case 9: return 0.5 * z * (1.0 + erf(z * 0.70710678118654752440));
        case 10: return z / (1.0 + fabs(z));
        case 11: return z * tanh((z > 0.0 ? z : 0.0) + log1p(exp(-fabs(z))));
        case 12: return exp(z);
        case 13: return z < 0.0 ? 0.0 : (z > 6.0 ? 6.0 : z);
        default: return z;
    }
}
/* swish and gelu are not monotone: their derivative is written from the pre-activation z */
static int act_zbased(int code) { return code >= 8 && code <= 13; }
static double act_d1z(int code, double z) {
    if (code == 8) {
        const double sg = 1.0 / (1.0 + exp(-z));
        return sg * (1.0 + z * (1.0 - sg));
    }
    if (code == 10) { const double r = 1.0 / (1.0 + fabs(z)); return r * r; }
    if (code == 11) {
        const double t = tanh((z > 0.0 ? z : 0.0) + log1p(exp(-fabs(z)))), g = 1.0 / (1.0 + exp(-z));
        return t + z * (1.0 - t * t) * g;
    }
    if (code == 12) return exp(z);
    if (code == 13) return (z > 0.0 && z < 6.0) ? 1.0 : 0.0;
    return 0.5 * (1.0 + erf(z * 0.70710678118654752440)) + z * 0.39894228040143267794 * exp(-0.5 * z * z);
}
static double act_d1(int code, double a, double par) {
    switch (code) {
        case 1: return 1.0 - a * a;
        case 2: return a > 0.0 ? 1.0 : 0.0;
        case 3: return a * (1.0 - a);
        case 4: return -expm1(-a);
        case 5: return a > 0.0 ? 1.0 : a + par;
        case 6: return a > 0.0 ? 1.0 : (a != a ? a : par);
        case 7: return a > 0.0 ? SELU_LAMBDA : a + SELU_LAMBDA * SELU_ALPHA;
        default: return 1.0;
    }
}

/* f (nx) and J (nx, nin) of the network at xi (nin); scratch holds activations + cotangents */
static void net_eval(const oracle_problem* p, const double* xi, double* f, double* J, double* act, double* cot, double* dact) {
    const int nl = p->nl, nin = p->nx + p->nu;
    int maxw = nin;
    for (int l = 0; l < nl; ++l) if (p->dout[l] > maxw) maxw = p->dout[l];
    /* forward: act[l] = output of layer l (post-activation), dact[l] = s'(z_l) */
    const double* in = xi;
    for (int l = 0; l < nl; ++l) {
        double* out = (l == nl - 1) ? f : act + (size_t)l * maxw;
        double* dv = dact + (size_t)l * maxw;
        const int wi = p->din[l], wo = p->dout[l];
        for (int j = 0; j < wo; ++j) out[j] = p->b[l][j];
        for (int i = 0; i < wi; ++i) {
            const double a = in[i];
            const double* w = p->W[l] + (size_t)i * wo;
            for (int j = 0; j < wo; ++j) out[j] += a * w[j];
        }
        for (int j = 0; j < wo; ++j) {
            const double z = out[j];
            if (p->act[l] != 0) out[j] = act_f(p->act[l], z, p->actp[l]);
            dv[j] = act_zbased(p->act[l]) ? act_d1z(p->act[l], z) : act_d1(p->act[l], out[j], p->actp[l]);
        }
        in = out;
    }
    /* reverse sweep per output */
    for (int k = 0; k < p->nx; ++k) {
        double* c = cot;
        double* cn = cot + maxw;
        const double dout_k = dact[(size_t)(nl - 1) * maxw + k];   /* output-layer activation */
        if (nl == 1) {
            for (int d = 0; d < nin; ++d) J[k * nin + d] = p->W[0][(size_t)d * p->dout[0] + k] * dout_k;
            continue;
        }
        {
            const int w = p->din[nl - 1];
            const double* da = dact + (size_t)(nl - 2) * maxw;
            for (int j = 0; j < w; ++j) c[j] = p->W[nl - 1][(size_t)j * p->nx + k] * dout_k * da[j];
        }
        for (int l = nl - 2; l >= 0; --l) {
            const int wi = p->din[l], wo = p->dout[l];
            double* dst = (l == 0) ? (J + k * nin) : cn;
            for (int i = 0; i < wi; ++i) {
                const double* w = p->W[l] + (size_t)i * wo;
                double s = 0.0;
                for (int j = 0; j < wo; ++j) s += w[j] * c[j];
                if (l > 0) s *= dact[(size_t)(l - 1) * maxw + i];
                dst[i] = s;
            }
            if (l > 0) { double* t = c; c = cn; cn = t; }
        }
    }
}

/* Phi (nx) and dPhi (nx,nin) for one row */
static void step_row(const oracle_problem* p, const double* xprev, const double* u, double* phi, double* dphi,
                     double* ws) {
    const int nx = p->nx, nu = p->nu, nin = nx + nu;
    int maxw = nin;
    for (int l = 0; l < p->nl; ++l) if (p->dout[l] > maxw) maxw = p->dout[l];
    double* xi = ws;                     /* nin */
    double* f = xi + nin;                /* nx */
    double* J = f + nx;                  /* nx*nin */
    double* k = J + nx * nin;            /* nx */
    double* dk = k + nx;                 /* nx*nin */
    double* acck = dk + nx * nin;        /* nx */
    double* accdk = acck + nx;           /* nx*nin */
    double* dkn = accdk + nx * nin;      /* nx*nin */
    double* act = dkn + nx * nin;        /* nl*maxw */
    double* cot = act + (size_t)p->nl * maxw; /* 2*maxw */
    double* dact = cot + 2 * (size_t)maxw;    /* nl*maxw */
    for (int i = 0; i < nx; ++i) xi[i] = xprev[i];
    for (int j = 0; j < nu; ++j) xi[nx + j] = u[j];
    net_eval(p, xi, f, J, act, cot, dact);
    if (p->kind != 2) {
        for (int i = 0; i < nx; ++i) {
            phi[i] = (p->kind == 0 ? xprev[i] : 0.0) + f[i];
            for (int d = 0; d < nin; ++d) dphi[i * nin + d] = J[i * nin + d] + ((p->kind == 0 && d == i) ? 1.0 : 0.0);
        }
        return;
    }
    memcpy(k, f, sizeof(double) * nx);
    memcpy(acck, f, sizeof(double) * nx);
    memcpy(dk, J, sizeof(double) * nx * nin);
    memcpy(accdk, J, sizeof(double) * nx * nin);
    for (int s = 0; s < 3; ++s) {
        const double c = (s == 2 ? 1.0 : 0.5) * p->DT, wgt = (s == 2 ? 1.0 : 2.0);
        for (int i = 0; i < nx; ++i) xi[i] = xprev[i] + c * k[i];
        net_eval(p, xi, f, J, act, cot, dact);
        for (int i = 0; i < nx; ++i)
            for (int d = 0; d < nin; ++d) {
                double v = 0.0;
                for (int e = 0; e < nx; ++e) v += J[i * nin + e] * dk[e * nin + d];
                dkn[i * nin + d] = J[i * nin + d] + c * v;
            }
        for (int i = 0; i < nx; ++i) { k[i] = f[i]; acck[i] += wgt * f[i]; }
        for (int e = 0; e < nx * nin; ++e) { dk[e] = dkn[e]; accdk[e] += wgt * dkn[e]; }
    }
    for (int i = 0; i < nx; ++i) {
        phi[i] = xprev[i] + p->DT / 6.0 * acck[i];
        for (int d = 0; d < nin; ++d) dphi[i * nin + d] = p->DT / 6.0 * accdk[i * nin + d] + (d == i ? 1.0 : 0.0);
    }
}

static size_t ws_doubles(const oracle_problem* p) {
    const int nx = p->nx, nin = p->nx + p->nu;
    int maxw = nin;
    for (int l = 0; l < p->nl; ++l) if (p->dout[l] > maxw) maxw = p->dout[l];
    return (size_t)nin + 3 * nx + 4 * (size_t)nx * nin + 2 * (size_t)p->nl * maxw + 2 * (size_t)maxw;
}

/* One batched evaluation.  Any output pointer may be NULL.  jac is dense (B,m,n). Returns threads used. */
int oracle_eval(const oracle_problem* p, int B, const double* Z, const double* X0, double* f, double* grad,
                double* g, double* jac, int nthreads) {
    const int H = p->H, nx = p->nx, nu = p->nu, nin = nx + nu, n = H * nin;
    const int m = H * nx + (p->box ? H * nx : 0);
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#endif
    const size_t wsz = ws_doubles(p);
#pragma omp parallel
    {
        double* ws = (double*)malloc(sizeof(double) * wsz);
        double* phi = (double*)malloc(sizeof(double) * nx);
        double* dphi = (double*)malloc(sizeof(double) * (size_t)nx * nin);
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            const double* z = Z + (size_t)b * n;
            if (f || grad) {
                double acc = 0.0;
                for (int t = 0; t < H; ++t) {
                    const double* x = z + t * nx;
                    const double* u = z + H * nx + t * nu;
                    for (int i = 0; i < nx; ++i) {
                        double qd = 0.0, qsd = 0.0;
                        for (int j = 0; j < nx; ++j) {
                            const double dxj = x[j] - p->xref[t * nx + j];
                            qd += p->Q[i * nx + j] * dxj;
                            qsd += (p->Q[i * nx + j] + p->Q[j * nx + i]) * dxj;
                        }
                        acc += (x[i] - p->xref[t * nx + i]) * qd + p->cx[t * nx + i] * x[i];
                        if (grad) grad[(size_t)b * n + t * nx + i] = qsd + p->cx[t * nx + i];
                    }
                    for (int i = 0; i < nu; ++i) {
                        double rd = 0.0, rsd = 0.0;
                        for (int j = 0; j < nu; ++j) {
                            const double duj = u[j] - p->uref[t * nu + j];
                            rd += p->R[i * nu + j] * duj;
                            rsd += (p->R[i * nu + j] + p->R[j * nu + i]) * duj;
                        }
                        acc += (u[i] - p->uref[t * nu + i]) * rd + p->cu[t * nu + i] * u[i];
                        if (grad) grad[(size_t)b * n + H * nx + t * nu + i] = rsd + p->cu[t * nu + i];
                    }
                }
                if (f) f[b] = acc;
            }
            if (g || jac) {
                double* Jb = jac ? jac + (size_t)b * m * n : NULL;
                if (Jb) memset(Jb, 0, sizeof(double) * (size_t)m * n);
                for (int t = 0; t < H; ++t) {
                    const double* xprev = (t == 0) ? X0 + (size_t)b * nx : z + (t - 1) * nx;
                    step_row(p, xprev, z + H * nx + t * nu, phi, dphi, ws);
                    for (int i = 0; i < nx; ++i) {
                        const int r = t * nx + i;
                        if (g) g[(size_t)b * m + r] = phi[i] - z[t * nx + i];
                        if (Jb) {
                            double* row = Jb + (size_t)r * n;
                            row[t * nx + i] -= 1.0;
                            if (t > 0) for (int j = 0; j < nx; ++j) row[(t - 1) * nx + j] += dphi[i * nin + j];
                            for (int j = 0; j < nu; ++j) row[H * nx + t * nu + j] += dphi[i * nin + nx + j];
                        }
                    }
                }
                if (p->box) {
                    for (int k2 = 0; k2 < H * nx; ++k2) {
                        if (g) g[(size_t)b * m + H * nx + k2] = z[k2];
                        if (Jb) Jb[(size_t)(H * nx + k2) * n + k2] = 1.0;
                    }
                }
            }
        }
        free(ws);
        free(phi);
        free(dphi);
    }
    return used;
}
