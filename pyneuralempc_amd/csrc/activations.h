// Activation family of the row / Hessian kernels (nempc_config.activations, NEMPC_ACT_*).
//
// The reference wraps any feed-forward Keras model (model/tensorflow.py:8-29,49-51) and lets TensorFlow differentiate
// it (tensorflow.py:53-109).  Here every derivative is written in terms of the layer's OUTPUT a = s(z) -- the one value
// the kernels keep from the forward pass (the tanh kernels have always used 1 - a^2):
//     s'(z) = d1(a),        s''(z) = r2(a) * d1(a)
//   linear    a = z                      d1 = 1                       r2 = 0
//   tanh      a = tanh z                 d1 = 1 - a^2                 r2 = -2a
//   relu      a = max(z, 0)              d1 = [a > 0]                 r2 = 0        (TensorFlow's relu gradient at 0 is 0)
//   sigmoid   a = 1 / (1 + e^-z)         d1 = a (1 - a)               r2 = 1 - 2a
//   softplus  a = log(1 + e^z)           d1 = 1 - e^-a (= sigmoid z)  r2 = e^-a
//   elu       a = z (z > 0), al(e^z - 1) d1 = 1 (a > 0), a + al       r2 = 0 (a > 0), 1          (al = alpha > 0; Keras' default 1)
//   leaky_relu a = z (z > 0), al z       d1 = 1 (a > 0), al           r2 = 0                     (al >= 0; run-time forms only)
//   selu      a = la z, la al(e^z - 1)   d1 = la (a > 0), a + la al   r2 = 0 (a > 0), 1          (Keras' constants; run-time forms only)
// r2 is what the forward-over-reverse Hessian sweeps need: d(delta * s') = s' d(delta) + delta * r2 * da, da the tangent
// of the activation itself.
//
// Two forms: Act<T, ACT> for the matrix-core kernels (the activation is a template parameter: one instantiation per
// activation, the tanh one unchanged), act_* (code, x) for the generic thread-per-row kernel (per-layer codes at run
// time, any mix, also on the output layer).
#pragma once

#include <hip/hip_runtime.h>

#include "nempc.h"

namespace nempc {

// tanh for the row kernels: t = 1 - 2 / (exp(2|x|) + 1), sign restored.  ocml's tanh(double) costs ~670 cycles per
// wave-instruction on gfx950.  On this chip every vector instruction -- double, single or integer -- takes the same 4
// issue cycles (v_rcp_f64: 16) and a v_mfma_f64 holds the vector pipe for all of its 64 (tools/ubench_dpops.hip: MFMA
// waves and vector waves on one SIMD serialise whatever the vector instruction is), so what a tanh costs the matrix
// kernels is its instruction COUNT.  24 issue slots here against 33 for the straightforward form (degree-13 Taylor,
// v_rndne + v_cvt_i32, two Newton steps, NaN select), same 2.2e-16 max abs error against tanhl on [-30, 30]
// (tools/ubench_tanh.hip; absolute accuracy is what the 1 - a^2 derivative factors need):
//  * |x| is clamped at 20 (tanh(20) rounds to 1) on its HIGH dword only: one compare, one select.  A NaN fails the
//    compare and flows through every later operation, so a diverged iterate stays visible without a select at the end
//  * n = rint(|x| * 2/ln2) by the 1.5*2^52 shift: one fma and one subtract, and the integer n is the low dword of
//    the shifted value (no v_rndne_f64, no v_cvt_i32_f64)
//  * s = |x| - n ln2/2 with ln2/2 as ONE double: its rounding error (1.9e-17) times n is an error of 2.2 n 1.9e-17 in
//    exp(2s), which reaches tanh scaled by 2e/(e+1)^2 ~ 2^(1-n): at most 2.4e-17 absolute (n = 2), so the second
//    Cody-Waite step buys nothing here
//  * exp(2s) on |s| <= ln2/4 by a degree-11 Chebyshev fit (relative error 1.7e-17 with the rounded coefficients)
//  * 1/d from v_rcp_f64 and ONE cubic step (three fmas)
__device__ __forceinline__ double nempc_tanh(double x) {
    const double a = __hiloint2double(fabs(x) > 20.0 ? 0x40340000 : __double2hiint(x), __double2loint(x));
    const double SHIFT = 6755399441055744.0;
    const double t = fma(fabs(a), 2.8853900817779268, SHIFT);
    const double nf = t - SHIFT;
    const double s = fma(-nf, 0.34657359027997264, fabs(a));      // ln2/2 in one piece, see above
    double p = 5.1425357017013815e-05;
    p = fma(p, s, 0.00028295822990378013);
    p = fma(p, s, 0.0014109307350312432);
    p = fma(p, s, 0.0063491802834760944);
    p = fma(p, s, 0.025396825459260305);
    p = fma(p, s, 0.08888888929481456);
    p = fma(p, s, 0.26666666666622724);
    p = fma(p, s, 0.6666666666638096);
    p = fma(p, s, 1.3333333333333344);
    p = fma(p, s, 2.0000000000000075);
    p = fma(p, s, 2.0);
    p = fma(p, s, 1.0);
    const double d = ldexp(p, __double2loint(t)) + 1.0;
    double q = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, q, 1.0);
    q = fma(fma(e, e, e), q, q);
    return copysign(fma(-2.0, q, 1.0), x);
}

// fp32: hardware exp2 / rcp; abs error ~1e-7, inside the fp32 configs' 1e-4 tolerance
__device__ __forceinline__ float nempc_tanh(float x) {
    const float ax = fminf(fabsf(x), 10.0f);
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);  // exp(2|x|)
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    return x != x ? x : copysignf(t, x);
}

// exp / expm1 / log1p of the other activations.  fp64: lean forms built like nempc_tanh (round 5) -- the library routines
// cost 35 .. 90 issue slots each and, inlined beside 150 - 250 live weight and activation registers, pushed the fp64
// sigmoid / softplus / elu kernels into scratch (up to 756 bytes per lane in the run-time-activation unit); what the kernels
// need is ABSOLUTE accuracy at the 1e-16 level (parity bar 1e-12 against NumPy's), NaN flowing through, 0 / inf at the ends:
//  * exp(x): x clamped to [-1024, 1024] on its high dword, n = rint(x / ln2) by the 1.5 * 2^52 shift, s = x / 2 - n ln2 / 2,
//    exp(2 s) by nempc_tanh's degree-11 fit, ldexp.  Relative error 2e-16 for |x| < 3, growing by 4e-17 per unit of |n|
//    (the one-piece ln2 / 2): 3e-15 where the value is 1e-12, i.e. absolute error below 2.3e-16 everywhere on x <= 0
//  * expm1(x): the same polynomial without its constant term IS expm1(2 s) to full relative accuracy; n == 0 returns it
//  * log1p(t), 0 <= t <= 1 (its only arguments: e^-|x|): 2 atanh(t / (2 + t)), the odd series to u^33 (u <= 1/3)
// fp32: the hardware exp2 / log2 (branch-free; absolute error ~1e-7, inside the fp32 configs' 1e-4 tolerance -- what
// nempc_tanh(float) does too).  tools/ubench_explog.hip measures all three against the host's long double routines.
__device__ __forceinline__ double nempc_exp2s_m1(double s) {          // exp(2 s) - 1, |s| <= ln2 / 4
    double p = 5.1425357017013815e-05;
    p = fma(p, s, 0.00028295822990378013);
    p = fma(p, s, 0.0014109307350312432);
    p = fma(p, s, 0.0063491802834760944);
    p = fma(p, s, 0.025396825459260305);
    p = fma(p, s, 0.08888888929481456);
    p = fma(p, s, 0.26666666666622724);
    p = fma(p, s, 0.6666666666638096);
    p = fma(p, s, 1.3333333333333344);
    p = fma(p, s, 2.0000000000000075);
    p = fma(p, s, 2.0);
    return p * s;
}
__device__ __forceinline__ void nempc_exp_reduce(double x, double& s, int& n) {
    const int hx = __double2hiint(x);
    const double xc = __hiloint2double(fabs(x) > 1024.0 ? ((hx & (int)0x80000000) | 0x40900000) : hx, __double2loint(x));
    const double SHIFT = 6755399441055744.0;
    const double t = fma(xc, 1.4426950408889634, SHIFT);                // x / ln2
    const double nf = t - SHIFT;
    s = fma(-nf, 0.34657359027997264, 0.5 * xc);
    n = __double2loint(t);
}
__device__ __forceinline__ double nempc_exp(double x) {
    double s;
    int n;
    nempc_exp_reduce(x, s, n);
    return ldexp(nempc_exp2s_m1(s) + 1.0, n);
}
__device__ __forceinline__ float nempc_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ double nempc_expm1(double x) {
    double s;
    int n;
    nempc_exp_reduce(x, s, n);
    const double q = nempc_exp2s_m1(s);
    return n == 0 ? q : ldexp(q + 1.0, n) - 1.0;
}
__device__ __forceinline__ float nempc_expm1(float x) { return nempc_exp(x) - 1.0f; }
__device__ __forceinline__ double nempc_log1p(double t) {             // 0 <= t <= 1
    const double d = 2.0 + t;
    double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    r = fma(fma(e, e, e), r, r);
    const double u = t * r, w = u * u;
    double p = 1.0 / 33.0;
    p = fma(p, w, 1.0 / 31.0);
    p = fma(p, w, 1.0 / 29.0);
    p = fma(p, w, 1.0 / 27.0);
    p = fma(p, w, 1.0 / 25.0);
    p = fma(p, w, 1.0 / 23.0);
    p = fma(p, w, 1.0 / 21.0);
    p = fma(p, w, 1.0 / 19.0);
    p = fma(p, w, 1.0 / 17.0);
    p = fma(p, w, 1.0 / 15.0);
    p = fma(p, w, 1.0 / 13.0);
    p = fma(p, w, 1.0 / 11.0);
    p = fma(p, w, 1.0 / 9.0);
    p = fma(p, w, 1.0 / 7.0);
    p = fma(p, w, 1.0 / 5.0);
    p = fma(p, w, 1.0 / 3.0);
    p = fma(p, w, 1.0);
    return (u + u) * p;
}
__device__ __forceinline__ float nempc_log1p(float x) { return __builtin_amdgcn_logf(1.0f + x) * 0.6931471805599453f; }

// sigmoid: fp64 as 1/2 + tanh(x / 2) / 2 on the kernels' own tanh (24 issue slots, absolute error 1.1e-16; a division alone is
// 20); fp32 1 / (1 + e^-x) (e^-x = inf gives 0)
__device__ __forceinline__ double nempc_sigmoid(double x) { return fma(0.5, nempc_tanh(0.5 * x), 0.5); }
__device__ __forceinline__ float nempc_sigmoid(float x) { return 1.0f / (1.0f + nempc_exp(-x)); }

template <typename T, int ACT>
struct Act;

template <typename T>
struct Act<T, NEMPC_ACT_LINEAR> {
    static __device__ __forceinline__ T f(T x) { return x; }
    static __device__ __forceinline__ T d1(T) { return T(1); }
    static __device__ __forceinline__ T r2(T) { return T(0); }
};
template <typename T>
struct Act<T, NEMPC_ACT_TANH> {
    static __device__ __forceinline__ T f(T x) { return nempc_tanh(x); }
    static __device__ __forceinline__ T d1(T a) { return fma(-a, a, T(1)); }      // one instruction (1 - a*a is two)
    static __device__ __forceinline__ T r2(T a) { return T(-2) * a; }
};
template <typename T>
struct Act<T, NEMPC_ACT_RELU> {
    static __device__ __forceinline__ T f(T x) { return x < T(0) ? T(0) : x; }       // (a NaN fails the compare and stays)
    static __device__ __forceinline__ T d1(T a) { return a > T(0) ? T(1) : T(0); }
    static __device__ __forceinline__ T r2(T) { return T(0); }
};
template <typename T>
struct Act<T, NEMPC_ACT_SIGMOID> {
    static __device__ __forceinline__ T f(T x) { return nempc_sigmoid(x); }
    static __device__ __forceinline__ T d1(T a) { return a * (T(1) - a); }
    static __device__ __forceinline__ T r2(T a) { return T(1) - T(2) * a; }
};
template <typename T>
struct Act<T, NEMPC_ACT_SOFTPLUS> {
    static __device__ __forceinline__ T f(T x) { return (x > T(0) ? x : (x != x ? x : T(0))) + nempc_log1p(nempc_exp(-fabs(x))); }
    static __device__ __forceinline__ T d1(T a) { return -nempc_expm1(-a); }
    static __device__ __forceinline__ T r2(T a) { return nempc_exp(-a); }
};
template <typename T>
struct Act<T, NEMPC_ACT_ELU> {
    static __device__ __forceinline__ T f(T x) { return x > T(0) ? x : nempc_expm1(x); }   // expm1(NaN) = NaN
    static __device__ __forceinline__ T d1(T a) { return a > T(0) ? T(1) : a + T(1); }
    static __device__ __forceinline__ T r2(T a) { return a > T(0) ? T(0) : (a != a ? a : T(1)); }
};

// run-time forms (generic kernel and the layered path: the code is wave-uniform, the switch a scalar branch).  `par` is the
// layer's nempc_config.act_param: alpha of elu / leaky_relu.
#define NEMPC_SELU_LAMBDA 1.0507009873554804934193349852946
#define NEMPC_SELU_ALPHA 1.6732632423543772848170429916717
template <typename T>
__device__ __forceinline__ T act_f(int code, T x, T par) {
    switch (code) {
        case NEMPC_ACT_TANH: return sizeof(T) == 8 ? (T)tanh((double)x) : (T)tanhf((float)x);
        case NEMPC_ACT_RELU: return Act<T, NEMPC_ACT_RELU>::f(x);
        case NEMPC_ACT_SIGMOID: return Act<T, NEMPC_ACT_SIGMOID>::f(x);
        case NEMPC_ACT_SOFTPLUS: return Act<T, NEMPC_ACT_SOFTPLUS>::f(x);
        case NEMPC_ACT_ELU: return x > T(0) ? x : par * nempc_expm1(x);                  // expm1(NaN) = NaN
        case NEMPC_ACT_LEAKY_RELU: return x > T(0) ? x : par * x;
        case NEMPC_ACT_SELU: return T(NEMPC_SELU_LAMBDA) * (x > T(0) ? x : T(NEMPC_SELU_ALPHA) * nempc_expm1(x));
        default: return x;
    }
}
template <typename T>
__device__ __forceinline__ T act_d1(int code, T a, T par) {
    switch (code) {
        case NEMPC_ACT_TANH: return Act<T, NEMPC_ACT_TANH>::d1(a);
        case NEMPC_ACT_RELU: return Act<T, NEMPC_ACT_RELU>::d1(a);
        case NEMPC_ACT_SIGMOID: return Act<T, NEMPC_ACT_SIGMOID>::d1(a);
        case NEMPC_ACT_SOFTPLUS: return Act<T, NEMPC_ACT_SOFTPLUS>::d1(a);
        case NEMPC_ACT_ELU: return a > T(0) ? T(1) : a + par;
        case NEMPC_ACT_LEAKY_RELU: return a > T(0) ? T(1) : (a != a ? a : par);          // (tf.nn.leaky_relu: alpha at 0)
        case NEMPC_ACT_SELU: return a > T(0) ? T(NEMPC_SELU_LAMBDA) : a + T(NEMPC_SELU_LAMBDA * NEMPC_SELU_ALPHA);
        default: return T(1);
    }
}
template <typename T>
__device__ __forceinline__ T act_r2(int code, T a, T par) {
    (void)par;
    switch (code) {
        case NEMPC_ACT_TANH: return Act<T, NEMPC_ACT_TANH>::r2(a);
        case NEMPC_ACT_SIGMOID: return Act<T, NEMPC_ACT_SIGMOID>::r2(a);
        case NEMPC_ACT_SOFTPLUS: return Act<T, NEMPC_ACT_SOFTPLUS>::r2(a);
        case NEMPC_ACT_ELU:
        case NEMPC_ACT_SELU: return a > T(0) ? T(0) : (a != a ? a : T(1));
        default: return T(0);
    }
}

// ---- per-layer activations of the register-resident matrix-core kernels.  ACT = an NEMPC_ACT_* code: the activation is
// the template parameter (one translation unit per activation, the layer index is ignored).  ACT = NEMPC_ACT_RUNTIME: the
// hidden layers' codes and parameters arrive in the launch arguments (ActSpec) -- any mix of the output-based family
// (linear ... selu), elu / leaky_relu with any alpha; the code is wave-uniform, the switch a scalar branch beside the
// layer's block of matrix instructions.  (The reference evaluates whatever per-layer activations the Keras model has:
// model/tensorflow.py:49-51.)
#define NEMPC_ACT_RUNTIME 100
#define NEMPC_MFMA_MAX_HIDDEN 4
struct ActSpec {
    int32_t code[NEMPC_MFMA_MAX_HIDDEN];
    double par[NEMPC_MFMA_MAX_HIDDEN];
};
template <typename T, int ACT>
struct ActL {
    static __device__ __forceinline__ T f(T x, const ActSpec& s, int l) {
        if constexpr (ACT == NEMPC_ACT_RUNTIME) {
            const int c = s.code[l];
            if (c == NEMPC_ACT_TANH) return nempc_tanh(x);            // (the kernels' own tanh, as in the tanh instantiations)
            return act_f<T>(c, x, (T)s.par[l]);
        } else {
            return Act<T, ACT>::f(x);
        }
    }
    static __device__ __forceinline__ T d1(T a, const ActSpec& s, int l) {
        if constexpr (ACT == NEMPC_ACT_RUNTIME) return act_d1<T>(s.code[l], a, (T)s.par[l]);
        else return Act<T, ACT>::d1(a);
    }
    static __device__ __forceinline__ T r2(T a, const ActSpec& s, int l) {
        if constexpr (ACT == NEMPC_ACT_RUNTIME) return act_r2<T>(s.code[l], a, (T)s.par[l]);
        else return Act<T, ACT>::r2(a);
    }
};

// swish, gelu, softsign, mish, exponential, relu6: value and derivatives from the PRE-activation z (the layered path's GEMM
// epilogue has z = acc + bias in registers).  swish: s = z g, s' = g (1 + z (1 - g)), s'' = g (1 - g) (2 + z (1 - 2 g)) with g = sigmoid(z);
// gelu: s = z Phi, s' = Phi + z phi, s'' = phi (2 - z^2) with Phi / phi the standard normal cdf / pdf.
// piecewise linear (s'' == 0 away from the kinks): no second-order constraint violation along such a network
__device__ __host__ __forceinline__ bool act_is_piecewise_linear(int code) {
    return code == NEMPC_ACT_RELU || code == NEMPC_ACT_LEAKY_RELU || code == NEMPC_ACT_RELU6;
}
__device__ __host__ __forceinline__ bool act_zbased(int code) { return code >= NEMPC_ACT_FIRST_ZBASED && code < NEMPC_ACT_COUNT; }
__device__ __forceinline__ double nempc_erf(double x) { return erf(x); }
__device__ __forceinline__ float nempc_erf(float x) { return erff(x); }
template <typename T>
__device__ __forceinline__ void act_from_z(int code, T z, T& a, T& d1, T& d2) {
    if (code == NEMPC_ACT_SWISH) {
        const T g = T(1) / (T(1) + nempc_exp(-z));
        a = z * g;
        d1 = g * (T(1) + z * (T(1) - g));
        d2 = g * (T(1) - g) * (T(2) + z * (T(1) - T(2) * g));
    } else if (code == NEMPC_ACT_GELU) {
        const T Phi = T(0.5) * (T(1) + nempc_erf(z * T(0.70710678118654752440)));
        const T phi = T(0.39894228040143267794) * nempc_exp(T(-0.5) * z * z);
        a = z * Phi;
        d1 = Phi + z * phi;
        d2 = phi * (T(2) - z * z);
    } else if (code == NEMPC_ACT_SOFTSIGN) {             // z / (1 + |z|);  1 / (1 + |z|)^2;  -2 sign(z) / (1 + |z|)^3
        const T r = T(1) / (T(1) + fabs(z));
        a = z * r;
        d1 = r * r;
        d2 = (z < T(0) ? T(2) : (z > T(0) ? T(-2) : z * T(0))) * r * r * r;      // (a NaN stays a NaN)
    } else if (code == NEMPC_ACT_MISH) {                 // z t, t = tanh(softplus z), g = sigmoid z
        const T sp = (z > T(0) ? z : (z != z ? z : T(0))) + nempc_log1p(nempc_exp(-fabs(z)));
        const T t = sizeof(T) == 8 ? (T)tanh((double)sp) : (T)tanhf((float)sp);
        const T g = T(1) / (T(1) + nempc_exp(-z));
        const T u = (T(1) - t * t) * g;                  // dt/dz
        a = z * t;
        d1 = t + z * u;
        d2 = T(2) * u + z * u * ((T(1) - g) - T(2) * t * g);
    } else if (code == NEMPC_ACT_EXPONENTIAL) {
        a = d1 = d2 = nempc_exp(z);
    } else {                                              // relu6: TensorFlow's gradient, 0 at both kinks
        a = z < T(0) ? T(0) : (z > T(6) ? T(6) : z);
        d1 = (z > T(0) && z < T(6)) ? T(1) : (z != z ? z : T(0));
        d2 = z != z ? z : T(0);
    }
}

}  // namespace nempc
