// tanh variants for the row kernels: issue cost on the vector pipe (which v_mfma_f64 blocks entirely, see
// tools/ubench_dpops.hip) and accuracy against host tanhl.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_tanh.hip -o tools/_bin/ubench_tanh && tools/_bin/ubench_tanh
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// round-1/2 form: degree-13 Taylor, rint + cvt, two Newton steps, NaN select at the end
__device__ __forceinline__ double tanh_v1(double x) {
    const double ax = fmin(fabs(x), 20.0);
    const double y = ax + ax;
    const double n = rint(y * 1.4426950408889634);
    double r = fma(-n, 6.93147180369123816490e-01, y);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double d = ldexp(p, (int)n) + 1.0;
    double q = __builtin_amdgcn_rcp(d);
    q = fma(fma(-d, q, 1.0), q, q);
    q = fma(fma(-d, q, 1.0), q, q);
    return x != x ? x : copysign(fma(-2.0, q, 1.0), x);
}

// lean form.  Every vector instruction costs the same 4 issue cycles on gfx950 (16 for v_rcp_f64) and v_mfma_f64 holds
// the vector pipe for its whole 64, so the count is what matters:
//  * |x| is clamped on its HIGH dword only (compare + one select; a NaN fails the compare and flows through to the
//    result, so no NaN select at the end; the low dword of a clamped value is immaterial, tanh(20) rounds to 1)
//  * n = rint(|x| * 2/ln2) by the 1.5*2^52 shift: one fma, one subtract, and the integer n is the low dword of the
//    shifted value (no v_rndne, no v_cvt_i32_f64)
//  * exp(2s) on |s| <= ln2/4 by a degree-DEG near-minimax polynomial (Chebyshev fit; tools/ubench_tanh.hip prints the error)
//  * 1/d: v_rcp_f64 + one cubic (Halley) step, three fmas instead of two Newton steps' four
template <int DEG>
__device__ __forceinline__ double tanh_lean(double x) {
    const int hx = __double2hiint(x);
    const double a = __hiloint2double(fabs(x) > 20.0 ? 0x40340000 : hx, __double2loint(x));      // |a| in [0, 20]
    const double SHIFT = 6755399441055744.0;                                   // 1.5 * 2^52
    const double t = fma(fabs(a), 2.8853900817779268, SHIFT);                  // 2/ln2
    const double nf = t - SHIFT;
    const double s = fma(-nf, 0.34657359027997264, fabs(a));      // ln2/2 in one piece: its rounding error matters only where tanh is flat
    double p;
    if (DEG == 11) {
        p = 5.1425357017013815e-05;
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
    } else {
        p = 0.00028289389815241956;
        p = fma(p, s, 0.0014151772567659443);
        p = fma(p, s, 0.006349185112831408);
        p = fma(p, s, 0.025396697946163286);
        p = fma(p, s, 0.08888888916792703);
        p = fma(p, s, 0.26666666834136904);
        p = fma(p, s, 0.6666666666651703);
        p = fma(p, s, 1.3333333333243524);
        p = fma(p, s, 2.000000000000002);
        p = fma(p, s, 2.0000000000000133);
        p = fma(p, s, 1.0);
    }
    const double d = ldexp(p, __double2loint(t)) + 1.0;
    double q = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, q, 1.0);
    q = fma(fma(e, e, e), q, q);
    return copysign(fma(-2.0, q, 1.0), x);
}

template <int MODE>
__global__ void tanh_rate(const double* in, double* out, int iters) {
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[(blockIdx.x * blockDim.x + threadIdx.x) * 8 + i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) v[i] = tanh_v1(v[i]) * 1.7;
            else if (MODE == 1) v[i] = tanh_lean<11>(v[i]) * 1.7;
            else v[i] = tanh_lean<10>(v[i]) * 1.7;
        }
    }
    for (int i = 0; i < 8; ++i) out[(blockIdx.x * blockDim.x + threadIdx.x) * 8 + i] = v[i];
}

__global__ void tanh_acc(const double* in, double* o0, double* o1, double* o2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { o0[i] = tanh_v1(in[i]); o1[i] = tanh_lean<11>(in[i]); o2[i] = tanh_lean<10>(in[i]); }
}

int main() {
    const int blocks = 256;
    const int n = 1 << 22;
    std::vector<double> hin(n);
    srand(1);
    for (int i = 0; i < n; ++i) hin[i] = ((double)rand() / RAND_MAX * 2 - 1) * ((i % 7 == 0) ? 30.0 : (i % 7 == 1 ? 0.3 : 3.0));
    const double specials[] = {0.0, -0.0, 1e-9, -1e-9, 25.0, -700.0, 1e-300, 20.0, -20.0, 19.999999, 1e300, -1e300,
                               INFINITY, -INFINITY, NAN, 0.17328679513998632, 0.1732867951399864, 5e-324};
    const int nsp = sizeof(specials) / sizeof(double);
    for (int i = 0; i < nsp; ++i) hin[i] = specials[i];
    double *din, *d0, *d1, *d2;
    CK(hipMalloc(&din, sizeof(double) * n)); CK(hipMalloc(&d0, sizeof(double) * n));
    CK(hipMalloc(&d1, sizeof(double) * n)); CK(hipMalloc(&d2, sizeof(double) * n));
    CK(hipMemcpy(din, hin.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    const char* names[3] = {"degree-13 Taylor, rint/cvt, 2 Newton", "lean, degree 11", "lean, degree 10"};
    for (int wps : {2, 8}) {
        const int threads = wps >= 4 ? 1024 : wps * 256, nb = blocks * (wps >= 4 ? wps / 4 : 1);
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int iters = 2000;
            auto launch = [&](int it) {
                if (mode == 0) hipLaunchKernelGGL(tanh_rate<0>, dim3(nb), dim3(threads), 0, 0, din, d0, it);
                if (mode == 1) hipLaunchKernelGGL(tanh_rate<1>, dim3(nb), dim3(threads), 0, 0, din, d0, it);
                if (mode == 2) hipLaunchKernelGGL(tanh_rate<2>, dim3(nb), dim3(threads), 0, 0, din, d0, it);
            };
            launch(10);
            CK(hipEventRecord(e0));
            launch(iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-40s waves/SIMD=%d  ns of a SIMD per tanh wave-instruction %.1f\n", names[mode], wps,
                   ms * 1e6 / ((double)iters * 8.0 * wps));
        }
    }
    hipLaunchKernelGGL(tanh_acc, dim3((n + 255) / 256), dim3(256), 0, 0, din, d0, d1, d2, n);
    std::vector<double> h0(n), h1(n), h2(n);
    CK(hipMemcpy(h0.data(), d0, sizeof(double) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), d1, sizeof(double) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), d2, sizeof(double) * n, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        if (hin[i] != hin[i]) continue;
        const long double t = tanhl((long double)hin[i]);
        e0 = fmax(e0, (double)fabsl(h0[i] - t)); e1 = fmax(e1, (double)fabsl(h1[i] - t)); e2 = fmax(e2, (double)fabsl(h2[i] - t));
    }
    printf("max abs err vs tanhl over %d points: v1 %.3e, lean-11 %.3e, lean-10 %.3e\n", n, e0, e1, e2);
    for (int i = 0; i < nsp; ++i)
        printf("  x = %-24.17g v1 %-24.17g lean-11 %-24.17g lean-10 %-24.17g libm %.17g\n", hin[i], h0[i], h1[i], h2[i], std::tanh(hin[i]));
    return 0;
}
