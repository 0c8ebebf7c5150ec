// Micro-benchmarks behind DESIGN.md's per-instruction numbers (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o /tmp/ubench && /tmp/ubench
// 1. v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 issue rate per SIMD (cycles per instruction)
// 2. cost of tanh: ocml tanh(double) vs the exp-based form used by the row kernel, with accuracy
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <int NACC>
__global__ void mfma_f64_rate(double* out, long long* cyc, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void mfma_f32_rate(float* out, long long* cyc, int iters) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// exp-based tanh: t = 1 - 2 / (exp(2|x|) + 1), sign restored
__device__ __forceinline__ double fast_exp_pos(double y) {
    // y in [0, 40]; n = round(y / ln2), r = y - n ln2 in [-ln2/2, ln2/2]
    const double L2E = 1.4426950408889634, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double n = rint(y * L2E);
    double r = fma(-n, LN2_HI, y);
    r = fma(-n, LN2_LO, r);
    // degree-13 Taylor (|r| <= 0.3466: r^14/14! ~ 4e-18)
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
    return ldexp(p, (int)n);
}

__device__ __forceinline__ double fast_tanh(double x) {
    const double ax = fmin(fabs(x), 20.0);
    const double e = fast_exp_pos(2.0 * ax);
    const double d = e + 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double t = fma(-2.0, r, 1.0);
    return copysign(t, x);
}

// table-driven variant: exp(y) = 2^(n/64) * exp(r), n = rint(y * 64/ln2), |r| <= ln2/128 -> degree-5 polynomial;
// 2^(j/64) from a 64-entry LDS table (the lookup runs on the LDS pipe, not the shared DP pipe)
__device__ double g_exp2_tab[64];
template <int NEWTON>
__device__ __forceinline__ double table_tanh(double x, const double* tab) {
    const double ax = fmin(fabs(x), 20.0);
    const double y = ax + ax;
    const double n = rint(y * 92.332482616893656877);            // 64 / ln2
    double r = fma(-n, 0.01083042469326756, y);                  // ln2/64 hi (21 trailing bits zero: n * hi exact)
    r = fma(-n, 2.9815858269852933e-12, r);                      // ln2/64 lo
    double p = 1.0 / 120.0;
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int ni = (int)n;
    const double e = ldexp(p * tab[ni & 63], ni >> 6);
    const double d = e + 1.0;
    double q = __builtin_amdgcn_rcp(d);
    q = fma(fma(-d, q, 1.0), q, q);
    if (NEWTON > 1) q = fma(fma(-d, q, 1.0), q, q);
    return copysign(fma(-2.0, q, 1.0), x);
}

template <int MODE>
__global__ void tanh_rate(const double* in, double* out, long long* cyc, int iters) {
    __shared__ double tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = g_exp2_tab[threadIdx.x];
    __syncthreads();
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[(blockIdx.x * blockDim.x + threadIdx.x) * 16 + i];
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) v[i] = tanh(v[i]) * 1.7;
            else if (MODE == 1) v[i] = fast_tanh(v[i]) * 1.7;
            else if (MODE == 2) v[i] = table_tanh<2>(v[i], tab) * 1.7;
            else v[i] = table_tanh<1>(v[i], tab) * 1.7;
        }
    }
    long long t1 = clock64();
    for (int i = 0; i < 16; ++i) out[(blockIdx.x * blockDim.x + threadIdx.x) * 16 + i] = v[i];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void tanh_acc(const double* in, double* o_ref, double* o_fast, double* o_t2, double* o_t1, int n) {
    __shared__ double tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = g_exp2_tab[threadIdx.x];
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        o_ref[i] = tanh(in[i]);
        o_fast[i] = fast_tanh(in[i]);
        o_t2[i] = table_tanh<2>(in[i], tab);
        o_t1[i] = table_tanh<1>(in[i], tab);
    }
}

// do f64 MFMA and f64 VALU overlap on one SIMD?  512 threads = 2 waves per SIMD; waves 0-3 run MFMAs,
// waves 4-7 run independent v_fma_f64 (mode 0: both, 1: MFMA waves only, 2: VALU waves only)
__global__ void coexec_f64(double* out, long long* cyc, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    long long t0 = clock64();
    if (mf && mode != 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    } else if (!mf && mode != 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fma(v[i], b, a);     // 64 DP FMAs per iteration = 256 cycles
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) cyc[blockIdx.x * 2 + (wave >> 2)] = t1 - t0;
}

static double median(std::vector<long long>& v) {
    std::sort(v.begin(), v.end());
    return (double)v[v.size() / 2];
}

int main() {
    const int blocks = 256, iters = 2000;
    double* dout;
    long long* dcyc;
    CK(hipMalloc(&dout, sizeof(double) * blocks * 1024 * 16));
    CK(hipMalloc(&dcyc, sizeof(long long) * blocks));
    std::vector<long long> cyc(blocks);
    auto report = [&](const char* name, int per_iter, int waves) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(cyc.data(), dcyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
        printf("%-44s waves/SIMD=%d  cycles per instr (per wave) = %.2f\n", name, waves,
               median(cyc) / ((double)iters * per_iter));
    };
    // one wave per SIMD (256 threads/block, 1 block/CU) and two waves per SIMD (512 threads)
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(mfma_f64_rate<1>, dim3(blocks), dim3(threads), 0, 0, dout, dcyc, iters);
        report("mfma_f64_16x16x4 dependent chain (1 acc)", 1, threads / 256);
        hipLaunchKernelGGL(mfma_f64_rate<4>, dim3(blocks), dim3(threads), 0, 0, dout, dcyc, iters);
        report("mfma_f64_16x16x4 4 independent acc", 4, threads / 256);
        hipLaunchKernelGGL(mfma_f32_rate<1>, dim3(blocks), dim3(threads), 0, 0, (float*)dout, dcyc, iters);
        report("mfma_f32_16x16x4 dependent chain (1 acc)", 1, threads / 256);
        hipLaunchKernelGGL(mfma_f32_rate<4>, dim3(blocks), dim3(threads), 0, 0, (float*)dout, dcyc, iters);
        report("mfma_f32_16x16x4 4 independent acc", 4, threads / 256);
    }
    // wall-clock throughput: every SIMD busy, 1 / 2 / 4 waves per SIMD
    for (int threads : {256, 512, 1024}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int big = 20000;
        hipLaunchKernelGGL(mfma_f64_rate<4>, dim3(blocks), dim3(threads), 0, 0, dout, dcyc, 100);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_f64_rate<4>, dim3(blocks), dim3(threads), 0, 0, dout, dcyc, big);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)blocks * (threads / 64) * big * 4.0 * 2048.0;
        printf("f64 mfma wall: %d waves/SIMD  %.2f ms  -> %.1f TFLOP/s\n", threads / 256, ms, flops / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_f32_rate<4>, dim3(blocks), dim3(threads), 0, 0, (float*)dout, dcyc, big);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("f32 mfma wall: %d waves/SIMD  %.2f ms  -> %.1f TFLOP/s\n", threads / 256, ms, flops / ms / 1e9);
    }
    {
        long long* dc2;
        CK(hipMalloc(&dc2, sizeof(long long) * blocks * 2));
        std::vector<long long> c2(blocks * 2);
        const char* names[3] = {"both", "mfma waves only", "valu waves only"};
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(coexec_f64, dim3(blocks), dim3(512), 0, 0, dout, dc2, 100, mode);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(coexec_f64, dim3(blocks), dim3(512), 0, 0, dout, dc2, 4000, mode);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("coexec f64 [%s]: wall %.3f ms (4000 iters: 16000 MFMAs = 1.02M cycles alone; 256k DP FMAs = 1.02M cycles alone)\n", names[mode], ms);
        }
    }
    // tanh
    const int n = blocks * 512 * 16;
    std::vector<double> hin(n);
    srand(1);
    for (int i = 0; i < n; ++i) hin[i] = ((double)rand() / RAND_MAX * 2 - 1) * ((i % 7 == 0) ? 30.0 : 3.0);
    hin[0] = 0.0; hin[1] = 1e-9; hin[2] = -1e-9; hin[3] = 25.0; hin[4] = -700.0; hin[5] = 1e-300;
    double *din, *dref, *dfast;
    CK(hipMalloc(&din, sizeof(double) * n));
    CK(hipMalloc(&dref, sizeof(double) * n));
    CK(hipMalloc(&dfast, sizeof(double) * n));
    CK(hipMemcpy(din, hin.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    {
        double htab[64];
        for (int j = 0; j < 64; ++j) htab[j] = std::exp2(j / 64.0);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_exp2_tab), htab, sizeof(htab)));
    }
    const int titers = 50;
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(tanh_rate<2>, dim3(blocks), dim3(threads), 0, 0, din, dout, dcyc, titers);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(cyc.data(), dcyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
        printf("table tanh, 2 Newton   waves/SIMD=%d  cycles per tanh (wave-instr) = %.1f\n", threads / 256,
               median(cyc) / (titers * 16.0));
        hipLaunchKernelGGL(tanh_rate<3>, dim3(blocks), dim3(threads), 0, 0, din, dout, dcyc, titers);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(cyc.data(), dcyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
        printf("table tanh, 1 Newton   waves/SIMD=%d  cycles per tanh (wave-instr) = %.1f\n", threads / 256,
               median(cyc) / (titers * 16.0));
        hipLaunchKernelGGL(tanh_rate<0>, dim3(blocks), dim3(threads), 0, 0, din, dout, dcyc, titers);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(cyc.data(), dcyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
        printf("ocml tanh(double)      waves/SIMD=%d  cycles per tanh (wave-instr) = %.1f\n", threads / 256,
               median(cyc) / (titers * 16.0));
        hipLaunchKernelGGL(tanh_rate<1>, dim3(blocks), dim3(threads), 0, 0, din, dout, dcyc, titers);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(cyc.data(), dcyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
        printf("exp-based fast_tanh    waves/SIMD=%d  cycles per tanh (wave-instr) = %.1f\n", threads / 256,
               median(cyc) / (titers * 16.0));
    }
    double *dt2, *dt1;
    CK(hipMalloc(&dt2, sizeof(double) * n));
    CK(hipMalloc(&dt1, sizeof(double) * n));
    hipLaunchKernelGGL(tanh_acc, dim3((n + 255) / 256), dim3(256), 0, 0, din, dref, dfast, dt2, dt1, n);
    std::vector<double> href(n), hfast(n), ht2(n), ht1(n);
    CK(hipMemcpy(ht2.data(), dt2, sizeof(double) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ht1.data(), dt1, sizeof(double) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(href.data(), dref, sizeof(double) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hfast.data(), dfast, sizeof(double) * n, hipMemcpyDeviceToHost));
    double e_ocml = 0, e_fast = 0, e_t2 = 0, e_t1 = 0;
    for (int i = 0; i < n; ++i) {
        const double t = std::tanh(hin[i]);
        e_ocml = fmax(e_ocml, fabs(href[i] - t));
        e_fast = fmax(e_fast, fabs(hfast[i] - t));
        e_t2 = fmax(e_t2, fabs(ht2[i] - t));
        e_t1 = fmax(e_t1, fabs(ht1[i] - t));
    }
    printf("max abs err vs host libm tanh: ocml %.3e, fast %.3e, table(2 Newton) %.3e, table(1 Newton) %.3e\n", e_ocml, e_fast,
           e_t2, e_t1);
    return 0;
}
