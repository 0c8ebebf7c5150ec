// Issue cost (cycles per wave-instruction, one wave per SIMD and two) of the double-precision vector instructions a
// tanh is made of, on gfx950:  hipcc --offload-arch=gfx950 -O3 tools/ubench_dpops.hip -o /tmp/ubench_dpops && /tmp/ubench_dpops
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define OPT(name, TYPE, text)                                                            \
    __global__ void k_##name(double* out, long long* cyc, int iters) {                  \
        TYPE v[8];                                                                       \
        for (int i = 0; i < 8; ++i) v[i] = (TYPE)(1.0 + threadIdx.x * 1e-3 + i);        \
        long long t0 = clock64();                                                        \
        for (int it = 0; it < iters; ++it) {                                             \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(text : "+v"(v[i]) : : "v40", "v41", "vcc"); \
        }                                                                                \
        long long t1 = clock64();                                                        \
        double s = 0;                                                                    \
        for (int i = 0; i < 8; ++i) s += (double)v[i];                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                  \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                 \
    }

#define OP1(name, text) OPT(name, double, text)
#define OP32(name, text) OPT(name, float, text)
OP1(fma, "v_fma_f64 %0, %0, %0, %0")
OP1(add, "v_add_f64 %0, %0, %0")
OP1(mul, "v_mul_f64 %0, %0, %0")
OP1(min, "v_min_f64 %0, |%0|, %0")
OP1(rcp, "v_rcp_f64 %0, %0")
OP1(rndne, "v_rndne_f64 %0, %0")
OP1(ldexp, "v_ldexp_f64 %0, %0, 3")
OP1(cvt_i32, "v_cvt_i32_f64 v40, %0")         // result discarded (v40 clobbered)
OP1(cvt_f32, "v_cvt_f32_f64 v40, %0")
OP32(cvt_f64, "v_cvt_f64_f32 v[40:41], %0")
OP32(rcp32, "v_rcp_f32 %0, %0")
OP32(exp32, "v_exp_f32 %0, %0")
OP32(fma32, "v_fma_f32 %0, %0, %0, %0")
OP1(pkfma32, "v_pk_fma_f32 %0, %0, %0, %0")
OP32(addu32, "v_add_u32 %0, %0, %0")
OP32(lshladd, "v_lshl_add_u32 %0, %0, 3, %0")
OP32(bfi, "v_bfi_b32 %0, %0, %0, %0")
OP1(cmp, "v_cmp_u_f64 vcc, %0, %0")
OP32(cndmask, "v_cndmask_b32 %0, %0, %0, vcc")
OP32(mov, "v_mov_b32 %0, %0")

// Does a vector instruction of kind KIND co-issue with v_mfma_f64_16x16x4_f64 of another wave on the same SIMD?
// 512 threads = 2 waves per SIMD; waves 0-3 run MFMAs, waves 4-7 the vector instruction (mode 0 both, 1 MFMA only, 2 vector only)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
// same question for the single-precision matrix instruction (v_mfma_f32_16x16x4_f32, 32 cycles)
template <int KIND>
__global__ void coexec32(double* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    double v[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i; f[i] = a + i; }
    if (mf && mode != 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);   // 8 x 32 cycles
    } else if (!mf && mode != 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(v[i]));
                    if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
                    if (KIND == 2) asm volatile("v_add_u32 %0, %0, %0" : "+v"(f[i]));
                }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
__global__ void coexec(double* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double v[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i; f[i] = (float)(a + i); }
    if (mf && mode != 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);   // 256 cycles
    } else if (!mf && mode != 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {                                                                 // 64 instr = 256 cycles
                    if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(v[i]));
                    if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
                    if (KIND == 2) asm volatile("v_add_u32 %0, %0, %0" : "+v"(f[i]));
                    if (KIND == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[i]));
                }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int blocks = 256, iters = 20000;
    double* dout; long long* dcyc;
    CK(hipMalloc(&dout, sizeof(double) * blocks * 2048));
    CK(hipMalloc(&dcyc, sizeof(long long) * blocks * 2));
    std::vector<long long> cyc(blocks);
#define RUN(name)                                                                                              \
    for (int wps : {1, 2, 8}) {                                                                                \
        const int threads = wps >= 4 ? 1024 : wps * 256, nb = blocks * (wps >= 4 ? wps / 4 : 1);               \
        hipEvent_t e0, e1;                                                                                     \
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));                                                      \
        hipLaunchKernelGGL(k_##name, dim3(nb), dim3(threads), 0, 0, dout, dcyc, 100);                          \
        CK(hipEventRecord(e0));                                                                                \
        hipLaunchKernelGGL(k_##name, dim3(nb), dim3(threads), 0, 0, dout, dcyc, iters);                        \
        CK(hipEventRecord(e1));                                                                                \
        CK(hipEventSynchronize(e1));                                                                           \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                                        \
        CK(hipMemcpy(cyc.data(), dcyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));                    \
        std::sort(cyc.begin(), cyc.begin() + blocks);                                                          \
        /* wall: ns of one SIMD per wave-instruction = ms / (iters * 8 instr * wps waves) */                   \
        printf("%-10s waves/SIMD=%d  clock64 ticks per wave-instr (per wave) %.2f   wall ns per wave-instr per SIMD %.3f\n", \
               #name, wps, (double)cyc[blocks / 2] / (iters * 8.0), ms * 1e6 / ((double)iters * 8.0 * wps));   \
    }
    RUN(fma) RUN(add) RUN(mul) RUN(min) RUN(rcp) RUN(rndne) RUN(ldexp) RUN(cvt_i32) RUN(cvt_f32) RUN(cvt_f64)
    RUN(rcp32) RUN(exp32) RUN(fma32) RUN(pkfma32) RUN(addu32) RUN(lshladd) RUN(bfi) RUN(cmp) RUN(cndmask) RUN(mov)
    const char* kinds[4] = {"v_fma_f64", "v_fma_f32", "v_add_u32", "v_rcp_f64"};
    const char* modes[3] = {"both", "mfma waves only", "vector waves only"};
#define CO(K)                                                                                      \
    for (int mode = 0; mode < 3; ++mode) {                                                         \
        hipEvent_t e0, e1;                                                                         \
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));                                          \
        hipLaunchKernelGGL(coexec<K>, dim3(blocks), dim3(512), 0, 0, dout, 100, mode);             \
        CK(hipEventRecord(e0));                                                                    \
        hipLaunchKernelGGL(coexec<K>, dim3(blocks), dim3(512), 0, 0, dout, 4000, mode);            \
        CK(hipEventRecord(e1));                                                                    \
        CK(hipEventSynchronize(e1));                                                               \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                            \
        printf("coexec f64 MFMA with %-10s [%-17s]: wall %.3f ms\n", kinds[K], modes[mode], ms);   \
    }
    CO(0) CO(1) CO(2) CO(3)
#define CO32(K)                                                                                    \
    for (int mode = 0; mode < 3; ++mode) {                                                         \
        hipEvent_t e0, e1;                                                                         \
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));                                          \
        hipLaunchKernelGGL(coexec32<K>, dim3(blocks), dim3(512), 0, 0, dout, 100, mode);           \
        CK(hipEventRecord(e0));                                                                    \
        hipLaunchKernelGGL(coexec32<K>, dim3(blocks), dim3(512), 0, 0, dout, 4000, mode);          \
        CK(hipEventRecord(e1));                                                                    \
        CK(hipEventSynchronize(e1));                                                               \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                            \
        printf("coexec f32 MFMA with %-10s [%-17s]: wall %.3f ms\n", kinds[K], modes[mode], ms);   \
    }
    CO32(0) CO32(1) CO32(2)
    return 0;
}
