// Dependent-issue latency of v_fma_f64 on gfx950: N independent chains in one wave (one wave per SIMD), cycles per
// instruction.   hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o /tmp/ubench_chain && /tmp/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int N>
__global__ void chain(double* out, long long* cyc, int iters) {
    double v[N];
    for (int i = 0; i < N; ++i) v[i] = 1.0 + threadIdx.x * 1e-3 + i;
    const double m = 0.999999, c = 1e-7;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < N; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < N; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int N>
void run(double* out, long long* cyc, int lanes) {
    const int iters = 2000;
    hipLaunchKernelGGL(chain<N>, dim3(1), dim3(lanes), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(chain<N>, dim3(1), dim3(lanes), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    long long c;
    CK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
    printf("chains %d, %3d lanes: %.2f cycles per v_fma_f64\n", N, lanes, (double)c / ((double)iters * 8 * N));
}

int main() {
    double* out; long long* cyc;
    CK(hipMalloc(&out, 1024 * sizeof(double))); CK(hipMalloc(&cyc, 64 * sizeof(long long)));
    for (int lanes : {64, 16}) {
        run<1>(out, cyc, lanes); run<2>(out, cyc, lanes); run<3>(out, cyc, lanes); run<4>(out, cyc, lanes); run<8>(out, cyc, lanes);
    }
    return 0;
}
