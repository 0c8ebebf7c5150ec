// Accuracy of the lean fp64 exp / expm1 / log1p / sigmoid of csrc/activations.h against the host's long double routines.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I pyneuralempc_amd/csrc tools/ubench_explog.hip -o tools/_bin/ubench_explog && tools/_bin/ubench_explog
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "activations.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void eval(const double* x, double* o, int n, int what) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    o[i] = what == 0 ? nempc::nempc_exp(v) : what == 1 ? nempc::nempc_expm1(v) : what == 2 ? nempc::nempc_log1p(v) : nempc::nempc_sigmoid(v);
}

int main() {
    const int N = 1 << 20;
    std::vector<double> x(N), o(N);
    double *dx, *dq;
    CK(hipMalloc(&dx, N * 8)); CK(hipMalloc(&dq, N * 8));
    struct { const char* name; int what; double lo, hi; } runs[] = {
        {"exp     on [-40, 0]", 0, -40.0, 0.0}, {"exp     on [-3, 3]", 0, -3.0, 3.0}, {"exp     on [0, 700]", 0, 0.0, 700.0},
        {"expm1   on [-40, 0]", 1, -40.0, 0.0}, {"expm1   on [-1e-3, 1e-3]", 1, -1e-3, 1e-3}, {"expm1   on [-1e-9, 0]", 1, -1e-9, 0.0},
        {"log1p   on [0, 1]", 2, 0.0, 1.0}, {"log1p   on [0, 1e-6]", 2, 0.0, 1e-6}, {"sigmoid on [-40, 40]", 3, -40.0, 40.0}};
    for (auto& r : runs) {
        for (int i = 0; i < N; ++i) x[i] = r.lo + (r.hi - r.lo) * ((double)rand() / RAND_MAX);
        x[0] = r.lo; x[1] = r.hi;
        CK(hipMemcpy(dx, x.data(), N * 8, hipMemcpyHostToDevice));
        eval<<<N / 256, 256>>>(dx, dq, N, r.what);
        CK(hipMemcpy(o.data(), dq, N * 8, hipMemcpyDeviceToHost));
        long double ea = 0, er = 0;
        for (int i = 0; i < N; ++i) {
            const long double v = x[i];
            const long double ref = r.what == 0 ? expl(v) : r.what == 1 ? expm1l(v) : r.what == 2 ? log1pl(v) : 1.0L / (1.0L + expl(-v));
            const long double d = fabsl((long double)o[i] - ref);
            if (d > ea) ea = d;
            if (ref != 0 && d / fabsl(ref) > er) er = d / fabsl(ref);
        }
        printf("%-28s max abs err %.3Le   max rel err %.3Le\n", r.name, ea, er);
    }
    // the ends and NaN
    double sp[8] = {-1e308, -1100.0, -745.0, 709.0, 710.0, 1e300, NAN, 0.0};
    CK(hipMemcpy(dx, sp, 64, hipMemcpyHostToDevice));
    for (int w = 0; w < 2; ++w) {
        eval<<<1, 64>>>(dx, dq, 8, w);
        double r[8];
        CK(hipMemcpy(r, dq, 64, hipMemcpyDeviceToHost));
        printf("%s at -1e308 -1100 -745 709 710 1e300 NaN 0: %g %g %g %g %g %g %g %g\n", w ? "expm1" : "exp  ", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
    }
    return 0;
}
