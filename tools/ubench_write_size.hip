// Calibration of rocprofv3's WRITE_SIZE for the store shapes of the fused dense epilogue (round-4 review item): what does
// the counter charge for an ISOLATED 8- or 16-byte write-through store, as opposed to the 16-byte-per-lane streams it is
// exact for (MI355X_MICROARCH.md)?  Each kernel writes `count` pieces of `piece` bytes, one per `stride` bytes, with
// `global_store_dwordx2 / x4 ... sc0 sc1` (the stores of fx_store_wt / fx_store_wt2) or plain stores.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_write_size tools/ubench_write_size.hip
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- tools/_bin/ubench_write_size      (tools/write_size_calibration.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int PIECE, bool WT>
__global__ void store_pieces(char* base, size_t stride, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    char* p = base + i * stride;
    if constexpr (PIECE == 8) {
        const double v = 1.0;
        if (WT) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    } else {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 v = {1.0, 2.0};
        if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const size_t count = 1 << 20;               // pieces per launch
    const size_t bytes = count * 512 + 4096;    // the widest stride
    char* buf;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    const dim3 block(256), grid((unsigned)(count / 256));
    // kernel name tells the shape: piece bytes, write-through or plain; the stride is the launch ORDER within a name:
    // 8 B pieces at strides 8 (a stream), 32, 64, 512; 16 B pieces at strides 16 (a stream), 32, 64, 512
    const size_t s8[4] = {8, 32, 64, 512}, s16[4] = {16, 32, 64, 512};
    for (int rep = 0; rep < 3; ++rep) {
        for (int k = 0; k < 4; ++k) {
            hipLaunchKernelGGL((store_pieces<8, true>), grid, block, 0, 0, buf, s8[k], count);
            hipLaunchKernelGGL((store_pieces<8, false>), grid, block, 0, 0, buf, s8[k], count);
            hipLaunchKernelGGL((store_pieces<16, true>), grid, block, 0, 0, buf, s16[k], count);
            hipLaunchKernelGGL((store_pieces<16, false>), grid, block, 0, 0, buf, s16[k], count);
            CK(hipDeviceSynchronize());
        }
    }
    printf("count %zu pieces per launch; per name the launches cycle over strides {8|16, 32, 64, 512} bytes, 3 repetitions\n", count);
    CK(hipFree(buf));
    return 0;
}
