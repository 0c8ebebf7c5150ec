// Reproducer of the compiler defect the build's ISA stage repairs (pyneuralempc_amd/_isa.py; DESIGN.md "The exec-restore
// spill defect"; AMD clang 22 / ROCm 7.2): ONE instantiation of the wave-per-tile row kernel -- the fp64, 128-wide,
// two-hidden-layer relu one with streamed weights, whose Discret rows lost the `+I` of entry (1,1) in round 3.
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I pyneuralempc_amd/csrc \
//           --cuda-device-only -S -o repro.s tools/isa_defect_repro.hip
//     python tools/isa_lint.py repro.s
//
// In the text the compiler emits, the block that joins the (possibly empty) extra-input staging region starts with the
// register allocator's spill store in front of the instruction that gives the lanes back:
//
//     .LBB0_49:
//         v_accvgpr_write_b32 a163, v18        ; runs with the region's lanes only -- none when ne == 0
//         s_or_b64 exec, exec, s[0:1]
//
// tests/test_build_isa_cpu.py compiles this file and asserts what `_isa.scan` says about it: a finding made of spill
// stores only, which `_isa.repair` moves.  A toolchain on which the scan comes back empty no longer has the defect (in
// this kernel), and that test says so.
#include "kernels_mfma_impl.h"

namespace nempc {
template __global__ void rows_mfma_kernel<double, 128, 2, false, 4, NEMPC_ACT_RELU>(MfmaParams);
}  // namespace nempc
