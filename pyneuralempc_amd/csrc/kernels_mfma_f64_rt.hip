// matrix-core kernels, double, per-layer activations at run time (see kernels_mfma_typed.inc, activations.h: ActL)
#define NEMPC_T double
#define NEMPC_ACT 100   // NEMPC_ACT_RUNTIME
#include "kernels_mfma_typed.inc"
