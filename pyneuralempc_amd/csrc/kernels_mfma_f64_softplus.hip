// matrix-core kernels, double, softplus hidden layers (see kernels_mfma_typed.inc)
#define NEMPC_T double
#define NEMPC_ACT 4   // NEMPC_ACT_SOFTPLUS
#include "kernels_mfma_typed.inc"
