// matrix-core kernels, float, sigmoid hidden layers (see kernels_mfma_typed.inc)
#define NEMPC_T float
#define NEMPC_ACT 3   // NEMPC_ACT_SIGMOID
#include "kernels_mfma_typed.inc"
