// matrix-core kernels, float, relu hidden layers (see kernels_mfma_typed.inc)
#define NEMPC_T float
#define NEMPC_ACT 2   // NEMPC_ACT_RELU
#include "kernels_mfma_typed.inc"
