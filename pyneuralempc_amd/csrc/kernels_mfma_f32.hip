// matrix-core kernels, float, tanh hidden layers (see kernels_mfma_typed.inc)
#define NEMPC_T float
#define NEMPC_ACT 1   // NEMPC_ACT_TANH
#include "kernels_mfma_typed.inc"
