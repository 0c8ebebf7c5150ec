// matrix-core kernels, double, tanh hidden layers (see kernels_mfma_typed.inc)
#define NEMPC_T double
#define NEMPC_ACT 1   // NEMPC_ACT_TANH
#include "kernels_mfma_typed.inc"
