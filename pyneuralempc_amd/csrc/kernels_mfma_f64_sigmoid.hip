// matrix-core kernels, double, sigmoid hidden layers (see kernels_mfma_typed.inc)
#define NEMPC_T double
#define NEMPC_ACT 3   // NEMPC_ACT_SIGMOID
#include "kernels_mfma_typed.inc"
