// matrix-core kernels, double, elu hidden layers (see kernels_mfma_typed.inc)
#define NEMPC_T double
#define NEMPC_ACT 5   // NEMPC_ACT_ELU
#include "kernels_mfma_typed.inc"
