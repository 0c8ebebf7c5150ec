// matrix-core kernels, float, elu hidden layers (see kernels_mfma_typed.inc)
#define NEMPC_T float
#define NEMPC_ACT 5   // NEMPC_ACT_ELU
#include "kernels_mfma_typed.inc"
