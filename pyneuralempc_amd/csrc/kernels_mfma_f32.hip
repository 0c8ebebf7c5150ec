#define NEMPC_T float
#include "kernels_mfma_typed.inc"
