#define NEMPC_T double
#include "kernels_mfma_typed.inc"
