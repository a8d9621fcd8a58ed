#define MDSP_GX_INST 3
#include "gx_inst.inc"
