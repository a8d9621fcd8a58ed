#define MDSP_GX_INST 1
#include "gx_inst.inc"
