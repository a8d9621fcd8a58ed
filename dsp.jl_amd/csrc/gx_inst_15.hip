#define MDSP_GX_INST 15
#include "gx_inst.inc"
