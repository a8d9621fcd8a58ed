#define MDSP_GX_INST 8
#include "gx_inst.inc"
