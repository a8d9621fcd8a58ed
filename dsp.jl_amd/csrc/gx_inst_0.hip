#define MDSP_GX_INST 0
#include "gx_inst.inc"
