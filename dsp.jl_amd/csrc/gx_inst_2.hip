#define MDSP_GX_INST 2
#include "gx_inst.inc"
