#define MDSP_GX_INST 6
#include "gx_inst.inc"
