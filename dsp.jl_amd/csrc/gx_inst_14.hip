#define MDSP_GX_INST 14
#include "gx_inst.inc"
