#define MDSP_GX_INST 11
#include "gx_inst.inc"
