#define MDSP_GX_INST 10
#include "gx_inst.inc"
