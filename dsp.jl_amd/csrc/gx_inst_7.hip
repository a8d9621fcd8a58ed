#define MDSP_GX_INST 7
#include "gx_inst.inc"
