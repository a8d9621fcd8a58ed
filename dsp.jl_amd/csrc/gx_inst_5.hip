#define MDSP_GX_INST 5
#include "gx_inst.inc"
