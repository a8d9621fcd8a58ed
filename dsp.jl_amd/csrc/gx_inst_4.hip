#define MDSP_GX_INST 4
#include "gx_inst.inc"
