#define MDSP_GX_INST 12
#include "gx_inst.inc"
