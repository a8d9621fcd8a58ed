#define MDSP_GX_INST 16
#include "gx_inst.inc"
