#define MDSP_GX_INST 17
#include "gx_inst.inc"
