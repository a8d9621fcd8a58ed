#define MDSP_GX_INST 9
#include "gx_inst.inc"
