#define MDSP_GX_INST 19
#include "gx_inst.inc"
