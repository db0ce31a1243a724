// Float32, any placement (a translation unit of its own: parallel compile)
#define RFLU_PL_F32 1
#include "panel_local.hip"
