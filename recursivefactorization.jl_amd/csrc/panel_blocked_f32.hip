// Float32 instantiations of the blocked leaf (a translation unit of its own: parallel compile)
#define RFLU_PB_F32 1
#include "panel_blocked.hip"
