// Float32, XCD-local variants of the cooperative leaf (a translation unit of its own: parallel compile)
#define RFLU_PL_F32 1
#define RFLU_PL_XCD 1
#include "panel_local.hip"
