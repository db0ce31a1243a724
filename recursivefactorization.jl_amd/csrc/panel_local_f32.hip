// Float32 instantiation of the XCD-local leaf kernel in its own translation unit (parallel compile, see build.py)
#define RFLU_PANEL_F32_TU
#include "panel_local.hip"
