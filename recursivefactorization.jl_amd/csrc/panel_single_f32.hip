// Float32 instantiations of the single-workgroup leaf (a translation unit of its own: parallel compile).
#define RFLU_PANEL_F32_TU 1
#include "panel_single.hip"
