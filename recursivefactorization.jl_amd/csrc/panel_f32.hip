// panel_f32.hip -- second translation unit of panel.hip: the Float32 instantiations of the cooperative leaf kernels, so that
// the two element types compile in parallel (each unrolls 64 pivot steps three times; one unit took 4.5 minutes).
#define RFLU_PANEL_F32_TU
#include "panel.hip"
