// conv_split_kernel instantiations (SPLIT_GROUP_WRES_W4; round 6, measured no-go: experiment builds only): the 32 -> 32 channel 3x3 layers (level 0) as half-size
// workgroups, two per CU -- see conv_split_kernel.h, profiles/r06_experiments/README.md section 6
#ifdef YOND_EXPERIMENTS
#define SPLIT_DBG_READER yond_split_debug_read_wres_w4
#include "conv_split_kernel.h"
SPLIT_GROUP_WRES_W4(SPLIT_INSTANTIATE)
#endif
