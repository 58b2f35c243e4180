// K2s instantiations (SPLIT_GROUP_S2_W4; round 6, measured no-go: experiment builds only): see conv_split_kernel.h, profiles/r06_experiments/README.md
#ifdef YOND_EXPERIMENTS
#define SPLIT_DBG_READER yond_split_debug_read_s2w4
#include "conv_split_kernel.h"

SPLIT_GROUP_S2_W4(SPLIT_INSTANTIATE)
#endif
