// conv_split_kernel instantiations: the 32 -> 32 channel 3x3 layers on two weight buffers, weights resident in LDS (see conv_split_kernel.h)
#define SPLIT_DBG_READER yond_split_debug_read_wres
#include "conv_split_kernel.h"
SPLIT_GROUP_WRES(SPLIT_INSTANTIATE)
