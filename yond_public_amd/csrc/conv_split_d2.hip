// conv_split_kernel instantiation: the stride-2 layer with the second output (SiLU in split planes; conv_split_kernel.h, D2)
#define SPLIT_DBG_READER yond_split_debug_read_d2
#include "conv_split_kernel.h"
SPLIT_GROUP_D2(SPLIT_INSTANTIATE)
