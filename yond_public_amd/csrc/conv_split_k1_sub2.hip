// conv_split_kernel instantiations: the level 1 -> 0 decoder GEMM with two sub-positions per channel tile (see conv_split_kernel.h)
#define SPLIT_DBG_READER yond_split_debug_read_k1_sub2
#include "conv_split_kernel.h"
SPLIT_GROUP_K1_SUB2(SPLIT_INSTANTIATE)
