// K2s instantiations (SPLIT_GROUP_HALF): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_half
#include "conv_split_kernel.h"

SPLIT_GROUP_HALF(SPLIT_INSTANTIATE)
