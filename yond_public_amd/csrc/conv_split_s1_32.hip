// K2s instantiations (SPLIT_GROUP_S1_32): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_s1_32
#include "conv_split_kernel.h"

SPLIT_GROUP_S1_32(SPLIT_INSTANTIATE)
