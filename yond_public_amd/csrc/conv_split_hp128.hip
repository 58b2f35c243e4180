// K2s instantiations (the h-only split-plane flow's 128-channel tiles): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_hp128
#include "conv_split_kernel.h"

SPLIT_GROUP_H128_OSP(SPLIT_INSTANTIATE)
SPLIT_GROUP_H128_ISP_OSP(SPLIT_INSTANTIATE)
