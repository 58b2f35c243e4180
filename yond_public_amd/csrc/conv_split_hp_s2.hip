// K2s instantiations (the h-only split-plane flow's stride-2 layers on 8-row tiles): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_hp_s2
#include "conv_split_kernel.h"

SPLIT_GROUP_H_S2_TALL(SPLIT_INSTANTIATE)
