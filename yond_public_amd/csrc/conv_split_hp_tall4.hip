// K2s instantiations (the h-only split-plane flow's 3x3 layers with four rows per wave): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_hp_tall4
#include "conv_split_kernel.h"

SPLIT_GROUP_H_TALL4(SPLIT_INSTANTIATE)
