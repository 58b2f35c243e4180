// K2s instantiations (the h-only split-plane flow's decoder GEMMs on taller / wider tiles): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_hp_k1t
#include "conv_split_kernel.h"

SPLIT_GROUP_H_K1_TALL(SPLIT_INSTANTIATE)
