// K2s instantiations (SPLIT_GROUP_H_K1S2: the split-plane data flow on h-only planes, BASELINE cfg 5): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_hp_k1s2
#include "conv_split_kernel.h"

SPLIT_GROUP_H_K1S2(SPLIT_INSTANTIATE)
