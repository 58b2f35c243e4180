// K2s instantiations: folded tiles of the stride-2 layers (output at most 16 pixels wide) -- see conv_split_kernel.h
#include "conv_split_kernel.h"

SPLIT_GROUP_FOLD_S2(SPLIT_INSTANTIATE)
