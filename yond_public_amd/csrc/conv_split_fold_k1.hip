// K2s instantiations: folded tiles of the decoder GEMMs (input at most 16 pixels wide) -- see conv_split_kernel.h
#include "conv_split_kernel.h"

SPLIT_GROUP_FOLD_K1(SPLIT_INSTANTIATE)
