// K2s instantiations: the taller tiles of the h-only (fp16 MFMA) path -- see conv_split_kernel.h
#include "conv_split_kernel.h"

SPLIT_GROUP_HALF_TALL(SPLIT_INSTANTIATE)
