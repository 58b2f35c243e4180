// K2s instantiations: 128-channel tiles of the h-only (fp16 MFMA) path -- see conv_split_kernel.h
#include "conv_split_kernel.h"

SPLIT_GROUP_HALF128(SPLIT_INSTANTIATE)
