// K2s instantiation: the decoder GEMM with the second, split-plane output -- see conv_split_kernel.h
#include "conv_split_kernel.h"

SPLIT_GROUP_K1_D2(SPLIT_INSTANTIATE)
