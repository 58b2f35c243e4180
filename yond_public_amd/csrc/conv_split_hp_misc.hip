// K2s instantiations (the h-only split-plane flow's last convolution with the fused output projection, the layers with a second output,
// the decoder GEMM with two sub-positions per tile): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_hp_misc
#include "conv_split_kernel.h"

SPLIT_GROUP_H_ISP_O4(SPLIT_INSTANTIATE)
SPLIT_GROUP_H_D2(SPLIT_INSTANTIATE)
SPLIT_GROUP_H_SUB2(SPLIT_INSTANTIATE)
