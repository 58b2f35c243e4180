// K2s instantiation: the decoder GEMM on 128-column tiles, split-plane inputs by LDS-DMA -- see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_k1_wide
#include "conv_split_kernel.h"

SPLIT_GROUP_K1_WIDE(SPLIT_INSTANTIATE)
