// K2s instantiations (SPLIT_GROUP_ISP_K1S2): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_isp_k1s2
#include "conv_split_kernel.h"

SPLIT_GROUP_ISP_K1S2(SPLIT_INSTANTIATE)
