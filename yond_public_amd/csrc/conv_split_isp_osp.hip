// K2s instantiations (SPLIT_GROUP_ISP_OSP): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_isp_osp
#include "conv_split_kernel.h"

SPLIT_GROUP_ISP_OSP(SPLIT_INSTANTIATE)
