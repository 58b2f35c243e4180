// K2s instantiations (SPLIT_GROUP_H_ISP_OSP: the split-plane data flow on h-only planes, BASELINE cfg 5): see conv_split_kernel.h
#define SPLIT_DBG_READER yond_split_debug_read_hp_isp_osp
#include "conv_split_kernel.h"

SPLIT_GROUP_H_ISP_OSP(SPLIT_INSTANTIATE)
