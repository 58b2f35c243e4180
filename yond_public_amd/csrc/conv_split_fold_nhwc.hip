// K2s instantiations: folded tiles of the plain [N][H][W][C] 3x3 layer (images at most 16 pixels wide) -- see conv_split_kernel.h
#include "conv_split_kernel.h"

SPLIT_GROUP_FOLD_NHWC(SPLIT_INSTANTIATE)
