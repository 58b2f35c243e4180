// K2s instantiation: folded tiles -- two 16-column sub-tiles per MFMA row for images at most 16 pixels wide -- see conv_split_kernel.h
#include "conv_split_kernel.h"

SPLIT_GROUP_FOLD(SPLIT_INSTANTIATE)
