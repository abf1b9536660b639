// One translation unit for the convolution kernels whose bodies the persistent per-XCD launch (mega.inc) calls as well.
#include "conv_igemm.hip"
#include "conv_halo.hip"
#include "mega.inc"
