// conv_cl2.hip compiled for IEEE-half operands and float32 output (cplxamd_conv2d_cl2h_fl): the forward / data-gradient
// convolution of the float32 layers' half split products (cplxmodule_amd/x3.py 'x2', conv.py).
#define CPLXAMD_CONV_F16 1
#define cl2 cl2h
#include "conv_cl2.hip"
