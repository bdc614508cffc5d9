// conv_cl_wgrad.hip compiled for IEEE-half operands (float32 weight gradients as before): cplxamd_conv2d_clh_wgrad*.
#define CPLXAMD_CONV_F16 1
#define clw clwh
#define cplxamd_conv2d_cl_wgrad_ws_bytes cplxamd_conv2d_clh_wgrad_ws_bytes
#define cplxamd_conv2d_cl_wgrad cplxamd_conv2d_clh_wgrad
#define cplxamd_conv2d_cl_wgrad_fl cplxamd_conv2d_clh_wgrad_fl
#include "conv_cl_wgrad.hip"
