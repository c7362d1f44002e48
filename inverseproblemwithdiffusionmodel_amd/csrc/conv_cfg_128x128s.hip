// tile configuration <2, 2, 2, 2, 16, 4> of the fp32 MFMA convolution (generated layout: one TU per configuration)
#include "conv_kernel.h"
namespace ipdm_conv {
int conv_cfg_128x128s(const ConvArgs& a, int ks, hipStream_t s) {
  return ks == 3 ? launch_cfg<2, 2, 2, 2, 16, 4, 8, 3>(a, s) : launch_cfg<2, 2, 2, 2, 16, 4, 8, 1>(a, s);
}
}  // namespace ipdm_conv
