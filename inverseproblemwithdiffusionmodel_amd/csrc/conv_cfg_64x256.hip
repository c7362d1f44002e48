// tile configuration <1, 4, 2, 2, 32, 1> of the fp32 MFMA convolution (generated layout: one TU per configuration)
#include "conv_kernel.h"
namespace ipdm_conv {
int conv_cfg_64x256(const ConvArgs& a, int ks, hipStream_t s) {
  return ks == 3 ? launch_cfg<1, 4, 2, 2, 32, 1, 8, 3>(a, s) : launch_cfg<1, 4, 2, 2, 32, 1, 8, 1>(a, s);
}
}  // namespace ipdm_conv
