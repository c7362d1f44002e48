// Dense 3x3 / 1x1 stride-1 convolution as an implicit GEMM on the gfx950 float32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 64 FLOP/clk/SIMD -- the fp32 MFMA roof is 157 TFLOP/s).
//
// GEMM view per image: D[co, pixel] = sum_{tap, ci} Wt[tap][ci][co] * P[ci][pixel + tap offset].
//   MFMA A operand (32 x 2): rows = output channels, k = two consecutive input channels
//   MFMA B operand (2 x 32): k = the same two input channels, columns = 32 pixels of the tile
//   accumulator (32 x 32):   lane <-> pixel (contiguous in NCHW, so stores are coalesced),
//                            registers <-> output channels
// One 256-thread workgroup owns (CO_T output channels) x (PH x PW pixels) of one image and walks the
// input channels in chunks of KC: the chunk's input patch (+halo, zero padded) and its 9*KC*CO_T weights
// are staged in LDS once and reused by all 9 taps, so every global byte feeds 9*CO_T (activations) or
// PH*PW (weights) multiply-adds.  The next chunk's global loads are issued before the current chunk's
// MFMA phase (register prefetch) and written to LDS after it.
//
// Fused on the way in:  InstanceNorm++ affine ((v - mu)*scale + shift, per image & channel) and the
//                       activation, applied while the patch goes global -> register -> LDS; padding
//                       stays exactly zero.
// Fused on the way out: bias, residual add, and optionally a second, ACTIVATED copy of the result
//                       (out_act = ELU(out)): the consumer convolution then reads ready-made operands.  Measured
//                       on MI355X this beats activating in the consumer's prologue, which repeats the VALU work
//                       for every co-tile and halo overlap (x2.7) inside an MFMA-bound loop: 96 vs 119 TFLOP/s.
//
// Reference call sites this replaces: torch.nn.Conv2d in ncsn/models/layers.py:28-60 (conv1x1, conv3x3,
// dilated_conv3x3) as used by ResidualBlock :401-456, RCUBlock :112-134, CRPBlock :62-83, MSFBlock
// :165-184 and NCSNv2Deepest.begin_conv/end_conv (ncsnv2.py:210-213).
#include "conv_kernel.h"

using namespace ipdm_conv;

namespace {

int forced_cfg() {
  static int v = -2;
  if (v == -2) {
    const char* e = getenv("IPDM_CONV_CFG");
    v = e ? atoi(e) : -1;
  }
  return v;
}

// Tile choice.  Wide images (W > 16): 256-pixel tiles (8 rows x 32); the 16x16 stage: 128-pixel tiles
// (8 rows x 16, LDS patch sized for dilation <= 4).  Measured on MI355X at B = 28 (scripts/bench_conv.py).
int dispatch_conv(const ConvArgs& a, int ks, hipStream_t s) {
  const int f = forced_cfg();       // tuning aid: IPDM_CONV_CFG=<id> forces one tile configuration
  const int64_t px = (int64_t)a.B * a.D * a.H * a.W;
  if (a.W <= 16) {
    if (f == 10) return conv_cfg_64x128s(a, ks, s);
    if (f == 11) return conv_cfg_128x128s(a, ks, s);
    if (f == 12) return conv_cfg_32x128s(a, ks, s);
    if (a.Cout >= 512 && a.Cout % 128 == 0 && px >= 4096) return conv_cfg_128x128s(a, ks, s);
    return conv_cfg_32x128s(a, ks, s);
  }
  if (a.dil > 1) return conv_cfg_64x256_dil(a, ks, s);
  if (f == 1) return conv_cfg_32x256(a, ks, s);
  if (f == 2) return conv_cfg_64x256(a, ks, s);
  if (f == 3) return conv_cfg_128x256(a, ks, s);
  if (a.Cout <= 32) return conv_cfg_32x256(a, ks, s);
  return conv_cfg_64x256(a, ks, s);
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout,
                                                          int Cin, int kk) {
  const int64_t total = (int64_t)Cout * Cin * kk;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int co = (int)(i % Cout);
    int64_t t = i / Cout;
    int ci = (int)(t % Cin);
    int tap = (int)(t / Cin);
    wt[i] = w[((int64_t)co * Cin + ci) * kk + tap];
  }
}

}  // namespace

extern "C" int ipdm_conv_pack_weight_f32(const float* w, float* wt, int Cout, int Cin, int k, void* stream) {
  IPDM_REQUIRE(w && wt && Cout > 0 && Cin > 0 && (k == 1 || k == 3 || k == 27));
  // k = 1 / 3: 2-D kernels [Cout][Cin][k][k]; k = 27: a 3x3x3 kernel [Cout][Cin][3][3][3] (taps flattened z, y, x)
  const int kk = k == 27 ? 27 : k * k;
  const int64_t total = (int64_t)Cout * Cin * kk;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, ipdm_stream(stream), w, wt, Cout,
                     Cin, kk);
  return ipdm_launch_status();
}

extern "C" int ipdm_conv2d_f32(const float* x, const float* wt, const float* bias, const float* coef, int act,
                               const float* residual, float* out, float* out_act, int act_out, int B, int Cin, int Cout,
                               int H, int W, int k, int dilation, int pool2, void* stream) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (k == 1 || k == 3) && dilation >= 1);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && wt && (out || out_act) && x != out && x != out_act);
  if (pool2) return IPDM_EUNSUPPORTED;           // ConvMeanPool epilogue: not fused yet (ipdm_meanpool2_f32)
  if (k == 3 && dilation > 4) return IPDM_EUNSUPPORTED;
  ConvArgs a;
  a.x = x; a.wt = wt; a.bias = bias; a.coef = coef; a.residual = residual; a.out = out; a.out_act = out_act; a.act_out = act_out;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = k == 3 ? dilation : 1; a.act = act;
  a.D = 1; a.kd = 1;
  a.tiles_x = a.tiles_y = a.co_tiles = 0; a.dbg = nullptr;
  return dispatch_conv(a, k, ipdm_stream(stream));
}

extern "C" int ipdm_conv3d_f32(const float* x, const float* wt, const float* bias, const float* coef, int act,
                               const float* residual, float* out, float* out_act, int act_out, int B, int Cin, int Cout,
                               int D, int H, int W, int k, int dilation, void* stream) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0 && (k == 1 || k == 3) && dilation >= 1);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && wt && (out || out_act) && x != out && x != out_act);
  if (k == 3 && dilation > 4) return IPDM_EUNSUPPORTED;
  ConvArgs a;
  a.x = x; a.wt = wt; a.bias = bias; a.coef = coef; a.residual = residual; a.out = out; a.out_act = out_act; a.act_out = act_out;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = k == 3 ? dilation : 1; a.act = act;
  a.D = D; a.kd = k;
  a.tiles_x = a.tiles_y = a.co_tiles = 0; a.dbg = nullptr;
  return dispatch_conv(a, k, ipdm_stream(stream));
}

// ---- Winograd F(2x2, 3x3) entry points (conv_wino.hip) ---------------------------------------------------------
static unsigned long long* g_wino_dbg = nullptr;
namespace ipdm_conv { unsigned long long* conv_debug_stamps() { return g_wino_dbg; } }
// tuning aid: a device buffer of 4 * n_blocks uint64 receives (start, loop start, loop end, end) s_memtime stamps
extern "C" int ipdm_debug_set_stamp_buffer(void* buf) {
  g_wino_dbg = (unsigned long long*)buf;
  return IPDM_OK;
}
extern "C" int ipdm_conv_wino_weight_f32(const float* w, float* U, int Cout, int Cin, void* stream) {
  IPDM_REQUIRE(w && U && Cout > 0 && Cin > 0);
  return conv_wino_weights(w, U, Cout, Cin, ipdm_stream(stream));
}

extern "C" int ipdm_conv2d_wino_supported(int Cin, int Cout, int H, int W, int dilation) {
  ConvArgs a;
  a.coef = nullptr; a.act = IPDM_ACT_NONE; a.dil = dilation; a.D = 1; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.B = 1;                       // the per-launch size limit (buffer descriptors) is checked again with the real batch
  return wino_ok(a, 3) ? 1 : 0;
}

extern "C" int ipdm_conv2d_wino_f32(const float* x, const float* U, const float* bias, const float* residual, float* out,
                                    float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int dilation,
                                    void* stream) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && dilation >= 1);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && U && (out || out_act) && x != out && x != out_act);
  ConvArgs a;
  a.x = x; a.wt = U; a.bias = bias; a.coef = nullptr; a.residual = residual; a.out = out; a.out_act = out_act;
  a.act_out = act_out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = dilation; a.act = IPDM_ACT_NONE;
  a.D = 1; a.kd = 1; a.tiles_x = a.tiles_y = a.co_tiles = 0; a.dbg = g_wino_dbg; a.dbg = g_wino_dbg;
  if (!wino_ok(a, 3)) return IPDM_EUNSUPPORTED;
  return conv_wino_launch(a, ipdm_stream(stream));
}
