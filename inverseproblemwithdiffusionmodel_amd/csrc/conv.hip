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
// Fused on the way out: bias, residual add.
//
// Reference call sites this replaces: torch.nn.Conv2d in ncsn/models/layers.py:28-60 (conv1x1, conv3x3,
// dilated_conv3x3) as used by ResidualBlock :401-456, RCUBlock :112-134, CRPBlock :62-83, MSFBlock
// :165-184 and NCSNv2Deepest.begin_conv/end_conv (ncsnv2.py:210-213).
#include "ipdm_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

struct ConvArgs {
  const float* x;
  const float* wt;
  const float* bias;
  const float* coef;
  const float* residual;
  float* out;
  int B, Cin, Cout, H, W, dil, act;
  int tiles_x, tiles_y, co_tiles;
};

// NCT x NPT MFMA tiles per wave, WCO x WPX waves (WCO*WPX == 4), PW = pixel-tile width (16 or 32),
// DMAX = largest dilation the LDS patch is sized for, KC = input channels per chunk, KS = 1 or 3.
template <int NCT, int NPT, int WCO, int WPX, int PW, int DMAX, int KC, int KS>
struct ConvCfg {
  static constexpr int TAPS = KS * KS;
  static constexpr int CO_T = 32 * NCT * WCO;
  static constexpr int ROWS_PER_TILE = 32 / PW;
  static constexpr int PH = NPT * WPX * ROWS_PER_TILE;
  static constexpr int HALO = KS == 3 ? DMAX : 0;
  static constexpr int PHP = PH + 2 * HALO;
  static constexpr int PWP = PW + 2 * HALO;
  // PW == 16: two image rows share one 32-lane group -> pitch = 16 (mod 32) keeps them on disjoint banks
  static constexpr int PITCH = PW == 32 ? PWP : (PWP <= 16 ? 16 : 48);
  static constexpr int PLANE = PHP * PITCH;
  static constexpr int W_ELEMS = TAPS * KC * CO_T;
  static constexpr int P_ELEMS = KC * PLANE;
  static constexpr int W_VEC_PER_THREAD = (W_ELEMS / 4 + 255) / 256;
  static constexpr int P_POS_PER_THREAD = (PHP * PWP + 255) / 256;
};

template <int NCT, int NPT, int WCO, int WPX, int PW, int DMAX, int KC, int KS>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
  using C = ConvCfg<NCT, NPT, WCO, WPX, PW, DMAX, KC, KS>;
  static_assert(WCO * WPX == 4, "four waves per workgroup");
  static_assert(KC % 2 == 0, "MFMA k-step is two input channels");
  __shared__ __align__(16) float Ws[C::W_ELEMS];
  __shared__ __align__(16) float Ps[C::P_ELEMS];

  // ---- which tile: XCD-aware remap so that the co-tiles of one pixel tile share an L2 ----
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, slot = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int co_tile = bid % a.co_tiles;
  int t = bid / a.co_tiles;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int co0 = co_tile * C::CO_T;
  const int y0 = ty * C::PH, x0 = tx * PW;
  const int d = KS == 3 ? a.dil : 0;          // halo actually used (<= DMAX)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int wco = wave / WPX, wpx = wave % WPX;
  const int HW = a.H * a.W;

  // ---- per-lane LDS read offsets ----
  const int a_base = h * C::CO_T + wco * NCT * 32 + j;
  int b_base[NPT];
#pragma unroll
  for (int n = 0; n < NPT; ++n) {
    const int tile = wpx * NPT + n;
    const int prow = PW == 32 ? tile : tile * 2 + (j >> 4);
    const int pcol = PW == 32 ? j : (j & 15);
    b_base[n] = h * C::PLANE + (prow + d) * C::PITCH + pcol + d;
  }

  f32x16 acc[NCT][NPT];
#pragma unroll
  for (int m = 0; m < NCT; ++m)
#pragma unroll
    for (int n = 0; n < NPT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // ---- staging registers (prefetch of the next chunk) ----
  float4 wreg[C::W_VEC_PER_THREAD];
  float preg[C::P_POS_PER_THREAD][KC];
  const bool w_vec_ok = (a.Cout % 4 == 0) && (co0 + C::CO_T <= a.Cout);
  const int pwv = PW + 2 * d, phv = C::PH + 2 * d;      // valid patch extent for this dilation

  auto load_chunk = [&](int c0) {
    // weights: rows (tap, kc) of CO_T contiguous floats in wt[tap][ci][co]
#pragma unroll
    for (int i = 0; i < C::W_VEC_PER_THREAD; ++i) {
      const int v = tid + i * 256;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < C::W_ELEMS / 4) {
        const int row = v / (C::CO_T / 4), q = v % (C::CO_T / 4);
        const int tap = row / KC, kc = row % KC;
        const int ci = c0 + kc;
        if (ci < a.Cin) {
          const float* src = a.wt + ((size_t)tap * a.Cin + ci) * a.Cout + co0 + q * 4;
          if (w_vec_ok) {
            val = *reinterpret_cast<const float4*>(src);
          } else {
            const int co = co0 + q * 4;
            if (co + 0 < a.Cout) val.x = src[0];
            if (co + 1 < a.Cout) val.y = src[1];
            if (co + 2 < a.Cout) val.z = src[2];
            if (co + 3 < a.Cout) val.w = src[3];
          }
        }
      }
      wreg[i] = val;
    }
    // input patch: position (r, c) of the (PH+2d) x (PW+2d) window, all KC channels
#pragma unroll
    for (int i = 0; i < C::P_POS_PER_THREAD; ++i) {
      const int p = tid + i * 256;
      const int r = p / pwv, c = p - r * pwv;
      const int gy = y0 - d + r, gx = x0 - d + c;
      const bool inb = (r < phv) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const float* src = a.x + ((size_t)b * a.Cin + c0) * HW + (size_t)gy * a.W + gx;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        float v = 0.f;
        if (inb && c0 + kc < a.Cin) v = src[(size_t)kc * HW];
        preg[i][kc] = v;
      }
    }
  };

  auto store_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < C::W_VEC_PER_THREAD; ++i) {
      const int v = tid + i * 256;
      if (v < C::W_ELEMS / 4) reinterpret_cast<float4*>(Ws)[v] = wreg[i];
    }
#pragma unroll
    for (int i = 0; i < C::P_POS_PER_THREAD; ++i) {
      const int p = tid + i * 256;
      const int r = p / pwv, c = p - r * pwv;
      if (r < phv) {
        const int gy = y0 - d + r, gx = x0 - d + c;
        const bool inb = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          float v = preg[i][kc];
          if (inb && c0 + kc < a.Cin) {
            if (a.coef) {
              const float* cf = a.coef + ((size_t)b * a.Cin + c0 + kc) * 3;
              v = (v - cf[0]) * cf[1] + cf[2];
            }
            v = ipdm_act(v, a.act);
          } else {
            v = 0.f;
          }
          Ps[kc * C::PLANE + r * C::PITCH + c] = v;
        }
      }
    }
  };

  const int n_chunks = (a.Cin + KC - 1) / KC;
  load_chunk(0);
  for (int ch = 0; ch < n_chunks; ++ch) {
    __syncthreads();                      // everyone finished reading the previous chunk from LDS
    store_chunk(ch * KC);
    __syncthreads();
    if (ch + 1 < n_chunks) load_chunk((ch + 1) * KC);   // in flight during the MFMA phase
    const int kc_valid = (a.Cin - ch * KC) < KC ? (a.Cin - ch * KC) : KC;
    const int ksteps = (kc_valid + 1) / 2;
#pragma unroll
    for (int tap = 0; tap < C::TAPS; ++tap) {
      const int dy = KS == 3 ? tap / 3 - 1 : 0, dx = KS == 3 ? tap % 3 - 1 : 0;
      const int tap_off = (dy * C::PITCH + dx) * d;
#pragma unroll
      for (int ks = 0; ks < KC / 2; ++ks) {
        if (ks < ksteps) {
          float av[NCT], bv[NPT];
#pragma unroll
          for (int m = 0; m < NCT; ++m) av[m] = Ws[(tap * KC + 2 * ks) * C::CO_T + a_base + 32 * m];
#pragma unroll
          for (int n = 0; n < NPT; ++n) bv[n] = Ps[2 * ks * C::PLANE + b_base[n] + tap_off];
#pragma unroll
          for (int m = 0; m < NCT; ++m)
#pragma unroll
            for (int n = 0; n < NPT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: bias, residual, coalesced stores (lane <-> pixel) ----
#pragma unroll
  for (int n = 0; n < NPT; ++n) {
    const int tile = wpx * NPT + n;
    const int prow = PW == 32 ? tile : tile * 2 + (j >> 4);
    const int pcol = PW == 32 ? j : (j & 15);
    const int gy = y0 + prow, gx = x0 + pcol;
    if (gy >= a.H || gx >= a.W) continue;
#pragma unroll
    for (int m = 0; m < NCT; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wco * NCT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < a.Cout) {
          const size_t o = ((size_t)b * a.Cout + co) * HW + (size_t)gy * a.W + gx;
          float v = acc[m][n][r];
          if (a.bias) v += a.bias[co];
          if (a.residual) v += a.residual[o];
          a.out[o] = v;
        }
      }
    }
  }
}

template <int NCT, int NPT, int WCO, int WPX, int PW, int DMAX, int KC, int KS>
int launch_conv(ConvArgs a, hipStream_t s) {
  using C = ConvCfg<NCT, NPT, WCO, WPX, PW, DMAX, KC, KS>;
  a.tiles_x = (a.W + PW - 1) / PW;
  a.tiles_y = (a.H + C::PH - 1) / C::PH;
  a.co_tiles = (a.Cout + C::CO_T - 1) / C::CO_T;
  const int64_t nblk = (int64_t)a.B * a.tiles_x * a.tiles_y * a.co_tiles;
  if (nblk > 0x7fffffff) return IPDM_EUNSUPPORTED;
  hipLaunchKernelGGL((conv_mfma_kernel<NCT, NPT, WCO, WPX, PW, DMAX, KC, KS>), dim3((unsigned)nblk), dim3(256), 0, s, a);
  return ipdm_launch_status();
}

template <int KS>
int dispatch_conv(const ConvArgs& a, hipStream_t s) {
  const int64_t px = (int64_t)a.B * a.H * a.W;
  if (a.W <= 16) {
    // small images (16x16 stage, dilations 1/2/4): 8x16-pixel tiles
    if (a.Cout >= 512 && px >= 4096) return launch_conv<1, 2, 2, 2, 16, 4, 8, KS>(a, s);   // 64 co x 128 px
    return launch_conv<1, 1, 1, 4, 16, 4, 8, KS>(a, s);                                    // 32 co x 128 px
  }
  if (a.dil > 1) return launch_conv<1, 4, 2, 2, 32, 4, 8, KS>(a, s);                       // dilated, wide images
  if (a.Cout <= 32) return launch_conv<1, 2, 1, 4, 32, 1, 8, KS>(a, s);                    // 32 co x 256 px
  if (a.Cout <= 64 || px < 65536) return launch_conv<1, 4, 2, 2, 32, 1, 8, KS>(a, s);      // 64 co x 256 px
  return launch_conv<4, 2, 1, 4, 32, 1, 8, KS>(a, s);                                      // 128 co x 256 px
}

}  // namespace

extern "C" int ipdm_conv_pack_weight_f32(const float* w, float* wt, int Cout, int Cin, int k, void* stream) {
  IPDM_REQUIRE(w && wt && Cout > 0 && Cin > 0 && (k == 1 || k == 3));
  const int64_t total = (int64_t)Cout * Cin * k * k;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, ipdm_stream(stream), w, wt, Cout,
                     Cin, k * k);
  return ipdm_launch_status();
}

extern "C" int ipdm_conv2d_f32(const float* x, const float* wt, const float* bias, const float* coef, int act,
                               const float* residual, float* out, int B, int Cin, int Cout, int H, int W, int k,
                               int dilation, int pool2, void* stream) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (k == 1 || k == 3) && dilation >= 1);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && wt && out && x != out);
  if (pool2) return IPDM_EUNSUPPORTED;           // ConvMeanPool epilogue: not fused yet (ipdm_meanpool2_f32)
  if (k == 3 && dilation > 4) return IPDM_EUNSUPPORTED;
  ConvArgs a;
  a.x = x; a.wt = wt; a.bias = bias; a.coef = coef; a.residual = residual; a.out = out;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = k == 3 ? dilation : 1; a.act = act;
  a.tiles_x = a.tiles_y = a.co_tiles = 0;
  return k == 3 ? dispatch_conv<3>(a, ipdm_stream(stream)) : dispatch_conv<1>(a, ipdm_stream(stream));
}
