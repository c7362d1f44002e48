// Winograd F(2x2, 3x3) convolution on the bf16 matrix cores with exactly split fp32 operands.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile, 4x4 input tile d, 3x3 filter g
// The 16 position-wise channel contractions  M_p[co, tile] = sum_ci U_p[co,ci] V_p[ci,tile]  run as in conv_bx3.hip:
// U and V are each split into three bf16 pieces and every 32x32x16 product is six v_mfma_f32_32x32x16_bf16 with fp32
// accumulation (fp32-faithful, see conv_bx3.hip).  Winograd cuts the multiply-adds 2.25x, the split costs 6 bf16
// MFMAs at 16x the fp32 MFMA rate: 6x fewer matrix-core cycles than the direct fp32 kernel.  On MI355X the dense
// bf16 loop is power-limited, so fewer MFMAs per output is what buys time.
//
// Workgroup = 64 output channels x 64 tiles (4 tile rows x 16 tiles = 8 x 32 output pixels) of one image, 512 threads.
// Wave w owns positions 2w and 2w+1 for ALL 64 channels x 64 tiles (2 positions x 2 channel tiles x 2 tile groups
// = 128 accumulator registers), so both of its operands are private to it:
//   A (U fragments): pre-split and laid out by ipdm_conv_wino_bx3_pack_weight as [pos][ci/16][co/32][piece][lane]
//        16-byte units; global -> VGPR directly, prefetched one position ahead, used for two tile groups.
//   B (V = B^T d B): the 4x4 patches go global -> regs (buffer loads, hardware zero padding), are transformed in fp32
//        and stored to LDS as fp32 Vs[pos][ci 16][tile 64] (conflict-free 4-byte stores, lane <-> tile); the owning
//        wave reads its 8 channels per lane back and splits them in registers right before the MFMAs.
// Two 64 KiB LDS stages, one barrier per 16-channel chunk.  Epilogue: the accumulators meet through LDS
// (M[pos][co 32][tile 64], one channel tile per round), every thread gathers the 16 positions of its (co, tile)
// pairs, applies A^T M A, bias / residual / activation and stores float2 rows.
//
// Eligible: 3x3, Cin % 16 == 0, Cout % 64 == 0, no fused input normalisation / activation; wide images (W >= 32,
// dilation 1) or the linear-tile-space variant for small / dilated images (polyphase, as conv_wino.hip).
#include "conv_kernel.h"

namespace ipdm_conv {

namespace {

constexpr int X_KC = 16;                 // input channels per chunk (one bf16 MFMA k-step)
constexpr int X_CO = 64;
constexpr int X_TX = 16, X_TY = 4;
constexpr int X_TILES = X_TX * X_TY;     // 64
constexpr int X_V_ELEMS = 16 * X_KC * X_TILES;                   // 16384 floats per stage
// wide images: the raw input region of a chunk ((2*TY+2) x (2*TX+2) pixels x 16 channels) is brought into LDS by
// LDS-DMA (buffer_load ... lds: no VGPRs, hardware zero padding) and the 4x4 patches are read from there
constexpr int X_RR = 2 * X_TY + 2, X_RC = 2 * X_TX + 2;          // 10 x 34
constexpr int X_RCH = 384;                                       // floats per channel (6 wave-instructions of 64 lanes)
constexpr int X_R_ELEMS = 32 * 256;                              // >= 16 x 384 (dword form), 27 quad pieces (DMA4, 8x8 tiles) and
                                                                 // eight wave-private 4 KiB blocks (f16x2 + DMA4)
constexpr size_t X_LDS_BYTES = (2 * (size_t)X_V_ELEMS + X_R_ELEMS) * sizeof(float);   // 128 KiB + 32 KiB = all of a CU's LDS

// U[p = a*4+b][co][ci] = sum_ij G[a][i] g[co][ci][i][j] G[b][j], split into three bf16 pieces, stored as MFMA
// A fragments: [p][cc = ci/16][ct = co/32][piece][h][r][8]  (lane (r, h) holds ci = 16cc + 8h .. +7 of co = 32ct + r)
__global__ __launch_bounds__(256) void wino_bx3_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                              int Cout, int Cin, int n_cc, int n_ct) {
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const int64_t total = (int64_t)16 * n_cc * n_ct * 512;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i & 7), r = (int)((i >> 3) & 31), h = (int)((i >> 8) & 1);
    const int64_t rest = i >> 9;
    const int ct = (int)(rest % n_ct);
    const int cc = (int)((rest / n_ct) % n_cc);
    const int p = (int)(rest / ((int64_t)n_ct * n_cc));
    const int co = ct * 32 + r, ci = cc * 16 + 8 * h + q;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
      const float* g = w + ((size_t)co * Cin + ci) * 9;
      const int a = p >> 2, b = p & 3;
      float t[3];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) t[jj] = G[a][0] * g[jj] + G[a][1] * g[3 + jj] + G[a][2] * g[6 + jj];
      v = t[0] * G[b][0] + t[1] * G[b][1] + t[2] * G[b][2];
    }
    const __bf16 hi = (__bf16)v;
    const float r1 = v - (float)hi;
    const __bf16 mi = (__bf16)r1;
    const __bf16 lo = (__bf16)(r1 - (float)mi);
    const int64_t base = rest * 3 * 512 + h * 256 + r * 8 + q;
    out[base] = __builtin_bit_cast(unsigned short, hi);
    out[base + 512] = __builtin_bit_cast(unsigned short, mi);
    out[base + 1024] = __builtin_bit_cast(unsigned short, lo);
  }
}

// ---- f16x2 weights (conv_kernel.h): U scaled per output channel by a power of two, two fp16 pieces,
//      [p][cc][ct][piece 2][h][r][8], followed by the inverse scales [n_ct * 32] (fp32) ----
__device__ __forceinline__ float wino_U(const float* g, int p) {
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const int a = p >> 2, b = p & 3;
  float t[3];
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) t[jj] = G[a][0] * g[jj] + G[a][1] * g[3 + jj] + G[a][2] * g[6 + jj];
  return t[0] * G[b][0] + t[1] * G[b][1] + t[2] * G[b][2];
}

// one workgroup per output channel: inv_scale[co] = 2^-k with max |U| * 2^k in [2^13, 2^14)
__global__ __launch_bounds__(256) void wino_hx2_scale_kernel(const float* __restrict__ w, float* __restrict__ inv_scale,
                                                             int Cout, int Cin, int n_co_pad) {
  __shared__ float red[256];
  const int co = blockIdx.x;
  float m = 0.f;
  if (co < Cout)
    for (int i = threadIdx.x; i < Cin * 16; i += 256) m = fmaxf(m, fabsf(wino_U(w + ((size_t)co * Cin + i / 16) * 9, i % 16)));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0 && co < n_co_pad) {
    int e = 0;
    const float mx = red[0];
    if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &e);       // mx = f * 2^e, f in [0.5, 1)
    // weight scale 2^(14 - e): mx * scale in [2^13, 2^14); the stored inverse also undoes the kernels' input pre-scale
    inv_scale[co] = (mx > 0.f && mx < INFINITY ? ldexpf(1.f, e - 14) : 1.f) / HX_WINO_PRESCALE;
  }
}

__global__ __launch_bounds__(256) void wino_hx2_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                              const float* __restrict__ inv_scale, int Cout, int Cin, int n_cc,
                                                              int n_ct) {
  const int64_t total = (int64_t)16 * n_cc * n_ct * 512;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i & 7), r = (int)((i >> 3) & 31), h = (int)((i >> 8) & 1);
    const int64_t rest = i >> 9;
    const int ct = (int)(rest % n_ct);
    const int cc = (int)((rest / n_ct) % n_cc);
    const int p = (int)(rest / ((int64_t)n_ct * n_cc));
    const int co = ct * 32 + r, ci = cc * 16 + 8 * h + q;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = wino_U(w + ((size_t)co * Cin + ci) * 9, p) * (1.f / (inv_scale[co] * HX_WINO_PRESCALE));
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const int64_t base = rest * 2 * 512 + h * 256 + r * 8 + q;
    out[base] = __builtin_bit_cast(unsigned short, hi);
    out[base + 512] = __builtin_bit_cast(unsigned short, lo);
  }
}

template <bool HX, bool SMALL>
__global__ __launch_bounds__(512) void conv_wino_bx3_kernel(ConvArgs a) {
  constexpr int NPC = HX ? 2 : 3;                            // operand pieces; FRAG: 16-byte units per (chunk, channel tile)
  constexpr int FRAG = 64 * NPC;
  extern __shared__ __align__(16) float lds[];
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (a.dbg) t0 = __builtin_amdgcn_s_memtime();
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, slot = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  // wide images: the channel tiles of one pixel tile are neighbours (they share the input region through L2);
  // small images: channel-tile-major, so that an XCD works on ONE channel tile's Winograd weights (16 x Cin x 64 x 6
  // bytes, 3 MiB at Cin = 512 -- eight of them would thrash the 4 MiB L2) across all images
  const int n_px = nblk / a.co_tiles;
  const int co_tile = SMALL ? bid / n_px : bid % a.co_tiles;
  int t = SMALL ? bid % n_px : bid / a.co_tiles;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int co0 = co_tile * X_CO;
  const int y0 = ty * (2 * X_TY), x0 = tx * (2 * X_TX);
  // SMALL: 64 consecutive tiles of the image's linear tile space; a dilation-d convolution is d*d undilated ones on
  // the d-subsampled images: tile t = ((sy*d + sx)*THS + tyy)*TWS + txx covers output (d*(2*tyy+i)+sy, d*(2*txx+j)+sx)
  const int d = a.dil;
  const int TWS = SMALL ? a.W / (2 * d) : 1, THS = SMALL ? a.H / (2 * d) : 1;   // (wide form: unused)
  const int tile0 = SMALL ? tx * X_TILES : 0;
  auto tile_origin = [&](int tl, int& py, int& px) {
    const int txx = tl % TWS;
    int r = tl / TWS;
    const int tyy = r % THS;
    r /= THS;
    py = d * (2 * tyy) + r / d;
    px = d * (2 * txx) + r % d;
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int HW = a.H * a.W;
  const int n_cc = a.Cin / X_KC, n_ct = a.Cout / 32;
  const int p0 = 2 * wave;                                   // this wave's positions: p0, p0 + 1

  // ---- staging geometry: this thread transforms tile `mytile` of channels 2*wave, 2*wave+1 of every chunk ----
  const int mytile = tid & 63;
  int row_off[4], col_off[4];                                // byte offsets; 0x20000000 marks padding
  {
    int my_py = 0, my_px = 0;
    bool my_tile_ok = true;
    if constexpr (SMALL) {
      my_tile_ok = tile0 + mytile < THS * TWS * d * d;
      tile_origin(my_tile_ok ? tile0 + mytile : 0, my_py, my_px);
    }
    const int tyl = mytile / X_TX, txl = mytile % X_TX;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int gy = SMALL ? my_py + d * (e - 1) : y0 + 2 * tyl - 1 + e;
      const int gx = SMALL ? my_px + d * (e - 1) : x0 + 2 * txl - 1 + e;
      const bool oky = my_tile_ok && gy >= 0 && gy < a.H, okx = gx >= 0 && gx < a.W;
      row_off[e] = oky ? gy * a.W * 4 : 0x20000000;          // out-of-range sum -> the buffer load returns 0
      col_off[e] = okx ? gx * 4 : 0x20000000;
    }
  }
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((size_t)a.B * a.Cin * HW * 4), 0x00020000);

  // f16x2: input pre-scale (1/4 for the transform's growth, times the image's dynamic scale when a.in_amax is given) and what
  // the epilogue multiplies by besides the weight scale
  [[maybe_unused]] float hx_in = HX_WINO_PRESCALE, hx_out = 1.f;
  if constexpr (HX) {
    if (a.in_amax) {
      float sd, si;
      hx_dynamic_scale(ipdm_amax_read_v(a.in_amax + (size_t)b * IPDM_AMAX_SLOT), sd, si);
      hx_in = HX_WINO_PRESCALE * sd;
      hx_out = si;
    }
  }
  float dreg[16];
  auto load_patch = [&](int ci) {                            // ci: absolute input channel (wave-uniform)
    const int soff = (int)(((size_t)b * a.Cin + ci) * HW * 4);
#pragma unroll
    for (int e = 0; e < 16; ++e)
      dreg[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, row_off[e / 4] + col_off[e % 4], soff, 0));
  };
  // B^T d B of the patch in dreg -> Vs[pos][kc][mytile] of stage st
  auto store_patch = [&](float* st, int kc) {
    const float(&dd)[16] = dreg;
    float tmp[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      tmp[0 * 4 + c] = dd[0 * 4 + c] - dd[2 * 4 + c];
      tmp[1 * 4 + c] = dd[1 * 4 + c] + dd[2 * 4 + c];
      tmp[2 * 4 + c] = dd[2 * 4 + c] - dd[1 * 4 + c];
      tmp[3 * 4 + c] = dd[1 * 4 + c] - dd[3 * 4 + c];
    }
    if constexpr (HX) {
      // f16x2 stage: Vs[pos][piece 2][channel pair 8][tile 64] 32-bit words, channel 2c in the low half of a word: the
      // reader's four words per piece ARE its MFMA operand.  This kernel stages one channel at a time: 16-bit stores.
      _Float16* vs = reinterpret_cast<_Float16*>(st) + ((kc >> 1) * X_TILES + mytile) * 2 + (kc & 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v4[4] = {tmp[r * 4 + 0] - tmp[r * 4 + 2], tmp[r * 4 + 1] + tmp[r * 4 + 2], tmp[r * 4 + 2] - tmp[r * 4 + 1],
                             tmp[r * 4 + 1] - tmp[r * 4 + 3]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float vsc = v4[c] * hx_in;                    // exact (power of two); undone in the epilogue
          const _Float16 hi = (_Float16)vsc;
          const _Float16 lo = (_Float16)(vsc - (float)hi);
          vs[((r * 4 + c) * 2 + 0) * (X_KC * X_TILES)] = hi;
          vs[((r * 4 + c) * 2 + 1) * (X_KC * X_TILES)] = lo;
        }
      }
    } else {
    float* vs = st + kc * X_TILES + mytile;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vs[(r * 4 + 0) * X_KC * X_TILES] = tmp[r * 4 + 0] - tmp[r * 4 + 2];
      vs[(r * 4 + 1) * X_KC * X_TILES] = tmp[r * 4 + 1] + tmp[r * 4 + 2];
      vs[(r * 4 + 2) * X_KC * X_TILES] = tmp[r * 4 + 2] - tmp[r * 4 + 1];
      vs[(r * 4 + 3) * X_KC * X_TILES] = tmp[r * 4 + 1] - tmp[r * 4 + 3];
    }
    }
  };

  // ---- wide images: raw region via LDS-DMA; wave w brings in channels 2w, 2w+1 (6 x 64 floats each) ----
  float* const rs = lds + 2 * X_V_ELEMS;
  int dma_off[6];
  if constexpr (!SMALL) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int e = k * 64 + lane;
      const int r = e / X_RC, c = e - r * X_RC;
      const int gy = y0 - 1 + r, gx = x0 - 1 + c;
      const bool ok = e < X_RR * X_RC && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      dma_off[k] = ok ? (gy * a.W + gx) * 4 : 0x40000000;      // out of range -> the DMA writes 0
    }
  }
  auto issue_dma = [&](int chunk) {
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const int kc = 2 * wave + cl;
      const int soff = (int)(((size_t)b * a.Cin + chunk * X_KC + kc) * HW * 4);
#pragma unroll
      for (int k = 0; k < 6; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(rs + kc * X_RCH + k * 64), 4,
                                                 dma_off[k], soff, 0, 0);
    }
  };
  const int r_lane = (2 * (mytile / X_TX)) * X_RC + 2 * (mytile % X_TX);
  auto read_patch = [&](int kc) {                              // 4x4 patch of channel kc (chunk-local) from the raw stage
    const float* rp = rs + kc * X_RCH + r_lane;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float2 lo = *reinterpret_cast<const float2*>(rp + r * X_RC);
      const float2 hi = *reinterpret_cast<const float2*>(rp + r * X_RC + 2);
      dreg[r * 4 + 0] = lo.x; dreg[r * 4 + 1] = lo.y; dreg[r * 4 + 2] = hi.x; dreg[r * 4 + 3] = hi.y;
    }
  };

  // ---- A fragments ----
  const uint4* wq = reinterpret_cast<const uint4*>(a.wt);
  const size_t pos_stride = (size_t)n_cc * n_ct * FRAG;
  const int a_lane = (co_tile * 2) * FRAG + lane;            // channel tile c adds FRAG
  auto load_A = [&](uint4 (&fr)[2][NPC], int p, int cc) {
    const uint4* base = wq + (size_t)p * pos_stride + (size_t)cc * n_ct * FRAG + a_lane;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int s = 0; s < NPC; ++s) fr[c][s] = base[c * FRAG + s * 64];
  };
  // ---- B operand: 8 channels (8h .. 8h+7) of tile (tg*32 + j) at position p, fp32 ----
  const int b_lane = (8 * h) * X_TILES + j;
  auto load_B = [&](float (&raw)[8], const float* cur, int pi, int tg) {
    const float* bp = cur + (p0 + pi) * (X_KC * X_TILES) + tg * 32 + b_lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) raw[q] = bp[q * X_TILES];
  };
  // f16x2: channel pairs 4h .. 4h+3 of both pieces -- the operand itself, no split
  auto load_Bh = [&](uint4 (&fr)[2], const float* cur, int pi, int tg) {
    const unsigned* bp = reinterpret_cast<const unsigned*>(cur) + (p0 + pi) * (X_KC * X_TILES) + (4 * h) * X_TILES + tg * 32 + j;
#pragma unroll
    for (int s = 0; s < 2; ++s)
      fr[s] = make_uint4(bp[s * 8 * X_TILES], bp[(s * 8 + 1) * X_TILES], bp[(s * 8 + 2) * X_TILES], bp[(s * 8 + 3) * X_TILES]);
  };

  f32x16 acc[2][2][2];                                       // [position][channel tile][tile group]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i >> 2][(i >> 1) & 1][i & 1][r] = 0.f;

  uint4 afr[2][2][NPC];
  const int n_chunks = n_cc;
  // prologue: stage 0 completely, and channel 2w of chunk 1 already in flight in dreg
  load_A(afr[0], p0, 0);
  if constexpr (SMALL) {
    load_patch(2 * wave);
    store_patch(lds, 2 * wave);
    load_patch(2 * wave + 1);
    store_patch(lds, 2 * wave + 1);
    load_patch((n_chunks > 1 ? X_KC : 0) + 2 * wave);
    __syncthreads();
  } else {
    issue_dma(0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    read_patch(2 * wave);
    store_patch(lds, 2 * wave);
    read_patch(2 * wave + 1);
    store_patch(lds, 2 * wave + 1);
    __syncthreads();                                           // raw stage free again
    issue_dma(n_chunks > 1 ? 1 : 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
  if (a.dbg) t1 = __builtin_amdgcn_s_memtime();

  // Steady state, branch-free.  Per chunk a wave runs four steps (position, tile group) of 12 MFMAs each; the side
  // work is dealt into the MFMA shadows with sched_group_barrier:
  //   every step : the LDS reads of the NEXT step's B operand, then its three-way split (VALU)
  //   step 0     : transform + store channel 2w of chunk c+1 (in flight since step 2 of chunk c-1), issue the loads
  //                of channel 2w+1; issue the A fragments of position p0+1
  //   step 2     : transform + store channel 2w+1 of chunk c+1, issue the loads of channel 2w of chunk c+2; issue
  //                the A fragments of position p0 of chunk c+1
  // (past the last chunk the channel index is clamped: redundant loads, stores into a stage nobody reads)
  for (int ch = 0; ch < n_chunks; ++ch) {
    const float* cur = lds + (ch & 1) * X_V_ELEMS;
    float* nxt = lds + ((ch + 1) & 1) * X_V_ELEMS;
    const int ch1 = ch + 1 < n_chunks ? ch + 1 : n_chunks - 1;
    const int ch2 = ch + 2 < n_chunks ? ch + 2 : n_chunks - 1;
    bf16x8 bs[2][3];
    uint4 bsh[2][2];
    float raw[8];
    if constexpr (HX) {
      load_Bh(bsh[0], cur, 0, 0);
    } else {
      load_B(raw, cur, 0, 0);
      split3(raw, bs[0][0], bs[0][1], bs[0][2]);
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<4>([&](auto sc) {
      constexpr int st = decltype(sc)::value;
      constexpr int pi = st >> 1, tg = st & 1;
      if constexpr (st < 3) {
        if constexpr (HX) load_Bh(bsh[(st + 1) & 1], cur, (st + 1) >> 1, (st + 1) & 1);
        else load_B(raw, cur, (st + 1) >> 1, (st + 1) & 1);
      }
      if constexpr (SMALL) {
        if constexpr (st == 0) {
          store_patch(nxt, 2 * wave);
          load_patch(ch1 * X_KC + 2 * wave + 1);
          load_A(afr[1], p0 + 1, ch);
        }
        if constexpr (st == 2) {
          store_patch(nxt, 2 * wave + 1);
          load_patch(ch2 * X_KC + 2 * wave);
          load_A(afr[0], p0, ch1);
        }
      } else {
        // the raw stage holds chunk c+1: transform it in step 0, refill it with chunk c+2 from step 1 on (three
        // steps for the DMA to land before the end-of-chunk barrier)
        if constexpr (st == 0) {
          read_patch(2 * wave);
          store_patch(nxt, 2 * wave);
          read_patch(2 * wave + 1);
          store_patch(nxt, 2 * wave + 1);
          load_A(afr[1], p0 + 1, ch);
        }
        if constexpr (st == 1) issue_dma(ch2);
        if constexpr (st == 2) load_A(afr[0], p0, ch1);
      }
      constexpr bool XFORM = SMALL ? (st == 0 || st == 2) : st == 0;
      constexpr bool VMEM = SMALL ? (st == 0 || st == 2) : st < 3;
      if constexpr (HX) {
        const f16x8 bh = __builtin_bit_cast(f16x8, bsh[st & 1][0]), bl = __builtin_bit_cast(f16x8, bsh[st & 1][1]);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f32x16 v = acc[pi][c][tg];
          v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][1]), bh, v, 0, 0, 0);
          v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][0]), bl, v, 0, 0, 0);
          v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][0]), bh, v, 0, 0, 0);
          acc[pi][c][tg] = v;
        }
        if constexpr (st < 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);        // next step's LDS reads first
        if constexpr (!SMALL && XFORM) __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, XFORM ? (SMALL ? 20 : 40) : 3, 0);
          if constexpr (XFORM) __builtin_amdgcn_sched_group_barrier(0x200, SMALL ? 6 : 11, 0);
          if constexpr (VMEM) __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
        }
      } else {
      if constexpr (st < 3) split3(raw, bs[(st + 1) & 1][0], bs[(st + 1) & 1][1], bs[(st + 1) & 1][2]);
      const bf16x8 bh = bs[st & 1][0], bm = bs[st & 1][1], bl = bs[st & 1][2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f32x16 v = acc[pi][c][tg];
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][2]), bh, v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][0]), bl, v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][1]), bm, v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][1]), bh, v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][0]), bm, v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][0]), bh, v, 0, 0, 0);
        acc[pi][c][tg] = v;
      }
      if constexpr (st < 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);          // next step's LDS reads first
      if constexpr (!SMALL && XFORM) __builtin_amdgcn_sched_group_barrier(0x100, 16, 0); // the patch reads
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x006, XFORM ? (SMALL ? 10 : 12) : 5, 0);  // VALU / SALU in its shadow
        if constexpr (XFORM) __builtin_amdgcn_sched_group_barrier(0x200, SMALL ? 2 : 3, 0);   // patch stores
        if constexpr (VMEM) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);          // patch / fragment loads, DMA
      }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!SMALL && st == 0) __syncthreads();        // everyone has read the raw stage
    });
    if constexpr (!SMALL) __builtin_amdgcn_s_waitcnt(0);       // the DMA of chunk c+2 has landed
    __syncthreads();
  }
  if (a.dbg) t2 = __builtin_amdgcn_s_memtime();

  // ---- epilogue: M -> LDS (one channel tile per round), gather 16 positions per (co, tile), A^T M A ----
  const int tile = tid & 63;
  const int cg = tid >> 6;                                   // channels 4*cg .. 4*cg+3 of the round's 32
  int oy = y0 + 2 * (tile / X_TX), ox = x0 + 2 * (tile % X_TX);
  bool out_ok = oy < a.H && ox < a.W;
  if constexpr (SMALL) {
    out_ok = tile0 + tile < THS * TWS * d * d;
    tile_origin(out_ok ? tile0 + tile : 0, oy, ox);
  }
  // bias and residual of BOTH rounds are fetched here, before the exchange (unconditional loads: an absent operand or an
  // out-of-range tile reads the weight blob instead, so hipcc can count them); next to their use every one of them
  // exposed its full latency, with nothing else resident on the CU to cover it
  const bool has_res = a.residual != nullptr, has_bias = a.bias != nullptr;
  const float* const res_p = has_res ? a.residual : a.wt;
  const float* const bias_p = has_bias ? a.bias : a.wt;
  const bool res_ok = has_res && out_ok;
  const int rd = res_ok ? d : 0;
  const float* const scale_p = HX ? reinterpret_cast<const float*>(wq + 16 * pos_stride) : a.wt;   // f16x2: inverse scales
  float bv[2][4], sv[2][4];
  float2 rv[2][4][2];
  float amx_o = 0.f, amx_a = 0.f;                             // max |stored value| of this thread (a.amax_out / a.amax_act)
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = co0 + c * 32 + cg * 4 + i;
      bv[c][i] = bias_p[has_bias ? b * a.bias_bstride + co : 0];
      sv[c][i] = scale_p[HX ? co : 0] * hx_out;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const size_t o = res_ok ? ((size_t)b * a.Cout + co) * HW + (size_t)(oy + (SMALL ? d * ii : ii)) * a.W + ox : 0;
        if constexpr (SMALL) rv[c][i][ii] = make_float2(res_p[o], res_p[o + rd]);
        else rv[c][i][ii] = *reinterpret_cast<const float2*>(res_p + o);
      }
    }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c > 0) __syncthreads();                              // the previous round's reads are done
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int tg = 0; tg < 2; ++tg)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int col = (r & 3) + 8 * (r >> 2) + 4 * h;
          lds[((p0 + pi) * 32 + col) * X_TILES + tg * 32 + j] = acc[pi][c][tg][r];
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {                            // the one counted wait for this round's loads sits here
      asm volatile("" : "+v"(bv[c][i]), "+v"(sv[c][i]));
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) asm volatile("" : "+v"(rv[c][i][ii].x), "+v"(rv[c][i][ii].y));
    }
    if (out_ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cl = cg * 4 + i;
        const int co = co0 + c * 32 + cl;
        float m[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) m[p] = lds[(p * 32 + cl) * X_TILES + tile];
        float tt[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          tt[0][q] = m[0 * 4 + q] + m[1 * 4 + q] + m[2 * 4 + q];
          tt[1][q] = m[1 * 4 + q] - m[2 * 4 + q] - m[3 * 4 + q];
        }
        const float bias = has_bias ? bv[c][i] : 0.f;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          float y0v = tt[ii][0] + tt[ii][1] + tt[ii][2], y1v = tt[ii][1] - tt[ii][2] - tt[ii][3];
          if constexpr (HX) {
            y0v = __builtin_fmaf(y0v, sv[c][i], bias);
            y1v = __builtin_fmaf(y1v, sv[c][i], bias);
          } else {
            y0v += bias;
            y1v += bias;
          }
          const size_t o = ((size_t)b * a.Cout + co) * HW + (size_t)(oy + (SMALL ? d * ii : ii)) * a.W + ox;
          float r0v = y0v, r1v = y1v;                         // what `out` stores (a.res_second: without the residual)
          if (has_res) {
            y0v += rv[c][i][ii].x;
            y1v += rv[c][i][ii].y;
          }
          r0v = a.res_second ? r0v : y0v;
          r1v = a.res_second ? r1v : y1v;
          y0v *= a.out_scale;
          y1v *= a.out_scale;
          r0v *= a.out_scale;
          r1v *= a.out_scale;
          amx_o = fmaxf(amx_o, fmaxf(fabsf(r0v), fabsf(r1v)));
          if constexpr (SMALL) {
            if (a.out) {
              a.out[o] = r0v;
              a.out[o + d] = r1v;
            }
            if (a.out_act) {
              const float e0 = a.act_out == IPDM_ACT_ELU ? fast_elu(y0v) : ipdm_act(y0v, a.act_out);
              const float e1 = a.act_out == IPDM_ACT_ELU ? fast_elu(y1v) : ipdm_act(y1v, a.act_out);
              amx_a = fmaxf(amx_a, fmaxf(fabsf(e0), fabsf(e1)));
              a.out_act[o] = e0;
              a.out_act[o + d] = e1;
            }
          } else {
            if (a.out) *reinterpret_cast<float2*>(a.out + o) = make_float2(r0v, r1v);
            if (a.out_act) {
              const float e0 = a.act_out == IPDM_ACT_ELU ? fast_elu(y0v) : ipdm_act(y0v, a.act_out);
              const float e1 = a.act_out == IPDM_ACT_ELU ? fast_elu(y1v) : ipdm_act(y1v, a.act_out);
              amx_a = fmaxf(amx_a, fmaxf(fabsf(e0), fabsf(e1)));
              *reinterpret_cast<float2*>(a.out_act + o) = make_float2(e0, e1);
            }
          }
        }
      }
    }
  }
  if (a.amax_out) ipdm_amax_commit(amx_o, a.amax_out + (size_t)b * IPDM_AMAX_SLOT, wave);
  if (a.amax_act) ipdm_amax_commit(amx_a, a.amax_act + (size_t)b * IPDM_AMAX_SLOT, wave);
  if (a.dbg) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* d4 = a.dbg + (size_t)blockIdx.x * 4;
      d4[0] = t0; d4[1] = t1; d4[2] = t2; d4[3] = t3;
    }
  }
}


// ---- wide images, persistent form -----------------------------------------------------------------------------------
// One workgroup per CU walks a list of (pixel tile, channel tile) pairs and never drains its staging pipeline: while
// the last chunk of tile i is multiplied, chunk 0 of tile i+1 is transformed into the other V stage and its chunk 1
// is brought into the raw stage, so only the FIRST tile of a workgroup pays the prologue (two dependent DMA round
// trips, ~13 % of a 128-channel tile).  The epilogue therefore has to live in ONE 64 KiB stage: four rounds of
// (channel tile, tile group) through M[pos 16][co 32][tile 32].
// Tiles are dealt XCD-aware: XCD x owns a contiguous range of the linear tile order (channel tile fastest, so the
// channel tiles of a pixel tile meet in that XCD's L2) and its workgroups stride through it.
// TX x TY tiles per workgroup: 16 x 4 (8 x 32 output pixels) for wide images, 8 x 8 (16 x 16) for 16-pixel images, whose
// workgroups are dealt channel-tile-major (CO_MAJOR: an XCD keeps ONE channel tile's 3 MiB of Winograd weights in L2).
// POOL: the ConvMeanPool epilogue (2x2 mean of every output tile) as its own instantiation, so that the unpooled kernel's
// register allocation is exactly what it was (a run-time branch cost it four spilled registers and ~7 % more VALU).
// DMA4 (W % 4 == 0, 16-byte aligned tensor): the raw region is fetched as 16-byte quads aligned to multiples of four
// pixels (a quad is then entirely inside or entirely outside the image, so the range check still pads), 25-27
// wave-instructions per chunk and workgroup instead of 96 -- an LDS-DMA instruction costs ~100 issue cycles next to
// MFMAs, whatever its width.
// STATS: the epilogue also reduces every (channel, tile group) of the workgroup's tile block to (count, mean, sum of squared
// deviations) of the stored result and writes them to a.stats (own instantiation, same reason as POOL).
// KSP: the K loop of a tile is dealt to a.ksplit workgroups (each a contiguous range of n_cc / ksplit chunks); every part
// writes its raw A^T M A to its own plane set of a.out (= the caller's partial-sum workspace [ksplit][B][Cout][H][W]) and
// bx3_splitk_reduce_kernel adds the parts in fixed order with bias / residual / activation.  For 16-pixel layers whose
// (image, channel tile) pairs alone leave most of the chip idle.
// POLY (16 x 16 images, any dilation d): the workgroup's 64 tiles are the image's d*d polyphase sub-images (a dilation-d
// convolution is d*d undilated ones on the d-subsampled images; tile t = ((sy*d + sx)*THS + tyy)*TWS + txx covers outputs
// (d*(2 tyy + i) + sy, d*(2 txx + j) + sx)).  The raw stage holds the WHOLE image of a chunk's 16 channels, padded: 17 rows
// (the last one all zero: where out-of-image patch rows point) of 24 floats (four zero columns on either side), filled by
// 26 16-byte LDS-DMA pieces per chunk whose out-of-image quads the range check zeroes; a thread gathers its 4 x 4 patch
// from there (16 LDS reads) instead of 16 scattered 4-byte global loads per patch -- the address path was what held the
// register-staged kernel at 12.7 k cycles per chunk on dilation-2 layers.
// CO32: 32 output channels per workgroup (one channel tile per wave instead of two: half the accumulators) -- for 16-pixel layers with
// fewer than 512 output channels, whose (image, 64-channel tile) pairs fill under half of the chip: twice the workgroups, the whole K
// loop in each, instead of two K halves + a reduction launch
template <bool HX, int TX, int TY, bool CO_MAJOR, bool DMA4, bool POOL = false, bool STATS = false, bool KSP = false, bool POLY = false,
          bool CO32 = false>
__global__ __launch_bounds__(512) void conv_wino_bx3_wide_kernel(ConvArgs a, int total_tiles) {
  constexpr int NPC = HX ? 2 : 3;                            // operand pieces; FRAG: 16-byte units per (chunk, channel tile)
  constexpr int FRAG = 64 * NPC;
  constexpr int NC = CO32 ? 1 : 2;                           // 32-channel tiles per workgroup
  static_assert(!CO32 || (HX && TX == 8 && TY == 8 && DMA4 && !POOL && !STATS && !KSP && !POLY), "CO32: the plain 8 x 8 form");
  static_assert(!POLY || (TX == 8 && TY == 8 && DMA4 && !POOL && !STATS && !KSP), "POLY: the 8 x 8 form, plain epilogue");
  constexpr int PPW = 24, PPH = 17;                           // POLY: padded row pitch / rows per channel
  constexpr int QC = POLY ? PPW / 4 : TX / 2 + 2, RC4 = 4 * QC;   // quads / floats per raw row (x0-4 .. x0+2TX+3)
  constexpr int QN = POLY ? PPH * QC : (2 * TY + 2) * QC;     // quads per channel
  constexpr int NI = (X_KC * QN + 63) / 64;                   // wave-instructions per chunk
  static_assert(NI >= 24 && NI <= 32 && NI * 256 <= X_R_ELEMS, "quad image fits the raw stage; pieces 0..23 exist");
  // f16x2 + 16-byte DMA: the raw stage is WAVE-PRIVATE -- wave w fetches exactly the two channels (2w, 2w+1) whose patches its
  // own threads transform, into its own 4 KiB block (2 * QN quads <= 256 = four wave-instructions).  No other wave ever reads
  // that block, so the raw data needs no barrier at all: a wave waits for ITS DMA with a counted vmcnt, reads its patches into
  // registers and re-issues the DMA of the chunk after -- a whole chunk ahead of its use; the one barrier left per chunk is
  // the V stage's.  (Shared raw stage: barrier -> DMA -> landing -> barrier -> patch reads -> barrier was the critical path,
  // 4.1 k cycles per chunk against 1.5 k of matrix work.)
  constexpr bool WPRIV = HX && DMA4;
  static_assert(!WPRIV || 2 * QN <= 256, "a wave's two channels fit four DMA instructions");
  static_assert(TX * TY == X_TILES, "64 tiles per workgroup");
  constexpr int RC = 2 * TX + 2, RR = 2 * TY + 2;            // raw region: 34 x 10 or 18 x 18 pixels (<= X_RCH)
  static_assert(RC * RR <= X_RCH, "raw region fits its LDS slot");
  extern __shared__ __align__(16) float lds[];
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (a.dbg) t0 = __builtin_amdgcn_s_memtime();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int HW = a.H * a.W;
  const int n_cc = a.Cin / X_KC, n_ct = a.Cout / 32;
  const int n_chunks = KSP ? n_cc / a.ksplit : n_cc;           // >= 2 (launcher)
  const int p0 = 2 * wave;

  // ---- this workgroup's tile list: first, stride, end (linear tile order, channel tile fastest) ----
  const int S = gridDim.x / 8;                                // workgroups per XCD (grid is a multiple of 8)
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const int q = total_tiles / 8, r8 = total_tiles % 8;
  const int x_start = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
  const int x_end = x_start + q + (xcd < r8 ? 1 : 0);
  // -DIPDM_WBX3_RUNS (tuning build): every workgroup walks a CONTIGUOUS run of the XCD's range (it then stays inside one or two images: fewer maxima
  // commits and scale reloads) instead of striding through it with the XCD's other workgroups
#ifdef IPDM_WBX3_RUNS
  constexpr bool RUNS = true;
#else
  constexpr bool RUNS = false;
#endif
  const int per_wg = (x_end - x_start + S - 1) / S;
  const int t_step = RUNS ? 1 : S;
  const int t_end = RUNS ? (x_start + (slot + 1) * per_wg < x_end ? x_start + (slot + 1) * per_wg : x_end) : x_end;
  int tile = RUNS ? x_start + slot * per_wg : x_start + slot;
  if (tile >= t_end) return;                                  // uniform: whole workgroup

  struct Geo { int b, y0, x0, co_tile, c0, ks; };               // c0: first chunk of this workgroup's K range (KSP)
  auto geo_of = [&](int L) {
    Geo g;
    g.ks = 0;
    g.c0 = 0;
    int n_px = total_tiles / a.co_tiles;
    if constexpr (KSP) {                                        // K part fastest: the parts of a tile share its input in L2
      g.ks = L % a.ksplit;
      L /= a.ksplit;
      n_px /= a.ksplit;
      g.c0 = g.ks * n_chunks;
    }
    g.co_tile = CO_MAJOR ? L / n_px : L % a.co_tiles;
    int t = CO_MAJOR ? L % n_px : L / a.co_tiles;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    g.y0 = (t % a.tiles_y) * (2 * TY);
    g.x0 = tx * (2 * TX);
    g.b = t / a.tiles_y;
    return g;
  };

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((size_t)a.B * a.Cin * HW * 4), 0x00020000);
  float* const rs = lds + 2 * X_V_ELEMS;
  const int mytile = tid & 63;

  // raw stage by LDS-DMA.  dword form: wave w brings in channels 2w, 2w+1 (6 x 64 floats each); quad form: the
  // chunk's 16 channels are one packed image of NI x 64 quads, wave w issues pieces w, w+8, w+16, w+24
  int dma_off[DMA4 ? 4 : 6];
  int dma_b = 0;                                              // image index the offsets belong to
  [[maybe_unused]] int dma_c0 = 0;                            // ... and the first chunk of its K range
  // every wave issues exactly four pieces per chunk (a wave without a k-th piece repeats its previous one: same bytes to
  // the same place), so that the number of DMA instructions in flight is a compile-time constant: with a wave-uniform
  // branch around them hipcc cannot count, and waits for vmcnt(0) -- i.e. for the DMA -- at the next fragment use
  auto dma_piece = [&](int k) { return wave + 8 * k < NI ? wave + 8 * k : wave + 8 * (k - 1); };
  auto set_dma_geo = [&](const Geo& g) {
    dma_b = g.b;
    if constexpr (KSP) dma_c0 = g.c0;
    if constexpr (DMA4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = (WPRIV ? k : dma_piece(k)) * 64 + lane;
        const int cl = e / QN, qq = e - cl * QN;                // WPRIV: cl = channel of the wave's pair (>= 2: padding lanes,
        const int cin = WPRIV ? 2 * wave + cl : cl;             //        which write zeros into the tail of the wave's own block)
        const int rr = qq / QC, qc = qq - rr * QC;
        const int gy = POLY ? (rr < 16 ? rr : -1) : g.y0 - 1 + rr, gx0 = g.x0 - 4 + 4 * qc;
        const bool ok = e < (WPRIV ? 2 : X_KC) * QN && gy >= 0 && gy < a.H && gx0 >= 0 && gx0 < a.W;
        dma_off[k] = ok ? (cin * HW + gy * a.W + gx0) * 4 : 0x40000000;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int e = k * 64 + lane;
        const int rr = e / RC, c = e - rr * RC;
        const int gy = g.y0 - 1 + rr, gx = g.x0 - 1 + c;
        const bool ok = e < RR * RC && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        dma_off[k] = ok ? (gy * a.W + gx) * 4 : 0x40000000;
      }
    }
  };
  auto issue_dma = [&](int chunk) {
    if constexpr (KSP) chunk += dma_c0;
    if constexpr (DMA4) {
      [[maybe_unused]] const int soff = (int)(((size_t)dma_b * a.Cin + chunk * X_KC) * HW * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)   // the 16-byte form only exists for gfx950: keep it out of the host pass
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            x_rsrc, (__attribute__((address_space(3))) void*)(rs + (WPRIV ? wave * 4 + k : dma_piece(k)) * 256), 16, dma_off[k],
            soff, 0, 0);
#endif
      }
    } else {
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        const int kc = 2 * wave + cl;
        const int soff = (int)(((size_t)dma_b * a.Cin + chunk * X_KC + kc) * HW * 4);
#pragma unroll
        for (int k = 0; k < 6; ++k)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(rs + kc * X_RCH + k * 64), 4,
                                                   dma_off[k], soff, 0, 0);
      }
    }
  };
  float dreg[16];
  // POLY: this thread's tile in the polyphase tile space, and the padded-image offsets of its patch rows / columns
  const int pd = POLY ? a.dil : 1;
  auto tile_origin = [&](int tl, int& py, int& px) {
    const int tws = a.W / (2 * pd), ths = a.H / (2 * pd);
    const int txx = tl % tws;
    int r = tl / tws;
    const int tyy = r % ths;
    r /= ths;
    py = pd * (2 * tyy) + r / pd;
    px = pd * (2 * txx) + r % pd;
  };
  [[maybe_unused]] int prow[4], pcol[4];
  if constexpr (POLY) {
    int py, px;
    tile_origin(mytile, py, px);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int gy = py + pd * (e - 1), gx = px + pd * (e - 1);
      prow[e] = (gy >= 0 && gy < a.H ? gy : 16) * PPW;        // row 16: zeros
      pcol[e] = gx >= 0 && gx < a.W ? gx + 4 : 0;             // column 0: zero
    }
  }
  const int r_lane = DMA4 ? (2 * (mytile / TX)) * RC4 + 2 * (mytile % TX) + 2 : (2 * (mytile / TX)) * RC + 2 * (mytile % TX);
  auto read_patch_to = [&](float (&dreg)[16], int kc) {
    if constexpr (POLY) {
      const float* rp = WPRIV ? rs + wave * 1024 + (kc & 1) * (QN * 4) : rs + kc * (QN * 4);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) dreg[rr * 4 + cc] = rp[prow[rr] + pcol[cc]];
    } else if constexpr (DMA4) {
      // patch columns 2txl+3 .. 2txl+6 of the 4-aligned rows.  A lane reads only its own aligned pair (2txl+4, 2txl+5);
      // column 2txl+3 is the left neighbour tile's second element and 2txl+6 the right neighbour's first (lanes of a
      // 16-lane DPP row are consecutive tiles of a tile row), the two tiles at the ends of a tile row read theirs:
      // 8 instead of 12 LDS instructions and half the LDS bytes per patch.
      const float* rp = (WPRIV ? rs + wave * 1024 + (kc & 1) * (QN * 4) : rs + kc * (QN * 4)) + r_lane;   // -> column 2txl+2
      const int txl = mytile % TX;
      const bool first = txl == 0, last = txl == TX - 1;
      const int edge = first ? 1 : 4;                         // column 2txl+3 (first) / 2txl+6 (last); others: unused
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float2 q1 = *reinterpret_cast<const float2*>(rp + rr * RC4 + 2);
        const float e = rp[rr * RC4 + edge];
        // (bound_ctrl form, no `old` operand: the row-end lanes, whose source lane does not exist, take `e` below -- with an
        //  old value hipcc materialises a zero per DPP move and cannot fold the move into the select)
        const float left = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q1.y), 0x111, 0xf, 0xf, true));
        const float right = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q1.x), 0x101, 0xf, 0xf, true));
        dreg[rr * 4 + 0] = first ? e : left;
        dreg[rr * 4 + 1] = q1.x;
        dreg[rr * 4 + 2] = q1.y;
        dreg[rr * 4 + 3] = last ? e : right;
      }
    } else {
      const float* rp = rs + kc * X_RCH + r_lane;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float2 lo = *reinterpret_cast<const float2*>(rp + rr * RC);
        const float2 hi = *reinterpret_cast<const float2*>(rp + rr * RC + 2);
        dreg[rr * 4 + 0] = lo.x; dreg[rr * 4 + 1] = lo.y; dreg[rr * 4 + 2] = hi.x; dreg[rr * 4 + 3] = hi.y;
      }
    }
  };
  auto read_patch = [&](int kc) { read_patch_to(dreg, kc); };
  auto store_patch = [&](float* st, int kc) {
    const float(&dd)[16] = dreg;
    float tmp[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      tmp[0 * 4 + c] = dd[0 * 4 + c] - dd[2 * 4 + c];
      tmp[1 * 4 + c] = dd[1 * 4 + c] + dd[2 * 4 + c];
      tmp[2 * 4 + c] = dd[2 * 4 + c] - dd[1 * 4 + c];
      tmp[3 * 4 + c] = dd[1 * 4 + c] - dd[3 * 4 + c];
    }
    float* vs = st + kc * X_TILES + mytile;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      vs[(rr * 4 + 0) * X_KC * X_TILES] = tmp[rr * 4 + 0] - tmp[rr * 4 + 2];
      vs[(rr * 4 + 1) * X_KC * X_TILES] = tmp[rr * 4 + 1] + tmp[rr * 4 + 2];
      vs[(rr * 4 + 2) * X_KC * X_TILES] = tmp[rr * 4 + 2] - tmp[rr * 4 + 1];
      vs[(rr * 4 + 3) * X_KC * X_TILES] = tmp[rr * 4 + 1] - tmp[rr * 4 + 3];
    }
  };

  // f16x2: B^T d B of the patch in dreg -> v (registers); the two channels 2w, 2w+1 of a tile are then split together and
  // stored as packed fp16 pairs, Vs[pos][piece 2][channel pair 8][tile 64] 32-bit words (channel 2c in the low half): the
  // reader's four words per piece ARE its MFMA operand -- the split is done once, by the writer, at 2 VALU per element
  auto transform = [&](float (&v)[16]) {                      // in place: B^T v B
    const float(&dd)[16] = v;
    float tmp[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      tmp[0 * 4 + c] = dd[0 * 4 + c] - dd[2 * 4 + c];
      tmp[1 * 4 + c] = dd[1 * 4 + c] + dd[2 * 4 + c];
      tmp[2 * 4 + c] = dd[2 * 4 + c] - dd[1 * 4 + c];
      tmp[3 * 4 + c] = dd[1 * 4 + c] - dd[3 * 4 + c];
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      v[rr * 4 + 0] = tmp[rr * 4 + 0] - tmp[rr * 4 + 2];
      v[rr * 4 + 1] = tmp[rr * 4 + 1] + tmp[rr * 4 + 2];
      v[rr * 4 + 2] = tmp[rr * 4 + 2] - tmp[rr * 4 + 1];
      v[rr * 4 + 3] = tmp[rr * 4 + 1] - tmp[rr * 4 + 3];
    }
  };
  // hx_in: the input pre-scale of the image whose chunk is being staged (1/4 for the transform's growth x the image's dynamic
  // scale, or just 1/4); hx_out: what the epilogue multiplies the current tile's result by besides the weight scale
  [[maybe_unused]] float hx_in = HX_WINO_PRESCALE, hx_out = 1.f;
  auto hx_scales_of = [&](int b, float& s_in, float& s_out) {
    s_in = HX_WINO_PRESCALE;
    s_out = 1.f;
    if constexpr (HX) {
      if (a.in_amax) {
        float sd, si;
        hx_dynamic_scale(ipdm_amax_read(a.in_amax + (size_t)b * IPDM_AMAX_SLOT), sd, si);
        s_in = HX_WINO_PRESCALE * sd;
        s_out = si;
      }
    }
  };
  auto store_pair = [&](float* st, const float (&va)[16], const float (&vb)[16], auto p_lo, auto p_hi) {
    unsigned* vs = reinterpret_cast<unsigned*>(st) + wave * X_TILES + mytile;
#pragma unroll
    for (int p = decltype(p_lo)::value; p < decltype(p_hi)::value; ++p) {
      unsigned hp, lp;
      split2_pk_scaled(va[p], vb[p], hx_in, hp, lp);
      vs[(p * 2 + 0) * (8 * X_TILES)] = hp;
      vs[(p * 2 + 1) * (8 * X_TILES)] = lp;
    }
  };
  // stage the chunk held in the raw stage into V stage `st` (this thread: its tile, channels 2w and 2w+1)
  auto stage_chunk = [&](float* st) {
    if constexpr (HX) {
      float va[16], vb[16];
      read_patch_to(va, 2 * wave);
      transform(va);
      read_patch_to(vb, 2 * wave + 1);
      transform(vb);
      store_pair(st, va, vb, std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});
    } else {
      read_patch(2 * wave);
      store_patch(st, 2 * wave);
      read_patch(2 * wave + 1);
      store_patch(st, 2 * wave + 1);
    }
  };

  const uint4* wq = reinterpret_cast<const uint4*>(a.wt);
  const size_t pos_stride = (size_t)n_cc * n_ct * FRAG;
  auto load_A = [&](uint4 (&fr)[NC][NPC], int p, int cc, int co_tile) {
    const uint4* base = wq + (size_t)p * pos_stride + ((size_t)cc * n_ct + co_tile * NC) * FRAG + lane;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int s = 0; s < NPC; ++s) fr[c][s] = base[c * FRAG + s * 64];
  };
  const int b_lane = (8 * h) * X_TILES + j;
  auto load_B = [&](float (&raw)[8], const float* cur, int pi, int tg) {
    const float* bp = cur + (p0 + pi) * (X_KC * X_TILES) + tg * 32 + b_lane;
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) raw[qq] = bp[qq * X_TILES];
  };
  auto load_Bh = [&](uint4 (&fr)[2], const float* cur, int pi, int tg) {      // f16x2: the operand itself, no split
    const unsigned* bp = reinterpret_cast<const unsigned*>(cur) + (p0 + pi) * (X_KC * X_TILES) + (4 * h) * X_TILES + tg * 32 + j;
#pragma unroll
    for (int s = 0; s < 2; ++s)
      fr[s] = make_uint4(bp[s * 8 * X_TILES], bp[(s * 8 + 1) * X_TILES], bp[(s * 8 + 2) * X_TILES], bp[(s * 8 + 3) * X_TILES]);
  };

  f32x16 acc[2][NC][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4 * NC; ++i)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[i / (2 * NC)][(i >> 1) % NC][i & 1][rr] = 0.f;
  };
  zero_acc();

  // ---- prologue of the first tile ----
  Geo cur_g = geo_of(tile);
  uint4 afr[2][NC][NPC];
  load_A(afr[0], p0, cur_g.c0, cur_g.co_tile);
  if constexpr (WPRIV) load_A(afr[1], p0 + 1, cur_g.c0, cur_g.co_tile);
  hx_scales_of(cur_g.b, hx_in, hx_out);
  set_dma_geo(cur_g);
  // workgroup barrier for LDS data only: __syncthreads() also drains vmcnt, i.e. would wait for a DMA just issued
  auto lds_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0): this wave's LDS stores / reads are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  issue_dma(0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  stage_chunk(lds);
  __syncthreads();
  issue_dma(1);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (a.dbg) t1 = __builtin_amdgcn_s_memtime();

  int g = 0;                                                  // chunks done so far: stage parity
  // pending maxima of the image whose tile passes this workgroup is in (see the epilogue)
  [[maybe_unused]] int pend_b = cur_g.b;
  [[maybe_unused]] unsigned pend_o = 0u, pend_a = 0u;
  [[maybe_unused]] auto flush_amax = [&]() {
    if (lane == 0) {
      if (a.amax_out) ipdm_amax_atomic(a.amax_out + (size_t)pend_b * IPDM_AMAX_SLOT, wave, __builtin_bit_cast(float, pend_o));
      if (a.amax_act) ipdm_amax_atomic(a.amax_act + (size_t)pend_b * IPDM_AMAX_SLOT, wave, __builtin_bit_cast(float, pend_a));
    }
  };
#ifdef IPDM_WBX3_TRACE
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // step timeline of chunk 2 of the second tile (diagnostic build)
#define IPDM_TR(k) if (a.dbg && trace_now) tr[k] = __builtin_amdgcn_s_memtime()
  unsigned long long te[18] = {};                              // epilogue timeline of the second tile: start, then per round
  int tiles_done = 0;                                           //   after the exchange stores / barrier / transform+stores / barrier
#define IPDM_TE(k) if (a.dbg && tiles_done == 1) te[k] = __builtin_amdgcn_s_memtime()
  unsigned long long tf[6] = {};                               // inside round 1's transform phase
#define IPDM_TF(k) if (a.dbg && tiles_done == 1) tf[k] = __builtin_amdgcn_s_memtime()
#else
#define IPDM_TF(k)
#define IPDM_TR(k)
#define IPDM_TE(k)
#endif
  while (true) {
    const int next_tile = tile + t_step;
    const bool has_next = next_tile < t_end;
    const Geo next_g = geo_of(has_next ? next_tile : tile);
    for (int ch = 0; ch < n_chunks; ++ch, ++g) {
#ifdef IPDM_WBX3_TRACE
      const bool trace_now = g == n_chunks + 2;
#endif
      IPDM_TR(0);
      const float* cur = lds + (g & 1) * X_V_ELEMS;
      float* nxt = lds + ((g + 1) & 1) * X_V_ELEMS;
      // the chunk two ahead in the stream (DMA target) and the one after this (A fragments of position p0)
      const bool dma_next = ch + 2 >= n_chunks;
      const int dma_chunk = dma_next ? ch + 2 - n_chunks : ch + 2;
      if (ch + 2 == n_chunks) set_dma_geo(next_g);            // from here on the raw stage belongs to the next tile
      const bool a_next = ch + 1 >= n_chunks;
      const int a_chunk = a_next ? next_g.c0 : cur_g.c0 + ch + 1;
      const int a_cot = a_next ? next_g.co_tile : cur_g.co_tile;
      if constexpr (HX) {
        [[maybe_unused]] float unused_out;
        if (a_next && next_g.b != cur_g.b) hx_scales_of(next_g.b, hx_in, unused_out);   // this chunk stages chunk 0 of the NEXT tile (its image's scale)
      }
      if constexpr (WPRIV) {
        // this wave's DMA of chunk c+1 (issued a chunk ago) and the fragments of position p0 have landed: fragments of position
        // p0+1 first (older than the next DMA in the in-order vmcnt queue), the wave's own patches into registers, and at once
        // the DMA of chunk c+2 into the block just read -- no barrier: nobody else reads it.  Transform / split / packed stores
        // are dealt over the four MFMA steps; the chunk ends with the V-stage barrier (LDS only: no vmcnt drain).
        uint4 bsh[2][2];
        float va[16], vb[16];
        __builtin_amdgcn_s_waitcnt(0x0F74);                     // vmcnt(4): everything but the four youngest -- position p0+1's
        read_patch_to(va, 2 * wave);                            // fragments, requested at the end of the previous chunk
        read_patch_to(vb, 2 * wave + 1);
        load_Bh(bsh[0], cur, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): the patches are in registers
        IPDM_TR(5);
        issue_dma(dma_chunk);
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto sc) {
          constexpr int st = decltype(sc)::value;
          constexpr int pi = st >> 1, tg = st & 1;
          if constexpr (st < 3) load_Bh(bsh[(st + 1) & 1], cur, (st + 1) >> 1, (st + 1) & 1);
          if constexpr (st == 0) transform(va);
          if constexpr (st == 1) transform(vb);
          if constexpr (st == 2) {
            store_pair(nxt, va, vb, std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
            load_A(afr[0], p0, a_chunk, a_cot);
          }
          if constexpr (st == 3) store_pair(nxt, va, vb, std::integral_constant<int, 8>{}, std::integral_constant<int, 16>{});
          const f16x8 bh = __builtin_bit_cast(f16x8, bsh[st & 1][0]), bl = __builtin_bit_cast(f16x8, bsh[st & 1][1]);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            f32x16 v = acc[pi][c][tg];
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][1]), bh, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][0]), bl, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][0]), bh, v, 0, 0, 0);
            acc[pi][c][tg] = v;
          }
          if constexpr (st < 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // next step's operand reads first
          if constexpr (st == 2) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);    // ... and the fragment requests
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
            if constexpr (st >= 2) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          IPDM_TR(1 + st);
        });
        // position p0+1's fragments of the NEXT chunk, the moment their registers are free (the last MFMA that reads them has
        // issued): a barrier wait and the next chunk's patch reads earlier than at its top, where step 2 stalled ~1 k cycles on them
        load_A(afr[1], p0 + 1, a_chunk, a_cot);
      } else if constexpr (HX) {
        // f16x2 chunk.  With three MFMAs per product the matrix work of a chunk (48 MFMAs per wave) no longer covers a DMA
        // round trip issued a quarter chunk in: the raw patches are read into registers FIRST, the barrier that frees the raw
        // stage follows at once and the DMA of chunk c+2 is issued right behind it (a whole chunk to land); the transform,
        // the split and the packed stores of chunk c+1 are dealt from registers over the four MFMA steps.  The A fragments
        // of position p0+1 are requested before the DMA (vmcnt retires in order: a load issued behind the DMA and consumed in
        // this chunk would wait for the DMA).
        uint4 bsh[2][2];
        float va[16], vb[16];
        load_A(afr[1], p0 + 1, cur_g.c0 + ch, cur_g.co_tile);
        read_patch_to(va, 2 * wave);
        read_patch_to(vb, 2 * wave + 1);
        load_Bh(bsh[0], cur, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                        // everyone has read the raw stage
        IPDM_TR(5);
        issue_dma(dma_chunk);
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto sc) {
          constexpr int st = decltype(sc)::value;
          constexpr int pi = st >> 1, tg = st & 1;
          if constexpr (st < 3) load_Bh(bsh[(st + 1) & 1], cur, (st + 1) >> 1, (st + 1) & 1);
          if constexpr (st == 0) transform(va);
          if constexpr (st == 1) transform(vb);
          if constexpr (st == 2) {
            store_pair(nxt, va, vb, std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
            load_A(afr[0], p0, a_chunk, a_cot);
          }
          if constexpr (st == 3) store_pair(nxt, va, vb, std::integral_constant<int, 8>{}, std::integral_constant<int, 16>{});
          const f16x8 bh = __builtin_bit_cast(f16x8, bsh[st & 1][0]), bl = __builtin_bit_cast(f16x8, bsh[st & 1][1]);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            f32x16 v = acc[pi][c][tg];
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][1]), bh, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][0]), bl, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][0]), bh, v, 0, 0, 0);
            acc[pi][c][tg] = v;
          }
          if constexpr (st < 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // next step's operand reads first
          if constexpr (st == 2) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);    // ... and the fragment requests
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
            if constexpr (st >= 2) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          IPDM_TR(1 + st);
        });
      } else {
      bf16x8 bs[2][3];
      uint4 bsh[2][2];
      float raw[8];
      if constexpr (HX) {
        load_Bh(bsh[0], cur, 0, 0);
      } else {
        load_B(raw, cur, 0, 0);
        split3(raw, bs[0][0], bs[0][1], bs[0][2]);
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<4>([&](auto sc) {
        constexpr int st = decltype(sc)::value;
        constexpr int pi = st >> 1, tg = st & 1;
        if constexpr (st < 3) {
          if constexpr (HX) load_Bh(bsh[(st + 1) & 1], cur, (st + 1) >> 1, (st + 1) & 1);
          else load_B(raw, cur, (st + 1) >> 1, (st + 1) & 1);
        }
        if constexpr (st == 0) {
          stage_chunk(nxt);
          load_A(afr[1], p0 + 1, cur_g.c0 + ch, cur_g.co_tile);
        }
        if constexpr (st == 1) issue_dma(dma_chunk);
        if constexpr (st == 2) load_A(afr[0], p0, a_chunk, a_cot);
        constexpr bool XFORM = st == 0;
        constexpr int MIDSTEP = 0;                                  // the mid-chunk barrier follows step 0
        if constexpr (HX) {
          const f16x8 bh = __builtin_bit_cast(f16x8, bsh[st & 1][0]), bl = __builtin_bit_cast(f16x8, bsh[st & 1][1]);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            f32x16 v = acc[pi][c][tg];
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][1]), bh, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][0]), bl, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[pi][c][0]), bh, v, 0, 0, 0);
            acc[pi][c][tg] = v;
          }
          constexpr int XRD = 16, XVALU = 28, XST = 6;              // patch reads, then per MFMA: VALU, packed-pair stores
          if constexpr (st < 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
          if constexpr (XFORM) __builtin_amdgcn_sched_group_barrier(0x100, XRD, 0);
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, XFORM ? XVALU : 3, 0);
            if constexpr (XFORM) __builtin_amdgcn_sched_group_barrier(0x200, XST, 0);
            if constexpr (st < 3) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
          }
        } else {
        if constexpr (st < 3) split3(raw, bs[(st + 1) & 1][0], bs[(st + 1) & 1][1], bs[(st + 1) & 1][2]);
        const bf16x8 bh = bs[st & 1][0], bm = bs[st & 1][1], bl = bs[st & 1][2];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          f32x16 v = acc[pi][c][tg];
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][2]), bh, v, 0, 0, 0);
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][0]), bl, v, 0, 0, 0);
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][1]), bm, v, 0, 0, 0);
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][1]), bh, v, 0, 0, 0);
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][0]), bm, v, 0, 0, 0);
          v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[pi][c][0]), bh, v, 0, 0, 0);
          acc[pi][c][tg] = v;
        }
        if constexpr (st < 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        constexpr int XRD = 16, XVALU = 12, XST = 3;                // patch reads, VALU per MFMA, patch stores
        if constexpr (XFORM) __builtin_amdgcn_sched_group_barrier(0x100, XRD, 0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, XFORM ? XVALU : 5, 0);
          if constexpr (XFORM) __builtin_amdgcn_sched_group_barrier(0x200, XST, 0);
          if constexpr (st < 3) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        IPDM_TR(1 + st);
        if constexpr (st == MIDSTEP) {
          __syncthreads();                                    // everyone has read the raw stage
          IPDM_TR(5);
        }
      });
      }
      if constexpr (WPRIV) {
        lds_barrier();
      } else {
        __builtin_amdgcn_s_waitcnt(0);                        // the DMA has landed
        IPDM_TR(6);
        __syncthreads();
      }
      IPDM_TR(7);
    }
    if (a.dbg) t2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue of this tile in the stage its last chunk was read from (the other stage and the raw stage already
    //      hold the next tile); rounds over (channel tile c, tile group tg) ----
    float* ms = lds + ((g - 1) & 1) * X_V_ELEMS;              // M[pos 16][co 32][tile 32]
    IPDM_TE(0);
    const int etile = tid & 31, ecg = tid >> 5;               // this thread: tile, channels 2*ecg, 2*ecg+1 of the 32
    const int co0 = cur_g.co_tile * (CO32 ? 32 : X_CO);
    // bias and residual of a round are fetched ONE ROUND AHEAD (round 0: before the first exchange): issued next to their
    // use, each of the four residual loads of a round exposed a full HBM latency to all eight waves at once
    // (8x8 tile blocks: same round, before the exchange -- one buffer; the second costs that instantiation nine spills)
    constexpr int NR = POOL ? 1 : 2;
    constexpr bool AHEAD = TX == 16;
    float2 resv[AHEAD ? 2 : 1][2][NR];
    float biasv[AHEAD ? 2 : 1][2];
    [[maybe_unused]] float scalev[AHEAD ? 2 : 1][2];            // f16x2: the channel's inverse weight scale (blob tail)
    const float* const scale_p = HX ? reinterpret_cast<const float*>(wq + 16 * pos_stride) : a.wt;
    // address of an output = [one 64-bit base per tile pass and tensor, scalar] + [32-bit byte offset: a scalar term per (round,
    // channel of the pair, row) + this thread's own term, fixed for the whole launch] -- the `global_* v, v, s[base]` form, one
    // vector add per address (the four-term element index recomputed per address was ~12 vector instructions, five of them
    // quarter-rate 64-bit multiplies: a third of the epilogue's issue time)
    [[maybe_unused]] const int oH = POOL ? a.H >> 1 : a.H, oW = POOL ? a.W >> 1 : a.W;   // dimensions of the written tensor
    [[maybe_unused]] constexpr int rstep = POOL ? 1 : 2;                                 // its rows / columns per tile
    [[maybe_unused]] const size_t tile_base =
        ((KSP ? (size_t)cur_g.ks * a.B * a.Cout * HW : 0) +                              // this K part's plane set of the workspace
         ((size_t)cur_g.b * a.Cout + co0) * ((size_t)oH * oW) + (size_t)(POOL ? cur_g.y0 >> 1 : cur_g.y0) * oW +
         (POOL ? cur_g.x0 >> 1 : cur_g.x0)) * 4;
    [[maybe_unused]] unsigned eoff4 = 4u * (unsigned)((ecg * 2) * (oH * oW) + rstep * (etile / TX) * oW + rstep * (etile % TX));
    // (opaque per tile: hipcc otherwise hoists all sixteen sums below out of the tile loop and spills them across the chunk loop)
    asm volatile("" : "+v"(eoff4));
    [[maybe_unused]] auto boff = [&](int c, int tg, int i, int ii) -> unsigned {
      return 4u * (unsigned)((c * 32 + i) * (oH * oW) + (rstep * (tg * 32 / TX) + ii) * oW) + eoff4;
    };
    auto out_index = [&](int c, int tg, int i, int ii) -> size_t {   // POLY only: element index of the row's first output; second: + d
      int py, px;
      tile_origin(tg * 32 + etile, py, px);
      return ((size_t)cur_g.b * a.Cout + co0 + c * 32 + ecg * 2 + i) * HW + (size_t)(py + pd * ii) * a.W + px;
    };
    auto in_range = [&](int tg) {
      if constexpr (POLY) return true;                          // exactly 64 tiles per image
      const int T = tg * 32 + etile;
      return cur_g.y0 + 2 * (T / TX) < a.H && cur_g.x0 + 2 * (T % TX) < a.W;
    };
    // (unconditional loads -- an absent operand or an out-of-range tile reads the weight blob -- so that hipcc can count
    // what is in flight: behind a branch it waits for vmcnt(0), i.e. also for the batch just issued)
    const bool has_res = a.residual != nullptr, has_bias = a.bias != nullptr;
    const float* const res_p = has_res ? a.residual : a.wt;
    const float* const bias_p = has_bias ? a.bias : a.wt;
    [[maybe_unused]] const char* const res_b = reinterpret_cast<const char*>(res_p) + (has_res ? tile_base : 0);
    [[maybe_unused]] char* const out_b = reinterpret_cast<char*>(a.out) + tile_base;           // (dereferenced only where non-null)
    [[maybe_unused]] char* const act_b = reinterpret_cast<char*>(a.out_act) + tile_base;
    auto prefetch = [&](auto rc) {
      constexpr int rnd = decltype(rc)::value;
      constexpr int c = rnd >> 1, tg = rnd & 1, bf = AHEAD ? rnd & 1 : 0;
      const bool res_ok = has_res && in_range(tg);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        biasv[bf][i] = bias_p[has_bias ? cur_g.b * a.bias_bstride + co0 + c * 32 + ecg * 2 + i : 0];
        if constexpr (HX) scalev[bf][i] = scale_p[co0 + c * 32 + ecg * 2 + i] * hx_out;
#pragma unroll
        for (int ii = 0; ii < NR; ++ii) {
          if constexpr (POLY) {
            const size_t o = res_ok ? out_index(c, tg, i, ii) : 0;
            resv[bf][i][ii] = make_float2(res_p[o], res_p[o + (res_ok ? pd : 0)]);
          } else {
            const unsigned ob = res_ok ? boff(c, tg, i, ii) : 0u;
            if constexpr (POOL) resv[bf][i][ii] = make_float2(*reinterpret_cast<const float*>(res_b + ob), 0.f);
            else resv[bf][i][ii] = *reinterpret_cast<const float2*>(res_b + ob);
          }
        }
      }
    };
    // maxima of what the PREVIOUS tile passes stored (wave-uniform, kept in scalar registers across the chunk loop) go out now,
    // when the image changes: the atomics then complete under this epilogue's own stores.  Issued at the END of a tile pass they
    // sat in front of the next chunk's fragment loads in the in-order vmcnt queue, and 256 waves per image-line made each take
    // microseconds (ipdm_common.h).
    if constexpr (!KSP) {
      if (cur_g.b != pend_b) {
        flush_amax();
        pend_b = cur_g.b;
        pend_o = pend_a = 0u;
      }
    }
    if constexpr (AHEAD) prefetch(std::integral_constant<int, 0>{});
    [[maybe_unused]] float amx_o = 0.f, amx_a = 0.f;            // max |stored value| of this thread over the tile pass
    static_for<2 * NC>([&](auto rc) {
      constexpr int rnd = decltype(rc)::value;
      constexpr int c = rnd >> 1, tg = rnd & 1, bf = AHEAD ? rnd & 1 : 0;
      if constexpr (!AHEAD) prefetch(rc);
#pragma unroll
      for (int pi = 0; pi < 2; ++pi)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int col = (rr & 3) + 8 * (rr >> 2) + 4 * h;
          ms[((p0 + pi) * 32 + col) * 32 + j] = acc[pi][c][tg][rr];
          acc[pi][c][tg][rr] = 0.f;                             // cleared for the next tile here, in the shadow of the exchange
        }
      IPDM_TE(1 + 4 * rnd);
      __syncthreads();
      IPDM_TE(2 + 4 * rnd);
      if constexpr (AHEAD && rnd < 2 * NC - 1) prefetch(std::integral_constant<int, rnd + 1>{});
      constexpr int NSV = POOL ? 1 : 4;
      [[maybe_unused]] float sv[2][NSV];                      // STATS: this thread's stored values of the round
      if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int k = 0; k < NSV; ++k) sv[i][k] = 0.f;
      }
      if constexpr (rnd == 1) { IPDM_TF(0); }
      if (in_range(tg)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int cl = ecg * 2 + i;
          float m[16];
#pragma unroll
          for (int p = 0; p < 16; ++p) m[p] = ms[(p * 32 + cl) * 32 + etile];
          float tt[2][4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            tt[0][qq] = m[0 * 4 + qq] + m[1 * 4 + qq] + m[2 * 4 + qq];
            tt[1][qq] = m[1 * 4 + qq] - m[2 * 4 + qq] - m[3 * 4 + qq];
          }
#ifdef IPDM_WBX3_TRACE
          if constexpr (rnd == 1) {
            asm volatile("" : "+v"(tt[0][0]), "+v"(tt[1][3]));
            if (i == 0) { IPDM_TF(1); } else { IPDM_TF(3); }
          }
#endif
          const float bias = has_bias ? biasv[bf][i] : 0.f;
          if constexpr (POOL) {
            // ConvMeanPool: (y[::2,::2] + y[1::2,::2] + y[::2,1::2] + y[1::2,1::2]) / 4 in the reference's order; the
            // output tile IS the 2x2 pooling window, so the full-resolution result is never written
            float y00 = tt[0][0] + tt[0][1] + tt[0][2], y01 = tt[0][1] - tt[0][2] - tt[0][3];
            float y10 = tt[1][0] + tt[1][1] + tt[1][2], y11 = tt[1][1] - tt[1][2] - tt[1][3];
            if constexpr (HX) {
              const float sc = scalev[bf][i];
              y00 = __builtin_fmaf(y00, sc, bias); y01 = __builtin_fmaf(y01, sc, bias);
              y10 = __builtin_fmaf(y10, sc, bias); y11 = __builtin_fmaf(y11, sc, bias);
            } else {
              y00 += bias; y01 += bias; y10 += bias; y11 += bias;
            }
            float v = (((y00 + y10) + y01) + y11) * 0.25f;
            const unsigned ob = boff(c, tg, i, 0);
            float vr = v;                                       // what `out` stores (a.res_second: without the residual)
            if (has_res) v += resv[bf][i][0].x;
            vr = a.res_second ? vr : v;
            v *= a.out_scale;
            vr *= a.out_scale;
            if constexpr (STATS) sv[i][0] = vr;
            if constexpr (!KSP) amx_o = fmaxf(amx_o, fabsf(vr));
            if (a.out) *reinterpret_cast<float*>(out_b + ob) = vr;
            if (a.out_act) {
              const float e = a.act_out == IPDM_ACT_ELU ? fast_elu(v) : ipdm_act(v, a.act_out);
              if constexpr (!KSP) amx_a = fmaxf(amx_a, fabsf(e));
              *reinterpret_cast<float*>(act_b + ob) = e;
            }
          } else {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            float y0v = tt[ii][0] + tt[ii][1] + tt[ii][2], y1v = tt[ii][1] - tt[ii][2] - tt[ii][3];
            if constexpr (HX) {
              y0v = __builtin_fmaf(y0v, scalev[bf][i], bias);
              y1v = __builtin_fmaf(y1v, scalev[bf][i], bias);
            } else {
              y0v += bias;
              y1v += bias;
            }
            [[maybe_unused]] size_t o = 0;
            [[maybe_unused]] unsigned ob = 0;
            if constexpr (POLY) o = out_index(c, tg, i, ii);
            else ob = boff(c, tg, i, ii);
            float r0v = y0v, r1v = y1v;                       // what `out` stores (a.res_second: without the residual)
            if (has_res) {
              y0v += resv[bf][i][ii].x;
              y1v += resv[bf][i][ii].y;
            }
            r0v = a.res_second ? r0v : y0v;
            r1v = a.res_second ? r1v : y1v;
            y0v *= a.out_scale;
            y1v *= a.out_scale;
            r0v *= a.out_scale;
            r1v *= a.out_scale;
            if constexpr (STATS) {
              sv[i][2 * ii] = r0v;
              sv[i][2 * ii + 1] = r1v;
            }
            if constexpr (!KSP) amx_o = fmaxf(amx_o, fmaxf(fabsf(r0v), fabsf(r1v)));
            if constexpr (POLY) {
              if (a.out) {
                a.out[o] = r0v;
                a.out[o + pd] = r1v;
              }
            } else {
              // (streaming stores: the result is read next by another kernel after 200+ MB of other traffic; `nt` measured -0.6 % per step)
              typedef float ntf2 __attribute__((ext_vector_type(2)));
              if (a.out) __builtin_nontemporal_store(ntf2{r0v, r1v}, reinterpret_cast<ntf2*>(out_b + ob));
            }
            if (a.out_act) {
              const float e0 = a.act_out == IPDM_ACT_ELU ? fast_elu(y0v) : ipdm_act(y0v, a.act_out);
              const float e1 = a.act_out == IPDM_ACT_ELU ? fast_elu(y1v) : ipdm_act(y1v, a.act_out);
              if constexpr (!KSP) amx_a = fmaxf(amx_a, fmaxf(fabsf(e0), fabsf(e1)));
              if constexpr (POLY) {
                a.out_act[o] = e0;
                a.out_act[o + pd] = e1;
              } else {
                *reinterpret_cast<float2*>(act_b + ob) = make_float2(e0, e1);
              }
            }
          }
          }
#ifdef IPDM_WBX3_TRACE
          if constexpr (rnd == 1) {
            if (i == 0) { IPDM_TF(2); } else { IPDM_TF(4); }
          }
#endif
        }
      }
      if constexpr (STATS) {
        // the 32 lanes of a half-wave hold the 32 tiles of tile group tg for channels 2*ecg, 2*ecg+1: count / mean / sum of
        // squared deviations about that mean, reduced by butterflies (fixed order -> deterministic), one partial per
        // (image, channel, tile block, tile group)
        const bool inr = in_range(tg);
        const unsigned long long bal = __ballot(inr);
        const float cnt = (float)(NSV * __popcll(h ? (bal >> 32) : (bal & 0xffffffffull)));
        const int tiles_img = a.tiles_x * a.tiles_y;
        const int tb = (cur_g.y0 / (2 * TY)) * a.tiles_x + cur_g.x0 / (2 * TX);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          // one pass: sums of (v - K) and (v - K)^2 about a shift K taken FROM the data (the first lane's first value of
          // this half-wave), so that the subtraction below cancels nothing the spread of the data does not
          const float k0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv[i][0]), 0));
          const float k1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv[i][0]), 32));
          const float K = h ? k1 : k0;
          float s1 = 0.f, s2 = 0.f;
          if (inr) {
#pragma unroll
            for (int k = 0; k < NSV; ++k) {
              const float dlt = sv[i][k] - K;
              s1 += dlt;
              s2 += dlt * dlt;
            }
          }
          s1 = half_wave_sum(s1);
          s2 = half_wave_sum(s2);
          if (etile == 0) {
            const int co = co0 + c * 32 + ecg * 2 + i;
            float* sp = a.stats + ((((size_t)cur_g.b * a.Cout + co) * tiles_img + tb) * 2 + tg) * 3;
            const float dm = cnt > 0.f ? s1 / cnt : 0.f;
            sp[0] = cnt; sp[1] = K + dm; sp[2] = fmaxf(s2 - s1 * dm, 0.f);
          }
        }
      }
      IPDM_TE(3 + 4 * rnd);
      __syncthreads();                                        // M is rewritten by the next round / the next tile's V
      IPDM_TE(4 + 4 * rnd);
    });
    if constexpr (!KSP) {                                       // a tile pass lies inside ONE image
      if (a.amax_out || a.amax_act) {
        const unsigned mo = __builtin_bit_cast(unsigned, ipdm_wave_max(amx_o)), ma = __builtin_bit_cast(unsigned, ipdm_wave_max(amx_a));
        const unsigned uo = __builtin_amdgcn_readfirstlane(mo), ua = __builtin_amdgcn_readfirstlane(ma);
        pend_o = pend_o > uo ? pend_o : uo;
        pend_a = pend_a > ua ? pend_a : ua;
      }
    }
#ifdef IPDM_WBX3_TRACE
    ++tiles_done;
#endif
    if (!has_next) break;
    tile = next_tile;
    if (next_g.b != cur_g.b) hx_scales_of(next_g.b, hx_in, hx_out);
    cur_g = next_g;
  }
  if constexpr (!KSP) flush_amax();
  if (a.dbg) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* d4 = a.dbg + (size_t)blockIdx.x * 4;
      d4[0] = t0; d4[1] = t1; d4[2] = t2; d4[3] = t3;
    }
#ifdef IPDM_WBX3_TRACE
    if (lane == 0) {
      unsigned long long* d8 = a.dbg + (size_t)gridDim.x * 4 + ((size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) d8[k] = tr[k];
      unsigned long long* d18 = a.dbg + (size_t)gridDim.x * (4 + 64) + ((size_t)blockIdx.x * 8 + wave) * 18;
#pragma unroll
      for (int k = 0; k < 18; ++k) d18[k] = te[k];
      unsigned long long* d6 = a.dbg + (size_t)gridDim.x * (4 + 64 + 8 * 18) + ((size_t)blockIdx.x * 8 + wave) * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) d6[k] = tf[k];
    }
#endif
  }
}
#undef IPDM_TR


bool x_small(const ConvArgs& a) { return a.W < 32 || a.dil > 1; }
// undilated images of up to 16 x 16 pixels with enough (image, channel tile) pairs to fill the chip: the persistent
// LDS-DMA kernel with an 8 x 8 tile block per image
bool x_small_dma(const ConvArgs& a) {
  static int min_pairs = -1;                     // IPDM_WBX3_SMALL_MIN: tuning aid
  if (min_pairs < 0) {
    const char* e = getenv("IPDM_WBX3_SMALL_MIN");
    min_pairs = e ? atoi(e) : 160;
  }
  return a.dil == 1 && a.W <= 16 && a.H <= 16 && a.W % 2 == 0 && a.H % 2 == 0 && a.Cin >= 2 * X_KC &&
         (int64_t)a.B * (a.Cout / X_CO) >= min_pairs;
}

// 16-pixel layers with fewer than 512 output channels as 32-channel workgroups (CO32): what the f16x2 family's PLAIN call runs for
// the shapes whose split-K rule says 2 (the host mirror then does not split: one launch instead of two K halves + a reduction;
// IPDM_WBX3_CO32=0: off; a function of the layer shape only)
bool x_co32(const ConvArgs& a) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("IPDM_WBX3_CO32");
    enabled = e ? atoi(e) : 1;
  }
  return enabled && a.hx && a.dil == 1 && a.W <= 16 && a.H <= 16 && a.W % 4 == 0 && a.H % 2 == 0 && a.Cin >= 2 * X_KC &&
         a.Cout < 512 && !a.pool2 && !a.stats;
}

int wino_persist() {                             // IPDM_WBX3_PERSIST=0: one workgroup per tile (tuning / fallback)
  static int persist = -1;
  if (persist < 0) {
    const char* e = getenv("IPDM_WBX3_PERSIST");
    persist = e ? atoi(e) : 1;
  }
  return persist;
}

int cus_per_xcd() {
  static int n = 0;
  if (!n) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= 8)
      n = v / 8;
    else
      n = 32;                                    // MI355X: 256 CUs in 8 XCDs
  }
  return n;
}

}  // namespace

bool wino_bx3_ok(const ConvArgs& a, int ks) {
  if (!(ks == 3 && a.D == 1 && a.Cin % X_KC == 0 && a.Cout % X_CO == 0 && !a.coef && a.act == IPDM_ACT_NONE)) return false;
  if (a.dil < 1 || a.dil > 4) return false;
  // buffer descriptors: the tensor must end below the padding marker (2^29 + 2^29 register path, 2^30 DMA path)
  if ((size_t)a.B * a.Cin * a.H * a.W * 4 >= (x_small(a) ? 0x1fffffffull : 0x3fffffffull)) return false;
  if (x_small(a)) return a.H % (2 * a.dil) == 0 && a.W % (2 * a.dil) == 0 && (a.H * a.W) / 4 >= 32;
  return a.H % 2 == 0 && a.W % 2 == 0 && a.H >= 8;
}

// dilated 16 x 16 images (the deepest stages of the score networks): the persistent kernel on the polyphase tile space with the
// whole padded image in the raw stage (a function of the layer shape; IPDM_WBX3_POLY=0: the register-staged kernel)
bool x_poly(const ConvArgs& a) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("IPDM_WBX3_POLY");
    enabled = e ? atoi(e) : 1;
  }
  return enabled && wino_persist() && a.dil > 1 && a.H == 16 && a.W == 16 && 16 % (2 * a.dil) == 0 && a.Cin >= 2 * X_KC &&
         !a.pool2 && !a.stats && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
}

template <bool HXV>
static int conv_wino_bx3_launch_t(ConvArgs a, hipStream_t s) {
  if (x_poly(a)) {
    a.tiles_x = a.tiles_y = 1;
    a.co_tiles = a.Cout / X_CO;
    const int64_t nblk = (int64_t)a.B * a.co_tiles;
    static bool poly_attr = false;
    if (!poly_attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<HXV, 8, 8, true, true, false, false, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS_BYTES);
      poly_attr = true;
    }
    const int per_xcd = (int)((nblk + 7) / 8);
    const int S = per_xcd < cus_per_xcd() ? per_xcd : cus_per_xcd();
    hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<HXV, 8, 8, true, true, false, false, false, true>), dim3((unsigned)(8 * S)),
                       dim3(512), X_LDS_BYTES, s, a, (int)nblk);
    return ipdm_launch_status();
  }
  const bool small = x_small(a);
  if (HXV && small && wino_persist() && x_co32(a) && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0) {
    a.tiles_x = (a.W + 15) / 16;
    a.tiles_y = (a.H + 15) / 16;
    a.co_tiles = a.Cout / 32;
    const int64_t nblk = (int64_t)a.B * a.tiles_x * a.tiles_y * a.co_tiles;
    if (nblk > 0x7fffffff) return IPDM_EUNSUPPORTED;
    static bool co32_attr = false;
    if (!co32_attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<true, 8, 8, true, true, false, false, false, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS_BYTES);
      co32_attr = true;
    }
    const int per_xcd = (int)((nblk + 7) / 8);
    const int S = per_xcd < cus_per_xcd() ? per_xcd : cus_per_xcd();
    hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<true, 8, 8, true, true, false, false, false, false, true>), dim3((unsigned)(8 * S)),
                       dim3(512), X_LDS_BYTES, s, a, (int)nblk);
    return ipdm_launch_status();
  }
  const bool small_dma = small && wino_persist() && x_small_dma(a);
  // the pooled epilogue lives in the persistent wide kernels only (the ConvMeanPool layers of the score nets are 32..128
  // pixels wide); everything else reports "unsupported" and the caller runs convolution + mean-pool separately
  if (a.pool2 && (small || !wino_persist() || a.Cin < 2 * X_KC || a.H % 2 || a.W % 2)) return IPDM_EUNSUPPORTED;
  if (small_dma) {
    a.tiles_x = (a.W + 15) / 16;
    a.tiles_y = (a.H + 15) / 16;
  } else if (small) {
    a.tiles_x = ((a.H * a.W) / 4 + X_TILES - 1) / X_TILES;
    a.tiles_y = 1;
  } else {
    a.tiles_x = (a.W + 2 * X_TX - 1) / (2 * X_TX);
    a.tiles_y = (a.H + 2 * X_TY - 1) / (2 * X_TY);
  }
  a.co_tiles = a.Cout / X_CO;
  const int64_t nblk = (int64_t)a.B * a.tiles_x * a.tiles_y * a.co_tiles;
  if (nblk > 0x7fffffff) return IPDM_EUNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    const void* kernels[] = {reinterpret_cast<const void*>(conv_wino_bx3_kernel<HXV, false>),
                             reinterpret_cast<const void*>(conv_wino_bx3_kernel<HXV, true>),
                             reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<HXV, 16, 4, false, false>),
                             reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<HXV, 8, 8, true, false>),
                             reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<HXV, 16, 4, false, true>),
                             reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<HXV, 8, 8, true, true>),
                             reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<HXV, 16, 4, false, false, true>),
                             reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<HXV, 16, 4, false, true, true>),
                             reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<HXV, 16, 4, false, true, false, true>),
                             reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<HXV, 16, 4, false, true, true, true>)};
    for (const void* k : kernels) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)X_LDS_BYTES);
      if (e != hipSuccess) return (int)e;
    }
    attr_set = true;
  }
  const int per_xcd = (int)((nblk + 7) / 8);
  const int S = per_xcd < cus_per_xcd() ? per_xcd : cus_per_xcd();    // one 128+24 KiB workgroup per CU
  static int dma4_ok = -1;                       // IPDM_WBX3_DMA4=0: dword LDS-DMA everywhere (tuning / fallback)
  if (dma4_ok < 0) {
    const char* e = getenv("IPDM_WBX3_DMA4");
    dma4_ok = e ? atoi(e) : 1;
  }
  const bool dma4 = dma4_ok && a.W % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
  if (a.stats && (small || !wino_persist() || a.Cin < 2 * X_KC)) return IPDM_EUNSUPPORTED;
  if (small_dma) {
    if (dma4)
      hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<HXV, 8, 8, true, true>), dim3((unsigned)(8 * S)), dim3(512), X_LDS_BYTES, s, a, (int)nblk);
    else
      hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<HXV, 8, 8, true, false>), dim3((unsigned)(8 * S)), dim3(512), X_LDS_BYTES, s, a, (int)nblk);
  } else if (small) {
    hipLaunchKernelGGL((conv_wino_bx3_kernel<HXV, true>), dim3((unsigned)nblk), dim3(512), X_LDS_BYTES, s, a);
  } else if (wino_persist() && a.Cin >= 2 * X_KC) {
    if (a.stats) {
      if (!dma4) return IPDM_EUNSUPPORTED;
      if (a.pool2)
        hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<HXV, 16, 4, false, true, true, true>), dim3((unsigned)(8 * S)), dim3(512), X_LDS_BYTES, s, a, (int)nblk);
      else
        hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<HXV, 16, 4, false, true, false, true>), dim3((unsigned)(8 * S)), dim3(512), X_LDS_BYTES, s, a, (int)nblk);
      return ipdm_launch_status();
    }
    if (a.pool2 && dma4)
      hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<HXV, 16, 4, false, true, true>), dim3((unsigned)(8 * S)), dim3(512), X_LDS_BYTES, s, a, (int)nblk);
    else if (a.pool2)
      hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<HXV, 16, 4, false, false, true>), dim3((unsigned)(8 * S)), dim3(512), X_LDS_BYTES, s, a, (int)nblk);
    else if (dma4)
      hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<HXV, 16, 4, false, true>), dim3((unsigned)(8 * S)), dim3(512), X_LDS_BYTES, s, a, (int)nblk);
    else
      hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<HXV, 16, 4, false, false>), dim3((unsigned)(8 * S)), dim3(512), X_LDS_BYTES, s, a, (int)nblk);
  } else {
    hipLaunchKernelGGL((conv_wino_bx3_kernel<HXV, false>), dim3((unsigned)nblk), dim3(512), X_LDS_BYTES, s, a);
  }
  return ipdm_launch_status();
}

int conv_wino_bx3_launch(ConvArgs a, hipStream_t s) {
  return a.hx ? conv_wino_bx3_launch_t<true>(a, s) : conv_wino_bx3_launch_t<false>(a, s);
}

// K parts that pay for a layer shape (a function of the SHAPE only, so that a sample's bits do not depend on its batch):
// undilated images of at most 16 x 16 pixels with fewer than 512 output channels -- (image, channel tile) pairs alone fill
// under half of the chip at the production batch -- run as two K halves when each half keeps at least two chunks
int wino_bx3_ksplit_for(int Cin, int Cout, int H, int W, int dilation) {
  static int enabled = -1;                       // IPDM_WBX3_KSPLIT=0: off (tuning / fallback)
  if (enabled < 0) {
    const char* e = getenv("IPDM_WBX3_KSPLIT");
    enabled = e ? atoi(e) : 1;
  }
  if (!enabled || !wino_persist()) return 1;
  if (dilation != 1 || W > 16 || H > 16 || W % 4 || H % 2 || Cout % X_CO || Cout >= 512) return 1;
  if (Cin % (2 * X_KC) || Cin / X_KC < 4) return 1;
  return 2;
}

template <bool HXV>
static int conv_wino_bx3_launch_ksplit_t(ConvArgs a, int ksplit, float* work, hipStream_t s) {
  if (ksplit != wino_bx3_ksplit_for(a.Cin, a.Cout, a.H, a.W, a.dil) || ksplit < 2 || !work) return IPDM_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(a.x) & 15) != 0) return IPDM_EUNSUPPORTED;          // 16-byte LDS-DMA form only
  const float* bias = a.bias;
  const float* residual = a.residual;
  float* out = a.out;
  float* out_act = a.out_act;
  a.tiles_x = (a.W + 15) / 16;
  a.tiles_y = (a.H + 15) / 16;
  a.co_tiles = a.Cout / X_CO;
  a.ksplit = ksplit;
  a.bias = nullptr; a.residual = nullptr; a.out_act = nullptr; a.out = work; a.pool2 = 0; a.stats = nullptr;
  float* const amax_out = a.amax_out;
  float* const amax_act = a.amax_act;
  a.amax_out = a.amax_act = nullptr;              // the parts are partial sums: the maxima belong to the reduce pass
  const int res_second = a.res_second;
  a.res_second = 0;
  const float out_scale = a.out_scale;
  a.out_scale = 1.f;                              // the parts are raw sums: bias / residual / scale belong to the reduce pass
  const int64_t nblk = (int64_t)a.B * a.tiles_x * a.tiles_y * a.co_tiles * ksplit;
  if (nblk > 0x7fffffff) return IPDM_EUNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_bx3_wide_kernel<HXV, 8, 8, true, true, false, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS_BYTES);
    attr_set = true;
  }
  const int per_xcd = (int)((nblk + 7) / 8);
  const int S = per_xcd < cus_per_xcd() ? per_xcd : cus_per_xcd();
  hipLaunchKernelGGL((conv_wino_bx3_wide_kernel<HXV, 8, 8, true, true, false, false, true>), dim3((unsigned)(8 * S)), dim3(512),
                     X_LDS_BYTES, s, a, (int)nblk);
  const int64_t plane = (int64_t)a.H * a.W, total = (int64_t)a.B * a.Cout * plane;
  hipLaunchKernelGGL(bx3_splitk_reduce_kernel, splitk_reduce_grid(a.B, (int64_t)a.Cout * plane, amax_out || amax_act), dim3(256), 0, s, work, ksplit, bias, residual,
                     out, out_act, a.act_out, a.Cout, plane, total, a.bias_bstride, out_scale, amax_out, amax_act, res_second);
  return ipdm_launch_status();
}

int conv_wino_bx3_launch_ksplit(ConvArgs a, int ksplit, float* work, hipStream_t s) {
  return a.hx ? conv_wino_bx3_launch_ksplit_t<true>(a, ksplit, work, s) : conv_wino_bx3_launch_ksplit_t<false>(a, ksplit, work, s);
}

int conv_wino_bx3_weights(const float* w, void* U, int Cout, int Cin, hipStream_t s) {
  const int n_cc = (Cin + 15) / 16, n_ct = (Cout + 31) / 32;
  const int64_t total = (int64_t)16 * n_cc * n_ct * 512;
  hipLaunchKernelGGL(wino_bx3_weight_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, s, w, (unsigned short*)U, Cout,
                     Cin, n_cc, n_ct);
  return ipdm_launch_status();
}

int conv_wino_hx2_weights(const float* w, void* U, int Cout, int Cin, hipStream_t s) {
  const int n_cc = (Cin + 15) / 16, n_ct = (Cout + 31) / 32;
  const int64_t total = (int64_t)16 * n_cc * n_ct * 512;
  float* inv_scale = reinterpret_cast<float*>(static_cast<char*>(U) + total * 2 * 2);      // behind the two fp16 pieces
  hipLaunchKernelGGL(wino_hx2_scale_kernel, dim3(n_ct * 32), dim3(256), 0, s, w, inv_scale, Cout, Cin, n_ct * 32);
  hipLaunchKernelGGL(wino_hx2_weight_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, s, w, (unsigned short*)U, inv_scale,
                     Cout, Cin, n_cc, n_ct);
  return ipdm_launch_status();
}

}  // namespace ipdm_conv

using namespace ipdm_conv;

extern "C" int64_t ipdm_conv_wino_hx2_weight_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return -1;
  return (int64_t)16 * ((Cin + 15) / 16) * ((Cout + 31) / 32) * 2048 + (int64_t)((Cout + 31) / 32) * 32 * 4;
}

extern "C" int ipdm_conv_wino_hx2_pack_weight(const float* w, void* U, int Cout, int Cin, void* stream) {
  IPDM_REQUIRE(w && U && Cout > 0 && Cin > 0);
  return conv_wino_hx2_weights(w, U, Cout, Cin, ipdm_stream(stream));
}

extern "C" int64_t ipdm_conv_wino_bx3_weight_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return -1;
  return (int64_t)16 * ((Cin + 15) / 16) * ((Cout + 31) / 32) * 3072;
}

extern "C" int ipdm_conv_wino_bx3_pack_weight(const float* w, void* U, int Cout, int Cin, void* stream) {
  IPDM_REQUIRE(w && U && Cout > 0 && Cin > 0);
  return conv_wino_bx3_weights(w, U, Cout, Cin, ipdm_stream(stream));
}

extern "C" int ipdm_conv2d_wino_bx3_supported(int Cin, int Cout, int H, int W, int dilation) {
  ConvArgs a;
  a.coef = nullptr; a.act = IPDM_ACT_NONE; a.dil = dilation; a.D = 1; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.B = 1;
  return wino_bx3_ok(a, 3) ? 1 : 0;
}

static int wino_bx3_entry(const float* x, const void* U, const float* bias, const float* residual, float* out,
                          float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int dilation, int pool2,
                          float* stats, void* stream, int hx = 0, const ipdm_conv_ext_t* ext = nullptr) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && dilation >= 1);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && U && (out || out_act) && x != out && x != out_act);
  ConvArgs a;
  a.x = x; a.wt = (const float*)U; a.bias = bias; a.coef = nullptr; a.residual = residual; a.out = out; a.out_act = out_act;
  a.act_out = act_out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = dilation; a.act = IPDM_ACT_NONE;
  a.D = 1; a.kd = 1; a.tiles_x = a.tiles_y = a.co_tiles = 0; a.dbg = conv_debug_stamps();
  a.pool2 = pool2 ? 1 : 0;
  a.stats = stats;
  a.hx = hx;
  conv_apply_ext(a, ext, hx);
  if (!wino_bx3_ok(a, 3)) return IPDM_EUNSUPPORTED;
  return conv_wino_bx3_launch(a, ipdm_stream(stream));
}

extern "C" int ipdm_conv2d_wino_bx3_f32(const float* x, const void* U, const float* bias, const float* residual,
                                        float* out, float* out_act, int act_out, int B, int Cin, int Cout, int H, int W,
                                        int dilation, int pool2, const ipdm_conv_ext_t* ext, void* stream) {
  return wino_bx3_entry(x, U, bias, residual, out, out_act, act_out, B, Cin, Cout, H, W, dilation, pool2, nullptr, stream, 0, ext);
}

/* Split-K form for 16-pixel layers with fewer than 512 output channels: ipdm_conv2d_wino_bx3_splitk returns the number of K
 * parts (1: use the plain call; a function of the layer shape only), the _f32 call runs the parts into `work`
 * (ksplit * B * Cout * H * W floats) and adds them in fixed order with bias / residual / activation. */
extern "C" int ipdm_conv2d_wino_bx3_splitk(int Cin, int Cout, int H, int W, int dilation) {
  ConvArgs a;
  a.coef = nullptr; a.act = IPDM_ACT_NONE; a.dil = dilation; a.D = 1; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.B = 1;
  if (!wino_bx3_ok(a, 3)) return 1;
  return wino_bx3_ksplit_for(Cin, Cout, H, W, dilation);
}

static int wino_bx3_splitk_entry(const float* x, const void* U, const float* bias, const float* residual, float* out,
                                 float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int dilation, int ksplit,
                                 float* work, void* stream, int hx, const ipdm_conv_ext_t* ext = nullptr) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && dilation >= 1 && ksplit >= 2);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && U && work && (out || out_act) && x != out && x != out_act);
  IPDM_REQUIRE(B <= 65535 || !ext || (!ext->out_amax && !ext->act_amax));
  ConvArgs a;
  a.x = x; a.wt = (const float*)U; a.bias = bias; a.coef = nullptr; a.residual = residual; a.out = out; a.out_act = out_act;
  a.act_out = act_out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = dilation; a.act = IPDM_ACT_NONE;
  a.D = 1; a.kd = 1; a.tiles_x = a.tiles_y = a.co_tiles = 0; a.dbg = nullptr;
  a.hx = hx;
  conv_apply_ext(a, ext, hx);
  if (!wino_bx3_ok(a, 3)) return IPDM_EUNSUPPORTED;
  return conv_wino_bx3_launch_ksplit(a, ksplit, work, ipdm_stream(stream));
}

extern "C" int ipdm_conv2d_wino_bx3_splitk_f32(const float* x, const void* U, const float* bias, const float* residual,
                                               float* out, float* out_act, int act_out, int B, int Cin, int Cout, int H,
                                               int W, int dilation, int ksplit, float* work, const ipdm_conv_ext_t* ext, void* stream) {
  return wino_bx3_splitk_entry(x, U, bias, residual, out, out_act, act_out, B, Cin, Cout, H, W, dilation, ksplit, work, stream, 0, ext);
}

/* f16x2 forms of the three calls above: same arguments, U from ipdm_conv_wino_hx2_pack_weight (conv_kernel.h: two fp16
 * pieces, three MFMAs per product; the shape rules -- _supported, _splitk, _stats_partials -- are shared) */
extern "C" int ipdm_conv2d_wino_hx2_f32(const float* x, const void* U, const float* bias, const float* residual,
                                        float* out, float* out_act, int act_out, int B, int Cin, int Cout, int H, int W,
                                        int dilation, int pool2, const ipdm_conv_ext_t* ext, void* stream) {
  return wino_bx3_entry(x, U, bias, residual, out, out_act, act_out, B, Cin, Cout, H, W, dilation, pool2, nullptr, stream, 1,
                        ext);
}

extern "C" int ipdm_conv2d_wino_hx2_splitk_f32(const float* x, const void* U, const float* bias, const float* residual,
                                               float* out, float* out_act, int act_out, int B, int Cin, int Cout, int H,
                                               int W, int dilation, int ksplit, float* work, const ipdm_conv_ext_t* ext,
                                               void* stream) {
  return wino_bx3_splitk_entry(x, U, bias, residual, out, out_act, act_out, B, Cin, Cout, H, W, dilation, ksplit, work, stream, 1,
                               ext);
}

// partials per plane the statistics epilogue writes for this layer shape (0: that epilogue does not serve it)
extern "C" int ipdm_conv2d_wino_bx3_stats_partials(int Cin, int Cout, int H, int W, int dilation, int pool2) {
  ConvArgs a;
  a.coef = nullptr; a.act = IPDM_ACT_NONE; a.dil = dilation; a.D = 1; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.B = 1;
  a.pool2 = pool2 ? 1 : 0;
  if (!wino_bx3_ok(a, 3) || x_small(a) || !wino_persist() || Cin < 2 * X_KC || W % 4 != 0) return 0;
  if (pool2 && (H % 2 || W % 2)) return 0;
  return 2 * ((W + 2 * X_TX - 1) / (2 * X_TX)) * ((H + 2 * X_TY - 1) / (2 * X_TY));
}

extern "C" int ipdm_conv2d_wino_bx3_stats_f32(const float* x, const void* U, const float* bias, const float* residual,
                                              float* out, float* out_act, int act_out, int B, int Cin, int Cout, int H,
                                              int W, int dilation, int pool2, float* stats, const ipdm_conv_ext_t* ext, void* stream) {
  IPDM_REQUIRE(stats != nullptr);
  if (ipdm_conv2d_wino_bx3_stats_partials(Cin, Cout, H, W, dilation, pool2) == 0) return IPDM_EUNSUPPORTED;
  return wino_bx3_entry(x, U, bias, residual, out, out_act, act_out, B, Cin, Cout, H, W, dilation, pool2, stats, stream, 0, ext);
}

extern "C" int ipdm_conv2d_wino_hx2_stats_f32(const float* x, const void* U, const float* bias, const float* residual,
                                              float* out, float* out_act, int act_out, int B, int Cin, int Cout, int H,
                                              int W, int dilation, int pool2, float* stats, const ipdm_conv_ext_t* ext,
                                              void* stream) {
  IPDM_REQUIRE(stats != nullptr);
  if (ipdm_conv2d_wino_bx3_stats_partials(Cin, Cout, H, W, dilation, pool2) == 0) return IPDM_EUNSUPPORTED;
  return wino_bx3_entry(x, U, bias, residual, out, out_act, act_out, B, Cin, Cout, H, W, dilation, pool2, stats, stream, 1,
                        ext);
}
