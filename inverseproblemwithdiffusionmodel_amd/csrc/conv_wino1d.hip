// 3x3 convolution as ONE-DIMENSIONAL Winograd F(2,3) along x, direct along y, on f16x2 operands (conv_kernel.h).
//
//   Y[y][2 px] = A^T sum_ky [ (G g[ky]) (.) (B^T d[y + ky]) ]        per output row y and pair of pixels, 4-pixel row patch d
//
// The three rows of the filter accumulate into the SAME accumulator, so there are 4 positions (x) instead of the 16 of
// conv_wino_bx3.hip, each a channel contraction with K = 3 Cin.  For the register budget of that kernel (128 accumulator
// registers per wave, 8 waves) a workgroup therefore holds 128 output channels x 8 rows x 32 pixels -- twice the channels of the
// 2-D form over the same pixel block -- and draws 12 instead of 2 x 16 weight values per (co, ci) from L2: 96 + 26 KiB per
// 16-channel chunk against 2 x (64 + 21).  It pays for that with 1.5x the matrix instructions (6 instead of 4 multiply-adds per
// output, co, ci), which the 2-D kernel has to spare: its chunk loop waits on L2 weight fragments, not on the matrix pipe.
// The exchange of the epilogue (4 positions) is half the 2-D form's per output, the input transform a quarter.
//
// Workgroup = 512 threads; wave w = (xi = w & 3, co half = w >> 2) owns position xi for two 32-channel tiles x four 32-column
// blocks (rows 2nb, 2nb+1 x 16 pixel pairs): acc[2][4] = 128 registers.
//   A (U fragments): [pos = ky*4 + xi][ci/16][co/32][piece 2][lane] 16-byte units, global -> VGPR, two slots, requested one
//        filter row (24 MFMAs) ahead.
//   B (V = B^T d, rows): raw rows by wave-private 16-byte LDS-DMA exactly as the 2-D kernel's (same 10 x 40 region per
//        channel), transformed along x, split once into packed fp16 pairs and stored as
//        Vs[row 12][xi 4][piece 2][q>>1][h][q&1][tile 16] words (channel pair cp = 4h + q); a staged row serves three filter rows.
// Two 48.75 KiB V stages + 32 KiB raw; the epilogue's two exchange buffers M[xi 4][co 64][col 32] overlay the stage last read and
// the 15 KiB behind it (one barrier per round: a round's stores go to the buffer the previous round did not read).
// Eligible: 3x3, dilation 1, Cin % 32 == 0, Cout % 128 == 0, W % 4 == 0, 16-byte aligned input (launcher).
#include "conv_kernel.h"

namespace ipdm_conv {

namespace {

constexpr int Y_KC = 16;
constexpr int Y_CO = 128;
constexpr int Y_ROWS = 8, Y_TX = 16;                 // output rows / pixel pairs per row of a workgroup's block
constexpr int Y_VROWS = 12;                          // 10 staged rows + 2 that only the idle lanes of the third item write
constexpr int Y_RS = 1040;                           // words per staged row: 16 (xi, piece, q>>1) x 64 + 16 (bank offset of a row)
constexpr int Y_STAGE = Y_VROWS * Y_RS;              // 12480 words
constexpr int Y_RAW = 8 * 1024;                      // eight wave-private 4 KiB blocks
constexpr int Y_M = 4 * 64 * 32;                     // floats of one exchange buffer M[xi 4][co 64][col 32]
// LDS: V stage 0 | raw stage | V stage 1 = exchange buffer 0, then exchange buffer 1 (its tail lies behind stage 1)
constexpr int Y_S1 = Y_STAGE + Y_RAW;                // word offset of stage 1
constexpr size_t Y_LDS_BYTES = ((size_t)Y_S1 + 2 * Y_M + 2 * Y_CO) * 4;   // + the epilogue's (scale, bias) table
static_assert(Y_STAGE <= 2 * Y_M && Y_LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr float HX1_PRESCALE = 0.5f;                 // the 1-D input transform at most doubles a value
constexpr int Y_QC = 10, Y_RC4 = 40, Y_QN = 100;     // quads / floats per raw row, quads per channel

__device__ __forceinline__ float wino1d_U(const float* g, int p) {       // p = (kz*3 + ky)*4 + xi; g: the filter's 9 (or 27) taps
  g += (p / 12) * 9;
  p %= 12;
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const int ky = p >> 2, xi = p & 3;
  return G[xi][0] * g[ky * 3 + 0] + G[xi][1] * g[ky * 3 + 1] + G[xi][2] * g[ky * 3 + 2];
}

// one workgroup per output channel: inv_scale[co] = 2^-k with max |U| * 2^k in [2^13, 2^14), times 1 / HX1_PRESCALE
__global__ __launch_bounds__(256) void wino1d_scale_kernel(const float* __restrict__ w, float* __restrict__ inv_scale, int Cout,
                                                           int Cin, int n_co_pad, int npos) {
  __shared__ float red[256];
  const int co = blockIdx.x;
  float m = 0.f;
  if (co < Cout)
    for (int i = threadIdx.x; i < Cin * npos; i += 256)
      m = fmaxf(m, fabsf(wino1d_U(w + ((size_t)co * Cin + i / npos) * (npos / 12 * 9), i % npos)));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0 && co < n_co_pad) {
    int e = 0;
    const float mx = red[0];
    if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &e);
    inv_scale[co] = (mx > 0.f && mx < INFINITY ? ldexpf(1.f, e - 14) : 1.f) / HX1_PRESCALE;
  }
}

__global__ __launch_bounds__(256) void wino1d_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                            const float* __restrict__ inv_scale, int Cout, int Cin, int n_cc,
                                                            int n_ct, int npos) {
  const int64_t total = (int64_t)npos * n_cc * n_ct * 512;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i & 7), r = (int)((i >> 3) & 31), h = (int)((i >> 8) & 1);
    const int64_t rest = i >> 9;
    const int ct = (int)(rest % n_ct);
    const int cc = (int)((rest / n_ct) % n_cc);
    const int p = (int)(rest / ((int64_t)n_ct * n_cc));
    const int co = ct * 32 + r, ci = cc * 16 + 8 * h + q;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = wino1d_U(w + ((size_t)co * Cin + ci) * (npos / 12 * 9), p) * (1.f / (inv_scale[co] * HX1_PRESCALE));
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const int64_t base = rest * 2 * 512 + h * 256 + r * 8 + q;
    out[base] = __builtin_bit_cast(unsigned short, hi);
    out[base + 512] = __builtin_bit_cast(unsigned short, lo);
  }
}

// OUTS 1: out, 2: out_act, 3: both (the launcher picks; no branch per store).  STATS: per (image, channel, pixel block) the stored `out`
// values are reduced to (count, mean, sum of squared deviations) partials [B][Cout][tiles_y * tiles_x][3] for the InstanceNorm++
// that follows (deterministic: fixed butterflies inside a half-wave, the four rounds of a channel tile accumulated in registers)
// POOL: the ConvMeanPool epilogue -- a column block's two rows are the two rows of the 2x2 pooling windows: the lower row's values come
// by one cross-row exchange, the upper row's lanes store the mean [B][Cout][H/2][W/2] (residual and statistics at that size)
// FIN: the input is act(InstanceNorm++(x)) of a raw tensor x (a.coef [B][Cin][3]: (x - mu) * scale + shift, then ELU): applied to the raw
// rows on their way through the producer's registers -- every raw value once per row item, the padding kept at zero -- instead of an
// affine + activation pass that writes the normalised tensor and reads it back (8 bytes of HBM per element)
// VOL: 3x3x3 convolution of volumes [B][C][D][H][W] -- the depth taps are further K chunks (chunk = (kz, 16 channels): the raw rows of
// plane z + kz - 1, hardware-zeroed outside the volume; 36 weight positions), and planes of 12 pixels or less go TWO per row block
// (depth slices z, z+1 side by side: the producer zeroes the two neighbour values that cross the seam)
// NBU = 3 (VOL, rows of exactly 12 used pixel pairs -- 24-pixel planes or two 12-pixel slices -- and 8-row blocks): the block's 8 x 12
// used columns are numbered row * 12 + pair and fill THREE 32-column blocks completely, instead of 12 of 16 pairs in each of four:
// a quarter of the MFMAs, operand reads and epilogue rounds gone
template <int OUTS, bool STATS = false, bool POOL = false, bool FIN = false, bool VOL = false, int NBU = 4>
__global__ __launch_bounds__(512) void conv_wino1d_kernel(ConvArgs a, int total_tiles) {
  extern __shared__ __align__(16) float lds[];
  unsigned* const ldsw = reinterpret_cast<unsigned*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  const int xi = wave & 3, chh = wave >> 2;
  const int HW = a.H * a.W;
  const int n_cc = a.Cin / Y_KC, n_ct = a.Cout / 32;
  static_assert(!VOL || (!STATS && !POOL && !FIN), "VOL: plain epilogues");
  static_assert(NBU == 4 || (NBU == 3 && VOL), "compact columns: the volume form");
  const int n_chunks = VOL ? 3 * n_cc : n_cc;                  // even, >= 2 (launcher)
  const int CS = VOL ? a.D * HW : HW;                          // channel stride of input / output
  const bool pack = VOL && a.W <= 12;                          // two depth slices per row block
  const int seam = a.W >> 1;                                   // pack: first pixel pair of the second slice
  unsigned long long t0 = 0, t1 = 0, t_loop = 0, t_epi = 0, tq = 0;   // tuning stamps (a.dbg)
#ifdef IPDM_W1D_TRACE
  unsigned long long te[4] = {0, 0, 0, 0};                            // epilogue phases: exchange stores, barrier, transform + stores, barrier
#endif
  if (a.dbg) t0 = __builtin_amdgcn_s_memtime();

  // ---- this workgroup's tile list (XCD-aware, as conv_wino_bx3_wide_kernel) ----
  const int S = gridDim.x / 8;
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const int q8 = total_tiles / 8, r8 = total_tiles % 8;
  const int x_start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int t_end = x_start + q8 + (xcd < r8 ? 1 : 0);
  int tile = x_start + slot;
  if (tile >= t_end) return;
  // phase groups (tuning, IPDM_W1D_STAGGER): group slot % 4 starts g * stagger cycles late, so that the workgroups' epilogues -- all
  // 256 hit HBM in the same microseconds otherwise -- are spread over the pass
  if (a.phase_step > 0) {
    const int g = slot & a.phase_mask;
    for (int i = 0; i < g * a.phase_step; ++i) __builtin_amdgcn_s_sleep(8);             // ~8 x 64 cycles each
  }

  struct Geo { int b, y0, x0, cob, z; };
  auto geo_of = [&](int L) {
    Geo g;
    g.cob = L % a.co_tiles;
    int t = L / a.co_tiles;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    g.y0 = (t % a.tiles_y) * Y_ROWS;
    g.x0 = tx * (2 * Y_TX);
    t /= a.tiles_y;
    g.z = 0;
    if constexpr (VOL) {
      const int nz = pack ? (a.D + 1) >> 1 : a.D;
      g.z = (t % nz) * (pack ? 2 : 1);
      t /= nz;
    }
    g.b = t;
    return g;
  };

  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((size_t)a.B * a.Cin * CS * 4), 0x00020000);
  float* const rs = lds + Y_STAGE;

  // raw stage: wave w fetches channels 2w, 2w+1 of a chunk (2 x 100 quads) into its own block, four 16-byte LDS-DMA instructions
  int dma_off[4];                                               // VOL: without the plane's offset (it changes with the chunk's depth tap)
  int dma_b = 0;
  [[maybe_unused]] int dma_z = 0, dma_isb = 0;                  // VOL: the tile's first depth slice; bit k: piece k reads the SECOND slice
  auto set_dma_geo = [&](const Geo& g) {
    dma_b = g.b;
    if constexpr (VOL) {
      dma_z = g.z;
      dma_isb = 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = k * 64 + lane;
      const int cl = e / Y_QN, qq = e - cl * Y_QN;
      const int cin = 2 * wave + cl;
      const int rr = qq / Y_QC, qc = qq - rr * Y_QC;
      const int gy = g.y0 - 1 + rr;
      int gx0 = g.x0 - 4 + 4 * qc;
      bool okx = gx0 >= 0 && gx0 < a.W;
      if constexpr (VOL) {
        if (pack) {                                             // raw columns 4 .. 4+W: slice z, 4+W .. 4+2W: slice z+1
          const bool in_b = gx0 >= a.W && gx0 < 2 * a.W;
          okx = okx || in_b;
          gx0 -= in_b ? a.W : 0;
          dma_isb |= in_b ? 1 << k : 0;
        }
      }
      const bool ok = e < 2 * Y_QN && gy >= 0 && gy < a.H && okx;
      dma_off[k] = ok ? (cin * CS + gy * a.W + gx0) * 4 : 0x40000000;
    }
  };
  auto issue_dma = [&](int chunk) {
    int cc = chunk;
    [[maybe_unused]] int off_a = 0, off_b = 0;
    [[maybe_unused]] bool ok_a = true, ok_b = false;
    if constexpr (VOL) {
      const int kz = chunk / n_cc;
      cc = chunk - kz * n_cc;
      const int za = dma_z + kz - 1, zb = za + 1;
      ok_a = (unsigned)za < (unsigned)a.D;
      ok_b = pack && (unsigned)zb < (unsigned)a.D && dma_z + 1 < a.D;
      off_a = za * HW * 4;
      off_b = zb * HW * 4;
    }
    [[maybe_unused]] const int soff = (int)(((size_t)dma_b * a.Cin + cc * Y_KC) * CS * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int off = dma_off[k];
      if constexpr (VOL) {
        const bool isb = (dma_isb >> k) & 1;
        const bool ok = off != 0x40000000 && (isb ? ok_b : ok_a);
        off = ok ? off + (isb ? off_b : off_a) : 0x40000000;
      }
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(rs + (wave * 4 + k) * 256), 16,
                                               off, soff, 0, 0);
#endif
    }
  };

  // ---- producer: this thread's three (row, pixel pair) items of the wave's channel pair ----
  const int pt = lane & 15, rsub = lane >> 4;
  const bool pfirst = pt == 0, plast = pt == Y_TX - 1;
  struct Raw { float2 qa, qb; float ea, eb; };
  Raw raw[3];
  auto read_items = [&]() {
    const float* rp = rs + wave * 1024 + rsub * Y_RC4 + 2 * pt + 2;       // -> column 2pt+2 of row rsub, first channel
    const int edge = pfirst ? 1 : 4;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      raw[it].qa = *reinterpret_cast<const float2*>(rp + it * 4 * Y_RC4 + 2);
      raw[it].ea = rp[it * 4 * Y_RC4 + edge];
      raw[it].qb = *reinterpret_cast<const float2*>(rp + Y_QN * 4 + it * 4 * Y_RC4 + 2);
      raw[it].eb = rp[Y_QN * 4 + it * 4 * Y_RC4 + edge];
    }
  };
  [[maybe_unused]] float hx_in = HX1_PRESCALE, hx_out = 1.f;
  auto hx_scales_of = [&](int b, float& s_in, float& s_out) {
    s_in = HX1_PRESCALE;
    s_out = 1.f;
    if (a.in_amax) {
      float sd, si;
      hx_dynamic_scale(ipdm_amax_read(a.in_amax + (size_t)b * IPDM_AMAX_SLOT), sd, si);
      s_in = HX1_PRESCALE * sd;
      s_out = si;
    }
  };
  // word offset of this thread's stores inside a staged row: channel pair cp = wave -> k half cp >> 2, q = cp & 3
  const int v_st = ((wave & 3) >> 1) * 64 + (wave >> 2) * 32 + (wave & 1) * 16 + pt;
  // FIN: coefficients of the wave's two channels and the in-image predicates of the chunk being staged
  [[maybe_unused]] float fin_mu[2] = {0.f, 0.f}, fin_sc[2] = {1.f, 1.f}, fin_sh[2] = {0.f, 0.f};
  [[maybe_unused]] int fin_row = 0;                             // image row of this thread's first item
  [[maybe_unused]] bool fin_pair = true, fin_edge = true;
  [[maybe_unused]] auto set_fin = [&](int b, int chunk, int y0, int x0) {
    if constexpr (FIN) {
      const float* cf = a.coef + ((size_t)b * a.Cin + chunk * Y_KC + 2 * wave) * 3;
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        fin_mu[cl] = cf[cl * 3];
        fin_sc[cl] = cf[cl * 3 + 1];
        fin_sh[cl] = cf[cl * 3 + 2];
      }
      fin_row = y0 - 1 + rsub;
      fin_pair = x0 + 2 * pt < a.W;
      fin_edge = pfirst ? x0 > 0 : x0 + 2 * Y_TX < a.W;
    }
  };
  auto xform_item = [&](auto itc, float (&va)[4], float (&vb)[4]) {
    constexpr int it = decltype(itc)::value;
    Raw r = raw[it];
    if constexpr (FIN) {
      const bool rok = (unsigned)(fin_row + 4 * it) < (unsigned)a.H;
      const bool pok = rok && fin_pair, eok = rok && fin_edge;
      auto fin = [&](float v, int cl, bool ok) {
        const float t = fast_elu((v - fin_mu[cl]) * fin_sc[cl] + fin_sh[cl]);
        return ok ? t : 0.f;                                      // the padding stays zero
      };
      r.qa.x = fin(r.qa.x, 0, pok); r.qa.y = fin(r.qa.y, 0, pok); r.ea = fin(r.ea, 0, eok);
      r.qb.x = fin(r.qb.x, 1, pok); r.qb.y = fin(r.qb.y, 1, pok); r.eb = fin(r.eb, 1, eok);
    }
    const float la = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, r.qa.y), 0x111, 0xf, 0xf, true));
    const float ra = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, r.qa.x), 0x101, 0xf, 0xf, true));
    const float lb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, r.qb.y), 0x111, 0xf, 0xf, true));
    const float rb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, r.qb.x), 0x101, 0xf, 0xf, true));
    float a0 = pfirst ? r.ea : la, a3 = plast ? r.ea : ra;
    float b0 = pfirst ? r.eb : lb, b3 = plast ? r.eb : rb;
    if constexpr (VOL) {
      if (pack) {                                               // neighbours across the seam belong to the other slice: padding
        const bool s_l = pt == seam, s_r = pt == seam - 1;
        a0 = s_l ? 0.f : a0; b0 = s_l ? 0.f : b0;
        a3 = s_r ? 0.f : a3; b3 = s_r ? 0.f : b3;
      }
    }
    va[0] = a0 - r.qa.y; va[1] = r.qa.x + r.qa.y; va[2] = r.qa.y - r.qa.x; va[3] = r.qa.x - a3;
    vb[0] = b0 - r.qb.y; vb[1] = r.qb.x + r.qb.y; vb[2] = r.qb.y - r.qb.x; vb[3] = r.qb.x - b3;
  };
  auto store_item = [&](unsigned* st, auto itc, const float (&va)[4], const float (&vb)[4], auto lo_c, auto hi_c) {
    constexpr int it = decltype(itc)::value;
    unsigned* vs = st + (it * 4 + rsub) * Y_RS + v_st;
#pragma unroll
    for (int x = decltype(lo_c)::value; x < decltype(hi_c)::value; ++x) {
      unsigned hp, lp;
      split2_pk_scaled(va[x], vb[x], hx_in, hp, lp);
      vs[(x * 2 + 0) * 128] = hp;
      vs[(x * 2 + 1) * 128] = lp;
    }
  };

  // ---- consumer operands ----
  const uint4* wq = reinterpret_cast<const uint4*>(a.wt);
  const size_t pos_stride = (size_t)n_cc * n_ct * 128;
  auto load_A = [&](uint4 (&fr)[2][2], int ky, int chunk, int cob) {
    int cc = chunk, kz = 0;
    if constexpr (VOL) {
      kz = chunk / n_cc;
      cc = chunk - kz * n_cc;
    }
    const uint4* base = wq + (size_t)((kz * 3 + ky) * 4 + xi) * pos_stride + ((size_t)cc * n_ct + cob * 4 + chh * 2) * 128 + lane;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int s = 0; s < 2; ++s) fr[c][s] = base[c * 128 + s * 64];
  };
  const int b_lane = (j >> 4) * Y_RS + xi * 256 + h * 32 + (j & 15);
  [[maybe_unused]] int b_cmp[3] = {0, 0, 0};                    // NBU == 3: this lane's (row, pair) of column 32 nb + j, as a word offset
  if constexpr (NBU == 3) {
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
      const int n = 32 * nb + j;
      b_cmp[nb] = (n / 12) * Y_RS + xi * 256 + h * 32 + n % 12;
    }
  }
  auto load_B = [&](uint4 (&fr)[2], const unsigned* cur, auto kyc, auto nbc) {
    constexpr int ky = decltype(kyc)::value, nb = decltype(nbc)::value;
    const unsigned* bp = NBU == 3 ? cur + ky * Y_RS + b_cmp[nb < 3 ? nb : 0] : cur + (2 * nb + ky) * Y_RS + b_lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) fr[s] = make_uint4(bp[s * 128], bp[s * 128 + 16], bp[s * 128 + 64], bp[s * 128 + 80]);
  };

  f32x16 acc[2][NBU];
#pragma unroll
  for (int i = 0; i < 2 * NBU; ++i)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[i / NBU][i % NBU][rr] = 0.f;

  auto lds_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue of the first tile ----
  Geo cur_g = geo_of(tile);
  uint4 afr[2][2][2];
  load_A(afr[0], 0, 0, cur_g.cob);
  load_A(afr[1], 1, 0, cur_g.cob);
  hx_scales_of(cur_g.b, hx_in, hx_out);
  set_dma_geo(cur_g);
  issue_dma(0);
  __builtin_amdgcn_s_waitcnt(0);                               // (the raw blocks are wave-private: no barrier before reading them)
  set_fin(cur_g.b, 0, cur_g.y0, cur_g.x0);
  read_items();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0): chunk 0's rows are in registers
  issue_dma(1);                                                // ... so chunk 1 lands while chunk 0 is transformed and stored
  __builtin_amdgcn_sched_barrier(0);
  static_for<3>([&](auto itc) {
    float va[4], vb[4];
    xform_item(itc, va, vb);
    store_item(ldsw, itc, va, vb, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
  });
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  if (a.dbg) t1 = tq = __builtin_amdgcn_s_memtime();
  [[maybe_unused]] int pend_b = cur_g.b;
  [[maybe_unused]] unsigned pend_o = 0u, pend_a = 0u;
  auto flush_amax = [&]() {
    if (lane == 0) {
      if (a.amax_out) ipdm_amax_atomic(a.amax_out + (size_t)pend_b * IPDM_AMAX_SLOT, wave, __builtin_bit_cast(float, pend_o));
      if (a.amax_act) ipdm_amax_atomic(a.amax_act + (size_t)pend_b * IPDM_AMAX_SLOT, wave, __builtin_bit_cast(float, pend_a));
    }
  };

  while (true) {
    const int next_tile = tile + S;
    const bool has_next = next_tile < t_end;
    const Geo next_g = geo_of(has_next ? next_tile : tile);
    for (int ch2 = 0; ch2 < n_chunks; ch2 += 2) {
      static_for<2>([&](auto cpc) {
        constexpr int cpar = decltype(cpc)::value;
        const int ch = ch2 + cpar;
        const unsigned* cur = ldsw + cpar * Y_S1;
        unsigned* nxt = ldsw + (1 - cpar) * Y_S1;
        const bool dma_next = ch + 2 >= n_chunks;
        const int dma_chunk = dma_next ? ch + 2 - n_chunks : ch + 2;
        if (ch + 2 == n_chunks) set_dma_geo(next_g);
        const bool a_next = ch + 1 >= n_chunks;                 // this chunk stages chunk 0 of the NEXT tile
        const int a_chunk = a_next ? 0 : ch + 1;
        const int a_cob = a_next ? next_g.cob : cur_g.cob;
        if (a_next && next_g.b != cur_g.b) {
          [[maybe_unused]] float unused_out;
          hx_scales_of(next_g.b, hx_in, unused_out);
        }
        if constexpr (FIN) {
          if (a_next) set_fin(next_g.b, 0, next_g.y0, next_g.x0);
          else set_fin(cur_g.b, ch + 1, cur_g.y0, cur_g.x0);
        }
        uint4 bsh[2][2];
        // the wave's DMA of chunk ch+1 (issued a chunk ago) has landed: three fragment groups are younger, two of them still wanted
        __builtin_amdgcn_s_waitcnt(0x0F78);                     // vmcnt(8)
        read_items();
        load_B(bsh[0], cur, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): the raw rows are in registers
        issue_dma(dma_chunk);
        __builtin_amdgcn_sched_barrier(0);
        float va[4], vb[4];
        static_for<3 * NBU>([&](auto ssc) {
          constexpr int ss = decltype(ssc)::value;
          constexpr int ky = ss / NBU, nb = ss % NBU;
          constexpr int aslot = (cpar * 3 + ky) & 1;
          if constexpr (ss < 3 * NBU - 1)
            load_B(bsh[(ss + 1) & 1], cur, std::integral_constant<int, (ss + 1) / NBU>{}, std::integral_constant<int, (ss + 1) % NBU>{});
          // producer work of item ky, dealt over the sub-steps of a filter row
          constexpr int it = ky;
          if constexpr (nb == 0) xform_item(std::integral_constant<int, it>{}, va, vb);
          if constexpr (nb == 1) store_item(nxt, std::integral_constant<int, it>{}, va, vb, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
          if constexpr (nb == 2) store_item(nxt, std::integral_constant<int, it>{}, va, vb, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
          const f16x8 bh = __builtin_bit_cast(f16x8, bsh[ss & 1][0]), bl = __builtin_bit_cast(f16x8, bsh[ss & 1][1]);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            f32x16 v = acc[c][nb];
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[aslot][c][1]), bh, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[aslot][c][0]), bl, v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[aslot][c][0]), bh, v, 0, 0, 0);
            acc[c][nb] = v;
          }
          if constexpr (nb == NBU - 1) {
            // this filter row's fragments are free: request the row after next (same slot)
            if constexpr (ky == 0) load_A(afr[aslot], 2, ch, cur_g.cob);
            if constexpr (ky == 1) load_A(afr[aslot], 0, a_chunk, a_cob);
            if constexpr (ky == 2) load_A(afr[aslot], 1, a_chunk, a_cob);
          }
          if constexpr (ss < 3 * NBU - 1) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // next sub-step's operand reads first
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
            if constexpr (nb == 1 || nb == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        lds_barrier();
      });
    }

    if (a.dbg) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_loop += t - tq;
      tq = t;
    }
    // ---- epilogue: eight rounds (channel tile c, column block nb) through M[rnd & 1][xi][co 64][col 32] ----
    constexpr bool W_OUT = (OUTS & 1) != 0, W_ACT = (OUTS & 2) != 0;
    float* const ms = lds + Y_S1;
    const int ecol = tid & 31, ecg = tid >> 5;                  // this thread: column, channels k*16 + ecg of the round's 64
    const int co0 = cur_g.cob * Y_CO;
    const float* const scale_p = reinterpret_cast<const float*>(wq + (VOL ? 36 : 12) * pos_stride);
    const int oHW = POOL ? HW >> 2 : (VOL ? CS : HW), oW = POOL ? a.W >> 1 : a.W;     // the written tensor's channel stride / row
    const size_t tile_base = (((size_t)cur_g.b * a.Cout + co0) * oHW + (size_t)cur_g.z * HW + (size_t)(POOL ? cur_g.y0 >> 1 : cur_g.y0) * oW +
                              (POOL ? cur_g.x0 >> 1 : cur_g.x0)) * 4;
    // pack: pixel pairs seam .. 2 seam - 1 are the second slice's pairs 0 .. seam - 1
    const bool e_isb = pack && (ecol & 15) >= seam;
    const int e_pair = e_isb ? (ecol & 15) - seam : (ecol & 15);
    unsigned eoff4 = 4u * (unsigned)(ecg * oHW + (POOL ? (ecol & 15) : (ecol >> 4) * oW + 2 * e_pair + (e_isb ? HW : 0)));
    asm volatile("" : "+v"(eoff4));
    // NBU == 3: column 32 nb + ecol is (row, pair) = divmod(., 12); pairs 6 .. 11 of a two-slice block are the second slice's 0 .. 5
    [[maybe_unused]] auto cmp_off = [&](int nb) -> unsigned {
      const int n = 32 * nb + ecol, row = n / 12, pr = n % 12;
      const bool isb = pack && pr >= 6;
      return 4u * (unsigned)(ecg * oHW + row * oW + 2 * (isb ? pr - 6 : pr) + (isb ? HW : 0));
    };
    auto boff = [&](int c, int nb, int k) -> unsigned {
      if constexpr (NBU == 3) return 4u * (unsigned)((((k >> 1) * 2 + c) * 32 + (k & 1) * 16) * oHW) + cmp_off(nb);
      return 4u * (unsigned)((((k >> 1) * 2 + c) * 32 + (k & 1) * 16) * oHW + (POOL ? nb : 2 * nb) * oW) + eoff4;
    };
    auto in_range = [&](int nb) {
      if constexpr (NBU == 3) return !(pack && (32 * nb + ecol) % 12 >= 6) || cur_g.z + 1 < a.D;     // (8 full rows, 12 used pairs)
      const bool row_ok = cur_g.y0 + 2 * nb + (ecol >> 4) < a.H;
      if constexpr (VOL) {
        if (pack) return row_ok && (ecol & 15) < 2 * seam && (!e_isb || cur_g.z + 1 < a.D);
      }
      return row_ok && cur_g.x0 + 2 * (ecol & 15) < a.W;
    };
    const bool has_res = a.residual != nullptr, has_bias = a.bias != nullptr;
    const bool elu = a.act_out == IPDM_ACT_ELU;                 // otherwise the activated copy is the identity (launcher)
    const float* const res_p = has_res ? a.residual : a.wt;
    const float* const bias_p = has_bias ? a.bias : a.wt;
    const char* const res_b = reinterpret_cast<const char*>(res_p) + (has_res ? tile_base : 0);
    [[maybe_unused]] char* const out_b = reinterpret_cast<char*>(a.out) + tile_base;
    [[maybe_unused]] char* const act_b = reinterpret_cast<char*>(a.out_act) + tile_base;
    constexpr int NV = POOL ? 1 : 2;                            // stored values per (thread, channel)
    typedef float vecv __attribute__((ext_vector_type(NV)));
    vecv resv[2][4];
    // (unconditional loads -- an absent operand or an out-of-range column reads the weight blob -- so that hipcc can count them)
    auto prefetch = [&](auto rc) {
      constexpr int rnd = decltype(rc)::value;
      constexpr int c = rnd / NBU, nb = rnd % NBU, bf = rnd & 1;
      const bool res_ok = has_res && in_range(nb) && (!POOL || (ecol >> 4) == 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned ob = res_ok ? boff(c, nb, k) : 0u;
        resv[bf][k] = *reinterpret_cast<const vecv*>(res_b + ob);
      }
    };
    if (cur_g.b != pend_b) {
      flush_amax();
      pend_b = cur_g.b;
      pend_o = pend_a = 0u;
    }
    // this pass's 128 (scale, bias) pairs go through a small LDS table: as vector loads prefetched a round ahead they held 16
    // registers through the whole epilogue
    float* const es = lds + Y_S1 + 2 * Y_M;
    if (tid < Y_CO) {
      es[tid] = scale_p[co0 + tid] * hx_out;
      es[Y_CO + tid] = has_bias ? bias_p[cur_g.b * a.bias_bstride + co0 + tid] : 0.f;
    }
    prefetch(std::integral_constant<int, 0>{});
    float amx_o = 0.f, amx_a = 0.f;
    [[maybe_unused]] float st_k[4], st_s1[4], st_s2[4];          // STATS: shift, sum (v - K), sum (v - K)^2 of this thread's four channels
    [[maybe_unused]] int st_c0 = 0, st_c1 = 0;                   //        in-range values of the two half-waves
#ifdef IPDM_W1D_TRACE
#define W1D_TE(k) if (a.dbg) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); te[k] += t_ - tl; tl = t_; }
    unsigned long long tl = a.dbg ? __builtin_amdgcn_s_memtime() : 0;
#else
#define W1D_TE(k)
#endif
    static_for<2 * NBU>([&](auto rc) {
      constexpr int rnd = decltype(rc)::value;
      constexpr int c = rnd / NBU, nb = rnd % NBU, bf = rnd & 1;
      float* const mw = ms + (rnd & 1) * Y_M;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int col = (rr & 3) + 8 * (rr >> 2) + 4 * h;
        mw[(xi * 64 + chh * 32 + col) * 32 + j] = acc[c][nb][rr];
        acc[c][nb][rr] = 0.f;
      }
      W1D_TE(0);
      __syncthreads();
      W1D_TE(1);
      if constexpr (rnd < 2 * NBU - 1) prefetch(std::integral_constant<int, rnd + 1>{});
      const bool inr = in_range(nb) && (!POOL || (ecol >> 4) == 0);   // this thread stores (POOL: the windows' upper rows do)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float m[2][4], sc[2], bi[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int k = half * 2 + kk;
          const int cidx = ((k >> 1) * 2 + c) * 32 + (k & 1) * 16 + ecg;
          sc[kk] = es[cidx];
          bi[kk] = es[Y_CO + cidx];
#pragma unroll
          for (int x = 0; x < 4; ++x) m[kk][x] = mw[(x * 64 + k * 16 + ecg) * 32 + ecol];
        }
        vecv ov[2], ev[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int k = half * 2 + kk;
          float yv[2];
          yv[0] = __builtin_fmaf(m[kk][0] + m[kk][1] + m[kk][2], sc[kk], bi[kk]);
          yv[1] = __builtin_fmaf(m[kk][1] - m[kk][2] - m[kk][3], sc[kk], bi[kk]);
          if constexpr (POOL) {
            // (y[::2,::2] + y[1::2,::2] + y[::2,1::2] + y[1::2,1::2]) / 4 in the reference's order (layers.py:291-313)
            const float p0 = __shfl_xor(yv[0], 16, 64), p1 = __shfl_xor(yv[1], 16, 64);
            yv[0] = (((yv[0] + p0) + yv[1]) + p1) * 0.25f;
          }
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            float y = yv[v], rv = yv[v];
            y += has_res ? resv[bf][k][v] : 0.f;
            rv = a.res_second ? rv : y;
            y *= a.out_scale;
            rv *= a.out_scale;
            ov[kk][v] = rv;
            if constexpr (STATS) {
              if constexpr (nb == 0) {
                if (v == 0) {
                  // the shift K is taken FROM the data (the half-wave's first value of the channel tile's first round), so that
                  // the subtraction below cancels nothing the spread of the data does not
                  const float k0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rv), 0));
                  const float k1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rv), 32));
                  st_k[k] = h ? k1 : k0;
                  st_s1[k] = st_s2[k] = 0.f;
                }
              }
              const float d = inr ? rv - st_k[k] : 0.f;
              st_s1[k] += d;
              st_s2[k] = __builtin_fmaf(d, d, st_s2[k]);
            }
            if constexpr (W_OUT) amx_o = fmaxf(amx_o, inr ? fabsf(rv) : 0.f);
            if constexpr (W_ACT) {
              const float e = elu ? fast_elu(y) : y;
              ev[kk][v] = e;
              amx_a = fmaxf(amx_a, inr ? fabsf(e) : 0.f);
            }
          }
        }
        if (inr) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const unsigned ob = boff(c, nb, half * 2 + kk);
#if defined(IPDM_W1D_NO_NT)
            if constexpr (W_OUT) *reinterpret_cast<vecv*>(out_b + ob) = ov[kk];
#elif defined(IPDM_W1D_NT_BIG)
            if constexpr (W_OUT) {                                  // (experiment: streaming stores only for tensors beyond the MALL)
              if ((size_t)a.B * a.Cout * HW > (size_t)IPDM_W1D_NT_BIG * 262144) __builtin_nontemporal_store(ov[kk], reinterpret_cast<vecv*>(out_b + ob));
              else *reinterpret_cast<vecv*>(out_b + ob) = ov[kk];
            }
#else
            if constexpr (W_OUT) __builtin_nontemporal_store(ov[kk], reinterpret_cast<vecv*>(out_b + ob));
#endif
#ifdef IPDM_W1D_ACT_NT
            if constexpr (W_ACT) __builtin_nontemporal_store(ev[kk], reinterpret_cast<vecv*>(act_b + ob));
#else
            if constexpr (W_ACT) *reinterpret_cast<vecv*>(act_b + ob) = ev[kk];
#endif
          }
        }
      }
      if constexpr (STATS) {
        const unsigned long long bal = __ballot(inr);
        if constexpr (nb == 0) st_c0 = st_c1 = 0;
        st_c0 += NV * __popcll(bal & 0xffffffffull);
        st_c1 += NV * __popcll(bal >> 32);
        if constexpr (nb == NBU - 1) {
          const float cnt = (float)(h ? st_c1 : st_c0);
          const int tb = (cur_g.y0 / Y_ROWS) * a.tiles_x + cur_g.x0 / (2 * Y_TX);
          const int n_tb = a.tiles_x * a.tiles_y;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float s1 = half_wave_sum(st_s1[k]), s2 = half_wave_sum(st_s2[k]);
            if (ecol == 0) {
              const int co = co0 + ((k >> 1) * 2 + c) * 32 + (k & 1) * 16 + ecg;
              float* sp = a.stats + (((size_t)cur_g.b * a.Cout + co) * n_tb + tb) * 3;
              const float dm = cnt > 0.f ? s1 / cnt : 0.f;
              sp[0] = cnt; sp[1] = st_k[k] + dm; sp[2] = fmaxf(s2 - s1 * dm, 0.f);
            }
          }
        }
      }
      W1D_TE(2);
    });
    __syncthreads();                                            // the next tile's second chunk is staged over the exchange buffers
    W1D_TE(3);
    if (a.amax_out || a.amax_act) {
      const unsigned mo = __builtin_bit_cast(unsigned, ipdm_wave_max(amx_o)), ma = __builtin_bit_cast(unsigned, ipdm_wave_max(amx_a));
      const unsigned uo = __builtin_amdgcn_readfirstlane(mo), ua = __builtin_amdgcn_readfirstlane(ma);
      pend_o = pend_o > uo ? pend_o : uo;
      pend_a = pend_a > ua ? pend_a : ua;
    }
    if (a.dbg) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_epi += t - tq;
      tq = t;
    }
    if (!has_next) break;
    tile = next_tile;
    if (next_g.b != cur_g.b) hx_scales_of(next_g.b, hx_in, hx_out);
    cur_g = next_g;
  }
  flush_amax();
  if (a.dbg && tid == 0) {
    unsigned long long* d4 = a.dbg + (size_t)blockIdx.x * 4;
    d4[0] = tq - t0; d4[1] = t1 - t0; d4[2] = t_loop; d4[3] = t_epi;
#ifdef IPDM_W1D_TRACE
    unsigned long long* d8 = a.dbg + (size_t)gridDim.x * 4 + (size_t)blockIdx.x * 4;
    d8[0] = te[0]; d8[1] = te[1]; d8[2] = te[2]; d8[3] = te[3];
#endif
  }
}

}  // namespace

bool wino1d_ok(const ConvArgs& a) {
  if (!(a.D == 1 && a.dil == 1 && a.Cin % (2 * Y_KC) == 0 && a.Cout % Y_CO == 0)) return false;
  if (a.coef ? a.act != IPDM_ACT_ELU : a.act != IPDM_ACT_NONE) return false;              // fused input: InstanceNorm++ + ELU only
  if (a.out_act && a.act_out != IPDM_ACT_ELU && a.act_out != IPDM_ACT_COPY) return false;    // the epilogue's branch-free activations
  if ((size_t)a.B * a.Cin * a.H * a.W * 4 >= 0x3fffffffull) return false;
  if (a.stats && !a.out) return false;
  return a.H % 2 == 0 && a.W % 4 == 0 && a.H >= 8 && a.W >= 32 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
}

int conv_wino1d_launch(ConvArgs a, hipStream_t s) {
  // IPDM_W1D_STAGGER=<cycles>[:<groups 2|4|8>[:<min passes per workgroup>]]: start delay per phase group
  static int stagger = -1, st_groups = 4, st_minpass = 0;
  if (stagger < 0) {
    const char* e = getenv("IPDM_W1D_STAGGER");
    stagger = 0;
    if (e) {
      int c = 0, g = 4, m = 0;
      const int n = sscanf(e, "%d:%d:%d", &c, &g, &m);
      if (n >= 1) stagger = c / 512;
      if (n >= 2 && (g == 2 || g == 4 || g == 8)) st_groups = g;
      if (n >= 3) st_minpass = m;
    }
  }
  a.tiles_x = (a.W + 2 * Y_TX - 1) / (2 * Y_TX);
  a.tiles_y = (a.H + Y_ROWS - 1) / Y_ROWS;
  a.co_tiles = a.Cout / Y_CO;
  const int64_t nblk = (int64_t)a.B * a.tiles_x * a.tiles_y * a.co_tiles;
  if (nblk > 0x7fffffff) return IPDM_EUNSUPPORTED;
  a.phase_step = stagger > 0 && nblk >= (int64_t)st_minpass * 256 ? stagger : 0;
  a.phase_mask = st_groups - 1;
  static bool attr_set = false;
  if (!attr_set) {
#define W1D_K(O, S_, P_) reinterpret_cast<const void*>(conv_wino1d_kernel<O, S_, P_, false>), reinterpret_cast<const void*>(conv_wino1d_kernel<O, S_, P_, true>)
    const void* kernels[] = {W1D_K(1, false, false), W1D_K(2, false, false), W1D_K(3, false, false), W1D_K(1, true, false), W1D_K(3, true, false),
                             W1D_K(1, false, true),  W1D_K(2, false, true),  W1D_K(3, false, true),  W1D_K(1, true, true),  W1D_K(3, true, true)};
#undef W1D_K
    for (const void* k : kernels) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Y_LDS_BYTES);
      if (e != hipSuccess) return (int)e;
    }
    attr_set = true;
  }
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
    cus = 256;
  const int per_xcd = (int)((nblk + 7) / 8);
  const int S = per_xcd < cus / 8 ? per_xcd : cus / 8;
  const int outs = (a.out ? 1 : 0) | (a.out_act ? 2 : 0);
#define W1D_LAUNCH(O, S_, P_)                                                                                                    \
  do {                                                                                                                           \
    if (a.coef) hipLaunchKernelGGL((conv_wino1d_kernel<O, S_, P_, true>), dim3((unsigned)(8 * S)), dim3(512), Y_LDS_BYTES, s, a, (int)nblk); \
    else hipLaunchKernelGGL((conv_wino1d_kernel<O, S_, P_, false>), dim3((unsigned)(8 * S)), dim3(512), Y_LDS_BYTES, s, a, (int)nblk);      \
  } while (0)
  if (a.pool2) {
    if (a.stats) { if (outs == 3) W1D_LAUNCH(3, true, true); else W1D_LAUNCH(1, true, true); }
    else if (outs == 3) W1D_LAUNCH(3, false, true);
    else if (outs == 1) W1D_LAUNCH(1, false, true);
    else W1D_LAUNCH(2, false, true);
  } else if (a.stats) {
    if (outs == 3) W1D_LAUNCH(3, true, false); else W1D_LAUNCH(1, true, false);
  } else if (outs == 3) W1D_LAUNCH(3, false, false);
  else if (outs == 1) W1D_LAUNCH(1, false, false);
  else W1D_LAUNCH(2, false, false);
#undef W1D_LAUNCH
  return ipdm_launch_status();
}

int conv_wino1d_weights(const float* w, void* U, int Cout, int Cin, int npos, hipStream_t s) {
  const int n_cc = (Cin + 15) / 16, n_ct = (Cout + 31) / 32;
  const int64_t total = (int64_t)npos * n_cc * n_ct * 512;
  float* inv_scale = reinterpret_cast<float*>(static_cast<char*>(U) + total * 2 * 2);
  hipLaunchKernelGGL(wino1d_scale_kernel, dim3(n_ct * 32), dim3(256), 0, s, w, inv_scale, Cout, Cin, n_ct * 32, npos);
  hipLaunchKernelGGL(wino1d_weight_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, s, w, (unsigned short*)U, inv_scale, Cout,
                     Cin, n_cc, n_ct, npos);
  return ipdm_launch_status();
}

}  // namespace ipdm_conv

using namespace ipdm_conv;

extern "C" int64_t ipdm_conv_wino1d_weight_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return -1;
  return (int64_t)12 * ((Cin + 15) / 16) * ((Cout + 31) / 32) * 2048 + (int64_t)((Cout + 31) / 32) * 32 * 4;
}

extern "C" int ipdm_conv_wino1d_pack_weight(const float* w, void* U, int Cout, int Cin, void* stream) {
  IPDM_REQUIRE(w && U && Cout > 0 && Cin > 0);
  return conv_wino1d_weights(w, U, Cout, Cin, 12, ipdm_stream(stream));
}

extern "C" int ipdm_conv2d_wino1d_supported(int Cin, int Cout, int H, int W) {
  ConvArgs a;
  a.x = nullptr; a.out = a.out_act = nullptr; a.act_out = IPDM_ACT_NONE; a.coef = nullptr; a.act = IPDM_ACT_NONE; a.dil = 1; a.D = 1; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.B = 1;
  return wino1d_ok(a) ? 1 : 0;
}

static int wino1d_entry(const float* x, const void* U, const float* bias, const float* coef, int act, const float* residual,
                        float* out, float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int pool2, float* stats, const ipdm_conv_ext_t* ext,
                        void* stream) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && U && (out || out_act) && x != out && x != out_act);
  ConvArgs a;
  a.x = x; a.wt = (const float*)U; a.bias = bias; a.coef = coef; a.residual = residual; a.out = out; a.out_act = out_act;
  a.act_out = act_out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = 1; a.act = coef ? act : IPDM_ACT_NONE;
  a.D = 1; a.kd = 1; a.tiles_x = a.tiles_y = a.co_tiles = 0; a.dbg = conv_debug_stamps();
  a.hx = 1;
  a.stats = stats;
  a.pool2 = pool2 ? 1 : 0;
  conv_apply_ext(a, ext, 1);
  if (!wino1d_ok(a)) return IPDM_EUNSUPPORTED;
  return conv_wino1d_launch(a, ipdm_stream(stream));
}

extern "C" int ipdm_conv2d_wino1d_f32(const float* x, const void* U, const float* bias, const float* coef, int act,
                                      const float* residual, float* out, float* out_act, int act_out, int B, int Cin, int Cout,
                                      int H, int W, int pool2, const ipdm_conv_ext_t* ext, void* stream) {
  IPDM_REQUIRE(coef || act == IPDM_ACT_NONE);
  return wino1d_entry(x, U, bias, coef, act, residual, out, out_act, act_out, B, Cin, Cout, H, W, pool2, nullptr, ext, stream);
}

// statistics epilogue: partials per plane (0: the layer shape is not served) and the call that fills stats[B][Cout][partials][3]
extern "C" int ipdm_conv2d_wino1d_stats_partials(int Cin, int Cout, int H, int W) {
  if (!ipdm_conv2d_wino1d_supported(Cin, Cout, H, W)) return 0;
  return ((W + 2 * Y_TX - 1) / (2 * Y_TX)) * ((H + Y_ROWS - 1) / Y_ROWS);
}

extern "C" int ipdm_conv2d_wino1d_stats_f32(const float* x, const void* U, const float* bias, const float* coef, int act,
                                            const float* residual, float* out, float* out_act, int act_out, int B, int Cin,
                                            int Cout, int H, int W, int pool2, float* stats, const ipdm_conv_ext_t* ext,
                                            void* stream) {
  IPDM_REQUIRE(stats != nullptr && out != nullptr && (coef || act == IPDM_ACT_NONE));
  return wino1d_entry(x, U, bias, coef, act, residual, out, out_act, act_out, B, Cin, Cout, H, W, pool2, stats, ext, stream);
}

// ---- 3x3x3 convolution of volumes on the same kernel (VOL instantiations) ----------------------------------------------------
namespace ipdm_conv {
bool wino1d_vol_ok(const ConvArgs& a) {
  if (!(a.D >= 1 && a.dil == 1 && a.Cin % (2 * Y_KC) == 0 && a.Cout % Y_CO == 0 && !a.coef && a.act == IPDM_ACT_NONE)) return false;
  if (a.out_act && a.act_out != IPDM_ACT_ELU && a.act_out != IPDM_ACT_COPY) return false;
  if (a.pool2 || a.stats) return false;
  if ((size_t)a.B * a.Cin * a.D * a.H * a.W * 4 >= 0x3fffffffull) return false;            // buffer descriptor / padding marker
  if ((size_t)a.D * a.H * a.W * Y_CO * 4 >= 0xffffffffull) return false;                   // 32-bit store offsets inside a channel block
  // rows of 16 .. any pixels (a 32-pixel block each) or planes of 12 pixels or less (two depth slices per block)
  return a.H >= 2 && a.W % 4 == 0 && (a.W <= 12 || a.W >= 16) && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
}

int conv_wino1d_vol_launch(ConvArgs a, hipStream_t s) {
  const bool pack = a.W <= 12;
  a.tiles_x = pack ? 1 : (a.W + 2 * Y_TX - 1) / (2 * Y_TX);
  a.tiles_y = (a.H + Y_ROWS - 1) / Y_ROWS;
  a.co_tiles = a.Cout / Y_CO;
  const int64_t nblk = (int64_t)a.B * (pack ? (a.D + 1) / 2 : a.D) * a.tiles_x * a.tiles_y * a.co_tiles;
  if (nblk > 0x7fffffff) return IPDM_EUNSUPPORTED;
  static int compact = -1;                       // IPDM_W1D_COMPACT=0: four column blocks everywhere (tuning / fallback)
  if (compact < 0) {
    const char* e = getenv("IPDM_W1D_COMPACT");
    compact = e ? atoi(e) : 1;
  }
  const bool nbu3 = compact && a.H % Y_ROWS == 0 && (pack ? a.W == 12 : a.W == 24);     // exactly 12 used pairs in each of 8 rows
  static bool attr_set = false;
  if (!attr_set) {
    const void* kernels[] = {reinterpret_cast<const void*>(conv_wino1d_kernel<1, false, false, false, true>),
                             reinterpret_cast<const void*>(conv_wino1d_kernel<2, false, false, false, true>),
                             reinterpret_cast<const void*>(conv_wino1d_kernel<3, false, false, false, true>),
                             reinterpret_cast<const void*>(conv_wino1d_kernel<1, false, false, false, true, 3>),
                             reinterpret_cast<const void*>(conv_wino1d_kernel<2, false, false, false, true, 3>),
                             reinterpret_cast<const void*>(conv_wino1d_kernel<3, false, false, false, true, 3>)};
    for (const void* k : kernels) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Y_LDS_BYTES);
      if (e != hipSuccess) return (int)e;
    }
    attr_set = true;
  }
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
    cus = 256;
  const int per_xcd = (int)((nblk + 7) / 8);
  const int S = per_xcd < cus / 8 ? per_xcd : cus / 8;
  const int outs = (a.out ? 1 : 0) | (a.out_act ? 2 : 0);
  if (nbu3) {
    if (outs == 3)
      hipLaunchKernelGGL((conv_wino1d_kernel<3, false, false, false, true, 3>), dim3((unsigned)(8 * S)), dim3(512), Y_LDS_BYTES, s, a, (int)nblk);
    else if (outs == 1)
      hipLaunchKernelGGL((conv_wino1d_kernel<1, false, false, false, true, 3>), dim3((unsigned)(8 * S)), dim3(512), Y_LDS_BYTES, s, a, (int)nblk);
    else
      hipLaunchKernelGGL((conv_wino1d_kernel<2, false, false, false, true, 3>), dim3((unsigned)(8 * S)), dim3(512), Y_LDS_BYTES, s, a, (int)nblk);
    return ipdm_launch_status();
  }
  if (outs == 3)
    hipLaunchKernelGGL((conv_wino1d_kernel<3, false, false, false, true>), dim3((unsigned)(8 * S)), dim3(512), Y_LDS_BYTES, s, a, (int)nblk);
  else if (outs == 1)
    hipLaunchKernelGGL((conv_wino1d_kernel<1, false, false, false, true>), dim3((unsigned)(8 * S)), dim3(512), Y_LDS_BYTES, s, a, (int)nblk);
  else
    hipLaunchKernelGGL((conv_wino1d_kernel<2, false, false, false, true>), dim3((unsigned)(8 * S)), dim3(512), Y_LDS_BYTES, s, a, (int)nblk);
  return ipdm_launch_status();
}
}  // namespace ipdm_conv

extern "C" int64_t ipdm_conv_wino1d_weight_bytes3d(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return -1;
  return (int64_t)36 * ((Cin + 15) / 16) * ((Cout + 31) / 32) * 2048 + (int64_t)((Cout + 31) / 32) * 32 * 4;
}

extern "C" int ipdm_conv_wino1d_pack_weight3d(const float* w, void* U, int Cout, int Cin, void* stream) {
  IPDM_REQUIRE(w && U && Cout > 0 && Cin > 0);
  return conv_wino1d_weights(w, U, Cout, Cin, 36, ipdm_stream(stream));
}

extern "C" int ipdm_conv3d_wino1d_supported(int Cin, int Cout, int D, int H, int W) {
  ConvArgs a;
  a.x = nullptr; a.out = a.out_act = nullptr; a.act_out = IPDM_ACT_NONE; a.coef = nullptr; a.act = IPDM_ACT_NONE; a.dil = 1;
  a.D = D; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.B = 1;
  return wino1d_vol_ok(a) ? 1 : 0;
}

extern "C" int ipdm_conv3d_wino1d_f32(const float* x, const void* U, const float* bias, const float* residual, float* out,
                                      float* out_act, int act_out, int B, int Cin, int Cout, int D, int H, int W,
                                      const ipdm_conv_ext_t* ext, void* stream) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && U && (out || out_act) && x != out && x != out_act);
  ConvArgs a;
  a.x = x; a.wt = (const float*)U; a.bias = bias; a.coef = nullptr; a.residual = residual; a.out = out; a.out_act = out_act;
  a.act_out = act_out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = 1; a.act = IPDM_ACT_NONE;
  a.D = D; a.kd = 3; a.tiles_x = a.tiles_y = a.co_tiles = 0; a.dbg = conv_debug_stamps();
  a.hx = 1;
  conv_apply_ext(a, ext, 1);
  if (!wino1d_vol_ok(a)) return IPDM_EUNSUPPORTED;
  return conv_wino1d_vol_launch(a, ipdm_stream(stream));
}
