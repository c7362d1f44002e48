// fp32 convolution on the gfx950 bf16 matrix cores by exact operand splitting ("bf16x3").
//
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate (157 TFLOP/s vs ~2.5 PFLOP/s), so the dense
// 3x3 / 1x1 contractions are computed here as
//     x = xh + xm + xl,   xh = bf16(x), xm = bf16(x - xh), xl = bf16(x - xh - xm)      (8 + 8 + 8 significand bits:
//     the three pieces carry the whole fp32 significand; both subtractions are exact in fp32)
//     w * x = wh*xh + (wh*xm + wm*xh) + (wh*xl + wm*xm + wl*xh) + (wm*xl + wl*xm + wl*xl)
// i.e. SIX v_mfma_f32_32x32x16_bf16 per 32x32x16 tile, accumulated in fp32 (every bf16 x bf16 product is exact in
// fp32).  The three dropped products are bounded by 2^-23 |w x| (|xm| <= 2^-8 |x|, |xl| <= 2^-16 |x|; typically
// 2^-25) -- the size of ONE fp32 rounding -- so the result is fp32-faithful: against a float64 convolution its error
// is at or below that of the exact-fp32 MFMA kernel (1.1-2.2e-6 vs 1.2-3.2e-6 of the output range on the network's
// layer shapes), and tests/test_kernels_gpu.py holds it to a tighter float64-referenced bound than the fp32 kernel.
// 6 x 32 cycles per 16-deep k-step against 8 x 64 for the fp32 instruction: 2.67x the matrix rate.
//
// GEMM view (as conv_kernel.h):  D[co, pixel] = sum_{tap, ci} W[tap][ci][co] * P[ci][pixel + tap offset]
//   A operand (32 x 16): rows = output channels, k = 16 input channels; lane (r, h) holds k = 8h .. 8h+7 of row r.
//        Weights are split and laid out ONCE by ipdm_conv_bx3_pack_weight so that a wave's fragment is one
//        contiguous KiB: [tap][ci/16][co/32][piece][h][r][8 x bf16].  Fragments go global -> VGPR directly
//        (global_load_dwordx4, prefetched one tap ahead); they never touch LDS.
//   B operand (16 x 32): k = the same 16 channels, columns = 32 pixels.  The input patch (+halo) of a 16-channel
//        chunk is loaded NCHW-coalesced (lane <-> pixel), activated/normalised if asked, split, and stored to LDS
//        channel-innermost: [piece][h][patch pixel][8 x bf16] -- so every tap's operand read is ONE conflict-free
//        ds_read_b128 per piece at a shifted pixel address.  Double-buffered; one barrier per chunk.
//   accumulator: lane <-> pixel, registers <-> output channels (coalesced NCHW stores), as the fp32 kernel.
// Two workgroups per CU (<= 256 registers, <= 80 KiB LDS): one wave's staging / waits run under the other's MFMAs.
#include "conv_kernel.h"

namespace ipdm_conv {

constexpr int BX3_KG = 8;       // K chunks (of 16 input channels) per accumulation group of the 16-pixel configurations

// ZT (3-D convolutions on small planes): depth slices per workgroup.  A workgroup's weight traffic is K x CO_T x 4 bytes whatever
// its pixel tile, so the kernel's L2 draw per output is inversely proportional to the pixels a workgroup owns; on 8 x 12 / 8 x 24
// slices one 128-pixel tile per workgroup asked an XCD's L2 for ~42 B/clk/CU of fragments (it delivers 27-30).  ZT = 2: the
// workgroup computes the SAME pixel tile of two consecutive depth slices -- every weight fragment feeds twice the MFMAs, the three
// depth taps of the pair read four input slices instead of six.
template <int NCT, int NPT, int WCO, int WPX, int PW, int DMAX, int KS, bool HX = false, int ZT = 1>
struct BxCfg {
  static constexpr int NPC = HX ? 2 : 3;                     // operand pieces (conv_kernel.h: f16x2 / three-way bf16)
  static constexpr int TAPS = KS * KS;
  static constexpr int CO_T = 32 * NCT * WCO;
  static constexpr int ROWS_PER_TILE = 32 / PW;
  static constexpr int PH = NPT * WPX * ROWS_PER_TILE;
  static constexpr int HALO = KS == 3 ? DMAX : 0;
  static constexpr int PHP = PH + 2 * HALO;
  static constexpr int PWP = PW + 2 * HALO;
  // PW == 16: a 32-lane half reads two image rows; a pitch of 0 (mod 16) pixels keeps their 16-byte pieces on
  // disjoint banks for ds_read_b128's lane groups
  static constexpr int PITCH = PW == 32 ? PWP : (PWP + 15) / 16 * 16;
  static constexpr int PLANE_S = PHP * PITCH;                // 16-byte units of one slice's patch
  static constexpr int PLANE = ZT * PLANE_S;                 // ... of the workgroup's slices, one behind the other
  static constexpr int STAGE = 2 * NPC * PLANE;              // [piece][h 2][PLANE]
  static constexpr int ITEMS = (2 * ZT * PHP * PWP + 255) / 256;  // (slice, pixel, 8-channel group) items per thread
  static constexpr int NPT_T = NPT * ZT;                     // pixel tiles per wave over all slices
  static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE * 16;
};

template <bool HX, int NCT, int NPT, int WCO, int WPX, int PW, int DMAX, int KS, bool FAST, int ZT = 1>
__global__ __launch_bounds__(256, 2) void conv_bx3_kernel(ConvArgs a) {
  using C = BxCfg<NCT, NPT, WCO, WPX, PW, DMAX, KS, HX, ZT>;
  constexpr int NPT_T = C::NPT_T;
  static_assert(ZT == 1 || (ZT == 2 && KS == 3), "two depth slices per workgroup: 3-D 3x3x3 layers");
  constexpr int NPC = C::NPC, FRAG = 64 * NPC;               // FRAG: 16-byte units per (tap, chunk, channel tile)
  static_assert(WCO * WPX == 4, "four waves per workgroup");
  extern __shared__ __align__(16) uint4 lds4[];

  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (a.dbg) t0 = __builtin_amdgcn_s_memtime();
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, slot = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int ks = bid % a.ksplit;                 // split-K part (fastest: the parts of a tile share its input in L2)
  bid /= a.ksplit;
  const int co_tile = bid % a.co_tiles;
  int t = bid / a.co_tiles;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int bz = t / a.tiles_y;
  const int zg = (a.D + ZT - 1) / ZT;            // depth groups per volume
  const int b = bz / zg, z = (bz - b * zg) * ZT; // first depth slice of this workgroup
  const int y0 = ty * C::PH, x0 = tx * PW;
  const int d = KS == 3 ? (DMAX == 1 ? 1 : a.dil) : 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int wco = wave / WPX, wpx = wave % WPX;
  const int HW = a.H * a.W;
  const size_t cs = (size_t)a.D * HW;

  const int n_cc = (a.Cin + 15) / 16;
  const int n_ct = (a.Cout + 31) / 32;
  // depth taps whose input slice exists (3-D only): a contiguous range around the centre tap
  int kz_lo = 0, n_kz = 1;
  if (a.kd == 3) {                               // (ZT slices: a tap is walked when ANY slice has its input; the others stage zeros)
    const bool lo = z + ZT - 1 - a.dil >= 0, hi = z + a.dil < a.D;
    kz_lo = lo ? 0 : 1;
    n_kz = 1 + (lo ? 1 : 0) + (hi ? 1 : 0);
  }
  const int n_chunks = n_cc * n_kz;
  // Split-K (16-pixel tile configurations only).  The K chunks are summed in GROUPS of BX3_KG: every group starts
  // from a zero accumulator and the group sums are added in order.  With ksplit == 1 the workgroup walks all groups
  // itself; with ksplit == number of groups each workgroup computes ONE group and the reduce pass adds them in the
  // same order -- bit-identical results either way, so the choice (which depends on how many tiles the batch
  // provides) never changes a sample's value.  An empty share (3-D border slices) writes zeros.
  int c_begin = 0, c_end = n_chunks;
  if (a.ksplit > 1) {
    c_begin = ks * BX3_KG < n_chunks ? ks * BX3_KG : n_chunks;
    c_end = c_begin + BX3_KG < n_chunks ? c_begin + BX3_KG : n_chunks;
  }

  // ---- B operand read offsets (16-byte units inside a stage) ----
  int b_base[NPT_T];
#pragma unroll
  for (int n = 0; n < NPT_T; ++n) {
    const int tile = wpx * NPT + n % NPT;
    const int prow = PW == 32 ? tile : tile * 2 + (j >> 4);
    const int pcol = PW == 32 ? j : (j & 15);
    b_base[n] = h * C::PLANE + (n / NPT) * C::PLANE_S + (prow + d) * C::PITCH + pcol + d;
  }

  // ---- staging geometry: item i = (patch pixel p, channel group g) ----
  const int pwv = PW + 2 * d, phv = C::PH + 2 * d;
  const int npos = phv * pwv;
  int it_lds[C::ITEMS], it_gofs[C::ITEMS], it_g[C::ITEMS];
  [[maybe_unused]] int it_s[C::ITEMS];           // ZT > 1: which of the workgroup's slices
  bool it_valid[C::ITEMS];
#pragma unroll
  for (int i = 0; i < C::ITEMS; ++i) {
    int idx = tid + i * 256;
    idx = idx < 2 * ZT * npos ? idx : 2 * ZT * npos - 1;
    const int sl = idx / (2 * npos);
    idx -= sl * 2 * npos;
    const int g = idx >= npos ? 1 : 0;
    const int p = idx - g * npos;
    const int r = p / pwv, c = p - r * pwv;
    const int gy = y0 - d + r, gx = x0 - d + c;
    it_g[i] = g;
    it_s[i] = sl;
    it_lds[i] = g * C::PLANE + sl * C::PLANE_S + r * C::PITCH + c;
    it_valid[i] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    it_gofs[i] = it_valid[i] ? gy * a.W + gx : 0;
  }

  // f16x2, dynamic range: this image's power-of-two input scale and its inverse (a.in_amax: per-image max |x|)
  [[maybe_unused]] float hx_in = 1.f, hx_out = 1.f;
  [[maybe_unused]] const bool hx_dyn = HX && a.in_amax != nullptr;
  if constexpr (HX) {
    if (hx_dyn) hx_dynamic_scale(ipdm_amax_read_v(a.in_amax + (size_t)b * IPDM_AMAX_SLOT), hx_in, hx_out);
  }
  float preg[C::ITEMS][8];
  auto chunk_kz_c0 = [&](int ch, int& kz, int& c0) {
    const int kzi = ch / n_cc;
    c0 = (ch - kzi * n_cc) * 16;
    kz = kz_lo + kzi;
  };
  // input slice of the workgroup's slice `sl` for depth tap kz, and whether it exists (clamped address when it does not)
  auto slice_in = [&](int kz, int sl, bool& ok) {
    const int zi = z + sl + (kz - a.kd / 2) * a.dil;
    ok = zi >= 0 && zi < a.D && z + sl < a.D;
    return ok ? zi : (z < a.D ? z : 0);
  };
  auto load_chunk = [&](int ch) {
    int kz, c0;
    chunk_kz_c0(ch, kz, c0);
    bool ok0, ok1 = false;
    const int zi0 = slice_in(kz, 0, ok0);
    const float* xb = a.x + (((size_t)b * a.Cin + c0) * a.D + zi0) * HW;
    [[maybe_unused]] const float* xb1 = xb;
    if constexpr (ZT > 1) xb1 = a.x + (((size_t)b * a.Cin + c0) * a.D + slice_in(kz, 1, ok1)) * HW;
#pragma unroll
    for (int i = 0; i < C::ITEMS; ++i) {
      const float* src = (ZT > 1 && it_s[i]) ? xb1 : xb;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int cl = it_g[i] * 8 + q;
        if constexpr (FAST) {
          preg[i][q] = src[(size_t)cl * cs + it_gofs[i]];
        } else {
          float v = 0.f;
          if (it_valid[i] && c0 + cl < a.Cin) v = src[(size_t)cl * cs + it_gofs[i]];
          preg[i][q] = v;
        }
      }
    }
  };
  auto store_chunk = [&](uint4* st, int ch) {
    int kz, c0;
    chunk_kz_c0(ch, kz, c0);
    bool zok0 = true, zok1 = true;                 // the slice's input for this depth tap exists (else: zeros)
    if constexpr (ZT > 1) {
      (void)slice_in(kz, 0, zok0);
      (void)slice_in(kz, 1, zok1);
    }
#pragma unroll
    for (int i = 0; i < C::ITEMS; ++i) {
      float v[8];
      const bool item_ok = it_valid[i] && (ZT == 1 || (it_s[i] ? zok1 : zok0));
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float val = preg[i][q];
        if constexpr (FAST) {
          val = item_ok ? val : 0.f;
        } else {
          const int ci = c0 + it_g[i] * 8 + q;
          if (item_ok && ci < a.Cin) {
            if (a.coef) {
              const float* cf = a.coef + ((size_t)b * a.Cin + ci) * 3;
              val = (val - cf[0]) * cf[1] + cf[2];
            }
            val = a.act == IPDM_ACT_ELU ? fast_elu(val) : ipdm_act(val, a.act);
          } else {
            val = 0.f;
          }
        }
        v[q] = val;
      }
      if constexpr (HX) {
        uint4 ph, pl;
        if (hx_dyn) split2_scaled(v, hx_in, ph, pl);           // (uniform branch: one image per workgroup)
        else split2(v, ph, pl);
        st[it_lds[i]] = ph;
        st[2 * C::PLANE + it_lds[i]] = pl;
      } else {
        bf16x8 ph, pm, pl;
        split3(v, ph, pm, pl);
        st[it_lds[i]] = __builtin_bit_cast(uint4, ph);
        st[2 * C::PLANE + it_lds[i]] = __builtin_bit_cast(uint4, pm);
        st[4 * C::PLANE + it_lds[i]] = __builtin_bit_cast(uint4, pl);
      }
    }
  };

  // ---- A fragments: global -> VGPR, [tapidx][cc][ct][piece][lane] in 16-byte units ----
  const uint4* wq = reinterpret_cast<const uint4*>(a.wt);
  const int ct0 = (co_tile * WCO + wco) * NCT;
  const size_t tap_stride = (size_t)n_cc * n_ct * FRAG;
  int ct_ofs[NCT];
#pragma unroll
  for (int m = 0; m < NCT; ++m) ct_ofs[m] = (ct0 + m < n_ct ? ct0 + m : n_ct - 1) * FRAG + lane;  // ragged: reload a valid one
  auto a_chunk_ptr = [&](int ch) {
    int kz, c0;
    chunk_kz_c0(ch, kz, c0);
    return wq + ((size_t)kz * C::TAPS * n_cc + (c0 >> 4)) * n_ct * FRAG;
  };
  auto load_A = [&](uint4 (&fr)[NCT][NPC], const uint4* base, int tap) {
    const uint4* p = base + tap * tap_stride;
#pragma unroll
    for (int m = 0; m < NCT; ++m)
#pragma unroll
      for (int s = 0; s < NPC; ++s) fr[m][s] = p[ct_ofs[m] + s * 64];
  };

  f32x16 acc[NCT][NPT_T];
#pragma unroll
  for (int m = 0; m < NCT; ++m)
#pragma unroll
    for (int n = 0; n < NPT_T; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // group sums added in order (16-pixel configurations, so that the split-K form gives the same bits; the two-slice form is
  // never split -- a rule of the layer shape, bx3_choose_ksplit -- and accumulates straight through)
  constexpr bool GROUPED = PW == 16 && ZT == 1;
  f32x16 tot[GROUPED ? NCT : 1][GROUPED ? NPT : 1];
  if constexpr (GROUPED) {
#pragma unroll
    for (int m = 0; m < NCT; ++m)
#pragma unroll
      for (int n = 0; n < NPT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[m][n][r] = 0.f;
  }
  uint4 afr[2][NCT][NPC];
  uint4 bfr[2][NPC];
  const int c_first = c_begin < n_chunks ? c_begin : n_chunks - 1;   // empty share: stage a valid chunk, use none
  load_A(afr[0], a_chunk_ptr(c_first), 0);
  load_chunk(c_first);
  store_chunk(lds4, c_first);
  __syncthreads();

  if (a.dbg) t1 = __builtin_amdgcn_s_memtime();
  constexpr int STEPS = C::TAPS * NPT_T;
  // operand reads of step s = (tap, pixel tile n): one ds_read_b128 per piece at the tap-shifted pixel
  auto load_B = [&](uint4 (&fr)[NPC], const uint4* cur, auto sc) {
    constexpr int st = decltype(sc)::value;
    constexpr int tap = st / NPT_T, n = st % NPT_T;
    constexpr int dy = KS == 3 ? tap / 3 - 1 : 0, dx = KS == 3 ? tap % 3 - 1 : 0;
    const uint4* bp = cur + b_base[n] + (dy * C::PITCH + dx) * d;
#pragma unroll
    for (int s = 0; s < NPC; ++s) fr[s] = bp[2 * s * C::PLANE];
  };

  for (int ch = c_begin; ch < c_end; ++ch) {
    const uint4* cur = lds4 + ((ch - c_begin) & 1) * C::STAGE;
    uint4* nxt = lds4 + ((ch - c_begin + 1) & 1) * C::STAGE;
    const bool more = ch + 1 < c_end;
    const uint4* a_cur = a_chunk_ptr(ch);
    const uint4* a_nxt = a_chunk_ptr(more ? ch + 1 : ch);
    if (more) load_chunk(ch + 1);
    load_B(bfr[0], cur, std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    // Software pipeline pinned with sched_group_barrier: the LDS reads of step s+1 (and, at the first step of a
    // tap, the global loads of the NEXT tap's A fragments) are issued before the six MFMAs of step s.
    static_for<STEPS>([&](auto sc) {
      constexpr int st = decltype(sc)::value;
      constexpr int tap = st / NPT_T, n = st % NPT_T;
      if constexpr (st + 1 < STEPS) load_B(bfr[(st + 1) & 1], cur, std::integral_constant<int, st + 1>{});
      if constexpr (n == 0) {
        if constexpr (tap + 1 < C::TAPS) load_A(afr[(tap + 1) & 1], a_cur, tap + 1);
        else load_A(afr[(tap + 1) & 1], a_nxt, 0);
      }
      if constexpr (HX) {
        const f16x8 bh = __builtin_bit_cast(f16x8, bfr[st & 1][0]), bl = __builtin_bit_cast(f16x8, bfr[st & 1][1]);
#pragma unroll
        for (int m = 0; m < NCT; ++m) {
          const f16x8 ah = __builtin_bit_cast(f16x8, afr[tap & 1][m][0]), al = __builtin_bit_cast(f16x8, afr[tap & 1][m][1]);
          f32x16 c = acc[m][n];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
          acc[m][n] = c;
        }
      } else {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bfr[st & 1][0]), bm = __builtin_bit_cast(bf16x8, bfr[st & 1][1]),
                     bl = __builtin_bit_cast(bf16x8, bfr[st & 1][NPC - 1]);
#pragma unroll
        for (int m = 0; m < NCT; ++m) {
          const bf16x8 ah = __builtin_bit_cast(bf16x8, afr[tap & 1][m][0]), am = __builtin_bit_cast(bf16x8, afr[tap & 1][m][1]),
                       al = __builtin_bit_cast(bf16x8, afr[tap & 1][m][NPC - 1]);
          f32x16 c = acc[m][n];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
          acc[m][n] = c;
        }
      }
      if constexpr (st + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, NPC, 0);        // DS reads first
      if constexpr (n == 0) __builtin_amdgcn_sched_group_barrier(0x020, NPC * NCT, 0);          // then the A loads
      __builtin_amdgcn_sched_group_barrier(0x008, (HX ? 3 : 6) * NCT, 0);                       // then the MFMAs
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr ((C::TAPS & 1) == 1) {            // the prefetched tap-0 fragments of the next chunk sit in slot 1
#pragma unroll
      for (int m = 0; m < NCT; ++m)
#pragma unroll
        for (int s = 0; s < NPC; ++s) afr[0][m][s] = afr[1][m][s];
    }
    if constexpr (GROUPED) {
      if (((ch + 1) % BX3_KG) == 0 || !more) {     // end of an accumulation group (uniform)
#pragma unroll
        for (int m = 0; m < NCT; ++m)
#pragma unroll
          for (int n = 0; n < NPT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              tot[m][n][r] += acc[m][n][r];
              acc[m][n][r] = 0.f;
            }
      }
    }
    if (more) store_chunk(nxt, ch + 1);
    __syncthreads();
  }

  if (a.dbg) t2 = __builtin_amdgcn_s_memtime();
  // ---- epilogue: bias, residual, coalesced stores (lane <-> pixel).  Loads are issued in batches ahead of their use:
  //      the bias of all of this thread's channels first, then the 16 residual values of a (pixel tile, channel tile)
  //      pair together, one pair ahead of the pair being stored.  (One element at a time -- load, wait, add, store --
  //      exposed every load's full latency, and behind run-time branches hipcc cannot count the stores in flight, so it
  //      also waited for the previous element's stores.)  The loads are unconditional (addresses clamped into the tensor,
  //      an absent bias / residual reads the weight blob instead) and each batch passes through an empty asm that
  //      "redefines" it, so the one counted wait sits there and no use further down waits again. ----
  const int co0 = co_tile * C::CO_T;
  const bool finish = a.ksplit <= 1;             // else: raw partial sum; bias / residual / activation in the reduce pass
  const bool has_res = finish && a.residual != nullptr, has_bias = finish && a.bias != nullptr;
  const float* const res_p = has_res ? a.residual : a.wt;
  const float* const bias_p = has_bias ? a.bias : a.wt;
  const size_t co_stride = has_res ? (size_t)a.D * HW : 0;
  // f16x2: the channels' inverse weight scales sit behind the fragments ([n_ct * 32] floats)
  const float* const scale_p = HX ? reinterpret_cast<const float*>(wq + (size_t)(a.kd == 3 ? 3 : 1) * C::TAPS * tap_stride) : a.wt;
  float bv[NCT][16];
  [[maybe_unused]] float sv[NCT][16];
#pragma unroll
  for (int m = 0; m < NCT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + (wco * NCT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      bv[m][r] = bias_p[has_bias ? b * a.bias_bstride + (co < a.Cout ? co : a.Cout - 1) : 0];
      if constexpr (HX) sv[m][r] = scale_p[co < n_ct * 32 ? co : 0] * hx_out;
    }
  constexpr int NBATCH = NPT_T * NCT;
  float rv[2][16];
  auto pixel_of = [&](int n, int& gy, int& gx) {   // n over the pixel tiles of all slices: slice n / NPT
    const int tile = wpx * NPT + n % NPT;
    gy = y0 + (PW == 32 ? tile : tile * 2 + (j >> 4));
    gx = x0 + (PW == 32 ? j : (j & 15));
  };
  auto zout_of = [&](int n) { return z + n / NPT; };
  auto load_res = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int n = q / NCT, m = q % NCT;
    int gy, gx;
    pixel_of(n, gy, gx);
    gy = gy < a.H ? gy : a.H - 1;
    gx = gx < a.W ? gx : a.W - 1;
    const int cob = co0 + (wco * NCT + m) * 32 + 4 * h;
    // channel cob + dco clamped to Cout - 1 (ragged last tile), as a 32-bit offset from channel cob's plane: the constant dco
    // times the plane stride is scalar, only the clamp value is per-thread (a per-element 64-bit product cost three quarter-rate
    // multiplies per residual value)
    const int cs32 = has_res ? (int)co_stride : 0;              // D * H * W < 2^26 (launcher)
    const int lim = a.Cout - 1 - cob, limoff = lim * cs32;
    const int zr = zout_of(n) < a.D ? zout_of(n) : a.D - 1;
    const float* const rp = res_p + (has_res ? (((size_t)b * a.Cout + cob) * a.D + zr) * HW + (size_t)gy * a.W + gx : 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dco = (r & 3) + 8 * (r >> 2);
      rv[q & 1][r] = rp[(ptrdiff_t)(dco <= lim ? dco * cs32 : limoff)];
    }
  };
#pragma unroll
  for (int m = 0; m < NCT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      asm volatile("" : "+v"(bv[m][r]));
      if constexpr (HX) asm volatile("" : "+v"(sv[m][r]));
    }
  load_res(std::integral_constant<int, 0>{});
  float amx_o = 0.f, amx_a = 0.f;                 // max |stored value| of this thread (a.amax_out / a.amax_act; one image per workgroup)
  static_for<NBATCH>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int n = q / NCT, m = q % NCT;
    if constexpr (q + 1 < NBATCH) load_res(std::integral_constant<int, q + 1>{});
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(rv[q & 1][r]));
    int gy, gx;
    pixel_of(n, gy, gx);
    if (gy < a.H && gx < a.W && zout_of(n) < a.D) {
      const int cob = co0 + (wco * NCT + m) * 32 + 4 * h;      // channel of r = 0; r adds (r & 3) + 8 * (r >> 2)
      size_t ob = (((size_t)b * a.Cout + cob) * a.D + zout_of(n)) * HW + (size_t)gy * a.W + gx;
      // (opaque: hipcc otherwise folds the channel step back into the product chain and recomputes the whole 64-bit index --
      //  six quarter-rate multiplies -- for every one of the 16 values; the step itself is a scalar, D * H * W < 2^26: launcher)
      asm volatile("" : "+v"(ob));
      const unsigned os32 = (unsigned)(a.D * HW);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dco = (r & 3) + 8 * (r >> 2);
        if (cob + dco < a.Cout) {
          const size_t o = ob + (size_t)((unsigned)dco * os32);
          float v;
          if constexpr (GROUPED) v = tot[m][n][r];
          else v = acc[m][n][r];
          if constexpr (HX) v *= sv[m][r];
          if (!finish) {
            a.partial[(size_t)ks * a.B * a.Cout * cs + o] = v;
            continue;
          }
          if (has_bias) v += bv[m][r];
          float vr = v;                                       // what `out` stores: without the residual when it is the second
          if (has_res) v += rv[q & 1][r];                     // output's alone (a.res_second)
          vr = a.res_second ? vr : v;
          v *= a.out_scale;
          vr *= a.out_scale;
          amx_o = fmaxf(amx_o, fabsf(vr));
          if (a.out) a.out[o] = vr;
          if (a.out_act) {
            const float e = a.act_out == IPDM_ACT_ELU ? fast_elu(v) : ipdm_act(v, a.act_out);
            amx_a = fmaxf(amx_a, fabsf(e));
            a.out_act[o] = e;
          }
        }
      }
    }
  });
  if (finish) {
    const int way = (int)(blockIdx.x & 1) * 4 + wave;          // (four waves per workgroup: neighbours take the other four ways)
    if (a.amax_out) ipdm_amax_commit(amx_o, a.amax_out + (size_t)b * IPDM_AMAX_SLOT, way);
    if (a.amax_act) ipdm_amax_commit(amx_a, a.amax_act + (size_t)b * IPDM_AMAX_SLOT, way);
  }
  if (a.dbg) {                                   // tuning aid (ipdm_debug_set_stamp_buffer); NULL in production
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* d4 = a.dbg + (size_t)blockIdx.x * 4;
      d4[0] = t0; d4[1] = t1; d4[2] = t2; d4[3] = t3;
    }
  }
}

template <bool HX, int NCT, int NPT, int WCO, int WPX, int PW, int DMAX, int KS, bool FAST, int ZT = 1>
int launch_bx3(ConvArgs a, hipStream_t s) {
  using C = BxCfg<NCT, NPT, WCO, WPX, PW, DMAX, KS, HX, ZT>;
  a.tiles_x = (a.W + PW - 1) / PW;
  a.tiles_y = (a.H + C::PH - 1) / C::PH;
  a.co_tiles = (a.Cout + C::CO_T - 1) / C::CO_T;
  if (ZT > 1 && a.ksplit > 1) return IPDM_EUNSUPPORTED;
  const int64_t nblk = (int64_t)a.B * ((a.D + ZT - 1) / ZT) * a.tiles_x * a.tiles_y * a.co_tiles * a.ksplit;
  if (nblk > 0x7fffffff || (int64_t)a.D * a.H * a.W >= (1 << 26)) return IPDM_EUNSUPPORTED;   // (32-bit plane offsets in the epilogue)
  auto kern = conv_bx3_kernel<HX, NCT, NPT, WCO, WPX, PW, DMAX, KS, FAST, ZT>;
  static bool attr_set = false;
  if (!attr_set) {
    if (C::LDS_BYTES > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::LDS_BYTES);
      if (e != hipSuccess) return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), C::LDS_BYTES, s, a);
  return ipdm_launch_status();
}

template <bool HX, int NCT, int NPT, int WCO, int WPX, int PW, int DMAX>
int launch_bx3_cfg_t(const ConvArgs& a, int ks, hipStream_t s) {
  const bool fast = a.Cin % 16 == 0 && a.act == IPDM_ACT_NONE && !a.coef;
  if (ks == 3) {
    return fast ? launch_bx3<HX, NCT, NPT, WCO, WPX, PW, DMAX, 3, true>(a, s)
                : launch_bx3<HX, NCT, NPT, WCO, WPX, PW, DMAX, 3, false>(a, s);
  }
  return fast ? launch_bx3<HX, NCT, NPT, WCO, WPX, PW, 1, 1, true>(a, s) : launch_bx3<HX, NCT, NPT, WCO, WPX, PW, 1, 1, false>(a, s);
}
template <int NCT, int NPT, int WCO, int WPX, int PW, int DMAX>
int launch_bx3_cfg(const ConvArgs& a, int ks, hipStream_t s) {
  return a.hx ? launch_bx3_cfg_t<true, NCT, NPT, WCO, WPX, PW, DMAX>(a, ks, s)
              : launch_bx3_cfg_t<false, NCT, NPT, WCO, WPX, PW, DMAX>(a, ks, s);
}

static int bx3_forced_cfg() {
  static int v = -2;
  if (v == -2) {
    const char* e = getenv("IPDM_BX3_CFG");
    v = e ? atoi(e) : -1;
  }
  return v;
}

// two depth slices per workgroup (BxCfg): undilated 3x3x3 layers on slices at most 16 pixels wide with >= 128 output channels --
// a rule of the LAYER SHAPE only (the form accumulates straight through, so it must never alternate with the grouped one)
bool bx3_two_slices(int D, int Cin, int Cout, int W, int k, int dil, int kd) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("IPDM_BX3_ZT2");
    on = e ? atoi(e) : 1;
  }
  return on && kd == 3 && k == 3 && dil == 1 && D >= 2 && W <= 16 && Cout >= 128 && Cin % 16 == 0;
}

template <bool HX>
static int launch_bx3_zt2(const ConvArgs& a, hipStream_t s) {
  const bool fast = a.Cin % 16 == 0 && a.act == IPDM_ACT_NONE && !a.coef;
  // (two channel tiles x two pixel tiles per wave and slice -- half the LDS operand reads per MFMA -- was measured too: 80 spilled
  //  registers, 4.64 ms against 3.90 per 256 -> 256 launch of config 4; one channel tile per wave it is)
  return fast ? launch_bx3<HX, 1, 4, 4, 1, 16, 1, 3, true, 2>(a, s) : launch_bx3<HX, 1, 4, 4, 1, 16, 1, 3, false, 2>(a, s);
}

int conv_bx3_dispatch(const ConvArgs& a, int ks, hipStream_t s) {
  const int f = bx3_forced_cfg();
  if (ks == 3 && a.ksplit <= 1 && f < 0 && bx3_two_slices(a.D, a.Cin, a.Cout, a.W, ks, a.dil, a.kd))
    return a.hx ? launch_bx3_zt2<true>(a, s) : launch_bx3_zt2<false>(a, s);
  if (a.W <= 16) {
    // 128 output channels per workgroup (a wave: 32 channels x all four 32-pixel tiles) where that still leaves two workgroups
    // per CU: a weight fragment then feeds 12 MFMAs instead of 6 -- the 64-channel form draws ~42 B/clk/CU of fragments from L2,
    // more than a CU gets (config 4's 3-D layers: 0.722 -> 0.677 s per level).  Same K order per output: same bits.
    const int64_t wgs128 = (int64_t)a.B * a.D * ((a.H + 7) / 8) * ((a.W + 15) / 16) * ((a.Cout + 127) / 128) * a.ksplit;
    const bool wide_co = f == 3 || (f != 4 && a.Cout >= 128 && wgs128 >= 512);
    if (a.dil > 1 && ks == 3)
      return wide_co ? launch_bx3_cfg<1, 4, 4, 1, 16, 4>(a, ks, s) : launch_bx3_cfg<1, 2, 2, 2, 16, 4>(a, ks, s);
    return wide_co ? launch_bx3_cfg<1, 4, 4, 1, 16, 1>(a, ks, s)                     // 128 co x (8 x 16) px
                   : launch_bx3_cfg<1, 2, 2, 2, 16, 1>(a, ks, s);                    //  64 co x (8 x 16) px
  }
  if (a.dil > 1 && ks == 3) return launch_bx3_cfg<1, 4, 2, 2, 32, 4>(a, ks, s);
  if (f == 1) return launch_bx3_cfg<1, 2, 1, 4, 32, 1>(a, ks, s);                    // 32 co x 256 px
  if (f == 2) return launch_bx3_cfg<2, 2, 2, 2, 32, 1>(a, ks, s);                    // 128 co x 128 px
  if (a.Cout <= 32) return launch_bx3_cfg<1, 2, 1, 4, 32, 1>(a, ks, s);
  return launch_bx3_cfg<1, 4, 2, 2, 32, 1>(a, ks, s);                                // 64 co x (8 x 32) px
}

// split-K second pass: out = bias + sum_s partial[s] (fixed order: deterministic) + residual; out_act = act(out)
__global__ __launch_bounds__(256) void bx3_splitk_reduce_kernel(const float* __restrict__ partial, int ksplit,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ residual, float* out,
                                                                float* out_act, int act_out, int Cout, int64_t plane,
                                                                int64_t total, int bias_bstride, float out_scale,
                                                                float* amax_out, float* amax_act, int res_second) {
  // grid = (blocks per image, images) when the maxima are wanted (a workgroup then stays inside one image), else 1-D grid-stride
  const int64_t per_image = plane * Cout;
  const bool by_image = gridDim.y > 1 || amax_out || amax_act;      // (launchers: B <= 65535 whenever the maxima are wanted)
  const int64_t lo = by_image ? (int64_t)blockIdx.y * per_image : 0, hi = by_image ? lo + per_image : total;
  float amx_o = 0.f, amx_a = 0.f;
  for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256) {
    float v = partial[i];
    for (int s = 1; s < ksplit; ++s) v += partial[(size_t)s * total + i];
    if (bias) v += bias[(i / per_image) * bias_bstride + (i / plane) % Cout];
    float vr = v;
    if (residual) v += residual[i];
    vr = res_second ? vr : v;
    v *= out_scale;
    vr *= out_scale;
    amx_o = fmaxf(amx_o, fabsf(vr));
    if (out) out[i] = vr;
    if (out_act) {
      const float e = act_out == IPDM_ACT_ELU ? fast_elu(v) : ipdm_act(v, act_out);
      amx_a = fmaxf(amx_a, fabsf(e));
      out_act[i] = e;
    }
  }
  __shared__ float red[2][4];
  if (amax_out) ipdm_amax_commit_block(amx_o, amax_out + (size_t)blockIdx.y * IPDM_AMAX_SLOT, (int)blockIdx.x, red[0]);
  if (amax_act) ipdm_amax_commit_block(amx_a, amax_act + (size_t)blockIdx.y * IPDM_AMAX_SLOT, (int)blockIdx.x, red[1]);
}

// Split-K pays only when the tiles alone leave most of the chip idle (small images at small batch).  It is offered
// for the 16-pixel tile configurations, whose accumulation is grouped (BX3_KG chunks): the answer is either 1 or the
// number of groups, and both give bit-identical results.
int bx3_choose_ksplit(int B, int D, int Cin, int Cout, int H, int W, int k, int dil) {
  static int forced = -2;                        // IPDM_BX3_KSPLIT=0: never split, =1: split whenever allowed
  if (forced == -2) {
    const char* e = getenv("IPDM_BX3_KSPLIT");
    forced = e ? atoi(e) : -1;
  }
  if (W > 16 || forced == 0) return 1;
  if (bx3_two_slices(D, Cin, Cout, W, k, dil, D > 1 && k == 3 ? 3 : 1)) return 1;      // (that form is never split)
  const int64_t tiles = (int64_t)B * D * ((H + 7) / 8) * ((Cout + 63) / 64);
  const int n_chunks = ((Cin + 15) / 16) * (D > 1 && k == 3 ? 3 : 1);
  const int n_groups = (n_chunks + BX3_KG - 1) / BX3_KG;
  if (n_groups < 2) return 1;
  if (forced == 1) return n_groups;
  return tiles <= 320 ? n_groups : 1;            // measured at B = 28 (224 tiles of 256 -> 256 @16^2): -0.6 ms per iteration
}

__global__ __launch_bounds__(256) void bx3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                       int Cout, int Cin, int kk, int n_cc, int n_ct) {
  const int64_t total = (int64_t)kk * n_cc * n_ct * 512;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i & 7), r = (int)((i >> 3) & 31), h = (int)((i >> 8) & 1);
    const int64_t rest = i >> 9;
    const int ct = (int)(rest % n_ct);
    const int cc = (int)((rest / n_ct) % n_cc);
    const int tap = (int)(rest / ((int64_t)n_ct * n_cc));
    const int co = ct * 32 + r, ci = cc * 16 + 8 * h + q;
    const float v = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * kk + tap] : 0.f;
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

// ---- f16x2 weights (conv_kernel.h): per-output-channel power-of-two scale, two fp16 pieces
//      [tap][ci/16][co/32][piece 2][h][r][8], followed by the inverse scales [n_ct * 32] (fp32) ----
__global__ __launch_bounds__(256) void hx2_scale_kernel(const float* __restrict__ w, float* __restrict__ inv_scale, int Cout,
                                                        int64_t per_co) {
  __shared__ float red[256];
  const int co = blockIdx.x;
  float m = 0.f;
  if (co < Cout)
    for (int64_t i = threadIdx.x; i < per_co; i += 256) m = fmaxf(m, fabsf(w[(int64_t)co * per_co + i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int e = 0;
    const float mx = red[0];
    const bool ok = mx > 0.f && mx < INFINITY;
    if (ok) (void)frexpf(mx, &e);                            // mx = f * 2^e, f in [0.5, 1): mx * 2^(14 - e) in [2^13, 2^14)
    inv_scale[co] = ok ? ldexpf(1.f, e - 14) : 1.f;
  }
}

__global__ __launch_bounds__(256) void hx2_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                       const float* __restrict__ inv_scale, int Cout, int Cin, int kk, int n_cc,
                                                       int n_ct) {
  const int64_t total = (int64_t)kk * n_cc * n_ct * 512;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i & 7), r = (int)((i >> 3) & 31), h = (int)((i >> 8) & 1);
    const int64_t rest = i >> 9;
    const int ct = (int)(rest % n_ct);
    const int cc = (int)((rest / n_ct) % n_cc);
    const int tap = (int)(rest / ((int64_t)n_ct * n_cc));
    const int co = ct * 32 + r, ci = cc * 16 + 8 * h + q;
    const float v = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * kk + tap] * (1.f / inv_scale[co]) : 0.f;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const int64_t base = rest * 2 * 512 + h * 256 + r * 8 + q;
    out[base] = __builtin_bit_cast(unsigned short, hi);
    out[base + 512] = __builtin_bit_cast(unsigned short, lo);
  }
}

}  // namespace ipdm_conv

using namespace ipdm_conv;

extern "C" int64_t ipdm_conv_hx2_weight_bytes(int Cout, int Cin, int k) {
  if (Cout <= 0 || Cin <= 0 || !(k == 1 || k == 3 || k == 27)) return -1;
  const int64_t kk = k == 27 ? 27 : k * k;
  return kk * ((Cin + 15) / 16) * ((Cout + 31) / 32) * 2048 + (int64_t)((Cout + 31) / 32) * 32 * 4;
}

extern "C" int ipdm_conv_hx2_pack_weight(const float* w, void* packed, int Cout, int Cin, int k, void* stream) {
  IPDM_REQUIRE(w && packed && Cout > 0 && Cin > 0 && (k == 1 || k == 3 || k == 27));
  const int kk = k == 27 ? 27 : k * k;
  const int n_cc = (Cin + 15) / 16, n_ct = (Cout + 31) / 32;
  const int64_t total = (int64_t)kk * n_cc * n_ct * 512;
  float* inv_scale = reinterpret_cast<float*>(static_cast<char*>(packed) + total * 4);
  hipLaunchKernelGGL(hx2_scale_kernel, dim3(n_ct * 32), dim3(256), 0, ipdm_stream(stream), w, inv_scale, Cout, (long long)Cin * kk);
  hipLaunchKernelGGL(hx2_pack_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, ipdm_stream(stream), w,
                     (unsigned short*)packed, inv_scale, Cout, Cin, kk, n_cc, n_ct);
  return ipdm_launch_status();
}

extern "C" int64_t ipdm_conv_bx3_weight_bytes(int Cout, int Cin, int k) {
  if (Cout <= 0 || Cin <= 0 || !(k == 1 || k == 3 || k == 27)) return -1;
  const int64_t kk = k == 27 ? 27 : k * k;
  return kk * ((Cin + 15) / 16) * ((Cout + 31) / 32) * 3072;
}

extern "C" int ipdm_conv_bx3_pack_weight(const float* w, void* packed, int Cout, int Cin, int k, void* stream) {
  IPDM_REQUIRE(w && packed && Cout > 0 && Cin > 0 && (k == 1 || k == 3 || k == 27));
  const int kk = k == 27 ? 27 : k * k;
  const int n_cc = (Cin + 15) / 16, n_ct = (Cout + 31) / 32;
  const int64_t total = (int64_t)kk * n_cc * n_ct * 512;
  hipLaunchKernelGGL(bx3_pack_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, ipdm_stream(stream), w,
                     (unsigned short*)packed, Cout, Cin, kk, n_cc, n_ct);
  return ipdm_launch_status();
}

static int conv3d_bx3_entry(const float* x, const void* wq, const float* bias, const float* coef, int act,
                            const float* residual, float* out, float* out_act, int act_out, int B, int Cin, int Cout, int D,
                            int H, int W, int k, int dilation, void* stream, int hx, const ipdm_conv_ext_t* ext = nullptr) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0 && (k == 1 || k == 3) && dilation >= 1);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && wq && (out || out_act) && x != out && x != out_act);
  if (k == 3 && dilation > 4) return IPDM_EUNSUPPORTED;
  ConvArgs a;
  a.x = x; a.wt = (const float*)wq; a.bias = bias; a.coef = coef; a.residual = residual; a.out = out; a.out_act = out_act;
  a.act_out = act_out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = k == 3 ? dilation : 1; a.act = act;
  a.D = D; a.kd = k;
  a.tiles_x = a.tiles_y = a.co_tiles = 0; a.dbg = conv_debug_stamps();
  a.hx = hx;
  conv_apply_ext(a, ext, hx);
  return conv_bx3_dispatch(a, k, ipdm_stream(stream));
}

extern "C" int ipdm_conv3d_bx3_f32(const float* x, const void* wq, const float* bias, const float* coef, int act,
                                   const float* residual, float* out, float* out_act, int act_out, int B, int Cin,
                                   int Cout, int D, int H, int W, int k, int dilation, const ipdm_conv_ext_t* ext, void* stream) {
  return conv3d_bx3_entry(x, wq, bias, coef, act, residual, out, out_act, act_out, B, Cin, Cout, D, H, W, k, dilation, stream, 0, ext);
}

/* f16x2 forms (conv_kernel.h): same arguments, wq from ipdm_conv_hx2_pack_weight */
extern "C" int ipdm_conv3d_hx2_f32(const float* x, const void* wq, const float* bias, const float* coef, int act,
                                   const float* residual, float* out, float* out_act, int act_out, int B, int Cin,
                                   int Cout, int D, int H, int W, int k, int dilation, const ipdm_conv_ext_t* ext, void* stream) {
  return conv3d_bx3_entry(x, wq, bias, coef, act, residual, out, out_act, act_out, B, Cin, Cout, D, H, W, k, dilation, stream, 1,
                          ext);
}

extern "C" int ipdm_conv_bx3_splitk(int B, int D, int Cin, int Cout, int H, int W, int k, int dilation) {
  return bx3_choose_ksplit(B, D, Cin, Cout, H, W, k, dilation);
}

// split-K form: `work` holds ksplit * B * Cout * D * H * W floats (ksplit from ipdm_conv_bx3_splitk, > 1);
// volume = 0: 2-D convolution (D must be 1), volume = 1: 3-D convolution with 27- / 1-tap weights
static int conv_bx3_splitk_entry(const float* x, const void* wq, const float* bias, const float* coef, int act,
                                 const float* residual, float* out, float* out_act, int act_out, int B, int Cin, int Cout,
                                 int D, int H, int W, int k, int dilation, int volume, int ksplit, float* work, void* stream,
                                 int hx, const ipdm_conv_ext_t* ext = nullptr) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0 && (k == 1 || k == 3) && dilation >= 1);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && wq && work && ksplit > 1 && (out || out_act) && x != out && x != out_act);
  IPDM_REQUIRE(B <= 65535 || !ext || (!ext->out_amax && !ext->act_amax));
  if (k == 3 && dilation > 4) return IPDM_EUNSUPPORTED;
  ConvArgs a;
  a.x = x; a.wt = (const float*)wq; a.bias = nullptr; a.coef = coef; a.residual = nullptr; a.out = nullptr; a.out_act = nullptr;
  a.act_out = IPDM_ACT_NONE; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = k == 3 ? dilation : 1; a.act = act;
  a.D = D; a.kd = volume ? k : 1;             // volume: the weights carry depth taps (ipdm_conv3d_bx3_f32 semantics)
  a.tiles_x = a.tiles_y = a.co_tiles = 0; a.dbg = nullptr;
  a.ksplit = ksplit; a.partial = work;
  a.hx = hx;
  conv_apply_ext(a, ext, hx);
  const float out_scale = a.out_scale;
  a.out_scale = 1.f;                            // the parts are raw partial sums; the reduce pass applies bias / residual / scale
  float* const amax_out = a.amax_out;
  float* const amax_act = a.amax_act;
  a.amax_out = a.amax_act = nullptr;            //   ... and takes the maxima
  int rc = conv_bx3_dispatch(a, k, ipdm_stream(stream));
  a.out_scale = out_scale;
  if (rc != IPDM_OK) return rc;
  const int64_t plane = (int64_t)D * H * W, total = (int64_t)B * Cout * plane;
  hipLaunchKernelGGL(bx3_splitk_reduce_kernel, splitk_reduce_grid(B, Cout * plane, amax_out || amax_act), dim3(256), 0,
                     ipdm_stream(stream), work, ksplit, bias, residual, out, out_act, act_out, Cout, (long long)plane,
                     (long long)total, a.bias_bstride, a.out_scale, amax_out, amax_act, a.res_second);
  return ipdm_launch_status();
}

extern "C" int ipdm_conv_bx3_splitk_f32(const float* x, const void* wq, const float* bias, const float* coef, int act,
                                        const float* residual, float* out, float* out_act, int act_out, int B, int Cin,
                                        int Cout, int D, int H, int W, int k, int dilation, int volume, int ksplit,
                                        float* work, const ipdm_conv_ext_t* ext, void* stream) {
  return conv_bx3_splitk_entry(x, wq, bias, coef, act, residual, out, out_act, act_out, B, Cin, Cout, D, H, W, k, dilation, volume,
                               ksplit, work, stream, 0, ext);
}

extern "C" int ipdm_conv_hx2_splitk_f32(const float* x, const void* wq, const float* bias, const float* coef, int act,
                                        const float* residual, float* out, float* out_act, int act_out, int B, int Cin,
                                        int Cout, int D, int H, int W, int k, int dilation, int volume, int ksplit,
                                        float* work, const ipdm_conv_ext_t* ext, void* stream) {
  return conv_bx3_splitk_entry(x, wq, bias, coef, act, residual, out, out_act, act_out, B, Cin, Cout, D, H, W, k, dilation, volume,
                               ksplit, work, stream, 1, ext);
}

static int conv2d_bx3_entry(const float* x, const void* wq, const float* bias, const float* coef, int act,
                            const float* residual, float* out, float* out_act, int act_out, int B, int Cin, int Cout, int H,
                            int W, int k, int dilation, void* stream, int hx, const ipdm_conv_ext_t* ext = nullptr) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (k == 1 || k == 3) && dilation >= 1);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && wq && (out || out_act) && x != out && x != out_act);
  if (k == 3 && dilation > 4) return IPDM_EUNSUPPORTED;
  ConvArgs a;
  a.x = x; a.wt = (const float*)wq; a.bias = bias; a.coef = coef; a.residual = residual; a.out = out; a.out_act = out_act;
  a.act_out = act_out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.dil = k == 3 ? dilation : 1; a.act = act;
  a.D = 1; a.kd = 1;
  a.tiles_x = a.tiles_y = a.co_tiles = 0; a.dbg = conv_debug_stamps();
  a.hx = hx;
  conv_apply_ext(a, ext, hx);
  return conv_bx3_dispatch(a, k, ipdm_stream(stream));
}

extern "C" int ipdm_conv2d_bx3_f32(const float* x, const void* wq, const float* bias, const float* coef, int act,
                                   const float* residual, float* out, float* out_act, int act_out, int B, int Cin,
                                   int Cout, int H, int W, int k, int dilation, const ipdm_conv_ext_t* ext, void* stream) {
  return conv2d_bx3_entry(x, wq, bias, coef, act, residual, out, out_act, act_out, B, Cin, Cout, H, W, k, dilation, stream, 0, ext);
}

extern "C" int ipdm_conv2d_hx2_f32(const float* x, const void* wq, const float* bias, const float* coef, int act,
                                   const float* residual, float* out, float* out_act, int act_out, int B, int Cin,
                                   int Cout, int H, int W, int k, int dilation, const ipdm_conv_ext_t* ext, void* stream) {
  return conv2d_bx3_entry(x, wq, bias, coef, act, residual, out, out_act, act_out, B, Cin, Cout, H, W, k, dilation, stream, 1,
                          ext);
}
