// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores for the undilated 3x3 layers with wide images.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile, 4x4 input tile d, 3x3 filter g
// turns the 9-tap convolution into 16 independent channel contractions  M_p[co, tile] = sum_ci U_p[co,ci] V_p[ci,tile]
// (p = position in the 4x4 transformed tile): 16 multiply-adds per 4 outputs instead of 36 -> 2.25x fewer MFMA
// cycles, still exact-fp32 fma chains.  U = G g G^T is precomputed once per model ([16][Cin][Cout]).
//
// Workgroup = 64 output channels x 64 tiles (4 tile rows x 16 tiles = 8 x 32 output pixels) of one image;
// 8 waves = 2 (position halves) x 2 (channel halves) x 2 (tile halves); a wave keeps 8 of the 16 positions of its
// 32 x 32 (co x tile) block in 128 accumulator registers; the output transform A^T M A is per-lane register
// arithmetic on each half plus one LDS hand-off, and the lane <-> tile map gives float2 (8-byte) coalesced stores.
// Per chunk of 8 input channels: U chunk (32 KiB) global -> regs -> LDS; the 4x4 input patches go global -> regs,
// are transformed (B^T d B, 32 adds) in the shadow of the MFMAs and land as V in LDS; two LDS stages, one barrier
// per chunk, operand reads software-pipelined one MFMA group ahead (same scheme as conv_kernel.h).
//
// Used by ipdm_conv2d_f32 when: 3x3, dilation 1, Cin % 8 == 0, Cout % 64 == 0, H and W even, W >= 32, no fused
// input normalisation/activation (the network hands activated tensors to its convolutions).
#include "conv_kernel.h"

namespace ipdm_conv {

namespace {

constexpr int W_KC = 8;                 // input channels per chunk
constexpr int W_CO = 64;                // output channels per workgroup
constexpr int W_TX = 16, W_TY = 4;      // tiles per workgroup (x, y)
constexpr int W_TILES = W_TX * W_TY;    // 64
constexpr int W_U_ELEMS = 16 * W_KC * W_CO;       // 8192 floats
constexpr int W_V_ELEMS = 16 * W_KC * W_TILES;    // 8192 floats
constexpr int W_STAGE = W_U_ELEMS + W_V_ELEMS;    // 64 KiB per stage
constexpr size_t W_LDS_BYTES = 2 * (size_t)W_STAGE * sizeof(float);

// U[p = a*4+b][ci][co] = sum_ij G[a][i] g[co][ci][i][j] G[b][j]
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout,
                                                          int Cin) {
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const int64_t total = (int64_t)Cout * Cin;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int co = (int)(idx % Cout), ci = (int)(idx / Cout);
    const float* g = w + ((size_t)co * Cin + ci) * 9;
    float t[4][3];
    for (int a = 0; a < 4; ++a)
      for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[j] + G[a][1] * g[3 + j] + G[a][2] * g[6 + j];
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b)
        U[((size_t)(a * 4 + b) * Cin + ci) * Cout + co] = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
  }
}

// 512 threads = 8 waves = 2 (position halves) x 2 (channel halves) x 2 (tile halves): two waves per SIMD, so one
// wave's load / LDS latencies are covered by its partner's MFMAs.  Wave `ph` owns positions 8*ph .. 8*ph+7 (rows
// 2*ph, 2*ph+1 of the 4x4 transformed tile) = 128 accumulator registers; the output transform is linear, so each
// wave reduces its half to a partial 2x2 result and the halves meet once through LDS in the epilogue.
template <bool SMALL>
__global__ __launch_bounds__(512) void conv_wino_kernel(ConvArgs a) {
  extern __shared__ __align__(16) float lds[];
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
  const int co0 = co_tile * W_CO;
  const int y0 = ty * (2 * W_TY), x0 = tx * (2 * W_TX);     // output-pixel origin of the workgroup (!SMALL)
  // SMALL: the workgroup owns 64 consecutive tiles of the image's linear tile space.  A dilation-d convolution
  // is d*d independent undilated convolutions on the d-subsampled images (polyphase), so tile number
  //   t = ((sy*d + sx)*THS + tyy)*TWS + txx   covers output pixels (d*(2*tyy+i) + sy, d*(2*txx+j) + sx)
  // and reads input pixels d apart; a 16x16 image is exactly 64 tiles for d = 1, 2 and 4.
  const int d = a.dil;
  const int TWS = SMALL ? a.W / (2 * d) : 1, THS = SMALL ? a.H / (2 * d) : 1;   // (wide form: unused)
  const int tile0 = SMALL ? tx * W_TILES : 0;               // tiles_x counts 64-tile groups, tiles_y == 1
  auto tile_origin = [&](int tl, int& py, int& px) {        // top-left OUTPUT pixel of tile tl (SMALL)
    const int txx = tl % TWS;
    int r = tl / TWS;
    const int tyy = r % THS;
    r /= THS;
    py = d * (2 * tyy) + r / d;
    px = d * (2 * txx) + r % d;
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int ph = wave >> 2, wco = (wave >> 1) & 1, wtl = wave & 1;
  const int HW = a.H * a.W;

  // ---- MFMA operand read offsets inside a stage (position offset added per group) ----
  const int a_base = (ph * 8 * W_KC + h) * W_CO + wco * 32 + j;                    // Us[p][kc][co]
  const int b_base = W_U_ELEMS + (ph * 8 * W_KC + h) * W_TILES + wtl * 32 + j;     // Vs[p][kc][tile]

  // ---- staging geometry: this thread transforms tile `mytile` of channel kc = tid / 64 ----
  const int mytile = tid & 63;
  const int kc_a = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps the channel base pointer in SGPRs
  const int tyl = mytile / W_TX, txl = mytile % W_TX;
  // 4x4 patch addresses = (clamped row offset) + (clamped column offset); padding is zero-selected after the load
  int row_off[4], col_off[4];
  unsigned rv = 0, cv = 0;
  int my_py = 0, my_px = 0;
  bool my_tile_ok = true;
  if constexpr (SMALL) {
    my_tile_ok = tile0 + mytile < THS * TWS * d * d;
    tile_origin(my_tile_ok ? tile0 + mytile : 0, my_py, my_px);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int gy = SMALL ? my_py + d * (e - 1) : y0 + 2 * tyl - 1 + e;
    const int gx = SMALL ? my_px + d * (e - 1) : x0 + 2 * txl - 1 + e;
    const bool oky = my_tile_ok && gy >= 0 && gy < a.H, okx = gx >= 0 && gx < a.W;
    row_off[e] = oky ? gy * a.W : 0;
    col_off[e] = okx ? gx : 0;
    rv |= oky ? (1u << e) : 0u;
    cv |= okx ? (1u << e) : 0u;
  }
  // Patch loads are BUFFER loads: 128-bit descriptor in SGPRs + 32-bit per-lane byte offset + scalar chunk offset, so
  // no 64-bit address VALU in the loop, and zero padding comes from the hardware range check (offset past the end of
  // the tensor returns 0) instead of a select.
  int p_boff[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const bool ok = (((rv >> (e / 4)) & (cv >> (e % 4))) & 1u) != 0;
    p_boff[e] = ok ? (row_off[e / 4] + col_off[e % 4]) * 4 : 0x7ffffff0;
  }
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((size_t)a.B * a.Cin * HW * 4), 0x00020000);

  // U chunk: float4 number v = tid + i*512 of the [16 pos][8 kc][64 co] block; consecutive i are 4 positions apart
  const int u_boff0 = (((tid >> 7) * a.Cin + ((tid >> 4) & 7)) * a.Cout + (tid & 15) * 4 + co0) * 4;   // bytes
  const int u_bstride = 4 * a.Cin * a.Cout * 4;

  f32x16 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

  float ureg[4][4];
  float dreg[16];

  // loads of one chunk in three parts (issued inside the first three MFMA groups of the previous chunk)
  auto load_part = [&](int c0, auto partc) {
    constexpr int part = decltype(partc)::value;
    if constexpr (part == 0) {
      // (ROCm 7.2's __builtin_amdgcn_raw_buffer_load_b128 lowers to a single-dword load: the weights keep plain
      //  16-byte global loads)
      const char* ub = reinterpret_cast<const char*>(a.wt) + (size_t)c0 * a.Cout * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 t4 = *reinterpret_cast<const float4*>(ub + u_boff0 + (size_t)i * u_bstride);
        ureg[i][0] = t4.x; ureg[i][1] = t4.y; ureg[i][2] = t4.z; ureg[i][3] = t4.w;
      }
    } else {
      const int soff = (int)(((size_t)b * a.Cin + c0 + kc_a) * HW * 4);
#pragma unroll
      for (int e = (part - 1) * 8; e < part * 8; ++e)
        dreg[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, p_boff[e], soff, 0));
    }
  };
  auto load_chunk = [&](int c0) {
    load_part(c0, std::integral_constant<int, 0>{});
    load_part(c0, std::integral_constant<int, 1>{});
    load_part(c0, std::integral_constant<int, 2>{});
  };

  // pieces 0..3: one float4 of U each; piece 4: B^T d B of this thread's (channel, tile) pair -> 16 LDS words
  auto store_piece = [&](float* st, auto qc) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q < 4) {
      reinterpret_cast<float4*>(st)[tid + q * 512] = make_float4(ureg[q][0], ureg[q][1], ureg[q][2], ureg[q][3]);
    } else {
      const float(&d)[16] = dreg;          // padding already reads as 0 (buffer range check)
      float tmp[16];                       // B^T d : rows (d0-d2, d1+d2, d2-d1, d1-d3)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        tmp[0 * 4 + c] = d[0 * 4 + c] - d[2 * 4 + c];
        tmp[1 * 4 + c] = d[1 * 4 + c] + d[2 * 4 + c];
        tmp[2 * 4 + c] = d[2 * 4 + c] - d[1 * 4 + c];
        tmp[3 * 4 + c] = d[1 * 4 + c] - d[3 * 4 + c];
      }
      float* vs = st + W_U_ELEMS + kc_a * W_TILES + mytile;
#pragma unroll
      for (int r = 0; r < 4; ++r) {        // (B^T d) B : columns the same way
        vs[(r * 4 + 0) * W_KC * W_TILES] = tmp[r * 4 + 0] - tmp[r * 4 + 2];
        vs[(r * 4 + 1) * W_KC * W_TILES] = tmp[r * 4 + 1] + tmp[r * 4 + 2];
        vs[(r * 4 + 2) * W_KC * W_TILES] = tmp[r * 4 + 2] - tmp[r * 4 + 1];
        vs[(r * 4 + 3) * W_KC * W_TILES] = tmp[r * 4 + 1] - tmp[r * 4 + 3];
      }
    }
  };

  // group g (8 per chunk): k-step ks = g / 2 (two channels), this wave's positions 4*(g%2) .. +3
  auto load_ops = [&](const float* cur, auto gc, float (&av)[4], float (&bv)[4]) {
    constexpr int g = decltype(gc)::value;
    constexpr int ks = g / 2, pg = g % 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      av[i] = cur[((pg * 4 + i) * W_KC + 2 * ks) * W_CO + a_base];
      bv[i] = cur[((pg * 4 + i) * W_KC + 2 * ks) * W_TILES + b_base];
    }
  };
  constexpr int GROUPS = 8, PIECES = 5, FIRST = GROUPS - PIECES;
  auto compute = [&](const float* cur, float* nxt, int c0_next, auto store_flag) {
    constexpr bool STORE = decltype(store_flag)::value;
    float av[2][4], bv[2][4];
    load_ops(cur, std::integral_constant<int, 0>{}, av[0], bv[0]);
    __builtin_amdgcn_sched_barrier(0);
    static_for<GROUPS>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int pg = g % 2;
      if constexpr (g + 1 < GROUPS) load_ops(cur, std::integral_constant<int, g + 1>{}, av[(g + 1) & 1], bv[(g + 1) & 1]);
      constexpr bool HAS_STORE = STORE && g >= FIRST;
      constexpr bool HAS_LOAD = STORE && g < 3;            // next chunk's global loads ride in groups 0..2
      if constexpr (HAS_LOAD) load_part(c0_next, std::integral_constant<int, g>{});
      if constexpr (HAS_STORE) store_piece(nxt, std::integral_constant<int, g - FIRST>{});
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[pg * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][i], bv[g & 1][i], acc[pg * 4 + i], 0, 0, 0);
      if constexpr (g + 1 < GROUPS) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      constexpr bool IS_V = HAS_STORE && (g - FIRST) >= 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (HAS_STORE) __builtin_amdgcn_sched_group_barrier(0x006, IS_V ? 12 : 2, 0);
        if constexpr (HAS_LOAD) {
          __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, g == 0 ? 1 : 2, 0);
        }
      }
      if constexpr (HAS_STORE) __builtin_amdgcn_sched_group_barrier(0x200, IS_V ? 16 : 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  const int n_chunks = a.Cin / W_KC;
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (a.dbg) t0 = __builtin_amdgcn_s_memtime();
  load_chunk(0);
  static_for<PIECES>([&](auto qc) { store_piece(lds, qc); });
  __syncthreads();
  if (a.dbg) t1 = __builtin_amdgcn_s_memtime();
  for (int ch = 0; ch + 1 < n_chunks; ++ch) {
    float* cur = lds + (ch & 1) * W_STAGE;
    float* nxt = lds + ((ch + 1) & 1) * W_STAGE;
    compute(cur, nxt, (ch + 1) * W_KC, std::true_type{});
    __syncthreads();
  }
  compute(lds + ((n_chunks - 1) & 1) * W_STAGE, nullptr, 0, std::false_type{});
  if (a.dbg) t2 = __builtin_amdgcn_s_memtime();

  // ---- epilogue: each wave reduces its 8 positions (rows 2ph, 2ph+1 of M) to a partial Y = A^T M A, the ph = 1
  //      wave hands its partial to the ph = 0 wave through LDS, which adds bias / residual and stores float2 ----
  // A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]: row a of M enters tt[0] with (1,1,1,0)[a] and tt[1] with (0,1,-1,-1)[a]
  const float c00 = ph == 0 ? 1.f : 1.f, c01 = ph == 0 ? 1.f : 0.f;     // tt[0] coefficients of rows (2ph, 2ph+1)
  const float c10 = ph == 0 ? 0.f : -1.f, c11 = ph == 0 ? 1.f : -1.f;   // tt[1] coefficients
  __syncthreads();                                                       // everyone is done with the stages
  float* ex = lds + ((wco * 2 + wtl) * 16) * 4 * 64;                     // [pair][r][4][lane]
  const int tile = wtl * 32 + j;
  int oy = y0 + 2 * (tile / W_TX), ox = x0 + 2 * (tile % W_TX);
  bool out_ok = oy < a.H && ox < a.W;
  if constexpr (SMALL) {
    out_ok = tile0 + tile < THS * TWS * d * d;
    tile_origin(out_ok ? tile0 + tile : 0, oy, ox);
  }
  float yv[16][4];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float tt[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float m0 = acc[c][r], m1 = acc[4 + c][r];                    // rows 2ph and 2ph+1, column c
      tt[0][c] = c00 * m0 + c01 * m1;
      tt[1][c] = c10 * m0 + c11 * m1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      yv[r][i * 2 + 0] = tt[i][0] + tt[i][1] + tt[i][2];
      yv[r][i * 2 + 1] = tt[i][1] - tt[i][2] - tt[i][3];
    }
  }
  // both waves of a pair finalize half of the 16 channel rows: wave ph keeps rows 8*ph .. 8*ph+7 and hands the
  // partials of the other 8 rows to its partner
  float* ex_out = ex + ((1 - ph) * 8 * 4) * 64;          // slots the PARTNER will read: [8 rows][4][lane]
  float* ex_in = ex + (ph * 8 * 4) * 64;
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      // static register indexing: select the row of the other half without a runtime index
      const float val = ph == 0 ? yv[8 + rr][v] : yv[rr][v];
      ex_out[(rr * 4 + v) * 64 + lane] = val;
    }
  }
  __syncthreads();
  if (out_ok) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = ph * 8 + rr;
      const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float own0 = ph == 0 ? yv[rr][i * 2 + 0] : yv[8 + rr][i * 2 + 0];
        const float own1 = ph == 0 ? yv[rr][i * 2 + 1] : yv[8 + rr][i * 2 + 1];
        float y0v = own0 + ex_in[(rr * 4 + i * 2 + 0) * 64 + lane] + bias;
        float y1v = own1 + ex_in[(rr * 4 + i * 2 + 1) * 64 + lane] + bias;
        const size_t o = ((size_t)b * a.Cout + co) * HW + (size_t)(oy + (SMALL ? d * i : i)) * a.W + ox;
        if constexpr (SMALL) {                 // the two outputs of a tile row are d pixels apart
          if (a.residual) {
            y0v += a.residual[o];
            y1v += a.residual[o + d];
          }
          if (a.out) {
            a.out[o] = y0v;
            a.out[o + d] = y1v;
          }
          if (a.out_act) {
            a.out_act[o] = a.act_out == IPDM_ACT_ELU ? fast_elu(y0v) : ipdm_act(y0v, a.act_out);
            a.out_act[o + d] = a.act_out == IPDM_ACT_ELU ? fast_elu(y1v) : ipdm_act(y1v, a.act_out);
          }
        } else {
          if (a.residual) {
            const float2 rr2 = *reinterpret_cast<const float2*>(a.residual + o);
            y0v += rr2.x;
            y1v += rr2.y;
          }
          if (a.out) *reinterpret_cast<float2*>(a.out + o) = make_float2(y0v, y1v);
          if (a.out_act) {
            const float e0 = a.act_out == IPDM_ACT_ELU ? fast_elu(y0v) : ipdm_act(y0v, a.act_out);
            const float e1 = a.act_out == IPDM_ACT_ELU ? fast_elu(y1v) : ipdm_act(y1v, a.act_out);
            *reinterpret_cast<float2*>(a.out_act + o) = make_float2(e0, e1);
          }
        }
      }
    }
  }
  if (a.dbg && tid == 0) {
    unsigned long long* d4 = a.dbg + (size_t)blockIdx.x * 4;
    d4[0] = t0; d4[1] = t1; d4[2] = t2; d4[3] = __builtin_amdgcn_s_memtime();
  }
}

}  // namespace

static bool wino_small(const ConvArgs& a) {
  // small images (or any dilation): linear tile space, needs H, W divisible by 2*dil
  return a.W < 32 || a.dil > 1;
}

bool wino_ok(const ConvArgs& a, int ks) {
  if (!(ks == 3 && a.D == 1 && a.Cin % W_KC == 0 && a.Cout % W_CO == 0 && !a.coef && a.act == IPDM_ACT_NONE)) return false;
  if (a.dil < 1 || a.dil > 4) return false;
  if ((size_t)a.B * a.Cin * a.H * a.W * 4 >= 0x7fffffffull) return false;   // buffer descriptors address < 2 GiB
  if (wino_small(a)) return a.H % (2 * a.dil) == 0 && a.W % (2 * a.dil) == 0 && (a.H * a.W) / 4 >= 32;
  return a.H % 2 == 0 && a.W % 2 == 0 && a.H >= 8;
}

// a.wt must be the Winograd-domain weights U [16][Cin][Cout] (ipdm_conv_wino_weight_f32)
int conv_wino_launch(ConvArgs a, hipStream_t s) {
  const bool small = wino_small(a);
  if (small) {
    a.tiles_x = ((a.H * a.W) / 4 + W_TILES - 1) / W_TILES;
    a.tiles_y = 1;
  } else {
    a.tiles_x = (a.W + 2 * W_TX - 1) / (2 * W_TX);
    a.tiles_y = (a.H + 2 * W_TY - 1) / (2 * W_TY);
  }
  a.co_tiles = a.Cout / W_CO;
  const int64_t nblk = (int64_t)a.B * a.tiles_x * a.tiles_y * a.co_tiles;
  if (nblk > 0x7fffffff) return IPDM_EUNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)W_LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)W_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  if (small) hipLaunchKernelGGL(conv_wino_kernel<true>, dim3((unsigned)nblk), dim3(512), W_LDS_BYTES, s, a);
  else hipLaunchKernelGGL(conv_wino_kernel<false>, dim3((unsigned)nblk), dim3(512), W_LDS_BYTES, s, a);
  return ipdm_launch_status();
}

int conv_wino_weights(const float* w, float* U, int Cout, int Cin, hipStream_t s) {
  const int64_t total = (int64_t)Cout * Cin;
  hipLaunchKernelGGL(wino_weight_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, s, w, U, Cout, Cin);
  return ipdm_launch_status();
}

}  // namespace ipdm_conv
