// Winograd F(2x2, 3x3) on the bf16 matrix cores with split fp32 operands -- wave-specialised form.
//
// Same arithmetic as conv_wino_bx3.hip (U pre-split [pos][ci/16][co/32][piece][lane], V = B^T d B in fp32 through
// LDS, split in registers right before the MFMAs, six bf16 MFMAs per 32x32x16 product), different division of labour.
// In the symmetric kernel every wave transforms, DMAs, loads fragments and multiplies: the HBM-latency input DMA and
// the L2-latency fragment loads retire through ONE in-order vmcnt queue per wave, and all eight waves are in the
// transform step at the same time with the matrix pipe idle.  Here the two waves of a SIMD have different jobs:
//   waves 0-3 (one per SIMD, "multipliers"): own one ROW of four Winograd positions for 64 channels x 32 tiles
//        (4 positions x 2 channel tiles = 128 accumulator registers); their only vector-memory traffic is the U
//        fragment stream, three (position, channel tile) units ahead; V comes from LDS.
//   waves 4-7 ("stagers"): issue the LDS-DMA of the raw input region three chunks ahead, transform the next chunk's
//        patches (two per lane) into the other V stage; nothing of theirs is ever waited on by a multiplier.
// One barrier per 16-channel chunk.  Workgroup = 64 channels x 32 tiles (TX x TY = 16 x 2: 4 x 32 output pixels, or
// 8 x 4 for 16-pixel images), persistent over an XCD-aware tile list like the symmetric kernel; epilogue: four rounds of
// M[pos 16][co 16][tile 32] through the V stage that was read last, all 512 threads gather / transform / store.
// LDS: two V stages (34 KiB each, channel rows 8-15 shifted by 32 floats so that the two halves of a B fragment read
// fall on disjoint banks) + three raw stages (15 KiB each) = 113 KiB.
// Needs W % 4 == 0 and a 16-byte aligned tensor (16-byte LDS-DMA quads); everything else stays on conv_wino_bx3.hip.
#include "conv_kernel.h"

namespace ipdm_conv {

namespace {

#ifndef Z_PROBE
#define Z_PROBE 0
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// two fp32 -> one dword of two bf16 (round to nearest even): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

constexpr int Z_KC = 16;
constexpr int Z_CO = 64;
constexpr int Z_TILES = 32;
constexpr int Z_PS = 16 * 32 + 32;                 // floats per position: [ci 16][tile 32] + the half shift
constexpr int Z_V_ELEMS = 16 * Z_PS;               // 8704 floats per V stage (>= 16 x 16 x 32 for the epilogue)
constexpr int Z_R_ELEMS = 15 * 256;                // 15 wave-instructions of 64 quads per raw stage
constexpr int Z_RSLOTS = 3;                        // raw stages: the DMA runs three chunks ahead of the multipliers
constexpr size_t Z_LDS_BYTES = (2 * (size_t)Z_V_ELEMS + Z_RSLOTS * Z_R_ELEMS) * sizeof(float);

template <int TX, int TY, bool CO_MAJOR>
__global__ __launch_bounds__(512) void conv_wino_bx3_spec_kernel(ConvArgs a, int total_tiles) {
  constexpr int probe = Z_PROBE;                              // diagnostic builds (-DZ_PROBE=<mask>) compile pieces out
  constexpr int QC = TX / 2 + 2, RC4 = 4 * QC;               // quads / floats per raw row (x0-4 .. x0+2TX+3)
  constexpr int QN = (2 * TY + 2) * QC;                       // quads per channel
  constexpr int NI = (Z_KC * QN + 63) / 64;                   // wave-instructions per chunk
  static_assert(NI <= 16 && NI * 256 <= Z_R_ELEMS, "quad image fits the raw stage");
  static_assert(TX * TY == Z_TILES, "32 tiles per workgroup");
  extern __shared__ __align__(16) float lds[];
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (a.dbg) t0 = __builtin_amdgcn_s_memtime();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mult = wave < 4;                                 // multiplier (0-3) or stager (4-7)
  const int role = wave & 3;                                  // position row / channel group
  const int h = lane >> 5, j = lane & 31;
  const int HW = a.H * a.W;
  const int n_cc = a.Cin / Z_KC, n_ct = a.Cout / 32;
  const int n_chunks = n_cc;                                  // >= 2 (launcher)
  const int p0 = 4 * role;

  // ---- this workgroup's tile list: first, stride, end (linear tile order) ----
  const int S = gridDim.x / 8;
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const int q = total_tiles / 8, r8 = total_tiles % 8;
  const int x_start = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
  const int x_end = x_start + q + (xcd < r8 ? 1 : 0);
  int tile = x_start + slot;
  if (tile >= x_end) return;                                  // uniform: whole workgroup

  struct Geo { int b, y0, x0, co_tile; };
  auto geo_of = [&](int L) {
    Geo g;
    const int n_px = total_tiles / a.co_tiles;
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
  float* const rs = lds + 2 * Z_V_ELEMS;
  const int mytile = lane & 31;

  // ---- stagers: raw region by 16-byte LDS-DMA; the chunk's 16 channels are one packed image of NI x 64 quads,
  //      stager r issues pieces r, r+4, r+8, r+12 ----
  int dma_off[4];
  int dma_b = 0;
  auto set_dma_geo = [&](const Geo& g) {
    dma_b = g.b;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = (role + 4 * k) * 64 + lane;
      const int cin = e / QN, qq = e - cin * QN;
      const int rr = qq / QC, qc = qq - rr * QC;
      const int gy = g.y0 - 1 + rr, gx0 = g.x0 - 4 + 4 * qc;
      const bool ok = e < Z_KC * QN && gy >= 0 && gy < a.H && gx0 >= 0 && gx0 < a.W;
      dma_off[k] = ok ? (cin * HW + gy * a.W + gx0) * 4 : 0x40000000;
    }
  };
  auto issue_dma = [&](int chunk, int rslot) {
    const int soff = (int)(((size_t)dma_b * a.Cin + chunk * Z_KC) * HW * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (role + 4 * k < NI) {                                // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)   // the 16-byte form only exists for gfx950: keep it out of the host pass
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            x_rsrc, (__attribute__((address_space(3))) void*)(rs + rslot * Z_R_ELEMS + (role + 4 * k) * 256), 16,
            dma_off[k], soff, 0, 0);
#endif
      }
  };
  float dreg[16];
  const int r_lane = (2 * (mytile / TX)) * RC4 + 2 * (mytile % TX) + 2;
  auto read_patch = [&](int kc, int rslot) {
    // a lane reads its own aligned pixel pair per patch row; the two outer columns come from the neighbouring tiles'
    // lanes by DPP (lanes of a 16-lane DPP row are consecutive tiles of a tile row), row ends read theirs
    const float* rp = rs + rslot * Z_R_ELEMS + kc * (QN * 4) + r_lane;
    const int txl = mytile % TX;
    const bool first = txl == 0, last = txl == TX - 1;
    const int edge = first ? 1 : 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float2 q1 = *reinterpret_cast<const float2*>(rp + rr * RC4 + 2);
      const float e = rp[rr * RC4 + edge];
      const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q1.y), 0x111, 0xf, 0xf, false));
      const float right = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q1.x), 0x101, 0xf, 0xf, false));
      dreg[rr * 4 + 0] = first ? e : left;
      dreg[rr * 4 + 1] = q1.x;
      dreg[rr * 4 + 2] = q1.y;
      dreg[rr * 4 + 3] = last ? e : right;
    }
  };
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
    float* vs = st + kc * 32 + (kc >> 3) * 32 + mytile;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      vs[(rr * 4 + 0) * Z_PS] = tmp[rr * 4 + 0] - tmp[rr * 4 + 2];
      vs[(rr * 4 + 1) * Z_PS] = tmp[rr * 4 + 1] + tmp[rr * 4 + 2];
      vs[(rr * 4 + 2) * Z_PS] = tmp[rr * 4 + 2] - tmp[rr * 4 + 1];
      vs[(rr * 4 + 3) * Z_PS] = tmp[rr * 4 + 1] - tmp[rr * 4 + 3];
    }
  };
  const int kc_a = 4 * role + h, kc_b = 4 * role + 2 + h;    // a stager lane's two channels of the chunk
  auto stage_chunk = [&](float* st, int rslot) {
    read_patch(kc_a, rslot);
    store_patch(st, kc_a);
    read_patch(kc_b, rslot);
    store_patch(st, kc_b);
  };

  // ---- multipliers ----
  const uint4* wq = reinterpret_cast<const uint4*>(a.wt);
  const size_t pos_stride = (size_t)n_cc * n_ct * 192;
  // one unit = the three pieces of (position p0 + u/2, channel tile u%2) of a chunk; four unit buffers, three units ahead
  auto unit_ptr = [&](int u, int cc, int co_tile) {
    return wq + (size_t)(p0 + (u >> 1)) * pos_stride + ((size_t)cc * n_ct + co_tile * 2 + (u & 1)) * 192 + lane;
  };
  auto load_A = [&](bf16x8 (&fr)[3], int u, int cc, int co_tile) {
    const uint4* base = unit_ptr(u, cc, co_tile);
#pragma unroll
    for (int s = 0; s < 3; ++s) fr[s] = __builtin_bit_cast(bf16x8, base[s * 64]);
  };
  const int b_lane = h * (8 * 32 + 32) + j;                   // channels 8h .. 8h+7 of tile j
  // B operand: V of one position, channels 8h .. 8h+7 of tile j, split into three bf16 pieces in packed pairs
  // (bq[buffer][piece][pair]); the split of the NEXT position is cut into eight slices that go into MFMA gaps
  u32x4 bq[2][3];
  float raw[8], res0[4], res1[4];
  auto load_raw = [&](const float* bpos) {
    const float* bp = bpos + b_lane;
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) raw[qq] = bp[qq * 32];
  };
  auto slice_a = [&](auto bc, auto qc) {                      // h and m pieces of pair q, residuals kept for slice_b
    constexpr int nb = decltype(bc)::value, qv = decltype(qc)::value;
    const float x0 = raw[2 * qv], x1 = raw[2 * qv + 1];
    const unsigned hp = cvt_pk_bf16(x0, x1);
    res0[qv] = x0 - __builtin_bit_cast(float, hp << 16);
    res1[qv] = x1 - __builtin_bit_cast(float, hp & 0xffff0000u);
    bq[nb][0][qv] = hp;
    bq[nb][1][qv] = cvt_pk_bf16(res0[qv], res1[qv]);
  };
  auto slice_b = [&](auto bc, auto qc) {
    constexpr int nb = decltype(bc)::value, qv = decltype(qc)::value;
    const unsigned mp = bq[nb][1][qv];
    const float s0 = res0[qv] - __builtin_bit_cast(float, mp << 16);
    const float s1 = res1[qv] - __builtin_bit_cast(float, mp & 0xffff0000u);
    bq[nb][2][qv] = cvt_pk_bf16(s0, s1);
  };

  f32x16 acc[4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[i >> 1][i & 1][rr] = 0.f;
  };
  zero_acc();

  // ---- prologue of the first tile ----
  Geo cur_g = geo_of(tile);
  bf16x8 afr[4][3];
  // one position of a chunk: 12 MFMAs (the two channel tiles' chains interleaved), everything else placed by hand into
  // their gaps and fenced there -- one multiplier per SIMD has no partner wave to cover a clump of VALU work:
  //   before: the raw V of the next position (4 x ds_read2_b32);  gaps 0-2: fragment unit 2pi+3;
  //   gaps 2-9: the split of the next position (6 / 5 VALU per gap);  gaps 10-11: unit 2pi+4 (unit 2pi's buffer)
  auto mstep = [&](auto sc, const float* bnext, int ch, int a_chunk, int a_cot) {
    constexpr int pi = decltype(sc)::value;
    constexpr int ub = (2 * pi) & 3, u1 = 2 * pi + 3, u2 = 2 * pi + 4;
    constexpr int cb = pi & 1;
    using NB = std::integral_constant<int, ((pi + 1) % 2)>;
    load_raw(bnext);
    const uint4* a1 = u1 < 8 ? unit_ptr(u1, ch, cur_g.co_tile) : unit_ptr(u1 - 8, a_chunk, a_cot);
    const uint4* a2 = u2 < 8 ? unit_ptr(u2, ch, cur_g.co_tile) : unit_ptr(u2 - 8, a_chunk, a_cot);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[cb][0]), bm = __builtin_bit_cast(bf16x8, bq[cb][1]),
                 bl = __builtin_bit_cast(bf16x8, bq[cb][2]);
    f32x16 v0 = acc[pi][0], v1 = acc[pi][1];
    __builtin_amdgcn_sched_barrier(0);
    static_for<12>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int c = i & 1, k = i >> 1;
      constexpr int ai = k == 0 ? 2 : (k == 2 || k == 3) ? 1 : 0;         // A piece: l h m m h h
      const bf16x8 bb = (k == 0 || k == 3 || k == 5) ? bh : (k == 1 ? bl : bm);   // B piece: h l m h m h
      if constexpr (c == 0)
        v0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ub][ai], bb, v0, 0, 0, 0);
      else
        v1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ub + 1][ai], bb, v1, 0, 0, 0);
      if constexpr (!(probe & 4)) {
        if constexpr (i < 3) afr[u1 & 3][i] = __builtin_bit_cast(bf16x8, a1[i * 64]);
        if constexpr (i == 10) {
          afr[u2 & 3][0] = __builtin_bit_cast(bf16x8, a2[0]);
          afr[u2 & 3][1] = __builtin_bit_cast(bf16x8, a2[64]);
        }
        if constexpr (i == 11) afr[u2 & 3][2] = __builtin_bit_cast(bf16x8, a2[128]);
      }
      if constexpr (i >= 2 && i < 10) {
        using Q = std::integral_constant<int, ((i - 2) / 2)>;
        if constexpr ((i & 1) == 0) slice_a(NB{}, Q{}); else slice_b(NB{}, Q{});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    acc[pi][0] = v0;
    acc[pi][1] = v1;
  };
  if (mult) {
    load_A(afr[0], 0, 0, cur_g.co_tile);
    load_A(afr[1], 1, 0, cur_g.co_tile);
    load_A(afr[2], 2, 0, cur_g.co_tile);
  } else {
    set_dma_geo(cur_g);
    issue_dma(0, 0);
    issue_dma(1, 1);
    issue_dma(2, 2);                                          // n_chunks >= 3 (launcher)
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (!mult) stage_chunk(lds, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (mult) {                                                 // position p0 of the first chunk
    load_raw(lds + p0 * Z_PS);
    static_for<4>([&](auto qc) {
      slice_a(std::integral_constant<int, 0>{}, qc);
      slice_b(std::integral_constant<int, 0>{}, qc);
    });
  }
  int rs_next = 1;                                            // raw slot of the chunk the stagers transform next
  if (a.dbg) t1 = __builtin_amdgcn_s_memtime();

  int g = 0;                                                  // chunks done so far: stage parity
  while (true) {
    const int next_tile = tile + S;
    const bool has_next = next_tile < x_end;
    const Geo next_g = geo_of(has_next ? next_tile : tile);
    for (int ch = 0; ch < n_chunks; ++ch, ++g) {
      const float* cur = lds + (g & 1) * Z_V_ELEMS;
      float* nxt = lds + ((g + 1) & 1) * Z_V_ELEMS;
      // the chunk after this one (A fragments of its first units)
      const bool a_next = ch + 1 >= n_chunks;
      const int a_chunk = a_next ? 0 : ch + 1;
      const int a_cot = a_next ? next_g.co_tile : cur_g.co_tile;
      if (mult) {
        static_for<3>([&](auto sc) { mstep(sc, cur + (p0 + decltype(sc)::value + 1) * Z_PS, ch, a_chunk, a_cot); });
      } else {
        // the chunk three ahead in the stream (DMA target: the slot transformed in the previous iteration), the next
        // one (transform)
        const bool dma_next = ch + 3 >= n_chunks;
        const int dma_chunk = dma_next ? ch + 3 - n_chunks : ch + 3;
        if (ch + 3 == n_chunks) set_dma_geo(next_g);          // from here on the DMA belongs to the next tile
        const int rs_dma = rs_next == 0 ? Z_RSLOTS - 1 : rs_next - 1;
        if constexpr (!(probe & 2)) issue_dma(dma_chunk, rs_dma);
        if constexpr (!(probe & 1)) stage_chunk(nxt, rs_next);
        rs_next = rs_next == Z_RSLOTS - 1 ? 0 : rs_next + 1;
      }
      // multipliers keep their fragment prefetch in flight across the barrier (only LDS is waited for); stagers wait
      // for their DMA and their V stores
      // (all but the DMA batch issued in this iteration: four pieces, three for the last stager)
      if (mult)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else if (role + 12 < NI)
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      // the last position of the chunk; its gaps read and split position p0 of the NEXT chunk (complete since the barrier)
      if (mult) mstep(std::integral_constant<int, 3>{}, nxt + p0 * Z_PS, ch, a_chunk, a_cot);
    }
    if (a.dbg) t2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue of this tile in the stage its last chunk was read from (the other stage and the raw stages already
    //      hold the next tile); rounds over (channel tile c, channel half hf) ----
    float* ms = lds + ((g - 1) & 1) * Z_V_ELEMS;              // M[pos 16][co 16][tile 32]
    const int etile = tid & 31, ecl = tid >> 5;               // this thread: tile, channel of the 16
    const int co0 = cur_g.co_tile * Z_CO;
    static_for<4>([&](auto rc) {
      constexpr int rnd = decltype(rc)::value;
      constexpr int c = rnd >> 1, hf = rnd & 1;
      if constexpr ((probe & 8) != 0) return;
      if (mult) {
#pragma unroll
        for (int pi = 0; pi < 4; ++pi)
#pragma unroll
          for (int rq = 0; rq < 8; ++rq) {
            const int col = (rq & 3) + 8 * (rq >> 2) + 4 * h;
            ms[((p0 + pi) * 16 + col) * 32 + j] = acc[pi][c][8 * hf + rq];
          }
      }
      __syncthreads();
      const int oy = cur_g.y0 + 2 * (etile / TX), ox = cur_g.x0 + 2 * (etile % TX);
      if (oy < a.H && ox < a.W) {
        const int co = co0 + c * 32 + hf * 16 + ecl;
        float m[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) m[p] = ms[(p * 16 + ecl) * 32 + etile];
        float tt[2][4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          tt[0][qq] = m[0 * 4 + qq] + m[1 * 4 + qq] + m[2 * 4 + qq];
          tt[1][qq] = m[1 * 4 + qq] - m[2 * 4 + qq] - m[3 * 4 + qq];
        }
        const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          float y0v = tt[ii][0] + tt[ii][1] + tt[ii][2] + bias;
          float y1v = tt[ii][1] - tt[ii][2] - tt[ii][3] + bias;
          const size_t o = ((size_t)cur_g.b * a.Cout + co) * HW + (size_t)(oy + ii) * a.W + ox;
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
      __syncthreads();                                        // M is rewritten by the next round / the next tile's V
    });
    if (!has_next) break;
    zero_acc();
    tile = next_tile;
    cur_g = next_g;
  }
  if (a.dbg) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* d4 = a.dbg + (size_t)blockIdx.x * 4;
      d4[0] = t0; d4[1] = t1; d4[2] = t2; d4[3] = t3;
    }
  }
}

}  // namespace

// small_dma: undilated images of up to 16 x 16 pixels (8 x 4 tile block, channel-tile-major); otherwise wide images.
// The caller has checked W % 4 == 0, the tensor's 16-byte alignment and Cin >= 48.
int conv_wino_bx3_spec_launch(ConvArgs a, bool small_dma, int cus_per_xcd, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    const void* ks[] = {reinterpret_cast<const void*>(conv_wino_bx3_spec_kernel<16, 2, false>),
                        reinterpret_cast<const void*>(conv_wino_bx3_spec_kernel<8, 4, true>)};
    for (const void* k : ks) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Z_LDS_BYTES);
      if (e != hipSuccess) return (int)e;
    }
    attr_set = true;
  }
  if (small_dma) {
    a.tiles_x = (a.W + 15) / 16;
    a.tiles_y = (a.H + 7) / 8;
  } else {
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + 3) / 4;
  }
  a.co_tiles = a.Cout / Z_CO;
  const int64_t n = (int64_t)a.B * a.tiles_x * a.tiles_y * a.co_tiles;
  if (n > 0x7fffffff) return IPDM_EUNSUPPORTED;
  const int px = (int)((n + 7) / 8);
  const int S = px < cus_per_xcd ? px : cus_per_xcd;
  if (small_dma)
    hipLaunchKernelGGL((conv_wino_bx3_spec_kernel<8, 4, true>), dim3((unsigned)(8 * S)), dim3(512), Z_LDS_BYTES, s, a, (int)n);
  else
    hipLaunchKernelGGL((conv_wino_bx3_spec_kernel<16, 2, false>), dim3((unsigned)(8 * S)), dim3(512), Z_LDS_BYTES, s, a, (int)n);
  return ipdm_launch_status();
}

}  // namespace ipdm_conv
