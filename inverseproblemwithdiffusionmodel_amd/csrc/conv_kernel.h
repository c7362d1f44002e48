// Templates of the fp32 MFMA implicit-GEMM convolution; design notes in conv.hip.
#pragma once
#include "ipdm_common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace ipdm_conv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// exact three-way split of eight fp32 values into bf16 pieces: v = h + m + l (8 + 8 + 8 significand bits; both
// residual subtractions are exact in fp32)
__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hi = (__bf16)v[i];
    const float r1 = v[i] - (float)hi;
    const __bf16 mi = (__bf16)r1;
    const float r2 = r1 - (float)mi;
    h[i] = hi;
    m[i] = mi;
    l[i] = (__bf16)r2;
  }
}

// ---- "f16x2": fp32 operands as TWO fp16 pieces, three fp16 MFMAs per product ------------------------------------------
// v = h + l + e with h = f16(v) (round to nearest even), l = f16(v - h) (the subtraction is exact in fp32) and
// |e| <= 2^-24 |v| while l is a normal fp16 number (|v| >~ 2^-3 / scale), |e| <= 2^-25 / scale below that: 11 + 11
// significand bits and two signs carry what an fp32 rounding keeps.  w * x = wh*xh + (wh*xl + wl*xh) + dropped, the
// dropped wl*xl + e-terms bounded by 3 * 2^-24 |w x| -- the size of the fp32 roundings an fp32 fma chain commits per
// term anyway.  Three v_mfma_f32_32x32x16_f16 (fp32 accumulate, every fp16 x fp16 product exact in fp32) instead of the
// six bf16 ones of the three-way bf16 split: half the matrix-core cycles, 4 instead of 6 bytes per weight, 2 instead of
// 5.5 VALU per split element.  Price: fp16's exponent range.  Weights are scaled per output channel by a power of two
// at pack time (max |w s| in [2^13, 2^14); the inverse scale rides in the blob and is applied in the epilogue's fma);
// activations must satisfy |x| < 65504 (the Winograd kernels pre-scale their 4-term input transform by 1/4 to keep that same
// bound) -- an input beyond it gives inf -> NaN in the output, never a silently wrong finite value; IPDM_CONV_IMPL=bx3 keeps
// the whole fp32 exponent range.  Measured against float64 on the networks' layer shapes the error is at or below the three-way bf16
// split's and the exact-fp32 MFMA kernel's.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// two fp32 values -> packed (hi, hi) and (lo, lo) fp16 pairs: v_cvt_pk_f16_f32, two v_fma_mix_f32 (v - f32(h), exact),
// v_cvt_pk_f16_f32 -- 2 VALU per element
__device__ __forceinline__ void split2_pk(float v0, float v1, unsigned& hp, unsigned& lp) {
  const f16x2 hh = {(_Float16)v0, (_Float16)v1};
  hp = __builtin_bit_cast(unsigned, hh);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hp), "v"(v0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hp), "v"(v1));
  const f16x2 ll = {(_Float16)r0, (_Float16)r1};
  lp = __builtin_bit_cast(unsigned, ll);
}
// the same with a power-of-two pre-scale s (the Winograd kernels: s = 1/4 undoes the growth of the 4-term input transform,
// so that their range contract is the direct kernels' |x| < 65504; the epilogue's inverse weight scale carries the 4):
// v_fma_mixlo_f16 / v_fma_mixhi_f16 (f16(v * s), exact scaling), two v_fma_mix_f32 (v * s - f32(h), exact), v_cvt_pk_f16_f32
__device__ __forceinline__ void split2_pk_scaled(float v0, float v1, float s, unsigned& hp, unsigned& lp) {
  unsigned h;                              // mixlo defines the low half (the high half, kept from whatever the register held,
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v0), "v"(s));        // is overwritten by mixhi): no zero-initialising move
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(v1), "v"(s));
  hp = h;
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v0), "v"(s), "v"(h));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v1), "v"(s), "v"(h));
  const f16x2 ll = {(_Float16)r0, (_Float16)r1};
  lp = __builtin_bit_cast(unsigned, ll);
}
constexpr float HX_WINO_PRESCALE = 0.25f;

// per-image DYNAMIC input scale (ConvArgs.in_amax: max |x| of every image, from ipdm_absmax_f32): the power of two s with
// amax * s in [2^14, 2^15) -- fp16's range then fits ANY fp32 input (the scale and its inverse are exact), and values down to
// 2^-17 of the image's maximum keep all 22 bits.  -> (s, 1 / s); amax zero / denormal / non-finite -> (1, 1).
__device__ __forceinline__ void hx_dynamic_scale(float amax, float& s, float& inv_s) {
  const unsigned e = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;     // amax in [2^(e-127), 2^(e-126))
  const bool ok = e >= 15u && e <= 253u;                                     // keeps both exponent fields in [1, 254]
  s = __builtin_bit_cast(float, ok ? (268u - e) << 23 : 0x3f800000u);        // 2^(141 - e)
  inv_s = __builtin_bit_cast(float, ok ? (e - 14u) << 23 : 0x3f800000u);     // 2^(e - 141)
}
// split2 with a power-of-two pre-scale (direct kernel, dynamic range)
__device__ __forceinline__ void split2_scaled(const float (&v)[8], float s, uint4& h, uint4& l) {
  split2_pk_scaled(v[0], v[1], s, h.x, l.x);
  split2_pk_scaled(v[2], v[3], s, h.y, l.y);
  split2_pk_scaled(v[4], v[5], s, h.z, l.z);
  split2_pk_scaled(v[6], v[7], s, h.w, l.w);
}

// eight consecutive k values -> the hi and the lo MFMA operand
__device__ __forceinline__ void split2(const float (&v)[8], uint4& h, uint4& l) {
  split2_pk(v[0], v[1], h.x, l.x);
  split2_pk(v[2], v[3], h.y, l.y);
  split2_pk(v[4], v[5], h.z, l.z);
  split2_pk(v[6], v[7], h.w, l.w);
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}).
// Every register-array index below is a constant expression, so nothing is ever demoted to scratch.
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// sum over the 32 lanes of a half-wave, the total in every lane: four DPP adds inside the 16-lane rows (quad swaps, then the
// half-row and row mirrors) and one cross-row exchange -- a fixed order, and a fifth of the latency of five bpermutes
__device__ __forceinline__ float half_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
  return v + __shfl_xor(v, 16, 64);
}


struct ConvArgs {
  const float* x;
  const float* wt;
  const float* bias;
  const float* coef;
  const float* residual;
  float* out;      // raw result (may be NULL when only out_act is wanted)
  float* out_act;  // act_out(result) (may be NULL)
  int act_out;
  int B, Cin, Cout, H, W, dil, act;
  int D, kd;       // depth slices per volume (1 = plain 2-D) and depth taps (1 or 3): 3-D convolution as extra K chunks
  int tiles_x, tiles_y, co_tiles;
  unsigned long long* dbg;   // optional in-kernel stamps (diagnostic builds / IPDM tuning only; NULL in production)
  int pool2 = 0;             // conv_wino_bx3 wide kernels only: write the 2x2 MEAN of every output tile (ConvMeanPool,
                             //   layers.py:291-313) to out / out_act [B][Cout][H/2][W/2]; residual is at that size too
  int ksplit = 1;            // conv_bx3 only: the K (input channel x depth tap) chunks are dealt to ksplit workgroups
  float* partial = nullptr;  //   per tile, each writing its raw partial sums to partial[ks][B][Cout][D*H*W]
  int bias_bstride = 0;      // split-operand kernels: bias index = image * bias_bstride + channel (Cout: one bias row per image, e.g.
                             //   conv bias + the time-embedding shift of a score_sde block; 0: the usual per-channel bias)
  float out_scale = 1.f;     //   ... and result = (conv + bias + residual) * out_scale (score_sde skip_rescale: 1 / sqrt 2)
  const float* in_amax = nullptr;   // f16x2 only: maxima vector of the input ([B][IPDM_AMAX_SLOT], ipdm.h) -> per-image power-of-two
                                    //   input scale (hx_dynamic_scale(ipdm_amax_read(.))); NULL: static range contract |x| < 65504
  int res_second = 0;        // split-operand kernels: the residual enters out_act only (out = conv + bias; ipdm_conv_ext_t)
  float* amax_out = nullptr; // split-operand kernels: maxima vectors ([B][IPDM_AMAX_SLOT], zeroed by the caller) of out / out_act: max
  float* amax_act = nullptr; //   |stored value| per image, one atomic max per wave (ipdm_common.h) -- the NEXT convolution's in_amax
  int hx = 0;                // conv_bx3 / conv_wino_bx3: 1 = the weights are an f16x2 blob (two fp16 pieces + per-channel inverse
                             //   scales), run the three-MFMA fp16 instantiation
  int phase_step = 0, phase_mask = 0;   // conv_wino1d (tuning, IPDM_W1D_STAGGER): workgroup slot & phase_mask starts g * phase_step x 512 cycles late
  float* stats = nullptr;    // conv_wino_bx3 wide kernel (16 x 4 tile block, 16-byte DMA) only: per-plane statistics of
                             //   the RESULT as deterministic partials [B][Cout][P][3] = (count, mean, sum of squared
                             //   deviations) per (tile block, tile group), P = 2 * tiles_y * tiles_x -- what the following
                             //   InstanceNorm++ needs, so that it does not read the tensor again
};

// split-K second pass (conv_bx3.hip): out = (bias + sum_s partial[s] (fixed order) + residual) * out_scale; out_act = act_out(out)
__global__ __launch_bounds__(256) void bx3_splitk_reduce_kernel(const float* __restrict__ partial, int ksplit,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ residual, float* out,
                                                                float* out_act, int act_out, int Cout, int64_t plane,
                                                                int64_t total, int bias_bstride, float out_scale,
                                                                float* amax_out, float* amax_act, int res_second);

// its grid: 1-D grid-stride, or (blocks per image, images) when per-image maxima are wanted
inline dim3 splitk_reduce_grid(int B, int64_t per_image, bool by_image) {
  if (!by_image || B > 65535) return dim3((unsigned)ipdm_ew_grid((int64_t)B * per_image, 256));
  int64_t bx = (per_image + 255) / 256, cap = (2048 + B - 1) / B;
  bx = bx < 1 ? 1 : (bx > cap ? cap : bx);
  return dim3((unsigned)bx, (unsigned)B);
}

// optional extras of the split-operand entry points (include/ipdm.h: ipdm_conv_ext_t) -> ConvArgs
inline void conv_apply_ext(ConvArgs& a, const ipdm_conv_ext_t* ext, int hx) {
  a.in_amax = hx && ext ? ext->in_amax : nullptr;
  a.bias_bstride = ext ? ext->bias_bstride : 0;
  a.out_scale = ext && ext->out_scale != 0.f ? ext->out_scale : 1.f;
  a.res_second = ext ? (ext->res_second != 0) : 0;
  a.amax_out = ext ? ext->out_amax : nullptr;
  a.amax_act = ext ? ext->act_amax : nullptr;
}

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
  static constexpr int BUF_ELEMS = W_ELEMS + P_ELEMS;            // one LDS stage
  static constexpr int W_VEC = (W_ELEMS / 4 + 255) / 256;        // float4 weight pieces per thread
  static constexpr int P_POS = (PHP * PWP + 255) / 256;          // patch positions per thread
  static constexpr int PIECES = W_VEC + P_POS * KC;              // LDS-store pieces per thread per chunk
  static constexpr int GROUPS = TAPS * (KC / 2);                 // MFMA groups per chunk
  static constexpr int PPG = (PIECES + GROUPS - 1) / GROUPS;     // store pieces interleaved into one MFMA group
  static constexpr int STORE_GROUPS = (PIECES + PPG - 1) / PPG;
  static constexpr int FIRST_STORE_GROUP = GROUPS - STORE_GROUPS; // stores go at the END of the chunk: the global
                                                                  // loads issued at its start have landed by then
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_ELEMS * sizeof(float);
};

// ELU in the convolution epilogues, five instructions (v_mul, v_exp, v_add, v_cmp, v_cndmask): exp(v) - 1 is within ~1.2e-7
// ABSOLUTE of expm1(v) everywhere (it loses RELATIVE accuracy for |v| << 1, where the value itself is negligible next to the
// O(1) values it is summed with in the next convolution).  The 4-term series that round 1-3 selected for |v| < 1/16 cost seven
// more VALU per value: ~900 issue cycles per wave and tile pass in the wide Winograd kernel (IPDM_ELU_SERIES restores it).
__device__ __forceinline__ float fast_elu(float v) {
  const float e = __expf(v) - 1.f;
#ifdef IPDM_ELU_SERIES
  const float p = v * (1.f + v * (0.5f + v * (0.16666667f + v * 0.041666668f)));
  const float neg = v > -0.0625f ? p : e;
  return v > 0.f ? v : neg;
#else
  return v > 0.f ? v : e;
#endif
}

// Pipeline per workgroup (the wave hides its own latencies):
//   prologue : global -> regs -> LDS stage 0
//   chunk ch : issue the global loads of chunk ch+1 into registers
//              TAPS x KC/2 groups of MFMAs on LDS stage ch&1; the LDS stores of chunk ch+1 (normalised +
//              activated on the way) into stage (ch+1)&1 are interleaved behind the LAST groups
//              one workgroup barrier
// FAST (Cin % KC == 0, Cout % CO_T == 0): the steady state is one branch-free basic block -- loads and LDS
// stores are unconditional (clamped addresses, zero-select for padding), ACT / NORM are compile-time.
// !FAST: every guard is a run-time test (ragged channel counts, any activation code).
template <int NCT, int NPT, int WCO, int WPX, int PW, int DMAX, int KC, int KS, bool FAST, int ACT, bool NORM>
__global__ __launch_bounds__(256, 1) void conv_mfma_kernel(ConvArgs a) {
  using C = ConvCfg<NCT, NPT, WCO, WPX, PW, DMAX, KC, KS>;
  static_assert(WCO * WPX == 4, "four waves per workgroup");
  static_assert(KC % 2 == 0, "MFMA k-step is two input channels");
  extern __shared__ __align__(16) float lds[];

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
  const int bz = t / a.tiles_y;               // (volume, depth slice) pair; D == 1: the image index
  const int b = bz / a.D, z = bz - b * a.D;
  const int co0 = co_tile * C::CO_T;
  const int y0 = ty * C::PH, x0 = tx * PW;
  const int d = KS == 3 ? a.dil : 0;          // halo actually used (<= DMAX)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int wco = wave / WPX, wpx = wave % WPX;
  const int HW = a.H * a.W;
  const size_t cs = (size_t)a.D * HW;          // channel stride of x / out ([B][C][D][H][W])

  // depth taps whose input slice exists (3-D only): each one contributes n_cc more K chunks
  const int n_cc = (a.Cin + KC - 1) / KC;
  int kz_list[3] = {0, 0, 0};
  int n_kz = 0;
  for (int kz = 0; kz < a.kd; ++kz) {
    const int zi = z + (kz - a.kd / 2) * a.dil;
    if (zi >= 0 && zi < a.D) kz_list[n_kz++] = kz;
  }

  // ---- per-lane LDS read offsets (floats, relative to a stage base) ----
  const int a_base = h * C::CO_T + wco * NCT * 32 + j;
  int b_base[NPT];
#pragma unroll
  for (int n = 0; n < NPT; ++n) {
    const int tile = wpx * NPT + n;
    const int prow = PW == 32 ? tile : tile * 2 + (j >> 4);
    const int pcol = PW == 32 ? j : (j & 15);
    b_base[n] = C::W_ELEMS + h * C::PLANE + (prow + d) * C::PITCH + pcol + d;
  }

  f32x16 acc[NCT][NPT];
#pragma unroll
  for (int m = 0; m < NCT; ++m)
#pragma unroll
    for (int n = 0; n < NPT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // ---- per-thread staging geometry (identical for every chunk) ----
  // Threads beyond the last piece are clamped onto it: they load and store the very same value (benign).
  const int pwv = PW + 2 * d, phv = C::PH + 2 * d;      // valid patch extent for this dilation
  int p_lds[C::P_POS];       // LDS offset of the position inside a channel plane
  int p_gofs[C::P_POS];      // offset inside a channel plane of x (clamped to 0 when padding)
  bool p_valid[C::P_POS];    // false: zero padding
#pragma unroll
  for (int i = 0; i < C::P_POS; ++i) {
    int p = tid + i * 256;
    p = p < phv * pwv ? p : phv * pwv - 1;
    const int r = p / pwv, c = p - r * pwv;
    const int gy = y0 - d + r, gx = x0 - d + c;
    p_lds[i] = r * C::PITCH + c;
    p_valid[i] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    p_gofs[i] = p_valid[i] ? gy * a.W + gx : 0;
  }
  int w_lds[C::W_VEC];       // float4 index inside the stage
  int w_gofs[C::W_VEC];      // float offset into wt of (tap, kc, q) for channel chunk 0 and co0 = 0
  int w_kc[C::W_VEC];
#pragma unroll
  for (int i = 0; i < C::W_VEC; ++i) {
    int v = tid + i * 256;
    v = v < C::W_ELEMS / 4 ? v : C::W_ELEMS / 4 - 1;
    const int row = v / (C::CO_T / 4), q = v % (C::CO_T / 4);
    w_lds[i] = v;
    w_kc[i] = row % KC;
    w_gofs[i] = ((row / KC) * a.Cin + (row % KC)) * a.Cout + q * 4;
  }

  float wreg[C::W_VEC][4];
  float preg[C::P_POS][KC];
  float cfreg[KC][3];        // InstanceNorm++ (mu, scale, shift) of the chunk's channels (wave-uniform -> SGPRs)

  auto load_chunk = [&](int ch) {
    const int kzi = ch / n_cc;
    const int c0 = (ch - kzi * n_cc) * KC;
    const int kz = kzi == 0 ? kz_list[0] : (kzi == 1 ? kz_list[1] : kz_list[2]);
    const int zi = z + (kz - a.kd / 2) * a.dil;
    const float* wb = a.wt + ((size_t)kz * C::TAPS * a.Cin + c0) * a.Cout + co0;
#pragma unroll
    for (int i = 0; i < C::W_VEC; ++i) {
      if constexpr (FAST) {
        const float4 t4 = *reinterpret_cast<const float4*>(wb + w_gofs[i]);
        wreg[i][0] = t4.x; wreg[i][1] = t4.y; wreg[i][2] = t4.z; wreg[i][3] = t4.w;
      } else {
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + w_kc[i] < a.Cin) {
          const float* src = wb + w_gofs[i];
          const int co = co0 + (w_lds[i] % (C::CO_T / 4)) * 4;
          if (co + 0 < a.Cout) val.x = src[0];
          if (co + 1 < a.Cout) val.y = src[1];
          if (co + 2 < a.Cout) val.z = src[2];
          if (co + 3 < a.Cout) val.w = src[3];
        }
        wreg[i][0] = val.x; wreg[i][1] = val.y; wreg[i][2] = val.z; wreg[i][3] = val.w;
      }
    }
    if constexpr (FAST && NORM) {
      const float* cf = a.coef + ((size_t)b * a.Cin + c0) * 3;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        cfreg[kc][0] = cf[kc * 3 + 0];
        cfreg[kc][1] = cf[kc * 3 + 1];
        cfreg[kc][2] = cf[kc * 3 + 2];
      }
    }
    const float* xb = a.x + (((size_t)b * a.Cin + c0) * a.D + zi) * HW;
#pragma unroll
    for (int i = 0; i < C::P_POS; ++i) {
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        if constexpr (FAST) {
          preg[i][kc] = xb[(size_t)kc * cs + p_gofs[i]];
        } else {
          float v = 0.f;
          if (p_valid[i] && c0 + kc < a.Cin) v = xb[(size_t)kc * cs + p_gofs[i]];
          preg[i][kc] = v;
        }
      }
    }
  };
  auto chunk_c0 = [&](int ch) { return (ch % n_cc) * KC; };

  // one LDS-store piece of the chunk held in (wreg, preg) into stage `st`; the piece index is a type
  auto store_piece = [&](float* st, int c0, auto qc) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q < C::W_VEC) {
      reinterpret_cast<float4*>(st)[w_lds[q]] = make_float4(wreg[q][0], wreg[q][1], wreg[q][2], wreg[q][3]);
    } else if constexpr (q < C::PIECES) {
      constexpr int i = (q - C::W_VEC) / KC, kc = (q - C::W_VEC) % KC;
      float v = preg[i][kc];
      if constexpr (FAST) {
        if constexpr (NORM) v = (v - cfreg[kc][0]) * cfreg[kc][1] + cfreg[kc][2];
        if constexpr (ACT == IPDM_ACT_ELU) v = fast_elu(v);
        else v = ipdm_act_t<ACT>(v);
        v = p_valid[i] ? v : 0.f;
      } else {
        if (p_valid[i] && c0 + kc < a.Cin) {
          if (a.coef) {
            const float* cf = a.coef + ((size_t)b * a.Cin + c0 + kc) * 3;
            v = (v - cf[0]) * cf[1] + cf[2];
          }
          v = ipdm_act(v, a.act);
        } else {
          v = 0.f;
        }
      }
      st[C::W_ELEMS + kc * C::PLANE + p_lds[i]] = v;
    }
  };

  // LDS operand reads of MFMA group g (tap = g / (KC/2), k-step = g % (KC/2))
  auto load_ops = [&](const float* cur, auto gc, float (&av)[NCT], float (&bv)[NPT]) {
    constexpr int g = decltype(gc)::value;
    constexpr int tap = g / (KC / 2), ks = g % (KC / 2);
    constexpr int dy = KS == 3 ? tap / 3 - 1 : 0, dx = KS == 3 ? tap % 3 - 1 : 0;
    const int tap_off = (dy * C::PITCH + dx) * d;
#pragma unroll
    for (int m = 0; m < NCT; ++m) av[m] = cur[(tap * KC + 2 * ks) * C::CO_T + a_base + 32 * m];
#pragma unroll
    for (int n = 0; n < NPT; ++n) bv[n] = cur[2 * ks * C::PLANE + b_base[n] + tap_off];
  };

  // MFMAs of one chunk from stage `cur`; when STORE, interleave the stores of the next chunk into `nxt`.
  // Software pipeline, pinned with sched_barrier: the LDS operand reads of group g+1 are issued BEFORE the
  // MFMAs of group g (left to itself hipcc emits read -> wait -> mfma for every single MFMA).
  auto compute = [&](const float* cur, float* nxt, int c0_next, auto store_flag) {
    constexpr bool STORE = decltype(store_flag)::value;
    float av[2][NCT], bv[2][NPT];
    load_ops(cur, std::integral_constant<int, 0>{}, av[0], bv[0]);
    __builtin_amdgcn_sched_barrier(0);
    static_for<C::GROUPS>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      // one scheduling region per group: operand reads of g+1, the MFMAs of g, and (STORE) one or more store
      // pieces whose VALU work (normalise, ELU, padding select) is dealt out BETWEEN the MFMAs so that it runs
      // in the shadow of the 64-cycle matrix instructions instead of behind them.
      if constexpr (g + 1 < C::GROUPS)
        load_ops(cur, std::integral_constant<int, g + 1>{}, av[(g + 1) & 1], bv[(g + 1) & 1]);
      constexpr bool HAS_STORE = STORE && g >= C::FIRST_STORE_GROUP;
      if constexpr (HAS_STORE) {
        static_for<C::PPG>([&](auto uc) {
          store_piece(nxt, c0_next, std::integral_constant<int, (g - C::FIRST_STORE_GROUP) * C::PPG + decltype(uc)::value>{});
        });
      }
#pragma unroll
      for (int m = 0; m < NCT; ++m)
#pragma unroll
        for (int n = 0; n < NPT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][m], bv[g & 1][n], acc[m][n], 0, 0, 0);
      if constexpr (g + 1 < C::GROUPS) __builtin_amdgcn_sched_group_barrier(0x100, NCT + NPT, 0);   // DS reads first
      constexpr int VALU_PER_MFMA = HAS_STORE ? (24 * C::PPG + NCT * NPT - 1) / (NCT * NPT) : 0;
#pragma unroll
      for (int i = 0; i < NCT * NPT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          // one MFMA
        if constexpr (HAS_STORE) __builtin_amdgcn_sched_group_barrier(0x006, VALU_PER_MFMA, 0);    // VALU/SALU in its shadow
      }
      if constexpr (HAS_STORE) __builtin_amdgcn_sched_group_barrier(0x200, C::PPG, 0);             // the piece's DS write(s)
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  const int n_chunks = n_cc * n_kz;            // n_kz >= 1: the centre depth tap always exists
  load_chunk(0);
  static_for<C::PIECES>([&](auto qc) { store_piece(lds, 0, qc); });
  __syncthreads();
  for (int ch = 0; ch + 1 < n_chunks; ++ch) {
    float* cur = lds + (ch & 1) * C::BUF_ELEMS;
    float* nxt = lds + ((ch + 1) & 1) * C::BUF_ELEMS;
    load_chunk(ch + 1);
    compute(cur, nxt, chunk_c0(ch + 1), std::true_type{});
    __syncthreads();
  }
  compute(lds + ((n_chunks - 1) & 1) * C::BUF_ELEMS, nullptr, 0, std::false_type{});

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
        if (FAST || co < a.Cout) {
          const size_t o = (((size_t)b * a.Cout + co) * a.D + z) * HW + (size_t)gy * a.W + gx;
          float v = acc[m][n][r];
          if (a.bias) v += a.bias[co];
          if (a.residual) v += a.residual[o];
          if (a.out) a.out[o] = v;
          if (a.out_act) a.out_act[o] = a.act_out == IPDM_ACT_ELU ? fast_elu(v) : ipdm_act(v, a.act_out);
        }
      }
    }
  }
}

template <int NCT, int NPT, int WCO, int WPX, int PW, int DMAX, int KC, int KS, bool FAST, int ACT, bool NORM>
int launch_conv(ConvArgs a, hipStream_t s) {
  using C = ConvCfg<NCT, NPT, WCO, WPX, PW, DMAX, KC, KS>;
  a.tiles_x = (a.W + PW - 1) / PW;
  a.tiles_y = (a.H + C::PH - 1) / C::PH;
  a.co_tiles = (a.Cout + C::CO_T - 1) / C::CO_T;
  const int64_t nblk = (int64_t)a.B * a.D * a.tiles_x * a.tiles_y * a.co_tiles;
  if (nblk > 0x7fffffff) return IPDM_EUNSUPPORTED;
  auto kern = conv_mfma_kernel<NCT, NPT, WCO, WPX, PW, DMAX, KC, KS, FAST, ACT, NORM>;
  static bool attr_set = false;                 // per instantiation; first set by an eager (non-captured) call
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

// true when the FAST instantiation of this tile configuration may be used
template <int NCT, int WCO, int KC>
inline bool fast_ok(const ConvArgs& a) {
  return a.Cin % KC == 0 && a.Cout % (32 * NCT * WCO) == 0;
}

// The (activation, normalisation) combinations the score network uses get FAST instantiations:
//   3x3: (none, -), (ELU, -), (ELU, InstanceNorm++)        1x1: (none, -)
// everything else (ragged channels, other activations, begin_conv's Cin = 1, end_conv's Cout = 1) runs the
// run-time-guarded instantiation of the same tile configuration.
template <int NCT, int NPT, int WCO, int WPX, int PW, int DMAX, int KC, int KS>
int launch_cfg(const ConvArgs& a, hipStream_t s) {
  if (fast_ok<NCT, WCO, KC>(a)) {
    if (a.act == IPDM_ACT_NONE && !a.coef)
      return launch_conv<NCT, NPT, WCO, WPX, PW, DMAX, KC, KS, true, IPDM_ACT_NONE, false>(a, s);
    if constexpr (KS == 3) {
      if (a.act == IPDM_ACT_ELU && !a.coef)
        return launch_conv<NCT, NPT, WCO, WPX, PW, DMAX, KC, KS, true, IPDM_ACT_ELU, false>(a, s);
      if (a.act == IPDM_ACT_ELU && a.coef)
        return launch_conv<NCT, NPT, WCO, WPX, PW, DMAX, KC, KS, true, IPDM_ACT_ELU, true>(a, s);
    }
  }
  return launch_conv<NCT, NPT, WCO, WPX, PW, DMAX, KC, KS, false, 0, false>(a, s);
}

// tile configurations, one translation unit each (conv_cfg_*.hip) to keep compile times parallel
int conv_cfg_32x256(const ConvArgs& a, int ks, hipStream_t s);     // <1,2,1,4,32,1>
int conv_cfg_64x256(const ConvArgs& a, int ks, hipStream_t s);     // <1,4,2,2,32,1>
int conv_cfg_128x256(const ConvArgs& a, int ks, hipStream_t s);    // <4,2,1,4,32,1>
int conv_cfg_64x256_dil(const ConvArgs& a, int ks, hipStream_t s); // <1,4,2,2,32,4>
int conv_cfg_32x128s(const ConvArgs& a, int ks, hipStream_t s);    // <1,1,1,4,16,4>
int conv_cfg_64x128s(const ConvArgs& a, int ks, hipStream_t s);    // <1,2,2,2,16,4>
int conv_cfg_128x128s(const ConvArgs& a, int ks, hipStream_t s);   // <2,2,2,2,16,4>

unsigned long long* conv_debug_stamps();   // conv.hip: buffer set by ipdm_debug_set_stamp_buffer (NULL = off)

// Winograd F(2x2,3x3) path (conv_wino.hip)
bool wino_ok(const ConvArgs& a, int ks);
int conv_wino_launch(ConvArgs a, hipStream_t s);
int conv_wino_weights(const float* w, float* U, int Cout, int Cin, hipStream_t s);

// Winograd F(2x2,3x3) on the bf16 matrix cores with split operands (conv_wino_bx3.hip)
bool wino_bx3_ok(const ConvArgs& a, int ks);
int conv_wino_bx3_launch(ConvArgs a, hipStream_t s);
int conv_wino_bx3_weights(const float* w, void* U, int Cout, int Cin, hipStream_t s);

}  // namespace ipdm_conv
