// 3x3 'same' convolutions with very few input OR very few output channels: the first and the last layer of every score
// network (begin_conv 1|2|3 -> ngf, end_conv ngf -> 1|2|3; reference ncsn/models/ncsnv2.py:40,45 and
// models/ncsnpp.py:143,226).  On the matrix-core kernels these two layers pad their thin side to a 16- or 32-wide MFMA
// operand and run at ~6 TFLOP/s; they are pure streaming problems (write, resp. read, ONE ngf-channel tensor), so
// they are written as such: fp32 FMA chains on the vector ALU, 16-byte accesses on the wide tensor, the thin tensor
// and the weights through the caches (weights: wave-uniform addresses -> scalar loads).  fp32 throughout, fixed summation
// order (channel-major, taps row-major): a sample's result does not depend on the batch it is computed in.
#include "ipdm_common.h"

namespace {

// ---- few input channels: out[b][co][y][4q..4q+3], one thread per (b, y, quad) and CO_BLK output channels -------------------
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_fewin_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ coef,
                                                            float* __restrict__ out, int B, int Cout, int H, int W,
                                                            int co_blk) {
  const int Q = W >> 2;
  const long long items = (long long)B * H * Q;
  const int co0 = blockIdx.y * co_blk;
  const int co1 = min(Cout, co0 + co_blk);
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
    const int q = (int)(it % Q);
    const long long t = it / Q;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float in[CIN][3][6];                               // columns 4q-1 .. 4q+4 of rows y-1 .. y+1
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      // optional input affine (x - c0) * c1 + c2 on the pixels INSIDE the image (the padding stays zero), as the fused
      // input normalisation of the matrix-core kernels: the score networks fold h = 2x - 1 into their first layer
      float c0 = 0.f, c1 = 1.f, c2 = 0.f;
      if (coef) {
        const float* cf = coef + ((long long)b * CIN + ci) * 3;
        c0 = cf[0]; c1 = cf[1]; c2 = cf[2];
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        const bool row_ok = yy >= 0 && yy < H;
        const float* rp = x + (((long long)b * CIN + ci) * H + (row_ok ? yy : 0)) * W + 4 * q;
        const float4 m = *reinterpret_cast<const float4*>(rp);
        const float l = rp[q > 0 ? -1 : 0], rr = rp[q < Q - 1 ? 4 : 3];
        in[ci][r][0] = (row_ok && q > 0) ? (l - c0) * c1 + c2 : 0.f;
        in[ci][r][1] = row_ok ? (m.x - c0) * c1 + c2 : 0.f;
        in[ci][r][2] = row_ok ? (m.y - c0) * c1 + c2 : 0.f;
        in[ci][r][3] = row_ok ? (m.z - c0) * c1 + c2 : 0.f;
        in[ci][r][4] = row_ok ? (m.w - c0) * c1 + c2 : 0.f;
        in[ci][r][5] = (row_ok && q < Q - 1) ? (rr - c0) * c1 + c2 : 0.f;
      }
    }
    float* op = out + (((long long)b * Cout + co0) * H + y) * W + 4 * q;
    for (int co = co0; co < co1; ++co, op += (long long)H * W) {
      const float* wp = w + (long long)co * CIN * 9;   // wave-uniform: scalar loads
      const float bv = bias ? bias[co] : 0.f;
      float a0 = bv, a1 = bv, a2 = bv, a3 = bv;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float wv = wp[(ci * 3 + r) * 3 + c];
            a0 = fmaf(wv, in[ci][r][c], a0);
            a1 = fmaf(wv, in[ci][r][c + 1], a1);
            a2 = fmaf(wv, in[ci][r][c + 2], a2);
            a3 = fmaf(wv, in[ci][r][c + 3], a3);
          }
      *reinterpret_cast<float4*>(op) = make_float4(a0, a1, a2, a3);
    }
  }
}

// ---- few output channels: one thread per (b, y, quad) walks all input channels ---------------------------------------------
// SHFL (W / 4 divides 64): the columns left and right of a thread's quad are the neighbouring lanes' own elements
// (cross-lane moves instead of six more 4-byte loads per channel: the address path, not HBM, was the limit).
// The image border costs nothing inside the channel loop: rows above / below the image are read from a valid row and
// accumulate into their own partial sums (top / middle / bottom row of taps), as do the two halo columns; what lies outside
// the image is dropped when the partial sums are added at the end (fixed order: middle, top, bottom, left, right).
template <int COUT, bool SHFL>
__global__ __launch_bounds__(256) void conv3x3_fewout_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int B, int Cin, int H, int W) {
  // a workgroup = 64 items x 4 channel groups (wave g takes channels g, g+4, ...: four times the waves in flight of one
  // thread per item -- 1.75 waves per SIMD left the loads' latency exposed); the four partial sums meet in LDS, added in
  // group order
  __shared__ float red[3][COUT][4][64];
  const int Q = W >> 2;
  const long long items = (long long)B * H * Q;
  const long long HW = (long long)H * W;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (long long it0 = (long long)blockIdx.x * 64; it0 < items; it0 += (long long)gridDim.x * 64) {
    const long long it = it0 + lane < items ? it0 + lane : items - 1;       // a tail lane repeats the last item (not stored)
    const bool live = it0 + lane < items;
    const int q = (int)(it % Q);
    const long long t = it / Q;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    // acc[co][r][j]: taps of kernel row r on the thread's own four columns; hl / hr [co][r]: the halo columns' taps
    float acc[COUT][3][4], hl[COUT][3], hr[COUT][3];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        hl[co][r] = 0.f; hr[co][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[co][r][j] = 0.f;
      }
    const bool up = y > 0, dn = y < H - 1, lf = q > 0, rt = q < Q - 1;
    const float* base = x + ((long long)b * Cin * H + y) * W + 4 * q;
    const int ro[3] = {up ? -W : 0, 0, dn ? W : 0};
    // four channels per trip, unrolled by hand: the cross-lane moves are convergent operations, which keep hipcc from
    // unrolling a loop of run-time length itself (it warned "loop not unrolled"), and one channel per trip leaves three
    // loads in flight per lane
    for (int ci0 = grp; ci0 < Cin; ci0 += 16)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ci = ci0 + 4 * u;
      if (ci >= Cin) break;                                   // wave-uniform
      const float* cp = base + ci * HW;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float* rp = cp + ro[r];
        const float4 m = *reinterpret_cast<const float4*>(rp);
        float l, rr;
        if constexpr (SHFL) {
          l = __shfl_up(m.w, 1, 64);
          rr = __shfl_down(m.x, 1, 64);
        } else {
          l = rp[lf ? -1 : 0];
          rr = rp[rt ? 4 : 3];
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float* wp = w + ((long long)co * Cin + ci) * 9 + r * 3;   // wave-uniform: scalar loads
          const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
          // output column j takes w0 * in[j-1] + w1 * in[j] + w2 * in[j+1]
          hl[co][r] = fmaf(w0, l, hl[co][r]);
          acc[co][r][0] = fmaf(w1, m.x, acc[co][r][0]);
          acc[co][r][0] = fmaf(w2, m.y, acc[co][r][0]);
          acc[co][r][1] = fmaf(w0, m.x, acc[co][r][1]);
          acc[co][r][1] = fmaf(w1, m.y, acc[co][r][1]);
          acc[co][r][1] = fmaf(w2, m.z, acc[co][r][1]);
          acc[co][r][2] = fmaf(w0, m.y, acc[co][r][2]);
          acc[co][r][2] = fmaf(w1, m.z, acc[co][r][2]);
          acc[co][r][2] = fmaf(w2, m.w, acc[co][r][2]);
          acc[co][r][3] = fmaf(w0, m.z, acc[co][r][3]);
          acc[co][r][3] = fmaf(w1, m.w, acc[co][r][3]);
          hr[co][r] = fmaf(w2, rr, hr[co][r]);
        }
      }
    }
    float o[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[co][1][j];
        if (up) v += acc[co][0][j];
        if (dn) v += acc[co][2][j];
        o[co][j] = v;
      }
      float left = hl[co][1], right = hr[co][1];
      if (up) { left += hl[co][0]; right += hr[co][0]; }
      if (dn) { left += hl[co][2]; right += hr[co][2]; }
      if (lf) o[co][0] += left;
      if (rt) o[co][3] += right;
    }
    if (grp > 0) {
#pragma unroll
      for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[grp - 1][co][j][lane] = o[co][j];
    }
    __syncthreads();
    if (grp == 0 && live) {
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        const float bv = bias ? bias[co] : 0.f;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = bv + (((o[co][j] + red[0][co][j][lane]) + red[1][co][j][lane]) + red[2][co][j][lane]);
        *reinterpret_cast<float4*>(out + (((long long)b * COUT + co) * H + y) * W + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    __syncthreads();                                          // red is rewritten by the next round
  }
}

}  // namespace

extern "C" int ipdm_conv3x3_thin_supported(int Cin, int Cout, int H, int W) {
  return (W % 4 == 0 && H > 0 && ((Cin >= 1 && Cin <= 3 && Cout >= 1) || (Cout >= 1 && Cout <= 3 && Cin >= 1))) ? 1 : 0;
}

extern "C" int ipdm_conv3x3_thin_f32(const float* x, const float* w, const float* bias, const float* coef, float* out, int B,
                                     int Cin, int Cout, int H, int W, void* stream) {
  IPDM_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0);
  if (!ipdm_conv3x3_thin_supported(Cin, Cout, H, W)) return IPDM_EUNSUPPORTED;
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && w && out && x != out);
  IPDM_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
  hipStream_t s = ipdm_stream(stream);
  const long long items = (long long)B * H * (W / 4);
  const int gx = ipdm_ew_grid(items, 256);
  if (Cin <= 3 && Cin <= Cout) {
    // enough (item block, channel block) pairs to fill the chip several times over, at least 8 channels per thread
    int co_blk = Cout;
    while (co_blk > 8 && (long long)gx * ((Cout + co_blk - 1) / co_blk) < 2048) co_blk = (co_blk + 1) / 2;
    const dim3 grid(gx, (Cout + co_blk - 1) / co_blk);
    if (Cin == 1) hipLaunchKernelGGL(conv3x3_fewin_kernel<1>, grid, dim3(256), 0, s, x, w, bias, coef, out, B, Cout, H, W, co_blk);
    else if (Cin == 2) hipLaunchKernelGGL(conv3x3_fewin_kernel<2>, grid, dim3(256), 0, s, x, w, bias, coef, out, B, Cout, H, W, co_blk);
    else hipLaunchKernelGGL(conv3x3_fewin_kernel<3>, grid, dim3(256), 0, s, x, w, bias, coef, out, B, Cout, H, W, co_blk);
  } else {
    if (coef) return IPDM_EUNSUPPORTED;          // an input affine on the wide side belongs to the producing layer
    const int Q = W / 4;
    const int gx = ipdm_ew_grid(items, 64);       // 64 items per workgroup (four channel groups each)
    const bool shfl = Q <= 64 && 64 % Q == 0;     // a row of quads never straddles a wave (items are dealt row-major)
#define IPDM_FEWOUT(CO)                                                                                                     \
  if (shfl) hipLaunchKernelGGL((conv3x3_fewout_kernel<CO, true>), dim3(gx), dim3(256), 0, s, x, w, bias, out, B, Cin, H, W); \
  else hipLaunchKernelGGL((conv3x3_fewout_kernel<CO, false>), dim3(gx), dim3(256), 0, s, x, w, bias, out, B, Cin, H, W)
    if (Cout == 1) { IPDM_FEWOUT(1); }
    else if (Cout == 2) { IPDM_FEWOUT(2); }
    else { IPDM_FEWOUT(3); }
#undef IPDM_FEWOUT
  }
  return ipdm_launch_status();
}
