// upfirdn2d + fused bias/activation for gfx950.
//
// Both ops are HBM-bound (4*(in+out) bytes per upfirdn2d call, 8 bytes/element for bias-act).
// upfirdn2d has three kernels:
//   * upfirdn2d_stream_kernel -- the shapes NCSN++ issues (2x FIR down / up-sampling, 4x4 taps, float4-shaped rows): no LDS,
//     taps in SGPRs, a rolling window of input rows in registers, only 16-byte (8-byte for the up-sampler's window)
//     lane accesses, polyphase tap selection at compile time; 5.2-5.4 TB/s on the 335 MB down-sampling call, 4.9-5.9 TB/s
//     on the up-sampling calls (scripts/bench_resample.py, B = 8; profiles/r02_*)
//   * upfirdn2d_tiled_kernel  -- other (up, down) in {1,2} with <= 4x4 taps: one input tile (+halo) per workgroup in LDS
//   * upfirdn2d_generic_kernel -- anything else, one thread per output with the reference's clamped tap ranges.
//
// Semantics follow op/upfirdn2d.py:168-209 (upfirdn2d_native) / op/upfirdn2d_kernel.cu:49-105 and
// op/fused_bias_act_kernel.cu:19-49 of the reference.
#include "ipdm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

__host__ __device__ __forceinline__ int floor_div_i(int a, int b) {
  int c = a / b;
  if (c * b > a) c--;
  return c;
}

// ---------------------------------------------------------------------------------------------
// T: storage type (float, _Float16, double -- the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// op/upfirdn2d_kernel.cu:311); A: accumulator (float for half / float: one rounding at the store instead of the reference's
// per-tap half arithmetic; double for double)
template <typename T> struct AccOf { typedef float type; };
template <> struct AccOf<double> { typedef double type; };

template <typename T, int UP, int DOWN, int K, int TOH, int TOW>
__global__ __launch_bounds__(256) void upfirdn2d_tiled_kernel(
    const T* __restrict__ in, const T* __restrict__ kernel, T* __restrict__ out,
    int in_h, int in_w, int out_h, int out_w, int kernel_h, int kernel_w, int pad_x0, int pad_y0) {
  typedef typename AccOf<T>::type A;
  constexpr int TIH = ((TOH - 1) * DOWN + K - 1) / UP + 2;
  constexpr int TIW = ((TOW - 1) * DOWN + K - 1) / UP + 2;
  constexpr int NT = (K + UP - 1) / UP;     // taps per axis that can hit a real sample
  __shared__ A sk[K][K];
  __shared__ A sx[TIH][TIW + 1];

  const int plane = blockIdx.x;
  const int oy0 = blockIdx.y * TOH;
  const int ox0 = blockIdx.z * TOW;
  const int tid = threadIdx.x;

  // flipped taps, zero-extended to K x K
  if (tid < K * K) {
    int ky = tid / K, kx = tid % K;
    A v = 0;
    if (ky < kernel_h && kx < kernel_w) v = (A)kernel[(kernel_h - 1 - ky) * kernel_w + (kernel_w - 1 - kx)];
    sk[ky][kx] = v;
  }
  // first input row / column any output of this tile touches
  const int tin_y0 = floor_div_i(oy0 * DOWN + UP - 1 - pad_y0, UP);
  const int tin_x0 = floor_div_i(ox0 * DOWN + UP - 1 - pad_x0, UP);
  const T* src = in + (size_t)plane * in_h * in_w;
  for (int i = tid; i < TIH * TIW; i += 256) {
    int r = i / TIW, c = i - r * TIW;
    int gy = tin_y0 + r, gx = tin_x0 + c;
    A v = 0;
    if (gy >= 0 && gy < in_h && gx >= 0 && gx < in_w) v = (A)src[(size_t)gy * in_w + gx];
    sx[r][c] = v;
  }
  __syncthreads();

  T* dst = out + (size_t)plane * out_h * out_w;
  for (int i = tid; i < TOH * TOW; i += 256) {
    int ty = i / TOW, tx = i - ty * TOW;
    int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= out_h || ox >= out_w) continue;
    int fy = floor_div_i(oy * DOWN + UP - 1 - pad_y0, UP);     // first contributing input row
    int fx = floor_div_i(ox * DOWN + UP - 1 - pad_x0, UP);
    int jy0 = fy * UP - (oy * DOWN - pad_y0);                  // its tap index in the flipped kernel
    int jx0 = fx * UP - (ox * DOWN - pad_x0);
    int ry = fy - tin_y0, rx = fx - tin_x0;
    A acc = 0;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      int jy = jy0 + a * UP;
      if (K % UP != 0 && jy >= K) break;
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        int jx = jx0 + b * UP;
        if (K % UP != 0 && jx >= K) break;
        acc += sx[ry + a][rx + b] * sk[jy][jx];
      }
    }
    dst[(size_t)oy * out_w + ox] = (T)acc;
  }
}

// ---------------------------------------------------------------------------------------------
// MI355X-first FIR resampler for the shapes NCSN++ issues (minor == 1, 4x4 taps zero-extended, rows of whole float4s).
// No LDS: a thread owns OW consecutive output columns and walks RS output rows down the image, keeping a rolling
// window of NR input rows x NW aligned float4s in registers.  Every global access is a 16-byte lane access
// (global_load_dwordx4 / global_store_dwordx4); the window's outer float4s overlap the neighbouring lanes' and are
// served by L1.  The window rotates through NR register slots by phase (no register moves): PH = NR / gcd(NR, ADV)
// row groups are unrolled per loop trip.  Tap -> window-element indices are compile-time, so the zero-inserted
// samples of the up-sampler are never multiplied (polyphase) and everything stays in VGPRs.
//   out[oy][ox] = sum_{ky,kx} upz[oy*DOWN + ky - PY0][ox*DOWN + kx - PX0] * w[ky][kx],  w = flipped kernel,
//   upz[u][v] = in[u/UP][v/UP] where both are multiples of UP, else 0          (op/upfirdn2d.py:168-209)
__host__ __device__ constexpr int cfloor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
__host__ __device__ constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }

template <int UP, int DOWN, int PX0, int PY0, int OW_>
struct FirGeom {
  static constexpr int K = 4;
  static constexpr int OW = OW_;                                                // output columns per thread (whole float4s)
  static constexpr int IB = OW * DOWN / UP;                                     // input columns per thread
  static constexpr int VW = IB % 4 == 0 ? 4 : 2;                                // floats per aligned window load
  static constexpr int F0 = cfloor_div(UP - 1 - PX0, UP);                       // first input column, relative
  static constexpr int L0 = cfloor_div((OW - 1) * DOWN + K - 1 - PX0, UP);      // last input column, relative
  static constexpr int WA = VW * cfloor_div(F0, VW);                            // aligned window start, relative
  static constexpr int NW = (L0 - WA) / VW + 1;                                 // vector loads per window row
  static constexpr int OG = UP;                                                 // output rows per row group
  static constexpr int RF = cfloor_div(UP - 1 - PY0, UP);                       // first / last input row of a group,
  static constexpr int RL = cfloor_div((OG - 1) * DOWN + K - 1 - PY0, UP);      // relative to g*DOWN
  static constexpr int NR = RL - RF + 1;
  static constexpr int ADV = DOWN;                                              // input rows per group
  static constexpr int PH = NR / cgcd(NR, ADV);
  static_assert(IB % 2 == 0 && (OW * DOWN) % UP == 0, "thread base must stay vector-aligned");
  static_assert(OW % 4 == 0 && NR > ADV, "unsupported geometry");
};

template <int UP, int DOWN, int PX0, int PY0, int OW_>
__global__ __launch_bounds__(256) void upfirdn2d_stream_kernel(const float* __restrict__ in,
                                                               const float* __restrict__ kernel,
                                                               float* __restrict__ out, int planes, int in_h, int in_w,
                                                               int out_h, int out_w, int kernel_h, int kernel_w,
                                                               int rows_per_thread) {
  using G = FirGeom<UP, DOWN, PX0, PY0, OW_>;
  constexpr int K = G::K, OW = G::OW, NW = G::NW, VW = G::VW, NR = G::NR, OG = G::OG, ADV = G::ADV, PH = G::PH;
  // flipped, zero-extended taps -> SGPRs (uniform loads)
  float w[K][K];
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int kx = 0; kx < K; ++kx)
    {   // unconditional scalar load from a clamped index, then select (no branch per tap)
      const float t = kernel[max(kernel_h - 1 - ky, 0) * kernel_w + max(kernel_w - 1 - kx, 0)];
      w[ky][kx] = (ky < kernel_h && kx < kernel_w) ? t : 0.f;
    }

  const int nq = (out_w + OW - 1) / OW;                       // column groups per row
  const int groups_total = (out_h + OG - 1) / OG;             // row groups per plane
  const int gps = rows_per_thread;                            // row groups per strip (per thread)
  const int ns = (groups_total + gps - 1) / gps;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;          // the host keeps planes*ns*nq below 2^31
  if (idx >= (unsigned)planes * (unsigned)ns * (unsigned)nq) return;
  const unsigned row_item = idx / (unsigned)nq;
  const int tx = (int)(idx - row_item * (unsigned)nq);
  const int plane = (int)(row_item / (unsigned)ns);
  const int strip = (int)(row_item - (unsigned)plane * (unsigned)ns);
  const float* src = in + (size_t)plane * in_h * in_w;
  float* dst = out + (size_t)plane * out_h * out_w;
  const int g0 = strip * gps;
  const int g1 = min(groups_total, g0 + gps);
  const int col0 = G::IB * tx + G::WA;                        // first column of the aligned window (multiple of VW)

  float win[NR][NW * VW];
  auto load_row = [&](int slot, int row) {
    // branch-free: always load from a clamped (valid) address, then zero what lies outside the image
    // (bit mask instead of a select: hipcc turns `ok ? load : 0` back into a branch around the load, which
    //  serialises the row's loads behind s_waitcnt)
    const bool rok = row >= 0 && row < in_h;
    const float* rp = src + (size_t)min(max(row, 0), in_h - 1) * in_w;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int c = col0 + VW * q;
      const unsigned m = (rok && c >= 0 && c + VW - 1 < in_w) ? 0xffffffffu : 0u;
      const float* ap = rp + min(max(c, 0), in_w - VW);
      if constexpr (VW == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(ap);
        win[slot][4 * q + 0] = __uint_as_float(v.x & m);
        win[slot][4 * q + 1] = __uint_as_float(v.y & m);
        win[slot][4 * q + 2] = __uint_as_float(v.z & m);
        win[slot][4 * q + 3] = __uint_as_float(v.w & m);
      } else {
        const uint2 v = *reinterpret_cast<const uint2*>(ap);
        win[slot][2 * q + 0] = __uint_as_float(v.x & m);
        win[slot][2 * q + 1] = __uint_as_float(v.y & m);
      }
    }
  };
  // window rows RF .. RL of group g0
#pragma unroll
  for (int r = 0; r < NR; ++r) load_row(r, g0 * ADV + G::RF + r);

  for (int g = g0; g < g1; g += PH) {
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
      if (g + ph < g1) {
        // logical window row i of this group lives in slot (i + ph*ADV) % NR
#pragma unroll
        for (int p = 0; p < OG; ++p) {
          const int oy = (g + ph) * OG + p;
          float acc[OW];
#pragma unroll
          for (int j = 0; j < OW; ++j) acc[j] = 0.f;
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            const int uy = p * DOWN + ky - PY0;               // compile-time after unrolling
            if (((uy % UP) + UP) % UP != 0) continue;
            const int ri = cfloor_div(uy, UP) - G::RF;        // logical window row
            const int slot = (ri + ph * ADV) % NR;
#pragma unroll
            for (int j = 0; j < OW; ++j) {
#pragma unroll
              for (int kx = 0; kx < K; ++kx) {
                const int ux = j * DOWN + kx - PX0;
                if (((ux % UP) + UP) % UP != 0) continue;
                const int wi = cfloor_div(ux, UP) - G::WA;     // index into the window row
                acc[j] = fmaf(win[slot][wi], w[ky][kx], acc[j]);
              }
            }
          }
          if (oy < out_h) {
#pragma unroll
            for (int j4 = 0; j4 < OW / 4; ++j4) {
              const int ox = OW * tx + 4 * j4;
              if (ox + 3 < out_w) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v o4 = {acc[4 * j4], acc[4 * j4 + 1], acc[4 * j4 + 2], acc[4 * j4 + 3]};
                f4v* op = reinterpret_cast<f4v*>(dst + (size_t)oy * out_w + ox);
                *op = o4;
              }
            }
          }
        }
        // slide: the ADV oldest logical rows are replaced by the next group's newest rows
        if (g + ph + 1 < g1) {
#pragma unroll
          for (int a = 0; a < ADV; ++a) {
            const int slot = (a + ph * ADV) % NR;              // logical row a of this group = oldest
            load_row(slot, (g + ph + 1) * ADV + G::RF + (NR - ADV) + a);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
struct UpfirdnParams {
  int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
  int major, in_h, in_w, minor, kernel_h, kernel_w, out_h, out_w;
};

template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_generic_kernel(const T* __restrict__ in, const T* __restrict__ kernel,
                                                                T* __restrict__ out, UpfirdnParams p) {
  typedef typename AccOf<T>::type A;
  const int64_t total = (int64_t)p.major * p.out_h * p.out_w * p.minor;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int mi = (int)(idx % p.minor);
    int64_t t = idx / p.minor;
    int ox = (int)(t % p.out_w);
    t /= p.out_w;
    int oy = (int)(t % p.out_h);
    int major = (int)(t / p.out_h);

    int mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
    int fy = floor_div_i(mid_y, p.up_y);
    int ky0 = mid_y + p.kernel_h - (fy + 1) * p.up_y;           // unflipped tap index of row fy
    int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0;
    int fx = floor_div_i(mid_x, p.up_x);
    int kx0 = mid_x + p.kernel_w - (fx + 1) * p.up_x;

    A acc = 0;
    for (int a = 0, ky = ky0; ky >= 0; ++a, ky -= p.up_y) {
      int gy = fy + a;
      if (gy < 0 || ky >= p.kernel_h) continue;
      if (gy >= p.in_h) break;
      for (int b = 0, kx = kx0; kx >= 0; ++b, kx -= p.up_x) {
        int gx = fx + b;
        if (gx < 0 || kx >= p.kernel_w) continue;
        if (gx >= p.in_w) break;
        acc += (A)in[(((size_t)major * p.in_h + gy) * p.in_w + gx) * p.minor + mi] * (A)kernel[ky * p.kernel_w + kx];
      }
    }
    out[idx] = (T)acc;
  }
}

template <typename T, int UP, int DOWN, int K, int TOH, int TOW>
int launch_tiled(const T* in, const T* kernel, T* out, const UpfirdnParams& p, hipStream_t s) {
  dim3 grid(p.major, (p.out_h + TOH - 1) / TOH, (p.out_w + TOW - 1) / TOW);
  hipLaunchKernelGGL((upfirdn2d_tiled_kernel<T, UP, DOWN, K, TOH, TOW>), grid, dim3(256), 0, s, in, kernel, out,
                     p.in_h, p.in_w, p.out_h, p.out_w, p.kernel_h, p.kernel_w, p.pad_x0, p.pad_y0);
  return ipdm_launch_status();
}

template <int UP, int DOWN, int PX0, int PY0, int OW>
int launch_stream(const float* in, const float* kernel, float* out, const UpfirdnParams& p, hipStream_t s) {
  using G = FirGeom<UP, DOWN, PX0, PY0, OW>;
  const long long nq = (p.out_w + G::OW - 1) / G::OW;
  const long long groups = (p.out_h + G::OG - 1) / G::OG;
  // row groups per thread.  Measured on MI355X (scripts/bench_resample.py, B = 8): what a launch wants is ~2^20 vector loads
  // in flight chip-wide -- long strips (few threads, each streaming many rows: the (NR - ADV)-row halo is amortised and the
  // window loads of a group are all independent) until the thread count drops below that; 128 K threads for the
  // down-sampler (8 loads per group), 350 K for the up-sampler (3).  IPDM_FIR_GPS overrides (tuning).
  static const int tune_gps = getenv("IPDM_FIR_GPS") ? atoi(getenv("IPDM_FIR_GPS")) : 0;
  long long gps = (long long)p.major * groups * nq * (G::ADV * G::NW) >> 20;
  gps = gps < 1 ? 1 : (gps > 32 ? 32 : gps);
  if (tune_gps > 0) gps = tune_gps;
  const long long ns = (groups + gps - 1) / gps;
  const long long threads = (long long)p.major * ns * nq;
  if (threads >= (1LL << 31)) return IPDM_EUNSUPPORTED;         // caller falls through to the tiled kernel
  hipLaunchKernelGGL((upfirdn2d_stream_kernel<UP, DOWN, PX0, PY0, OW>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                     in, kernel, out, p.major, p.in_h, p.in_w, p.out_h, p.out_w, p.kernel_h, p.kernel_w, (int)gps);
  return ipdm_launch_status();
}

// ---------------------------------------------------------------------------------------------
// scalar form for the other storage types of the reference's dispatch (half, double; op/fused_bias_act_kernel.cu:79);
// arithmetic in float (half) / double
template <typename T>
__global__ __launch_bounds__(256) void bias_act_any_kernel(const T* __restrict__ x, const T* __restrict__ b,
                                                           const T* __restrict__ ref, T* __restrict__ y, int64_t n,
                                                           int step_b, int size_b, int code, float alpha, float scale) {
  typedef typename AccOf<T>::type A;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    A t = (A)x[i];
    if (b) t += (A)b[(i / step_b) % size_b];
    A o;
    switch (code) {
      case 30: o = t > (A)0 ? t : t * (A)alpha; break;
      case 31: o = (ref ? (A)ref[i] : (A)0) > (A)0 ? t : t * (A)alpha; break;
      case 12:
      case 32: o = (A)0; break;
      default: o = t; break;
    }
    y[i] = (T)(o * (A)scale);
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                       const float* __restrict__ ref, float* __restrict__ y,
                                                       int64_t n, int step_b, int size_b, int code, float alpha,
                                                       float scale) {
  constexpr int V = VEC ? 4 : 1;
  const int64_t nv = n / V;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float v[V], r[V];
    if constexpr (VEC) {
      float4 t = reinterpret_cast<const float4*>(x)[i];
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      if (ref) {
        float4 q = reinterpret_cast<const float4*>(ref)[i];
        r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
      }
    } else {
      v[0] = x[i];
      if (ref) r[0] = ref[i];
    }
    if (b) {
      float bb = b[((i * V) / step_b) % size_b];   // VEC requires step_b % 4 == 0: one bias per vector
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] += bb;
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float t = v[j], o;
      switch (code) {
        case 30: o = t > 0.f ? t : t * alpha; break;
        case 31: o = (ref ? r[j] : 0.f) > 0.f ? t : t * alpha; break;
        case 12:
        case 32: o = 0.f; break;
        default: o = t; break;
      }
      v[j] = o * scale;
    }
    if constexpr (VEC) reinterpret_cast<float4*>(y)[i] = make_float4(v[0], v[1], v[2], v[3]);
    else y[i] = v[0];
  }
}

}  // namespace

template <typename T>
static int upfirdn2d_entry(const T* in, const T* kernel, T* out, int major, int in_h, int in_w, int minor, int kernel_h,
                           int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                           int pad_y1, void* stream) {
  IPDM_REQUIRE(in && kernel && out);
  IPDM_REQUIRE(major >= 0 && in_h > 0 && in_w > 0 && minor > 0 && kernel_h > 0 && kernel_w > 0);
  IPDM_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0);
  UpfirdnParams p;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
  p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kernel_h = kernel_h; p.kernel_w = kernel_w;
  p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h) / down_y + 1;
  p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w) / down_x + 1;
  IPDM_REQUIRE(p.out_h > 0 && p.out_w > 0);
  if (major == 0) return IPDM_OK;
  hipStream_t s = ipdm_stream(stream);
  const bool sq = up_x == up_y && down_x == down_y && minor == 1 && kernel_h <= 4 && kernel_w <= 4;
  if constexpr (std::is_same<T, float>::value) {
    // the NCSN++ resampling calls (models/up_or_down_sampling.py:191-257: down2 pad (1,1), up2 pad (2,1) with [1,3,3,1]
    // taps) on float4-shaped rows -> register-streaming kernel
    const bool vec_ok = sq && in_w % 4 == 0 && p.out_w % 4 == 0 && pad_x0 == pad_y0 &&
                        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    int rc = IPDM_EUNSUPPORTED;
    // (the up-sampler writes 4x what it reads: one float4 of outputs per thread and row keeps every store instruction a
    //  contiguous KiB per wave -- 4.9 vs 3.6 TB/s at 128^2 -> 256^2 against two float4s per thread)
    if (vec_ok && up_x == 1 && down_x == 2 && pad_x0 == 1) rc = launch_stream<1, 2, 1, 1, 4>(in, kernel, out, p, s);
    if (vec_ok && up_x == 2 && down_x == 1 && pad_x0 == 2) rc = launch_stream<2, 1, 2, 2, 4>(in, kernel, out, p, s);
    if (rc != IPDM_EUNSUPPORTED) return rc;
  }
  if (sq && up_x == 1 && down_x == 2) return launch_tiled<T, 1, 2, 4, 16, 64>(in, kernel, out, p, s);
  if (sq && up_x == 2 && down_x == 1) return launch_tiled<T, 2, 1, 4, 32, 64>(in, kernel, out, p, s);
  if (sq && up_x == 1 && down_x == 1) return launch_tiled<T, 1, 1, 4, 32, 64>(in, kernel, out, p, s);
  const int64_t total = (int64_t)major * p.out_h * p.out_w * minor;
  hipLaunchKernelGGL(upfirdn2d_generic_kernel<T>, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, s, in, kernel, out, p);
  return ipdm_launch_status();
}

extern "C" int ipdm_upfirdn2d_f32(const float* in, const float* kernel, float* out, int major, int in_h, int in_w,
                                  int minor, int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y,
                                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
  return upfirdn2d_entry<float>(in, kernel, out, major, in_h, in_w, minor, kernel_h, kernel_w, up_x, up_y, down_x, down_y,
                                pad_x0, pad_x1, pad_y0, pad_y1, stream);
}

/* IEEE half storage (in / kernel / out: 2-byte elements), fp32 arithmetic, one rounding at the store */
extern "C" int ipdm_upfirdn2d_f16(const void* in, const void* kernel, void* out, int major, int in_h, int in_w,
                                  int minor, int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y,
                                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
  return upfirdn2d_entry<_Float16>(static_cast<const _Float16*>(in), static_cast<const _Float16*>(kernel),
                                   static_cast<_Float16*>(out), major, in_h, in_w, minor, kernel_h, kernel_w, up_x, up_y,
                                   down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, stream);
}

extern "C" int ipdm_upfirdn2d_f64(const double* in, const double* kernel, double* out, int major, int in_h, int in_w,
                                  int minor, int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y,
                                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
  return upfirdn2d_entry<double>(in, kernel, out, major, in_h, in_w, minor, kernel_h, kernel_w, up_x, up_y, down_x, down_y,
                                 pad_x0, pad_x1, pad_y0, pad_y1, stream);
}

template <typename T>
static int bias_act_any_entry(const T* x, const T* b, const T* ref, T* y, int64_t n, int step_b, int size_b, int act, int grad,
                              float alpha, float scale, void* stream) {
  IPDM_REQUIRE(n >= 0);
  if (n == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y);
  if (size_b <= 0) b = nullptr;
  if (b) IPDM_REQUIRE(step_b > 0);
  hipLaunchKernelGGL(bias_act_any_kernel<T>, dim3(ipdm_ew_grid(n, 256)), dim3(256), 0, ipdm_stream(stream), x, b, ref, y, n,
                     step_b, size_b, act * 10 + grad, alpha, scale);
  return ipdm_launch_status();
}

extern "C" int ipdm_fused_bias_act_f16(const void* x, const void* b, const void* ref, void* y, int64_t n, int step_b,
                                       int size_b, int act, int grad, float alpha, float scale, void* stream) {
  return bias_act_any_entry<_Float16>(static_cast<const _Float16*>(x), static_cast<const _Float16*>(b),
                                      static_cast<const _Float16*>(ref), static_cast<_Float16*>(y), n, step_b, size_b, act,
                                      grad, alpha, scale, stream);
}

extern "C" int ipdm_fused_bias_act_f64(const double* x, const double* b, const double* ref, double* y, int64_t n, int step_b,
                                       int size_b, int act, int grad, float alpha, float scale, void* stream) {
  return bias_act_any_entry<double>(x, b, ref, y, n, step_b, size_b, act, grad, alpha, scale, stream);
}

extern "C" int ipdm_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* y, int64_t n,
                                       int step_b, int size_b, int act, int grad, float alpha, float scale,
                                       void* stream) {
  IPDM_REQUIRE(n >= 0);
  if (n == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y);
  if (size_b <= 0) b = nullptr;
  if (b) IPDM_REQUIRE(step_b > 0);
  const int code = act * 10 + grad;
  hipStream_t s = ipdm_stream(stream);
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(ref)) & 15) == 0;
  const bool vec = aligned && (n % 4 == 0) && (!b || step_b % 4 == 0);
  if (vec)
    hipLaunchKernelGGL(bias_act_kernel<true>, dim3(ipdm_ew_grid(n / 4, 256)), dim3(256), 0, s, x, b, ref, y, n, step_b,
                       size_b, code, alpha, scale);
  else
    hipLaunchKernelGGL(bias_act_kernel<false>, dim3(ipdm_ew_grid(n, 256)), dim3(256), 0, s, x, b, ref, y, n, step_b,
                       size_b, code, alpha, scale);
  return ipdm_launch_status();
}
