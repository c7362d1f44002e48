// k-space side of the ALD step on gfx950: centred orthonormal 2-D FFT held entirely in one CU's LDS
// (128x128 complex64 = 128 KiB of the 160 KiB), multi-coil SENSE forward / adjoint / SSOS, the
// closed form of the reference's one-SGD-step L2Penalty proximal, and the fused Langevin + proximal
// iteration tail.  One 1024-thread workgroup owns one image; coils are looped inside the workgroup so
// the coil sum is deterministic (no atomics).  All of this is launch/latency-bound work (~1 MiB of
// algorithmic traffic per sample per step) that rides beside the score network.
//
// The fftshift/ifftshift pairs of i2k_complex / k2i_complex (ncsn/linear_transforms/__init__.py:36-57)
// are folded into (-1)^(r+c) sign flips before and after an ordinary FFT (exact for sizes % 4 == 0);
// sizes the LDS path cannot take go through a direct centred DFT with an exact integer phase index.
#include "kspace_fft.h"

namespace {

using namespace ipdm_kspace;

#define FFT_LDS_SETUP(H, W)                                   \
  extern __shared__ __align__(16) unsigned char smem_raw[];  \
  FftLds L;                                                   \
  L.buf = reinterpret_cast<float2*>(smem_raw);               \
  L.tw = L.buf + (size_t)(H) * (W);                          \
  L.twN = (H) > (W) ? (H) : (W);                             \
  fft_make_twiddles(L);

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FFT_THREADS) void fft2c_lds_kernel(const float2* in, float2* out,
                                                                int H, int W, int inverse) {
  FFT_LDS_SETUP(H, W)
  const int HW = H * W;
  const float2* src = in + (size_t)blockIdx.x * HW;
  float2* dst = out + (size_t)blockIdx.x * HW;
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    int r = e / W, c = e - r * W;
    float s = sign_rc(r, c);
    float2 v = src[e];
    L.buf[e] = make_float2(v.x * s, v.y * s);
  }
  __syncthreads();
  fft2_lds(L, H, W, inverse != 0);
  const float scale = rsqrtf((float)HW);
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    int r = e / W, c = e - r * W;
    float s = sign_rc(r, c) * scale;
    float2 v = L.buf[e];
    dst[e] = make_float2(v.x * s, v.y * s);
  }
}

// direct centred DFT along the last axis, output transposed: in [batch][R][N] -> out [batch][N][R]
__global__ __launch_bounds__(256) void dft_rows_transposed_kernel(const float2* __restrict__ in,
                                                                  float2* __restrict__ out, int R, int N, int inverse) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float2* row = reinterpret_cast<float2*>(smem_raw);
  float2* tw = row + N;
  const int b = blockIdx.y, r = blockIdx.x;
  const float2* src = in + ((size_t)b * R + r) * N;
  const double sgn = inverse ? 2.0 : -2.0;
  const float scale = (float)(1.0 / sqrt((double)N));
  for (int q = threadIdx.x; q < N; q += blockDim.x) {
    double s, c;
    sincospi(sgn * (double)q / (double)N, &s, &c);
    tw[q] = make_float2((float)c * scale, (float)s * scale);
    row[q] = src[q];
  }
  __syncthreads();
  const int cshift = N / 2;
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    int dk = ((k - cshift) % N + N) % N;
    int p = (int)(((int64_t)(N - cshift) * dk) % N);     // (m - c)(k - c) mod N at m = 0
    float2 acc = make_float2(0.f, 0.f);
    for (int m = 0; m < N; ++m) {
      float2 w = tw[p], x = row[m];
      acc.x += x.x * w.x - x.y * w.y;
      acc.y += x.x * w.y + x.y * w.x;
      p += dk;
      if (p >= N) p -= N;
    }
    out[((size_t)b * N + k) * R + r] = acc;
  }
}

// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(FFT_THREADS) void sense_forward_kernel(const float2* __restrict__ x,
                                                                    const float* __restrict__ sens,
                                                                    const uint8_t* __restrict__ mask, int mask_t,
                                                                    float2* __restrict__ y, int B, int H, int W) {
  FFT_LDS_SETUP(H, W)
  const int HW = H * W;
  const int b = blockIdx.x, coil = blockIdx.y;
  const float2* src = x + (size_t)b * HW;
  const float* sm = sens ? sens + (size_t)coil * HW : nullptr;      // NULL: single coil, S = 1
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    int r = e / W, c = e - r * W;
    float s = sm ? sign_rc(r, c) * sm[e] : sign_rc(r, c);
    float2 v = src[e];
    L.buf[e] = make_float2(v.x * s, v.y * s);
  }
  __syncthreads();
  fft2_lds(L, H, W, false);
  const float scale = rsqrtf((float)HW);
  float2* dst = y + ((size_t)coil * B + b) * HW;
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    int r = e / W, c = e - r * W;
    float2 v = L.buf[e];
    float s = mask_at(mask, mask_t, b, W, c) ? sign_rc(r, c) * scale : 0.f;
    dst[e] = make_float2(v.x * s, v.y * s);
  }
}

// s [n_coils][B][H][W] -> out[b] = sum_c S_c * ifft2c(s[c][b])  (or root-sum-of-squares for SSOS).
// The coil sum is accumulated in the (L2-resident) output image, coil by coil in index order, like the
// reference's `X_out += ...` loop: no accumulator registers live across the FFT, deterministic.
template <bool SSOS>
__global__ __launch_bounds__(FFT_THREADS) void sense_adjoint_kernel(const float2* __restrict__ s,
                                                                    const float* __restrict__ sens,
                                                                    const uint8_t* __restrict__ mask, int mask_t,
                                                                    int apply_mask, float* out, int B,
                                                                    int n_coils, int H, int W) {
  FFT_LDS_SETUP(H, W)
  const int HW = H * W;
  const int b = blockIdx.x;
  const float scale = rsqrtf((float)HW);
  for (int coil = 0; coil < n_coils; ++coil) {
    const float2* src = s + ((size_t)coil * B + b) * HW;
    for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
      int r = e / W, c = e - r * W;
      float sg = sign_rc(r, c);
      if (apply_mask && !mask_at(mask, mask_t, b, W, c)) sg = 0.f;
      float2 v = src[e];
      L.buf[e] = make_float2(v.x * sg, v.y * sg);
    }
    __syncthreads();
    fft2_lds(L, H, W, true);
    const bool last = coil == n_coils - 1;
    for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
      float2 v = L.buf[e];
      size_t gi = (size_t)b * HW + e;
      if constexpr (SSOS) {
        float a = (v.x * v.x + v.y * v.y) * (scale * scale);
        if (coil > 0) a += out[gi];
        out[gi] = last ? sqrtf(a) : a;
      } else {
        int r = e / W, c = e - r * W;
        float w = sign_rc(r, c) * scale * sens[(size_t)coil * HW + e];
        float2 a = make_float2(v.x * w, v.y * w);
        float2* o = reinterpret_cast<float2*>(out) + gi;
        if (coil > 0) {
          float2 prev = *o;
          a.x += prev.x;
          a.y += prev.y;
        }
        *o = a;
      }
    }
    __syncthreads();
  }
}

// Langevin update of one sample's two planes, in place (the first phase of the fused iteration tails):
//   x += step*g + noise_scale*n, n injected (n_re/n_im) or Philox keyed by (seed, global sample id, step, plane)
__device__ __forceinline__ void langevin_phase(float* xr, float* xi, const float* __restrict__ g_re,
                                               const float* __restrict__ g_im, const float* __restrict__ n_re,
                                               const float* __restrict__ n_im, float step, float noise_scale,
                                               uint64_t seed, int64_t sample_offset, int64_t step_id, int b, int HW) {
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    size_t gi = (size_t)b * HW + e;
    float nr, ni;
    if (n_re) {
      nr = n_re[gi];
      ni = n_im[gi];
    } else {
      float q[4];
      const int lane4 = e & 3;
      ipdm_philox_normal4(seed, sample_offset + b, step_id, 0, (uint32_t)(e >> 2), q);
      nr = lane4 == 0 ? q[0] : lane4 == 1 ? q[1] : lane4 == 2 ? q[2] : q[3];
      ipdm_philox_normal4(seed, sample_offset + b, step_id, 1, (uint32_t)(e >> 2), q);
      ni = lane4 == 0 ? q[0] : lane4 == 1 ? q[1] : lane4 == 2 ? q[2] : q[3];
    }
    xr[e] = xr[e] + step * g_re[gi] + nr * noise_scale;
    xi[e] = xi[e] + step * g_im[gi] + ni * noise_scale;
  }
}

// Langevin update (optional) + L2Penalty closed form, planar real/imag, in place.
//   phase 0: z = x + step*g + noise_scale*n           -> stored back to x (global, L2-resident)
//   per coil: LDS = S_c z ; FFT ; residual on sampled columns ; IFFT ; work += S_c * (.)
//   final:   x = z - coef * work
// `work` ([B][H][W] c64) carries the coil sum so that no accumulator registers live across the FFTs.
template <bool LANGEVIN>
__global__ __launch_bounds__(FFT_THREADS) void ald_sense_step_kernel(
    float* x_re, float* x_im, const float* __restrict__ g_re, const float* __restrict__ g_im,
    const float* __restrict__ n_re, const float* __restrict__ n_im, float step, float noise_scale, uint64_t seed,
    int64_t sample_offset, int64_t step_id, const ipdm_sched_t* __restrict__ sched, const float2* __restrict__ y,
    const float* __restrict__ sens, const uint8_t* __restrict__ mask, int mask_t, float coef, float2* work, int B,
    int n_coils, int H, int W) {
  FFT_LDS_SETUP(H, W)
  if (sched) {
    step = sched->step;
    noise_scale = sched->noise_scale;
    coef = sched->coef;
    step_id = sched->step_id;
  }
  const int HW = H * W;
  const int b = blockIdx.x;
  const float scale = rsqrtf((float)HW);
  float* xr = x_re + (size_t)b * HW;
  float* xi = x_im + (size_t)b * HW;
  float2* wk = work + (size_t)b * HW;
  if constexpr (LANGEVIN)
    langevin_phase(xr, xi, g_re, g_im, n_re, n_im, step, noise_scale, seed, sample_offset, step_id, b, HW);
  if (coef == 0.f) return;
  // pass 2c: forward transform of S_c z ; pass 2c+1: inverse transform of the masked residual
  for (int pass = 0; pass < 2 * n_coils; ++pass) {
    const int coil = pass >> 1;
    const bool inv = pass & 1;
    const float* sm = sens + (size_t)coil * HW;
    if (!inv) {
      for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
        int r = e / W, c = e - r * W;
        float sg = sign_rc(r, c) * sm[e];
        L.buf[e] = make_float2(xr[e] * sg, xi[e] * sg);
      }
    }
    __syncthreads();
    fft2_lds(L, H, W, inv);
    if (!inv) {
      // residual on the sampled columns, re-modulated for the inverse transform:
      //   sign*(sign*scale*v - y) = scale*v - sign*y
      const float2* yc = y + ((size_t)coil * B + b) * HW;
      for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
        int r = e / W, c = e - r * W;
        float2 v = L.buf[e];
        float2 res = make_float2(0.f, 0.f);
        if (mask_at(mask, mask_t, b, W, c)) {
          float sg = sign_rc(r, c);
          float2 yy = yc[e];
          res = make_float2(v.x * scale - sg * yy.x, v.y * scale - sg * yy.y);
        }
        L.buf[e] = res;
      }
    } else {
      const bool last = coil == n_coils - 1;
      for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
        int r = e / W, c = e - r * W;
        float2 v = L.buf[e];
        float w = sign_rc(r, c) * scale * sm[e];
        float2 a = make_float2(v.x * w, v.y * w);
        if (coil > 0) {
          float2 prev = wk[e];
          a.x += prev.x;
          a.y += prev.y;
        }
        if (last) {
          xr[e] = xr[e] - coef * a.x;
          xi[e] = xi[e] - coef * a.y;
        } else {
          wk[e] = a;
        }
      }
    }
  }
}

// The same iteration tail with the coils in parallel: a rank's batch is 13-14 samples, i.e. 14 of 256 CUs busy for four
// dependent (FFT, inverse FFT) pairs in the one-workgroup-per-sample kernel above.  Here workgroup (coil, b) forms the
// Langevin update z in registers (every coil workgroup of a sample computes the same z: the noise is injected or a pure
// function of (seed, sample, step, element)), does ITS coil's transform pair and writes r_c = conj-free S_c F^-1[M (F S_c z - y_c)]
// to work[b][coil]; ald_sense_combine_kernel then forms z once more and x = z - coef * (((r_0 + r_1) + r_2) + ...) in the
// sequential kernel's order: bit-identical results, 56 workgroups instead of 14, one transform pair deep instead of four.
__device__ __forceinline__ void langevin_value(const float* xr, const float* xi, const float* __restrict__ g_re,
                                               const float* __restrict__ g_im, const float* __restrict__ n_re,
                                               const float* __restrict__ n_im, float step, float noise_scale, uint64_t seed,
                                               int64_t sample_offset, int64_t step_id, int b, int HW, int e, float& zr,
                                               float& zi) {
  const size_t gi = (size_t)b * HW + e;
  float nr, ni;
  if (n_re) {
    nr = n_re[gi];
    ni = n_im[gi];
  } else {
    float q[4];
    const int lane4 = e & 3;
    ipdm_philox_normal4(seed, sample_offset + b, step_id, 0, (uint32_t)(e >> 2), q);
    nr = lane4 == 0 ? q[0] : lane4 == 1 ? q[1] : lane4 == 2 ? q[2] : q[3];
    ipdm_philox_normal4(seed, sample_offset + b, step_id, 1, (uint32_t)(e >> 2), q);
    ni = lane4 == 0 ? q[0] : lane4 == 1 ? q[1] : lane4 == 2 ? q[2] : q[3];
  }
  zr = xr[e] + step * g_re[gi] + nr * noise_scale;
  zi = xi[e] + step * g_im[gi] + ni * noise_scale;
}

template <bool LANGEVIN>
__global__ __launch_bounds__(FFT_THREADS) void ald_sense_coil_kernel(
    const float* x_re, const float* x_im, const float* __restrict__ g_re, const float* __restrict__ g_im,
    const float* __restrict__ n_re, const float* __restrict__ n_im, float step, float noise_scale, uint64_t seed,
    int64_t sample_offset, int64_t step_id, const ipdm_sched_t* __restrict__ sched, const float2* __restrict__ y,
    const float* __restrict__ sens, const uint8_t* __restrict__ mask, int mask_t, float coef, float2* work, int B,
    int n_coils, int H, int W) {
  FFT_LDS_SETUP(H, W)
  if (sched) {
    step = sched->step;
    noise_scale = sched->noise_scale;
    coef = sched->coef;
    step_id = sched->step_id;
  }
  if (coef == 0.f) return;                                    // the combine pass then only applies the Langevin update
  const int HW = H * W;
  const int coil = blockIdx.x, b = blockIdx.y;
  const float scale = rsqrtf((float)HW);
  const float* xr = x_re + (size_t)b * HW;
  const float* xi = x_im + (size_t)b * HW;
  const float* sm = sens + (size_t)coil * HW;
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    float zr = xr[e], zi = xi[e];
    if constexpr (LANGEVIN)
      langevin_value(xr, xi, g_re, g_im, n_re, n_im, step, noise_scale, seed, sample_offset, step_id, b, HW, e, zr, zi);
    const int r = e / W, c = e - r * W;
    const float sg = sign_rc(r, c) * sm[e];
    L.buf[e] = make_float2(zr * sg, zi * sg);
  }
  __syncthreads();
  fft2_lds(L, H, W, false);
  const float2* yc = y + ((size_t)coil * B + b) * HW;
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    const int r = e / W, c = e - r * W;
    const float2 v = L.buf[e];
    float2 res = make_float2(0.f, 0.f);
    if (mask_at(mask, mask_t, b, W, c)) {
      const float sg = sign_rc(r, c);
      const float2 yy = yc[e];
      res = make_float2(v.x * scale - sg * yy.x, v.y * scale - sg * yy.y);
    }
    L.buf[e] = res;
  }
  __syncthreads();
  fft2_lds(L, H, W, true);
  float2* wk = work + ((size_t)b * n_coils + coil) * HW;
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    const int r = e / W, c = e - r * W;
    const float2 v = L.buf[e];
    const float w = sign_rc(r, c) * scale * sm[e];
    wk[e] = make_float2(v.x * w, v.y * w);
  }
}

template <bool LANGEVIN>
__global__ __launch_bounds__(256) void ald_sense_combine_kernel(
    float* x_re, float* x_im, const float* __restrict__ g_re, const float* __restrict__ g_im,
    const float* __restrict__ n_re, const float* __restrict__ n_im, float step, float noise_scale, uint64_t seed,
    int64_t sample_offset, int64_t step_id, const ipdm_sched_t* __restrict__ sched, float coef,
    const float2* __restrict__ work, int B, int n_coils, int HW) {
  if (sched) {
    step = sched->step;
    noise_scale = sched->noise_scale;
    coef = sched->coef;
    step_id = sched->step_id;
  }
  const int b = blockIdx.y;
  float* xr = x_re + (size_t)b * HW;
  float* xi = x_im + (size_t)b * HW;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < HW; e += gridDim.x * 256) {
    float zr = xr[e], zi = xi[e];
    if constexpr (LANGEVIN)
      langevin_value(xr, xi, g_re, g_im, n_re, n_im, step, noise_scale, seed, sample_offset, step_id, b, HW, e, zr, zi);
    if (coef != 0.f) {
      const float2* wk = work + (size_t)b * n_coils * HW + e;
      float2 a = wk[0];
      for (int c = 1; c < n_coils; ++c) {                     // a_c = r_c + a_(c-1): the one-workgroup kernel's order
        const float2 rc = wk[(size_t)c * HW];
        a = make_float2(rc.x + a.x, rc.y + a.y);
      }
      zr = zr - coef * a.x;
      zi = zi - coef * a.y;
    }
    xr[e] = zr;
    xi[e] = zi;
  }
}

// Single-coil iteration tail (RandomUndersamplingFourier: A = M F, no coil maps), planar real/imag, in place:
// Langevin update (optional) + one of the reference's three single-coil data-consistency operators
//   mode 0  L2Penalty   x = z - coef * F^-1[ M (M F z - y) ]               (proximal_op.py:19-51, coef = 0.05 a/(l K), K = B)
//   mode 1  SingleCoil  x = F^-1[ (F z + coef*y) / (1 + coef*M) ]          (proximal_op.py:72-94, coef = alpha/lamda)
//   mode 2  projection  x = F^-1[ coef*y + (1-coef) M F z + (1-M) F z ]   (undersampling_fourier.py:89-97, coef = lamda)
// The image stays in LDS between the two transforms; z is re-read from x (L2-resident) for mode 0.
template <bool LANGEVIN>
__global__ __launch_bounds__(FFT_THREADS) void ald_singlecoil_step_kernel(
    float* x_re, float* x_im, const float* __restrict__ g_re, const float* __restrict__ g_im,
    const float* __restrict__ n_re, const float* __restrict__ n_im, float step, float noise_scale, uint64_t seed,
    int64_t sample_offset, int64_t step_id, const ipdm_sched_t* __restrict__ sched, const float2* __restrict__ y,
    const uint8_t* __restrict__ mask, int mask_t, float coef, int mode, int B, int H, int W) {
  FFT_LDS_SETUP(H, W)
  if (sched) {
    step = sched->step;
    noise_scale = sched->noise_scale;
    coef = sched->coef;
    step_id = sched->step_id;
  }
  const int HW = H * W;
  const int b = blockIdx.x;
  const float scale = rsqrtf((float)HW);
  float* xr = x_re + (size_t)b * HW;
  float* xi = x_im + (size_t)b * HW;
  if constexpr (LANGEVIN)
    langevin_phase(xr, xi, g_re, g_im, n_re, n_im, step, noise_scale, seed, sample_offset, step_id, b, HW);
  if (mode == 0 && coef == 0.f) return;
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    int r = e / W, c = e - r * W;
    float sg = sign_rc(r, c);
    L.buf[e] = make_float2(xr[e] * sg, xi[e] * sg);
  }
  __syncthreads();
  fft2_lds(L, H, W, false);
  // k-space value K = sg*scale*v; the inverse transform wants sg*K' -> every formula is written on scale*v and sg*y
  const float2* yb = y + (size_t)b * HW;
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    int r = e / W, c = e - r * W;
    float2 v = L.buf[e];
    v.x *= scale;
    v.y *= scale;
    const bool m = mask_at(mask, mask_t, b, W, c);
    float2 o;
    if (mode == 0) {
      o = make_float2(0.f, 0.f);
      if (m) {
        float sg = sign_rc(r, c);
        float2 yy = yb[e];
        o = make_float2(v.x - sg * yy.x, v.y - sg * yy.y);
      }
    } else if (mode == 1) {
      float sg = sign_rc(r, c) * coef;
      float2 yy = yb[e];
      float inv = m ? 1.f / (1.f + coef) : 1.f;
      o = make_float2((v.x + sg * yy.x) * inv, (v.y + sg * yy.y) * inv);
    } else {
      float sg = sign_rc(r, c) * coef;
      float2 yy = yb[e];
      float keep = m ? 1.f - coef : 1.f;
      o = make_float2(sg * yy.x + keep * v.x, sg * yy.y + keep * v.y);
    }
    L.buf[e] = o;
  }
  __syncthreads();
  fft2_lds(L, H, W, true);
  for (int e = threadIdx.x; e < HW; e += FFT_THREADS) {
    int r = e / W, c = e - r * W;
    float2 v = L.buf[e];
    float w = sign_rc(r, c) * scale;
    if (mode == 0) {
      xr[e] = xr[e] - coef * (v.x * w);
      xi[e] = xi[e] - coef * (v.y * w);
    } else {
      xr[e] = v.x * w;
      xi[e] = v.y * w;
    }
  }
}

template <typename K>
static int set_lds_limit(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return IPDM_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes);
  return e == hipSuccess ? IPDM_OK : (int)e;
}

template <bool LANGEVIN>
static int launch_sense_step_coils(float* x_re, float* x_im, const float* g_re, const float* g_im, const float* n_re,
                                   const float* n_im, float step, float noise_scale, uint64_t seed, int64_t sample_offset,
                                   int64_t step_id, const ipdm_sched_t* sched, const float2* y, const float* sens,
                                   const uint8_t* mask, int mask_t, float coef, float2* work, int B, int n_coils, int H,
                                   int W, hipStream_t st) {
  if (B > 65535) return IPDM_EUNSUPPORTED;
  const size_t lds = lds_bytes(H, W);
  const int rc = set_lds_limit(ald_sense_coil_kernel<LANGEVIN>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(ald_sense_coil_kernel<LANGEVIN>, dim3(n_coils, B), dim3(FFT_THREADS), lds, st, x_re, x_im, g_re, g_im,
                     n_re, n_im, step, noise_scale, seed, (long long)sample_offset, (long long)step_id, sched, y, sens, mask,
                     mask_t, coef, work, B, n_coils, H, W);
  const int HW = H * W;
  int gx = (HW + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(ald_sense_combine_kernel<LANGEVIN>, dim3(gx, B), dim3(256), 0, st, x_re, x_im, g_re, g_im, n_re, n_im,
                     step, noise_scale, seed, (long long)sample_offset, (long long)step_id, sched, coef, work, B, n_coils, HW);
  return ipdm_launch_status();
}

}  // namespace

extern "C" int64_t ipdm_fft2c_workspace_bytes(int batch, int H, int W) {
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  return (lds_fft_ok(H, W) || ipdm_kspace_large::large_ok(H, W)) ? 0 : (int64_t)batch * H * W * 8;
}

// IPDM_SENSE_COILS=0: the one-workgroup-per-sample kernel (tuning / comparison; same bits)
static int sense_coil_parallel() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("IPDM_SENSE_COILS");
    v = e ? atoi(e) : 1;
  }
  return v;
}

extern "C" int64_t ipdm_sense_workspace_bytes(int B, int n_coils, int H, int W) {
  if (B <= 0 || n_coils <= 0 || H <= 0 || W <= 0) return 0;
  if (lds_fft_ok(H, W)) return (int64_t)B * n_coils * H * W * 8;   // one plane per (sample, coil): the coil-parallel path
  return ipdm_kspace_large::workspace_bytes(B, n_coils, H, W);
}

extern "C" int ipdm_fft2c_c64(const float* in, float* out, int batch, int H, int W, int inverse, float* workspace,
                              void* stream) {
  IPDM_REQUIRE(batch >= 0 && H > 0 && W > 0 && H <= 1024 && W <= 1024);
  if (batch == 0) return IPDM_OK;
  IPDM_REQUIRE(in && out);
  hipStream_t s = ipdm_stream(stream);
  if (lds_fft_ok(H, W)) {
    size_t lds = lds_bytes(H, W);
    int rc = set_lds_limit(fft2c_lds_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(fft2c_lds_kernel, dim3(batch), dim3(FFT_THREADS), lds, s, reinterpret_cast<const float2*>(in),
                       reinterpret_cast<float2*>(out), H, W, inverse);
    return ipdm_launch_status();
  }
  if (ipdm_kspace_large::large_ok(H, W))
    return ipdm_kspace_large::fft2c(reinterpret_cast<const float2*>(in), reinterpret_cast<float2*>(out), batch, H, W,
                                    inverse, s);
  IPDM_REQUIRE(workspace);
  // pass 1: DFT along W, [b][H][W] -> ws [b][W][H]; pass 2: DFT along H, ws -> out [b][H][W]
  hipLaunchKernelGGL(dft_rows_transposed_kernel, dim3(H, batch), dim3(256), (size_t)2 * W * sizeof(float2), s,
                     reinterpret_cast<const float2*>(in), reinterpret_cast<float2*>(workspace), H, W, inverse);
  hipLaunchKernelGGL(dft_rows_transposed_kernel, dim3(W, batch), dim3(256), (size_t)2 * H * sizeof(float2), s,
                     reinterpret_cast<const float2*>(workspace), reinterpret_cast<float2*>(out), W, H, inverse);
  return ipdm_launch_status();
}

extern "C" int ipdm_sense_forward_c64(const float* x, const float* sens, const uint8_t* mask, int mask_t, float* y,
                                      int B, int n_coils, int H, int W, void* stream) {
  IPDM_REQUIRE(B >= 0 && n_coils > 0 && H > 0 && W > 0 && mask_t > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && mask && y && (sens || n_coils == 1));
  if (ipdm_kspace_large::large_ok(H, W))
    return ipdm_kspace_large::sense_forward(reinterpret_cast<const float2*>(x), sens, mask, mask_t,
                                            reinterpret_cast<float2*>(y), B, n_coils, H, W, ipdm_stream(stream));
  if (!lds_fft_ok(H, W)) return IPDM_EUNSUPPORTED;
  size_t lds = lds_bytes(H, W);
  int rc = set_lds_limit(sense_forward_kernel, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(sense_forward_kernel, dim3(B, n_coils), dim3(FFT_THREADS), lds, ipdm_stream(stream),
                     reinterpret_cast<const float2*>(x), sens, mask, mask_t, reinterpret_cast<float2*>(y), B, H, W);
  return ipdm_launch_status();
}

extern "C" int ipdm_sense_adjoint_c64(const float* s, const float* sens, const uint8_t* mask, int mask_t,
                                      int apply_mask, float* x, float* workspace, int B, int n_coils, int H, int W,
                                      void* stream) {
  IPDM_REQUIRE(B >= 0 && n_coils > 0 && H > 0 && W > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(s && sens && x);
  if (apply_mask) IPDM_REQUIRE(mask && mask_t > 0);
  if (ipdm_kspace_large::large_ok(H, W)) {
    IPDM_REQUIRE(workspace);
    return ipdm_kspace_large::sense_adjoint(reinterpret_cast<const float2*>(s), sens, mask, mask_t > 0 ? mask_t : 1,
                                            apply_mask, reinterpret_cast<float2*>(x), nullptr,
                                            reinterpret_cast<float2*>(workspace), B, n_coils, H, W, ipdm_stream(stream));
  }
  if (!lds_fft_ok(H, W)) return IPDM_EUNSUPPORTED;
  size_t lds = lds_bytes(H, W);
  int rc = set_lds_limit(sense_adjoint_kernel<false>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(sense_adjoint_kernel<false>, dim3(B), dim3(FFT_THREADS), lds, ipdm_stream(stream),
                     reinterpret_cast<const float2*>(s), sens, mask, mask_t > 0 ? mask_t : 1, apply_mask, x, B, n_coils,
                     H, W);
  return ipdm_launch_status();
}

extern "C" int ipdm_sense_ssos_c64(const float* s, float* out, float* workspace, int B, int n_coils, int H, int W,
                                   void* stream) {
  IPDM_REQUIRE(B >= 0 && n_coils > 0 && H > 0 && W > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(s && out);
  if (ipdm_kspace_large::large_ok(H, W)) {
    IPDM_REQUIRE(workspace);
    return ipdm_kspace_large::sense_adjoint(reinterpret_cast<const float2*>(s), nullptr, nullptr, 1, 0, nullptr, out,
                                            reinterpret_cast<float2*>(workspace), B, n_coils, H, W, ipdm_stream(stream));
  }
  if (!lds_fft_ok(H, W)) return IPDM_EUNSUPPORTED;
  size_t lds = lds_bytes(H, W);
  int rc = set_lds_limit(sense_adjoint_kernel<true>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(sense_adjoint_kernel<true>, dim3(B), dim3(FFT_THREADS), lds, ipdm_stream(stream),
                     reinterpret_cast<const float2*>(s), nullptr, nullptr, 1, 0, out, B, n_coils, H, W);
  return ipdm_launch_status();
}

extern "C" int ipdm_sense_l2prox_f32(const float* z_re, const float* z_im, const float* y, const float* sens,
                                     const uint8_t* mask, int mask_t, float coef, float* out_re, float* out_im,
                                     float* work, int B, int n_coils, int H, int W, void* stream) {
  IPDM_REQUIRE(B >= 0 && n_coils > 0 && H > 0 && W > 0 && mask_t > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(z_re && z_im && y && sens && mask && out_re && out_im && work);
  const bool large = ipdm_kspace_large::large_ok(H, W);
  if (!large && !lds_fft_ok(H, W)) return IPDM_EUNSUPPORTED;
  hipStream_t st = ipdm_stream(stream);
  const size_t bytes = (size_t)B * H * W * sizeof(float);
  if (out_re != z_re && hipMemcpyAsync(out_re, z_re, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return (int)hipGetLastError();
  if (out_im != z_im && hipMemcpyAsync(out_im, z_im, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return (int)hipGetLastError();
  if (large)
    return ipdm_kspace_large::prox_step(out_re, out_im, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0ull, 0, 0, nullptr,
                                        reinterpret_cast<const float2*>(y), sens, mask, mask_t, coef, 0,
                                        reinterpret_cast<float2*>(work), B, n_coils, H, W, st);
  if (sense_coil_parallel())
    return launch_sense_step_coils<false>(out_re, out_im, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0ull, 0, 0, nullptr,
                                          reinterpret_cast<const float2*>(y), sens, mask, mask_t, coef,
                                          reinterpret_cast<float2*>(work), B, n_coils, H, W, st);
  size_t lds = lds_bytes(H, W);
  int rc = set_lds_limit(ald_sense_step_kernel<false>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(ald_sense_step_kernel<false>, dim3(B), dim3(FFT_THREADS), lds, st, out_re, out_im, nullptr,
                     nullptr, nullptr, nullptr, 0.f, 0.f, 0ull, 0ll, 0ll, nullptr, reinterpret_cast<const float2*>(y), sens,
                     mask, mask_t, coef, reinterpret_cast<float2*>(work), B, n_coils, H, W);
  return ipdm_launch_status();
}

extern "C" int ipdm_ald_sense_step_f32(float* x_re, float* x_im, const float* g_re, const float* g_im,
                                       const float* noise_re, const float* noise_im, float step, float noise_scale,
                                       uint64_t seed, int64_t sample_offset, int64_t step_id,
                                       const ipdm_sched_t* dev_sched, const float* y, const float* sens, const uint8_t* mask, int mask_t, float coef, float* work,
                                       int B, int n_coils, int H, int W, void* stream) {
  IPDM_REQUIRE(B >= 0 && n_coils > 0 && H > 0 && W > 0 && mask_t > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x_re && x_im && g_re && g_im && y && sens && mask && work);
  IPDM_REQUIRE((noise_re == nullptr) == (noise_im == nullptr));
  if (ipdm_kspace_large::large_ok(H, W))
    return ipdm_kspace_large::prox_step(x_re, x_im, g_re, g_im, noise_re, noise_im, step, noise_scale, seed, sample_offset,
                                        step_id, dev_sched, reinterpret_cast<const float2*>(y), sens, mask, mask_t, coef, 0,
                                        reinterpret_cast<float2*>(work), B, n_coils, H, W, ipdm_stream(stream));
  if (!lds_fft_ok(H, W)) return IPDM_EUNSUPPORTED;
  if (sense_coil_parallel())
    return launch_sense_step_coils<true>(x_re, x_im, g_re, g_im, noise_re, noise_im, step, noise_scale, seed, sample_offset,
                                         step_id, dev_sched, reinterpret_cast<const float2*>(y), sens, mask, mask_t, coef,
                                         reinterpret_cast<float2*>(work), B, n_coils, H, W, ipdm_stream(stream));
  size_t lds = lds_bytes(H, W);
  int rc = set_lds_limit(ald_sense_step_kernel<true>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(ald_sense_step_kernel<true>, dim3(B), dim3(FFT_THREADS), lds, ipdm_stream(stream), x_re, x_im,
                     g_re, g_im, noise_re, noise_im, step, noise_scale, seed, (long long)sample_offset,
                     (long long)step_id, dev_sched, reinterpret_cast<const float2*>(y), sens, mask, mask_t, coef,
                     reinterpret_cast<float2*>(work), B, n_coils, H, W);
  return ipdm_launch_status();
}

extern "C" int ipdm_singlecoil_prox_f32(const float* z_re, const float* z_im, const float* y, const uint8_t* mask,
                                        int mask_t, float coef, int mode, float* out_re, float* out_im, float* workspace,
                                        int B, int H, int W, void* stream) {
  IPDM_REQUIRE(B >= 0 && H > 0 && W > 0 && mask_t > 0 && mode >= 0 && mode <= 2);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(z_re && z_im && y && mask && out_re && out_im);
  const bool large = ipdm_kspace_large::large_ok(H, W);
  if (!large && !lds_fft_ok(H, W)) return IPDM_EUNSUPPORTED;
  hipStream_t st = ipdm_stream(stream);
  const size_t bytes = (size_t)B * H * W * sizeof(float);
  if (out_re != z_re && hipMemcpyAsync(out_re, z_re, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return (int)hipGetLastError();
  if (out_im != z_im && hipMemcpyAsync(out_im, z_im, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return (int)hipGetLastError();
  if (large) {
    IPDM_REQUIRE(workspace);
    return ipdm_kspace_large::prox_step(out_re, out_im, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0ull, 0, 0, nullptr,
                                        reinterpret_cast<const float2*>(y), nullptr, mask, mask_t, coef, mode,
                                        reinterpret_cast<float2*>(workspace), B, 1, H, W, st);
  }
  size_t lds = lds_bytes(H, W);
  int rc = set_lds_limit(ald_singlecoil_step_kernel<false>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(ald_singlecoil_step_kernel<false>, dim3(B), dim3(FFT_THREADS), lds, st, out_re, out_im, nullptr,
                     nullptr, nullptr, nullptr, 0.f, 0.f, 0ull, 0ll, 0ll, nullptr, reinterpret_cast<const float2*>(y), mask,
                     mask_t, coef, mode, B, H, W);
  return ipdm_launch_status();
}

extern "C" int ipdm_ald_singlecoil_step_f32(float* x_re, float* x_im, const float* g_re, const float* g_im,
                                            const float* noise_re, const float* noise_im, float step, float noise_scale,
                                            uint64_t seed, int64_t sample_offset, int64_t step_id,
                                            const ipdm_sched_t* dev_sched, const float* y, const uint8_t* mask, int mask_t,
                                            float coef, int mode, float* workspace, int B, int H, int W, void* stream) {
  IPDM_REQUIRE(B >= 0 && H > 0 && W > 0 && mask_t > 0 && mode >= 0 && mode <= 2);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x_re && x_im && g_re && g_im && y && mask);
  IPDM_REQUIRE((noise_re == nullptr) == (noise_im == nullptr));
  if (ipdm_kspace_large::large_ok(H, W)) {
    IPDM_REQUIRE(workspace);
    return ipdm_kspace_large::prox_step(x_re, x_im, g_re, g_im, noise_re, noise_im, step, noise_scale, seed, sample_offset,
                                        step_id, dev_sched, reinterpret_cast<const float2*>(y), nullptr, mask, mask_t, coef,
                                        mode, reinterpret_cast<float2*>(workspace), B, 1, H, W, ipdm_stream(stream));
  }
  if (!lds_fft_ok(H, W)) return IPDM_EUNSUPPORTED;
  size_t lds = lds_bytes(H, W);
  int rc = set_lds_limit(ald_singlecoil_step_kernel<true>, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(ald_singlecoil_step_kernel<true>, dim3(B), dim3(FFT_THREADS), lds, ipdm_stream(stream), x_re, x_im,
                     g_re, g_im, noise_re, noise_im, step, noise_scale, seed, (long long)sample_offset,
                     (long long)step_id, dev_sched, reinterpret_cast<const float2*>(y), mask, mask_t, coef, mode, B, H, W);
  return ipdm_launch_status();
}
