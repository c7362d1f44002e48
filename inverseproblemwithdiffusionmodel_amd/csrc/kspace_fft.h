// Shared device code of the k-space kernels (kspace.hip: whole image in one CU's LDS; kspace_large.hip: row / column
// passes for images beyond the LDS): radix-4/2 Stockham FFT over lines held in LDS, sign and mask helpers.
#pragma once
#include "ipdm_common.h"

namespace ipdm_kspace {

constexpr int FFT_THREADS = 1024;
constexpr int FFT_MAX_ELEMS = 16384;           // 128 KiB of float2
constexpr int FFT_EPT = FFT_MAX_ELEMS / FFT_THREADS;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// LDS image + twiddle table.  tw[q] = exp(-2*pi*i*q/twN), twN = max(H, W).
struct FftLds {
  float2* buf;
  float2* tw;
  int twN;
};

__device__ __forceinline__ void fft_make_twiddles(const FftLds& L) {
  for (int q = threadIdx.x; q < L.twN; q += blockDim.x) {
    double s, c;
    sincospi(-2.0 * (double)q / (double)L.twN, &s, &c);
    L.tw[q] = make_float2((float)c, (float)s);
  }
}

// One in-place Stockham stage of radix R over `nlines` lines of length N.
//   element (line, n) lives at buf[line*ls + n*es];  lines_fast: consecutive threads -> consecutive lines.
// Every thread reads all its butterflies, the workgroup barriers, then everyone writes.
template <int R>
__device__ __forceinline__ void fft_stage(const FftLds& L, int N, int Ns, int es, int ls, int nlines, bool lines_fast,
                                          bool inverse) {
  constexpr int BPT = FFT_EPT / R;             // butterflies per thread at the largest image
  const int nb = N / R;                        // butterflies per line
  const int total = nb * nlines;
  const int twstep = L.twN / (Ns * R);
  float2 v[BPT][R];
  int dst[BPT];
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    int i = threadIdx.x + u * FFT_THREADS;
    dst[u] = -1;
    if (i < total) {
      int line, j;
      if (lines_fast) { j = i / nlines; line = i - j * nlines; }
      else { line = i / nb; j = i - line * nb; }
      int k = j & (Ns - 1);
      int base = line * ls;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        float2 x = L.buf[base + (j + t * nb) * es];
        if (t > 0) {
          float2 w = L.tw[(k * t * twstep) & (L.twN - 1)];
          if (inverse) w.y = -w.y;
          x = cmul(x, w);
        }
        v[u][t] = x;
      }
      dst[u] = base + (((j - k) * R) + k) * es;
      if constexpr (R == 2) {
        float2 a = v[u][0], b = v[u][1];
        v[u][0] = make_float2(a.x + b.x, a.y + b.y);
        v[u][1] = make_float2(a.x - b.x, a.y - b.y);
      } else {
        float2 a = v[u][0], b = v[u][1], c = v[u][2], d = v[u][3];
        float2 apc = make_float2(a.x + c.x, a.y + c.y), amc = make_float2(a.x - c.x, a.y - c.y);
        float2 bpd = make_float2(b.x + d.x, b.y + d.y), bmd = make_float2(b.x - d.x, b.y - d.y);
        // forward: -i*(b-d) = (bmd.y, -bmd.x); inverse: +i*(b-d) = (-bmd.y, bmd.x)
        float2 jb = inverse ? make_float2(-bmd.y, bmd.x) : make_float2(bmd.y, -bmd.x);
        v[u][0] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
        v[u][1] = make_float2(amc.x + jb.x, amc.y + jb.y);
        v[u][2] = make_float2(apc.x - bpd.x, apc.y - bpd.y);
        v[u][3] = make_float2(amc.x - jb.x, amc.y - jb.y);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    if (dst[u] >= 0) {
#pragma unroll
      for (int t = 0; t < R; ++t) L.buf[dst[u] + t * Ns * es] = v[u][t];
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void fft_lines(const FftLds& L, int N, int es, int ls, int nlines, bool lines_fast,
                                          bool inverse) {
  int Ns = 1;
  while (Ns * 4 <= N) {
    fft_stage<4>(L, N, Ns, es, ls, nlines, lines_fast, inverse);
    Ns *= 4;
  }
  if (Ns < N) fft_stage<2>(L, N, Ns, es, ls, nlines, lines_fast, inverse);
}

// plain (uncentred, unnormalised) 2-D FFT of buf[H][W]; caller applies the (-1)^(r+c) flips and 1/sqrt(HW).
__device__ __forceinline__ void fft2_lds(const FftLds& L, int H, int W, bool inverse) {
  fft_lines(L, W, 1, W, H, false, inverse);   // along rows
  fft_lines(L, H, W, 1, W, true, inverse);    // along columns
}

__device__ __forceinline__ float sign_rc(int r, int c) { return ((r + c) & 1) ? -1.f : 1.f; }

__host__ __device__ __forceinline__ bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline bool lds_fft_ok(int H, int W) {
  return is_pow2(H) && is_pow2(W) && H >= 4 && W >= 4 && (int64_t)H * W <= FFT_MAX_ELEMS;
}
static inline size_t lds_bytes(int H, int W) { return ((size_t)H * W + (size_t)(H > W ? H : W)) * sizeof(float2); }


__device__ __forceinline__ bool mask_at(const uint8_t* mask, int mask_t, int b, int W, int c) {
  return mask[(size_t)(mask_t == 1 ? 0 : b % mask_t) * W + c] != 0;
}

}  // namespace ipdm_kspace

// row / column-pass operators for power-of-two images beyond the LDS (kspace_large.hip)
namespace ipdm_kspace_large {
bool large_ok(int H, int W);
int64_t workspace_bytes(int B, int n_coils, int H, int W);
int fft2c(const float2* in, float2* out, int batch, int H, int W, int inverse, hipStream_t s);
int sense_forward(const float2* x, const float* sens, const uint8_t* mask, int mask_t, float2* y, int B, int n_coils,
                  int H, int W, hipStream_t s);
int sense_adjoint(const float2* sm, const float* sens, const uint8_t* mask, int mask_t, int apply_mask, float2* x_out,
                  float* ssos_out, float2* ws, int B, int n_coils, int H, int W, hipStream_t s);
int prox_step(float* x_re, float* x_im, const float* g_re, const float* g_im, const float* n_re, const float* n_im,
              float step, float noise_scale, uint64_t seed, int64_t sample_offset, int64_t step_id,
              const ipdm_sched_t* sched, const float2* y, const float* sens, const uint8_t* mask, int mask_t, float coef,
              int mode, float2* ws, int B, int n_coils, int H, int W, hipStream_t s);
}  // namespace ipdm_kspace_large
