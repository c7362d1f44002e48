// Shared helpers for the libipdm.so kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ipdm.h"

#define IPDM_WAVE 64

#define IPDM_REQUIRE(cond)          \
  do {                              \
    if (!(cond)) return IPDM_EINVAL; \
  } while (0)

// returns the launch status without synchronising (callers may be capturing a graph)
static inline int ipdm_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? IPDM_OK : (int)e;
}

static inline hipStream_t ipdm_stream(void* s) { return (hipStream_t)s; }

// memory-bound elementwise grids: cap at 256 CUs x 8 blocks and grid-stride the rest
static inline int ipdm_ew_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float ipdm_act(float v, int act) {
  switch (act) {
    case IPDM_ACT_ELU: return v > 0.f ? v : expm1f(v);
    case IPDM_ACT_RELU: return v > 0.f ? v : 0.f;
    case IPDM_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case IPDM_ACT_SWISH: return v / (1.f + expf(-v));
    default: return v;
  }
}

template <int ACT>
__device__ __forceinline__ float ipdm_act_t(float v) {
  if constexpr (ACT == IPDM_ACT_ELU) return v > 0.f ? v : expm1f(v);
  else if constexpr (ACT == IPDM_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == IPDM_ACT_LRELU02) return v > 0.f ? v : 0.2f * v;
  else if constexpr (ACT == IPDM_ACT_SWISH) return v / (1.f + expf(-v));
  else return v;
}

// 64-lane wavefront reductions (DPP/shuffle based)
__device__ __forceinline__ float ipdm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double ipdm_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- per-image activation maxima (the f16x2 convolutions' dynamic range; include/ipdm.h, "maxima vectors") ----------------
// A maxima vector is [B][IPDM_AMAX_SLOT] floats: eight WAYS per image, one 64-byte line each (the first float of the line).
// Producers accumulate max |value| with atomic max into the way of their wave; a consumer takes the max over the eight ways.
// Why ways, and why a line each: float atomics execute at the memory side and serialise per 64-byte LINE (~26 ns each); the
// persistent convolution kernels finish a tile pass on all CUs at once, 256 waves per image, and vmcnt retires in order -- with
// one word per image every later load of those waves waited microseconds for its own atomic (measured: +18 % convolution time).
constexpr int IPDM_AMAX_WAY_STRIDE = 16;                        // floats between ways

__device__ __forceinline__ float ipdm_wave_max(float v) {       // v >= 0 in every lane; the maximum in every lane
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false)));
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
// |v| >= 0: the unsigned order of the bit patterns is the float order; max is exact and order-independent (deterministic)
__device__ __forceinline__ void ipdm_amax_atomic(float* image_slot, int way, float m) {
  atomicMax(reinterpret_cast<unsigned*>(image_slot + (way & (IPDM_AMAX_WAYS - 1)) * IPDM_AMAX_WAY_STRIDE), __builtin_bit_cast(unsigned, m));
}
// per-thread maximum -> ONE atomic per wave on the image's slot (way = the wave's global index)
__device__ __forceinline__ void ipdm_amax_commit(float m, float* image_slot, int way) {
  m = ipdm_wave_max(m);
  if ((threadIdx.x & 63) == 0) ipdm_amax_atomic(image_slot, way, m);
}
// per-thread maximum -> ONE atomic per 256-thread WORKGROUP (simple kernels with LDS to spare; `red`: four floats of LDS);
// every thread of the workgroup must call it
__device__ __forceinline__ void ipdm_amax_commit_block(float m, float* image_slot, int way, float* red) {
  m = ipdm_wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) ipdm_amax_atomic(image_slot, way, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}
// consumer, vector form: lanes 0..7 load a way each, wave max -> the image's value in every lane (the loads are ordinary VMEM:
// the compiler overlaps them with the kernel's other prologue loads; one image per workgroup)
__device__ __forceinline__ float ipdm_amax_read_v(const float* image_slot) {
  const int lane = threadIdx.x & 63;
  const float v = image_slot[(lane & (IPDM_AMAX_WAYS - 1)) * IPDM_AMAX_WAY_STRIDE];
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ipdm_wave_max(v))));
}
// consumer: max over the ways of one image (wave-uniform address: eight scalar loads)
__device__ __forceinline__ float ipdm_amax_read(const float* image_slot) {
  unsigned w0, w1, w2, w3, w4, w5, w6, w7;
  asm volatile(
      "s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\ts_load_dword %3, %8, 0xc0\n\t"
      "s_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\ts_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(w0), "=&s"(w1), "=&s"(w2), "=&s"(w3), "=&s"(w4), "=&s"(w5), "=&s"(w6), "=&s"(w7)
      : "s"(image_slot)
      : "memory");
  const unsigned a = w0 > w1 ? w0 : w1, b = w2 > w3 ? w2 : w3, c = w4 > w5 ? w4 : w5, d = w6 > w7 ? w6 : w7;
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return __builtin_bit_cast(float, ab > cd ? ab : cd);
}

// ---- Philox4x32-10 counter-based generator --------------------------------------------------
// key     = (seed lo, seed hi)                      -- the seed alone: two seeds never share a stream
// counter = (quad index within the plane, step lo, sample id lo, plane | step bits 32..39 << 8 | sample id bits 32..47 << 16)
// Every (seed, sample, step, plane, quad) is a distinct (key, counter) pair, so no two draws of one run, and no two
// runs with different seeds, reuse a block.  Four uniforms -> four normals (Box-Muller).
struct IpdmPhilox {
  uint32_t c[4];
  uint32_t k[2];
};
__host__ __device__ __forceinline__ void ipdm_philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__host__ __device__ __forceinline__ void ipdm_philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    ipdm_philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// the 128 random bits of quad `q` of plane `plane` of sample `sample` at step `step` (host + device: the host copy is
// what ipdm_philox_block_host exposes to the CPU tests)
__host__ __device__ __forceinline__ void ipdm_philox_block(uint64_t seed, int64_t sample, int64_t step, int plane,
                                                           uint32_t q, uint32_t (&c)[4]) {
  const uint64_t us = (uint64_t)sample, ut = (uint64_t)step;
  c[0] = q;
  c[1] = (uint32_t)ut;
  c[2] = (uint32_t)us;
  c[3] = ((uint32_t)plane & 0xffu) | (((uint32_t)(ut >> 32) & 0xffu) << 8) | (((uint32_t)(us >> 32) & 0xffffu) << 16);
  ipdm_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}
// four standard normals for quad `q` (elements 4q..4q+3) of plane `plane` of sample `sample` at step `step`
__device__ __forceinline__ void ipdm_philox_normal4(uint64_t seed, int64_t sample, int64_t step, int plane,
                                                    uint32_t q, float (&out)[4]) {
  uint32_t c[4];
  ipdm_philox_block(seed, sample, step, plane, q, c);
  const float two32 = 2.3283064365386963e-10f;                 // 2^-32
  float u0 = ((float)c[0] + 0.5f) * two32, u1 = ((float)c[1] + 0.5f) * two32;
  float u2 = ((float)c[2] + 0.5f) * two32, u3 = ((float)c[3] + 0.5f) * two32;
  u0 = fminf(fmaxf(u0, 1.0e-10f), 1.0f);
  u2 = fminf(fmaxf(u2, 1.0e-10f), 1.0f);
  float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
  float s0, c0, s1, c1;
  sincospif(2.f * u1, &s0, &c0);
  sincospif(2.f * u3, &s1, &c1);
  out[0] = r0 * c0; out[1] = r0 * s0; out[2] = r1 * c1; out[3] = r1 * s1;
}
