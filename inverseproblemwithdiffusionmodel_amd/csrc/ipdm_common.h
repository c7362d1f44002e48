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
