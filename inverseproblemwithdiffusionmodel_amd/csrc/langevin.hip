// Langevin noise-inject / step (reference: ncsn/models/ALD_optimizers.py:114-117, 238-241) and the
// Philox normal generator the fused kernels share.  HBM-bound: 12 bytes/element with Philox noise,
// 16 with injected noise.
#include "ipdm_common.h"

namespace {

template <bool PHILOX>
__global__ __launch_bounds__(256) void langevin_kernel(float* x, const float* __restrict__ g,
                                                       const float* __restrict__ noise, float step, float noise_scale,
                                                       uint64_t seed, int64_t sample_offset, int64_t step_id,
                                                       const ipdm_sched_t* __restrict__ sched, int64_t n_samples,
                                                       int64_t sample_elems) {
  if (sched) {
    step = sched->step;
    noise_scale = sched->noise_scale;
    step_id = sched->step_id;
  }
  // one thread per quad of elements inside a sample; samples on blockIdx.y
  const int64_t quads = (sample_elems + 3) / 4;
  for (int64_t smp = blockIdx.y; smp < n_samples; smp += gridDim.y) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
      float nz[4];
      const int64_t base = smp * sample_elems + q * 4;
      const int cnt = (int)((sample_elems - q * 4) < 4 ? (sample_elems - q * 4) : 4);
      if constexpr (PHILOX) ipdm_philox_normal4(seed, sample_offset + smp, step_id, 0, (uint32_t)q, nz);
      const bool vec = cnt == 4 && ((base & 3) == 0);
      if (vec) {
        float4 xv = *reinterpret_cast<const float4*>(x + base);
        float4 gv = *reinterpret_cast<const float4*>(g + base);
        if constexpr (!PHILOX) {
          float4 nv = *reinterpret_cast<const float4*>(noise + base);
          nz[0] = nv.x; nz[1] = nv.y; nz[2] = nv.z; nz[3] = nv.w;
        }
        xv.x = xv.x + step * gv.x + nz[0] * noise_scale;
        xv.y = xv.y + step * gv.y + nz[1] * noise_scale;
        xv.z = xv.z + step * gv.z + nz[2] * noise_scale;
        xv.w = xv.w + step * gv.w + nz[3] * noise_scale;
        *reinterpret_cast<float4*>(x + base) = xv;
      } else {
        for (int j = 0; j < cnt; ++j) {
          float n = PHILOX ? nz[j] : noise[base + j];
          x[base + j] = x[base + j] + step * g[base + j] + n * noise_scale;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void philox_normal_kernel(float* out, uint64_t seed, int64_t sample_offset,
                                                            int64_t step_id, int plane, int64_t n_samples,
                                                            int64_t sample_elems) {
  const int64_t quads = (sample_elems + 3) / 4;
  for (int64_t smp = blockIdx.y; smp < n_samples; smp += gridDim.y) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
      float nz[4];
      ipdm_philox_normal4(seed, sample_offset + smp, step_id, plane, (uint32_t)q, nz);
      for (int j = 0; j < 4 && q * 4 + j < sample_elems; ++j) out[smp * sample_elems + q * 4 + j] = nz[j];
    }
  }
}

inline dim3 sample_grid(int64_t n_samples, int64_t sample_elems) {
  int64_t quads = (sample_elems + 3) / 4;
  int gx = (int)((quads + 255) / 256);
  if (gx > 512) gx = 512;
  int gy = (int)(n_samples > 1024 ? 1024 : n_samples);
  return dim3(gx, gy);
}

}  // namespace

extern "C" int ipdm_langevin_step_f32(float* x, const float* g, const float* noise, float step, float noise_scale,
                                      uint64_t seed, int64_t sample_offset, int64_t step_id,
                                      const ipdm_sched_t* dev_sched, int64_t n_samples, int64_t sample_elems,
                                      void* stream) {
  IPDM_REQUIRE(n_samples >= 0 && sample_elems >= 0);
  if (n_samples == 0 || sample_elems == 0) return IPDM_OK;
  IPDM_REQUIRE(x && g);
  dim3 grid = sample_grid(n_samples, sample_elems);
  if (noise)
    hipLaunchKernelGGL(langevin_kernel<false>, grid, dim3(256), 0, ipdm_stream(stream), x, g, noise, step, noise_scale,
                       seed, (long long)sample_offset, (long long)step_id, dev_sched, (long long)n_samples,
                       (long long)sample_elems);
  else
    hipLaunchKernelGGL(langevin_kernel<true>, grid, dim3(256), 0, ipdm_stream(stream), x, g, noise, step, noise_scale,
                       seed, (long long)sample_offset, (long long)step_id, dev_sched, (long long)n_samples,
                       (long long)sample_elems);
  return ipdm_launch_status();
}

extern "C" int ipdm_philox_normal_f32(float* out, uint64_t seed, int64_t sample_offset, int64_t step_id, int plane,
                                      int64_t n_samples, int64_t sample_elems, void* stream) {
  IPDM_REQUIRE(n_samples >= 0 && sample_elems >= 0);
  if (n_samples == 0 || sample_elems == 0) return IPDM_OK;
  IPDM_REQUIRE(out);
  hipLaunchKernelGGL(philox_normal_kernel, sample_grid(n_samples, sample_elems), dim3(256), 0, ipdm_stream(stream), out,
                     seed, (long long)sample_offset, (long long)step_id, plane, (long long)n_samples,
                     (long long)sample_elems);
  return ipdm_launch_status();
}

// host-side copy of the generator's integer stage (no GPU needed): the CPU tests check it against the published
// Philox4x32-10 known-answer vectors and that different seeds / steps / samples never share a block
extern "C" int ipdm_philox_block_host(uint64_t seed, int64_t sample, int64_t step_id, int plane, uint32_t quad,
                                      uint32_t* out4) {
  IPDM_REQUIRE(out4);
  uint32_t c[4];
  ipdm_philox_block(seed, sample, step_id, plane, quad, c);
  for (int i = 0; i < 4; ++i) out4[i] = c[i];
  return IPDM_OK;
}
