// Diagnostic for ranks / streams that SHARE one MI355X (sharding.py): kernels launched on stream B ("victims": a
// checksum of scalar fp32 VALU math, the same with v_pk_mul_f32 / v_pk_add_f32, and the library's bilinear resize, whose
// compiler-generated code uses the packed instructions) are compared bit for bit with a quiet run while stream A runs an
// "aggressor" (the library's convolution kernels; synthetic kernels: a chain of f16 MFMAs on constant operands, and the same
// chain fed by global loads / by ds_read_b128 / by both).  Result on MI355X / ROCm 7.2 (profiles/r03_shared_card_probe.txt):
// the packed-fp32 victims are corrupted by the library's direct 3x3 split-operand kernels AND by the 30-line synthetic kernel
// whose MFMA operands come from LDS reads -- the disturbance does not need this library; scalar-fp32 victims are never hit.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude scripts/shared_card_probe.hip -o _variants/shared_card_probe \
//         -Linverseproblemwithdiffusionmodel_amd -lipdm -Wl,-rpath,'$ORIGIN/../inverseproblemwithdiffusionmodel_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ipdm.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void count_diff(const unsigned* a, const unsigned* b, long long n, unsigned* cnt) {
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  unsigned c = 0;
  for (; i < n; i += st) c += a[i] != b[i];
  if (c) atomicAdd(cnt, c);
}

// victims: every thread folds R rounds of a few multiplies / adds of its own inputs into a checksum.
// MODE 0: scalar v_mul_f32 / v_add_f32; 1: v_pk_mul_f32 / v_pk_add_f32; 2: v_pk_fma_f32; 3: v_fma_f32; 4: integer only
template <int MODE>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, unsigned* out, int rounds) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float a0 = in[i * 4 + 0], a1 = in[i * 4 + 1], b0 = in[i * 4 + 2], b1 = in[i * 4 + 3];
  unsigned sum = 0;
  for (int r = 0; r < rounds; ++r) {
    float r0, r1;
    if (MODE == 0) {
      asm volatile("v_mul_f32 %0, %2, %4\n v_mul_f32 %1, %3, %5\n s_nop 1\n v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2"
                   : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
    } else if (MODE == 1) {
      f32x2 a = {a0, a1}, b = {b0, b1}, p, q;
      asm volatile("v_pk_mul_f32 %0, %2, %3\n s_nop 1\n v_pk_add_f32 %1, %0, %2 op_sel:[0,1] op_sel_hi:[1,0]"
                   : "=&v"(p), "=&v"(q) : "v"(a), "v"(b));
      r0 = q.x; r1 = q.y;
    } else if (MODE == 2) {
      f32x2 a = {a0, a1}, b = {b0, b1}, p;
      asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=&v"(p) : "v"(a), "v"(b));
      r0 = p.x; r1 = p.y;
    } else if (MODE == 3) {
      asm volatile("v_fma_f32 %0, %2, %4, %3\n v_fma_f32 %1, %3, %5, %2" : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
    } else {
      unsigned x = __builtin_bit_cast(unsigned, a0), y = __builtin_bit_cast(unsigned, b1);
      asm volatile("v_mul_lo_u32 %0, %2, %3\n v_add_u32 %1, %2, %3" : "=&v"(x), "=&v"(y) : "v"(x), "v"(y));
      r0 = __builtin_bit_cast(float, x & 0x3fffffffu); r1 = __builtin_bit_cast(float, y & 0x3fffffffu);
    }
    sum = sum * 31u + __builtin_bit_cast(unsigned, r0) + 7u * __builtin_bit_cast(unsigned, r1);
    a0 += 0.25f; b1 -= 0.125f;
  }
  out[i] = sum;
}

// synthetic MFMA aggressors: CHAIN dependent MFMAs on one accumulator, then the next accumulator; no memory traffic
template <int CHAIN>
__global__ __launch_bounds__(256, 2) void mfma_hog(float* sink, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int c = 0; c < CHAIN; ++c) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  if (s == 12345.678f) sink[0] = s;
}

// synthetic aggressor with the direct kernel's instruction mix: per step two 16-byte global loads (L2-resident weights), two
// ds_read_b128 and three chained f16 MFMAs whose operands ARE the loaded values
template <int NLOAD, int NDS>
__global__ __launch_bounds__(256, 2) void mix_hog(const uint4* __restrict__ wbuf, float* sink, int iters) {
  extern __shared__ __align__(16) uint4 mixl[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += 256) mixl[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x38003800u, 0x34003400u);
  __syncthreads();
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  uint4 a0 = wbuf[lane], a1 = wbuf[64 + lane], b0 = mixl[lane], b1 = mixl[64 + lane];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (NLOAD) { na0 = wbuf[((it * 4 + k) * 128 + lane) & 4095]; na1 = wbuf[((it * 4 + k) * 128 + 64 + lane) & 4095]; }
      if (NDS == 1) { nb0 = mixl[((it + k) * 64 + lane) & 2047]; nb1 = mixl[((it + k) * 64 + 1024 + lane) & 2047]; }
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b0), acc[k], 0, 0, 0);
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b1), acc[k], 0, 0, 0);
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0), acc[k], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
  }
  float sm = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) sm += acc[k][r];
  if (sm == 12345.678f) sink[0] = sm;
}

int main() {
  const int B = 48, C = 256, IH = 8;
  const long long n_in = (long long)B * C * IH * IH;
  const int NV = 1 << 20;
  std::vector<float> hx(n_in), hw((size_t)C * C * 9), hv((size_t)NV * 4);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = 0.02f * rnd();
  for (auto& v : hv) v = rnd();
  float *w, *cx, *cout_, *sink, *vin;
  unsigned *vout_, *vref_, *cnt;
  CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&cx, n_in * 4)); CK(hipMalloc(&cout_, n_in * 4)); CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&vin, hv.size() * 4)); CK(hipMalloc(&vout_, NV * 4)); CK(hipMalloc(&vref_, NV * 4)); CK(hipMalloc(&cnt, 4));
  CK(hipMemcpy(cx, hx.data(), n_in * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(vin, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
  void *pk_hx, *pk_bx;
  CK(hipMalloc(&pk_hx, ipdm_conv_hx2_weight_bytes(C, C, 3))); CK(hipMalloc(&pk_bx, ipdm_conv_bx3_weight_bytes(C, C, 3)));
  if (ipdm_conv_hx2_pack_weight(w, pk_hx, C, C, 3, nullptr) || ipdm_conv_bx3_pack_weight(w, pk_bx, C, C, 3, nullptr)) { printf("pack failed\n"); return 1; }
  CK(hipDeviceSynchronize());
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));

  // 64 x 64 shapes
  const int B2 = 8, C2 = 128, H2 = 64;
  const long long n2 = (long long)B2 * C2 * H2 * H2;
  std::vector<float> hx2v(n2), hw2((size_t)C2 * C2 * 9), hw1((size_t)C * C);
  for (auto& v : hx2v) v = rnd();
  for (auto& v : hw2) v = 0.03f * rnd();
  for (auto& v : hw1) v = 0.05f * rnd();
  float *x2, *o2, *w2, *w1, *wt32;
  CK(hipMalloc(&x2, n2 * 4)); CK(hipMalloc(&o2, n2 * 4)); CK(hipMalloc(&w2, hw2.size() * 4)); CK(hipMalloc(&w1, hw1.size() * 4));
  CK(hipMalloc(&wt32, hw.size() * 4));
  CK(hipMemcpy(x2, hx2v.data(), n2 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice));
  void *p1_hx, *p1_bx, *p2_hx, *p2_bx, *u_hx, *u_bx;
  CK(hipMalloc(&p1_hx, ipdm_conv_hx2_weight_bytes(C, C, 1))); CK(hipMalloc(&p1_bx, ipdm_conv_bx3_weight_bytes(C, C, 1)));
  CK(hipMalloc(&p2_hx, ipdm_conv_hx2_weight_bytes(C2, C2, 3))); CK(hipMalloc(&p2_bx, ipdm_conv_bx3_weight_bytes(C2, C2, 3)));
  CK(hipMalloc(&u_hx, ipdm_conv_wino_hx2_weight_bytes(C2, C2))); CK(hipMalloc(&u_bx, ipdm_conv_wino_bx3_weight_bytes(C2, C2)));
  int rc = 0;
  rc |= ipdm_conv_hx2_pack_weight(w1, p1_hx, C, C, 1, nullptr); rc |= ipdm_conv_bx3_pack_weight(w1, p1_bx, C, C, 1, nullptr);
  rc |= ipdm_conv_hx2_pack_weight(w2, p2_hx, C2, C2, 3, nullptr); rc |= ipdm_conv_bx3_pack_weight(w2, p2_bx, C2, C2, 3, nullptr);
  rc |= ipdm_conv_wino_hx2_pack_weight(w2, u_hx, C2, C2, nullptr); rc |= ipdm_conv_wino_bx3_pack_weight(w2, u_bx, C2, C2, nullptr);
  rc |= ipdm_conv_pack_weight_f32(w, wt32, C, C, 3, nullptr);
  if (rc) { printf("pack failed %d\n", rc); return 1; }
  CK(hipDeviceSynchronize());

  // third victim: the library's bilinear resize 8x8 -> 16x16 of 12288 planes
  const long long nb_out = (long long)B * C * 16 * 16;
  float *bout, *bref;
  CK(hipMalloc(&bout, nb_out * 4)); CK(hipMalloc(&bref, nb_out * 4));
  const char* vnames[] = {"v_mul/v_add f32", "v_pk_mul/v_pk_add f32", "ipdm_bilinear_f32 8->16"};
  const char* anames[] = {"none", "hx2 direct 3x3 8x8", "bx3 direct 3x3 8x8", "hx2 direct 1x1 8x8", "bx3 direct 1x1 8x8",
                          "hx2 direct 3x3 dil2 8x8", "bx3 direct 3x3 dil2 8x8", "hx2 direct 3x3 64x64", "bx3 direct 3x3 64x64",
                          "hx2 winograd 64x64", "bx3 winograd 64x64", "fp32 mfma conv 3x3 8x8", "mfma chain 3",
                          "synthetic mfma + global loads", "synthetic mfma + ds_read_b128", "synthetic mfma + both"};
  auto run_victim = [&](int v, unsigned* out) {
    const int rounds = 64;
    if (v == 0) hipLaunchKernelGGL(victim<0>, dim3(NV / 256), dim3(256), 0, sb, vin, out, rounds);
    else if (v == 1) hipLaunchKernelGGL(victim<1>, dim3(NV / 256), dim3(256), 0, sb, vin, out, rounds);
    else ipdm_bilinear_f32(cx, reinterpret_cast<float*>(out), B * C, IH, IH, 16, 16, 0, 0, sb);
  };
  for (int v = 2; v >= 0; --v) {
    unsigned* const vout = v == 2 ? reinterpret_cast<unsigned*>(bout) : vout_;
    unsigned* const vref = v == 2 ? reinterpret_cast<unsigned*>(bref) : vref_;
    const long long nchk = v == 2 ? nb_out : (long long)NV;
    run_victim(v, vref);
    CK(hipStreamSynchronize(sb));
    for (int g = 0; g < 16; ++g) {
      CK(hipMemsetAsync(cnt, 0, 4, sb));
      const int reps = 200;
      int arc = 0;
      for (int r = 0; r < reps; ++r) {
        switch (g) {
          case 1: arc |= ipdm_conv2d_hx2_f32(cx, pk_hx, nullptr, nullptr, 0, nullptr, cout_, nullptr, 0, B, C, C, IH, IH, 3, 1, nullptr, sa); break;
          case 2: arc |= ipdm_conv2d_bx3_f32(cx, pk_bx, nullptr, nullptr, 0, nullptr, cout_, nullptr, 0, B, C, C, IH, IH, 3, 1, nullptr, sa); break;
          case 3: arc |= ipdm_conv2d_hx2_f32(cx, p1_hx, nullptr, nullptr, 0, nullptr, cout_, nullptr, 0, B, C, C, IH, IH, 1, 1, nullptr, sa); break;
          case 4: arc |= ipdm_conv2d_bx3_f32(cx, p1_bx, nullptr, nullptr, 0, nullptr, cout_, nullptr, 0, B, C, C, IH, IH, 1, 1, nullptr, sa); break;
          case 5: arc |= ipdm_conv2d_hx2_f32(cx, pk_hx, nullptr, nullptr, 0, nullptr, cout_, nullptr, 0, B, C, C, IH, IH, 3, 2, nullptr, sa); break;
          case 6: arc |= ipdm_conv2d_bx3_f32(cx, pk_bx, nullptr, nullptr, 0, nullptr, cout_, nullptr, 0, B, C, C, IH, IH, 3, 2, nullptr, sa); break;
          case 7: arc |= ipdm_conv2d_hx2_f32(x2, p2_hx, nullptr, nullptr, 0, nullptr, o2, nullptr, 0, B2, C2, C2, H2, H2, 3, 1, nullptr, sa); break;
          case 8: arc |= ipdm_conv2d_bx3_f32(x2, p2_bx, nullptr, nullptr, 0, nullptr, o2, nullptr, 0, B2, C2, C2, H2, H2, 3, 1, nullptr, sa); break;
          case 9: arc |= ipdm_conv2d_wino_hx2_f32(x2, u_hx, nullptr, nullptr, o2, nullptr, 0, B2, C2, C2, H2, H2, 1, 0, nullptr, sa); break;
          case 10: arc |= ipdm_conv2d_wino_bx3_f32(x2, u_bx, nullptr, nullptr, o2, nullptr, 0, B2, C2, C2, H2, H2, 1, 0, nullptr, sa); break;
          case 11: arc |= ipdm_conv2d_f32(cx, wt32, nullptr, nullptr, 0, nullptr, cout_, nullptr, 0, B, C, C, IH, IH, 3, 1, 0, sa); break;
          case 12: hipLaunchKernelGGL(mfma_hog<3>, dim3(512), dim3(256), 0, sa, sink, 130); break;
          case 13: hipLaunchKernelGGL((mix_hog<1, 0>), dim3(512), dim3(256), 32768, sa, reinterpret_cast<const uint4*>(w), sink, 100); break;
          case 14: hipLaunchKernelGGL((mix_hog<0, 1>), dim3(512), dim3(256), 32768, sa, reinterpret_cast<const uint4*>(w), sink, 100); break;
          case 15: hipLaunchKernelGGL((mix_hog<1, 1>), dim3(512), dim3(256), 32768, sa, reinterpret_cast<const uint4*>(w), sink, 100); break;
          default: break;
        }
        run_victim(v, vout);
        hipLaunchKernelGGL(count_diff, dim3(512), dim3(256), 0, sb, vout, vref, nchk, cnt);
      }
      CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
      unsigned hc = 0;
      CK(hipMemcpy(&hc, cnt, 4, hipMemcpyDeviceToHost));
      printf("victim %-24s aggressor %-26s (rc %d): %u differing words over %d launches\n", vnames[v], anames[g], arc, hc, reps);
      fflush(stdout);
    }
  }
  return 0;
}
