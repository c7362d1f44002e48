/*
 * ipdm.h -- C ABI of libipdm.so: the MI355X (gfx950) kernels of the Annealed-Langevin-Dynamics
 * reconstruction hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless
 * a parameter is documented as host.  All launches are asynchronous on `stream` (a hipStream_t cast
 * to void*; NULL = the default stream), allocate nothing and never synchronise, so a caller may
 * capture any sequence of them into a hipGraph.
 *
 * Return value: 0 on success; a positive hipError_t if the runtime rejected a launch; a negative
 * IPDM_E* code if the arguments were rejected before anything was launched.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repository root).  INTEGRATION.md shows the ctypes / pybind stub a reference maintainer would add.
 *
 * Layouts: images are NCHW planar, row-major, float32.  Complex data is interleaved (re, im)
 * float32 pairs ("c64"), the memory layout of torch.complex64 / numpy.complex64.
 */
#ifndef IPDM_H
#define IPDM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPDM_OK 0
#define IPDM_EINVAL (-1)       /* bad size / NULL pointer / unsupported combination            */
#define IPDM_EUNSUPPORTED (-2) /* valid request this build has no kernel for (size limits)     */

/* activation codes used by the fused normalisation / convolution prologues */
#define IPDM_ACT_NONE 0
#define IPDM_ACT_ELU 1
#define IPDM_ACT_RELU 2
#define IPDM_ACT_LRELU02 3
#define IPDM_ACT_SWISH 4
#define IPDM_ACT_COPY 5           /* identity, as an `act_out` code: "produce the second output without an activation" (with
                                     ipdm_conv_ext_t.res_second: out = conv, out_act = conv + residual) */

/* "maxima vectors": per-image max |x| of an activation tensor (or an upper bound of it), what the f16x2 convolutions take as
 * `in_amax` (dynamic range) and what producers hand over (`out_amax`, `act_amax`, `amax_out`, `amax_bound` below).  Layout:
 * [B][IPDM_AMAX_SLOT] floats, 8 ways per image at a stride of 16 floats (one 64-byte line each: memory-side atomics serialise per
 * line); the value of an image is the max over its ways.  Atomic producers need the whole vector ZEROED by the caller;
 * ipdm_absmax_f32 and the coefficient kernels write every way themselves. */
#define IPDM_AMAX_WAYS 8
#define IPDM_AMAX_SLOT 128

/* library identification: returns IPDM_ABI_VERSION, writes the gfx arch string the kernels were built for */
#define IPDM_ABI_VERSION 4
int ipdm_abi_version(void);
const char* ipdm_build_arch(void);

/* ------------------------------------------------------------------------------------------------
 * StyleGAN2 resampling ops (reference: op/upfirdn2d.cpp:12-23 `upfirdn2d`, kernels
 * op/upfirdn2d_kernel.cu:49-369; op/fused_bias_act.cpp:11-21 `fused_bias_act`, kernel
 * op/fused_bias_act_kernel.cu:19-99).
 * ---------------------------------------------------------------------------------------------- */

/* in [major][in_h][in_w][minor] -> out [major][out_h][out_w][minor],
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kernel_h)/down_y + 1 (likewise out_w); zero-insert upsample,
 * pad (negative pad crops), correlate with the FLIPPED kernel, decimate.  kernel [kernel_h][kernel_w]. */
int ipdm_upfirdn2d_f32(const float* in, const float* kernel, float* out,
                       int major, int in_h, int in_w, int minor, int kernel_h, int kernel_w,
                       int up_x, int up_y, int down_x, int down_y,
                       int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* the other storage types of the reference's dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF, op/upfirdn2d_kernel.cu:311):
 * IEEE half (2-byte elements behind the void pointers; fp32 arithmetic, one rounding at the store) and double */
int ipdm_upfirdn2d_f16(const void* in, const void* kernel, void* out,
                       int major, int in_h, int in_w, int minor, int kernel_h, int kernel_w,
                       int up_x, int up_y, int down_x, int down_y,
                       int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
int ipdm_upfirdn2d_f64(const double* in, const double* kernel, double* out,
                       int major, int in_h, int in_w, int minor, int kernel_h, int kernel_w,
                       int up_x, int up_y, int down_x, int down_y,
                       int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* y[i] = act(x[i] + b[(i / step_b) % size_b]) * scale;  act*10+grad as in the reference kernel:
 * 10/11 linear, 12 zero, 30 leaky-relu(alpha), 31 leaky-relu gradient gated by ref, 32 zero.
 * b == NULL or size_b == 0: no bias.  ref == NULL: treated as zeros. */
int ipdm_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* y,
                            int64_t n, int step_b, int size_b, int act, int grad,
                            float alpha, float scale, void* stream);
/* half / double storage (op/fused_bias_act_kernel.cu:79 dispatches the same three types) */
int ipdm_fused_bias_act_f16(const void* x, const void* b, const void* ref, void* y,
                            int64_t n, int step_b, int size_b, int act, int grad,
                            float alpha, float scale, void* stream);
int ipdm_fused_bias_act_f64(const double* x, const double* b, const double* ref, double* y,
                            int64_t n, int step_b, int size_b, int act, int grad,
                            float alpha, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * k-space operators (reference: ncsn/linear_transforms/__init__.py:36-57 i2k_complex/k2i_complex;
 * ncsn/linear_transforms/undersampling_fourier.py:77-97 RandomUndersamplingFourier,
 * :140-170 SENSE.__call__/conj_op/SSOS; ncsn/models/proximal_op.py:19-51 L2Penalty, :72-94 SingleCoil).
 * mask is uint8 [mask_t][W] (1 = sampled k-space column); mask_t == 1 broadcasts over the batch,
 * otherwise image b uses row b % mask_t (the reference's (T,1,1,W) broadcast against a (T,1,H,W) stack).
 * sens maps are float32 [n_coils][H][W] (real, as the reference's "exp" maps).
 * ---------------------------------------------------------------------------------------------- */

/* centred orthonormal 2-D DFT of `batch` images [H][W] c64; inverse != 0 -> k2i_complex.  Any H, W
 * in [1, 1024]; power-of-two sizes with H*W <= 16384 take the single-launch LDS FFT. in may equal out. */
int ipdm_fft2c_c64(const float* in, float* out, int batch, int H, int W, int inverse,
                   float* workspace /* batch*H*W c64, only for the non-LDS path; may be NULL otherwise */,
                   void* stream);
/* bytes of workspace ipdm_fft2c_c64 needs for (batch,H,W); 0 when the LDS path applies */
int64_t ipdm_fft2c_workspace_bytes(int batch, int H, int W);

/* y[c][b] = mask * fft2c(S_c * x[b])        x [B][H][W] c64 -> y [n_coils][B][H][W] c64 */
/* (sens == NULL with n_coils == 1: the single-coil operator RandomUndersamplingFourier.__call__, y = M F x,
 *  undersampling_fourier.py:77-82) */
int ipdm_sense_forward_c64(const float* x, const float* sens, const uint8_t* mask, int mask_t,
                           float* y, int B, int n_coils, int H, int W, void* stream);
/* Scratch the SENSE / single-coil operators below need for (B, n_coils, H, W):
 *   power-of-two images up to 128x128 (image resident in LDS): n_coils*B*H*W*8 bytes for the proximal operators (one
 *   plane per (sample, coil): the coils of a sample run in parallel workgroups), nothing for forward / adjoint / SSOS;
 *   larger power-of-two images (e.g. the reference's 256x256 ACDC slices, helpers/load_data.py:274; row / column FFT
 *   passes): n_coils*B*H*W*8 bytes for every operator except ipdm_sense_forward_c64 (single-coil: n_coils = 1). */
int64_t ipdm_sense_workspace_bytes(int B, int n_coils, int H, int W);
/* x[b] = sum_c S_c * ifft2c(mask? mask*s : s)    apply_mask == 0 reproduces SENSE.conj_op; workspace: see above
 * (may be NULL up to 128x128) */
int ipdm_sense_adjoint_c64(const float* s, const float* sens, const uint8_t* mask, int mask_t,
                           int apply_mask, float* x, float* workspace, int B, int n_coils, int H, int W, void* stream);
/* out[b] = sqrt(sum_c |ifft2c(s[c][b])|^2)    float32 [B][H][W] */
int ipdm_sense_ssos_c64(const float* s, float* out, float* workspace, int B, int n_coils, int H, int W, void* stream);

/* L2Penalty with a SENSE operator, closed form of its one SGD step:
 *   x = z - coef * A^H(A z - y),   coef = 0.05 * (alpha/lamda) / (n_coils * W)  (computed by the caller)
 * z given as planar real / imaginary float32 [B][H][W] (the sampler's x_mod_real / x_mod_imag), result
 * written planar to out_re / out_im (may alias z_re / z_im). */
int ipdm_sense_l2prox_f32(const float* z_re, const float* z_im, const float* y, const float* sens,
                          const uint8_t* mask, int mask_t, float coef, float* out_re, float* out_im,
                          float* work /* ipdm_sense_workspace_bytes(B, n_coils, H, W) */, int B, int n_coils, int H, int W,
                          void* stream);

/* Per-iteration scalars held in DEVICE memory, so that a captured hipGraph of one iteration can be
 * replayed for every noise level without re-instantiation: the host (or a tiny kernel) rewrites the
 * struct between replays. */
typedef struct ipdm_sched_t {
  float step;        /* step_lr * (sigma / sigma_L)^2                       */
  float noise_scale; /* sqrt(2 * step)                                      */
  float coef;        /* proximal coefficient (see ipdm_sense_l2prox_f32)    */
  float sigma;       /* current noise level (informational)                 */
  int64_t step_id;   /* global iteration counter: Philox key                */
  float seg_scale;   /* segmentation-likelihood weight lh_weight / sigma (ALD_optimizers.py:283) */
  float reserved;    /* keeps the struct 32 bytes                            */
} ipdm_sched_t;

/* One fused Annealed-Langevin iteration tail for the SENSE sampler (reference:
 * ncsn/models/ALD_optimizers.py:238-247 + :288-327 + proximal_op.py:19-51):
 *   x_re += step*g_re + sqrt(2*step)*n_re ; x_im likewise ; x = L2prox(x_re + i x_im)
 * noise_re/noise_im NULL -> counter-based Philox4x32-10 normals keyed by (seed, sample id, step_id),
 * sample id = sample_offset + b, so results do not depend on how samples are sharded over GPUs. */
int ipdm_ald_sense_step_f32(float* x_re, float* x_im, const float* g_re, const float* g_im,
                            const float* noise_re, const float* noise_im,
                            float step, float noise_scale, uint64_t seed, int64_t sample_offset, int64_t step_id,
                            const ipdm_sched_t* dev_sched /* device; non-NULL overrides step/noise_scale/coef/step_id */,
                            const float* y, const float* sens, const uint8_t* mask, int mask_t, float coef,
                            float* work /* ipdm_sense_workspace_bytes(B, n_coils, H, W) */, int B, int n_coils, int H, int W,
                            void* stream);

/* Single-coil data-consistency operators (A = M F, RandomUndersamplingFourier, no coil maps) on planar real/imag
 * float32 [B][H][W], y [B][H][W] complex64; out may alias z.  mode:
 *   0  L2Penalty on a single-coil operator (proximal_op.py:19-51): x = z - coef * F^-1[M (M F z - y)],
 *      coef = 0.05 * (alpha/lamda) / B  (the .mean() of the reference's loss runs over the batch; computed by the caller)
 *   1  SingleCoil closed form (proximal_op.py:72-94): x = F^-1[(F z + coef*y) / (1 + coef*M)], coef = alpha/lamda
 *   2  RandomUndersamplingFourier.projection (undersampling_fourier.py:89-97) = Constrained.__call__ (proximal_op.py:62-69):
 *      x = F^-1[coef*y + (1-coef) M F z + (1-M) F z], coef = lamda */
int ipdm_singlecoil_prox_f32(const float* z_re, const float* z_im, const float* y, const uint8_t* mask, int mask_t,
                             float coef, int mode, float* out_re, float* out_im,
                             float* workspace /* ipdm_sense_workspace_bytes(B, 1, H, W); NULL allowed up to 128x128 */,
                             int B, int H, int W, void* stream);

/* One fused Annealed-Langevin iteration tail for the single-coil samplers (the reference's
 * scripts/acdc_inv_seg_sampling_keep_center_prox_real_imag.py:79-89 and cine_inv_sampling_keep_center_prox_real_imag.py:78-88
 * build ALDInvSegProximalRealImag on RandomUndersamplingFourier + get_proximal(...)): Langevin update of both planes,
 * then ipdm_singlecoil_prox_f32's operator `mode`, in place; scalars / noise as ipdm_ald_sense_step_f32. */
int ipdm_ald_singlecoil_step_f32(float* x_re, float* x_im, const float* g_re, const float* g_im,
                                 const float* noise_re, const float* noise_im,
                                 float step, float noise_scale, uint64_t seed, int64_t sample_offset, int64_t step_id,
                                 const ipdm_sched_t* dev_sched, const float* y, const uint8_t* mask, int mask_t,
                                 float coef, int mode, float* workspace /* as ipdm_singlecoil_prox_f32 */, int B, int H,
                                 int W, void* stream);

/* Langevin update alone (ALD_optimizers.py:117): x += step*g + noise_scale*noise ; noise NULL -> Philox. */
int ipdm_langevin_step_f32(float* x, const float* g, const float* noise, float step, float noise_scale,
                           uint64_t seed, int64_t sample_offset, int64_t step_id, const ipdm_sched_t* dev_sched,
                           int64_t n_samples, int64_t sample_elems, void* stream);

/* Philox normals exactly as the fused kernels draw them, for tests: out [n_samples][sample_elems] */
int ipdm_philox_normal_f32(float* out, uint64_t seed, int64_t sample_offset, int64_t step_id, int plane,
                           int64_t n_samples, int64_t sample_elems, void* stream);

/* HOST function (no GPU): the 128 random bits behind elements 4*quad..4*quad+3 of (seed, sample, step_id, plane).
 * key = seed, counter = (quad, step, sample, plane): distinct seeds / steps / samples never share a block
 * (replaces torch.randn_like's device generator, ALD_optimizers.py:238-241, by a sharding-invariant stream). */
int ipdm_philox_block_host(uint64_t seed, int64_t sample, int64_t step_id, int plane, uint32_t quad, uint32_t* out4);

/* ------------------------------------------------------------------------------------------------
 * Score-network glue (reference: ncsn/models/normalization.py:150-176 InstanceNorm2dPlus;
 * ncsn/models/layers.py:11-23 get_act, :62-83 CRPBlock max-pool, :165-184 MSFBlock bilinear sum,
 * :291-313 ConvMeanPool mean; ncsn/models/ncsnv2.py:270-271 input rescale, :295-297 sigma division).
 * ---------------------------------------------------------------------------------------------- */

/* per (b,c) plane statistics -> coef [B][C][3] = (mu, scale, shift) such that
 *   InstanceNorm2dPlus(x)[b,c,:,:] = (x - mu) * scale + shift
 * gamma/alpha/beta are the module's parameters ([C] each; beta may be NULL). Two launches. */
int ipdm_instnorm_plus_coef_f32(const float* x, const float* alpha, const float* gamma, const float* beta,
                                float* coef, int B, int C, int HW,
                                float* amax_bound /* NULL, or a maxima vector: an upper bound of max |normalised value| per image, from the
                                                     coefficients alone (|x - mu| * rstd < sqrt(HW)) -- the in_amax of the f16x2
                                                     convolution that reads the normalised tensor, at no pass over it */,
                                void* stream);
/* y = act((x - mu) * scale + shift) with coef from above; x may equal y */
int ipdm_affine_act_f32(const float* x, const float* coef, float* y, int B, int C, int HW, int act, void* stream);
/* y = act(x) elementwise */
int ipdm_act_f32(const float* x, float* y, int64_t n, int act, void* stream);
/* y = a*x + b (scalar a, b) */
int ipdm_scale_shift_f32(const float* x, float* y, int64_t n, float a, float b, void* stream);
/* out = x + y (out may alias either) */
int ipdm_add_f32(const float* x, const float* y, float* out, int64_t n, void* stream);
/* out[b,...] = x[b,...] / sigmas[labels[b]];  labels == NULL: out[b,...] = x[b,...] / sigmas[b] */
int ipdm_div_sigma_f32(const float* x, const float* sigmas, const int64_t* labels, float* out,
                       int B, int64_t sample_elems, void* stream);
/* 3x3 'same' convolution (stride 1, dilation 1, zero padding) with at most three input OR at most three output channels:
 * the first / last layer of every score network (nn.Conv2d begin_conv / end_conv, ncsn/models/ncsnv2.py:40,45; conv3x3 of
 * models/ncsnpp.py:143,226).  Streaming fp32 kernels on the vector ALU (the matrix-core kernels pad the thin side to an
 * MFMA operand): w is the reference's own layout [Cout][Cin][3][3], bias may be NULL; W % 4 == 0, 16-byte aligned x / out.
 * ipdm_conv3x3_thin_supported tells whether a shape is served (else IPDM_EUNSUPPORTED). */
int ipdm_conv3x3_thin_supported(int Cin, int Cout, int H, int W);
int ipdm_conv3x3_thin_f32(const float* x, const float* w, const float* bias,
                          const float* coef /* NULL, or [B][Cin][3] = (c0, c1, c2): the input is (x - c0) * c1 + c2 inside the
                                               image (few-input-channel form only: `h = 2x - 1`, ncsnv2.py:64) */,
                          float* out, int B, int Cin, int Cout, int H, int W, void* stream);
/* MaxPool2d(kernel 5, stride 1, padding 2) on [planes][H][W] */
int ipdm_maxpool5_f32(const float* x, float* y, int planes, int H, int W, void* stream);
/* 2x2 mean pooling (ConvMeanPool's tail) [planes][H][W] -> [planes][H/2][W/2]; H, W even */
int ipdm_meanpool2_f32(const float* x, float* y, int planes, int H, int W, void* stream);
/* bilinear resize, align_corners=True: out = act(resize(x) [+ out]); accumulate != 0 adds the previous out */
int ipdm_bilinear_f32(const float* x, float* out, int planes, int in_h, int in_w, int out_h, int out_w,
                      int accumulate, int act /* applied to the value written */,
                      int planes_per_image, float* amax_out /* NULL, or a maxima vector ([planes / planes_per_image] images) zeroed by the
                                                               caller: per-image max |value written| (atomic max), see ipdm_conv_ext_t */,
                      void* stream);
/* trilinear resize, align_corners=True, of [planes][D][H][W] volumes (the 3-D MSF block's F.interpolate,
 * ncsn/models/layers3d.py:185,214); same accumulate / act convention */
int ipdm_trilinear_f32(const float* x, float* out, int planes, int in_d, int in_h, int in_w, int out_d, int out_h,
                       int out_w, int accumulate, int act, int planes_per_image, float* amax_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * NCSN++ / predictor-corrector extras (reference: torch.nn.GroupNorm(eps 1e-6) in models/layerspp.py:66,219
 * and models/ncsnpp.py:194; torch.nn.Linear models/ncsnpp.py:85-90; AttnBlockpp.forward models/layerspp.py:75-91;
 * skip_rescale (x + h)/sqrt(2) models/layerspp.py:88-91,271-274; ReverseDiffusionPredictor / LangevinCorrector
 * update arithmetic sde/sampling.py:200-205, 267-287).
 * ---------------------------------------------------------------------------------------------- */
/* GroupNorm as per-(b,c) coefficients (mu, weight*rstd, bias) for ipdm_affine_act_f32 / the conv prologue */
int ipdm_groupnorm_coef_f32(const float* x, const float* weight, const float* bias, float* coef,
                            int B, int C, int HW, int G, float eps,
                            float* plane_amax /* may be NULL; [B][C]: max |x| per plane from the same pass (single-read
                                                 plane kernels only, IPDM_EUNSUPPORTED otherwise) */,
                            void* stream);
/* GroupNorm (+ activation) of torch.cat([x1, x2], dim=1) WITHOUT the concatenation (the up path of NCSN++ feeds every block
 * `torch.cat([h, hs.pop()], dim=1)`, models/ncsnpp.py:351): coefficients [B][C1+C2][3] from the two tensors' planes, then one
 * pass that writes the normalised, activated, concatenated tensor.  IPDM_EUNSUPPORTED outside the single-read plane kernels
 * (HW % 4 != 0, HW > 65536, more than 65535 planes): concatenate and use the one-tensor calls. */
/* ... from the statistics partials of the convolution(s) that produced the tensor(s) (the _stats_f32 convolution calls: [B][C][P][3]
 * = count, mean, sum of squared deviations per pixel block): the tensor is not read.  part2 / C2 (may be NULL / 0): the channels
 * >= C1 of torch.cat([x1, x2], dim=1). */
int ipdm_groupnorm_coef_partials_f32(const float* part1, int C1, const float* part2, int C2, int P, const float* weight,
                                     const float* bias, float* coef /* [B][C1+C2][3] */, int B, int G, float eps, void* stream);
int ipdm_groupnorm_coef_cat_f32(const float* x1, int C1, const float* x2, int C2, const float* weight, const float* bias,
                                float* coef, int B, int HW, int G, float eps, float* plane_amax /* may be NULL; [B][C1+C2] */,
                                void* stream);
int ipdm_affine_act_cat_f32(const float* x1, int C1, const float* x2, int C2, const float* coef, float* y, int B, int HW,
                            int act, void* stream);
/* y[b][o] = bias[o] + sum_i act(x[b][i]) * W[o][i]   (torch.nn.Linear weight layout [Out][In]) */
int ipdm_linear_f32(const float* x, const float* W, const float* bias, float* y, int B, int In, int Out,
                    int act, void* stream);
/* out[b,:,i] = sum_j softmax_j(scale * q[b,:,i].k[b,:,j]) v[b,:,j]   q,k,v,out [B][C][N] */
int ipdm_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int C, int N,
                       float scale, void* stream);
/* out = a*x + b*y */
int ipdm_axpby_f32(const float* x, const float* y, float* out, int64_t n, float a, float b, void* stream);
/* out[s] = x[s] + a[s]*y[s] + c[s]*z[s] with per-sample device coefficients a, c [n_samples]; z/c may be NULL */
int ipdm_sample_axpy2_f32(const float* x, const float* y, const float* z, const float* a, const float* c,
                          float* out, int n_samples, int64_t sample_elems, void* stream);
/* norms[s] = ||x[s]||_2 */
int ipdm_sample_norm_f32(const float* x, float* norms, int n_samples, int64_t sample_elems, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense 3x3 / 1x1 convolution, float32 MFMA implicit GEMM (reference: torch.nn.Conv2d call sites in
 * ncsn/models/layers.py:28-60 conv1x1 / conv3x3 / dilated_conv3x3, stride 1, padding = dilation*(k/2)).
 *   out[b,co] = bias[co] + sum_ci W[co,ci] * pre(x[b,ci])  (+ residual[b,co])
 *   pre(v) = act((v - mu)*scale + shift) when coef != NULL (fused InstanceNorm2dPlus), else act(v)
 * wt is the weight repacked by ipdm_conv_pack_weight_f32: [k*k][Cin][Cout].
 * out_act (optional) receives act_out(out): the activated copy the NEXT convolution consumes (RCU / CRP chains
 * apply the non-linearity before every conv; producing it once in the epilogue is cheaper than activating in
 * each consumer's prologue).  out may be NULL when only the activated copy is needed.
 * pool2 != 0 additionally applies the 2x2 mean (ConvMeanPool) in the epilogue: out is [B][Cout][H/2][W/2].
 * ---------------------------------------------------------------------------------------------- */
int ipdm_conv_pack_weight_f32(const float* w /* [Cout][Cin][k][k] */, float* wt, int Cout, int Cin, int k, void* stream);
int ipdm_conv2d_f32(const float* x, const float* wt, const float* bias, const float* coef, int act,
                    const float* residual, float* out, float* out_act, int act_out,
                    int B, int Cin, int Cout, int H, int W, int k, int dilation, int pool2, void* stream);

/* Winograd F(2x2,3x3) variant of the 3x3 / dilation-1 convolution: 16 multiply-adds per 4 outputs instead of 36
 * (2.25x fewer MFMA cycles), exact-fp32 fma chains in the transformed domain.  U = G g G^T is produced once per
 * layer by ipdm_conv_wino_weight_f32 ([16][Cin][Cout]).  ipdm_conv2d_wino_supported tells whether a shape is
 * eligible (Cin % 8 == 0, Cout % 64 == 0; wide images: even H, W, dilation 1; small or dilated images: H and W
 * divisible by 2*dilation -- a dilated convolution is run as dilation^2 interleaved undilated ones); otherwise use
 * ipdm_conv2d_f32. */
int ipdm_conv_wino_weight_f32(const float* w /* [Cout][Cin][3][3] */, float* U, int Cout, int Cin, void* stream);
int ipdm_conv2d_wino_supported(int Cin, int Cout, int H, int W, int dilation);
int ipdm_conv2d_wino_f32(const float* x, const float* U, const float* bias, const float* residual, float* out,
                         float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int dilation,
                         void* stream);

/* tuning aid: when set (device buffer of 4*n_blocks uint64), the Winograd kernel's workgroups write s_memtime stamps
 * (start, loop start, loop end, end); NULL (default) disables it */
int ipdm_debug_set_stamp_buffer(void* buf);

/* 3-D variant for the temporal prior (reference: nn.Conv3d(k=3, padding=dilation, dilation) call sites in
 * ncsn/models/layers3d.py and ncsn/models/ncsn3d.py:137,141): x [B][Cin][D][H][W], weights packed by
 * ipdm_conv_pack_weight_f32(k = 27) to [27][Cin][Cout] (k == 1: [1][Cin][Cout]).  Each depth slice is an
 * (H x W) image of the 2-D kernel; the three depth taps are three more passes over the K loop. */
int ipdm_conv3d_f32(const float* x, const float* wt, const float* bias, const float* coef, int act,
                    const float* residual, float* out, float* out_act, int act_out,
                    int B, int Cin, int Cout, int D, int H, int W, int k, int dilation, void* stream);

/* MaxPool3d(kernel 5, stride 1, padding 2) on [planes][D][H][W] (volumes up to 8192 voxels) */
int ipdm_maxpool3d5_f32(const float* x, float* y, int planes, int D, int H, int W, void* stream);
/* tap gather for the temporal (1,1,4)/stride-2 convolutions of NCSN3DShallow: x [planes][S][T_in] ->
 * out [planes][4][S][T_out]; mode 0 = strided conv (T_out = T_in/2), mode 1 = transposed conv (T_out = 2*T_in).
 * A 1x1 ipdm_conv2d_f32 over the 4*C gathered channels then IS the temporal convolution. */
int ipdm_temporal_taps_f32(const float* x, float* out, int planes, int S, int T_in, int T_out, int mode, void* stream);

/* Optional extras of the split-operand convolution calls below (`ext` may be NULL = all defaults).  A HOST struct, read when
 * the call is made (its device pointer member is dereferenced by the kernel):
 *   in_amax       f16x2 calls only: maxima vector of the input (ipdm_absmax_f32, or a producer's out_amax / act_amax / amax_out /
 *                 amax_bound) -> dynamic range, see the f16x2 note; NULL = static
 *   bias_bstride  bias index = image * bias_bstride + channel: Cout gives one bias ROW PER IMAGE (the reference's
 *                 `h += Dense_0(act(temb))[:, :, None, None]` after a convolution, models/layerspp.py:252-254, folded into the
 *                 epilogue); 0 = the usual per-channel bias
 *   out_amax / act_amax   see the struct
 *   out_scale     result = (conv + bias + residual) * out_scale -- the skip_rescale of the score_sde blocks,
 *                 (x + h) / sqrt(2) (models/layerspp.py:271-274), folded into the epilogue; 0 is read as 1 */
typedef struct {
  const float* in_amax;
  int bias_bstride;
  float out_scale;
  int res_second;    /* != 0: the residual enters the SECOND output only -- out = (conv + bias) * out_scale, out_act =          */
                     /* act_out((conv + bias + residual) * out_scale).  CRPBlock's `path = conv(pool(path)); x = path + x`          */
                     /* (ncsn/models/layers.py:76-83) as one launch: `path` and the running sum leave the same epilogue           */
  float* out_amax;   /* maxima vector ZEROED by the caller, or NULL: per-image max |out| of what the call stores, accumulated with  */
  float* act_amax;   /* atomic max (exact, order-independent) -- likewise for out_act.  What the NEXT f16x2 convolution takes as   */
                     /* in_amax: the dynamic range then costs no pass over the tensor (the reference has no counterpart: fp32)      */
} ipdm_conv_ext_t;

/* ---- fp32 convolution on the bf16 matrix cores ("bf16x3": exact three-way operand split, six
 * v_mfma_f32_32x32x16_bf16 per k-step, fp32 accumulation; fp32-faithful results at 2.67x the fp32 MFMA rate) ----
 * Same call sites, arguments and fused input / output options as ipdm_conv2d_f32 / ipdm_conv3d_f32; only the
 * packed-weight format differs: an opaque blob of ipdm_conv_bx3_weight_bytes(Cout, Cin, k) bytes written by
 * ipdm_conv_bx3_pack_weight (k = 1, 3, or 27 for a 3x3x3 kernel), valid for every batch / image size. */
int64_t ipdm_conv_bx3_weight_bytes(int Cout, int Cin, int k);
int ipdm_conv_bx3_pack_weight(const float* w /* [Cout][Cin][k][k] or [Cout][Cin][3][3][3] */, void* packed, int Cout,
                              int Cin, int k, void* stream);
int ipdm_conv2d_bx3_f32(const float* x, const void* packed, const float* bias, const float* coef, int act,
                        const float* residual, float* out, float* out_act, int act_out,
                        int B, int Cin, int Cout, int H, int W, int k, int dilation, const ipdm_conv_ext_t* ext, void* stream);
int ipdm_conv3d_bx3_f32(const float* x, const void* packed, const float* bias, const float* coef, int act,
                        const float* residual, float* out, float* out_act, int act_out,
                        int B, int Cin, int Cout, int D, int H, int W, int k, int dilation, const ipdm_conv_ext_t* ext, void* stream);

/* Split-K form of the two calls above for shapes whose tiles alone leave most of the chip idle (small images at
 * small batch): ipdm_conv_bx3_splitk returns how many K parts pay (1 = use the plain call); with ksplit > 1 the parts
 * write raw partial sums to `work` (ksplit * B * Cout * D * H * W floats) and a second pass adds them in fixed order
 * (deterministic) together with bias / residual / activation.  volume: 0 = 2-D weights (D = 1), 1 = 3-D weights. */
int ipdm_conv_bx3_splitk(int B, int D, int Cin, int Cout, int H, int W, int k, int dilation);
int ipdm_conv_bx3_splitk_f32(const float* x, const void* packed, const float* bias, const float* coef, int act,
                             const float* residual, float* out, float* out_act, int act_out,
                             int B, int Cin, int Cout, int D, int H, int W, int k, int dilation, int volume, int ksplit,
                             float* work, const ipdm_conv_ext_t* ext, void* stream);

/* One torch.optim.Adam step (no weight decay / amsgrad) in place on x along the ASCENT direction g, i.e. with
 * param.grad = -g as the reference's MAP optimizers hand it over (ncsn/models/MAP_optimizers.py:72-74,103-105);
 * m, v: first / second moment buffers (zero-initialised by the caller), step = 1, 2, ... */
int ipdm_adam_ascent_f32(float* x, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                         float eps, int step, void* stream);

/* Winograd F(2x2,3x3) with split-bf16 operands (3x3, Cin % 16 == 0, Cout % 64 == 0, see ..._supported): the same
 * call sites and output options as ipdm_conv2d_wino_f32; weights transformed, split and laid out once per layer into
 * a blob of ipdm_conv_wino_bx3_weight_bytes(Cout, Cin) bytes. */
int64_t ipdm_conv_wino_bx3_weight_bytes(int Cout, int Cin);
int ipdm_conv_wino_bx3_pack_weight(const float* w /* [Cout][Cin][3][3] */, void* U, int Cout, int Cin, void* stream);
int ipdm_conv2d_wino_bx3_supported(int Cin, int Cout, int H, int W, int dilation);
/* pool2 != 0 (wide images, dilation 1): ConvMeanPool (layers.py:291-313) in one launch -- out / out_act / residual are
 * [B][Cout][H/2][W/2] and hold the 2x2 mean of the convolution (+ residual, activation); IPDM_EUNSUPPORTED where the
 * pooled epilogue is not built (the caller then runs the convolution and ipdm_meanpool2_f32 separately) */
int ipdm_conv2d_wino_bx3_f32(const float* x, const void* U, const float* bias, const float* residual, float* out,
                             float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int dilation,
                             int pool2, const ipdm_conv_ext_t* ext, void* stream);
/* The same launch with the statistics epilogue: besides the result it writes, per (image, output channel), P partials
 * (count, mean, sum of squared deviations about that mean) of the stored result -- one per (tile block, tile group), a
 * function of the image geometry only -- to stats [B][Cout][P][3]; ipdm_instnorm_plus_coef_partials_f32 turns them into the
 * InstanceNorm++ coefficients of the FOLLOWING normalisation (normalization.py:163-176) without reading the tensor again.
 * ipdm_conv2d_wino_bx3_stats_partials returns P for a layer shape, 0 where the epilogue does not exist (small / dilated
 * images, W % 4 != 0): callers then use ipdm_instnorm_plus_coef_f32 on the tensor as before. */
/* Split-K form of the Winograd launch for 16-pixel layers with fewer than 512 output channels, whose (image, channel tile)
 * pairs alone leave most of the chip idle: ipdm_conv2d_wino_bx3_splitk returns the number of K parts for a layer shape (1: use
 * the plain call; a function of the shape only, never of the batch); with ksplit > 1 the parts write raw partial results to
 * `work` (ksplit * B * Cout * H * W floats) and a second pass adds them in fixed order with bias / residual / activation. */
int ipdm_conv2d_wino_bx3_splitk(int Cin, int Cout, int H, int W, int dilation);
int ipdm_conv2d_wino_bx3_splitk_f32(const float* x, const void* U, const float* bias, const float* residual, float* out,
                                    float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int dilation,
                                    int ksplit, float* work, const ipdm_conv_ext_t* ext, void* stream);
int ipdm_conv2d_wino_bx3_stats_partials(int Cin, int Cout, int H, int W, int dilation, int pool2);
int ipdm_conv2d_wino_bx3_stats_f32(const float* x, const void* U, const float* bias, const float* residual, float* out,
                                   float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int dilation,
                                   int pool2, float* stats, const ipdm_conv_ext_t* ext, void* stream);
/* ---- "f16x2": the split-operand kernels above with TWO fp16 pieces per fp32 operand and THREE v_mfma_f32_32x32x16_f16 per
 * product (fp32 accumulation) -- half the matrix-core work and two thirds of the weight bytes of the three-way bf16 split.
 * fp32-faithful (error against float64 at or below the bf16 split's and the exact-fp32 kernel's on the networks' layer
 * shapes) under a RANGE contract: weights are scaled per output channel by a power of two at pack time (the inverse scale
 * rides in the blob); activations must satisfy |x| < 65504 (the Winograd calls pre-scale their input transform to keep that bound) -- beyond that the result
 * is NaN / inf (never a wrong finite number); the bx3 calls keep the whole fp32 exponent range.  DYNAMIC RANGE: `ext->in_amax`
 * (NULL = the static contract above) is the maxima vector of the input (ipdm_absmax_f32 measures one; producers hand theirs
 * over, see ipdm_conv_ext_t); the kernels
 * then scale every image by the power of two that puts its maximum in [2^14, 2^15) and undo it in the epilogue (both exact), so
 * ANY fp32 input is in range and values down to 2^-17 of an image's maximum keep all 22 bits.  (Ignored -- pass NULL -- when the
 * call fuses an input normalisation / activation: the normalised values are what is split.)  Same other arguments, semantics
 * and replaced reference interface (torch.nn.Conv2d / Conv3d inside ncsn/models/layers.py:28-60) as their bx3 twins; the
 * shape rules (ipdm_conv_bx3_splitk, ipdm_conv2d_wino_bx3_supported / _splitk / _stats_partials) are shared. */
int ipdm_absmax_f32(const float* x, float* amax /* maxima vector [n_images][IPDM_AMAX_SLOT], zeroed by the call */, int n_images,
                    int64_t per_image, void* stream);
int64_t ipdm_conv_hx2_weight_bytes(int Cout, int Cin, int k);
int ipdm_conv_hx2_pack_weight(const float* w /* [Cout][Cin][k][k] or [Cout][Cin][3][3][3] */, void* packed, int Cout,
                              int Cin, int k, void* stream);
int ipdm_conv2d_hx2_f32(const float* x, const void* packed, const float* bias, const float* coef, int act,
                        const float* residual, float* out, float* out_act, int act_out,
                        int B, int Cin, int Cout, int H, int W, int k, int dilation, const ipdm_conv_ext_t* ext, void* stream);
int ipdm_conv3d_hx2_f32(const float* x, const void* packed, const float* bias, const float* coef, int act,
                        const float* residual, float* out, float* out_act, int act_out,
                        int B, int Cin, int Cout, int D, int H, int W, int k, int dilation, const ipdm_conv_ext_t* ext, void* stream);
int ipdm_conv_hx2_splitk_f32(const float* x, const void* packed, const float* bias, const float* coef, int act,
                             const float* residual, float* out, float* out_act, int act_out,
                             int B, int Cin, int Cout, int D, int H, int W, int k, int dilation, int volume, int ksplit,
                             float* work, const ipdm_conv_ext_t* ext, void* stream);
int64_t ipdm_conv_wino_hx2_weight_bytes(int Cout, int Cin);
int ipdm_conv_wino_hx2_pack_weight(const float* w /* [Cout][Cin][3][3] */, void* U, int Cout, int Cin, void* stream);
int ipdm_conv2d_wino_hx2_f32(const float* x, const void* U, const float* bias, const float* residual, float* out,
                             float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int dilation,
                             int pool2, const ipdm_conv_ext_t* ext, void* stream);
/* 1-D Winograd form of the same 3x3 convolution on f16x2 operands (conv_wino1d.hip): F(2,3) along x, the filter's three rows as
 * K -- 4 positions instead of 16, 128 output channels per workgroup, 12 instead of 16 weight values per (co, ci) and 1.5x the
 * matrix instructions.  Same arguments as ipdm_conv2d_wino_hx2_f32 without the dilation (undilated launches; Cin % 32 == 0,
 * Cout % 128 == 0, W % 4 == 0 and >= 32, H even and >= 8: ipdm_conv2d_wino1d_supported; activated copy: ELU or
 * IPDM_ACT_COPY); its own weight blob. */
int64_t ipdm_conv_wino1d_weight_bytes(int Cout, int Cin);
int ipdm_conv_wino1d_pack_weight(const float* w /* [Cout][Cin][3][3] */, void* U, int Cout, int Cin, void* stream);
int ipdm_conv2d_wino1d_supported(int Cin, int Cout, int H, int W);
/* 3x3x3 convolution of volumes [B][C][D][H][W] on the same kernel: the depth taps are further K chunks (36 weight positions; blob
 * from ipdm_conv_wino1d_pack_weight3d, w [Cout][Cin][3][3][3]); planes of 12 pixels or less go two depth slices per row block.
 * Undilated, Cin % 32 == 0, Cout % 128 == 0, W % 4 == 0 and (W <= 12 or W >= 16): ipdm_conv3d_wino1d_supported. */
int64_t ipdm_conv_wino1d_weight_bytes3d(int Cout, int Cin);
int ipdm_conv_wino1d_pack_weight3d(const float* w, void* U, int Cout, int Cin, void* stream);
int ipdm_conv3d_wino1d_supported(int Cin, int Cout, int D, int H, int W);
int ipdm_conv3d_wino1d_f32(const float* x, const void* U, const float* bias, const float* residual, float* out,
                           float* out_act, int act_out, int B, int Cin, int Cout, int D, int H, int W,
                           const ipdm_conv_ext_t* ext, void* stream);
/* coef != NULL: the input is act(InstanceNorm++(x)) -- coef [B][Cin][3] from ipdm_instnorm_plus_coef_f32, act = IPDM_ACT_ELU --
 * applied inside the kernel (as ipdm_conv2d_hx2_f32's fused input); ext->in_amax is then the coefficient kernel's bound. */
int ipdm_conv2d_wino1d_f32(const float* x, const void* U, const float* bias, const float* coef, int act,
                           const float* residual, float* out, float* out_act, int act_out, int B, int Cin, int Cout, int H,
                           int W, int pool2, const ipdm_conv_ext_t* ext, void* stream);
/* ... with the statistics epilogue (as ipdm_conv2d_wino_hx2_stats_f32): stats[B][Cout][P][3] = (count, mean, sum of squared
 * deviations) of `out` per 8 x 32 pixel block, P = ipdm_conv2d_wino1d_stats_partials (0: not served) */
int ipdm_conv2d_wino1d_stats_partials(int Cin, int Cout, int H, int W);
int ipdm_conv2d_wino1d_stats_f32(const float* x, const void* U, const float* bias, const float* coef, int act,
                                 const float* residual, float* out, float* out_act, int act_out, int B, int Cin, int Cout,
                                 int H, int W, int pool2, float* stats, const ipdm_conv_ext_t* ext, void* stream);
int ipdm_conv2d_wino_hx2_splitk_f32(const float* x, const void* U, const float* bias, const float* residual, float* out,
                                    float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int dilation,
                                    int ksplit, float* work, const ipdm_conv_ext_t* ext, void* stream);
int ipdm_conv2d_wino_hx2_stats_f32(const float* x, const void* U, const float* bias, const float* residual, float* out,
                                   float* out_act, int act_out, int B, int Cin, int Cout, int H, int W, int dilation,
                                   int pool2, float* stats, const ipdm_conv_ext_t* ext, void* stream);
int ipdm_instnorm_plus_coef_partials_f32(const float* partials, int P, const float* alpha, const float* gamma,
                                         const float* beta /* may be NULL */, float* coef /* [B][C][3] */, int B, int C,
                                         int HW /* elements per plane: only the bound needs it */, float* amax_bound /* as above */,
                                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * Segmentation-likelihood guidance (reference: ncsn/models/__init__.py:197-215 compute_seg_grad through a MONAI UNet --
 * stride-2 Convolution / transposed Convolution blocks with InstanceNorm + PReLU --, ALD_optimizers.py:272-286).
 * The strided (transposed) convolutions and their input-gradients run on ipdm_conv2d_bx3_f32 between these:
 * ---------------------------------------------------------------------------------------------- */

/* out [planes][2H][2W]: out[2i][2j] = x[i][j], zero elsewhere */
int ipdm_zero_insert2_f32(const float* x, float* out, int planes, int H, int W, void* stream);
/* out [planes][Ho][Wo] = x[2i + oy][2j + ox]: a stride-2 convolution is the stride-1 one sampled at (oy, ox) + 2(i, j) */
int ipdm_subsample2_f32(const float* x, float* out, int planes, int H, int W, int oy, int ox, int Ho, int Wo, void* stream);
/* per-plane InstanceNorm (biased variance, eps, no affine) + PReLU with one slope (slope NULL: identity):
 * xhat = (x - mean) * rstd, y = prelu(xhat); rstd [planes] and xhat are what the backward needs */
int ipdm_in_prelu_fwd_f32(const float* x, const float* slope, float* xhat, float* y, float* rstd, int planes, int HW,
                          float eps, void* stream);
/* input-gradient of ipdm_in_prelu_fwd_f32 */
int ipdm_in_prelu_bwd_f32(const float* gy, const float* xhat, const float* rstd, const float* slope, float* gx, int planes,
                          int HW, void* stream);
/* g = d/dlogits sum log softmax(logits)[label]: g[b][c][p] = [c == label[b][p]] - softmax_c; logits [B][C][HW], label int64 */
int ipdm_seg_loglh_grad_f32(const float* logits, const int64_t* label, float* g, int B, int C, int64_t HW, void* stream);
/* y += scale * x (* mask[i % mask_period], the "FG" mode); scale = dev_sched->seg_scale when dev_sched is non-NULL */
int ipdm_axpy_sched_f32(float* y, const float* x, const int64_t* mask, int64_t mask_period, const ipdm_sched_t* dev_sched,
                        float scale, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * On-device reporting (reference: helpers/metrics.py:21-102, helpers/visualizations.py:93,117,121).  Deterministic
 * float64 reductions; metrics are per image (one result per image in `out`, device float64).
 * ---------------------------------------------------------------------------------------------- */

/* out[i] = |x[i]| */
int ipdm_magnitude_c64(const float* x /* n complex64 */, float* out, int64_t n, void* stream);
/* samples [n_samples][HW] complex64 -> planes [7][HW] float64: sum |x|, sum |x|^2, sum angle, sum angle^2, sum Re,
 * sum Im, sum |angle| over the samples (compute_mean_and_std, helpers/metrics.py:77-92, as partial sums a shard can
 * all-reduce; the reference's phase std is np.std(np.abs(angle)), hence the last plane) */
int ipdm_posterior_moments_c64(const float* samples, double* planes, int n_samples, int64_t HW, void* stream);
/* NRMSE_wrapper (helpers/metrics.py:70-74): skimage normalized_root_mse(img, ref, "euclidean") = ||img - ref|| / ||img||,
 * img [n_images][elems]; ref [n_images][elems] or one [elems] image shared by all (ref_broadcast) */
int ipdm_nrmse_f32(const float* img, const float* ref, double* out, int n_images, int64_t elems, int ref_broadcast,
                   void* stream);
/* SSIM_wrapper (helpers/metrics.py:55-67) on single-channel [H][W] images: skimage structural_similarity defaults
 * (7x7 uniform window, K1 0.01, K2 0.03, sample covariance, mean over the valid interior) with an explicit data_range */
int ipdm_ssim_f32(const float* img, const float* ref, double* out, int n_images, int H, int W, int ref_broadcast,
                  double data_range, void* stream);

/* Total-variation baseline (reference: scripts/acdc_SENSE_TV.py:76-83 + ncsn/models/MAP_optimizers.py:26-52 MAPModel with
 * kornia.losses.TotalVariation): x, g complex64 [n_images][H][W] (interleaved re/im).
 * ipdm_tv_c64: out[b] = sum |x[i+1,j]-x[i,j]| + sum |x[i,j+1]-x[i,j]| (float64);
 * ipdm_tv_grad_c64: the gradient autograd gives for the complex parameter (unit difference vectors, 0 at a zero difference). */
int ipdm_tv_c64(const float* x, double* out, int n_images, int H, int W, void* stream);
int ipdm_tv_grad_c64(const float* x, float* g, int n_images, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IPDM_H */
