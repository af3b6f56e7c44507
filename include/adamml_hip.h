/* libadamml_hip.so -- C ABI of the MI355X-native AdaMML hot path (gfx950 only).
 *
 * The reference (IBM/AdaMML) has no FFI: its hot path is a chain of torch operators invoked from
 *   models/resnet.py:195-223, models/sound_mobilenet_v2.py:152-162, models/policy_net.py:142-149,312-373,
 *   models/common.py:28-33, models/joint_resnet_mobilenetv2.py:84-128, models/adamml.py:42-91.
 * Each entry point below names the operator call site(s) it replaces.  Conventions:
 *   - plain C types; every pointer is a DEVICE pointer valid on the current HIP device unless noted;
 *   - activations are NHWC bf16 with the channel count padded to a multiple of 8;
 *   - a "lazy" activation is (raw, scale, shift, act): value = act(scale[c]*raw + shift[c]); scale==NULL -> raw;
 *   - the caller owns all memory (no allocation inside); work is enqueued on `stream`, never synchronised;
 *   - return 0 on success, negative ADAMML_E* otherwise; adamml_last_error_string() describes the failure;
 *   - GROUPS: AdaMML calls every backbone once per segment with per-call BatchNorm statistics (models/adamml.py:84-86).
 *     All entry points that touch BatchNorm state take a `groups` count so that the S segment calls are ONE launch
 *     over [groups*N, H, W, C] tensors with per-group statistics / scale / shift -- same arithmetic, S times fewer
 *     launches and S times larger grids.  A `*_gstride` argument is the element stride between the groups of a
 *     per-channel vector (0 = one vector shared by all groups, e.g. eval-mode BatchNorm).
 */
#ifndef ADAMML_HIP_H
#define ADAMML_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;

/* The library is built with -fvisibility=hidden: the entry points declared here (ADAMML_API) are its whole dynamic symbol table
 * (tests/test_host_cpu.py::test_library_exports_every_declared_symbol checks both directions). */
#define ADAMML_API __attribute__((visibility("default")))

/* Per-channel reductions (BatchNorm statistics, BatchNorm-backward sums) are order-fixed and exact (csrc/common.h): inside a
 * workgroup every partial has one owner lane and is folded in a fixed order; across workgroups the fp32 workgroup partial is added
 * with ONE native 64-bit integer atomic per channel into exponent-indexed integer bins.  An accumulator is ADAMML_STAT_SLOTS = 32
 * such bins ([slot][2*C] 8-byte entries, zeroed by the caller), OPAQUE between the kernel that fills it and adamml_bn_finalize /
 * adamml_bn_bwd_finalize / adamml_stats_collapse, which decode the bins in a fixed order. */
#define ADAMML_STAT_SLOTS 32

#define ADAMML_ACT_NONE 0
#define ADAMML_ACT_RELU 1
#define ADAMML_ACT_RELU6 2

typedef struct {
    int32_t N, H, W, Cin;       /* input  [N,H,W,Cin]  (Cin padded to x8)            */
    int32_t OH, OW, Cout;       /* output [N,OH,OW,Cout]                              */
    int32_t KH, KW, stride, pad;
    int32_t up;                 /* internal: zero-upsampling of the input (dgrad); 1 for forward */
    int32_t act;                /* activation of the lazy INPUT transform             */
    int32_t accumulate;         /* epilogue adds into the existing output             */
    int32_t groups;             /* BatchNorm groups batched in one launch (0/1 = one): tensors are [groups*N, ...], statistic /
                                   sum buffers [groups][SLOTS][2C], BatchNorm vectors [groups][...] (see in_gstride)          */
    int32_t in_gstride;         /* element stride between the groups of in_scale / in_shift (0 = shared by all groups)        */
} adamml_conv_desc_t;

ADAMML_API int adamml_version(void);
ADAMML_API const char* adamml_last_error_string(void);
/* Reproducible reductions: every per-channel statistic / BatchNorm-backward sum is order-fixed inside a workgroup and accumulated
 * exactly across workgroups (integer bins: csrc/common.h), so two runs of the same step are bit-identical.  This is the ONLY mode
 * and the library keeps no mutable state; the two entry points remain for callers written against the switch of earlier versions:
 * adamml_set_deterministic(1) and (0) both return ADAMML_OK and change nothing -- a `try: set(1) ... finally: set(0)` caller keeps
 * working; (0) prints one warning to stderr per process --, adamml_get_deterministic() == 1 always.  The statistic buffers
 * ([groups][ADAMML_STAT_SLOTS][2C] doubles, zeroed by the caller) are OPAQUE between the kernel that fills them and
 * adamml_bn_finalize / adamml_bn_bwd_finalize / adamml_stats_collapse, which decode them. */
ADAMML_API int adamml_set_deterministic(int on);
ADAMML_API int adamml_get_deterministic(void);

/* nn.Conv2d(bias=False) forward (models/resnet.py:37-43,138; sound_mobilenet_v2.py:37,60; policy_net.py:40,48,76,84)
 * fused with the PRODUCER's BatchNorm+ReLU/ReLU6 on load and with the per-channel sum / sum-of-squares of its own
 * output (stats[slot][0..C) = sum, [slot][C..2C) = sumsq, fp64 [ADAMML_STAT_SLOTS][2C], caller zeroes) for this layer's
 * train-mode BatchNorm. */
ADAMML_API int adamml_conv_fwd(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale,
                    const float* in_shift, void* y, double* stats, hipStream_t stream);
/* 1 when adamml_conv_fwd applies the lazy input transform of this KxK conv ONCE per element (LDS-resident input patch:
 * 3x3 / stride 1 / pad 1 / 64 -> 64 channels), so the caller need not materialise the normalised input first; 0 when the
 * implicit-GEMM loader would re-apply it once per tap. */
ADAMML_API int adamml_conv_fused_input_supported(const adamml_conv_desc_t* d);
/* 1 when the 1x1 conv of FORWARD descriptor d is served by the barrier-free streaming kernels of the narrow MobileNetV2 layers
 * (csrc/conv1x1_narrow.hip; models/policy_net.py:63-95, models/sound_mobilenet_v2.py:43-69) instead of the tile-loop implicit GEMM,
 * for kind 0: adamml_conv_fwd, 1: adamml_conv_bwd_weight (with a workspace), 2: adamml_conv_bwd_data_dual, 3: adamml_conv_bwd_data
 * without accumulation, 4: adamml_conv_bwd_data_bn / adamml_conv_bwd_data accumulating.  Results are identical either way; callers
 * use it to label launches with the device kernel that runs (bench.py's roofline groups launches by kernel). */
ADAMML_API int adamml_conv1x1_narrow_supported(const adamml_conv_desc_t* d, int kind);
/* The same question for the activation-stationary streaming kernels of the WIDE 1x1 convs of ResNet-50 layers 3-4 (csrc/conv1x1_wide.hip;
 * models/resnet.py:94-113 at the widths of models/resnet.py:150-154): 1 when the launch of FORWARD descriptor d behind kind 0:
 * adamml_conv_fwd, 3: adamml_conv_bwd_data without accumulation, 4: adamml_conv_bwd_data accumulating runs there instead of on the
 * tile-loop implicit GEMM (other kinds: 0).  Outputs are bit-identical either way (same K order and rounding points); the forward
 * statistics differ in summation order only.  ADAMML_WIDE_STREAM=0 in the environment disables the kernels (A/B aid; read at every call). */
ADAMML_API int adamml_conv1x1_wide_supported(const adamml_conv_desc_t* d, int kind);
/* Forward 1x1 / stride-1 conv with BatchNorm + residual add + activation in its epilogue (a bottleneck's conv3 + bn3 + add +
 * ReLU, models/resnet.py:104-112; a MobileNetV2 projection + add) -- for the cases where the BatchNorm vectors are known before
 * the launch: eval mode, or train mode with statistics from adamml_gram_stats.  bn_vec [groups][4][Cout] (scale, shift, ..) of
 * THIS conv's BatchNorm; idn [same shape as out] or NULL, lazily normalised with id_scale / id_shift [Cout] (group stride
 * id_gstride floats) when given; out = act(scale*z + shift + idn'), mask_out (optional) = 1 bit per element, act'(out) != 0.
 * The raw conv output z is never written. */
ADAMML_API int adamml_conv_fwd_bn_add_supported(const adamml_conv_desc_t* d);
/* 1 when an adamml_conv_fwd_bn_add launch WITH an identity operand is served by the barrier-free streaming kernel of
 * csrc/conv1x1_fadd_stream.hip (the ResNet-50 layer-2 shape: 128 -> 512 channels, >= 4096 pixels per group; out and mask_out
 * bit-identical) rather than the tile kernel -- a label for profilers.  ADAMML_FADD_STREAM=0 disables it (read at every call). */
ADAMML_API int adamml_conv_fwd_bn_add_streams(const adamml_conv_desc_t* d);
ADAMML_API int adamml_conv_fwd_bn_add(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                           const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                           void* out, uint8_t* mask_out, hipStream_t stream);
/* conv3 + bn3 + residual add + activation of a bottleneck AND the 1x1 conv1 of the NEXT bottleneck in one barrier-free streaming
 * kernel (csrc/conv1x1_fadd_next.hip; models/resnet.py:104-112 then :94-96 of the next block): the block output `out` (+ mask_out) is
 * written as by adamml_conv_fwd_bn_add, and its tile -- still in LDS -- is multiplied with w_next [next_cout][Cout] (forward pack) into
 * y_next [same pixels][next_cout] with the per-channel sums of the stored values in stats_next ([groups][SLOTS][2*next_cout], caller
 * zeroes; may be NULL), exactly what adamml_conv_fwd(out -> y_next) would produce (bit-identical y_next), without re-reading `out`.
 * ResNet-50 layer 1 only: Cin = 64, Cout = 256, next_cout = 64 (the weights of both convs are LDS-resident). */
ADAMML_API int adamml_conv_fwd_bn_add_next_supported(const adamml_conv_desc_t* d, int next_cout);
ADAMML_API int adamml_conv_fwd_bn_add_next(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                                void* out, uint8_t* mask_out, const void* w_next, void* y_next, double* stats_next, hipStream_t stream);
/* The same conv + BatchNorm + add + ReLU when the block output feeds ONLY a temporal max-pool (the last block of a ResNet stage:
 * models/resnet.py:205-209 -> models/common.py:4-33, kernel 3 / stride 2 / pad 1 over the `frames` frames of a clip; d->N = clips * frames
 * per group): the epilogue pools over the frames and writes pooled [groups * clips * frames / 2][OH*OW][Cout] and, when `code` is given,
 * 2 bits per pooled element (one uint16 per 8 channels: window tap 0..2 of the FIRST maximum, as nn.MaxPool3d, or 3 when the maximum
 * is <= 0, i.e. the ReLU passes no gradient) for adamml_temporal_pool_bwd_code.  The full-rate block output, its activation mask and the
 * pool's own pass (adamml_temporal_pool_fwd) never touch HBM.  frames in {2, 4, 8}, Cout % 128 == 0, act = ReLU, idn required. */
ADAMML_API int adamml_conv_fwd_bn_add_tpool_supported(const adamml_conv_desc_t* d, int frames, int act, int lazy_input);
/* which device kernel serves the launch (a label for profilers): 0 = the tile kernel, 1 = the streaming kernel of csrc/conv1x1_fadd_next.hip
 * (64 -> 256 channels), 2 = the wave-slice streaming kernel of csrc/conv1x1_fadd_stream.hip (128 -> 512); same pooled tensor and codes. */
ADAMML_API int adamml_conv_fwd_bn_add_tpool_streams(const adamml_conv_desc_t* d, int frames);
ADAMML_API int adamml_conv_fwd_bn_add_tpool(const adamml_conv_desc_t* d, const void* x, const void* w_packed, const float* in_scale, const float* in_shift,
                                 const float* bn_vec, const void* idn, const float* id_scale, const float* id_shift, int id_gstride, int act,
                                 int frames, void* pooled, uint16_t* code, hipStream_t stream);
/* Train-mode BatchNorm statistics of z = W a without z: sums[g][co] = W[co,:] . s_g, sums[g][Cout+co] = W[co,:] G_g W[co,:]^T from
 * the Gram matrix G [groups][Cin][Cin] = a^T a and the column sums s [groups][Cin] of the conv INPUT (fp32), W = the bf16
 * forward pack [Cout][Cin].  sums [groups][2*Cout] doubles: pass to adamml_bn_finalize with nslots = 1. */
ADAMML_API int adamml_gram_stats(const void* w_packed, const float* G, const float* s, double* sums, int Cout, int Cin, int groups, hipStream_t stream);
/* G [groups][C][C] = a^T a and s [groups][C] = sum_p a over the P pixels of each group, a = act(scale x + shift) rounded to bf16
 * as the conv loaders stage it (scale == NULL: a = x), in one streaming pass over x (csrc/gram.hip) -- the inputs of
 * adamml_gram_stats and of the algebraic BatchNorm backward (models/resnet.py:103-111: bn3 statistics without storing conv3's
 * output).  C in {64, 128}; workspace of adamml_gram_colsum_workspace() bytes (per-workgroup partials, summed in a fixed order). */
ADAMML_API int adamml_gram_colsum_supported(int C);
ADAMML_API size_t adamml_gram_colsum_workspace(size_t P, int C, int groups);
ADAMML_API int adamml_gram_colsum(const void* x, const float* scale, const float* shift, int gstride, int act, float* G, float* s, size_t P, int C,
                       int groups, void* workspace, size_t workspace_bytes, hipStream_t stream);
/* autograd of adamml_conv_fwd w.r.t. its input (d = forward descriptor; w packed with mode 1) */
ADAMML_API int adamml_conv_bwd_data(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx,
                         int accumulate, hipStream_t stream);
/* same, when dx is the gradient w.r.t. a lazily normalised tensor act(BN(z_in)) with a single consumer: the epilogue
 * multiplies by act'(scale*z_in+shift), stores g' and accumulates the BatchNorm-backward sums (sum g', sum g'*zhat) into
 * sums[groups][ADAMML_STAT_SLOTS][2*Cin]; bn_vec = [groups][4][Cin] (scale, shift, mean, invstd).  adamml_bn_bwd_reduce is
 * then skipped. */
ADAMML_API int adamml_conv_bwd_data_bn(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx, const void* z_in,
                            const float* bn_vec, int act, double* sums, hipStream_t stream);

/* Data gradient of a 1x1 / stride-1 conv fused with the BatchNorm backward "apply" pass of its own output's BatchNorm
 * (activation NONE: bn3 / downsample BN of a bottleneck, the linear bottleneck of MobileNetV2): the loader reads the
 * masked gradient g and the raw conv output z and forms dz = A g + B z + C per channel (aff [groups][3][Cout] from
 * adamml_bn_bwd_affine) on the way into LDS, so adamml_bn_bwd_apply (read g, z; write dz) never runs for this layer;
 * dz_side (optional, same shape as g) receives dz once for adamml_conv_bwd_weight.  z_in / bn_vec / act / sums (all or none):
 * the BatchNorm-fused epilogue of adamml_conv_bwd_data_bn. */
ADAMML_API int adamml_conv_bwd_data_dual_supported(const adamml_conv_desc_t* d);   /* 1x1, stride 1, Cout <= 512 (where it pays) */
ADAMML_API int adamml_bn_bwd_affine(const float* coef, const float* vec, float* aff, int C, int groups, hipStream_t stream);
ADAMML_API int adamml_conv_bwd_data_dual(const adamml_conv_desc_t* d, const void* g, const void* z, const float* aff, void* dz_side,
                              const void* w_dgrad_packed, void* dx, int accumulate, const void* z_in, const float* bn_vec, int act,
                              double* sums, hipStream_t stream);

/* ---- algebraic BatchNorm backward through a 1x1 conv z = W a followed by a linear BatchNorm (bn3 / downsample of a bottleneck).
 * With dz = A g' + B z + C per channel (adamml_bn_bwd_affine) and z = W a:
 *     dx = (W^T diag(A)) g' + (W^T diag(B) W) a + W^T C        dW = A (.) (g'^T a) + B (.) (W G) + C (x) s,   G = a^T a, s = sum_p a
 * so neither the block-output-sized z nor dz is read or written in backward (7 -> 5 passes over that tensor per bottleneck).
 *   adamml_conv_bwd_weight_grouped: per-group products out[g] = dz_g^T x_g (OVERWRITTEN; dz may be lazy too: Gram matrix with dz = x)
 *   adamml_alg_pack:      w_alg [groups][Cin][Cout + Cin] bf16 and epi_add [groups][Cin] from W (fp32 [Cout][Cin]) and aff
 *   adamml_conv_bwd_data_alg: ONE GEMM per pixel tile over the concatenated input [g' | a] (a with its lazy transform, d->act /
 *                          d->in_gstride), + epi_add, then the usual accumulate or BatchNorm-fused epilogue
 *   adamml_alg_wgrad_combine: dW += sum_g A_g (.) P_g + B_g (.) (W G_g) + C_g (x) s_g */
ADAMML_API int adamml_conv_bwd_weight_grouped(const adamml_conv_desc_t* d, const void* dz, const float* dz_scale, const float* dz_shift, int dz_act,
                                   int dz_gstride, const void* x, const float* in_scale, const float* in_shift, float* out, int cin_true,
                                   void* workspace, size_t workspace_bytes, hipStream_t stream);
ADAMML_API int adamml_lazy_colsum(const void* x, const float* scale, const float* shift, int gstride, int act, float* s, size_t P, int C, int groups,
                       hipStream_t stream);     /* s[g][c] = sum_p act(scale x + shift), overwritten */
ADAMML_API int adamml_alg_sumfix(const float* w, const float* P, const float* vec, double* sums, int Cout, int Cin, int groups, hipStream_t stream);
   /* sum(g' zhat) from P = g'^T a when the producer of g' ran with z == NULL (adamml_conv_bwd_data_res / adamml_temporal_pool_bwd_res
      accept z_a == NULL: sum(g') only) */
/* m_pre (optional) = the [groups][Cin][Cin] products W^T diag(B_g) W and wg_pre (optional) = [Cout][groups*Cin] products W G_g when the
 * caller computed them with adamml_gemm_f32 (large Cin); NULL: formed inside the kernels. */
ADAMML_API int adamml_alg_pack(const float* w, const float* aff, const float* m_pre, void* w_alg, float* epi_add, int Cout, int Cin, int groups,
                    hipStream_t stream);
ADAMML_API int adamml_alg_wgrad_combine(const float* w, const float* aff, const float* P, const float* G, const float* wg_pre, const float* s, float* dw,
                             int Cout, int Cin, int groups, hipStream_t stream);
ADAMML_API int adamml_conv_bwd_data_alg(const adamml_conv_desc_t* d, const void* g, const void* a, const float* a_scale, const float* a_shift,
                             const void* w_alg, const float* epi_add, void* dx, int accumulate, const void* z_in, const float* bn_vec,
                             int act, double* sums, hipStream_t stream);

/* Data gradient of a 1x1 / stride-1 conv whose INPUT is the output of a residual add  out = act(bn_a(z_a) + idn)
 * (models/resnet.py:110-111; sound_mobilenet_v2.py:67), finishing that add's backward in the epilogue:
 *   g' = (W^T dz [+ dx, when accumulate: the identity-path gradient already stored there]) * act'(res_out)
 * is written to dx, and sum(g'), sum(g' * zhat_a) go to sums_a ([groups][SLOTS][2*Cin], caller zeroes); when the add's
 * other operand is BatchNorm'd too (downsample branch) z_b / vec_b / sums_b receive the same for it (else all NULL).
 * res_mask (optional): the 1-bit-per-element act'(res_out) != 0 mask written by adamml_bn_act_add_mask; when given it is read
 * instead of res_out (1/16 of the bytes).
 * Replaces adamml_conv_bwd_data(accumulate) + adamml_residual_bwd: the block-output gradient is written once. */
ADAMML_API int adamml_conv_bwd_data_res_supported(const adamml_conv_desc_t* d);
/* 1 when an adamml_conv_bwd_data_res launch in the algebraic backward's form (accumulate, res_mask, z_a == z_b == NULL) is served by
 * the barrier-free streaming kernel of csrc/res_prod_stream.hip (the ResNet-50 layer-2 shape: d->Cin == 512, d->Cout == 128, >= 4096
 * pixels per group; dx bit-identical) rather than the tile kernel -- a label for profilers.  ADAMML_RES_PROD_STREAM=0 disables it. */
ADAMML_API int adamml_conv_bwd_data_res_streams(const adamml_conv_desc_t* d);
ADAMML_API int adamml_conv_bwd_data_res(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx,
                             int accumulate, const void* res_out, const uint8_t* res_mask, int res_act, const void* z_a, const float* vec_a,
                             double* sums_a, const void* z_b, const float* vec_b, double* sums_b, hipStream_t stream);

/* adamml_conv_bwd_data_res in the form the algebraic BatchNorm backward uses it (accumulate onto the identity-path gradient in dx, 1-bit
 * mask, sum(g') only) which ALSO accumulates the per-group product  prod[g][c_out][c] = sum_p g'[p][c_out] * a[p][c]  of the gradient
 * tile it has just formed with a second, lazily normalised tensor a [pixels][a_channels] (value act(a_scale*a + a_shift), group stride
 * a_gstride): a = the input of the conv whose OUTPUT gradient g' is (models/resnet.py:103-111: conv3 of the previous bottleneck), i.e.
 * the g'^T a that adamml_conv_bwd_weight_grouped would compute in a separate pass over g' and a.  a_channels == 64, d->Cin % 128 == 0;
 * workspace of adamml_conv_bwd_data_res_prod_workspace() bytes; prod is overwritten. */
ADAMML_API int adamml_conv_bwd_data_res_prod_supported(const adamml_conv_desc_t* d, int a_channels);
ADAMML_API size_t adamml_conv_bwd_data_res_prod_workspace(const adamml_conv_desc_t* d);
/* 1 when the launch is served by the barrier-free streaming kernel of csrc/res_prod_stream.hip (the layer-1 shape: d->Cin == 256,
 * d->Cout == 64, a_channels == 64; same results: dx bit-identical) rather than the tile kernel -- a label for profilers, as
 * adamml_conv1x1_narrow_supported. */
ADAMML_API int adamml_conv_bwd_data_res_prod_streams(const adamml_conv_desc_t* d, int a_channels);
ADAMML_API int adamml_conv_bwd_data_res_prod(const adamml_conv_desc_t* d, const void* dz, const void* w_dgrad_packed, void* dx, const uint8_t* res_mask,
                                  int res_act, double* sums_a, const void* a, const float* a_scale, const float* a_shift, int a_act,
                                  int a_gstride, int a_channels, float* prod, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* autograd w.r.t. the weight: dw (fp32 OIHW, cin_true input channels) += dz^T * im2col(act(x)).  The pixel axis is
 * split over workgroups; with a workspace of adamml_conv_bwd_weight_workspace() bytes the partial tiles are written
 * with plain stores and summed by a second launch (device-scope fp32 atomics run at ~20 G/s on MI355X and would
 * otherwise bound the kernel); workspace == NULL falls back to atomic accumulation. */
ADAMML_API size_t adamml_conv_bwd_weight_workspace(const adamml_conv_desc_t* d, int cin_true);
ADAMML_API int adamml_conv_bwd_weight(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale,
                           const float* in_shift, float* dw, int cin_true, void* workspace, size_t workspace_bytes,
                           hipStream_t stream);
/* fp32 OIHW master weight -> bf16 GEMM operand.  mode 0: [Cout][KH*KW][cin_pad] (forward);
 * mode 1: [cin_pad][KH*KW flipped][Cout] (data gradient); mode 2: depthwise [KH*KW][C] fp32. */
ADAMML_API int adamml_pack_conv_weight(const float* w, void* out, int cout, int cin_true, int cin_pad, int kh, int kw, int mode,
                            hipStream_t stream);

/* ResNet stem fast path (models/resnet.py:138-139): 7x7 stride-2 pad-3 conv of a <=4-channel image (NHWC, channels padded
 * to 8) to 64 channels from an LDS-resident input patch, K = 7*8*4 = 224 instead of 49*8 = 392, with the statistics of
 * adamml_conv_fwd.  w_stem_packed = bf16 [64][7][8][4] from adamml_pack_stem_weight (fp32 OIHW master weight).  The plain
 * input only (no lazy BatchNorm transform); adamml_conv_stem_supported() tells whether a descriptor qualifies, otherwise
 * callers use adamml_conv_fwd (the stem has no data gradient). */
ADAMML_API int adamml_conv_stem_supported(const adamml_conv_desc_t* d);
ADAMML_API int adamml_pack_stem_weight(const float* w, void* out, int cout, int cin_true, hipStream_t stream);
ADAMML_API int adamml_conv_stem_fwd(const adamml_conv_desc_t* d, const void* x, const void* w_stem_packed, void* y, double* stats,
                         hipStream_t stream);
/* weight gradient of the same conv: dw (fp32 OIHW, cin_true channels) += dz^T * im2col(x), pixels as the MFMA reduction
 * dimension, im2col fragments read transposed from the LDS input patch; workspace >= adamml_conv_stem_bwd_weight_workspace */
ADAMML_API size_t adamml_conv_stem_bwd_weight_workspace(const adamml_conv_desc_t* d);
ADAMML_API int adamml_conv_stem_bwd_weight(const adamml_conv_desc_t* d, const void* dz, const void* x, float* dw, int cin_true,
                                void* workspace, size_t workspace_bytes, hipStream_t stream);

/* depthwise 3x3 conv (groups == channels; sound_mobilenet_v2.py:58, policy_net.py:66,80) */
ADAMML_API int adamml_dwconv_fwd(const adamml_conv_desc_t* d, const void* x, const float* w_tapmajor, const float* in_scale,
                      const float* in_shift, void* y, double* stats, hipStream_t stream);
ADAMML_API int adamml_dwconv_bwd_data(const adamml_conv_desc_t* d, const void* dz, const float* w_tapmajor, void* dx,
                           int accumulate, hipStream_t stream);
/* adamml_dwconv_bwd_data whose output is the gradient w.r.t. the ACTIVATED value of a lazily normalised tensor z_in (the expansion
 * conv's BatchNorm + ReLU6 output, sound_mobilenet_v2.py:52-57 / policy_net.py:72-79): the mask act'(bn(z_in)) is applied before the
 * store and sum(g'), sum(g' zhat) are accumulated into sums [groups][SLOTS][2C] -- replaces adamml_bn_bwd_reduce over (g, z_in).
 * pad 1, stride 1 or 2, no accumulation. */
ADAMML_API int adamml_dwconv_bwd_data_bn_supported(const adamml_conv_desc_t* d);
ADAMML_API int adamml_dwconv_bwd_data_bn(const adamml_conv_desc_t* d, const void* dz, const float* w_tapmajor, void* dx, const void* z_in,
                              const float* bn_vec, int act, double* sums, hipStream_t stream);
ADAMML_API size_t adamml_dwconv_bwd_weight_workspace(const adamml_conv_desc_t* d);
/* The whole backward of a stride-1 depthwise conv + train-mode BatchNorm + ReLU6 in one pass (models/sound_mobilenet_v2.py:58-61,
 * models/policy_net.py:80-83; replaces adamml_bn_bwd_apply + adamml_dwconv_bwd_weight + adamml_dwconv_bwd_data_bn: four passes over the
 * block's widest tensor instead of eight).  g = gradient w.r.t. the ACTIVATED depthwise output, already masked by its activation (the
 * projection's data gradient does that), z = the raw depthwise output, aff = [groups][3][C] from adamml_bn_bwd_affine
 * (dz = A g + B z + C, rounded to bf16 as the per-layer form stores it); x = the RAW expansion output the conv read through its lazy
 * BatchNorm, x_vec = [groups][4][C] its BatchNorm vectors, x_act its activation.  Outputs: dx = gradient w.r.t. the activated expansion
 * output, masked (same taps, same order as adamml_dwconv_bwd_data_bn), sums [groups][SLOTS][2C] += sum(dx), sum(dx xhat), and
 * dw [C][3][3] += the weight gradient (split partials in `workspace`, adamml_dwconv_bwd_fused_workspace bytes). */
ADAMML_API int adamml_dwconv_bwd_fused_supported(const adamml_conv_desc_t* d);
ADAMML_API size_t adamml_dwconv_bwd_fused_workspace(const adamml_conv_desc_t* d);
ADAMML_API int adamml_dwconv_bwd_fused(const adamml_conv_desc_t* d, const void* g, const void* z, const float* aff, const float* w_tapmajor,
                            const void* x, const float* x_vec, int x_act, void* dx, double* sums, float* dw, void* workspace,
                            size_t workspace_bytes, hipStream_t stream);
/* 3x3 / stride-2 / pad-1 stem of a ONE-channel fp32 image -- the first conv of the Sound-MobileNetV2 and of the policy MobileNetV2 on a
 * log-spectrogram (models/sound_mobilenet_v2.py:96, models/policy_net.py:108 with input_channels = 1) -- reading the caller's fp32 tensor
 * directly: image n of BatchNorm group g at x + g * group_stride + n * image_stride floats (for the [B, S, H, W] input of
 * models/adamml.py:49-53 with the segment as the group: image_stride = S*H*W, group_stride = H*W).  The spectrogram's range (-5 +- 3)
 * costs bf16 two to three bits; here neither the input nor the 3x3 weights (w_tapmajor [9][Cout] fp32, adamml_pack_conv_weight mode 2)
 * are rounded, only the output y [groups*N][OH][OW][Cout] bf16; stats as adamml_conv_fwd.  No data gradient (network input). */
ADAMML_API int adamml_conv_stem1_supported(const adamml_conv_desc_t* d);
ADAMML_API int adamml_conv_stem1_fwd(const adamml_conv_desc_t* d, const float* x, size_t image_stride, size_t group_stride, const float* w_tapmajor,
                          void* y, double* stats, hipStream_t stream);
ADAMML_API size_t adamml_conv_stem1_bwd_weight_workspace(const adamml_conv_desc_t* d);
ADAMML_API int adamml_conv_stem1_bwd_weight(const adamml_conv_desc_t* d, const void* dz, const float* x, size_t image_stride, size_t group_stride,
                                 float* dw /* [Cout][3][3], accumulated */, void* workspace,
                                 size_t workspace_bytes, hipStream_t stream);
ADAMML_API int adamml_dwconv_bwd_weight(const adamml_conv_desc_t* d, const void* dz, const void* x, const float* in_scale,
                             const float* in_shift, float* dw, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* Every entry point from here to adamml_gap_fwd is batched over `groups` independent BatchNorm groups (the S segment
 * calls of one backbone, models/adamml.py:151-160): activations are [groups][P][C], statistic accumulators
 * [groups][nslots][2C] fp64, BatchNorm vectors bn_vec = [groups][4][C] fp32 (scale, shift, mean, invstd), backward
 * coefficients coef = [groups][3][C].  A *_gstride is the element stride between the groups of a (scale, shift) pair
 * (0 = one pair shared by all groups, e.g. eval mode).
 *
 * nn.BatchNorm2d: train-mode statistics -> bn_vec consumed lazily by the next op, running-stat momentum update
 * (unbiased variance) applied once per group IN GROUP ORDER, exactly as `groups` successive module calls would.
 * count = elements per channel per group (global count under SyncBN). */
/* out[groups][2C] = sum over the ADAMML_STAT_SLOTS copies (before a SyncBN all-reduce) */
ADAMML_API int adamml_stats_collapse(const double* stats, double* out, int C, int groups, hipStream_t stream);
ADAMML_API int adamml_bn_finalize(const double* stats, int nslots, int groups, double count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, float* bn_vec, int C,
                       hipStream_t stream);
/* eval-mode BatchNorm folded to (scale, shift) from the running statistics */
ADAMML_API int adamml_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                          float eps, float* scale, float* shift, int C, hipStream_t stream);
/* out = act(scale*z + shift + identity) -- BN apply + residual add + ReLU of a bottleneck / inverted-residual block
 * (models/resnet.py:104-111, sound_mobilenet_v2.py:66-67, policy_net.py:92-93).  identity may be lazy or NULL. */
ADAMML_API int adamml_bn_act_add(const void* z, const float* scale, const float* shift, int z_gstride, int act, const void* idn,
                      const float* id_scale, const float* id_shift, int id_gstride, void* out, size_t P, int C, int groups,
                      hipStream_t stream);
/* same, additionally writing mask_out[(p*C + c) / 8] bit (c % 8) = act'(stored out) != 0 for the residual backward */
ADAMML_API int adamml_bn_act_add_mask(const void* z, const float* scale, const float* shift, int z_gstride, int act, const void* idn,
                           const float* id_scale, const float* id_shift, int id_gstride, void* out, uint8_t* mask_out, size_t P,
                           int C, int groups, hipStream_t stream);
/* g = g_out * act'(out) evaluated from the stored block output */
ADAMML_API int adamml_act_bwd_from_output(const void* g_out, const void* out, int act, void* g, size_t n, hipStream_t stream);
/* residual-add backward: g2 = g_out * act'(out) fused with the BatchNorm-backward sums of up to two lazily normalised
 * operands of the add (za/veca/sumsa and zb/vecb/sumsb); any of the two may be NULL */
ADAMML_API int adamml_residual_bwd(const void* g_out, const void* out, int act, void* g2, const void* za, const float* veca, double* sumsa,
                        const void* zb, const float* vecb, double* sumsb, size_t P, int C, int groups, hipStream_t stream);
/* BatchNorm backward: per-channel sums of g' = g*act'(scale*z+shift) and g'*zhat (sums [groups][SLOTS][2C], caller zeroes) */
ADAMML_API int adamml_bn_bwd_reduce(const void* g, const void* z, const float* bn_vec, int act, double* sums, size_t P, int C, int groups,
                         hipStream_t stream);
/* dgamma += grad_scale * sum_groups sum(g' zhat); dbeta += grad_scale * sum_groups sum(g');
 * coef[g][0..C) = gamma*invstd, [C..2C) = sum g'/count, [2C..3C) = sum g' zhat/count.
 * grad_scale = 1 normally; 1/world under SyncBatchNorm, where `sums` are already summed over the ranks and the
 * data-parallel gradient average sums dgamma / dbeta over the ranks once more (torch's SyncBatchNorm keeps
 * grad_weight / grad_bias local, train_adamml.py:126-129). */
ADAMML_API int adamml_bn_bwd_finalize(const double* sums, int nslots, int groups, double count, const float* gamma, const float* bn_vec,
                           float* dgamma, float* dbeta, float* coef, int C, float grad_scale, hipStream_t stream);
/* adamml_bn_bwd_finalize and adamml_bn_bwd_affine (aff [groups][3][C]: dz = A g' + B z + C) in one launch -- the pair runs back to back in
 * front of every data gradient with a BatchNorm-backward loader (adamml_conv_bwd_data_dual, adamml_conv_bwd_data_alg,
 * adamml_dwconv_bwd_fused); same values, one launch less on the critical path. */
ADAMML_API int adamml_bn_bwd_finalize_affine(const double* sums, int nslots, int groups, double count, const float* gamma, const float* bn_vec,
                                  float* dgamma, float* dbeta, float* coef, float* aff, int C, float grad_scale, hipStream_t stream);
/* dz = coef0 * (g' - coef1 - zhat*coef2) */
ADAMML_API int adamml_bn_bwd_apply(const void* g, const void* z, const float* bn_vec, int act, const float* coef, void* dz, size_t P, int C,
                        int groups, hipStream_t stream);

/* nn.MaxPool2d(3, 2, 1) on a lazy input (models/resnet.py:141,202); idx = argmax tap (uint8) for the backward.
 * z_sel (NULL or [groups*N, OH, OW, C] bf16) receives the RAW input value at the arg-max tap: with it the BatchNorm backward of
 * the pool's input takes sum(g'), sum(g' zhat) from adamml_bn_bwd_reduce(g_y, z_sel, ...) over the windows (a quarter of the
 * pixels) instead of adamml_maxpool2d_bwd_bn_reduce over the input.  N = images per group. */
ADAMML_API int adamml_maxpool2d_fwd(const void* x, const float* scale, const float* shift, int gstride, int act, void* y, uint8_t* idx,
                         void* z_sel, int N, int H, int W, int C, int OH, int OW, int groups, hipStream_t stream);
ADAMML_API int adamml_maxpool2d_bwd(const void* g_y, const uint8_t* idx, void* g_x, int N, int H, int W, int C, int OH, int OW,
                         int accumulate, hipStream_t stream);
/* TemporalPooling (models/common.py:4-33): k3 s2 p1 over the frame axis; mode 0 = max, 1 = avg (zeros counted).
 * NB = clips per group. */
ADAMML_API int adamml_temporal_pool_fwd(const void* x, const float* scale, const float* shift, int gstride, int act, void* y, int NB,
                             int T, size_t HWC, int C, int mode, int groups, hipStream_t stream);
ADAMML_API int adamml_temporal_pool_bwd(const void* g_y, const void* x, const float* scale, const float* shift, int gstride, int act,
                             void* g_x, int NB, int T, size_t HWC, int C, int mode, int groups, hipStream_t stream);
/* MaxPool2d(3,2,1) backward fused with the BatchNorm backward of the pool's (lazily normalised) input -- the ResNet stem
 * (models/resnet.py:199-202), whose only consumer is the pool.  The routed gradient is recomputed from g_y / idx in both
 * passes and never stored: _reduce accumulates sum(g'), sum(g' zhat) into sums [groups][SLOTS][2C]; _apply (after
 * adamml_bn_bwd_finalize produced coef) writes dz.  N = images per group; g_y / idx: [groups*N, OH, OW, C]. */
ADAMML_API int adamml_maxpool2d_bwd_bn_reduce(const void* g_y, const uint8_t* idx, const void* z, const float* vec, int act, double* sums,
                                   int N, int H, int W, int C, int OH, int OW, int groups, hipStream_t stream);
ADAMML_API int adamml_maxpool2d_bwd_bn_apply(const void* g_y, const uint8_t* idx, const void* z, const float* vec, int act, const float* coef,
                                  void* dz, int N, int H, int W, int C, int OH, int OW, int groups, hipStream_t stream);

/* Temporal MAX-pool backward fused with the residual-add backward of the block that produced the pool's input
 * (models/common.py:28-33 directly after models/resnet.py:110-111; the pool is that block output's only consumer):
 *   g2[n,t] = route(g_y)[n,t] * act'(out[n,t]),  sums_a += (sum g2, sum g2 * zhat_a)   per channel and group.
 * out / g2 / z_a: [groups*NB*T, HW, C]; g_y: [groups*NB*To, HW, C]; T in {2,4,8}.  Replaces adamml_temporal_pool_bwd +
 * adamml_residual_bwd (6.5 -> 3.5 passes over the block-output tensor). */
ADAMML_API int adamml_temporal_pool_bwd_res_supported(int T, int C, int mode);
ADAMML_API int adamml_temporal_pool_bwd_res(const void* g_y, const void* out, int act, void* g2, const void* z_a, const float* vec_a,
                                 double* sums_a, int NB, int T, int HW, int C, int groups, hipStream_t stream);
/* The same backward from the 2-bit codes adamml_conv_fwd_bn_add_tpool stored instead of the block output:
 *   g2[n,t] = sum over the windows `to` containing frame t of (code[n,to] == tap of t ? g_y[n,to] : 0)   (code 3 routes nothing),
 * rounded to bf16 as the unfused pair does, and sums_a += sum g2 per channel and group (second moment 0: the algebraic BatchNorm
 * backward derives it, adamml_alg_sumfix).  Bit-identical to adamml_temporal_pool_bwd_res(z_a = NULL) on the block output. */
ADAMML_API int adamml_temporal_pool_bwd_code(const void* g_y, const uint16_t* code, void* g2, double* sums_a, int NB, int T, int HW, int C, int groups,
                                  hipStream_t stream);
/* adamml_temporal_pool_bwd_code AND the first product its output feeds, in one pass (round 5): while a tile of the expanded gradient g2
 * is on chip it is multiplied with the matching tile of the block's conv3 input a [groups*NB*T, HW, Cin] (raw, with its lazy
 * BatchNorm in_scale / in_shift / in_act): prod [groups][C][Cin] = g2^T a per group, OVERWRITTEN -- the product
 * adamml_conv_bwd_weight_grouped(dz = g2) computes for the algebraic BatchNorm backward of conv3 (adamml_alg_sumfix needs it before the
 * coefficients exist), without reading g2 back.  g2 bit-identical, sums_a as adamml_temporal_pool_bwd_code.
 * (T, C, Cin) = (8, 256, 64): the end of ResNet-50 stage 1. */
ADAMML_API int adamml_temporal_pool_bwd_code_prod_supported(int T, int C, int Cin);
ADAMML_API size_t adamml_temporal_pool_bwd_code_prod_workspace(int NB, int T, int HW, int C, int Cin, int groups);
ADAMML_API int adamml_temporal_pool_bwd_code_prod(const void* g_y, const uint16_t* code, void* g2, double* sums_a, const void* a, const float* in_scale,
                                       const float* in_shift, int in_gstride, int in_act, float* prod, void* workspace, size_t workspace_bytes,
                                       int NB, int T, int HW, int C, int Cin, int groups, hipStream_t stream);

/* AdaptiveAvgPool2d(1) on a lazy input -> fp32 [groups*N,C] (resnet.py:212, sound_mobilenet_v2.py:157, policy_net.py:147) */
ADAMML_API int adamml_gap_fwd(const void* x, const float* scale, const float* shift, int gstride, int act, float* out, int N, int HW,
                   int C, int groups, hipStream_t stream);
ADAMML_API int adamml_gap_bwd(const float* g, void* g_x, int N, int HW, int C, hipStream_t stream);
/* Fused classifier head (models/resnet.py:212-221, models/sound_mobilenet_v2.py:155-158): AdaptiveAvgPool2d(1) -> Dropout ->
 * Linear -> mean over the T remaining frames of a clip, one launch.  x: lazy [clips*T, HW, C] bf16 (clips = all groups);
 * keep_mask [clips*T, C] bytes (NULL: no dropout; kept features are scaled by inv_keep = 1/(1-p)) -- the caller draws it, so a
 * parity run can supply the reference's mask; weight [K, C], bias [K] fp32.  Outputs: feat [clips*T, C] fp32 (pooled, masked:
 * the left operand of the weight gradient), logits [clips, K]. */
ADAMML_API int adamml_head_fwd(const void* x, const float* scale, const float* shift, int gstride, int act, const uint8_t* keep_mask, float inv_keep,
                    const float* weight, const float* bias, float* feat, float* logits, int clips, int T, int HW, int C, int K, int groups,
                    hipStream_t stream);
/* g [clips, K] -> g_x [clips*T, HW, C] bf16 (gradient w.r.t. the activated pool input) and g_rows [clips*T, K] = g / T (may be
 * NULL): dW += g_rows^T feat is an adamml_gemm_f32 call, db += column sums of g is adamml_colsum_f32. */
ADAMML_API int adamml_head_bwd(const float* g, const uint8_t* keep_mask, float inv_keep, const float* weight, void* g_x, float* g_rows, int clips, int T,
                    int HW, int C, int K, hipStream_t stream);
/* out[c] (+)= sum_r a[r, c] for a small row-major fp32 matrix (rows added in order: deterministic) */
ADAMML_API int adamml_colsum_f32(const float* a, float* out, int rows, int cols, int accumulate, hipStream_t stream);

/* AdaMML.data_layer (models/adamml.py:42-67): NCHW fp32 clip tensor [B, S*F*C, H, W] -> per-segment NHWC bf16
 * frames [S][B*Fk][OH][OW][c_pad] with optional bilinear resize (align_corners=False) and frame stride. */
ADAMML_API int adamml_clip_to_nhwc(const float* x, void* y, int B, int S, int F, int C, int H, int W, int OH, int OW,
                        int frame_step, int c_pad, hipStream_t stream);

/* Decoded-frame input path: what utils/video_transforms.py:302-343 (Stack -> ToTorchFormatTensor(div) -> GroupNormalize) and
 * AdaMML.data_layer do between the decoder and the first conv, in one launch from the uint8 frames.  x: [B][H][W][S*F*C]
 * uint8 (Stack's HW(FC) array per video); value = ((u8 / 255 if div255) - mean[c % n_mean]) / std[c % n_mean] in fp32, then the
 * same re-layout / bilinear resize / frame stride as adamml_clip_to_nhwc.  mean / std: HOST arrays of n_mean <= 4 floats
 * (models/adamml.py:93-99). */
ADAMML_API int adamml_clip_u8_to_nhwc(const uint8_t* x, void* y, int B, int S, int F, int C, int H, int W, int OH, int OW, int frame_step,
                           int c_pad, const float* mean, const float* std, int n_mean, int div255, hipStream_t stream);
/* RGB-diff modality computed on the GPU from decoded RGB frames (utils/video_dataset.py:32-38,75-84: for every frame group the
 * loader reads D+1 consecutive frames and stores D difference images uint8((next - cur + 255) * 0.5)):
 * x [B, H, W, S*F*(D+1)*3] uint8 RGB frames -> y [S, B*Fk, OH, OW, c_pad] bf16 with 3*D difference channels per frame group,
 * then ToTorchFormatTensor (/255) + GroupNormalize + the data-layer resize exactly as adamml_clip_u8_to_nhwc.  D = 5 in the
 * reference (num_consecutive_frames, utils/video_dataset.py:310-311). */
ADAMML_API int adamml_clip_u8_rgbdiff_to_nhwc(const uint8_t* x, void* y, int B, int S, int F, int D, int H, int W, int OH, int OW, int frame_step,
                                   int c_pad, const float* mean, const float* std, int n_mean, hipStream_t stream);

/* y[M,N] = act(x[M,K] @ w[N,K]^T + bias) in fp32 with arbitrary strides (nn.Linear / LSTMCell gates and their
 * gradients: policy_net.py:228-231,278-279,351-362; resnet.py:215; sound_mobilenet_v2.py:158) */
ADAMML_API int adamml_gemm_f32(const float* a, int64_t a_sm, int64_t a_sk, const float* b, int64_t b_sn, int64_t b_sk, float* c,
                    int64_t c_sm, int64_t c_sn, const float* bias, int act, int accumulate, int M, int N, int K,
                    hipStream_t stream);

/* All weight packs of a backbone in one launch.  table (device): n rows of 6 x int64 {w (fp32 OIHW), out, cout | cin_true << 32,
 * cin_pad | kh << 32, kw | mode << 32, first block}; row i owns blocks [first_i, first_{i+1}) of adamml_pack_block_elems()
 * output elements each (modes as adamml_pack_conv_weight); total_blocks = first block past the last row. */
ADAMML_API int adamml_pack_block_elems(void);
ADAMML_API int adamml_pack_conv_weights_batched(const int64_t* table, int n, int64_t total_blocks, hipStream_t stream);

/* flat fused optimizer steps (train_adamml.py:251-257 SGD-momentum / Adam with weight decay) */
ADAMML_API int adamml_sgd_step(float* p, const float* g, float* mom, size_t n, float lr, float momentum, float weight_decay,
                    int nesterov, int first_step, hipStream_t stream);
ADAMML_API int adamml_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, hipStream_t stream);

/* ---- policy causality head and late fusion -------------------------------------------------------------------------
 * PolicyNet.forward, causality_modeling='lstm' (models/policy_net.py:341-370): nn.LSTMCell over the S segments with the
 * previous segment's logits fed back, one nn.Linear(256, 2) head per modality, F.gumbel_softmax(tau, hard=True)[:, -1]
 * (:283-290).  The recurrence is per video, so one workgroup runs one video through all S steps (one launch, no
 * per-segment launches).  The caller first computes the feature half of the gates for ALL segments with
 * adamml_gemm_f32: gates_x[S,B,4H] = feat[S*B,F] @ W_ih[:, :F]^T + b_ih;  this entry point adds b_hh, W_hh h and
 * W_ih[:, F:] prev (w_prev = W_ih + F, ld_ih = F + 2M).  `expo` [S,M,B,2] is the Exponential(1) draw behind the Gumbel
 * noise (torch.empty_like(logits).exponential_() in the reference), supplied by the caller so runs are reproducible.
 * fc_w / fc_b are HOST arrays of M device pointers (fcs[m].weight [2,H], fcs[m].bias [2]); M <= 4; hidden must be 256.
 * Outputs: decisions [S,M,B], logits [S,M,B,2]; saved for the backward pass: h_all / c_all [S+1,B,H] (slot 0 = the zero
 * initial state), gate_act [S,B,4H] (post-nonlinearity i,f,g,o), prev_all [S,B,2M], ysoft [S,M,B,2]. */
ADAMML_API int adamml_policy_head_fwd(const float* gates_x, const float* w_prev, int ld_ih, const float* w_hh, const float* b_hh,
                           const float* const* fc_w, const float* const* fc_b, const float* expo, float tau,
                           float* decisions, float* logits, float* h_all, float* c_all, float* gate_act, float* prev_all,
                           float* ysoft, int S, int B, int M, int hidden, hipStream_t stream);
/* Reverse recurrence: d_decisions [S,M,B] (straight-through: the gradient flows through y_soft), d_logits_in [S,M,B,2]
 * or NULL.  Writes d_gates [S,B,4H] (gradient of the pre-activation gates) and d_logits [S,M,B,2] (total gradient of each
 * step's logits); the weight / bias / feature gradients are GEMMs over those two (adamml_gemm_f32), issued by the caller. */
ADAMML_API int adamml_policy_head_bwd(const float* d_decisions, const float* d_logits_in, const float* w_prev, int ld_ih,
                           const float* w_hh, const float* const* fc_w, float tau, const float* c_all,
                           const float* gate_act, const float* ysoft, float* d_gates, float* d_logits, int S, int B, int M,
                           int hidden, hipStream_t stream);
/* The gate alone, for the head without causality modelling (policy_net.py:330-340): rows of 2 logits. */
ADAMML_API int adamml_gumbel_gate_fwd(const float* logits, const float* expo, float tau, float* decisions, float* ysoft, int rows,
                           hipStream_t stream);
ADAMML_API int adamml_gumbel_gate_bwd(const float* d_decisions, const float* ysoft, float tau, float* d_logits, int rows,
                           hipStream_t stream);

/* Decision-gated late fusion and segment mean (models/joint_resnet_mobilenetv2.py:94,112-127; models/adamml.py:86-88):
 *   out[b,c] = 1/S sum_s sum_m w_m * (decisions[s,m,b] * x[m][s*B+b, c]),  w = cat(lf_weights, 1 - sum(lf_weights)) when
 * lf_weights != NULL ([M-1], learnable), else 1/M.  x / d_x: HOST arrays of M device pointers to [S*B, C] fp32 logits;
 * decisions may be NULL (no gating).  Backward: d_x[m] (entries may be NULL), d_decisions [S,M,B] or NULL, and
 * d_lf_part [S*B, M] or NULL = per-row d out / d w_m (the caller folds rows and applies d w / d lf_weights). */
ADAMML_API int adamml_fusion_fwd(const float* const* x, const float* decisions, const float* lf_weights, float* out, int S, int B,
                      int C, int M, hipStream_t stream);
ADAMML_API int adamml_fusion_bwd(const float* const* x, const float* decisions, const float* lf_weights, const float* g_out,
                      float* const* d_x, float* d_decisions, float* d_lf_part, int S, int B, int C, int M,
                      hipStream_t stream);

/* ---- launch plans (csrc/plan.hip) -------------------------------------------------------------------------------------------
 * The static launch sequence of one backbone call, recorded once by the host executor and replayed by ONE call: every record is a call
 * of one of the stream-taking entry points above (`fn` = its index in the alphabetically sorted list of those entry points,
 * adamml_plan_num_entry_points() of them), a wait of one of the caller's streams on another, or a memset.  Arguments are 8-byte slots:
 * pointers and integers as they are, float / double arguments as the bits of a double, descriptors as pointers to HOST copies the
 * caller keeps alive.  A bit of `slot_mask` marks an argument as an index into the `slots` array of adamml_plan_run (the pointers that
 * change from replay to replay: the call's input, the incoming gradient).  Nothing is allocated or synchronised; an error of a
 * record ends the replay with that record's code.  The reference has no counterpart: its launches are issued by the PyTorch
 * dispatcher, one Python call each (utils/utils.py:359-400). */
#define ADAMML_PLAN_MAX_ARGS 21
#define ADAMML_PLAN_CALL 0
#define ADAMML_PLAN_WAIT 1
#define ADAMML_PLAN_ZERO 2
typedef struct {
    int32_t kind, fn, nargs, stream;      /* stream: index into the `streams` array of adamml_plan_run */
    uint32_t slot_mask, reserved;
    uint64_t a[ADAMML_PLAN_MAX_ARGS];
} adamml_plan_op_t;
ADAMML_API int adamml_plan_num_entry_points(void);
ADAMML_API int adamml_plan_run(const adamml_plan_op_t* ops, int n_ops, const hipStream_t* streams, int n_streams, const hipEvent_t* events, int n_events,
                    const uint64_t* slots, int n_slots);       /* ops / streams / events / slots: HOST arrays */
ADAMML_API int adamml_plan_events_create(hipEvent_t* events, int n);     /* HOST array, filled with timing-less events */
ADAMML_API int adamml_plan_events_destroy(hipEvent_t* events, int n);
/* Stream-ordered strided device copy (byte pitches): dst[r][0..width) = src[r][0..width), r < rows. */
ADAMML_API int adamml_copy2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
