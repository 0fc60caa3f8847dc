/* gdlhip.h -- C ABI of libgdlhip.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * DOFA-ViT + UperNet semantic-segmentation hot path of NRCan/geo-deep-learning.
 *
 * The reference has no native boundary of its own (it is 100 % Python on torch.nn /
 * timm / smp): the drop-in boundary is L1 "tensor backend"
 * (SURVEY.md section 1, row L1).  Every entry point below cites the reference call it
 * replaces.  Conventions:
 *   - plain pointers + sizes, no torch types; all tensors are DEVICE pointers owned by
 *     the caller (PyTorch's caching allocator); the library allocates nothing.
 *   - activations are NHWC / token-major ("channels last"); strides are in ELEMENTS.
 *   - every function enqueues on the caller's hipStream_t and returns 0 or a negative
 *     gdl_status; it never throws or aborts.  gdl_last_error() gives a message.
 *   - dtype codes: GDL_F32 = 0, GDL_BF16 = 1.
 */
#ifndef GDLHIP_H_
#define GDLHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gdl_stream_t; /* hipStream_t */

enum { GDL_F32 = 0, GDL_BF16 = 1 };
enum {
  GDL_ACT_NONE = 0,
  GDL_ACT_RELU = 1,
  GDL_ACT_GELU = 2,         /* erf GELU */
  GDL_ACT_MUL_GELU_GRAD = 3, /* backward of a fused Linear+GELU: v *= gelu'(u), u passed in `resid` */
  GDL_ACT_RESID_RELU = 4     /* ReLU applied AFTER the residual add: ResNet BasicBlock tail relu(bn(conv) + identity) */
};
enum {
  GDL_OK = 0,
  GDL_ERR_INVALID = -1, /* bad argument (shape / alignment / dtype) */
  GDL_ERR_LAUNCH = -2,  /* hipLaunchKernel or a runtime call failed */
  GDL_ERR_UNSUPPORTED = -3
};

int gdl_version(void);
const char* gdl_last_error(void);

/* ---- implicit-GEMM convolution / linear (MFMA) ---------------------------------------
 * out[b,oy,ox,n] = epilogue( alpha * sum_{r,s,c} in[b, oy*stride+r-pad, ox*stride+s-pad, c]
 *                                               * w[n, (r*S+s)*C + c] )
 * epilogue: v += bias[n]; if (scale) v = v*scale[n] + (shift ? shift[n] : 0);
 *           v = act(v); if (batch_scale) v *= batch_scale[b]; if (resid) v += resid[b,oy,ox,n];
 *           store as out_dtype.
 * A plain Linear / 1x1 conv is R=S=1 (M = B*Ho*Wo rows).  Batched GEMM via nz.
 * Replaces: F.conv2d / nn.Linear + BatchNorm(eval, folded) + ReLU/GELU + LayerScale +
 * residual add -- timm Block (dofa_v2.py:248-263), ConvModule (models/utils.py:10-52,
 * multilevel_neck.py:28-67), upernet.py:62-101, fcn_head.py:36-84; with transposed /
 * flipped weights it is also the data-gradient of those convs.
 * Requirements: C % (16/elem_size) == 0 (whole 16-byte pieces; a channel count that does not fill
 * the 128-byte K chunk is zero-filled by the kernel), all base pointers and row strides 16-byte
 * aligned, dtype of `in` and `w` equal (GDL_F32 -> exact-f32 MFMA, GDL_BF16 -> bf16 MFMA
 * with f32 accumulate).
 */
typedef struct {
  const void* in;
  int dtype; /* operand dtype of in and w */
  int B, H, W, C;
  int64_t in_sB, in_sH, in_sW;
  int Ho, Wo, R, S, stride, pad;
  const void* w;
  int64_t w_sN; /* elements between consecutive n rows (>= R*S*C) */
  int N;
  void* out;
  int out_dtype;
  int64_t out_sB, out_sH, out_sW;
  float alpha;
  const float* bias;
  const float* scale;
  const float* shift;
  int act;
  const float* batch_scale; /* [B] per-sample factor applied before the residual (DropPath) */
  const void* resid;
  int resid_dtype;
  int64_t res_sB, res_sH, res_sW;
  /* batching over grid.z: z -> (z / nz_inner, z % nz_inner) */
  int nz, nz_inner;
  int64_t in_sZ0, in_sZ1, w_sZ0, w_sZ1, out_sZ0, out_sZ1;
  void* aux_out; /* optional: alpha*acc + bias, i.e. the value BEFORE scale/shift/act (same dtype and
                    strides as out), saved for the backward pass (GELU input, LayerScale input) */
  float* stats_partial; /* optional, only where gdl_conv_gemm_stats_rows() > 0: [rows][2][N] f32 partial sums (sum, sum of
                           squares) of the bf16-ROUNDED outputs over 32 * TM output pixels each -- the train-mode BatchNorm
                           statistics of a ConvModule without a pass over its output (models/utils.py:10-52); reduce with
                           gdl_bn_stats_finalize(partials, rows, N, B * Ho * Wo, ...) */
} gdl_conv_args;

int gdl_conv_gemm(const gdl_conv_args* a, gdl_stream_t stream);
/* Number of partial rows gdl_conv_gemm writes to `stats_partial` for this call, 0 when the call cannot emit statistics
 * (bf16 in / out, bias-only epilogue, pixel-dense output, whole 128 / 256-pixel tiles on the coalesced-epilogue kernels).
 * `stats_partial` itself is ignored by the query. */
int64_t gdl_conv_gemm_stats_rows(const gdl_conv_args* a);
/* Which kernel variant the call above will launch (0 = 64x64 tiles, 1 = 128x128, 2 = 256x256 / 8 waves, 3 = 256x256 with
 * alternating loader halves, 4 = 256x256 3x3 with shared activation staging, 5 = 256x64 for narrow outputs, 6 = dual-resident
 * 256x128 tile (GELU layers), 7 = direct 3x3 for 8 / 16 / 32 input channels, 8 = 256x256 with one wave per SIMD (>= 40
 * K-steps), 9 = persistent 256x256 for dense 1x1 layers, 10 = persistent 256x256 with one wave per SIMD whose finished tile
 * is parked in registers and stored from the MFMA shadows of the next tile's K loop) and its algorithmic flops 2*M*N*K (for
 * roofline accounting in bench.py). */
int gdl_conv_gemm_plan(const gdl_conv_args* a, int64_t* flops);

/* Weight gradient of the same convolution:
 * dw[n, (r*S+s)*C + c] (+)= sum_{b,oy,ox} dy[b,oy,ox,n] * in[b, oy*stride+r-pad, ox*stride+s-pad, c]
 * dw is f32 [N][R*S*C]; accumulate != 0 adds into dw.  (autograd of F.conv2d wrt weight.) */
typedef struct {
  const void* in;
  const void* dy;
  int dtype;
  int B, H, W, C;
  int64_t in_sB, in_sH, in_sW;
  int Ho, Wo, R, S, stride, pad;
  int N;
  int64_t dy_sB, dy_sH, dy_sW;
  float* dw;
  int64_t dw_sN;
  int accumulate;
  float* workspace;       /* split-K partials, >= gdl_conv_wgrad_workspace() bytes, or NULL */
  int64_t workspace_bytes;
  /* batched mode (nz > 1): grid.z -> (z / nz_inner, z % nz_inner) selects independent problems (the per-head
   * dK = dS^T Q and dV = P^T dO of attention backward); gdl_conv_wgrad_workspace() covers nz * splits partials */
  int nz, nz_inner;
  int64_t in_sZ0, in_sZ1, dy_sZ0, dy_sZ1, dw_sZ0, dw_sZ1;
} gdl_wgrad_args;

int64_t gdl_conv_wgrad_workspace(const gdl_wgrad_args* a);
int gdl_conv_wgrad(const gdl_wgrad_args* a, gdl_stream_t stream);

/* ---- normalisation -------------------------------------------------------------------
 * LayerNorm over the last dim (biased variance, eps inside sqrt): F.layer_norm as used by
 * timm Block norm1/norm2 (dofa_v2.py:248-263) and nn.TransformerEncoderLayer (dofa_v2.py:73-85).
 * x: f32 rows with stride x_stride; y: dtype y_dtype, dense [rows][D]. */
int gdl_layernorm_fwd(const float* x, int64_t x_stride, const float* gamma, const float* beta,
                      void* y, int y_dtype, int64_t rows, int D, float eps, gdl_stream_t stream);

/* BatchNorm2d, training mode, NHWC [P pixels][C] (F.batch_norm(training=True)).
 * stats: per-channel sum and sum of squares in f64-free two-level f32 (deterministic). */
int gdl_bn_stats(const void* x, int dtype, int64_t P, int C, int64_t x_sP, float* mean,
                 float* var_biased, float* running_mean, float* running_var, float momentum,
                 float* workspace, int64_t workspace_bytes, gdl_stream_t stream);
int64_t gdl_bn_stats_workspace(int64_t P, int C);
/* y = relu?( (x-mean)*rsqrt(var+eps)*gamma + beta ), in place allowed */
int gdl_bn_apply(const void* x, void* y, int dtype, int64_t P, int C, int64_t x_sP, int64_t y_sP,
                 const float* mean, const float* var, const float* gamma, const float* beta,
                 float eps, int relu, gdl_stream_t stream);
/* backward of y = relu?(bn(x)) from the saved conv output x (the ReLU mask is recomputed as
 * bn(x) > 0).  Two steps so SyncBatchNorm can all-reduce the sums in between:
 *   reduce: dbeta = sum g, dgamma = sum g*xhat  (g = dy * mask)
 *   dx    : gamma*rstd*(g - dbeta_sum/P_total - xhat*dgamma_sum/P_total)                      */
int gdl_bn_bwd_reduce(const void* x, const void* dy, int dtype, int64_t P, int C, int64_t x_sP,
                      int64_t dy_sP, const float* mean, const float* var, const float* gamma,
                      const float* beta, float eps, int relu, float* dgamma, float* dbeta,
                      float* workspace, int64_t workspace_bytes, gdl_stream_t stream);
int gdl_bn_bwd_dx(const void* x, const void* dy, void* dx, int dtype, int64_t P, int C, int64_t x_sP,
                  int64_t dy_sP, int64_t dx_sP, const float* mean, const float* var,
                  const float* gamma, const float* beta, float eps, int relu,
                  const float* dgamma_sum, const float* dbeta_sum, int64_t P_total,
                  gdl_stream_t stream);
/* SyncBatchNorm (torch.nn.SyncBatchNorm under `sync_batchnorm: true`, configs/dofa_config_RGB.yaml:13): the cross-rank exchange is
 * ONE all-reduce(SUM) of a count-weighted message per direction.  gdl_syncbn_pack writes this rank's [count * mean | count * E[x^2] |
 * count] (2 C + 1 floats) into `out` (a slice of the message buffer); gdl_syncbn_unpack reads the summed message back into the
 * GLOBAL mean / biased variance and updates the running estimates with the unbiased variance over the global count (which stays
 * on the device).  gdl_bn_bwd_dx_sync = gdl_bn_bwd_dx with the all-reduced sums and that device-side count (`total_count` points
 * at entry 2 C of the forward message). */
int gdl_syncbn_pack(const float* mean, const float* var, double count, int C, float* out, gdl_stream_t stream);
int gdl_syncbn_unpack(const float* packed, int C, float* mean, float* var, float* running_mean, float* running_var, float momentum,
                      gdl_stream_t stream);
int gdl_bn_bwd_dx_sync(const void* x, const void* dy, void* dx, int dtype, int64_t P, int C, int64_t x_sP, int64_t dy_sP, int64_t dx_sP,
                       const float* mean, const float* var, const float* gamma, const float* beta, float eps, int relu,
                       const float* dgamma_sum, const float* dbeta_sum, const float* total_count, gdl_stream_t stream);
/* The whole train-mode BatchNorm(+ReLU) of a SMALL map in one launch per direction (one workgroup = four channels over all
 * pixels: statistics, running-estimate update, normalised output / sums, parameter gradients, dx).  For the maps of at most a
 * few thousand pixels that a per-GPU batch of 4 (configs/dofa_config_RGB.yaml:85) gives most ConvModules of the decoder
 * (models/utils.py:10-52, upernet.py:62-127): the three launches of each direction above are shorter than the gap between
 * dependent launches there.  Same arithmetic as gdl_bn_stats + gdl_bn_apply resp. gdl_bn_bwd_reduce + gdl_bn_bwd_dx.
 * Single-process statistics only (SyncBatchNorm exchanges the sums between the two passes). */
int gdl_bn_small_fwd(const void* x, void* y, int dtype, int64_t P, int C, int64_t x_sP, int64_t y_sP, const float* gamma,
                     const float* beta, float eps, int relu, float* mean, float* var, float* running_mean, float* running_var,
                     float momentum, gdl_stream_t stream);
int gdl_bn_small_bwd(const void* x, const void* dy, void* dx, int dtype, int64_t P, int C, int64_t x_sP, int64_t dy_sP, int64_t dx_sP,
                     const float* mean, const float* var, const float* gamma, const float* beta, float eps, int relu,
                     float* dgamma, float* dbeta, gdl_stream_t stream);

/* ---- transformer-block backward (timm Block dofa_v2.py:248-263, MiT Block mix_transformer.py:160-221) ----
 * Column reductions (parameter gradients) go through per-block partials in `ws`
 * (>= gdl_colreduce_workspace(rows, C, planes) bytes) and a deterministic final pass; no atomics. */
int64_t gdl_colreduce_workspace(int64_t rows, int C, int planes);
/* autograd of F.layer_norm: dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) [+ dres], g = dy*gamma;
 * dgamma (+)= sum dy*xhat, dbeta (+)= sum dy.  x f32 rows (stride x_stride), dy dense [rows][D] of dy_dtype,
 * dres optional f32 gradient of the residual stream added into dx.  planes = 2. */
int gdl_layernorm_bwd(const float* x, int64_t x_stride, const void* dy, int dy_dtype, const float* gamma,
                      const float* dres, int64_t dres_stride, float* dx, int64_t dx_stride, int64_t rows, int D,
                      float eps, float* dgamma, float* dbeta, int accumulate_params, float* ws, int64_t ws_bytes,
                      gdl_stream_t stream);
/* out[c] (+)= sum_rows x[row, c]: bias gradient of nn.Linear / conv.  planes = 1. */
int gdl_colsum(const void* x, int dtype, int64_t rows, int C, int64_t x_stride, float* out, int accumulate,
               float* ws, int64_t ws_bytes, gdl_stream_t stream);
/* backward of y = x + batch_scale[b] * gamma[c] * z (LayerScale + DropPath, dofa_v2.py:255-262):
 * dz = g*batch_scale*gamma in dz_dtype; dgamma (+)= sum g*batch_scale*z.  gamma / batch_scale optional. */
int gdl_layerscale_bwd(const float* g, const void* z, int z_dtype, const float* gamma, const float* batch_scale,
                       int64_t rows, int64_t rows_per_batch, int C, void* dz, int dz_dtype, float* dgamma,
                       int accumulate, float* ws, int64_t ws_bytes, gdl_stream_t stream);
/* dS = P * (dP - sum_k dP*P) * scale on the first n_valid columns of each row, pad columns -> 0 (in place ok) */
int gdl_softmax_bwd_rows(const void* p, const void* dp, void* ds, int dtype, int64_t rows, int n_valid, int n_cols,
                         float scale, gdl_stream_t stream);
/* backward of y = gelu(dwconv3x3(u) + bias) (Mix-FFN, mix_transformer.py:533-546 + :52-63):
 * dpre = dy * gelu'(dwconv3x3(u)+bias); dw9[9][C] (+)= sum dpre * u@tap; dbias (+)= sum dpre.  planes = 10.
 * (du is gdl_dwconv3x3 of dpre with the taps reversed.) */
int gdl_dwconv3x3_gelu_bwd(const void* u, const void* dy, int dtype, int B, int H, int W, int C, const float* w9,
                           const float* bias, void* dpre, float* dw9, float* dbias, int accumulate, float* ws,
                           int64_t ws_bytes, gdl_stream_t stream);
/* data gradient of a strided conv: cols[(b,oy,ox)][(r,s,c)] (= dy x W, a GEMM) gathered into dx[b,y,x,c]
 * (autograd of F.conv2d wrt input for OverlapPatchEmbed / the sr conv, mix_transformer.py:224-276, :95-99) */
int gdl_col2im(const void* cols, int dtype, int B, int Ho, int Wo, int R, int S, int C, int stride, int pad, int H,
               int W, void* dx, int dx_dtype, int64_t dx_sB, int64_t dx_sH, int64_t dx_sW, gdl_stream_t stream);

/* ---- channel-adaptive stem of the "dynamic" SegFormer encoder (DynamicChannelEmbed, mix_transformer.py:762-859) ----
 * Input-independent part (weight_gen :781-786 and the position half of channel_attention[0] :797-801), all f32:
 *   hid[C][HD] = relu(pos W0^T + b0); cw[C][E] = tanh(hid W2^T + b2); hb[C][H1] = pos W1b^T + b1
 * pos[C][PD] is the sinusoidal band code (:810-821); W1b = channel_attention.0.weight[:, E:], row stride w1b_ld. */
int gdl_chan_weights_fwd(const float* pos, int C, int PD, int HD, int E, int H1, const float* W0, const float* b0,
                         const float* W2, const float* b2, const float* W1b, int64_t w1b_ld, const float* b1,
                         float* hid, float* cw, float* hb, gdl_stream_t stream);
/* its backward: (dcw, dhb) -> dW0 [HD][PD], db0, dW2 [E][HD], db2, dW1b [H1][PD], db1 (overwritten) */
int gdl_chan_weights_bwd(const float* pos, int C, int PD, int HD, int E, int H1, const float* W2, const float* hid,
                         const float* cw, const float* dcw, const float* dhb, float* dW0, float* db0, float* dW2,
                         float* db2, float* dW1b, float* db1, gdl_stream_t stream);
/* Per output pixel (:823-853): conv [B][C][P][E] f32 is the shared 7x7 conv of every band (gdl_patchify + gdl_conv_gemm);
 *   xw[c] = conv[c]*cw[c]; s[c] = w2 . relu(W1a xw[c] + hb[c]) + b2s; a = softmax_c(s); agg[b][p][:] = sum_c a[c] xw[c]
 * W1a = channel_attention.0.weight[:, :E] (row stride w1a_ld); attn [B][P][C] is returned for inspection. */
int gdl_chan_pool_fwd(const float* conv, int B, int C, int64_t P, int E, int H1, const float* cw, const float* w1a,
                      int64_t w1a_ld, const float* hb, const float* w2, float b2s, float* agg, float* attn,
                      gdl_stream_t stream);
/* its backward: dagg -> dconv [B][C][P][E] and grads = [dW1a H1*E | dhb C*H1 | dw2 H1 | dcw C*E] (overwritten;
 * d b2s is identically zero: softmax is shift invariant) */
int64_t gdl_chan_pool_workspace(int B, int C, int64_t P, int E, int H1);
int gdl_chan_pool_bwd(const float* conv, int B, int C, int64_t P, int E, int H1, const float* cw, const float* w1a,
                      int64_t w1a_ld, const float* hb, const float* w2, float b2s, const float* dagg, float* dconv,
                      float* grads, float* ws, int64_t ws_bytes, gdl_stream_t stream);

/* torchmetrics.segmentation.MeanIoU(input_format="index") update (segmentation_dofa.py:71-76,313): for each sample
 * and class the exact counts counts[b][0][k] = |pred==k & target==k|, [b][1][k] = |pred==k|, [b][2][k] = |target==k|
 * (int64, zeroed by the call); pred / target are [B][P] class indices, values outside 0..K-1 are ignored. */
int gdl_iou_counts(const int64_t* pred, const int64_t* target, int B, int64_t P, int K, int64_t* counts,
                   gdl_stream_t stream);

/* ---- GPU-side augmentation (kornia pipeline of segmentation_dofa.py:91-121,201-211, fused with the normalise step)
 * img: raw NCHW tile of `kind` (GDL_RAW_*); mean/std non-NULL -> each tap is normalised like gdl_normalize_raw first.
 * params: f32 [B][8] per sample {kind, k, y0, x0, h, w, -, -}; kind 0 none, 1 horizontal flip, 2 vertical flip,
 * 3 rot90 by k quarter turns (torch.rot90, square tiles), 4 crop [y0,y0+h) x [x0,x0+w) resized to H x W (bilinear
 * align_corners=False for the image, nearest for the mask).  mask / out_mask optional int64 [B][H][W]. */
int gdl_augment(const void* img, int kind, float* out, const int64_t* mask, int64_t* out_mask, int B, int C, int H,
                int W, const float* mean, const float* stdv, const float* params, gdl_stream_t stream);

/* ---- ResNet encoder / UNet++ decoder pieces (smp.UnetPlusPlus, segmentation_unetplus.py:126-131; torchvision
 * resnet.py BasicBlock / maxpool; smp decoders/unetplusplus/decoder.py DecoderBlock) ----------------------------
 * NHWC, strides in elements, channels a multiple of 8 (bf16) / 4 (f32); tensors may be slices of a concat buffer. */
/* F.max_pool2d(kernel 3, stride 2, padding 1); backward routes each window's gradient to its FIRST maximum */
int gdl_maxpool3x3s2_fwd(const void* in, int dtype, int B, int H, int W, int C, int64_t in_sB, int64_t in_sH,
                         int64_t in_sW, void* out, int64_t out_sB, int64_t out_sH, int64_t out_sW, gdl_stream_t stream);
int gdl_maxpool3x3s2_bwd(const void* in, const void* dout, void* din, int dtype, int B, int H, int W, int C,
                         int64_t in_sB, int64_t in_sH, int64_t in_sW, int64_t d_sB, int64_t d_sH, int64_t d_sW,
                         int64_t g_sB, int64_t g_sH, int64_t g_sW, gdl_stream_t stream);
/* F.interpolate(scale_factor=2, mode="nearest"): [B,H,W,C] -> [B,2H,2W,C]; backward sums each 2x2 block */
int gdl_nearest2x_fwd(const void* in, int dtype, int B, int H, int W, int C, int64_t in_sB, int64_t in_sH,
                      int64_t in_sW, void* out, int64_t out_sB, int64_t out_sH, int64_t out_sW, gdl_stream_t stream);
int gdl_nearest2x_bwd(const void* dout, int dtype, int B, int H, int W, int C, int64_t d_sB, int64_t d_sH,
                      int64_t d_sW, void* din, int64_t g_sB, int64_t g_sH, int64_t g_sW, gdl_stream_t stream);
/* out = relu(a + b) on dense tensors of n elements; dx = dy * (y > 0) */
int gdl_add_relu(const void* a, const void* b, void* out, int dtype, int64_t n, gdl_stream_t stream);
int gdl_relu_bwd(const void* y, const void* dy, void* dx, int dtype, int64_t n, gdl_stream_t stream);
/* dense [P][C] (any C, e.g. the 5-class logit gradient) -> [P][Cpad] in out_dtype, extra channels zero */
int gdl_pad_channels(const void* in, int in_dtype, int64_t P, int C, void* out, int out_dtype, int Cpad,
                     gdl_stream_t stream);

/* ---- resampling (NHWC) ---------------------------------------------------------------
 * F.interpolate(mode="bilinear", align_corners=False) (models/utils.py:96-137,
 * upernet.py:127-150, models/utils.py:81-93).  accumulate: out += result. */
int gdl_bilinear_fwd(const void* in, int in_dtype, int B, int Hi, int Wi, int C, int64_t in_sB,
                     int64_t in_sH, int64_t in_sW, void* out, int out_dtype, int Ho, int Wo,
                     int64_t out_sB, int64_t out_sH, int64_t out_sW, int accumulate,
                     gdl_stream_t stream);
/* out = base + F.interpolate(in -> Ho x Wo): UperNet's top-down path `laterals[i - 1] = laterals[i - 1] + resize(laterals[i])`
 * (models/decoders/upernet.py:127-135) in ONE pass -- `base` has out's dtype and strides and is not modified.  bf16 maps with
 * 16-byte aligned rows and a resize factor of exactly 2 or 4 take the fused kernel; any other call copies base and accumulates
 * (gdl_copy_cast + gdl_bilinear_fwd(accumulate = 1)): same values. */
int gdl_bilinear_fwd_add(const void* in, int in_dtype, int B, int Hi, int Wi, int C, int64_t in_sB,
                         int64_t in_sH, int64_t in_sW, const void* base, void* out, int out_dtype, int Ho, int Wo,
                         int64_t out_sB, int64_t out_sH, int64_t out_sW, gdl_stream_t stream);
/* Backward of conv3x3(pad 1)(F.interpolate(x, bilinear)) (multilevel_neck.py:56-67,157-158) at LOW resolution: with
 * U = the resize and S_t = the shift of filter tap t (both act on pixels only), dx = sum_t W_t^T G_t and
 * dW_t = sum_q G_t[q] (x) x[q] where G_t = U^T S_t^T dy.  This is the gather that builds the nine maps: dy dense
 * [B,Ho,Wo,N] -> g dense [B,Hi,Wi,9*N], tap block 8 - t (= gdl_pack_dgrad's flipped tap order, so the data gradient is
 * gdl_conv_gemm(g, w_dgrad) as a 1x1 convolution with K = 9 N, and the weight gradient nine 1x1 gdl_conv_wgrad calls on
 * channel slices of g).  Any real resize factor up to 10 (non-integer ratios included).  Deterministic. */
int gdl_resize_conv3x3_bwd_gather(const void* dy, int dtype, int B, int Ho, int Wo, int N, void* g, int Hi, int Wi,
                                  gdl_stream_t stream);
/* 1 when gdl_resize_conv3x3_bwd_gather runs its one-pass matrix-core form for this shape (bf16, N % 64 == 0, factor 2 or 4:
 * dy read once, no intermediate) -- the caller then skips the two-pass entry and its workspace */
int gdl_resize_conv3x3_bwd_gather_one_pass(int dtype, int B, int Ho, int Wo, int N, int Hi, int Wi);
/* gdl_resize_conv3x3_bwd_gather of the BatchNorm(+ReLU) BACKWARD of dz, without that gradient ever existing in memory: for a
 * training ConvModule whose convolution is conv3x3(resize(x)) (multilevel_neck.py:56-67,157-158: conv -> BN -> ReLU), the
 * gradient that reaches the convolution is dy = gamma rstd (dz [bn(x) > 0] - sum(dz')/P - xhat sum(dz' xhat)/P) -- what
 * gdl_bn_bwd_dx writes -- and it is only ever consumed by this gather.  The kernel forms dy per element from dz and the saved
 * convolution output x on the way into its staging buffer (one read of dz and x instead of read dz, read x, write dy, read dy).
 * Arguments as gdl_bn_bwd_dx (dgamma_sum / dbeta_sum = the outputs of gdl_bn_bwd_reduce, P_total their pixel count) +
 * gdl_resize_conv3x3_bwd_gather; coef_ws: 4 * N floats of scratch (16-byte aligned).  Only where
 * gdl_resize_conv3x3_bwd_gather_one_pass() == 1 (bf16, N % 64 == 0, factor 2 or 4); fails otherwise. */
int gdl_resize_conv3x3_bwd_gather_bn(const void* dz, const void* x, int dtype, int B, int Ho, int Wo, int N, void* g, int Hi, int Wi,
                                     const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                     int relu, const float* dgamma_sum, const float* dbeta_sum, int64_t P_total, float* coef_ws,
                                     gdl_stream_t stream);
/* the same result in two separable passes (rows, then columns) through a workspace of three [B,Hi,Wo,N] maps: fewer
 * multiply-adds per loaded vector; the intermediate is rounded to dy's dtype */
int64_t gdl_resize_conv3x3_bwd_gather_workspace(int dtype, int B, int Wo, int N, int Hi);
int gdl_resize_conv3x3_bwd_gather2(const void* dy, int dtype, int B, int Ho, int Wo, int N, void* g, int Hi, int Wi, void* ws,
                                   int64_t ws_bytes, gdl_stream_t stream);
/* FORWARD of conv3x3(pad 1)(F.interpolate(x, bilinear, factor 2/4/8)) at LOW resolution (multilevel_neck.py:157-158 levels
 * with scale 2 and 4; upernet.py:144-152 `fpn_bottleneck` over its upsampled levels): the resize U and the tap shifts S_t act
 * on pixels, the filter slices W_t on channels, so y = sum_t S_t U (W_t x).  The nine tap products z = [W_0 x, ..., W_8 x]
 * are one 1x1 gdl_conv_gemm over the low-resolution pixels (dense [B,hs,ws,9*N], tap block t = 3 r + s); this call forms
 * out[b,oy,ox,n] = addvec[n] + sum_k sum_(r,s) bilinear(zs[k]_(r,s))[oy+r-1, ox+s-1] (positions outside the output = the
 * convolution's zero padding) for 1..3 sources of one dtype, optional ReLU, dense [B,Ho,Wo,N].  Deterministic. */
int gdl_resize_conv3x3_fwd_sum(const void* const* zs, const int* hs, const int* ws, int nsrc, int dtype, int B, int N, void* out,
                               int Ho, int Wo, const float* addvec, int relu, gdl_stream_t stream);
/* The same sum for ONE source at ANY upsampling ratio (plain gather: 9 taps x 4 bilinear corners per output vector, f32
 * accumulation): the level of a pyramid whose size does not divide the output's -- DOFA-large's top FPN level is
 * int(73 * 0.5) = 36 wide against 292 (models/utils.py:106-110, upernet.py:144-152).  accumulate != 0: out += the sum (after
 * gdl_resize_conv3x3_fwd_sum wrote the integer-factor levels); addvec / relu as above, applied after the accumulation. */
int gdl_resize_conv3x3_fwd_sum_any(const void* z, int hs, int ws, int dtype, int B, int N, void* out, int Ho, int Wo, int accumulate,
                                   const float* addvec, int relu, gdl_stream_t stream);
/* the same with the train-mode BatchNorm statistics of the result as a side output (models/utils.py:10-52, ConvModule = conv ->
 * BatchNorm -> ReLU): per-channel mean / biased variance of `out` (and the running-stat update of nn.BatchNorm2d when the
 * running buffers are given) from per-block partial sums -- no separate pass over the output.  bf16, N % 64 == 0.
 * workspace: gdl_resize_conv3x3_fwd_sum_bn_rows(B, Ho, Wo) * 2 * N floats. */
int64_t gdl_resize_conv3x3_fwd_sum_bn_rows(int B, int Ho, int Wo);
int gdl_resize_conv3x3_fwd_sum_bn(const void* const* zs, const int* hs, const int* ws, int nsrc, int dtype, int B, int N, void* out,
                                  int Ho, int Wo, const float* addvec, float* workspace, int64_t ws_bytes, float* mean, float* var,
                                  float* running_mean, float* running_var, float momentum, gdl_stream_t stream);
/* final reduction of [nsplit][2][C] partial sums (sum, sum of squares over P pixels in total) into mean / biased variance
 * (+ running statistics): the second half of gdl_bn_stats, for producers that emit the partials themselves */
int gdl_bn_stats_finalize(const float* partials, int nsplit, int C, int64_t P, float* mean, float* var, float* running_mean,
                          float* running_var, float momentum, gdl_stream_t stream);
/* strided NHWC copy with dtype conversion: `x.to(dtype)` under autocast, `.contiguous()` of a channel slice
 * (models/utils.py:50-52 inputs, torch.cat slices of upernet.py:103-109 in backward) */
int gdl_copy_cast(const void* in, int in_dtype, int B, int H, int W, int C, int64_t in_sB, int64_t in_sH, int64_t in_sW,
                  void* out, int out_dtype, int64_t out_sB, int64_t out_sH, int64_t out_sW, gdl_stream_t stream);
/* out = sum_k bilinear(src_k -> Ho x Wo) for 1..3 DENSE sources [B, hs[k], ws[k], C] of one dtype, written once (dense
 * [B,Ho,Wo,C], same dtype).  segformer_mlp.py:97-125: `linear_fuse(cat([resize(_c4), resize(_c3), resize(_c2), _c1]))` is
 * evaluated per level at the level's own resolution (a 1x1 convolution commutes with the resize); this sums the upsampled
 * partial results, which then enter the finest level's GEMM as its residual operand. */
int gdl_bilinear_sum_fwd(const void* const* srcs, const int* hs, const int* ws, int nsrc, int dtype, int B, int C,
                         void* out, int Ho, int Wo, gdl_stream_t stream);
/* din (+)= bilinear^T(dout) */
int gdl_bilinear_bwd(const void* dout, int dout_dtype, int B, int Ho, int Wo, int C,
                     int64_t dout_sB, int64_t dout_sH, int64_t dout_sW, void* din, int din_dtype,
                     int Hi, int Wi, int64_t din_sB, int64_t din_sH, int64_t din_sW, int accumulate,
                     gdl_stream_t stream);
/* nn.AdaptiveAvgPool2d(S) (models/utils.py:73-79) */
int gdl_adaptive_avgpool_fwd(const void* in, int dtype, int B, int Hi, int Wi, int C, int64_t in_sB,
                             int64_t in_sH, int64_t in_sW, void* out, int out_dtype, int So,
                             gdl_stream_t stream);
int gdl_adaptive_avgpool_bwd(const void* dout, int dtype, int B, int So, int C, void* din,
                             int din_dtype, int Hi, int Wi, int64_t din_sB, int64_t din_sH,
                             int64_t din_sW, int accumulate, gdl_stream_t stream);

/* ---- attention -----------------------------------------------------------------------
 * qkv [B,N,3,H,hd] (output of timm Attention.qkv).  Q and K are consumed in place
 * (strided GEMM operands); only V is re-laid out: vt [B,H,hd,Npad] (keys >= N zero). */
int gdl_v_transpose(const void* v, int dtype, int B, int N, int H, int hd, int64_t v_sB, int64_t v_sN,
                    void* vt, int Npad, gdl_stream_t stream);
/* row softmax over the first n_valid of n_cols columns; pad columns written as 0 */
int gdl_softmax_rows(const void* in, void* out, int dtype, int64_t rows, int n_valid, int n_cols,
                     gdl_stream_t stream);
/* fused flash attention forward (bf16, hd = 64): F.scaled_dot_product_attention inside timm
 * Attention.  Reads q,k straight from qkv [B,N,3,H,64], v from vt; o[b, n, h*64 + d]. */
int gdl_flash_attn_fwd(const void* q, int64_t q_sB, int64_t q_sN, const void* k, int64_t k_sB,
                       int64_t k_sN, const void* vt, void* o, int B, int H, int Nq, int Nkv, int Npad,
                       float scale, gdl_stream_t stream);

/* Second-generation fused attention (csrc/attention_v2.hip): forward with V consumed row-major (transposed inside the
 * LDS by ds_read_b64_tr_b16 -- no V^T pass) and the log-sum-exp of the scaled scores as a second output
 * (lse [B,H,Nq] f32, may be NULL); backward of F.scaled_dot_product_attention (timm Attention, dofa_v2.py:248-263;
 * MiT Attention, mix_transformer.py:66-157) with the probabilities recomputed tile by tile from lse: nothing of size
 * Nq x Nkv is materialised.  q / o / dout / dq are [B,Nq,H*64] token rows, k / v / dk / dv [B,Nkv,H*64], all bf16 with
 * unit channel stride and arbitrary (16-byte aligned) batch / token strides, e.g. slices of a packed qkv tensor.
 * dvec [B,H,Nq] f32 is scratch (rowsum(dO * O)).  ws: gdl_flash_attn_bwd_workspace() bytes of 16-byte aligned
 * scratch (0 for most shapes, then ws may be NULL): with few keys and many queries (MiT's spatial reduction) the dK / dV
 * kernel cuts the query range into parts whose f32 partial sums are added in a fixed order.  Deterministic (no atomics). */
int gdl_flash_attn_fwd2(const void* q, int64_t q_sB, int64_t q_sN, const void* k, int64_t k_sB, int64_t k_sN,
                        const void* v, int64_t v_sB, int64_t v_sN, void* o, int64_t o_sB, int64_t o_sN, float* lse,
                        int B, int H, int Nq, int Nkv, float scale, gdl_stream_t stream);
int gdl_flash_attn_bwd(const void* q, int64_t q_sB, int64_t q_sN, const void* k, int64_t k_sB, int64_t k_sN,
                       const void* v, int64_t v_sB, int64_t v_sN, const void* o, int64_t o_sB, int64_t o_sN,
                       const void* dout, int64_t do_sB, int64_t do_sN, const float* lse, float* dvec,
                       void* dq, int64_t dq_sB, int64_t dq_sN, void* dk, int64_t dk_sB, int64_t dk_sN,
                       void* dv, int64_t dv_sB, int64_t dv_sN, int B, int H, int Nq, int Nkv, float scale,
                       float* ws, int64_t ws_bytes, gdl_stream_t stream);
int64_t gdl_flash_attn_bwd_workspace(int B, int H, int Nq, int Nkv);

/* ---- DOFA patch embed (dofa_v2.py:157-181) --------------------------------------------
 * im2col of conv2d(stride=P, padding=1, kernel P): in NCHW f32 [B,C,H,W] ->
 * cols [B*Gh*Gw][Kpad] (k = (c*P + r)*P + s, zero padded to Kpad), dtype out_dtype. */
int gdl_patchify(const float* img, int B, int C, int H, int W, int P, int stride, int pad, int Gh,
                 int Gw, void* cols, int out_dtype, int Kpad, gdl_stream_t stream);
/* depthwise 3x3 conv (pad 1, stride 1) + bias (+ exact-erf GELU) on NHWC -- the Mix-FFN DWConv of
 * SegFormer (mix_transformer.py:533-546 + :56-63).  w9 is [9][C] f32 (tap-major), bias [C]. */
int gdl_dwconv3x3(const void* in, int dtype, int B, int H, int W, int C, const float* w9,
                  const float* bias, int gelu, void* out, int out_dtype, gdl_stream_t stream);
/* generated kernel G [C][P*P][D] f32 (TransformerWeightGenerator.fc_weight output viewed as at
 * dofa_v2.py:157-166) -> GEMM weight [D][Kpad], k = c*P*P + r*P + s, times `scaler` (0.01) */
int gdl_dofa_pack_kernel(const float* g, int C, int PP, int D, float scaler, void* out,
                         int out_dtype, int Kpad, gdl_stream_t stream);
/* backward of gdl_dofa_pack_kernel: dg[C][PP*D] = scaler * transpose(dw[D][Kpad]) (f32) */
int gdl_dofa_unpack_grad(const float* dw, int C, int PP, int D, float scaler, int Kpad, float* dg, gdl_stream_t stream);

/* position_embedding (dofa_v2.py:9-35): out[m] = [sin(pos[m]*omega[d]) | cos(pos[m]*omega[d])], f32;
 * omega [D/2] is the module's constant frequency table (1 / 10000^(d/(D/2))). */
int gdl_sincos_embed(const float* pos, const float* omega, int M, int D, float* out, gdl_stream_t stream);
/* eval BatchNorm2d folded for the conv epilogue: scale = g/sqrt(var+eps), shift = b - mean*scale */
int gdl_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                int C, float* scale, float* shift, gdl_stream_t stream);
/* forward weights [N][T][C] -> data-gradient weights [C][T flipped][N] (autograd of F.conv2d wrt
 * its input is gdl_conv_gemm on these with pad' = R-1-pad) */
int gdl_pack_dgrad(const void* w, int w_dtype, int N, int T, int C, void* out, int out_dtype,
                   gdl_stream_t stream);

/* ---- elementwise ----------------------------------------------------------------------*/
int gdl_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, gdl_stream_t stream);
int gdl_scale_f32(const float* in, float* out, int64_t n, float s, gdl_stream_t stream);
/* out[r,:] = a[r % a_rows,:] + (b ? b[r % b_rows,:] : 0)  (token assembly / broadcast), f32 */
int gdl_add_rows(const float* a, int64_t a_rows, int64_t a_stride, const float* b, int64_t b_rows,
                 int64_t b_stride, float* out, int64_t out_stride, int64_t rows, int D,
                 gdl_stream_t stream);
/* u8 tile -> (x/255 - mean[c]) / std[c], NCHW f32 (utils/tensors.py:10-35, wds_dataset.py:230-236) */
int gdl_normalize_u8(const uint8_t* in, float* out, int B, int C, int64_t HW, const float* mean,
                     const float* std, gdl_stream_t stream);
/* the same for any raw sample dtype the reference's `.float()` accepts (wds_dataset.py:230-236) */
enum { GDL_RAW_U8 = 0, GDL_RAW_U16 = 1, GDL_RAW_I16 = 2, GDL_RAW_F32 = 3 };
int gdl_normalize_raw(const void* in, int kind, float* out, int B, int C, int64_t HW, const float* mean,
                      const float* stdv, gdl_stream_t stream);
/* x[o, :inner] *= s[o]  (per-sample DropPath scaling, timm DropPath) */
int gdl_scale_outer(void* x, int dtype, const float* s, int64_t outer, int64_t inner,
                    gdl_stream_t stream);

/* ---- classifier tail -------------------------------------------------------------------
 * 1x1 conv to K<=16 classes (+bias) on NHWC features -> f32 NHWC logits [P][K]
 * (segmentation_head.py:22-26, fcn_head.py:73, segformer_mlp.py:64-65).  Dense bf16 features with C = 128 / 256 / 512 / 768 / 1024 run
 * as a skinny MFMA GEMM (16-pixel tiles, f32 weights as bf16 hi + lo fragments: f32-grade logits; a Dropout2d chan_scale [B][C] is
 * folded into the weights per image when pix_per_img % 16 == 0 and C % 256 == 0); every other shape one wave per pixel: */
int gdl_head_1x1(const void* feat, int dtype, int64_t P, int C, int64_t f_sP, const float* w,
                 const float* bias, const float* chan_scale, int64_t pix_per_img, float* out, int K,
                 gdl_stream_t stream);
/* its backward: dfeat (may be NULL), dw [K][C], db [K] (may be NULL) */
int64_t gdl_head_1x1_bwd_workspace(int64_t P, int C, int K);
int gdl_head_1x1_bwd(const void* feat, int dtype, const float* dlog, int64_t P, int C, int64_t f_sP,
                     const float* w, const float* chan_scale, int64_t pix_per_img, void* dfeat,
                     int64_t d_sP, float* dw, float* db, int K, float* workspace,
                     int64_t workspace_bytes, gdl_stream_t stream);
/* bilinear (align_corners=False) NHWC [B,Hi,Wi,K] -> NCHW f32 [B,K,Ho,Wo] logits (dofa.py:89-105) */
int gdl_upsample_logits(const float* in, int B, int Hi, int Wi, int K, float* out, int Ho, int Wo,
                        gdl_stream_t stream);
int64_t gdl_upsample_logits_bwd_workspace(int B, int K, int Hi, int Wo);
int gdl_upsample_logits_bwd(const float* dout, int B, int Ho, int Wo, int K, float* din, int Hi,
                            int Wi, float* workspace, int64_t workspace_bytes, gdl_stream_t stream);
/* softmax(dim=1).argmax(dim=1) on NCHW f32 logits -> int64 mask (segmentation_dofa.py:281) */
int gdl_softmax_argmax(const float* logits, int B, int K, int64_t HW, int64_t* mask,
                       gdl_stream_t stream);
/* the same mask from the head's own NHWC f32 map [B,Hi,Wi,K]: softmax(dim=1).argmax(dim=1) of F.interpolate(logits, (Ho, Wo))
 * (dofa.py:89-95 + segmentation_dofa.py:278-281) without the resized tensor; bit-identical to gdl_upsample_logits +
 * gdl_softmax_argmax.  K >= 2 classes. */
int gdl_upsample_argmax(const float* low, int B, int Hi, int Wi, int K, int64_t* mask, int Ho, int Wo,
                        gdl_stream_t stream);
/* f.softmax(output, dim=1) (K > 1) / f.sigmoid (K == 1) of the exported inference model (tools/script_model.py:55-59) */
int gdl_class_probs(const float* logits, int B, int K, int64_t HW, float* probs, gdl_stream_t stream);
/* smp DiceLoss(mode="multiclass", smooth=0, eps=1e-7) forward+backward on NCHW f32 logits
 * (configs/dofa_config_RGB.yaml:58-61).  sums [3*K] = (intersection, sum p, count y)
 * is produced by fwd and consumed by bwd.  dlogits = upstream[0]*grad_scale*dL/dlogits. */
int64_t gdl_dice_loss_workspace(int B, int K, int64_t HW);
int gdl_dice_loss_fwd(const float* logits, const int64_t* target, int B, int K, int64_t HW, float eps,
                      float* sums, float* loss, float* workspace, int64_t workspace_bytes,
                      gdl_stream_t stream);
int gdl_dice_loss_bwd(const float* logits, const int64_t* target, int B, int K, int64_t HW, float eps,
                      const float* sums, const float* upstream, float grad_scale, float* dlogits,
                      int accumulate, gdl_stream_t stream);
/* The same loss WITHOUT the full-resolution logits (round 5): the reference's training step computes
 * DiceLoss(F.interpolate(head(x), size=image_size, mode="bilinear")) (dofa.py:89-105, segmentation_dofa.py:226-229) and only needs the
 * loss and its gradient.  low = the [B, Hi, Wi, K] f32 map of gdl_head_1x1; the bilinear logit of every [Ho, Wo] pixel is
 * evaluated on the fly (same expression as gdl_upsample_logits).  _fwd: sums / loss as gdl_dice_loss_fwd, ws of
 * gdl_dice_loss_lowres_workspace() bytes.  _bwd: dlow [B, Hi, Wi, K] f32 = d loss / d low in one pass (gather form, fixed order), scaled by
 * upstream[0] (device scalar, may be null) * grad_scale.  Upsampling factors up to 64 (DOFA's auxiliary head: 18 -> 512).  With K <= 8 and a workspace of
 * gdl_dice_loss_lowres_bwd_workspace() bytes the gradient is formed tile by tile (every full-resolution softmax evaluated once,
 * partial patches summed in a fixed order by a second kernel); otherwise by one gather kernel (ws may be null). */
int64_t gdl_dice_loss_lowres_workspace(int B, int K, int Ho, int Wo);
int gdl_dice_loss_lowres_fwd(const float* low, const int64_t* target, int B, int K, int Hi, int Wi, int Ho, int Wo, float eps,
                             float* sums, float* loss, float* ws, int64_t ws_bytes, gdl_stream_t stream);
int64_t gdl_dice_loss_lowres_bwd_workspace(int B, int K, int Hi, int Wi, int Ho, int Wo);
int gdl_dice_loss_lowres_bwd(const float* low, const int64_t* target, int B, int K, int Hi, int Wi, int Ho, int Wo, float eps,
                             const float* sums, const float* upstream, float grad_scale, float* dlow, float* ws, int64_t ws_bytes,
                             gdl_stream_t stream);

/* smp DiceLoss(mode="binary", smooth=0, eps=1e-7) on `total` = B*H*W logits of the single class (the reference's
 * UNet++ config: configs/unetplus_config_RGB.yaml:40-47 with num_classes 1; smp 0.5.0 losses/dice.py): p =
 * exp(logsigmoid(x)), sums over batch and pixels, loss = (1 - 2I/(sum p + sum y)) * [sum y > 0].  Workspace: the
 * multiclass one for K = 1 (gdl_dice_loss_workspace(B, 1, HW)); sums [3] = (I, sum p, sum y). */
int gdl_dice_binary_loss_fwd(const float* logits, const int64_t* target, int64_t total, float eps, float* sums,
                             float* loss, float* workspace, int64_t workspace_bytes, gdl_stream_t stream);
int gdl_dice_binary_loss_bwd(const float* logits, const int64_t* target, int64_t total, float eps,
                             const float* sums, const float* upstream, float grad_scale, float* dlogits,
                             int accumulate, gdl_stream_t stream);

/* ---- fused bilinear x4 upsample -> 3x3 conv (multilevel_neck.py:157-158, scale 4) -------------------------------
 * gdl_pad_nhwc: NHWC border padding by (pad_h, pad_w), replicate (zero_mode 0) or zeros (1): out [B,H+2ph,W+2pw,C] dense.
 * gdl_subpix4_weights: the 16 phase weight sets of the sub-pixel decomposition from the 3x3 weights w [N][9*C]
 * (f32, K order (dy,dx,c)): g22 / g23 / g32 / g33 = [4 phases][N][R*S*C] for the phase groups with R x S low-res
 * taps, lines = [4][N][3*C] (top, bottom, left, right border-line convolutions); see csrc/subpixel.hip. */
int gdl_pad_nhwc(const void* in, int dtype, int B, int H, int W, int C, int64_t in_sB, int64_t in_sH, int64_t in_sW,
                 void* out, int pad_h, int pad_w, int zero_mode, gdl_stream_t stream);
int gdl_subpix4_weights(const float* w, int N, int C, int out_dtype, void* g22, void* g23, void* g32, void* g33,
                        void* lines, gdl_stream_t stream);

/* ---- optimizer -------------------------------------------------------------------------
 * torch.optim.Adam step (configs/dofa_config_RGB.yaml:62-65) on one flat f32 tensor, with the
 * global-norm clip coefficient read from device memory (gradient_clip_val 1.0, :11). */
int gdl_sumsq(const float* x, int64_t n, float* out_accum, gdl_stream_t stream);
/* coef = min(1, max_norm / (sqrt(sumsq) + 1e-6))  (torch.nn.utils.clip_grad_norm_) */
int gdl_clip_coef(const float* sumsq, float max_norm, float* coef, gdl_stream_t stream);
/* Multi-tensor forms: ONE launch for all parameter tensors.  `table` is a DEVICE array of
 * nchunks rows {param*, grad*, exp_avg*, exp_avg_sq*, count, shadow*} (int64 each; pointers already offset
 * to the chunk, count <= 65536; all f32 dense; shadow = 0 or a bf16 copy of the parameter in the same element order, which
 * the update rewrites with the new values -- the compute-dtype GEMM operand of the next forward, so that no per-parameter
 * cast runs between steps).  step-wide bias corrections bc1/bc2. */
int gdl_multi_sumsq(const int64_t* table, int nchunks, float* out_accum, gdl_stream_t stream);
int gdl_multi_adam(const int64_t* table, int nchunks, float lr, float beta1, float beta2, float eps,
                   float weight_decay, float bc1, float bc2, const float* clip_coef,
                   gdl_stream_t stream);
/* capturable form (hipGraph replay of a whole training step, torch.optim.Adam(capturable=True)): step count and
 * hyper-parameters in DEVICE memory -- state = {step, lr, beta1, beta2, eps, weight_decay, bc1, bc2} f32; gdl_adam_tick
 * advances the step and refreshes the two bias corrections, gdl_multi_adam_dev reads everything from `state` */
int gdl_adam_tick(float* state, double beta1, double beta2, gdl_stream_t stream);   /* betas in double: bc = 1 - beta^step as the host computes it */
int gdl_multi_adam_dev(const int64_t* table, int nchunks, const float* state, const float* clip_coef, gdl_stream_t stream);
/* The bf16 GEMM operands DERIVED from 3x3 conv parameters, rebuilt in one launch behind the update (the reference has no
 * counterpart: cuDNN re-reads the f32 parameter; here they are the operands of models/decoders/upernet.py:144-152 and
 * models/necks/multilevel_neck.py:157-158 in their low-resolution forms and of every data gradient).  `table`: DEVICE array of
 * `rows` rows {src f32 [N][T][C], dst bf16, N, T, C, c0, Cs, mode, first_tile, tiles_c} (int64 each; a tile = 32 n x 32 c of
 * one tap, rows ordered by first_tile, total_tiles = their sum).  mode 0: dst[n][t][c-c0] (channel slice); 1: dst[t][n][c-c0]
 * (tap-major); 2: dst[c][T-1-t][n] (= gdl_pack_dgrad).  Rounding as gdl_cast: bit-identical to the separate launches. */
int gdl_multi_repack(const int64_t* table, int rows, int64_t total_tiles, gdl_stream_t stream);
int gdl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, float bc1, float bc2,
                  const float* clip_coef, gdl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GDLHIP_H_ */
