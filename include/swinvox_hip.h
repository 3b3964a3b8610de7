/*
 * swinvox_hip.h -- C ABI of libswinvox_hip.so, the MI355X (gfx950) implementation of the SwinVox
 * Encoder -> Decoder -> Merger -> Refiner forward/backward path.
 *
 * The reference (SandeepaInduwaraSamaranayake/SwinVox) has no FFI layer: its hot path is a chain of
 * stock PyTorch operators dispatched from its models/ package.  Each entry point below replaces one operator
 * class of that chain; the comment above every group cites the
 * reference file:line whose operator it stands in for.  The host side (the swinvox_amd/models package) binds
 * these through ctypes and mirrors the reference's nn.Module surface.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer; typed pointers are fp32 (or as declared), void* activation pointers hold the
 *     element type named by the call's `act_dtype` (SV_F32 / SV_BF16, see below); the library never allocates,
 *     frees or retains memory (all buffers, including workspaces, are owned by the caller);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only enqueues work;
 *   - return value: 0 on success, <0 on error (SV_ERR_*); sv_last_error() returns a message for the
 *     calling thread; no entry point aborts the process;
 *   - activations are channels-last: 2-D maps [N,H,W,C], 3-D grids [N,D,H,W,C], token lists [rows,C];
 *     a row stride `ld*` (in elements) lets a call read/write a column slice of a wider buffer;
 *   - re-entrant and thread-safe: no global mutable state besides the thread-local error string.
 */
#ifndef SWINVOX_HIP_H
#define SWINVOX_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SV_OK 0
#define SV_ERR_INVALID (-1)
#define SV_ERR_LAUNCH (-2)

enum { SV_ACT_NONE = 0, SV_ACT_RELU = 1, SV_ACT_GELU = 2, SV_ACT_LRELU = 3 };
#define SV_BN_SLOTS 16 /* BatchNorm statistic accumulators are [SV_BN_SLOTS][2*C] doubles (contention spreading) */
enum { SV_MATH_F32 = 0, SV_MATH_BF16 = 1, SV_MATH_FP8 = 2 }; /* MFMA input type of the contraction kernels; accumulation stays fp32.
   SV_MATH_FP8 (OCP e4m3, per-tile scales) is accepted by sv_window_attention_fwd only (QK^T and PV of BASELINE configuration 5);
   sv_window_attention_bwd treats it as SV_MATH_BF16 */
/* Storage type of ACTIVATIONS (and activation gradients) in HBM.  Entry points with an `act_dtype` argument take their
 * activation tensors as void*: every such tensor of one call has this element type.  Parameters, parameter gradients,
 * statistics (mean/rstd/scale/shift/sums), workspaces and drop-path scales are always fp32 (or double where noted).
 * Arithmetic is fp32 in both cases; SV_BF16 halves the HBM traffic of the path (values are rounded to nearest-even on store). */
enum { SV_F32 = 0, SV_BF16 = 1 };

const char* sv_last_error(void);
int sv_version(void);

/* ------------------------------------------------------------------------------------------------
 * Contraction engine (implicit GEMM on MFMA): stands in for nn.Linear / nn.Conv2d / nn.Conv3d /
 * nn.ConvTranspose3d forward, data-gradient and weight-gradient.  Reference call sites:
 * timm Linear layers behind models/swin_transformer.py:78; models/encoder.py:22-23,36,41-111;
 * models/cross_view_attention.py:38,46,49-53; models/decoder.py:24-46; models/merger.py:20-54;
 * models/refiner.py:21-70.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sv_geom {
  int N;              /* images (or rows for a Linear: N = rows, all grids 1x1x1) */
  int Di, Hi, Wi;     /* grid of the GATHERED tensor  */
  int Do, Ho, Wo;     /* grid of the PRODUCED tensor  */
  int Ci, Co;         /* channels of gathered / produced tensor (in-memory, i.e. including padding) */
  int kd, kh, kw;     /* kernel extent */
  int sd, sh, sw;     /* stride */
  int pd, ph, pw;     /* padding */
  int ldi;            /* element stride between consecutive positions of the gathered tensor (>= Ci) */
} sv_geom;

typedef struct sv_epilogue {
  const float* bias;     /* [Co] or NULL */
  const void* residual;  /* ACTIVATION (act_dtype): same positions as the output, row stride ldr, or NULL:  out = residual + scale*val */
  int ldr;
  const float* row_scale; /* optional per-image scale of val before the residual add (drop-path), index = row / rows_per_scale */
  int rows_per_scale;
  void* pre_act;         /* ACTIVATION (act_dtype), optional: receives val (after bias, before activation), same layout as out */
  double* stats;         /* optional [SV_BN_SLOTS][2*Co] DOUBLES: atomically accumulates sum and sum-of-squares of the stored output per channel */
  int act;               /* SV_ACT_* applied to val (after bias) */
  float slope;           /* LeakyReLU slope */
  const void* act_grad_src; /* ACTIVATION (act_dtype), optional: val *= act'(act_grad_src[pos]) (backward through an activation), layout of out */
  int act_grad_kind;     /* SV_ACT_* of that activation */
  int ldc;               /* output row stride (elements) */
  int col_off;           /* first output column */
} sv_epilogue;

/* w_packed: [Co][taps][Ci] produced by sv_pack_weight(s); fp32 with act_dtype = SV_F32, bf16 with act_dtype = SV_BF16.
 * gather form:  out[o, co] = sum_{tap,ci} in[o*s - p + tap, ci] * w[co, tap, ci]      (conv forward; tconv data-grad; Linear) */
int sv_conv_gather(const void* in, const void* w_packed, void* out, const sv_geom* g, const sv_epilogue* e,
                   int math, int act_dtype, void* stream);
/* Halo-tile kernels behind sv_conv_gather / sv_tconv_gather (bf16 storage + bf16 MFMA, plain store with optional statistics): 3 x 3 / stride 1 /
 * padding 1 with Ci = Co = 64 (forward and data gradient) and the 4 x 4 / pads (2, 1) ResNet stem on the space-to-depth image (Ci = 16, Co = 64)
 * keep the weights in LDS and read each input patch once instead of once per tap.  mode 0: off, 1 (default; SV_CONV_HALO in the environment):
 * calls of >= 256 tiles of 8 x 32 positions, 2: every call of those shapes.  Results differ from the gather engine's only in fp32 summation order. */
int sv_set_conv_halo(int mode);
int sv_conv_halo_mode(void);
int sv_set_conv_halo_wgrad(int mode); /* the same switch for the halo-tile WEIGHT gradient of the 3 x 3 / stride 1 convolutions with 64 k channels on both
                                         sides behind sv_conv_wgrad (SV_CONV_HALO_WGRAD; 1 = calls of >= 4 tiles per split) */
long long sv_conv_halo_launches(void); /* calls taken by the halo-tile kernels so far in this process (tests: the path they mean to exercise) */
/* 1 when sv_conv_gather would run this call on the wide dense kernel (256 x 128 tile, LDS-DMA operand ring: Linear / 1x1 layers with
 * bf16 storage, K % 32 == 0, K >= 128, Co >= 128, 8-aligned rows and >= 256 tiles), else 0.  SV_GEMM_WIDE=0 in the environment disables it. */
int sv_conv_gather_is_wide(const void* in, const void* w, void* out, const sv_geom* g, const sv_epilogue* e, int math, int act_dtype);
/* scatter-as-gather form: out[o, co] = sum_{tap,ci : (o+p-tap)%s==0} in[(o+p-tap)/s, ci] * w[co, tap, ci]
 * (transposed-conv forward; strided-conv data-grad), decomposed per output parity class            */
int sv_tconv_gather(const void* in, const void* w_packed, void* out, const sv_geom* g, const sv_epilogue* e,
                    int math, int act_dtype, void* stream);
/* 1 when sv_conv_wgrad would run this call on the wide weight-gradient kernel (256 x 128 tile of dW, LDS-DMA ring fed by producer waves:
 * Linear / 1x1 stride-1 layers with bf16 storage, rows % 64 == 0 and >= 16384, >= 128 8-aligned columns on both sides), else 0.
 * SV_WGRAD_WIDE=0 in the environment disables it. */
int sv_conv_wgrad_is_wide(const void* anchor, int lda, const void* gathered, const sv_geom* g, int cg_valid, int math, int act_dtype);
/* weight gradient: dw[ca, cg, tap] += sum_r anchor[r, ca] * gathered[r*s - p + tap, cg]   (fp32 atomics; dw pre-zeroed
 * or holding a running sum).  g->Do.. = anchor grid, g->Di.. = gathered grid, g->Co = anchor channels (row stride lda),
 * g->Ci = gathered channels (stride g->ldi); only cg < cg_valid is written; dw index = (ca*cg_valid + cg)*taps + tap.
 * When the kernel has more than one tap the partial sums are gathered in `workspace`
 * (sv_conv_wgrad_workspace_floats(g) floats, caller-owned, contents destroyed) and folded into dw by a second kernel. */
size_t sv_conv_wgrad_workspace_floats(const sv_geom* g);
int sv_conv_wgrad(const void* anchor, int lda, const void* gathered, float* dw, const sv_geom* g, int cg_valid,
                  float* workspace, float* dbias /* optional: dbias[ca] += sum_r anchor[r, ca] */, int math, int act_dtype,
                  void* stream);
/* LDS-halo MFMA stencils for 3x3x3 / stride 1 / pad 1 convolutions with <= 16 output channels per tile (merger.py:20-54),
 * bf16 operands.  x: channels-last positions with row stride ldx, cin_load (multiple of 4) elements read per position,
 * zero-extended to 16*groups channels; w_bf16: [16*ntiles16][27][16*groups] bf16 (forward: rows = output channels;
 * data-gradient: rows = input channels, taps flipped).  Writes out[pos*ldc + col_off + n] for n < cout
 * (= residual[pos*ldr + n] + value when residual != NULL); optional per-channel statistics as in sv_epilogue.stats
 * (ntiles16 == 1 only).  When ldc, col_off (and ldr) are multiples of 4 and the row has room, the columns
 * cout .. roundup4(cout)-1 are treated as PADDING of the row and written too (zero, + residual): 8/16-byte row stores.
 * Persistent kernel: one workgroup walks many 8x8x8 bricks (D % 8 == 0, H % 8 == 0, W % 8 == 0), prefetching the next brick while it contracts the current one.
 * Planar channel storage (the dense per-layer buffers that replace the reference's torch.cat, merger.py:84): with
 * x_plane_stride != 0 memory channel c of the input lives at x[(c / ldx) * x_plane_stride + pos*ldx + c % ldx]; with
 * out_plane_stride != 0 column n goes to out[(n / ldc) * out_plane_stride + pos*ldc + n % ldc] (col_off 0, no residual).
 * Strides in elements, multiples of 4; 0 = interleaved rows (sv_stencil3_wgrad: the same for its gathered operand x). */
int sv_stencil3_fwd(const void* x, int ldx, int cin_load, int groups, const void* w_bf16, int ntiles16,
                    const float* bias, void* out, int ldc, int col_off, int cout, const void* residual, int ldr,
                    double* stats, int I, int D, int H, int W, long long x_plane_stride, long long out_plane_stride, int act_dtype,
                    void* stream);
/* dw[co][ci][27] += sum_vox dy[vox][co] * x[vox + tap][c]; memory channel c maps to ci = (c / c_stride)*c_valid + c % c_stride;
 * optional dbias[co] += sum_vox dy[vox][co].  workspace: NULL (every workgroup adds into dw directly) or
 * sv_stencil3_wgrad_workspace_floats(cout, cin) floats, ZERO on entry: partial sums are spread over slot images and
 * folded into dw by a second small kernel (removes the ~1000-way atomic contention per weight) */
size_t sv_stencil3_wgrad_workspace_floats(int cout, int cin);
int sv_stencil3_wgrad(const void* x, int ldx, int cin_load, int groups, const void* dy, int lddy, int cout_load,
                      float* dw, float* dbias, float* workspace, int cout, int cin, int c_stride, int c_valid, int I, int D, int H, int W,
                      long long x_plane_stride, int act_dtype, void* stream);
/* ConvTranspose3d(kernel 4, stride 2, padding 1), 32 input channels -> 8 output channels (rows of 8 bf16; cout < 8 leaves zero columns),
 * bf16 operands, on an LDS halo brick: the decoder's last up-sampling layer (models/decoder.py:37-40).  x: [I][D][H][W][32];
 * w_packed: the forward pack [cout][64 taps][32] of sv_pack_weight (swap = 1); out: [I][2D][2H][2W][8] = value + bias;
 * stats as in sv_epilogue.stats.  D, H, W must be multiples of 4, 4, 8. */
int sv_tconv4s2_fwd(const void* x, const void* w_packed, const float* bias, void* out, double* stats, int I, int D, int H, int W,
                    int cin, int cout, void* stream);
/* dst[a][t][b] (b padded with zeros to pad_to) from the fp32 parameter src[a][b][t] (swap=0), or dst[b][t][a..pad_to] (swap=1);
 * dst elements are out_dtype (SV_F32 / SV_BF16) */
int sv_pack_weight(const float* src, void* dst, int A, int B, int T, int swap, int pad_to, int out_dtype, void* stream);
/* the same for a whole table of weights in ONE launch.  descs_dev: DEVICE array; rows_out / inner_out as derived by
 * sv_pack_weight (rows_out = swap ? B : A, inner_out = max(pad_to, swap ? A : B)); block0 = exclusive prefix sum of
 * ceil(rows_out*T*inner_out / sv_pack_weights_block_elems()) over the table, total_blocks = its grand total */
typedef struct sv_pack_desc {
  const float* src; void* dst;
  int A, B, T, swap, rows_out, inner_out, block0, reserved;
} sv_pack_desc;
int sv_pack_weights_block_elems(void);
int sv_pack_weights(const sv_pack_desc* descs_dev, int n, int total_blocks, int out_dtype, void* stream);
/* per-column sum over rows: out[c] (+)= sum_r x[r*ld + c]  (bias gradients) */
int sv_colsum(const void* x, int rows, int cols, int ld, float* out, int accumulate, int act_dtype, void* stream);
/* element-wise storage conversion (module boundaries: nn.Module inputs / outputs are fp32) */
int sv_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation.  Token LayerNorm = timm norm1/norm2/patch_embed.norm/downsample.norm (eps 1e-5); with
 * merge_H/merge_W > 0 the rows are the PatchMerging 2x2 gather (order h0w0,h1w0,h0w1,h1w1) of a
 * [I,merge_H,merge_W,C/4] map.  Image LayerNorm = nn.LayerNorm([C,H,W]) + Dropout(0.05) of
 * models/swin_transformer.py:64-69,82-89 on NHWC data with the affine pre-transposed to [HW,C].
 * BatchNorm = nn.BatchNorm2d/3d of models/encoder.py, cross_view_attention.py:56, decoder.py, merger.py,
 * refiner.py on channels-last [M,C] (biased batch variance, unbiased running update, eps 1e-5).
 * x / y / dx / dy / z / dz / residual / dres are activations (act_dtype); everything else is fp32 (double where declared).
 * ---------------------------------------------------------------------------------------------- */
int sv_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     long long rows, int C, float eps, int merge_H, int merge_W, int act_dtype, void* stream);
/* workspace: sv_layernorm_bwd_workspace_floats(C) floats, ZERO on entry (slot-spread dgamma/dbeta partial sums + ticket) */
size_t sv_layernorm_bwd_workspace_floats(int C);
int sv_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                     void* dx, float* dgamma, float* dbeta, float* workspace, long long rows, int C, int merge_H, int merge_W,
                     int accumulate_dx, int act_dtype, void* stream);
size_t sv_ln_image_workspace_floats(int I, int L);
int sv_ln_image_fwd(const void* x, const float* w, const float* b, void* y, float* meanrstd, float* workspace,
                    int I, int L, float eps, float drop_p, uint32_t seed, const uint32_t* seed_epoch, int act_dtype, void* stream);
int sv_ln_image_bwd(const void* dy, const void* x, const float* w, const float* meanrstd, void* dx, float* dw,
                    float* db, double* sums_ws /* [2*I] doubles */, int I, int L, float drop_p, uint32_t seed,
                    const uint32_t* seed_epoch, int act_dtype, void* stream);
int sv_bn_stats(const void* x, long long M, int C, int ld, double* sums, int act_dtype, void* stream);   /* sums: [SV_BN_SLOTS][2*C] doubles (slot 0 is used) */
int sv_bn_finalize(const double* sums, long long count, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float momentum, float eps, int training, float* scale, float* shift,
                   float* save_mean, float* save_rstd, int C, void* stream);
/* Padded rows: when C is not a multiple of 4, C <= 16 and every row stride involved is a multiple of 4 elements and
 * >= roundup4(C), sv_scale_shift_act and sv_bn_bwd treat the columns C .. roundup4(C)-1 as PADDING of the row: whatever they
 * hold on input is ignored (it may be uninitialised memory) and ZERO is written there (y / dx / dres are stored as whole
 * 4-channel vectors).  A 9-channel tensor kept
 * in 12-wide rows therefore never needs a separate zero fill (swinvox_amd/models/merger.py relies on this). */
int sv_scale_shift_act(const void* x, int ldx, const float* scale, const float* shift, const void* residual, int ldr,
                       void* y, int ldy, long long M, int C, int act, float slope, int act_dtype, void* stream);
/* ResNet stem: BatchNorm + activation + MaxPool2d(3, stride 2, padding 1) in one pass over the convolution's output x [N, H, W, C] (rows of exactly C
 * elements, C % 4 == 0, 256 % (C / 4) == 0): pooled / idx [N, (H+1)/2, (W+1)/2, C] (idx: arg-max tap 0..8 per element, first maximum in scan order).
 * The normalised activation is never stored.  sv_bn_maxpool_bwd: gradient of the pooled map -> dx (w.r.t. x) with the pool's backward computed on the
 * fly inside both passes of the BatchNorm backward; dgamma / dbeta +=; sums_ws: sv_bn_bwd_workspace_doubles(C) doubles, zero on entry.
 * Replaces sv_scale_shift_act + sv_maxpool2d_fwd and sv_maxpool2d_bwd + sv_bn_bwd of reference models/encoder.py:22-23 (resnet50 bn1 / relu / maxpool). */
int sv_bn_act_maxpool_fwd(const void* x, const float* scale, const float* shift, void* pooled, void* idx, int N, int H, int W, int C,
                          int act, float slope, int act_dtype, void* stream);
int sv_bn_maxpool_bwd(const void* dpooled, const void* idx, const void* x, const float* gamma, const float* save_mean, const float* save_rstd,
                      const float* fwd_scale, const float* fwd_shift, int N, int H, int W, int C, int act, float slope, int training,
                      void* dx, float* dgamma, float* dbeta, double* sums_ws, int act_dtype, void* stream);
/* The same fusion around MaxPool3d(2) (floor) for the Refiner's down-sampling layers (reference models/refiner.py:21-39): x [N, D, H, W, C] ->
 * pooled / idx [N, D/2, H/2, W/2, C] (tap = 4 dz + 2 dy + dx, as sv_maxpool3d_fwd); the backward writes dx for every input position, the planes of an
 * odd grid that no window covers included. */
int sv_bn_act_maxpool3d_fwd(const void* x, const float* scale, const float* shift, void* pooled, void* idx, int N, int D, int H, int W, int C,
                            int act, float slope, int act_dtype, void* stream);
int sv_bn_maxpool3d_bwd(const void* dpooled, const void* idx, const void* x, const float* gamma, const float* save_mean, const float* save_rstd,
                        const float* fwd_scale, const float* fwd_shift, int N, int D, int H, int W, int C, int act, float slope, int training,
                        void* dx, float* dgamma, float* dbeta, double* sums_ws, int act_dtype, void* stream);
size_t sv_bn_bwd_workspace_doubles(int C);   /* size of sums_ws below */
int sv_bn_bwd(const void* dz, int lddz, const void* z, int ldz, const void* x, int ldx, const float* gamma,
              const float* save_mean, const float* save_rstd, long long M, int C, int act, float slope, int training,
              void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta, double* sums_ws /* sv_bn_bwd_workspace_doubles(C) doubles, ZERO on entry */,
              const float* fwd_scale, const float* fwd_shift /* optional: with z == NULL the activation mask is recomputed as
              x*fwd_scale + fwd_shift > 0 (valid when the forward added no residual) - saves one tensor read per pass */,
              int act_dtype, void* stream);
/* Activation mask as SIGN WORDS instead of the stored output (BatchNorm in front of a residual sum: bn3 and the down-sampling branch of every
 * bottleneck, reference models/encoder.py:22-23): sv_scale_shift_act_signs also writes, per row and per 256 channels, four 64-bit words - bit l of
 * word j = (value of channel 256 b + 4 l + j before the activation) > 0 - i.e. [M][C/64] uint64, 1/16 of the bytes of a bf16 output;
 * sv_bn_bwd_signs takes its mask from them (and always returns the masked gradient dz * act' through dres, which its second pass reads back
 * instead of (dz, z)).  C % 256 == 0 (sv_bn_signs_supported), rows 4-aligned. */
int sv_bn_signs_supported(int C);
int sv_scale_shift_act_signs(const void* x, int ldx, const float* scale, const float* shift, const void* residual, int ldr, void* y, int ldy,
                             long long M, int C, int act, float slope, void* signs, int act_dtype, void* stream);
int sv_bn_bwd_signs(const void* dz, int lddz, const void* signs, const void* x, int ldx, const float* gamma, const float* save_mean,
                    const float* save_rstd, long long M, int C, int act, float slope, int training, void* dx, int lddx, void* dres, int lddres,
                    float* dgamma, float* dbeta, double* sums_ws, int act_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attention cores.  Window attention = timm WindowAttention + SwinTransformerBlock roll/partition/mask
 * (call site models/swin_transformer.py:78): qkv [I*H*W, 3C] rows in natural (h,w) order, columns
 * [q|k|v][head][32]; table [169, heads] (fp32 parameter); out [I*H*W, C].  Cross-view attention = models/
 * cross_view_attention.py:78-105: qkv [B*V*P, 3R] channels-last rows (view image, position), scores over
 * the V views of one sample scaled by 1/sqrt(head_dim*V).  qkv / out / dout / dqkv are activations (act_dtype;
 * SV_BF16 requires math = SV_MATH_BF16); softmax statistics and the bias-table gradient stay fp32.
 * ---------------------------------------------------------------------------------------------- */
int sv_window_attention_fwd(const void* qkv, const float* table, void* out, int I, int H, int W, int C, int heads,
                            int shift, int math, int act_dtype, void* stream);
/* workspace: NULL, or sv_window_attention_bwd_workspace_floats(heads) floats, ZERO on entry (slot images of dtable, folded
 * by a second small kernel; used by the bf16-MFMA kernels) */
size_t sv_window_attention_bwd_workspace_floats(int heads);
int sv_window_attention_bwd(const void* qkv, const float* table, const void* dout, void* dqkv, float* dtable, float* workspace,
                            int I, int H, int W, int C, int heads, int shift, int math, int act_dtype, void* stream);
int sv_cross_view_attention_fwd(const void* qkv, void* out, int B, int V, int P, int R, int heads, int act_dtype, void* stream);
int sv_cross_view_attention_bwd(const void* qkv, const void* dout, void* dqkv, int B, int V, int P, int R, int heads,
                                int act_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Layout / pooling / tail kernels (reference sites in the comments of csrc/elementwise.hip).  void* tensors are
 * activations (act_dtype); weights, their gradients, pooling indices and drop-path scales keep their declared types.
 * ---------------------------------------------------------------------------------------------- */
int sv_transpose(const void* src, void* dst, int batch, int R, int C, int lds, int ldd, long long src_bstride,
                 long long dst_bstride, int act_dtype, void* stream);                       /* dst[b][c][r] = src[b][r][c] */
int sv_add_n(const void* a, const void* b, const void* c, const void* d, void* out, long long M, int C, int ldo, int act_dtype, void* stream);
int sv_axpby(const void* a, const void* b, void* out, float alpha, float beta, long long n, int act_dtype, void* stream);
int sv_relu_bwd(const void* dy, const void* y, void* out, long long n, int act_dtype, void* stream);               /* refiner.py:51 (ReLU after layer5) */
int sv_maxpool2d_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int act_dtype, void* stream);   /* 3x3 s2 p1, encoder.py:23 (resnet maxpool) */
int sv_maxpool2d_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int act_dtype, void* stream);   /* writes every dx element (no pre-zeroing) */
int sv_avgpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int ldy, int col_off, int act_dtype, void* stream); /* encoder.py:123 */
int sv_avgpool2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int ldy, int col_off, int act_dtype, void* stream);
int sv_decoder_seed_fwd(const void* feat, void* out, int I, int C, int act_dtype, void* stream);                      /* decoder.py:59-67 */
int sv_decoder_seed_bwd(const void* dout, void* dfeat, int I, int C, int act_dtype, void* stream);
int sv_maxpool3d_fwd(const void* x, void* y, uint8_t* idx, int N, int D, int H, int W, int C, int act_dtype, void* stream); /* refiner.py:25,31,37 */
int sv_maxpool3d_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int D, int H, int W, int C, int act_dtype, void* stream);
/* seed_epoch (all stochastic entry points; may be NULL): device word mixed into `seed` on the device, so that launches replayed
 * from a captured hipGraph - whose scalar arguments are frozen - draw fresh masks when the caller advances the word per replay. */
int sv_dropout(const void* x, void* y, long long n, float p, uint32_t seed, const uint32_t* seed_epoch, int act_dtype, void* stream);            /* cross_view_attention.py:57,131 */
int sv_droppath_scale(float* scale, int I, float p, uint32_t seed, const uint32_t* seed_epoch, void* stream);                        /* timm DropPath */
int sv_rowscale(const void* x, const float* scale, void* y, long long rows, int C, int rows_per_scale, int act_dtype, void* stream);
int sv_dwconv2x2_fwd(const void* x, const float* w, const float* b, void* y, int I, int C, int act_dtype, void* stream); /* cross_view_attention.py:26-32,68 */
int sv_dwconv2x2_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* db, int I, int C, int act_dtype, void* stream);
int sv_upsample3to7_add_fwd(const void* small, const void* x, int ldx, void* y, int I, int C, int act_dtype, void* stream); /* cross_view_attention.py:110-120 */
int sv_upsample3to7_bwd(const void* dy, void* dsmall, int I, int C, int act_dtype, void* stream);
int sv_decoder_head_fwd(const void* x8, const float* w, const float* bias, void* raw12, void* vol, long long M, int act_dtype, void* stream); /* decoder.py:83-94 */
int sv_decoder_head_bwd(const void* draw12, const void* dvol, const void* x8, const float* w, void* dx8, float* dw, float* dbias,
                        long long M, int act_dtype, void* stream);
int sv_merge_views_fwd(const void* wlogit, const void* vol, void* out, int B, int V, int S, int act_dtype, void* stream);  /* merger.py:91-104 */
int sv_merge_views_bwd(const void* wlogit, const void* vol, const void* out, const void* dout, void* dwlogit, void* dvol,
                       int B, int V, int S, int act_dtype, void* stream);
/* Fused Swin MLP branch  x2 = x1 + s * fc2(GELU(fc1(LayerNorm(x1))))  (timm Mlp + norm2 + DropPath of a SwinTransformerBlock behind
 * models/swin_transformer.py:78) with no hidden activation in HBM: bf16 activations / bf16 MFMA only, C in {96, 128, 192}
 * (sv_swin_mlp_supported).  x1, x2, dx1, dx2: [M, C] bf16 token rows; w1 [4C, C], b1 [4C], w2 [C, 4C], b2 [C], ln_g / ln_b [C]: fp32
 * parameters in their native layouts; row_scale (may be NULL): one drop-path factor per rows_per_scale consecutive rows.
 *  sv_swin_mlp_pack:  w1, w2 -> `packs` (16 C^2 bf16 elements): four images in MFMA fragment order, read by _fwd and _bwd.
 *  sv_swin_mlp_fwd:   x2 from x1.
 *  sv_swin_mlp_bwd:   dx1 = dx2 + d(branch)/dx1, recomputing LayerNorm and the pre-activation; dgamma / dbeta += LayerNorm gradients.
 *  sv_swin_mlp_wgrad: dw1, db1, dw2, db2 += weight gradients, recomputing the hidden activation per hidden-unit chunk;
 *                     w1_rows = bf16 [4C][C] (fc1.weight), w2t_rows = bf16 [4C][C] (fc2.weight transposed).                          */
int sv_swin_mlp_supported(int C);
int sv_swin_mlp_pack(const float* w1, const float* w2, void* packs, int C, void* stream);
int sv_swin_mlp_fwd(const void* x1, void* x2, const float* ln_g, const float* ln_b, const void* packs, const float* b1, const float* b2,
                    const float* row_scale, int rows_per_scale, long long M, int C, float eps, void* stream);
int sv_swin_mlp_bwd(const void* x1, const void* dx2, void* dx1, const float* ln_g, const float* ln_b, const void* packs, const float* b1,
                    const float* row_scale, int rows_per_scale, float* dgamma, float* dbeta, long long M, int C, float eps, void* stream);
int sv_swin_mlp_wgrad(const void* x1, const void* dx2, const float* ln_g, const float* ln_b, const void* w1_rows, const void* w2t_rows,
                      const float* b1, const float* row_scale, int rows_per_scale, float* dw1, float* db1, float* dw2, float* db2,
                      long long M, int C, float eps, void* stream);

/* Fused attention branch of a stage-0 Swin block (timm SwinTransformerBlock._attn + the residual of forward(), reached from
 * models/swin_transformer.py:78): x1 = x + row_scale * proj(window_attention(qkv(LayerNorm(x)))) in one kernel, the roll / 7x7 partition /
 * reverse and the shift mask folded into token index math as in sv_window_attention_fwd.  Built for C = 96, 3 heads, bf16 token rows
 * (sv_swin_attn_block_supported): both weight matrices stay in LDS for the lifetime of a workgroup.
 *  x, x1: [I*H*W, C] bf16;  ln_g / ln_b [C], wqkv [3C, C], bqkv [3C], table [169, heads], wproj [C, C], bproj [C]: fp32 parameters in their
 *  native layouts;  row_scale (may be NULL): one drop-path factor per image.
 *  Side outputs for the (unfused) backward, all or none: ln1 [M, C] bf16 + mean / rstd [M] fp32 (LayerNorm output and statistics),
 *  qkv [M, 3C] bf16, att [M, C] bf16 (head outputs before the projection) - the tensors sv_layernorm_fwd, the qkv linear and
 *  sv_window_attention_fwd would have stored.                                                                                          */
int sv_swin_attn_block_supported(int C, int heads, int act_dtype, int math);
int sv_swin_attn_block_fwd(const void* x, const float* ln_g, const float* ln_b, const float* wqkv, const float* bqkv, const float* table,
                           const float* wproj, const float* bproj, const float* row_scale, void* x1, void* ln1, float* mean, float* rstd,
                           void* qkv, void* att, int I, int H, int W, int C, int heads, int shift, float eps, int act_dtype, void* stream);
/* Fused BACKWARD of the same branch (data path): reads dx1 [M,96], the qkv rows and LayerNorm statistics the forward stored, and x; writes
 * dqkv [M,288] (the engine's weight-gradient kernels read it: d qkv.weight / bias from (dqkv, ln1), d proj.weight / bias from (s dx1, att)) and
 * dx = dx1 + LayerNormBackward(dqkv Wqkv) [M,96]; accumulates into dgamma / dbeta of norm1 [96] and dtable [169,3].  dbr (optional) receives
 * s * dx1 when a drop-path scale is given (the projection's weight gradient needs it).  workspace: sv_window_attention_bwd_workspace_floats(3)
 * floats, zero on entry.  9 passes over the token map instead of the 17 of sv_conv_gather -> sv_window_attention_bwd -> sv_conv_gather ->
 * sv_layernorm_bwd.  Replaces the autograd of timm SwinTransformerBlock._attn + norm1 + DropPath behind reference models/swin_transformer.py:78
 * (reference core/train.py:272). */
int sv_swin_attn_block_bwd(const void* dx1, const void* qkv, const void* x, const float* mean, const float* rstd, const float* ln_g,
                           const float* wqkv, const float* wproj, const float* table, const float* row_scale, void* dqkv, void* dx,
                           void* dbr, float* dgamma, float* dbeta, float* dtable, float* workspace, int I, int H, int W, int C,
                           int heads, int shift, int act_dtype, void* stream);

/* Layout kernels of the ResNet stem (models/encoder.py:22: Conv2d(3, 64, 7, stride 2, pad 3) as a 4x4 / stride-1 convolution on the
 * space-to-depth image [I][112][112][(sy, sx, c) = 16]) and of the merger's stencil weights (merger.py:20-54):
 *  sv_stem_space_to_depth: images [I,3,224,224] -> x16;  sv_stem_pack: w [64,3,7,7] fp32 -> [64][16 taps][16] (out_dtype);
 *  sv_stem_unpack_grad: dw [64,3,7,7] += dw16 [64][16][4][4];
 *  sv_merger_pack: w [cout][cin][27] fp32 -> bf16 forward pack [16][27][16 | 48] or data-gradient pack [16 | 48][27][16] (taps flipped);
 *                  concat = 1: the 36 input channels sit at columns 12 g + j of the four 12-wide planes.                              */
/* Encoder input in one pass: images [I, 3, S, S] (fp32 when images_f32, else act_dtype; S % 4 == 0) -> x16 [I, S/2, S/2, 16] (the stem's space-to-depth
 * image, as sv_stem_space_to_depth) and xp [I, S/4, S/4, 48] with xp[.., (ky, kx, c)] = images[c][4 py + ky][4 px + kx]: the rows on which timm's
 * PatchEmbed Conv2d(3, C, kernel 4, stride 4) (behind reference models/swin_transformer.py:78) is a Linear(48, C) with the weight re-indexed
 * [co][c][ky][kx] -> [co][(ky, kx, c)].  Replaces the fp32 -> storage cast, the NCHW -> NHWC transpose and sv_stem_space_to_depth. */
int sv_encoder_prep(const void* images, int images_f32, void* x16, void* xp, int I, int S, int act_dtype, void* stream);
/* Refiner head Conv3d(1, Co, k = 4, p = 2) on a D^3 grid (reference models/refiner.py:21-26) as a (4, 1, 1)-tap convolution over 16 channels (the stem's
 * trick): xc [N, D, D+1, D+1, 16] with xc[.., Y, X, 4 cy + cx] = x[.., Y + cy - 2, X + cx - 2] (zero outside); sv_head_unpack_dx folds the
 * 16-channel data gradient of that convolution back into dx [N, D, D, D]. */
int sv_head_pack_x(const void* x, void* xc, int N, int D, int act_dtype, void* stream);
int sv_head_unpack_dx(const void* dxc, void* dx, int N, int D, int act_dtype, void* stream);
int sv_stem_space_to_depth(const void* images, void* x16, int I, int act_dtype, void* stream);
int sv_stem_pack(const float* w, void* wp, int out_dtype, void* stream);
int sv_stem_unpack_grad(const float* dw16, float* dw, void* stream);
int sv_merger_pack(const float* w, void* wp_bf16, int cout, int cin, int dgrad, int concat, void* stream);

/* harness-side kernels on fp32 module outputs */
int sv_mean_views(const float* vol, float* out, int B, int V, int S, void* stream);                      /* core/train.py:246 */
int sv_bce_logits(const float* x, const float* t, long long n, float* loss_accum, float* dx, const float* gscale_dev, void* stream); /* core/train.py:165,249,255 */
int sv_iou_counts(const float* logits, const float* gt, const float* thresholds_dev, int nth, int B, int S, float* counts, void* stream); /* core/test.py:141-163: counts[B][nth][4] = {intersection = TP, union, FP, FN} of sigmoid(logits) >= th vs gt, nth <= 8 */

/* Optimiser step on one flat fp32 buffer per module (parameters, gradients and moments share one layout).
 *  sv_grad_sumsq: slots16[blockIdx & 15] += sum (g * gscale)^2 in double; the caller zeroes the 16 slots.  It is the squared
 *    L2 norm torch.nn.utils.clip_grad_norm_ takes over a module's gradients (core/train.py:279-282).
 *  sv_adam_step / sv_sgd_step: torch.optim.Adam (coupled L2 weight decay, no amsgrad) / torch.optim.SGD (momentum) as
 *    configured at core/train.py:98-131 and stepped at :287-292.  Every gradient is first multiplied by
 *    gscale * min(1, max_norm / (sqrt(sum slots16) + 1e-6)) - the data-parallel mean and the clip coefficient, read on the
 *    device; max_norm <= 0 disables clipping.  `step` counts the calls from 1.  When slots16 is given and its sum is not
 *    finite (an inf / NaN gradient element) the step is SKIPPED on the device: p, m, v stay untouched and
 *    skipped_steps[0] (device counter, may be NULL) += 1 - the behaviour of the reference's GradScaler.step() after
 *    unscale_ (core/train.py:276-293).  Bias correction / "first step" use step - skipped_steps[0].                  */
int sv_grad_sumsq(const float* g, long long n, float gscale, double* slots16, void* stream);
int sv_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
                 double weight_decay, long long step, float gscale, const double* slots16, float max_norm,
                 long long* skipped_steps, void* stream);
int sv_sgd_step(float* p, const float* g, float* momentum_buf, long long n, double lr, double momentum, double weight_decay,
                long long step, float gscale, const double* slots16, float max_norm, long long* skipped_steps, void* stream);

/* Input preparation on the device (the reference does it per sample on the CPU in DataLoader workers).
 *  sv_binvox_decode: utils/binvox_rw.py:118-149 (read_as_3d_array) for B volumes of equal dims: `rle` holds the concatenated
 *    (value, count) byte pairs that follow the "data" header line, volume b owning pairs [pair_offsets[b], pair_offsets[b+1]);
 *    out[b] = float(np.repeat(values, counts) != 0).reshape(d0, d1, d2), transposed (0, 2, 1) when fix_coords (the loader's
 *    default, utils/data_loaders.py:83-86).  decoded[b] = number of voxels the runs describe; the caller rejects a volume
 *    whose count differs from d0*d1*d2 (the reference's reshape raises).
 *  sv_augment_views: utils/data_transforms.py applied in the order of core/train.py:44-59 to I = B*V renderings stored as
 *    8-bit [I][Hs][Ws][C] (C = 4 with alpha, or 3), without a bounding box: centre crop (crop_h, crop_w) when the image is
 *    larger, cv2.resize INTER_LINEAR to (out_h, out_w), RandomBackground (alpha == 0 -> bg), ColorJitter, RandomNoise,
 *    Normalize, RandomFlip, RandomPermuteRGB, ToTensor -> out [I][3][out_h][out_w] fp32.  One sv_aug_sample per SAMPLE (the
 *    reference draws these once per __getitem__ and shares them between the V views), one flip byte per IMAGE.  The
 *    validation pipeline (:60-65) is the same call with identity jitter / zero noise / no flips / identity permutation.
 *    grey_sum_ws: I doubles of scratch (mean grey level per image for the contrast step).                               */
typedef struct sv_aug_sample {
  float bg[3];            /* background colour in [0, 1], channel order of the stored image                              */
  float jitter_value[3];  /* blend factors: [0] brightness, [1] contrast, [2] saturation (1 = unchanged)                 */
  int jitter_order[3];    /* the order the three adjustments are applied in (a permutation of 0, 1, 2)                    */
  float noise[3];         /* RandomNoise offset per stored channel (noise_rgb reversed, data_transforms.py:386-390)       */
  int perm[3];            /* output channel c = stored channel perm[c]                                                    */
  float mean[3], std[3];  /* Normalize                                                                                    */
} sv_aug_sample;
int sv_binvox_decode(const unsigned char* rle, const long long* pair_offsets, int B, int d0, int d1, int d2, int fix_coords,
                     float* out, int* decoded, void* stream);
int sv_augment_views(const unsigned char* src, int I, int V, int Hs, int Ws, int C, int crop_h, int crop_w, int out_h, int out_w,
                     const sv_aug_sample* params_dev, const unsigned char* flip_dev, double* grey_sum_ws, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SWINVOX_HIP_H */
