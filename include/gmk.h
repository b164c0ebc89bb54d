/* gmk.h — C ABI of libgmk.so: MI355X (gfx950) kernels for the DDPM train + sample hot path.
 *
 * The reference (matwilso/generative_models) has no FFI: every arithmetic op of its diffusion path is a
 * stock torch op.  Each entry point below names the reference call site(s) it replaces (paths relative to
 * the reference checkout).  Conventions:
 *   - returns 0 on success, GMK_ERR_ARG (<0) for an argument error, or a positive hipError_t;
 *     gmk_last_error() returns a per-thread message.  No exceptions, no allocation, no synchronisation.
 *   - every pointer is a BORROWED device pointer (e.g. torch `tensor.data_ptr()`), contiguous, 16-byte aligned;
 *     workspaces are passed in by the caller; all work is enqueued on `stream` (a hipStream_t).
 *   - re-entrant: callable from any host thread (autograd's backward thread, one process per GPU).
 *   - activations inside the network are NHWC `[B][H][W][C]` in `dtype` (GMK_F32, GMK_BF16 or GMK_F16), C a
 *     multiple of 128; a kernel that reads forward activations next to gradients takes the two types separately (`x_dtype`); parameters, statistics, embeddings and everything in the diffusion algebra are fp32;
 *     images at the model boundary are NCHW fp32 exactly as the reference passes them.
 */
#ifndef GMK_H
#define GMK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMK_F32 0
#define GMK_BF16 1
#define GMK_F16 2   /* fp16 storage: forward activations and forward weight packs of the 16-bit mode (gradients stay GMK_BF16); the
                     * reference's GPU path computes its forward under fp16 autocast (diffusion_model.py:68) */
#define GMK_ERR_ARG (-1)

/* gather modes of the implicit-GEMM convolution (how an output pixel + filter tap maps to a source pixel) */
#define GMK_CONV_NORMAL 0      /* stride 1                      simple_unet.py:163,172,117 */
#define GMK_CONV_STRIDE2 1     /* stride 2 (Downsample)         simple_unet.py:81,97,100   */
#define GMK_CONV_UPSAMPLE2 2   /* nearest x2 folded into a stride-1 conv   simple_unet.py:120-121 */
#define GMK_CONV_TRANSPOSED2 3 /* data-gradient of GMK_CONV_STRIDE2 (a transposed convolution) */

int gmk_version(void);
const char* gmk_last_error(void);
/* profiling aid: which kernel the calling thread's last gmk_conv_igemm / gmk_conv_wgrad / gmk_gn_* call launched
 * (1 conv_igemm_kernel, 2 conv_igemm_dma_kernel, 3 conv3x3_halo_kernel, 4 conv3x3_halo_ws_kernel, 7 conv3x3_halo_ws_kernel with the folded 1x1 skip convolution, 8 / 9 / 10 conv_subpixel_ws_kernel (upsample / transposed / upsample data gradient), 5 a halo kernel on the zero-stuffed source of GMK_CONV_TRANSPOSED2 (GMK_CONV_KERNEL=3), 6 the four parity-phase launches of the LDS-DMA kernel for GMK_CONV_TRANSPOSED2, 11 conv_wgrad_kernel,
 * 12 conv_wgrad_slots_kernel, 13 conv_wgrad_slots_ws_kernel, 17 its four-plane form for GMK_CONV_STRIDE2, 16 conv_wgrad_subpixel_ws_kernel, 14 conv1x1_pair_stream_kernel, 15 conv1x1_wgrad_stream_kernel, 21 gn_silu_fwd_reg_kernel, 22 gn_silu_fwd_kernel, 23 gn_silu_bwd_hybrid_kernel, 24 gn_silu_bwd_kernel) */
int gmk_last_kernel(void);
/* development aid: force kernel variants (0 = automatic; see GMK_CONV_KERNEL / GMK_WGRAD_KERNEL / GMK_GN_KERNEL); -1 = unset */
int gmk_set_kernel_choice(int conv, int wgrad, int gn);
/* development aid: a free integer (GMK_DEV_VARIANT) that experimental code paths may read for in-process A/B runs
 * (tools/small_batch_probe.py, tools/step_stamps.py); 0 / unset = the shipped behaviour */
int gmk_set_dev_variant(int v);
/* development aid (tools/step_stamps.py): while a device buffer of at least 512 bytes per workgroup is set, the wave-specialised 3x3 halo
 * launches (gmk_conv_igemm's 16-bit stride-1 3x3 path, gmk_conv3x3_skipfold) run an instrumented instantiation whose wave 4 (producer) and
 * wave 0 (consumer) write per-K-step shader-cycle sums [workgroup][2][64] into it; buf = NULL returns to the shipped kernels */
int gmk_dev_set_stamp_buffer(void* buf, int64_t bytes);
/* number of CUs the persistent convolution kernels may occupy (8..256, default 256 = the whole chip).  New with the build (the
 * reference has no multi-GPU path): data-parallel runs leave a few CUs to RCCL's all-reduce kernels, which otherwise queue
 * behind a chip-filling persistent grid */
int gmk_set_cu_limit(int n);
int gmk_get_cu_limit(void);
/* fp32 mode (dtype GMK_F32: the reference's own precision, gms/main.py:161-168 runs the test loss in fp32).  Default: exact fp32 MFMA chains
 * (v_mfma_f32_32x32x2_f32) - the parity mode, 1e-3 on every reference vector including the ill-conditioned closed-form set.  Round 6 adds a
 * fast form, gmk_set_fp32_exact(0) / GMK_FP32_SPLIT=1: the convolutions and weight gradients split each fp32 operand into bf16 hi + lo and form
 * a product as hi hi + hi lo + lo hi on the bf16 matrix cores with fp32 accumulation (~ 1e-5 per product, a fifth of the exact chains' matrix
 * time; the fp32 weight packs are then stored split, so re-pack after switching).  gmk_fp32_split() says which is active. */
int gmk_set_fp32_exact(int exact);
int gmk_fp32_split(void);
/* number of bytes of scratch gmk_conv_wgrad needs for the given problem (split-K slabs) */
int64_t gmk_conv_wgrad_workspace_bytes(int64_t n_pixels, int taps, int cout, int ktot);

/* ---- parameter packing -------------------------------------------------------------------------------
 * nn.Conv2d weight `[Cout][Cin][k][k]` fp32 (simple_unet.py:81,117,163,172,177) ->
 *   w_fwd   `[tap][Cout][Cin]`  dtype  (K-contiguous rows for the forward implicit GEMM)
 *   w_dgrad `[tap'][Cin][Cout]` dtype, tap' = spatially flipped tap (rows for the data-gradient GEMM)
 * either output may be NULL. */
int gmk_pack_conv_weight(const float* w, void* w_fwd, void* w_dgrad, int cout, int cin, int ksize, int dtype,
                         void* stream);
/* the same for up to 40 convolutions in ONE launch (the re-pack after every optimiser step): tensor e is the fp32
 * weight at arena + w_off[e] (elements) and is written as [w_fwd | w_dgrad], n = cout*cin*ksize^2 elements each, at
 * packs + pack_off[e] (elements of `dtype`).  fwd_f16 (optional; with dtype GMK_BF16 only): entries with a non-zero flag get their
 * w_fwd half in fp16 (same 2-byte slots), the operand type of the 16-bit mode's forward.  The six tables are HOST arrays of `count` ints. */
int gmk_pack_conv_weights_multi(const float* arena, void* packs, int count, const int* w_off, const int* pack_off,
                                const int* cout, const int* cin, const int* ksize, const int* fwd_f16, int dtype, void* stream);

/* ---- GroupNorm + SiLU (simple_unet.py:39-40,161-162,169-170; always adjacent in the reference) ---------
 * x,y: NHWC [B][HW][C]; `groups` groups over these C channels (a 2C-channel concatenated input is handled as
 * two calls of 16 groups each, simple_unet.py:150,161); mean/rstd: fp32 [B][groups] (written).
 * groups < 0: -groups channels per group (1..16, any size: nn.GroupNorm(32, 96) has 3), ceil(C / -groups) groups whose last one may be
 * partial - for zero-padded widths, where it lies in the padding; mean/rstd then have ceil(C / -groups) columns.  Same convention in
 * gmk_gn_silu_bwd. */
int gmk_gn_silu_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                    int B, int HW, int C, int groups, float eps, const float* stats_part, int tile_pixels, int ntiles,
                    float drop_p, uint64_t drop_seed, uint64_t drop_offset, const float* xadd, int xadd_stride, int dtype,
                    void* stream);
/* xadd (optional, fp32 [B][xadd_stride]): x enters as x[b][p][c] + xadd[b][c] — the producing convolution's bias and the
 * embedding broadcast-add of simple_unet.py:183 (`h + emb_out[..., None, None]`) are applied here, in the HBM-bound kernel,
 * instead of in the MFMA kernel's epilogue; pass the same pointer to gmk_gn_silu_bwd.
 * stats_part (optional): partial sums emitted by the producing convolution (see gmk_conv_igemm gn_stats); when given,
 * the statistics pass over x is skipped (x is read once).
 * drop_p > 0: nn.Dropout(p) behind the SiLU (simple_unet.py:171, training mode): element e of y (NHWC order) is kept and
 * scaled by 1/(1-p) iff gmk_rng_uniform(seed = drop_seed, offset = drop_offset)[e] >= drop_p, else zeroed; pass the same
 * triple to gmk_gn_silu_bwd. */
/* backward of the above.  dx = d/dx + dadd1 + dadd2 (optional NHWC addends, e.g. the identity-skip gradient).
 * dgamma_part/dbeta_part: fp32 [B][C] per-sample partials (reduce with gmk_colsum); dxsum: optional fp32
 * [B][dxsum_stride] per-sample channel sums of the final dx (bias / embedding gradients).
 * dtype: type of dy / dadd1 / dadd2 / dx; x_dtype: type of the saved input x (the same, or GMK_F16 next to GMK_BF16 gradients). */
int gmk_gn_silu_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                    const float* rstd, const void* dadd1, const void* dadd2, void* dx, float* dgamma_part,
                    float* dbeta_part, float* dxsum, int dxsum_stride, int B, int HW, int C, int groups,
                    float drop_p, uint64_t drop_seed, uint64_t drop_offset, const float* xadd, int xadd_stride, int dtype,
                    int x_dtype, void* stream);
/* dst[i] = (dst type) src[i], n a multiple of 8: fp16 <-> bf16 storage conversion (fp16 results saturate at the largest finite value).
 * No reference call site: the self-attention extension (north_star, BASELINE configs[4]) keeps bf16 internals and converts the fp16
 * forward stream at its boundary. */
int gmk_cast16(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);
/* out[b][c] = sum over pixels of x[b][:, c]  (NHWC, fp32 result [B][out_stride]) */
int gmk_chansum(const void* x, float* out, int out_stride, int B, int HW, int C, int dtype, void* stream);
/* out[c] (+)= sum_r part[r*stride + c], r < R, c < C  (fp32) */
int gmk_colsum(const float* part, int64_t stride, float* out, int R, int C, int accumulate, void* stream);
/* up to 16 independent gmk_colsum problems in one launch (host arrays of length n) */
int gmk_colsum_multi(int n, const float* const* part, const int64_t* stride, float* const* out, const int* R,
                     const int* C, void* stream);
/* 2x2 sum-pool NHWC [B][2H][2W][C] -> [B][H][W][C]: backward of F.interpolate(nearest, x2), simple_unet.py:120 */
int gmk_sumpool2x2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);

/* ---- 3x3 / 1x1 convolution, C_out = 128-wide tiles, MFMA implicit GEMM ---------------------------------
 * out[b][oy][ox][n] = bias[n] + emb[b*emb_stride + n] + residual[b][oy][ox][n]
 *                     + sum_{tap,k} srcK[b][sy][sx][k] * w[tap][n][k]
 * src0/src1: NHWC sources of c0 / c1 channels (c1 = 0: one source) — the channel concatenation
 * torch.cat([x, skip], 1) (simple_unet.py:150) is never materialised.  (hs, ws): source spatial size;
 * (ho, wo): output spatial size; `mode` one of GMK_CONV_*.  w: packed rows `[tap][w_rows][c0+c1]` of
 * `dtype`; output channels n0 .. n0+cout-1 use rows n0.. of each tap (cout % 128 == 0).  bias/emb/residual
 * may be NULL.  Forward convolutions pass a w_fwd pack; data gradients pass a w_dgrad pack.
 * gn_stats (optional, may be NULL): the convolution that PRODUCES a GroupNorm input can emit that GroupNorm's
 * statistics from its epilogue, saving the consumer one full read of the tensor.  Only conv3x3_halo_kernel
 * (gmk_last_kernel() == 3) fills it: fp32 [ntiles][8][2][cout/4][2] = for every 32-pixel group of every tile
 * (tile = 256/wo whole rows of the global row list b*ho + y) and each of the <= 2 samples the group touches, the sum
 * and the sum of squares of the stored (rounded) outputs of every 4-channel unit.  Pass the same buffer to
 * gmk_gn_silu_fwd together with tile_pixels = (256/wo)*wo and ntiles = ceil(B*ho / (256/wo)). */
int gmk_conv_igemm(const void* src0, const void* src1, int c0, int c1, int B, int hs, int ws, int ho, int wo,
                   int ksize, int mode, const void* w, int w_rows, int n0, int cout, const float* bias,
                   const float* emb, int emb_stride, const void* residual, void* out, int out_cstride,
                   float* gn_stats, int64_t gn_stats_bytes, const float* gn_scale, const float* gn_shift, int gn_stride,
                   int dtype, void* stream);
/* gn_scale / gn_shift (optional, fp32 [B][gn_stride], gn_stride >= c0 + c1): the sources are RAW tensors and the convolution
 * applies nn.GroupNorm + nn.SiLU to them on the way into LDS, y = silu(x * gn_scale[b][k] + gn_shift[b][k]) with the tables of
 * gmk_gn_stats (simple_unet.py:161-163,169-172: the normalised tensor is never materialised).  Only where
 * gmk_conv_gn_fusable(...) returns 1; otherwise the call fails. */
int gmk_conv_gn_fusable(int B, int H, int W, int c0, int c1, int cout);
/* 1x1 convolution with TWO 128-channel output blocks (packed weight rows n0 .. n0+127 -> out_a, n0+128 .. n0+255 -> out_b, both
 * NHWC [B][H][W][128]) from ONE pass over the source: the data gradient of the up-path ResBlocks' 1x1 skip_connection
 * (simple_unet.py:176, Conv2d(2C, C, 1)) with respect to the two halves of its concatenated input (torch.cat, :141-147).
 * Two gmk_conv_igemm calls would read the output gradient twice. */
int gmk_conv1x1_pair(const void* src, int c, int B, int H, int W, const void* w, int w_rows, int n0, void* out_a, void* out_b,
                     int dtype, void* stream);
/* conv2 of an up-path ResBlock with its 1x1 skip convolution folded in - `skip_connection(x) + h` of simple_unet.py:174-186 as ONE launch
 * (the reference evaluates Conv2d(2C, C, 1) on torch.cat([x, skip]) and adds the result to out_layers' 3x3 convolution, :177-186):
 *   out[b][y][x][n] = bias[n] + bias_sk[n] + sum_{tap,k} src[b][y+dy][x+dx][k] * w[tap][n0+n][k]
 *                                          + sum_k cat(sk0, sk1)[b][y][x][k] * wsk[nsk0+n][k]
 * src: NHWC, c0 channels (the GroupNorm+SiLU output, simple_unet.py:169-172); sk0 / sk1: the two NHWC halves of the block input, cs
 * channels each; w: packed `[9][w_rows][c0]`, wsk: packed `[wsk_rows][2 cs]` (a w_fwd pack of the 1x1 convolution).  The skip output is
 * never written to or read back from HBM and is added in fp32 (the two-launch path rounds it to 16 bits in between).  Shapes: where
 * gmk_conv3x3_skipfold_ok(...) returns 1 (c0 = cs = cout = 128, a problem the LDS-halo kernel takes); otherwise the call fails. */
int gmk_conv3x3_skipfold_ok(int B, int H, int W, int c0, int cs, int cout);
int gmk_conv3x3_skipfold(const void* src, int c0, int B, int H, int W, const void* w, int w_rows, int n0, int cout,
                         const float* bias, const void* sk0, const void* sk1, int cs, const void* wsk, int wsk_rows, int nsk0,
                         const float* bias_sk, void* out, int out_cstride, int dtype, void* stream);
/* Sub-pixel (output-parity) form of the two x2 resampling convolutions: a LOW-resolution source [B][H][W][cin] -> out [B][2H][2W][cout].
 *   GMK_SUBPIXEL_UPSAMPLE    `Upsample` (simple_unet.py:112-122: F.interpolate(nearest, x2), then Conv2d(C, C, 3, padding=1)).  Output pixel
 *                            (2i + a, 2j + b) only meets low-resolution rows {i + a - 1, i + a} and columns {j + b - 1, j + b}: w is the pack of
 *                            gmk_pack_upsample_weight, `[4 (2a + b) + 2 ty + tx][w_rows][cin]` = the 3x3 weights that land on one low-resolution
 *                            pixel summed in fp32 and rounded once - 16 tap-products per low-resolution pixel instead of the 36 of
 *                            gmk_conv_igemm(GMK_CONV_UPSAMPLE2) (same sum, other rounding order: not bit-identical to it)
 *   GMK_SUBPIXEL_TRANSPOSED  data gradient of `Downsample` (simple_unet.py:75-84, Conv2d(C, C, 3, stride=2, padding=1)): w is the ordinary
 *                            w_dgrad pack `[9][w_rows][cin]` of gmk_pack_conv_weight; parities meet 1 / 2 / 2 / 4 of its taps (the same products
 *                            in the same order as gmk_conv_igemm(GMK_CONV_TRANSPOSED2) on the zero-stuffed gradient, minus its 27 of 36 multiplications by zero)
 * bias optional; residual (NHWC, the output's shape) optional with GMK_SUBPIXEL_TRANSPOSED only.  16-bit types, cin = cout = 128, shapes where gmk_conv_subpixel_ok(...) returns 1; otherwise
 * the call fails (callers fall back to gmk_conv_igemm).  GMK_SUBPIXEL=0 in the environment makes gmk_conv_subpixel_ok answer 0 (A/B switch).
 *   GMK_SUBPIXEL_UPSAMPLE_DGRAD  the data gradient of `Upsample` - HIGH -> LOW: src is the output gradient [B][2H][2W][cin], out [B][H][W][cout] =
 *                            sumpool2x2(dgrad3x3(src)) of the reference's autograd in ONE launch: the transpose of the sub-pixel forward, 16 tap-products
 *                            per low-resolution pixel over the four parity views of src; w = the w_sub_dgrad pack of gmk_pack_upsample_weight
 *                            (`[16][w_rows = Cin of the convolution][Cout]`); (H, W) is the LOW-resolution grid in every mode
 * gmk_pack_upsample_weight: w fp32 [Cout][Cin][3][3] -> w_sub (forward, `dtype`) and / or w_sub_dgrad (`dgrad_dtype`), either may be NULL. */
#define GMK_SUBPIXEL_UPSAMPLE 0
#define GMK_SUBPIXEL_TRANSPOSED 1
#define GMK_SUBPIXEL_UPSAMPLE_DGRAD 2
int gmk_conv_subpixel_ok(int B, int H, int W, int cin, int cout, int dtype);
int gmk_pack_upsample_weight(const float* w, void* w_sub, void* w_sub_dgrad, int cout, int cin, int dtype, int dgrad_dtype, void* stream);
int gmk_conv_subpixel(const void* src, int B, int H, int W, int cin, const void* w, int w_rows, int n0, int cout, int mode,
                      const float* bias, const void* residual, void* out, int out_cstride, int dtype, void* stream);
/* Weight gradient of `Upsample` (simple_unet.py:112-122) in the sub-pixel form: dy = the HIGH-resolution output gradient [B][2H][2W][dy_cstride]
 * (bf16), x = the saved LOW-resolution input [B][H][W][cin] (bf16, or fp16 re-rounded to bf16 on its way into LDS), dw = the reference's
 * [Cout][Cin][3][3] fp32 (overwritten).  The 16 tap gradients of the pre-summed 2x2-tap matrices are accumulated over the low-resolution slots
 * (16 tap-products per low-resolution pixel; gmk_conv_wgrad(GMK_CONV_UPSAMPLE2) multiplies 36) and folded onto the 9 taps by a deterministic
 * two-stage reduce.  cin = cout = 128, W <= 62, where gmk_conv_wgrad_subpixel_ok(...) returns 1 (GMK_SUBPIXEL=0 / GMK_WGRAD_KERNEL != 0: never);
 * workspace: gmk_conv_wgrad_subpixel_workspace_bytes(...) bytes. */
int gmk_conv_wgrad_subpixel_ok(int B, int H, int W, int cin, int cout);
int64_t gmk_conv_wgrad_subpixel_workspace_bytes(int B, int H, int W, int cin, int cout);
int gmk_conv_wgrad_subpixel(const void* dy, int dy_cstride, const void* x, int B, int H, int W, int cin, int cout, float* dw,
                            void* workspace, int64_t workspace_bytes, int dtype, int x_dtype, void* stream);
/* statistics-only GroupNorm for the above: mean / rstd [B][groups] and the affine tables (columns [0, C) of rows of tab_stride
 * floats: a concatenated input passes the same table with a column offset); xadd as in gmk_gn_silu_fwd */
int gmk_gn_stats(const void* x, const float* gamma, const float* beta, float* mean, float* rstd, float* tab_scale,
                 float* tab_shift, int tab_stride, int B, int HW, int C, int groups, float eps, const float* xadd,
                 int xadd_stride, int dtype, void* stream);
/* weight gradient: dw[n][k][tap] (reference layout `[Cout][Cin][k][k]`, fp32, overwritten or accumulated) =
 *   sum_pixels dy[b][oy][ox][n0_dy + n] * srcK[b][sy][sx][k],  same gather as the forward of `mode`.
 * workspace: gmk_conv_wgrad_workspace_bytes(B*ho*wo, ksize*ksize, cout, c0+c1) bytes.
 * dtype: type of dy; x_dtype: type of the saved activations src0 / src1 - the same, or GMK_F16 next to GMK_BF16 gradients (the
 * activations are re-rounded to bf16 on their way into LDS: bf16 x bf16 products, fp32 accumulation). */
int gmk_conv_wgrad(const void* dy, int dy_cstride, const void* src0, const void* src1, int c0, int c1, int B,
                   int hs, int ws, int ho, int wo, int ksize, int mode, float* dw, int cout, void* workspace,
                   int64_t workspace_bytes, int dtype, int x_dtype, void* stream);

/* ---- stem / head convolutions (degenerate channel counts; simple_unet.py:92-94 and :41) ----------------
 * stem: x NCHW fp32 [B][cin][H][W] (cin <= 4) -> y NHWC [B][H][W][C]; w [C][cin][3][3], bias [C] */
int gmk_stem_fwd(const float* x, const float* w, const float* bias, void* y, int B, int cin, int H, int W, int C,
                 int dtype, void* stream);
/* dw_part: fp32 [nblk][C*cin*9] partials in the reference layout [C][cin][3][3], nblk = gmk_stem_wgrad_blocks(B*H*W);
 * reduce over nblk with gmk_colsum */
int gmk_stem_wgrad_blocks(int64_t n_pixels);
int gmk_stem_wgrad(const float* x, const void* dy, float* dw_part, int B, int cin, int H, int W, int C, int dtype,
                   void* stream);
/* head: a NHWC [B][H][W][C] -> out NCHW fp32 [B][cout][H][W] (cout <= 4); w [cout][C][3][3], bias [cout] */
int gmk_head_fwd(const void* a, const float* w, const float* bias, float* out, int B, int cout, int H, int W, int C,
                 int dtype, void* stream);
int gmk_head_dgrad(const float* dout, const float* w, void* da, int B, int cout, int H, int W, int C, int dtype,
                   void* stream);
/* dw_part: fp32 [nblk][cout*C*9 + cout]: weight partials in the reference layout [cout][C][3][3], then cout bias
 * partials; nblk = gmk_head_wgrad_blocks(B*H*W); reduce over nblk with gmk_colsum */
int gmk_head_wgrad_blocks(int64_t n_pixels);
int gmk_head_wgrad(const float* dout, const void* a, float* dw_part, int B, int cout, int H, int W, int C, int dtype,
                   void* stream);

/* ---- embedding path (simple_unet.py:20-34,45-64,166,205-224), fp32 -------------------------------------- */
/* out[b][0:32] = cos(t[b]*f_k), out[b][32:64] = sin(t[b]*f_k); freqs: fp32 [32] table computed by the host
 * exactly as simple_unet.py:215-219 does */
int gmk_timestep_embedding(const float* t, const float* freqs, float* out, int B, void* stream);
/* onehot[b][0:10]: F.one_hot(guide with -1 -> 0) as fp32 (simple_unet.py:53-56); keep[b] = guide[b] != -1 */
int gmk_guide_onehot(const int64_t* guide, float* onehot, float* keep, int B, void* stream);
/* classifier-free label drop of DiffusionModel.train_step (diffusion_model.py:67 `y[torch.rand(B) < cf_drop_prob] = -1`), in
 * place on the caller's labels like the reference: y[b] = -1 where gmk_rng_uniform(seed, offset)[b] < p */
int gmk_label_drop(int64_t* y, int B, float p, uint64_t seed, uint64_t offset, void* stream);
/* out[0] = mean(x[0..n)), fixed summation order: the batch mean of the per-sample losses (diffusion_model.py:78) */
int gmk_mean(const float* x, int n, float* out, void* stream);
/* C[i][j] = (accumulate ? C[i][j] : 0) + rowscale[i] * (bias[j] + bias2[j] + sum_k fa(A[i*sa0 + k*sa1]) * fb(B[k*sb0 + j*sb1]))
 * fa / fb = SiLU when bit 0 / bit 1 of `silu` is set, else identity; bias / bias2 / rowscale may be NULL (bias2: the conv1
 * bias that rides with the 12 emb_layers of simple_unet.py:166,183; rowscale = the
 * `guide != -1` row mask of simple_unet.py:57).  Small strided fp32 GEMM behind every nn.Linear forward/backward
 * of the embedding path. */
int gmk_gemm_f32(const float* A, int64_t sa0, int64_t sa1, const float* B, int64_t sb0, int64_t sb1, float* C,
                 int64_t ldc, int M, int N, int K, const float* bias, const float* bias2, const float* rowscale,
                 int silu, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/* scratch the GEMM wants for its deterministic split-K (few output tiles, long K); 0 if none */
int64_t gmk_gemm_f32_workspace_bytes(int M, int N, int K);
/* dpre[i] = dpost[i] * SiLU'(pre[i]) * (rowscale ? rowscale[i / ncols] : 1) */
int gmk_silu_bwd(const float* dpost, const float* pre, const float* rowscale, float* dpre, int64_t n, int ncols,
                 void* stream);
/* out[i] = in[i] * rowscale[i / ncols] */
int gmk_scale_rows(const float* in, const float* rowscale, float* out, int64_t n, int ncols, void* stream);

/* ---- diffusion algebra (gaussian_diffusion.py, diffusion_utils.py), fp32, images as flat [B][n] ---------- */
/* counter-based Philox4x32-10 streams: element i of stream (seed, offset) is reproducible anywhere */
int gmk_rng_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
int gmk_rng_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* logsnr[b] = -2 log(tan(a u[b] + b)) (diffusion_utils.py:198-201); z = x sqrt(sigmoid(l)) + eps sqrt(sigmoid(-l))
 * (gaussian_diffusion.py:95-100, diffusion_utils.py:65-73) */
int gmk_q_sample(const float* x, const float* eps, const float* u, float* logsnr, float* z, int B, int64_t n,
                 void* stream);
/* gaussian_diffusion.py:58-77,165-169 and their backward in one pass over each sample:
 * x_hat = clip(x_from_output(v)); eps_hat = eps_from_x; loss_b = max(mean (x_hat-x)^2, mean (eps_hat-eps)^2);
 * dv (optional) = d(grad_scale * sum_b loss_b)/dv.  loss_b/x_mse/eps_mse: fp32 [B].
 * mean_type (the reference's `mean_type`, :58-73) says what the network output v is: 0 'v' (x = alpha z - sigma v),
 * 1 'eps' (x = (z - sigma v)/alpha), 2 'x' (x = v).  ('both' splits a 1-channel output along W in the reference — dead.) */
int gmk_v_loss(const float* v, const float* z, const float* x, const float* eps, const float* logsnr, float* loss_b,
               float* x_mse, float* eps_mse, float* dv, float grad_scale, int loss_type, int mean_type, int B, int64_t n,
               void* stream);
/* loss_type 0 = 'snr_trunc' (max of the two MSEs, :168-169), 1 = 'snr' (eps MSE only, :170-171, distillation step1);
 * x / eps are the denoising targets (x0, eps) or the teacher's (x_target, eps_target). */
/* one reverse step on a batch (gaussian_diffusion.py:189-243,174-187,292):
 *   v: conditional net output; v_uncond/cond_w: NULL or the unconditional output + per-sample guidance weight;
 *   noise: NULL -> DDIM update, else ancestral ('noisy') update with that noise; is_last: the i == 0 select.
 *   z_next is written; x_pred / eps_pred are optional outputs.
 *   z_dup (optional): a second copy of z_next - the other half of the 2B-image batch the guided sampler feeds the network (:176-177
 *   evaluates the net twice on the same z); logsnr_next (optional, B floats, 2B with z_dup): filled with logsnr_s, which is the next
 *   iteration's logsnr_t (u_t(i-1) = u_s(i), :288-290) - the loop then needs neither torch.cat nor torch.full per step. */
int gmk_sampler_step(const float* v, const float* v_uncond, const float* cond_w, const float* z, const float* noise,
                     float logsnr_t, float logsnr_s, int is_last, float* z_next, float* x_pred, float* eps_pred,
                     float* z_dup, float* logsnr_next, int mean_type, int B, int64_t n, void* stream);

/* ---- self-attention core (north_star "optional self-attention block", BASELINE config 5) -------------
 * The reference SimpleUnet has no attention block: these have NO reference call site (parity unpinned; their definition is the
 * CPU restatement `attention_block` the tests check them against).  QK^T / PV and their gradients are one batched GEMM on the matrix cores:
 *   C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][n][k]      (both operands K-contiguous; leading dimensions and batch
 *   strides in elements; bf16 operands: K and all strides multiples of 8; out_dtype GMK_F32 or GMK_BF16) */
int gmk_bgemm_nt(const void* A, int64_t a_batch, int64_t lda, const void* B, int64_t b_batch, int64_t ldb, void* C,
                 int64_t c_batch, int64_t ldc, int batch, int M, int N, int K, float alpha, int in_dtype, int out_dtype,
                 void* stream);
/* out[b][c][r] = in[b][r][c]  (R x Cc matrices, leading dimensions / batch strides in elements) */
int gmk_transpose(const void* in, int64_t in_batch, int64_t ld_in, void* out, int64_t out_batch, int64_t ld_out, int batch,
                  int R, int Cc, int dtype, void* stream);
/* P[r][:] = softmax(scale * S[r][:]) over rows of N <= 1024 fp32 scores; P in out_dtype */
int gmk_softmax_fwd(const float* S, void* P, int64_t rows, int N, float scale, int out_dtype, void* stream);
/* dS = scale * P * (dP - sum_j dP_j P_j)  (P, dS in dtype; dP fp32) */
/* fused forward of the block's core: o[b] = softmax(scale * q[b] k[b]^T) v[b] with q, k, v the three C-column blocks of qkv
 * [B][N][3C] (bf16), one workgroup per sample, K / V resident in LDS, no N x N matrix in HBM unless p_out (bf16 [B][N][N], the
 * probabilities the backward pass needs) is given.  fp8 != 0: both contractions on the fp8 (OCP e4m3) matrix cores with fp32
 * accumulation - BASELINE config 5's "fp8 MFMA".  C = 128, N in {64, 128, 256}. */
int gmk_attention_fwd(const void* qkv, void* o, void* p_out, int B, int N, int C, float scale, int fp8, void* stream);
/* fused backward of the same core (round 5): dqkv [B][N][3C] (bf16) = (dq | dk | dv) given d_o [B][N][C] (bf16), the forward's inputs qkv and output o.
 * P is recomputed block by block in registers: no N x N matrix in HBM (the three-kernel path kept P [B][N][N] and a fp32 dP of that shape).  stats:
 * fp32 scratch [B][N][2] (log2-domain LSE and sum_c dO O per query, written by the first of the two kernels).  bf16 arithmetic with fp32 accumulation
 * also behind the fp8 forward (straight-through for the e4m3 rounding); deterministic.  C = 128, N in {64, 128, 256}. */
int gmk_attention_bwd(const void* qkv, const void* o, const void* d_o, void* dqkv, float* stats, int B, int N, int C, float scale, void* stream);
int gmk_softmax_bwd(const void* P, const float* dP, void* dS, int64_t rows, int N, float scale, int dtype, void* stream);

/* ---- progressive distillation (gaussian_diffusion.py:87-91,105-154) ------------------------------------------ */
/* logsnr[b] = schedule(u[b] - shift) with u given, or u = fp32(i_times[b] + 1) / num_steps (discrete time, :90-91);
 * u_out (optional) receives the shifted u */
int gmk_logsnr_schedule(const float* u, const int64_t* i_times, int num_steps, float shift, float* u_out, float* logsnr,
                        int B, void* stream);
/* one DDIM step with per-sample times logsnr_t[b] -> logsnr_s[b] (teacher steps inside the loss, :116,:129-144) */
int gmk_ddim_step_vec(const float* v, const float* v_uncond, const float* cond_w, const float* z, const float* logsnr_t,
                      const float* logsnr_s, float* z_next, float* x_pred, float* eps_pred, int mean_type, int B, int64_t n,
                      void* stream);
/* x_target = (z_teacher - f z_t) / (alpha_s - f alpha_t), f = exp((softplus(l) - softplus(l_s)) / 2); the i == 0 rows
 * take x_pred_teacher; eps_target = eps_from_x(z_t, x_target, l)  (:147-154) */
int gmk_distill_target(const float* z_teacher, const float* z_t, const float* x_pred_teacher, const float* logsnr,
                       const float* logsnr_s, const int64_t* i_times, float* x_target, float* eps_target, int B, int64_t n,
                       void* stream);

/* ---- optimiser (torch.optim.Adam defaults as diffusion_model.py:56 uses it), flat fp32 arena ------------- */
int gmk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GMK_H */
