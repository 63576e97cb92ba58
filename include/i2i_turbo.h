/*
 * i2i_turbo.h -- C ABI of the MI355X (gfx950) kernels behind Pix2Pix_Turbo / CycleGAN_Turbo .forward().
 *
 * The reference (GaParmar/img2img-turbo) has no FFI/plugin interface: its hot path
 * (src/pix2pix_turbo.py:186-219, src/cyclegan_turbo.py:199-207, src/model.py:14-54) reaches its
 * kernels through torch.nn.functional inside diffusers/peft.  Each entry point below names the
 * library call(s) it replaces on that path.  All functions are `extern "C"`, take plain pointers and
 * sizes (no torch types), never allocate device memory, never synchronise, and launch asynchronously
 * on the caller's hipStream_t (graph-capturable).  Return value: 0 = ok, negative = error; the message
 * is available from i2i_last_error().
 *
 * Activations are NHWC ("tokens" [B, H*W, C] are the same memory), element type = `dtype`.
 * Channel counts of every activation tensor are multiples of 8 (inputs are padded once at the boundary).
 * Weights are packed [Cout][KH][KW][Cin] (k contiguous) in `dtype`; biases and norm affine parameters
 * are fp32.
 */
#ifndef I2I_TURBO_H
#define I2I_TURBO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2I_ABI_VERSION 10

typedef enum { I2I_F32 = 0, I2I_BF16 = 1, I2I_F16 = 2,
               I2I_U8 = 3   /* only as src_dtype / dst_dtype of the boundary layout ops: uint8 images, HWC interleaved */
} i2i_dtype;

typedef enum {
    I2I_OK = 0,
    I2I_ERR_BAD_ARG = -1,      /* unsupported shape / alignment / null pointer */
    I2I_ERR_LAUNCH = -2,       /* HIP reported an error at launch */
    I2I_ERR_UNSUPPORTED = -3,
    I2I_ERR_RUNTIME = -4       /* other HIP runtime failure (graph capture, events) */
} i2i_status;

typedef enum {
    I2I_OP_IGEMM = 1,
    I2I_OP_GN_STATS = 2,
    I2I_OP_LAYERNORM = 3,
    I2I_OP_SOFTMAX = 4,
    I2I_OP_NCHW_TO_NHWC = 5,
    I2I_OP_NHWC_TO_NCHW = 6,
    I2I_OP_POSTERIOR = 7,
    I2I_OP_DDPM_POSTQUANT = 8,
    I2I_OP_ATTENTION = 9,
    I2I_OP_GN_APPLY = 10,
    I2I_OP_EMBED = 11,
    I2I_OP_LORA_MERGE = 12,
    I2I_OP_RESIZE_U8 = 13,
    I2I_OP_NOP = 14            /* one empty kernel launch (bench.py's calibration: microseconds per hipGraph node) */
} i2i_opcode;

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear / batched matmul on MFMA.
 *   C[m][n] = epilogue( alpha * sum_k A[m][k] * B[n][k] )
 * Replaces F.conv2d (3x3 s1/s2, 1x1), F.linear and the q.k^T / p.v matmuls of
 * F.scaled_dot_product_attention, with the surrounding F.group_norm+F.silu (as an A-operand
 * prologue), F.interpolate(nearest 2x) and torch.cat (as A-operand gathers), F.pad(0,1,0,1), bias,
 * residual add, skip add (src/model.py:41-43) and GEGLU fused in.  LoRA (peft) is merged into B at
 * load (W' = W + s.B.A), so one launch covers base + adapter.
 *
 * A operand: virtual concat of up to two NHWC sources [nimg][hin][win][c0 | c1]; m -> (img, oy, ox),
 *            k -> (ky, kx, ci); input coord = up_map(o*stride + k - pad) (>> ups, or the explicit up_h/up_w map), zero outside.
 *            z (grid z) adds zb*a_bs_b + zh*a_bs_h with zb = z / zh_count, zh = z % zh_count.
 * B operand: [N][ldb] k-contiguous (weights, or K of q.k^T, or V^T of p.v).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const void* a0; const void* a1;
    int32_t c0, c1;            /* channels per source (c1 = 0: single source); multiples of 8 */
    int32_t lda0, lda1;        /* pixel stride in elements (>= c0 / c1, multiple of 8) */
    int64_t a_bs_b, a_bs_h;    /* batch strides (elements) applied to BOTH sources */
    int32_t nimg, hin, win;    /* source geometry (before the optional 2x upsample) */
    int32_t ho, wo;            /* output geometry; M = nimg*ho*wo */
    int32_t ks, stride, pad, ups;
    const void* b; int32_t ldb; int64_t b_bs_b, b_bs_h;
    int32_t M, N, K;           /* K = ks*ks*(c0+c1) */
    const float* gn_ss;        /* optional [nimg][c0+c1][2] = (scale, shift) from I2I_OP_GN_STATS */
    int32_t act;               /* 0 none, 1 SiLU (after the affine) */
    const float* bias; int32_t bias_mode;   /* 0 none, 1 per column n, 2 per row m */
    float alpha;
    const void* res; int32_t ldr; int64_t r_bs_b, r_bs_h;   /* optional residual, same dtype */
    void* c; int32_t ldc; int64_t c_bs_b, c_bs_h;
    int32_t zcount, zh_count;  /* grid z = zcount batches; zh_count heads per batch (>= 1) */
    int32_t geglu;             /* 1: B rows interleaved [a16|g16]; out[:, n/2] = a * gelu(g); ldc for N/2 */
    int32_t out_f32;           /* 1: store fp32 regardless of dtype (attention scores) */
    int32_t tile;              /* 0 = auto; else forces a kernel / tile config id (tests / tuning):
                                  1-5 register-staged igemm, 10-19 / 30-39 halo conv3x3, 20-26 LDS-DMA igemm (26: 64x32 tiles),
                                  40-49 wide-tile conv3x3 (32x32x16 MFMA, conv3x3_w32.hip; 40 = its own auto; with `subpix` its
                                  sub-pixel upsampler form), 50-56 wide GEMM (32x32x16 MFMA, gemm_w32.hip; 50 = its own auto,
                                  51 256x160, 52 128x160, 53 256x128, 54 128x128 workgroup tiles; 55 / 56 = 52 / 54 with a 2-deep
                                  ring, two workgroups per CU); 60-69 the narrow-input 3x3 conv (8 padded input channels into a multiple
                                  of 128 output channels: the VAE's conv_in; conv_narrow.hip).
                                  `bias` (bias_mode 1) must be 4-byte aligned; the 3x3 conv routes need it 16-byte aligned */
    int32_t splitk;            /* > 1: split the K loop over grid z; needs `ws`; no GEGLU, zcount == 1 */
    void* ws;                  /* fp32 workspace, >= splitk * M * N floats (split-K partial slabs) */
    float* gn_part;            /* optional: GroupNorm partial sums of the OUTPUT tensor (as stored), written by the
                                  epilogue as [nimg][parts][gn_part_groups][2] = (sum, sum of squares) with
                                  parts = i2i_igemm_gn_parts(); finished by I2I_OP_GN_STATS with finalize_only */
    int32_t gn_part_groups;
    int32_t subpix;            /* 1 (with ups = 1, ks = 3, stride 1, pad 1): `b` holds the SUB-PIXEL form of the
                                  upsample+conv, [4 parities (a,b)][N][2*2*cin] with ldb = 4*cin, tap weights
                                  pre-summed per output parity (packer.subpixel_weights); K stays 9*cin */
    int32_t act_out;           /* epilogue activation after alpha/bias, before the residual: 0 none, 1 GELU (erf form,
                                  CLIP ViT-H MLP), 2 quick_gelu x*sigmoid(1.702x) (CLIP ViT-L).  LDS-DMA igemm only */
    int32_t up_h, up_w;        /* ups = 1 only: explicit size of the nearest-upsampled plane (F.interpolate(size=...), the
                                  UNet's forward_upsample_size path for latent sizes that are not multiples of 8);
                                  0,0 = (2*hin, 2*win).  Source index = min(floor(i * in/up), in-1) as ATen computes it */
    /* Optional SECOND contraction accumulated into the same output before the epilogue:
     *     acc += k2_a[m][0..k2_c) . k2_b[n][0..k2_c)^T        (so `alpha` scales it like the first contraction),
     * a 1x1 convolution over another NHWC tensor at OUTPUT resolution ([nimg][ho][wo][k2_lda], same dtype; k2_b is [N][k2_ldb]).
     * It is the decoder's `sample = sample + skip_conv_i(skip * gamma)` (src/model.py:41-43) folded into the Upsample2D conv
     * that produces `sample` (no read-modify-write pass over the stream), and a ResnetBlock2D's conv_shortcut(input) folded
     * into its conv2.  k2_c a multiple of 64; taken by the wide-tile conv only (i2i_igemm_route() == "conv3x3_w32_kernel" or
     * "conv3x3_w32_kernel<SUBPIX>"), anything else returns I2I_ERR_UNSUPPORTED. */
    const void* k2_a; const void* k2_b;
    int32_t k2_c, k2_lda, k2_ldb;
    /* LayerNorm folded into this GEMM (ABI v9; the wide GEMM only, ks = 1, one source, no split-K): `a0` holds the UN-normalised rows x and
     * the launch computes  C = LN(x) . W^T + b  (F.layer_norm over the K channels, eps = ln_eps, followed by F.linear: the norm1 -> to_q / to_k /
     * to_v, norm2 -> to_q and norm3 -> ff.net.0.proj pairs of diffusers' BasicTransformerBlock under src/pix2pix_turbo.py:199) as
     *     C[m][n] = rstd_m * (x_m . W'_n) - rstd_m * mu_m * ln_cs[n] + bias[n],
     * with W'[n][k] = W[n][k] * gamma[k] in `b` (folded by i2i_lora_merge: kscale), ln_cs[n] = sum_k W'[n][k] over the STORED (rounded)
     * weights and bias[n] = b[n] + sum_k W[n][k] * beta[k] (both written by i2i_lora_merge).  mu_m / rstd_m are accumulated from the row
     * fragments in the K loop (no extra pass over x, no normalised copy of x in HBM).  NULL = off. */
    const float* ln_cs; float ln_eps;
    /* Transposed column range (with ln_cs only): output columns n >= n_trans are written TRANSPOSED to c2[(n - n_trans) * ldc2 + m]
     * (the self-attention V^T [C][B*T] the flash kernel reads), columns below it to `c` as usual: to_q | to_k | to_v in ONE launch.
     * n_trans a multiple of the tile width (160 / 128), M a multiple of 8; 0 = off. */
    int32_t n_trans; void* c2; int32_t ldc2;
} i2i_igemm_params;

/* GroupNorm statistics -> per (image, channel) (scale, shift) so that GN(x)[c] = x*scale + shift.
 * Replaces the reduction half of F.group_norm; the affine+SiLU half runs in the consumer's prologue.
 * Two-stage and atomics-free (deterministic).  One-pass sums (sum, sum of squares) in fp32 with a second, shifted pass over
 * exactly the groups where |mean| > 16 sigma (F.group_norm is two-pass; DESIGN.md "Numerics").  `partial` needs nimg*nparts*groups*2 floats. */
typedef struct {
    const void* x0; const void* x1; int32_t c0, c1, ld0, ld1;
    int32_t nimg, hw, groups; float eps;
    const float* gamma; const float* beta;    /* [c0+c1] fp32 */
    float* partial; int32_t nparts;
    float* ss;                                /* out [nimg][c0+c1][2] */
    int32_t finalize_only;                    /* 1: `partial` was produced by a conv epilogue (gn_part).  x0 (+ ld0) may still be
                                                 given: groups whose one-pass variance is below mean^2 / 256 (cancellation) are
                                                 then re-read against the first-pass mean; with x0 = NULL the one-pass numbers stand */
    int32_t* counters;                        /* optional (ABI v7), >= nimg*groups ints, ZERO before the first launch (the kernel leaves
                                                 them zero): lets the single-launch path for small tensors cut every image's pixels
                                                 into up to `nparts` slices, one workgroup each; the workgroup that arrives last at
                                                 its (image, group set) ticket sums the slices in a fixed order and writes `ss`.
                                                 CONCURRENCY: `counters` / `partial` are state of the launch in flight -- an op (and therefore
                                                 a program, a captured graph or a loaded plan file that contains it) must not run on two
                                                 streams at once; consecutive launches on one stream are fine (the kernel leaves the
                                                 counters at zero).  The same holds for every scratch slab an op names (gn_part, ws). */
} i2i_gn_stats_params;

/* Standalone GN apply (+SiLU): y = act(x*scale+shift).  Used where the consumer cannot apply it in its
 * operand staging (LDS-DMA igemm on the small UNet planes, unfused VAE attention), and when fusion is disabled. */
typedef struct {
    const void* x; void* y; const float* ss; int32_t nimg, hw, c, act;
    int32_t ldx, ldy;          /* pixel strides in elements (0 = c): y may be a channel slice of a concat buffer */
    int32_t ss_ld, ss_off;     /* ss is [nimg][ss_ld][2]; this tensor's channels start at ss_off (0,0 = c,0) */
    const void* x1;            /* ABI v10: optional SECOND source [nimg*hw][ldx1] of c1 channels (torch.cat([x, x1], 1) in front of the norm:
                                  the resnets of the UNet's up blocks): its channels follow x's in y and in ss -- one launch normalises the
                                  concatenated input (it was one launch per source) */
    int32_t c1, ldx1;          /* c1 multiple of 8; ldx1 = 0 means c1 */
} i2i_gn_apply_params;

/* LayerNorm over the last dim (F.layer_norm, eps 1e-5, affine). rows x c, c multiple of 8. */
typedef struct {
    const void* x; void* y; const float* gamma; const float* beta; int32_t rows, c, ldx, ldy; float eps;
} i2i_layernorm_params;

/* Row softmax of fp32 scores: p = softmax(scale * s) written as `dtype` with zero padding up to ldp.
 * The softmax step of F.scaled_dot_product_attention on the unfused path. */
typedef struct {
    const float* s; void* p; int64_t rows; int32_t cols, lds, ldp; float scale;
} i2i_softmax_params;

/* Fused flash-style attention (F.scaled_dot_product_attention, no mask, not causal).
 * q [b][tq][ldq] / k [b][tk][ldk] with head h at column h*d; vt is V^T: [b][heads*d][ldvt] (tk contiguous).
 * o = softmax(scale * q k^T) v.  The 16-bit d = 64 kernel works in log2 units: it multiplies q by scale * log2(e) once at load (one
 * more rounding of q to the 16-bit type) UNLESS scale * log2(e) == 1, i.e. the caller passes q already multiplied by the factor and
 * scale = ln 2 -- the same function, and what the model's planner does (the factor is folded into to_q's weights before they are
 * rounded: Packer out_scale / scale0). */
typedef struct {
    const void* q; const void* k; const void* vt; void* o;
    int32_t batch, heads, d, tq, tk, ldq, ldk, ldvt, ldo;
    int64_t q_bs, k_bs, vt_bs, o_bs; float scale;
    int32_t causal;            /* 1: query i attends to keys <= i (CLIP text tower); needs tq == tk */
    int32_t ksplit;            /* > 1 (the 16-bit LDS-DMA kernels, d = 512 and d = 64, no causal mask): the keys are divided among ksplit
                                  workgroups per query tile and a second launch merges the partial results -- for launches with too few
                                  query tiles to fill the chip (batch 1: 32 tiles of 128 queries on 256 CUs; the d = 64 self-attention
                                  over 4096 tokens: 64 serial key tiles per workgroup).  Needs `ws`; 0 / 1 = off */
    void* ws;                  /* fp32 workspace for ksplit: batch*heads*ksplit*tq*(d + 2) floats, 16-byte aligned */
} i2i_attention_params;

/* Token + position embedding of the CLIP text tower (transformers CLIPTextEmbeddings; the tokenizer side stays on the
 * host): y[row][:] = tok[ids[row]][:] + pos[row % T][:], rows = B*T, tables and output in `dtype`. */
typedef struct {
    const int64_t* ids; const void* tok; const void* pos; void* y; int32_t rows, T, c, vocab;
} i2i_embed_params;

/* Boundary layout ops.  NCHW fp32/`dtype` <-> NHWC `dtype` with channel padding (zeros).
 * With src_dtype / dst_dtype = I2I_U8 the outer tensor is a uint8 image batch [n][h][w][c] (HWC, as PIL / numpy hand
 * it over) and the callers' pre/post-processing is folded in:
 *   in : y = u8/255 * mul + add     (F.to_tensor, src/inference_paired.py:50; Normalize([0.5],[0.5]) = mul 2, add -1,
 *                                     src/inference_unpaired.py:47); with binarize_below: the sketch threshold, see the struct
 *   out: u8 = trunc(clamp01(v*mul + add) * 255)   (ToPILImage()(x*0.5+0.5) = mul .5, add .5: src/inference_paired.py:72) */
typedef struct {
    const void* x; void* y; int32_t n, c, h, w, cpad; int32_t src_dtype; float mul, add;
    int32_t binarize_below;    /* I2I_U8 sources only, 0 = off: v = (u8 < binarize_below) ? 1 : 0 instead of u8/255, then v*mul + add.
                                  128 = the sketch script's `F.to_tensor(image) < 0.5` (src/inference_paired.py:57-58: 127/255 < 0.5 <= 128/255) */
} i2i_nchw_to_nhwc_params;
typedef struct {
    const void* x; void* y; int32_t n, c, h, w, ldx; int32_t dst_dtype; int32_t clamp; /* clamp to [-1,1] */
    float mul, add;            /* applied after the clamp; 0,0 means 1,0 */
} i2i_nhwc_to_nchw_params;

/* DiagonalGaussianDistribution.sample() * scaling_factor (+ stochastic mix, src/pix2pix_turbo.py:210):
 * z = (mean + exp(.5*clamp(logvar,-30,20)) * eps) * sf ; u = z*r + noise*(1-r)  (r = 1 => u = z).
 * moments NHWC [n][hw][ldm] (mean = ch 0..lat-1, logvar = lat..2lat-1); eps/noise NCHW fp32 [n][lat][hw]
 * (noise may have n = 1 and is broadcast); u NHWC padded to ldu channels (zeros). */
typedef struct {
    const void* moments; const float* eps; const float* noise; void* u;
    int32_t n, hw, lat, ldm, ldu, noise_n; float sf, r;
    const float* r_dev; /* optional device scalar overriding `r` at run time (the program / hipGraph stays valid when the
                           caller changes r: gradio_sketch2image.py sweeps it per request) */
    float* u_f32;      /* optional fp32 copy [n][hw][lat] for the scheduler step (keeps latent math fp32) */
    int32_t moments_f32; /* 1: moments are fp32 (producer igemm stored with out_f32) */
} i2i_posterior_params;

/* DDPMScheduler.step at the single timestep + /scaling_factor + post_quant_conv (1x1, lat x lat):
 * x0 = (u - sqrt(1-abar)*e)/sqrt(abar) ; y = Wpq.(x0/sf) + bpq.  All fp32 in registers. */
typedef struct {
    const void* u; const void* e; void* y; const float* wpq; const float* bpq;
    int32_t n, hw, lat, ldu, lde, ldy; float sqrt_abar, sqrt_1m_abar, sf;
    int32_t u_f32, e_f32;  /* 1: that operand is fp32 (ldu/lde in fp32 elements) */
} i2i_ddpm_params;

/* Device-side LoRA re-merge into a packed weight tensor (peft's merged form; the reference keeps adapters as side branches
 * and rescales them per call: unet.set_adapters(["default"], weights=[r]) / set_weights_and_activate_adapters(vae, ..., [r])
 * src/pix2pix_turbo.py:206-207, decoder.gamma = r :217):
 *   dst[n][k] = cvt( (w0[n][k] + r * sum_j b[n][j] * a[j][k]) * (use_gamma ? gamma : 1) ),  (r, gamma) = rg[0..1] on the device.
 * w0 fp32 master in the packed layout, a fp32 [rank][K] with lora_alpha/rank folded in (adapters concatenated along rank),
 * b fp32 [N][rank].  K multiple of 4.  rg = NULL means r = gamma = 1. */
typedef struct {
    void* dst; const float* w0; const float* a; const float* b;
    int32_t N, K, rank, use_gamma;
    const float* rg;
    /* LayerNorm fold (ABI v9, all NULL = off): the consumer GEMM computes LN(x) . W^T from the un-normalised rows (i2i_igemm_params.ln_cs):
     *   dst[n][k]   = cvt( w'[n][k] * kscale[k] ),  w' = (w0 + r b.a) * g as above, kscale = the LayerNorm weight
     *   colsum[n]   = sum_k float(dst[n][k])                 (of the ROUNDED values: a constant row then cancels exactly)
     *   bias_out[n] = (bias0 ? bias0[n] : 0) + sum_k w'[n][k] * kshift[k],   kshift = the LayerNorm bias
     * summed per row in a fixed order (deterministic, r -> r' -> r reproduces the bits). */
    const float* kscale; const float* kshift; const float* bias0;
    float* colsum; float* bias_out;
} i2i_lora_merge_params;

/* One separable pass of Pillow's LANCZOS resampling on uint8 HWC image batches [n][hin][win][c] (c <= 4), bit-identical to
 * Image.resize(size, Image.LANCZOS) when the horizontal pass (axis = 1: win -> nout columns) is followed by the vertical pass
 * (axis = 0: hin -> nout rows): the reference's host-side resizes (src/inference_paired.py:38-41, src/inference_unpaired.py:40,53).
 * bounds[o] = (first tap, tap count), coeffs[o][ksize] = 22-bit fixed-point weights, both device int32 arrays computed on the
 * host (image_ops.py restates Pillow's precompute_coeffs / normalize_coeffs_8bpc); out = clip8((2^21 + sum in*k) >> 22). */
typedef struct {
    const void* src; void* dst; int32_t n, hin, win, c;
    int32_t axis;              /* 1 = horizontal pass (resample x), 0 = vertical pass (resample y) */
    int32_t nout, ksize;       /* output extent along `axis`; weights per output coordinate */
    const int32_t* bounds; const int32_t* coeffs;
} i2i_resize_u8_params;

typedef struct { int32_t unused; } i2i_nop_params;

typedef struct {
    int32_t opcode;    /* i2i_opcode */
    int32_t dtype;     /* i2i_dtype */
    union {
        i2i_igemm_params igemm;
        i2i_gn_stats_params gn_stats;
        i2i_gn_apply_params gn_apply;
        i2i_layernorm_params layernorm;
        i2i_softmax_params softmax;
        i2i_attention_params attention;
        i2i_nchw_to_nhwc_params to_nhwc;
        i2i_nhwc_to_nchw_params to_nchw;
        i2i_posterior_params posterior;
        i2i_ddpm_params ddpm;
        i2i_embed_params embed;
        i2i_lora_merge_params lora_merge;
        i2i_resize_u8_params resize_u8;
        i2i_nop_params nop;
    } u;
} i2i_op;

/* ---- library ---- */
int i2i_abi_version(void);
const char* i2i_backend(void);            /* "gfx950" (product) */
const char* i2i_last_error(void);
size_t i2i_sizeof_op(void);               /* sanity check for FFI struct layout */

/* ---- single-op entry points (each = one or two kernel launches on `stream`) ---- */
int i2i_igemm(const i2i_igemm_params* p, int dtype, void* stream);
/* Number of partial-sum slots per image this op would write through gn_part (its spatial tile count), or 0 if
 * the kernel that will run it cannot produce GroupNorm partials (planner query; launches nothing). */
int i2i_igemm_gn_parts(const i2i_igemm_params* p, int dtype, int groups);
/* Name of the kernel family i2i_igemm() routes this op to ("conv3x3_w32_kernel", "conv3x3_w32_kernel<SUBPIX>",
 * "conv3x3_halo_kernel", "conv3x3_halo_kernel<SUBPIX>", "gemm_w32_kernel", "conv_narrow_kernel", "igemm_dma_kernel", "igemm_kernel"): reporting only (bench.py groups its per-op
 * timings by it; the planner does not have to mirror the routing rules).  Launches nothing; never NULL. */
const char* i2i_igemm_route(const i2i_igemm_params* p, int dtype);
int i2i_gn_stats(const i2i_gn_stats_params* p, int dtype, void* stream);
int i2i_gn_apply(const i2i_gn_apply_params* p, int dtype, void* stream);
int i2i_layernorm(const i2i_layernorm_params* p, int dtype, void* stream);
int i2i_softmax(const i2i_softmax_params* p, int dtype, void* stream);
int i2i_attention(const i2i_attention_params* p, int dtype, void* stream);
int i2i_nchw_to_nhwc(const i2i_nchw_to_nhwc_params* p, int dtype, void* stream);
int i2i_nhwc_to_nchw(const i2i_nhwc_to_nchw_params* p, int dtype, void* stream);
int i2i_posterior(const i2i_posterior_params* p, int dtype, void* stream);
int i2i_ddpm_postquant(const i2i_ddpm_params* p, int dtype, void* stream);
int i2i_embed(const i2i_embed_params* p, int dtype, void* stream);
int i2i_lora_merge(const i2i_lora_merge_params* p, int dtype, void* stream);
int i2i_resize_u8(const i2i_resize_u8_params* p, int dtype, void* stream);   /* dtype ignored (uint8 data) */

/* ---- calibration micro-kernels (csrc/calib.hip; bench.py's `calib` block: what THIS box delivers on three elementary loads, so that
 * lines measured on different boxes of a pool can be compared).  Not on the forward path. */
int i2i_nop(void* stream);                                                         /* one empty kernel (1 wave) */
/* 1024 waves (one per SIMD) each issue `iters` x 8 independent v_mfma_f32_32x32x16 on operands read from `operands` (>= 64 KiB of dtype
 * data, random for a power-realistic reading); FLOPs = 1024 * iters * 8 * 32768.  `sink`: >= 1024*64 floats (keeps the work alive). */
int i2i_calib_mfma(int dtype, int iters, const void* operands, float* sink, void* stream);
/* dst[i] = src[i] over `bytes` (multiple of 16), 16 bytes per lane, grid-stride: 2 * bytes of HBM traffic */
int i2i_calib_stream(const void* src, void* dst, size_t bytes, void* stream);

/* ---- programs: a forward pass is a flat array of ops executed in order on one stream ---- */
int i2i_run(const i2i_op* ops, int n_ops, void* stream);
/* Same, with a hipEvent pair around every op; ms[i] = duration of op i (synchronises at the end). */
int i2i_run_timed(const i2i_op* ops, int n_ops, void* stream, float* ms);
/* Capture the program into a hipGraph (launch-bound bs=1 path) and replay it. */
int i2i_graph_create(const i2i_op* ops, int n_ops, void** graph_out);
int i2i_graph_launch(void* graph, void* stream);
int i2i_graph_destroy(void* graph);

/* ---- plan files (ABI v8): a planned forward saved by the Python planner (img2img_turbo_amd/plan_file.py: the op program, every buffer
 * its pointers refer to -- packed weights with their contents, activations as zero-filled scratch --, named input / output buffers and a
 * relocation table), loaded and run without Python.  The whole-forward entry for C / C++ hosts: what `model(x, caption_enc=..., eps=...)`
 * (src/pix2pix_turbo.py:186-219) is for a Python host, for one fixed (batch, size, dtype, mode).  Names of the pix2pix / CycleGAN plans:
 * "x" (fp32 NCHW images in [-1, 1], or uint8 NHWC with the u8 boundary), "ctx" ([1 | B][77][1024] text states in the plan's dtype),
 * "eps" (fp32 [B][4][H/8][W/8] posterior noise), "noise" (stochastic plans), "out" (images, NCHW in the plan's output dtype or uint8 NHWC).
 * load / write / read are synchronous; run only enqueues (i2i_run); i2i_plan_ops() hands the program to i2i_graph_create(). */
int i2i_plan_load(const char* path, void** plan_out);
int i2i_plan_io(void* plan, const char* name, void** dev_ptr, size_t* bytes);       /* device address + size of a named buffer */
int i2i_plan_write(void* plan, const char* name, const void* host_src, size_t bytes);
int i2i_plan_read(void* plan, const char* name, void* host_dst, size_t bytes);     /* synchronises the device first */
int i2i_plan_ops(void* plan, const i2i_op** ops, int* n_ops);
int i2i_plan_run(void* plan, void* stream);
int i2i_plan_destroy(void* plan);

#ifdef __cplusplus
}
#endif
#endif /* I2I_TURBO_H */
