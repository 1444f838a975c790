/*
 * kdiff_hip.h -- C ABI of libkdiff_hip.so: the MI355X (gfx950) kernels behind the k-diffusion
 * sampling hot path.
 *
 * The reference (crowsonkb/k-diffusion) is pure Python and has no FFI of its own; what this ABI
 * replaces are the reference's *native call sites* -- the third-party CUDA extensions and the
 * torch.compile'd functions it calls on the hot path.  Each entry point cites the reference
 * interface it stands in for (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer (device memory);
 *   - nothing here allocates, frees or synchronises; every launch goes to `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream);
 *   - return value: 0 on success, negative KD_E* on a rejected call (bad shape / alignment /
 *     unsupported size); kd_last_error() then describes it.  Nothing throws;
 *   - activations are token-major ("NHWC"): [batch, h, w, channels] row-major fp32;
 *     qkv buffers are [batch*h*w, 3, n_heads, 64] (feature index t*(nh*64) + head*64 + e,
 *     k_diffusion/models/image_transformer_v2.py:386,422,431,467); the head dim is 64 in EVERY arithmetic mode (fp32-parity and bf16
 *     alike: every shipped config, k_diffusion/config.py:135-136).  The reference's `d_head` (image_transformer_v2.py:355-363:
 *     n_heads = d_model // d_head, RoPE on d_head // 2) is therefore not an argument of any entry point -- n_heads is, and the feature
 *     width is n_heads * 64; the Python mirror's model constructor refuses other values with a ValueError that says so
 *     (k-diffusion_amd/models/image_transformer_v2.py) instead of handing the kernels a layout they would misread;
 *   - reproducibility: a call's result is a function of its arguments and of which kernel serves it; the choice between a throughput
 *     kernel and a few-rows / block form of the same projection follows the ROW COUNT and the device's CU count (cost rules of the
 *     launchers), and the forms sum in different orders.  So the same sample can differ in its last bits between batch sizes or chips with
 *     another CU count (inside every tolerance stated here; forms documented as bit-identical to each other are exactly that).  The options
 *     that pin a form (x3s_max_rows / b16s_max_rows = 0, attn_block_bf16 / proj_block_bf16 = 0) give batch-invariant results.
 */
#ifndef KDIFF_HIP_H
#define KDIFF_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define KD_OK 0
#define KD_EINVAL (-1)   /* bad argument / unsupported shape */
#define KD_ELAUNCH (-2)  /* HIP launch error */

int kd_version(void);
const char* kd_last_error(void);
/* Tuning / A-B switches of the library (benchmarks and tests; defaults are the shipped configuration).  The library reads no
 * environment variables: the Python package maps its documented KDIFF_* variables onto these calls.  Thread-safe.
 *   fp32 kernels : "skinny" (1) "astat" (1) "ksplit" (1) "astat_max_k" (512) "astat_waves" (4) "astat_storewait" (0)
 *                  "gemm_debug" (0; profiling ablations of benchmarks/: 1 no C stores, 2 no MFMA, 8 GEGLU without erf)
 *                  "x3" (1; 0 = the round-1 / round-2 kernels for KD_PREC_SPLIT3 projections) "x3_splits" (0 = cost model)
 *                  "x3_half" (1; 0 = one workgroup per CU for the K = 256 projections) "x3_res" (0; 1 = the A-stationary kernel for the K = 512
 *                  residual projection) "x3r" (1; gemm_x3r.hip for projections without a norm in front, K >= 256; 0 = round-1
 *                  kernel, 2 = every eligible shape) "x3r_lw" (1; 0 = staging requests inside the compute waves' K loop instead of loader waves)
 *                  "x3r_split" (1; 0 = TokenSplit + lerp on the round-1 tile kernel) "ffn_x3" (1; 0 = kd_ffn_f32_supported answers no) "ffn_x3_half" (1; 0 = one workgroup per CU
 *                  at K = 128) "attn_x3" (1; 0 = the round-1 attention cores also for split-stored operands)
 *                  "x3_min_rows" (512) "x3r_min_rows" (128) "ffn_x3_min_panels_256" (7/8 of the CUs): row counts from which the throughput
 *                  kernels are chosen / advised
 *                  "x3s_max_rows" (4096; 0 = off) the few-rows latency form of the KD_PREC_SPLIT3 projections (gemm_x3s.hip: 32 rows x one
 *                  half tile per workgroup, K split over 8 waves), taken up to that many rows where its cost estimate beats the
 *                  throughput kernel's; "x3s_max_wgs" (-1; >= 0 replaces the estimate by a cap on the grid) "x3s_scale_lds" (0; 1 = the
 *                  AdaRMSNorm scale vector through LDS where a workgroup's rows share it: same results, measured level)
 *   bf16 kernels : "bf16_fast" (1; 0 = generic kernel only) "wstat" (1) "wstat_waves" (0 = per shape) "wstat_max_slices" (24)
 *                  "wstat_prefetch" (0; 1 next-chunk prefetch, 2 software-pipelined tiles) "astat_bf16" (1) "astat_splits" (0 = auto)
 *                  "tiled_bm" (0 = auto, 128, 256) "tiled_lw" (1; 0 = no loader waves in the tiled kernel at one tile per CU) "attn_global_qw" (8) "patch_fast" (1; 0 = patch-in / patch-out through the generic kernel)
 *                  "ffn_fused" (1; 0 = kd_ffn_bf16_supported answers no) "ffn_fused_256" (0) "ffn_bf16_min_rows" (16384: rows from which
 *                  kd_ffn_bf16_supported advises the fused block)
 *                  "b16s_max_rows" (4096; 0 = off) the few-rows latency form of the bf16 projections (gemm_b16s.hip, the bf16 sibling of
 *                  gemm_x3s.hip), taken up to that many rows where its grid is one round of the chip (two behind a norm at <= 512 rows);
 *                  "b16s_max_wgs" (-1; >= 0: a cap on the grid instead)
 *                  "code_warm" (8: the first wave of that many workgroups of a launch -- one per XCD -- reads the kernel's own code
 *                  range into L2 at entry, so that a kernel that has not run for a while does not walk its code through one
 *                  instruction-cache miss after the other; 0 = off.  Pure prefetch: results do not depend on it) */
/* value == INT_MIN puts the option back to its built-in default. */
int kd_set_option(const char* name, int value);
int kd_get_option(const char* name, int dflt);

/* ------------------------------------------------------------------------------------------
 * Fused GEMM  C = epilogue( prologue(A) @ W^T ).   W is [N_w, K] row-major (nn.Linear.weight).
 * Replaces: nn.Linear / F.linear call sites image_transformer_v2.py:126-129 (Linear),
 * :89-95,132-139 (linear_geglu / LinearGEGLU, torch.compile'd), fused with the ops around
 * them: rms_norm/AdaRMSNorm :98-103,155-166 (prologue), residual add :396,443,476,493,
 * TokenMerge :586-595, TokenSplit + lerp :610-621, TokenSplitWithoutSkip :598-607 and the
 * NCHW<->NHWC movedims :723,760, and the Karras preconditioner k_diffusion/layers.py:88-90.
 */
enum { KD_A_PLAIN = 0, KD_A_MERGE2x2 = 1, KD_A_PATCH_NCHW = 2 };
/* arithmetic of the products: exact fp32 MFMA (bit-for-bit an fmaf chain), or each fp32 operand split into two
 * bf16 (hi + lo, 16 significand bits) with hi*hi + hi*lo + lo*hi on the bf16 MFMA and fp32 accumulation */
enum { KD_PREC_EXACT = 0, KD_PREC_SPLIT3 = 1,
       KD_PREC_BF16 = 2 /* kd_gemm_bf16 only: bf16 activations in HBM, one bf16 MFMA per product, fp32 accumulate -- the
                           arithmetic of the reference under torch.autocast(bfloat16) (image_transformer_v2.py:98-103,376-384) */ };
enum { KD_EPI_STORE = 0, KD_EPI_RESIDUAL = 1, KD_EPI_GEGLU = 2, KD_EPI_SPLIT_LERP = 3, KD_EPI_UNPATCH_NCHW = 4,
       KD_EPI_QKV = 5 /* store a qkv projection with q,k prepared: cosine-sim scaling + axial RoPE
                         (image_transformer_v2.py:106-121,187-231) applied in the epilogue, v untouched */ };

typedef struct {
  int M, N, K;          /* C is [M, N]; K = reduction length (multiple of 4)                      */
  int a_mode, epi;      /* KD_A_*, KD_EPI_*                                                       */
  int norm;             /* 1: A'[m,k] = A[m,k] * scale[b(m)*scale_stride + k] * rsqrt(mean_k A^2 + eps) */
  int rows_per_sample;  /* b(m) = m / rows_per_sample (tokens per image)                          */
  int scale_stride;     /* K for per-sample AdaRMSNorm scales, 0 for a shared RMSNorm gain        */
  int gh, gw;           /* token grid of the COARSE side (merge/split: h,w of the merged grid;
                           patch modes: h,w of the patch grid)                                    */
  int ph, pw, chan;     /* patch modes: patch size and image channels                             */
  float eps;            /* rms-norm epsilon                                                       */
  float out_add;        /* KD_EPI_STORE: constant added to every output (AdaRMSNorm's "+1")       */
  float sigma_data;     /* patch modes: Karras preconditioner                                     */
  const float* A;       /* plain: [M,K]; merge: [B,2gh,2gw,K/4]; patch: image [B,chan,gh*ph,gw*pw] */
  const float* W;       /* [N,K]  (GEGLU: [2N,K], value rows first)                               */
  float* C;             /* store/residual/geglu: [M,N]; split: [B,2gh,2gw,N/4]; unpatch: image     */
  const float* R;       /* residual: [M,N]; split: skip [B,2gh,2gw,N/4]; unpatch: x_in image       */
  const float* scale;   /* norm scales                                                            */
  const float* sigma;   /* [B] per-sample sigma (patch modes; NULL = no preconditioning)           */
  const float* fac;     /* split: device pointer to the lerp factor                               */
  int precision;        /* KD_PREC_*                                                              */
  const void* Wp;       /* KD_PREC_SPLIT3: packed split image of W from kd_pack_weight_bf16x3     */
  int n_heads;          /* KD_EPI_QKV: N == 3 * n_heads * 64; token of row m = m % rows_per_sample   */
  const float* qk_scale;  /* KD_EPI_QKV: [n_heads] cosine-sim scale (self_attn.scale)                 */
  const float* rope_cos;  /* KD_EPI_QKV: [rows_per_sample, n_heads, 16] cos / sin of AxialRoPE theta  */
  const float* rope_sin;
  int qkv_packed;       /* KD_EPI_QKV + KD_PREC_SPLIT3: store every 4-column chunk of q, k, v as [hi: 4 x bf16][lo: 4 x bf16]
                           (hi = bf16(x), lo = bf16(x - hi)) in the 16 bytes of its 4 floats -- the operand format of the
                           split-bf16x3 attention cores, which then take it as stored (their prep = 2)        */
  const float* rope_pos;  /* kd_gemm_bf16 + KD_EPI_QKV: [rows_per_sample, 2] axial position (y, x) of every token       */
  const float* rope_freq; /* kd_gemm_bf16 + KD_EPI_QKV: [n_heads, 8] AxialRoPE freqs / (2 pi) (angles in revolutions); 32-byte aligned */
  /* ---- KD_PREC_SPLIT3 with PRE-SPLIT operands (round 3; kd_gemm_f32 only, norm must be 0, a_mode KD_A_PLAIN) ---------------------
   * a_split: A is not fp32 but two bf16 planes [M, K] (2 bytes per element): `A` = hi = bf16_rne(a), `A_lo` = bf16_rne(a - hi), as
   *          written by kd_norm_split_f32 or by a producer GEMM with c_split.  Both operands then move by LDS-DMA (gemm_x3t.hip).
   * c_split: the result is stored as two such planes ([M, N]: `C` = hi, `C_lo` = lo) instead of fp32 -- the A operand of the next
   *          a_split GEMM (KD_EPI_GEGLU -> down projection).  Not with KD_EPI_QKV (its split form is qkv_packed).                 */
  int a_split, c_split;
  const void* A_lo;
  void* C_lo;
  int per_row;          /* kd_gemm_f32, products with one row per SAMPLE (the conditioning chain): always use the per-row fp32 FMA
                           kernel, also above its 128-row default limit.  A row's result then does not depend on how many rows
                           share the launch, so the conditioning of a whole sigma schedule (steps x batch rows in one launch)
                           is bit-identical with computing it step by step */
} KdGemm;

int kd_gemm_f32(const KdGemm* desc, void* stream);

/* The same fused GEMM in bf16 arithmetic (precision must be KD_PREC_BF16).  Buffers: A, C, R are bf16 row-major (2 bytes per
 * element) EXCEPT the image side of the patch modes, which stays fp32 like the solver state: A of KD_A_PATCH_NCHW, C and R of
 * KD_EPI_UNPATCH_NCHW.  scale / sigma / fac / qk_scale are fp32.  Wp = image from kd_pack_weight_bf16 (W itself is not read).
 * KD_EPI_QKV computes the RoPE angles in the epilogue from rope_pos / rope_freq (hardware sin / cos) instead of reading
 * rope_cos / rope_sin tables.  Products: one v_mfma_f32_32x32x16_bf16 each, fp32 accumulation; RMS statistics, GELU, RoPE,
 * lerp and the Karras scalings in fp32 registers; one rounding to bf16 at the store. */
int kd_gemm_bf16(const KdGemm* desc, void* stream);
long long kd_packed_weight_bytes_bf16(int N, int K, int geglu);
/* W [N or 2N (geglu), K] fp32 -> blocks [n-tile][k-step][128 rows][64 k] bf16 (16 KiB each, the kernels' swizzled LDS image).
 * `geglu` selects the layout, as two bits: bit 0 = GEGLU rows (value / gate rows interleaved per 32 outputs), bit 1 = the k order in
 * which an MFMA result holds a row (inside every group of 16 k: 0-3, 8-11, 4-7, 12-15): 2 = the down projection of kd_ffn_bf16,
 * 3 = its up projection when the out projection is fused in front (KdFfn.attn). */
int kd_pack_weight_bf16(const float* W, void* out, int N, int K, int geglu, void* stream);

/* Fused feed-forward block in bf16 arithmetic:  out = x + down_proj(GEGLU(up_proj(AdaRMSNorm(x)))).
 * Replaces FeedForwardBlock.forward, image_transformer_v2.py:487-493 (norm :155-166, LinearGEGLU :132-139, dropout = identity at
 * inference, Linear :126-129, residual :493) -- the pair kd_gemm_bf16(KD_EPI_GEGLU) + kd_gemm_bf16(KD_EPI_RESIDUAL) without the
 * d_ff-wide hidden activation ever leaving the chip.  x, out: bf16 [M, K] (out may be x); scale: fp32 norm scales, row m uses
 * scale + (m / rows_per_sample) * scale_stride; Wp_up = kd_pack_weight_bf16(up_proj.weight [2 d_ff, K], N = d_ff, geglu = 1);
 * Wp_down = kd_pack_weight_bf16(down_proj.weight [K, d_ff], N = K, K = d_ff, geglu = 2).
 * Shapes: K == 128 or 256, d_ff % 64 == 0; anything else returns KD_EINVAL and the caller uses the pair.  kd_ffn_bf16_supported
 * additionally says where the fused form is the FASTER one (K == 128 and M >= 16384; K == 256 only with option "ffn_fused_256"). */
typedef struct KdFfn {
  const void* x;
  void* out;
  const float* scale;
  int scale_stride, rows_per_sample;
  float eps;
  const void* Wp_up;
  const void* Wp_down;
  int M, K, d_ff;
  /* Both NULL or both set: the attention block's out projection fused in front of the block (image_transformer_v2.py:473-476 then
   * :487-493):  x' = x + attn Wout^T;  out = x' + down(GEGLU(up(norm(x')))).  attn = [M, K] in the activation type of the call (the
   * attention core's output, heads merged), Wp_out = the packed out_proj.weight [K, K] (N = K, K, geglu = 0) and Wp_up must then be packed
   * with geglu = 3 (the k order in which an MFMA result holds a row).  kd_ffn_f32 takes it at K == 128 and K == 256 (fp32 attn,
   * kd_pack_weight_bf16x3 images); kd_ffn_bf16 at K == 128 (bf16 attn, kd_pack_weight_bf16 images; other K: KD_EINVAL). */
  const void* attn;
  const void* Wp_out;
} KdFfn;
int kd_ffn_bf16_supported(int M, int K, int d_ff);
int kd_ffn_bf16(const KdFfn* desc, void* stream);
/* The same block in fp32-parity arithmetic (KD_PREC_SPLIT3: fp32 x / out, three split-bf16 MFMA terms per product, fp32 accumulate):
 * Wp_up = kd_pack_weight_bf16x3(up_proj.weight, N = d_ff, K, geglu = 1), Wp_down = kd_pack_weight_bf16x3(down_proj.weight [K, d_ff],
 * N = K, K = d_ff, geglu = 2).  K == 128 or 256, d_ff % 64 == 0; out may be x.  kd_ffn_f32_supported says where it is the faster form
 * (M >= 2048; option "ffn_x3" = 0 answers no).  At K == 128 the kernel runs two workgroups per CU over half tiles of 32 hidden features
 * (option "ffn_x3_half" = 0: the one-workgroup form; same results). */
int kd_ffn_f32_supported(int M, int K, int d_ff);
int kd_ffn_f32(const KdFfn* desc, void* stream);

/* One-off packing of a weight for KD_PREC_SPLIT3 (weights are static during sampling): W [N or 2N (geglu == 1), K]
 * fp32 -> `out`, kd_packed_weight_bytes(N, K, geglu) bytes: [n-tile][k-step][hi|lo][128 rows][32 bf16] in the
 * kernel's swizzled LDS order, zero-padded.  N is the OUTPUT width (GEGLU: d_ff, W has 2*d_ff rows).
 * `geglu`, two bits as for kd_pack_weight_bf16: 0 plain, 1 GEGLU rows, 2 the k order of kd_ffn_f32's down projection (inside every
 * group of 16 k: 0-3, 8-11, 4-7, 12-15), 3 both (its up projection behind a fused out projection). */
long long kd_packed_weight_bytes(int N, int K, int geglu);
int kd_pack_weight_bf16x3(const float* W, void* out, int N, int K, int geglu, void* stream);

/* AdaRMSNorm / RMSNorm (image_transformer_v2.py:98-103, :155-166) of fp32 rows, written as the two bf16 planes of an a_split GEMM
 * operand: y[m, :] = x[m, :] * (scale[b(m) * scale_stride + :] * rsqrt(mean(x[m, :]^2) + eps)), hi = bf16_rne(y), lo = bf16_rne(y - hi);
 * b(m) = m / rows_per_sample.  scale == NULL: plain split of x (no norm).  K % 8 == 0, K <= 2048.  Used for the widths the fused norm ->
 * projection kernels (kd_gemm_f32 with norm = 1: K = 128 / 256 / 512, in registers) do not take -- 384, 640, ..., 2048 (K % 128 == 0):
 * the planes then feed an a_split kd_gemm_f32.  None of the shipped configs has such a width; tests/test_model_gpu.py covers 384 and 1152. */
int kd_norm_split_f32(const float* x, const float* scale, int scale_stride, int rows_per_sample, void* hi, void* lo, int M, int K,
                      float eps, void* stream);

/* Stand-alone RMS norm over the last dim (mapping network, image_transformer_v2.py:142-152):
 * y[m,:] = x[m,:] * scale[:] * rsqrt(mean(x[m,:]^2) + eps).  d <= 4096, d % 4 == 0. */
int kd_rmsnorm_f32(const float* x, const float* scale, float* y, int rows, int d, float eps, void* stream);

/* Conditioning front end (image_transformer_v2.py:734-740 + layers.py:285-293 FourierFeatures):
 * ff[b, :] = [cos(2*pi*c*w_j), sin(2*pi*c*w_j)], c = log(sigma_b)/4, w = time_emb.weight [half]. */
int kd_fourier_sigma_f32(const float* sigma, const float* weight, float* ff, int batch, int half, void* stream);
/* generic FourierFeatures for aug_cond: ff[b,:] = [cos(f), sin(f)], f = 2*pi * in[b,:] @ w^T, w [half, in_dim] */
int kd_fourier_f32(const float* in, const float* weight, float* ff, int batch, int in_dim, int half, void* stream);
/* out[b,:] = a[b,:] + (b_rows ? b[b,:] : b[:]) + (emb ? emb[ids[b],:] : 0) + (c ? c[b,:] : 0) */
int kd_cond_sum_f32(float* out, const float* a, const float* b, int b_rows, const float* emb,
                    const long long* ids, const float* c, int batch, int d, void* stream);

/* ------------------------------------------------------------------------------------------
 * q/k preparation, in place on a qkv buffer [tokens_total, 3, nh, 64]:
 *   q,k <- rope( x * sqrt(scale_h) * rsqrt(sum x^2 + eps) ),   v untouched.
 * Replaces scale_for_cosine_sim (image_transformer_v2.py:106-114) + apply_rotary_emb_
 * (:187-231, in place on a view).  cos/sin: [tokens_per_sample, nh, 16] tables of
 * AxialRoPE.forward's theta (:245-248), computed once on the host.
 * The attention kernels below take `prep`:
 *   0  q, k already prepared (by this call or by the KD_EPI_QKV epilogue), fp32
 *   1  raw fp32 q, k: prepared on the fly (needs scale_h, cos_t, sin_t)
 *   2  prepared AND stored split (KdGemm.qkv_packed: [hi: 4 x bf16][lo: 4 x bf16] per 4-column chunk): the split-bf16x3
 *      cores use the operands as stored; not available with KD_PREC_EXACT
 * `precision` (KD_PREC_EXACT / KD_PREC_SPLIT3) selects the arithmetic of the two products, as in KdGemm; the neighbourhood
 * core has the split-bf16x3 form only (the argument is accepted for symmetry). */
int kd_qk_prep_f32(float* qkv, const float* scale_h, const float* cos_t, const float* sin_t,
                   int batch, int tokens_per_sample, int nh, float eps, void* stream);

/* Dense softmax attention per (sample, head) over all T tokens (any T; T > 256 streams key blocks with an online
 * softmax and is served by the split-bf16x3 core only), softmax scale 1.0.
 * Replaces F.scaled_dot_product_attention / flash_attn_qkvpacked_func at
 * image_transformer_v2.py:383,392.  out: [batch*T, nh*64]. */
int kd_attn_global_f32(const float* qkv, float* out, int batch, int T, int nh,
                       int prep, const float* scale_h, const float* cos_t, const float* sin_t, float eps,
                       int precision, void* stream);

/* Shifted-window attention (window ws x ws tokens, ws in {4, 8, 16}), roll/window/mask/unwindow
 * folded into addressing.  Replaces apply_window_attention image_transformer_v2.py:319-337
 * (+ :253-316).  shift is 0 or ws/2. */
int kd_attn_window_f32(const float* qkv, float* out, int batch, int H, int W, int nh, int ws, int shift,
                       int prep, const float* scale_h, const float* cos_t, const float* sin_t, float eps,
                       int precision, void* stream);

/* 2-D neighbourhood attention, kernel ks x ks, window clamped inside the image, dilation 1, heads-last.  Replaces
 * natten.functional.na2d(q,k,v,kernel_size,scale=1.0) image_transformer_v2.py:428 (and the unfused pair :437-439; kernel_size is
 * the block's constructor argument, :399-410).  ks: odd, 3 .. 13 with split-stored operands (prep = 2: what the package's fp32-parity
 * mode runs -- 7 is the shipped size, 3 / 5 / 9 the same kernel with other constants, 11 / 13 a densely packed patch form);
 * ks == 7 only for fp32 operands (prep 0 / 1) and with option "attn_x3" = 0.  H, W >= ks. */
int kd_attn_na2d_f32(const float* qkv, float* out, int batch, int H, int W, int nh, int ks,
                     int prep, const float* scale_h, const float* cos_t, const float* sin_t, float eps,
                     int precision, void* stream);

/* The three attention cores in bf16 arithmetic (KD_PREC_BF16): qkv is the bf16 output of kd_gemm_bf16's KD_EPI_QKV epilogue
 * ([tokens, 3, nh, 64], q and k already prepared), out is bf16 [tokens, nh * 64].  bf16 MFMA products, fp32 scores / softmax /
 * accumulators.  Same reference call sites as the fp32 entry points above; kd_attn_na2d_bf16 takes kernel sizes 3 .. 13 (odd;
 * 3 .. 9 in the tuned form, 11 and 13 in a general one). */
int kd_attn_global_bf16(const void* qkv, void* out, int batch, int T, int nh, void* stream);
int kd_attn_window_bf16(const void* qkv, void* out, int batch, int H, int W, int nh, int ws, int shift, void* stream);
int kd_attn_na2d_bf16(const void* qkv, void* out, int batch, int H, int W, int nh, int ks, void* stream);

/* The global-attention block in front of its out projection as ONE launch (round 5; k_diffusion/models/image_transformer_v2.py:370-392:
 * norm -> qkv_proj -> scale_for_cosine_sim_qkv -> apply_rotary_emb_ -> scaled_dot_product_attention / flash_attn_qkvpacked_func).  `d` is the
 * descriptor of the block's qkv projection exactly as kd_gemm_bf16 takes it (A = residual stream, Wp = packed qkv weight, scale /
 * scale_stride / rows_per_sample = the AdaRMSNorm scales, qk_scale, rope_pos, rope_freq, n_heads; epi = KD_EPI_QKV, norm = 1, precision =
 * KD_PREC_BF16), except that C receives the ATTENTION OUTPUT [M, n_heads * 64] bf16: q, k, v never reach HBM.  One workgroup per (sample,
 * head).  Shapes: 256 tokens per sample, K = 64 * n_heads in {256, 512}, N = 3 K, M % 256 == 0 (kd_attn_block_bf16_supported tells);
 * anything else returns KD_EINVAL and the caller issues kd_gemm_bf16 + kd_attn_global_bf16, whose results this entry reproduces bit for bit.
 * (Round 5 also offered the block's out projection in the same launch behind a cross-workgroup rendezvous; it measured level with the separate
 * launch and was removed in round 6 -- a spin-wait between workgroups has no place in a library that must never hang or corrupt silently.) */
int kd_attn_block_bf16_supported(int tokens_per_sample, int width, int n_heads);
int kd_attn_block_bf16(const KdGemm* d, void* stream);

/* AdaRMSNorm -> wide projection in the same form (round 5): the FF block's norm -> up projection + GEGLU (image_transformer_v2.py:487-491) and,
 * for the levels whose attention core is a launch of its own, norm -> qkv projection + cosine-sim scale + RoPE (:370-380, :415-425).  A workgroup
 * per (256-row group, slice of six 64-row half blocks of the packed weight), the group's rows normalised once into register fragments, six passes
 * over K.  `d` is the projection's descriptor as kd_gemm_bf16 takes it (epi = KD_EPI_GEGLU with N = d_ff, or KD_EPI_QKV with N = 3 K; norm = 1,
 * precision = KD_PREC_BF16); the result is bit-identical to kd_gemm_bf16(d).  Shapes: rows per sample a multiple of 256, K in {256, 512},
 * d_ff % 192 == 0 (kd_proj_block_bf16_supported tells); else KD_EINVAL. */
int kd_proj_block_bf16_supported(int tokens_per_sample, int width, int n, int epi);
int kd_proj_block_bf16(const KdGemm* d, void* stream);

/* fp8 arithmetic mode (round 6; BASELINE configs[4] "fp8 MFMA weights" -- no reference counterpart: convert_for_inference.py:23 stops at
 * fp16 / bf16): the AdaRMSNorm -> wide projections of the K = 256 / 512 levels (image_transformer_v2.py:370-380 / :415-425 norm -> qkv_proj +
 * cosine-sim scale + RoPE; :487-491 norm -> up_proj + GEGLU; norm -> plain store) on the block-scaled fp8 matrix instruction
 * (v_mfma_scale_f32_32x32x64_f8f6f4, 2x the bf16 MFMA rate).  Weights: OCP e4m3 with ONE power-of-two scale per output channel
 * (checkpoint.quantize_fp8's rule: an fp8 checkpoint enters bit for bit; any other fp32 weight is rounded by kd_pack_weight_mx8).
 * Activations: x (bf16) * AdaRMSNorm scale (fp32) quantised per (row, 32-k block) to e4m3 with a power-of-two block scale 2^ceil(log2(amax / 448))
 * (OCP microscaling: one E8M0 byte per block, no saturation); fp32 accumulation; the RMS row factor (fp32 statistics of the unquantised row)
 * and the epilogues as in kd_gemm_bf16.  `d` is the projection's descriptor as kd_gemm_bf16 takes it (bf16 A / C, norm = 1, epi = KD_EPI_STORE /
 * KD_EPI_QKV / KD_EPI_GEGLU, precision = KD_PREC_BF16, rows_per_sample set) except that Wp points at the kd_pack_weight_mx8 image.
 * Shapes: K in {256, 512}, N a multiple of 128 (GEGLU: d_ff a multiple of 64), M >= 128 (kd_gemm_mx8_supported tells); else KD_EINVAL.
 * With c_split = 1 (KD_EPI_GEGLU only) the result leaves as the NEXT fp8 product's operand: C = e4m3 rows [M, N] (bytes), C_lo = one E8M0
 * byte per (row, 32 features) [M, N / 32], the activations' rule.
 * Second form, norm = 0 and a_split = 1 (:492 down_proj + the skip add; epi = KD_EPI_STORE / KD_EPI_RESIDUAL): A = such e4m3 rows [M, K],
 * A_lo = their scale bytes [M, K / 32] (4-byte aligned), both operands by LDS-DMA; K in {256, 512, 768, 1536}, N a multiple of 128; C / R bf16.
 * The arithmetic is restated in oracle/hdit.py (mx8_quantize_rows / mx8_quantize_weight). */
long long kd_packed_weight_bytes_mx8(int N, int K, int geglu);
int kd_pack_weight_mx8(const float* W, void* out, int N, int K, int geglu, void* stream);
int kd_gemm_mx8_supported(int M, int N, int K, int epi, int norm);
int kd_gemm_mx8(const KdGemm* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Solver step arithmetic (k_diffusion/sampling.py), one fused elementwise launch per step with
 * host-precomputed fp32 coefficients.  Operation order follows the reference expression trees
 * exactly (no FMA contraction) so results are bit-identical to the reference's CPU path.
 *   KD_STEP_EULER      : out = x + ((x - den) / c0) * c1                    (:129-134, :176, :558-560)
 *   KD_STEP_HEUN_PRED  : aux_out = (x - den)/c0 ; out = x + aux_out * c1     (:170,:179)
 *   KD_STEP_HEUN_CORR  : d2 = (x2 - den)/c0 ; out = x + ((aux + d2)/2) * c1   (:181-183)  [x2 = in2]
 *   KD_STEP_DPMPP_2M1  : out = c0 * x - c1 * den                             (:600, also :533,:535,:571,:579)
 *   KD_STEP_DPMPP_2M2  : out = c0 * x - c1 * (c2 * den - c3 * in2)            (:604-605)   [old = in2]
 *   KD_STEP_ADD_NOISE  : out = x + ((den * c0) * c1) * c2                    (:154,:572,:580,:648)  [den = noise]
 *   KD_STEP_LERP2      : out = c0 * den + c1 * in2                           (:578)
 *   KD_STEP_AXPY       : out = x + den * c0                                  (:127, :276 terms)
 *   KD_STEP_EULER_FROM : out = x + ((in2 - den) / c0) * c1                   (:213-214, :241-242)  [x2 = in2]
 *   KD_STEP_AXPBY      : out = c0 * x + c1 * den                             (:638, :679)
 *   KD_STEP_ADD_DIFF   : out = x + c0 * (den - in2)                          (:643, :645)
 *   KD_STEP_TO_D       : out = (x - den) / c0                                (:46-48 to_d)
 */
enum { KD_STEP_EULER = 0, KD_STEP_HEUN_PRED = 1, KD_STEP_HEUN_CORR = 2, KD_STEP_DPMPP_2M1 = 3,
       KD_STEP_DPMPP_2M2 = 4, KD_STEP_ADD_NOISE = 5, KD_STEP_LERP2 = 6, KD_STEP_AXPY = 7,
       KD_STEP_EULER_FROM = 8, KD_STEP_AXPBY = 9, KD_STEP_ADD_DIFF = 10, KD_STEP_TO_D = 11 };
int kd_sampler_step_f32(int op, const float* x, const float* den, const float* in2, float* out, float* aux,
                        float c0, float c1, float c2, float c3, long long n, void* stream);

/* Karras preconditioner for a foreign inner model (k_diffusion/layers.py:88-90):
 *   kd_precond_in : y = x * c_in(sigma_b)
 *   kd_precond_out: y = f * c_out(sigma_b) + x * c_skip(sigma_b)          per_sample = C*H*W */
int kd_precond_in_f32(const float* x, const float* sigma, float* y, float sigma_data, int batch, long long per_sample, void* stream);
int kd_precond_out_f32(const float* f, const float* x, const float* sigma, float* y, float sigma_data, int batch,
                       long long per_sample, void* stream);

/* DPM-Solver (k_diffusion/sampling.py:333-480) in the reference's operation order:
 *   kd_dpm_eps_f32:     out = (x - denoised) / sigma                                  (DPMSolver.eps :354)
 *   kd_dpm_combine_f32: out = x - a * eps [- b * (eps_r - eps) when eps_r != NULL]      (x_1 :363, u1 :371, x_2 :373, u2 :384, x_3 :387)
 *   kd_dpm_error_f32:   the adaptive solver's local error (:464-465) as kd_dpm_error_partials() per-block partial sums of
 *                       ((x_low - x_high) / max(atol, rtol * max(|x_low|, |x_prev|)))^2 on a fixed grid (reproducible total). */
int kd_dpm_eps_f32(float* out, const float* x, const float* denoised, float sigma, long long n, void* stream);
int kd_dpm_combine_f32(float* out, const float* x, const float* eps, const float* eps_r, float a, float b, long long n, void* stream);
int kd_dpm_error_partials(void);
int kd_dpm_error_f32(const float* x_low, const float* x_high, const float* x_prev, float atol, float rtol, long long n, float* partial,
                     void* stream);

/* Foreign-model wrappers (k_diffusion/external.py): the image-sized arithmetic of their forward()s,
 *   y[b, :] = f[b, :] * a[b] (+ x[b, :] * c[b] when x != NULL)          (:38, :113, :162)
 * and the discrete schedule's sigma <-> t maps over an ascending table log_sigmas[n] (:66-84). */
int kd_rows_affine_f32(const float* f, const float* x, const float* a, const float* c, float* y, int batch, long long per_sample, void* stream);
int kd_sigma_to_t_f32(const float* sigma, const float* log_sigmas, float* t, int count, int n, int quantize, void* stream);
int kd_t_to_sigma_f32(const float* t, const float* log_sigmas, float* sigma, int count, int n, void* stream);

/* Brownian-interval noise (stands in for torchsde.BrownianTree behind
 * k_diffusion/sampling.py:65-114): out[b, i] = sign * (W_b,i(t1) - W_b,i(t0)) * inv_norm where W is
 * a virtual Brownian tree on [T0, T1] (depth-`depth` dyadic bridge, Philox4x32-10 keyed by
 * seeds[b], counter = (element index, tree node)); path-consistent across nested queries.
 * kd_brownian_cached_f32 is the same function with the end-point values W(t0) / W(t1) kept by the
 * caller (the tensors torchsde's tree caches on the host, sampling.py:72-79): have0 / have1 != 0
 * reads that end point from w0 / w1 [batch, per_sample] instead of descending the tree; otherwise it
 * is computed and, when the pointer is non-NULL, stored there for a later query. */
int kd_brownian_f32(float* out, const unsigned long long* seeds, int batch, long long per_sample,
                    double T0, double T1, double t0, double t1, float mult, int depth, void* stream);
int kd_brownian_cached_f32(float* out, float* w0, float* w1, int have0, int have1, const unsigned long long* seeds, int batch,
                           long long per_sample, double T0, double T1, double t0, double t1, float mult, int depth, void* stream);

/* Index-addressed standard normals: the start noise of a sampling run (/root/reference sample.py:59, `torch.randn([n, C, H, W],
 * device=device) * sigma_max`) and the randn_like of the ancestral samplers (k_diffusion/sampling.py:61-62), drawn on the device as a
 * function of (seeds[b], draw, element) only -- so image i of a seeded job is the same for any batch size / GPU count (the reference's
 * draw comes from rank-local generator state).  out[b, e] = scale * z, z from Philox4x32-10 (key seeds[b], counter (e >> 2, draw | 2^63))
 * through two Box-Muller pairs per block; `draw` < 2^63 numbers the calls of one run (0 = start noise).  out: 16-byte aligned. */
int kd_randn_f32(float* out, const unsigned long long* seeds, int batch, long long per_sample, unsigned long long draw, float scale,
                 void* stream);

/* Final image conversion (k_diffusion/utils.py:27-34 to_pil_image): u8 = trunc((clamp(x,-1,1)+1)/2*255)
 * (torchvision's to_pil_image does mul(255).byte(), i.e. truncation) */
int kd_to_uint8(const float* x, unsigned char* y, long long n, void* stream);

/* Per-launch timing hooks for bench.py (HIP events recorded on `stream` around each launch). */
int kd_prof_enable(int on);
int kd_prof_count(void);
int kd_prof_get(int i, char* name, int name_cap, float* ms, double* flops, double* bytes);
int kd_prof_reset(void);
/* In-kernel time line probe (benchmarks/): while `dev_ptr` (16 x uint64 of device memory) is set, workgroup 0 of the bf16 GEMM kernels
 * (W-stationary, A-stationary, tiled) and of kd_ffn_bf16 writes s_memtime stamps: [0] entry, [2] exit, [1] / [3] s_memrealtime at
 * entry / exit (shader clock under load = ([2]-[0]) / ([3]-[1]) x 100 MHz), [4] end of the row prologue / first blocks in,
 * [5] end of the first tile's K loop (tiled, ffn: of the whole loop), [6] end of its epilogue, [7] number of ring blocks / tiles.
 * kd_ffn_f32 also stamps its third d_ff tile: [8] start, [9] up projection done, [10] GEGLU done, [11] down k-steps done, and [12]
 * the end of the tile loop.  NULL switches it off.  Not for concurrent launches.
 * Extended form (round 4; the fp32-parity kernels of gemm_x3.hip, gemm_x3r.hip, ffn_x3.hip, attn_x3.hip): when entry [15] holds 0x4b44 the buffer
 * must provide 32 + 3 * grid entries, and EVERY workgroup's first thread writes s_memrealtime (100 MHz ticks) at its entry to [32 + 3 b] and at
 * its exit to [32 + 3 b + 1] (b = blockIdx.x; [.. + 2]: exit of the workgroup's first loader wave where there is one): launch ramp, rounds,
 * tail and the spread of the workgroups' lifetimes (benchmarks/wg_timeline.py, benchmarks/x3r_bench.py). */
int kd_prof_clock_buffer(void* dev_ptr);

/* A forward's launch list in ONE host call (round 4).  The Python mirror used to issue the 58 - 66 launches of a model call one ctypes call
 * at a time; kd_run_list walks an array of calls and invokes the entry point each one names with the arguments it carries, in order, on
 * `stream`; it stops at the first failure, returns that entry point's code and stores the index in *failed (if not NULL).  Descriptors are
 * passed by ADDRESS (p[0]) and read at the call, so a caller may keep patching them between runs.  Arguments of the other entry points:
 * their pointer arguments in declaration order in p[], their int arguments in declaration order in i[], their one float (eps) in f.
 *   KD_OP_GEMM_F32 / KD_OP_GEMM_BF16 : p[0] = const KdGemm*          KD_OP_FFN_F32 / KD_OP_FFN_BF16 : p[0] = const KdFfn*
 *   KD_OP_ATTN_GLOBAL_F32 : p = qkv, out, scale_h, cos_t, sin_t; i = batch, T, nh, prep, precision; f = eps
 *   KD_OP_ATTN_WINDOW_F32 : p as above; i = batch, H, W, nh, ws, shift, prep, precision        KD_OP_ATTN_NA2D_F32 : i = batch, H, W, nh, ks, prep, precision
 *   KD_OP_ATTN_GLOBAL_BF16 : p = qkv, out; i = batch, T, nh      KD_OP_ATTN_WINDOW_BF16 : i = batch, H, W, nh, ws, shift      KD_OP_ATTN_NA2D_BF16 : i = batch, H, W, nh, ks
 *   KD_OP_NORM_SPLIT_F32 : p = x, scale, hi, lo; i = scale_stride, rows_per_sample, M, K; f = eps
 *   KD_OP_ATTN_BLOCK_BF16 / KD_OP_PROJ_BLOCK_BF16 / KD_OP_GEMM_MX8 : p[0] = const KdGemm* */
enum { KD_OP_GEMM_F32 = 0, KD_OP_GEMM_BF16 = 1, KD_OP_FFN_F32 = 2, KD_OP_FFN_BF16 = 3,
       KD_OP_ATTN_GLOBAL_F32 = 4, KD_OP_ATTN_WINDOW_F32 = 5, KD_OP_ATTN_NA2D_F32 = 6,
       KD_OP_ATTN_GLOBAL_BF16 = 7, KD_OP_ATTN_WINDOW_BF16 = 8, KD_OP_ATTN_NA2D_BF16 = 9, KD_OP_NORM_SPLIT_F32 = 10,
       KD_OP_ATTN_BLOCK_BF16 = 11, KD_OP_PROJ_BLOCK_BF16 = 12, KD_OP_GEMM_MX8 = 13 };
typedef struct {
  int op;             /* KD_OP_* */
  float f;
  const void* p[5];
  int i[8];
} KdCall;
int kd_run_list(const KdCall* calls, int n, void* stream, int* failed);

#ifdef __cplusplus
}
#endif
#endif
