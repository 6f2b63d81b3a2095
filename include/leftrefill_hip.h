/*
 * leftrefill_hip.h -- C ABI of the MI355X (gfx950) kernels behind LeftRefill's diffusion-sampling hot path.
 *
 * Boundary contract:
 *   - extern "C", raw device pointers + explicit sizes + a hipStream_t (passed as void*), no torch types;
 *   - no allocation, no synchronisation, no global mutable state inside; the caller owns every buffer,
 *     including workspaces; every call is stream-ordered and re-entrant;
 *   - return value: 0 = ok, > 0 = hipError_t of the launch, < 0 = argument error (LR_E_*).
 *
 * Layout convention on the device: activations are NHWC fp16 ("token-major": row m = (n*H + y)*W + x,
 * channels contiguous), weights are [Cout][tap][Cin] fp16 (K contiguous), statistics and accumulators fp32.
 *
 * Each entry point cites the reference call site it replaces (paths relative to the reference repo).
 */
#ifndef LEFTREFILL_HIP_H
#define LEFTREFILL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR_E_ARG (-1)      /* bad size / null pointer */
#define LR_E_ALIGN (-2)    /* channel count / leading dimension not a multiple the kernel needs */
#define LR_E_UNSUPPORTED (-3)

typedef void* lr_stream_t; /* hipStream_t */
typedef uint16_t lr_half;  /* 16-bit activation / weight element: IEEE binary16 bits, or bfloat16 bits in the *_bf16 entry points */
/* 16-bit compute type.  Every kernel exists for both: fp16 is what the reference's autocast inference and its fp16 AMP training
 * use (train_inpainting.py:52,127); bf16 is BASELINE configs[4].  Entry points named lr_<op>_bf16 have the signature of
 * lr_<op> (lr_<op>_f16) and read / write bfloat16 (declared at the end of this header); lr_gemm_args carries a `dtype` field. */
#define LR_DTYPE_F16 0
#define LR_DTYPE_BF16 1
/* lr_gemm_args.pipe */
#define LR_PIPE_DEFAULT 0
#define LR_PIPE_W8_DEEP 4
#define LR_PIPE_HALO 8

/* ABI version; bump on any signature change. */
int lr_abi_version(void);

/* ---- layout converters at the UNet boundary ------------------------------------------------------------------
 * replaces: `h = x.type(self.dtype)` + the NCHW<->token-major rearranges, openaimodel.py:775, attention.py:402,412;
 *           `torch.cat([x] + c_concat, dim=1)` ddpm.py:1349 (optional second source x2).
 * y[n,h,w,0:C1] = x1[n,:,h,w]; y[...,C1:C1+C2] = x2 (may be NULL with C2 = 0); y[...,C1+C2:Cpad] = 0. */
int lr_nchw_f32_to_nhwc_f16(const float* x1, int C1, const float* x2, int C2, lr_half* y, int Cpad, int N, int H,
                            int W, lr_stream_t s);
/* out_nchw[n,c,h,w] = y[n,h,w,c] for c < C; out_is_f32 selects float or half output (UNet returns eps). */
int lr_nhwc_f16_to_nchw(const lr_half* y, int Cstride, int C, void* out_nchw, int out_is_f32, int N, int H, int W,
                        lr_stream_t s);

/* ---- GroupNorm(32) [+ SiLU] -------------------------------------------------------------------------------------
 * replaces: GroupNorm32 / normalization (util.py:202-219, eps 1e-5) followed by nn.SiLU in ResBlock.in_layers /
 *           out_layers / UNetModel.out (openaimodel.py:200-231,726-731) and Normalize (attention.py:90-91, eps 1e-6).
 * Input is the virtual concat [x1 (C1 ch) | x2 (C2 ch)] (th.cat([h, hs.pop()], 1), openaimodel.py:781); x2 may be NULL.
 * Two launches: stats writes per-chunk partial (sum, sumsq) to `partials` (caller provides N*LR_GN_CHUNKS*64 floats; the
 * number of chunks actually used is a deterministic function of N and HW) -- no atomics, bitwise reproducible; apply
 * finalises mean/rstd in fp64 from the partials and writes y [N*HW][C1+C2] fp16. */
#define LR_GN_CHUNKS 256
int lr_groupnorm_stats(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, float* partials,
                       lr_stream_t s);
int lr_groupnorm_apply(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, const float* partials,
                       const float* gamma, const float* beta, float eps, int silu, lr_half* y, lr_stream_t s);
/* Statistics from the producers instead of a pass over x: p1 / p2 = gn_stats_out of the GEMMs that wrote x1 / x2
 * ([M / R1][C1][2], [M / R2][C2][2], M = N*HW; R1, R2 must divide HW).  finalize reduces them (fp64, fixed order) to
 * group sums in the layout lr_groupnorm_apply_n reads with nchunks = 1: partials [N][1][32][2]. */
int lr_groupnorm_finalize(const float* p1, int C1, int R1, const float* p2, int C2, int R2, int N, int HW, float* partials,
                          lr_stream_t s);

/* ---- the UNet's `out` block in one launch (ABI 21) -------------------------------------------------------------------------
 * replaces: `self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3,
 *           padding=1)))` applied as `self.out(h)` (reference ldm/modules/diffusionmodules/openaimodel.py:714-718, 812) and the
 *           NHWC -> NCHW conversion of its result: GroupNorm(32) + SiLU + 3x3 pad-1 conv to Cout <= 4 channels, input read once.
 *   x [B][H][W][C] fp16 (NHWC tokens), gpart [B][chunks][32][2] per-group (sum, sumsq) partials of x from its producer
 *   (lr_gemm_args.gn_group_out), gamma / beta [C] fp32; w [>= Cout][ldw] fp16 with k = tap * C + channel (the packed conv weight,
 *   rows >= Cout ignored), bias [>= Cout] fp32 or NULL; y [B][Cout][H][W] fp16 (NCHW).  The normalised + SiLU'd activation is
 *   rounded to fp16 before the product exactly like the two-launch path stores it.
 *   C % 64 == 0, H % 8 == 0, W % 16 == 0, Cout <= 4, 9 C small enough for LDS (C <= 704): otherwise LR_E_UNSUPPORTED. */
int lr_gn_conv_out_f16(const lr_half* x, int B, int H, int W, int C, const float* gpart, int chunks, const float* gamma, const float* beta,
                       float eps, const lr_half* w, int ldw, const float* bias, int Cout, lr_half* y, lr_stream_t s);
/* lr_groupnorm_apply with an explicit chunk count of `partials` ([N][nchunks][32][2]) */
int lr_groupnorm_apply_n(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, const float* partials,
                         int nchunks, const float* gamma, const float* beta, float eps, int silu, lr_half* y, lr_stream_t s);

/* ---- LayerNorm over the channel dimension -------------------------------------------------------------------------
 * replaces: nn.LayerNorm norm1/2/3 in BasicTransformerBlock (attention.py:271-273, eps 1e-5). x,y [M][C] fp16. */
int lr_layernorm(const lr_half* x, const float* gamma, const float* beta, float eps, lr_half* y, int M, int C,
                 lr_stream_t s);

/* ---- timestep embedding + small-M linear (time MLP) -------------------------------------------------------------
 * replaces: timestep_embedding (util.py:154-174; cos first), UNetModel.time_embed (openaimodel.py:528-532) and the
 *           22 ResBlock.emb_layers (217-223) batched as one [sum Cout][1280] weight. */
int lr_timestep_embedding(const int64_t* t, int N, int dim, lr_half* out, lr_stream_t s);
/* out[m][n] = act_out( sum_k act_in(a[m][k]) * w[n][k] + bias[n] ), M <= 16; act: 0 none, 1 SiLU. K % 8 == 0. */
int lr_linear_small_m(const lr_half* a, int lda, const lr_half* w, const float* bias, lr_half* out, int ldo, int M,
                      int N, int K, int act_in, int act_out, lr_stream_t s);

/* ---- implicit-GEMM convolution / linear on MFMA --------------------------------------------------------------------
 * replaces: conv3x3 s1/s2 (+bias) in ResBlock / Downsample / Upsample / input / output conv (openaimodel.py:106,150,
 *           200-231,546,730), the 1x1 skip_connection (240), nn.Linear in CrossAttention / SpatialTransformer /
 *           GEGLU / FeedForward (attention.py:54,74,156-163,359,381); fused epilogues replace `h + emb_out`
 *           (openaimodel.py:272), `skip(x) + h` (274), the attention/FF residual adds (attention.py:280-282,419) and
 *           `x * F.gelu(gate)` (attention.py:56-58).
 *
 *   C[m][n] = sum_{tap,c} A(m,tap,c) * Wt[n][tap*(C1+C2)+c]  (+ bias[n]) (+ rowvec[m / rows_per_batch][n]) (+ R[m][n])
 *   rows m = (b*H + y)*W + x over the OUTPUT grid; A gathers from the source grid [Hs][Ws]:
 *     taps = 1: pointwise (Linear / 1x1);  taps = 9: 3x3 pad 1 with `stride` (1|2) and optional nearest-2x `up`sample.
 *   Source is the virtual channel concat [p1 (C1) | p2 (C2)];  C1, C2 multiples of 64.
 *   geglu == 1: Wt/bias rows are pre-interleaved in 16-row groups [u16 | g16 | ...]; output has N/2 columns:
 *     out = (u + bu) * gelu_erf(g + bg).
 *   geglu == 2: plain erf-GELU of (acc + bias [+ rowvec]) before the residual (the text tower's MLP, nn.GELU).
 */
typedef struct lr_gemm_args {
  const lr_half* p1; int32_t C1;
  const lr_half* p2; int32_t C2;
  int32_t B, H, W;          /* output grid; M = B*H*W */
  int32_t Hs, Ws;           /* source grid */
  int32_t taps, stride, up; /* up: 0 none | 1 nearest 2x | 2 zero-insertion 2x (odd source rows / columns read as 0: with
                               flipped, transposed weights this is the input gradient of a stride-2 conv) */
  int32_t asym;             /* 0: 3x3 pad 1 on every side; 1: pad only bottom/right (F.pad (0,1,0,1) + padding 0, the VAE
                               Downsample, ldm/modules/diffusionmodules/model.py:83-86) */
  const lr_half* wt; int32_t N;      /* weights [N][taps*(C1+C2)] */
  const float* bias;                 /* [N] or NULL */
  const lr_half* rowvec; int32_t ld_rowvec; /* [B][ld_rowvec] per-sample vector added to every row of sample b, or NULL */
  const lr_half* resid; int32_t ld_resid;   /* [M][ld_resid] or NULL */
  lr_half* out; int32_t ld_out;             /* [M][ld_out] */
  int32_t geglu;
  int32_t tile_n;           /* 0 = auto; 64 | 128 | 160 (tile_m 128) or 128 | 160 | 256 | 320 (tile_m 256) */
  int32_t tile_m;           /* 0 = auto; 128 (4 waves, 2-stage) | 256 (8 waves, 3-stage counted-vmcnt pipeline) */
  /* split-K (small-M shapes that cannot fill 256 CUs): fp32 partial tiles go to `workspace`
   * [splits][M][N] and a second launch reduces them in a fixed order and applies the epilogue.
   * splits: 0 = auto, 1 = off.  workspace may be NULL (=> no split).  Not available with geglu. */
  int32_t splits;
  float* workspace; int64_t workspace_bytes;
  /* LayerNorm folded into a pointwise GEMM (replaces nn.LayerNorm norm1/2/3 + the Linear that follows it,
   * attention.py:271-283): p1 is the RAW x [M][K], wt = W * gamma (gamma folded along K), bias = W beta + b,
   * ln_colsum[n] = sum_k wt[n][k] (fp32, of the fp16-rounded wt):
   *     out[m][n] = rstd[m] * (acc[m][n] - mean[m] * ln_colsum[n]) + bias[n]
   * mean / rstd of row m are finalised inside the kernel from ln_stats [M][ln_parts][2] = per-row partial (sum, sumsq)
   * written by the GEMM that produced x (stats_out below).  ln_stats == NULL: plain GEMM. */
  const float* ln_stats; int32_t ln_parts; float ln_eps;
  const float* ln_colsum;
  /* stats_out != NULL: per-row (sum, sumsq) of the fp16-rounded output over each wave's column range,
   * [M][lr_gemm_stats_parts(args)][2] fp32 (fixed order, no atomics). Not with split-K. */
  float* stats_out;
  /* gn_stats_out != NULL: per-channel (sum, sumsq) of the fp16-rounded output over each block of R = lr_gemm_gn_rows(args)
   * consecutive rows, [ceil(M / R)][N][2] fp32 -- the statistics pass of the GroupNorm that consumes this tensor
   * (openaimodel.py:254-274) comes out of the producer's epilogue (of the reduce kernel, R = 32, when the call splits K);
   * lr_groupnorm_finalize turns the blocks of one sample (H*W must be a multiple of R) into per-group sums.
   * Fixed order, no atomics.  Not with GEGLU. */
  float* gn_stats_out;
  int32_t dtype;            /* LR_DTYPE_F16 | LR_DTYPE_BF16: type of p1, p2, wt, rowvec, resid, out (geglu == 2 is fp16 only) */
  int32_t pipe;             /* pipeline variant of the tile: LR_PIPE_DEFAULT (0) = the tile's standard instance (tile_m 128: 4 waves,
                             * 2-stage ring, 2-4 blocks per CU; 256 x {128,160}: 3-stage ring; 256 x {256,320}: 2-stage ring).
                             * LR_PIPE_W8_DEEP (4), tile_m 128 with tile_n 128 | 160: 8 waves (4 x 2, wave tile 32 x tile_n/2), 4-stage
                             * ring, one block per CU (the 4096- / 1024-row levels).
                             * LR_PIPE_HALO (8, ABI 23), tile_m 256 with tile_n 160 | 320: the halo-tile conv -- 3x3, stride 1, pad 1, no
                             * upsample, W a multiple of 16 and H a multiple of 16 (or H = 8 with an even number of samples: a tile is then the
                             * 8 x 16 pixels of two samples); split-K by whole 64-channel chunks: a block owns a 16 x 16 pixel tile, the 18 x 18 input patch of a
                             * 64-channel chunk is copied to LDS once and the nine taps are shifted reads of it (conv_halo.hip); K is
                             * accumulated chunk-major, so results agree with the other instances to fp32 rounding, not bit for bit.
                             * Anything else: LR_E_UNSUPPORTED */
  /* gn_group_out != NULL (ABI 20): per-GROUP (sum, sumsq) of the fp16-rounded output for the GroupNorm(32) of a consumer that
   * normalises THIS tensor alone (N / 32 channels per group), [samples][chunks][32][2] fp32 with chunks = lr_gemm_gn_group_chunks(args)
   * row tiles per sample (a sample = gn_hw rows; 0 = H * W): the tile reduces its per-channel sums to the groups it covers, so
   * lr_groupnorm_apply_n(partials = gn_group_out, nchunks = chunks) runs without lr_groupnorm_finalize.  LR_E_ARG when the plan
   * cannot produce it (lr_gemm_gn_group_chunks == 0: tile width not a whole number of groups, tiles straddling samples).  Fixed
   * order, no atomics; behind split-K it comes out of the reduce kernel (32-row chunks).  May be combined with gn_stats_out. */
  float* gn_group_out; int32_t gn_hw;
  /* wt_bstride != 0 (ABI 20): per-sample weights / bias -- rows of sample b = m / (H * W) use wt + b * wt_bstride and
   * bias + b * bias_bstride (elements).  Pointwise calls only (taps == 1), H * W a multiple of the tile's rows, no split-K.
   * Used for the GroupNorm of SpatialTransformer folded into proj_in (lr_gn_fold_weights_f16). */
  int32_t wt_bstride, bias_bstride;
  /* wt_pm != 0 (ABI 20): wt is piece-major, [K / 64][N][64] -- the 64-element (128-byte) pieces of all N rows for K-step 0, then for
   * K-step 1, ... (lr packing: w.reshape(N, K / 64, 64).permute(1, 0, 2)).  Same arithmetic, same K order; a tile's weight slice of one
   * K-step is then one contiguous run instead of tile_n pieces 2 K bytes apart.  Not with wt_bstride. */
  int32_t wt_pm;
  /* skip1 != NULL (ABI 22): pointwise K extension -- out = conv3x3([p1 | p2]) + W_s [skip1 | skip2] in ONE accumulation:
   *   wt = [N][9 (C1 + C2) + Cs1 + Cs2] (the 3x3 part first, then the pointwise weights over skip1's, then skip2's channels),
   *   bias = the sum of the two layers' biases.  skip1 / skip2: [B*H*W, Cs1 / Cs2] at the OUTPUT resolution (virtual channel concat).
   * replaces: `self.skip_connection(x) + h` of ResBlock._forward (openaimodel.py:274) where skip_connection is the 1x1 conv of a block
   *           whose width changes (253-259): the separate GEMM, its [M, N] output and the residual read of this conv's epilogue.
   * With taps == 1 the main part is pointwise too: out = W_a [p1 | p2] + W_s [skip1 | skip2] (wt = [N][C1 + C2 + Cs1 + Cs2]) -- used for
   * SpatialTransformer.proj_out composed with the last feed-forward Linear (attention.py:75-77, 412-419): proj_out(ff2(g) + x) + x_in =
   * (Wp W2) g + Wp x + (Wp b2 + bp) + x_in, one GEMM with resid = x_in.
   * stride 1, no upsample, no GEGLU / LayerNorm fold / per-sample weights: anything else LR_E_UNSUPPORTED.  resid / rowvec / statistics
   * outputs work as without it. */
  const lr_half* skip1; const lr_half* skip2; int32_t Cs1, Cs2;
  /* splitk_mode (ABI 23): 0 = the split-K partials are reduced by a second launch (fixed order); 1 (developer builds only -- measured
   * slower, profiles/r05_splitk_inlaunch.txt; the product library answers LR_E_UNSUPPORTED, ABI 24) = in-launch reduce where the plan
   * allows it (8-wave instances with 160- / 320-column tiles, tiles x splits <= 256 so that every K-slice block of the grid is
   * resident at once): every slice block publishes its partial tile write-through, waits for its tile's other slices on an
   * agent-scope counter (bounded spin) and reduces its share of the tile's rows in slice order with the full epilogue -- still
   * deterministic, no second launch.  Plans that do not qualify fall back to mode 0 silently. */
  int32_t splitk_mode;
} lr_gemm_args;
/* row tiles per sample of gn_group_out for this call, 0 if the plan cannot produce per-group sums */
int lr_gemm_gn_group_chunks(const lr_gemm_args* args);
/* rows per block of gn_stats_out (a function of the tile that will be used) */
int lr_gemm_gn_rows(const lr_gemm_args* args);
/* plan[0..3] = (tile_m, tile_n, splits, pipe) the call would use: explicit requests as given, zeros resolved by the static
 * heuristics (a pure function of the shape -- never of timing; the Python front end ships its tuned choices as a table). */
int lr_gemm_plan(const lr_gemm_args* args, int32_t* plan);
/* number of per-row partials this call writes to stats_out (a function of N and the tile that will be used) */
int lr_gemm_stats_parts(const lr_gemm_args* args);
/* bytes of workspace lr_gemm_conv_f16 would like for this problem (0 if it will not split) */
int64_t lr_gemm_workspace_bytes(const lr_gemm_args* args);
int lr_gemm_conv_f16(const lr_gemm_args* args, lr_stream_t s);
/* number of bounded-spin timeouts of the in-launch split-K reduce since the library was loaded (0 in a healthy run; synchronises
 * the device; -1 on a runtime error).  A timeout means a K-slice block did not see its tile's other slices arrive: wrong results
 * for that launch instead of a hang. */
int lr_gemm_splitk_timeouts(void);

/* ---- fused scaled-dot-product attention (flash-style, d_head = 64) ---------------------------------------------
 * replaces: xformers.ops.memory_efficient_attention (attention.py:236) == softmax(q k^T * d^-0.5) v of the vanilla
 *           path (attention.py:173-195), self (Nkv = HW or view_num*h^2) and cross (Nkv = 77).
 * q [B][Nq][ldq], k/v [B][Nkv][ldk|ldv], o [B][Nq][ldo]; head h occupies columns [h*64, h*64+64). */
int lr_attention_f16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o,
                     int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s);

/* Causal self-attention (query i sees keys j <= i): the OpenCLIP text tower of the prompt encoder
 * (ldm/modules/encoders/Refill_modules.py:189-201, model.attn_mask), N = 77 tokens, heads of 64. */
int lr_attention_causal_f16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o,
                            int ldo, int B, int heads, int N, float scale, lr_stream_t s);
/* Same attention with V supplied pre-transposed: vt [B][heads*64][ld_vt] from lr_transpose_v_f16 (ld_vt = Nkv rounded up
 * to 64, tail keys zero, keys permuted inside every group of 16 to the MFMA k-slot order).  The V tile then streams into
 * LDS by DMA like K; worth it for long key sequences (self-attention), the transpose costs one read + write of V. */
int lr_attention_vt_f16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* vt, int ld_vt, lr_half* o,
                        int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s);
int lr_transpose_v_f16(const lr_half* v, int ldv, lr_half* vt, int ld_vt, int B, int heads, int Nkv, lr_stream_t s);

/* ---- fused cross-attention block (C = 320, 5 heads of 64: level 0 of the SD2 UNet) --------------------------------
 * replaces: `x = self.attn2(self.norm2(x), context=context) + x` (attention.py:281) = norm2 (LayerNorm, attention.py:272),
 *           CrossAttention.to_q, softmax(q k^T * d^-0.5) v over the <= 96 context tokens, to_out[0] + bias
 *           (attention.py:165-196) and the residual add, in ONE launch: x is read once, out written once, q and the attention
 *           output never leave the registers (instead of LayerNorm-folded to_q GEMM -> attention -> to_out GEMM).
 *   x, out [M][320]; M = B * HW rows, a block of 128 rows stays inside one sample (M % 128 == 0, HW % 128 == 0).
 *   wq [320][320] = to_q.weight * gamma (LayerNorm folded along K), bq [320] = to_q.weight @ beta (fp32);
 *   k  [B * Lc][ldk]: K projection of the context whose columns are permuted inside every head: position 32 p + 8 f + i
 *      (f < 4, i < 8) holds channel 32 p + 16 (i >> 2) + 4 f + (i & 3) -- project the context with the same permutation
 *      of to_k.weight's rows;
 *   vt [B][5][2][64][64]: lr_xattn_pack_vt_f16 of the V projection (transposed, key slots in the same order);
 *   wo [5][320][64] = to_out[0].weight as per-head pieces (piece h = columns 64 h .. 64 h + 63, 40 KB contiguous) with the columns of
 *      every piece in that order, bo [320] = to_out[0].bias (fp32);
 *   stats_out (optional) [M][2]: per-row (sum, sumsq) of the rounded output (lr_gemm_args.ln_stats of the next GEMM, ln_parts = 1).
 * Anything else (other widths, Lc > 96, ragged M): LR_E_UNSUPPORTED -- callers keep the three-kernel path for those. */
typedef struct lr_xattn_args {
  const lr_half* x; lr_half* out;
  const lr_half* wq; const float* bq;
  const lr_half* k; int32_t ldk;
  const lr_half* vt;
  const lr_half* wo; const float* bo;
  float* stats_out;
  int32_t M, HW, C, heads, Lc;
  float ln_eps, scale;
  /* pre_a != NULL: the out-projection of the preceding self-attention runs in front, in the same launch
   * (`x = self.attn1(self.norm1(x)) + x`, attention.py:280):  x1 = pre_a pre_w^T + pre_b + x;  out = x1 + to_out(attention(LayerNorm(x1) ...)).
   * pre_a [M][320] = the self-attention output, pre_w [320][320] = attn1.to_out[0].weight (natural layout), pre_b [320] its bias (fp32);
   * wq must then have its COLUMNS in the k-slot order (inside every 64 columns: position 32 p + 8 f + i holds column
   * 32 p + 16 (i >> 2) + 4 f + (i & 3)): the normalised x1 reaches the q projection in accumulator order, never through memory. */
  const lr_half* pre_a; const lr_half* pre_w; const float* pre_b;
} lr_xattn_args;
int lr_xattn_block_f16(const lr_xattn_args* args, lr_stream_t s);
int lr_xattn_pack_vt_f16(const lr_half* v, int ldv, lr_half* vt, int B, int heads, int Lc, lr_stream_t s);

/* ---- entry of a SpatialTransformer's block (C = 320: level 0 of the SD2 UNet; ABI 24) --------------------------------------
 * replaces: `x = self.proj_in(x)` (use_linear, attention.py:405-408) and, of the block that follows, `self.norm1(x)` (attention.py:280)
 *           with attn1's `q = self.to_q(x); k = self.to_k(context); v = self.to_v(context)` on context = x (attention.py:168-172), in ONE
 *           launch instead of a K = 320 GEMM and a LayerNorm-folded 960-column GEMM:
 *               x1 = x wp^T + bp;     qkv = LayerNorm(x1) [Wq; Wk; Wv]^T
 *   x [M][320] (the GroupNorm-ed tokens), M % 256 == 0;  wp [320][320] = proj_in.weight, bp [320] its bias (fp32);
 *   wqkv [NQ][320] = [to_q; to_k; to_v].weight * gamma (LayerNorm folded along K, natural column order), bqkv [NQ] = W beta (fp32);
 *      NQ a multiple of 192 (960 for the SD2 block);
 *   x1 [M][320] and qkv [M][ld_qkv] out (x1 is rounded to 16 bits before the LayerNorm, like the two-launch path stores it).
 * Other widths / ragged M: LR_E_UNSUPPORTED -- callers keep the two-GEMM path. */
typedef struct lr_stin_args {
  const lr_half* x; const lr_half* wp; const float* bp; const lr_half* wqkv; const float* bqkv;
  lr_half* x1; lr_half* qkv;
  int32_t M, C, NQ, ld_qkv;
  float ln_eps;
  /* gn_part != NULL: x holds the RAW tokens and `x = self.norm(x)` of SpatialTransformer.forward (attention.py:399-404: GroupNorm(32, eps 1e-6,
   * affine), no activation) is applied to the rows as they are loaded -- same arithmetic as lr_groupnorm_apply_n, bit for bit, without
   * writing the normalised tensor.  gn_part [M / gn_hw][gn_chunks][32][2] = per-group (sum, sumsq) partials of x from its producer
   * (lr_gemm_args.gn_group_out / lr_ffn_args.gn_group_out), gn_hw rows per sample (gn_hw % 256 == 0), gn_gamma / gn_beta [320] fp32. */
  const float* gn_part; const float* gn_gamma; const float* gn_beta;
  int32_t gn_chunks, gn_hw;
  float gn_eps;
} lr_stin_args;
int lr_stin_block_f16(const lr_stin_args* args, lr_stream_t s);

/* ---- row-resident [LayerNorm +] Linear (C = 640 | 1280: levels 1 / 2 of the SD2 UNet; ABI 24) ---------------------------------
 * replaces: `self.norm1(x)` + attn1's fused `to_q / to_k / to_v` (attention.py:280, 168-172; geglu = 0, N = 3 C) and `self.norm3(x)` +
 *           GEGLU.proj + `x * F.gelu(gate)` (attention.py:282, 51-58; geglu = 1, N = 8 C -> 4 C output columns) where the tiled GEMM's
 *           K loop is only 10 steps long:   out = [gate of] LayerNorm(x) w^T + bias,  rows held in registers, weights streamed.
 *   x [M][C], C = 640 | 1280 (levels 1 / 2), M % 128 == 0;  w [N][C] = weight * gamma (LayerNorm folded along K), bias [N] = W beta + b (fp32); N % 64 == 0;
 *   geglu = 1: w / bias rows interleaved in 16-row groups [u16 | g16 | ...] (the layout of lr_gemm_args.geglu == 1), out [M][N / 2];
 *   out [M][ld_out].  The LayerNorm is two-pass in registers on the 16-bit x (no producer statistics needed).
 * Other widths / ragged M: LR_E_UNSUPPORTED -- callers keep the LayerNorm-folded GEMM. */
typedef struct lr_rowlin_args {
  const lr_half* x; const lr_half* w; const float* bias; lr_half* out;
  int32_t M, C, N, ld_out, geglu;
  float ln_eps;
  int32_t ln;      /* 1: LayerNorm the rows first (w / bias carry gamma / beta); 0: plain Linear (SpatialTransformer.proj_in, attention.py:405-408) */
  /* gn_part != NULL (ln = 0, geglu = 0): x holds RAW tokens and SpatialTransformer.norm (GroupNorm(32), attention.py:399-404) is applied to the
   * rows as they are loaded, exactly as in lr_stin_args (gn_hw % 128 == 0). */
  const float* gn_part; const float* gn_gamma; const float* gn_beta;
  int32_t gn_chunks, gn_hw;
  float gn_eps;
} lr_rowlin_args;
int lr_rowlin_f16(const lr_rowlin_args* args, lr_stream_t s);

/* ---- fused feed-forward block (C = 320: level 0 of the SD2 UNet) ------------------------------------------------------
 * replaces: `x = self.ff(self.norm3(x)) + x` (attention.py:282) = norm3 (LayerNorm), GEGLU.proj + `x * F.gelu(gate)` (attention.py:51-58),
 *           FeedForward.net[2] Linear + bias (attention.py:74-78) and the residual add, in ONE launch: the [M][H] hidden activation
 *           (168 MB at M = 65536, H = 1280) is never written (instead of LayerNorm-folded GEGLU GEMM -> Linear GEMM).
 *   x, out [M][320], M % 128 == 0;
 *   w1 [2H][320] = GEGLU.proj.weight * gamma (LayerNorm folded along K) with its rows interleaved in 16-row groups [u16 | g16 | ...]
 *      (the layout of lr_gemm_args.geglu == 1), b1 [2H] = proj.weight @ beta + proj.bias in the same row order (fp32);
 *   w2 [H / 64][320][64] = net[2].weight as 64-column pieces (each 40 KB contiguous) in the k-slot order of lr_xattn_args.wo (inside a
 *      piece: position 32 p + 8 f + i holds column 32 p + 16 (i >> 2) + 4 f + (i & 3)), b2 [320] = net[2].bias (fp32);  H % 64 == 0, H <= 2048;
 *   stats_out (optional) [M][2][2]: per-row (sum, sumsq) of the rounded output, one partial per column half (ln_parts = 2).
 * Other widths / ragged M: LR_E_UNSUPPORTED -- callers keep the two-GEMM path. */
typedef struct lr_ffn_args {
  const lr_half* x; lr_half* out;
  const lr_half* w1; const float* b1;
  const lr_half* w2; const float* b2;
  float* stats_out;
  int32_t M, C, H;
  float ln_eps;
  /* post_w != NULL: the Linear after the block runs in the same launch (SpatialTransformer.proj_out + `x + x_in`, attention.py:412-419):
   *   x3 = x + ff(LayerNorm(x));  out = x3 post_w^T + post_b + post_resid.
   * post_w [5][320][64] = proj_out.weight as 64-column pieces in k-slot order (like w2), post_b [320] fp32, post_resid [M][320];
   * gn_stats_out (optional) [M / 128][320][2] = per-channel (sum, sumsq) of the rounded output over each block of 128 rows, for the
   * GroupNorm that consumes `out` (lr_groupnorm_finalize with R = 128); stats_out is then the row statistics of `out`. */
  const lr_half* post_w; const float* post_b; const lr_half* post_resid; float* gn_stats_out;
  /* gn_group_out (optional, with post_w; ABI 20) [M / gn_hw][gn_hw / 128][32][2]: per-GROUP (sum, sumsq) of the rounded output over each
   * block of 128 rows (10 channels per group) -- lr_groupnorm_apply_n reads it directly (nchunks = gn_hw / 128), no finalize launch. */
  float* gn_group_out; int32_t gn_hw;
} lr_ffn_args;
int lr_ffn_block_f16(const lr_ffn_args* args, lr_stream_t s);

/* ---- row softmax of materialised logits (VAE AttnBlock: single head, d_head = C = 512) ---------------------------
 * replaces: `w_ = w_ * (int(c)**(-0.5)); w_ = softmax(w_, dim=2)` (ldm/modules/diffusionmodules/model.py:186-187) between
 *           the two bmm's (185, 192), which run through lr_gemm_conv_f16 (logits = q k^T with wt = k; out = p v with
 *           wt = v^T).  p[m][:] = softmax(scale * s[m][:]), fp32 math, fp16 in/out, N % 8 == 0, N <= 16384; p may alias s. */
int lr_softmax_rows_f16(const lr_half* s, lr_half* p, int M, int N, float scale, lr_stream_t st);

/* ---- re-arranged multi-view token gather / scatter ----------------------------------------------------------------
 * replaces: multiview_attention.py:436-448 (gather to [target, ref_0..]) and 452-462 (scatter back; target -> every
 *           canvas) for concat_target=True.  x [b*v][2*s*s][C] canvases (left = ref_i, right = target), seq [b][(v+1)*s*s][C].
 * (concat_target=False is a pure reshape and needs no kernel.) */
int lr_mv_gather(const lr_half* x, lr_half* seq, int b, int v, int s, int C, lr_stream_t st);
/* ---- row copies with index tables (ABI 23) -----------------------------------------------------------------------------------
 * replaces: the rearranges around the cross-view self-attention when the canvases of a sample live on different ranks
 *           (`rearrange(x, '(b v) hw c -> b (v hw) c')`, the concat_target slicing and the write-back, reference
 *           ldm/modules/multiview_attention.py:436-462): pack rows + per-row statistics into one message, unpack a received message
 *           into sequence order and into this rank's own rows, assemble the canvas -- up to 4 independent jobs in ONE launch.
 * job: for r < n_rows copy row_bytes bytes  src + src_idx[r] * src_pitch + src_off  ->  dst + dst_idx[r] * dst_pitch + dst_off
 *      (a NULL index table = the identity).  All byte counts / offsets / pitches / pointers multiples of 8. */
typedef struct lr_row_copy_job {
  const void* src; int64_t src_pitch, src_off;
  void* dst; int64_t dst_pitch, dst_off;
  int32_t row_bytes, n_rows;
  const int32_t* src_idx; const int32_t* dst_idx;
} lr_row_copy_job;
int lr_row_copy(const lr_row_copy_job* jobs, int n_jobs, lr_stream_t s);
int lr_mv_scatter(const lr_half* seq, lr_half* x, int b, int v, int s, int C, lr_stream_t st);

/* ---- fused classifier-free-guidance + DDIM update ----------------------------------------------------------------
 * replaces: p_sample_ddim's ~15 elementwise kernels (ddim.py:343,366,377-381): e = e_u + s (e_c - e_u);
 *           pred_x0 = (x - sqrt(1-a_t) e)/sqrt(a_t); x_prev = sqrt(a_prev) pred_x0 + sqrt(1-a_prev-sigma^2) e + sigma*noise.
 * x, x_prev, pred_x0, noise: fp32 [numel]; eps: fp16 or fp32 [2*numel], uncond half first (ddim.py:317-333).
 * noise may be NULL (sigma = 0).  Coefficients are host scalars (no device->host sync, cf. ddim.py:359-362). */
int lr_ddim_cfg_step(const float* x, const void* eps, int eps_is_f32, const float* noise, float* x_prev,
                     float* pred_x0, int64_t numel, float cfg_scale, float a_t, float a_prev, float sigma_t,
                     float sqrt_one_minus_at, lr_stream_t s);

/* ==== backward of the same operators (training with frozen weights: input gradients only) =========================
 * replaces: what torch.autograd derives for the reference's modules under `loss.backward()` (train_inpainting.py:141 ->
 *           LatentDiffusion.p_losses, ldm/models/diffusion/ddpm.py:900-935), recomputed per block by
 *           CheckpointFunction.backward (ldm/modules/diffusionmodules/util.py:133-151).
 * Input gradients of conv / linear are lr_gemm_conv_f16 itself on flipped / transposed weights (stride-2: `up = 2`). */

/* LayerNorm: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma.  x, dy, dx [M][C] fp16. */
int lr_layernorm_bwd(const lr_half* x, const lr_half* dy, const float* gamma, float eps, lr_half* dx, int M, int C,
                     lr_stream_t s);
/* GroupNorm(32)[+SiLU] over the virtual concat [x1 | x2]: fwd_partials = lr_groupnorm_stats of the same input;
 * bwd_partials: N*LR_GN_CHUNKS*64 floats of scratch; dy [N*HW][C1+C2]; dx1 [N*HW][C1], dx2 [N*HW][C2] (NULL if C2 = 0). */
int lr_groupnorm_bwd(const lr_half* x1, int C1, const lr_half* x2, int C2, const lr_half* dy, int N, int HW,
                     const float* fwd_partials, const float* gamma, const float* beta, float eps, int silu,
                     float* bwd_partials, lr_half* dx1, lr_half* dx2, lr_stream_t s);
/* ABI 25: the two backward passes with the gradient of the residual branch around the normalisation added in fp32 before the one
 * rounding -- `x + f(LayerNorm(x))` (attention.py:279-283) and `skip_connection(x) + f(GroupNorm(x))` (openaimodel.py:254-274,
 * attention.py:399-419) send two gradients to x; dres* (same layout as x*, NULL = none) is the one that does not pass through the
 * normalisation, so the fan-in sum costs no pass of its own.  fwd_chunks: chunk count of fwd_partials ([N][fwd_chunks][32][2]: the
 * per-group sums a producing GEMM wrote through lr_gemm_args.gn_group_out, or lr_groupnorm_finalize's [N][1][32][2]); 0 = the
 * lr_groupnorm_stats layout of lr_groupnorm_bwd. */
int lr_layernorm_bwd_res(const lr_half* x, const lr_half* dy, const lr_half* dres, const float* gamma, float eps, lr_half* dx,
                         int M, int C, lr_stream_t s);
int lr_groupnorm_bwd_res(const lr_half* x1, int C1, const lr_half* x2, int C2, const lr_half* dy, const lr_half* dres1,
                         const lr_half* dres2, int N, int HW, const float* fwd_partials, int fwd_chunks, const float* gamma,
                         const float* beta, float eps, int silu, float* bwd_partials, lr_half* dx1, lr_half* dx2, lr_stream_t s);
/* GEGLU: pre = projection + bias in the packed [u16 | g16] column layout (lr_gemm_conv_f16 with geglu = 0 on the packed
 * weights), dy [M][H] -> dpre [M][2H] (same layout): du = dy * gelu(g), dg = dy * u * gelu'(g). */
int lr_geglu_bwd(const lr_half* pre, const lr_half* dy, lr_half* dpre, int M, int H, lr_stream_t s);
/* training forward: out [M][H] = u * gelu(g) from the stored packed projection (kept for lr_geglu_bwd instead of recomputing). */
int lr_geglu_fwd(const lr_half* pre, lr_half* out, int M, int H, lr_stream_t s);
/* Attention: forward that also saves the log2-domain log-sum-exp (lse [B][heads][Nq] fp32), and the backward that
 * recomputes P from it (two deterministic kernels: dQ over key tiles; dK, dV over query tiles; plus D = rowsum(dO o O)).
 * kt / dot / ld_kt: ignored since ABI 25 (rounds 2-5 took lr_transpose_v_f16 copies of q / k / dout in qt / kt / dot; the kernels now gather
 * the k-major operands from the natural tiles with the LDS transpose read); the fields keep the struct layout, pass NULL / 0.
 * qt / ld_qt (ABI 26): query split of the dK / dV kernel for few keys against many queries (the 77-key cross-attention of the 64 x 128
 * level is ONE key block per (batch, head)): ld_qt = number of query slices (0 / 1 = none, <= 64), qt = fp32 workspace of
 * ld_qt * B * heads * ceil(Nkv / 128) * 2 * 128 * 64 floats (16-byte aligned); the slices' partial dK / dV are summed in slice order by a
 * second launch (deterministic).  NULL / 0 keep the single-slice kernel.
 * dsum: scratch [B][heads][Nq] fp32.  dq [B][Nq][lddq], dk / dv [B][Nkv][lddk | lddv], head h in columns h*64.. like q/k/v. */
int lr_attention_lse_f16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o,
                         int ldo, float* lse, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s);
typedef struct lr_attn_bwd_args {
  const lr_half* q; const lr_half* k; const lr_half* v; const lr_half* o; const lr_half* dout;
  const lr_half* qt; const lr_half* kt; const lr_half* dot;
  const float* lse; float* dsum;
  lr_half* dq; lr_half* dk; lr_half* dv;
  int32_t ldq, ldk, ldv, ldo, lddo, ld_qt, ld_kt, lddq, lddk, lddv;
  int32_t B, heads, Nq, Nkv;
  float scale;
} lr_attn_bwd_args;
int lr_attention_bwd_f16(const lr_attn_bwd_args* args, lr_stream_t s);

/* multi-view re-arrangement: gradient of lr_mv_gather (dseq -> dx: canvases other than 0 get zero in their right half) and
 * of lr_mv_scatter (dx -> dseq: the target slot sums the right halves of all canvases). */
int lr_mv_gather_bwd(const lr_half* dseq, lr_half* dx, int b, int v, int s, int C, lr_stream_t st);
int lr_mv_scatter_bwd(const lr_half* dx, lr_half* dseq, int b, int v, int s, int C, lr_stream_t st);
/* nearest-2x upsample: y[n][h][w][:] = sum of the four fine pixels of x [N][2H][2W][C]. */
int lr_sumpool2x2(const lr_half* x, lr_half* y, int N, int H, int W, int C, lr_stream_t s);

/* ---- bfloat16 twins: same signatures and semantics as the fp16 entry points above, every lr_half is bfloat16 bits -------- */
int lr_groupnorm_stats_bf16(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, float* partials,
    lr_stream_t s);
int lr_groupnorm_apply_bf16(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, const float*
    partials, const float* gamma, const float* beta, float eps, int silu, lr_half* y, lr_stream_t s);
int lr_groupnorm_apply_n_bf16(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, const float*
    partials, int nchunks, const float* gamma, const float* beta, float eps, int silu, lr_half* y, lr_stream_t s);
int lr_layernorm_bf16(const lr_half* x, const float* gamma, const float* beta, float eps, lr_half* y, int M, int C,
    lr_stream_t s);
int lr_layernorm_bwd_bf16(const lr_half* x, const lr_half* dy, const float* gamma, float eps, lr_half* dx, int M, int
    C, lr_stream_t s);
int lr_groupnorm_bwd_bf16(const lr_half* x1, int C1, const lr_half* x2, int C2, const lr_half* dy, int N, int HW,
    const float* fwd_partials, const float* gamma, const float* beta, float eps, int silu, float* bwd_partials,
    lr_half* dx1, lr_half* dx2, lr_stream_t s);
int lr_layernorm_bwd_res_bf16(const lr_half* x, const lr_half* dy, const lr_half* dres, const float* gamma, float eps, lr_half*
    dx, int M, int C, lr_stream_t s);
int lr_groupnorm_bwd_res_bf16(const lr_half* x1, int C1, const lr_half* x2, int C2, const lr_half* dy, const lr_half* dres1,
    const lr_half* dres2, int N, int HW, const float* fwd_partials, int fwd_chunks, const float* gamma, const float* beta,
    float eps, int silu, float* bwd_partials, lr_half* dx1, lr_half* dx2, lr_stream_t s);
int lr_nchw_f32_to_nhwc_bf16(const float* x1, int C1, const float* x2, int C2, lr_half* y, int Cpad, int N, int H, int
    W, lr_stream_t s);
int lr_nhwc_f16_to_nchw_bf16(const lr_half* y, int Cstride, int C, void* out, int out_is_f32, int N, int H, int W,
    lr_stream_t s);
int lr_timestep_embedding_bf16(const int64_t* t, int N, int dim, lr_half* out, lr_stream_t s);
int lr_linear_small_m_bf16(const lr_half* a, int lda, const lr_half* w, const float* bias, lr_half* out, int ldo, int
    M, int N, int K, int act_in, int act_out, lr_stream_t s);
int lr_mv_gather_bf16(const lr_half* x, lr_half* seq, int b, int v, int s, int C, lr_stream_t st);
int lr_mv_scatter_bf16(const lr_half* seq, lr_half* x, int b, int v, int s, int C, lr_stream_t st);
int lr_ddim_cfg_step_bf16(const float* x, const void* eps, int eps_is_f32, const float* noise, float* x_prev, float*
    pred_x0, int64_t numel, float cfg_scale, float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at,
    lr_stream_t s);
int lr_geglu_fwd_bf16(const lr_half* pre, lr_half* out, int M, int H, lr_stream_t s);
int lr_geglu_bwd_bf16(const lr_half* pre, const lr_half* dy, lr_half* dpre, int M, int H, lr_stream_t s);
int lr_sumpool2x2_bf16(const lr_half* x, lr_half* y, int N, int H, int W, int C, lr_stream_t s);
int lr_mv_gather_bwd_bf16(const lr_half* dseq, lr_half* dx, int b, int v, int s, int C, lr_stream_t st);
int lr_mv_scatter_bwd_bf16(const lr_half* dx, lr_half* dseq, int b, int v, int s, int C, lr_stream_t st);
int lr_gn_conv_out_bf16(const lr_half* x, int B, int H, int W, int C, const float* gpart, int chunks, const float* gamma, const float* beta,
                        float eps, const lr_half* w, int ldw, const float* bias, int Cout, lr_half* y, lr_stream_t s);
int lr_attention_bf16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o, int
    ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s);
int lr_attention_causal_bf16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half*
    o, int ldo, int B, int heads, int N, float scale, lr_stream_t s);
int lr_attention_lse_bf16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o,
    int ldo, float* lse, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s);
int lr_attention_vt_bf16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* vt, int ld_vt, lr_half*
    o, int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s);
int lr_transpose_v_bf16(const lr_half* v, int ldv, lr_half* vt, int ld_vt, int B, int heads, int Nkv, lr_stream_t s);
int lr_attention_bwd_bf16(const lr_attn_bwd_args* a, lr_stream_t s);
int lr_xattn_block_bf16(const lr_xattn_args* args, lr_stream_t s);
int lr_ffn_block_bf16(const lr_ffn_args* args, lr_stream_t s);
int lr_stin_block_bf16(const lr_stin_args* args, lr_stream_t s);
int lr_rowlin_bf16(const lr_rowlin_args* args, lr_stream_t s);
int lr_xattn_pack_vt_bf16(const lr_half* v, int ldv, lr_half* vt, int B, int heads, int Lc, lr_stream_t s);

/* ==== developer builds only (-DLR_DEV_VARIANTS, tools/build_variant.sh) ==================================================================
 * The product library (python -m leftrefill_amd.build) exports none of the following and reads no environment variable or other
 * process-global switch: every behaviour is selected by the arguments of a call.  A developer build additionally compiles the kernel
 * variants that were measured and not adopted (the 8-wave ping-pong attention kernel, the in-launch split-K reduce of
 * lr_gemm_args.splitk_mode = 1, alternative tile orders, the GroupNorm fold below; profiles/README.md has the measurements) and a small
 * name -> integer table that selects them; leftrefill_amd/_lib.py forwards LR_* environment variables to lr_dev_set. */
#ifdef LR_DEV_VARIANTS
int lr_dev_set(const char* name, int value);
int lr_dev_unset(const char* name);
/* ---- GroupNorm folded into the pointwise GEMM that consumes it (ABI 20) -------------------------------------------------
 * replaces: `x = self.norm(x)` of SpatialTransformer.forward (attention.py:399-404: Normalize = GroupNorm(32, eps 1e-6, affine), no
 *           activation) in front of proj_in (attention.py:405-408): per sample b the normalisation is a per-channel scale / shift
 *           a[b][c] = gamma[c] rstd[b][g(c)], t[b][c] = beta[c] - mean[b][g(c)] a[b][c], so
 *              proj_in(norm(x))[m][n] = sum_c (W[n][c] a[b][c]) x[m][c] + (bias[n] + sum_c W[n][c] t[b][c])
 *           -- the GEMM runs on the RAW x with per-sample weights (lr_gemm_args.wt_bstride) and the normalised tensor is never written.
 *   gpart [B][chunks][32][2]: per-group (sum, sumsq) partials of x from its producer (lr_gemm_args.gn_group_out), HW rows per sample;
 *   w [N][C] fp16, bias [N] fp32 or NULL  ->  w_out [B][N][C] = fp16(W a_b),  bias_out [B][N] = bias + W beta - rounded(W a_b) mean_b
 *   (the mean term uses the ROUNDED weights, so it cancels exactly what the matrix cores accumulate for a constant input).
 *   C % 32 == 0, C % 8 == 0, C <= 2048. */
int lr_gn_fold_weights_f16(const float* gpart, int chunks, int B, int HW, int C, const float* gamma, const float* beta, float eps,
                           const lr_half* w, const float* bias, int N, lr_half* w_out, float* bias_out, lr_stream_t s);
int lr_gn_fold_weights_bf16(const float* gpart, int chunks, int B, int HW, int C, const float* gamma, const float* beta, float eps,
                            const lr_half* w, const float* bias, int N, lr_half* w_out, float* bias_out, lr_stream_t s);
#endif /* LR_DEV_VARIANTS */

#ifdef __cplusplus
}
#endif
#endif /* LEFTREFILL_HIP_H */
