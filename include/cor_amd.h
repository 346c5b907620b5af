/* cor_amd.h — C ABI of libcor_amd.so: the MI355X (gfx950) kernels behind the CORE retrieval-time forward path.
 *
 * The reference (wangtong627/COR) is 100 % Python and has no FFI layer: its "plugin boundary" for this path is
 * the nn.Module API (lib/build_model.py:14-20 factory, lib/sam_with_sup_branch.py:57-104 forward). The Python
 * mirror of that API lives in cor_amd/lib/; every piece of arithmetic it needs is one of the entry points below,
 * each replacing the torch op sequence cited as "ref:". A reference maintainer binds them with ctypes
 * (see INTEGRATION.md).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless the name ends in _host; no allocation happens inside;
 *  - `stream` is a hipStream_t passed as void*; kernels are only enqueued, never synchronised;
 *  - return 0 on success, a positive hipError_t for a launch failure, COR_EINVAL (-1) for bad arguments,
 *    COR_ENOSUPPORT (-2) for a shape/dtype combination that has no kernel (never a silent fallback);
 *  - activations are token-major row matrices [rows, C] (channels-last); dtypes are COR_F32 or COR_BF16;
 *    parameters that stay in fp32 (bias, LayerNorm affine, rel-pos tables) are `const float*`.
 */
#ifndef COR_AMD_H
#define COR_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define COR_EINVAL (-1)
#define COR_ENOSUPPORT (-2)
/* Work order, OR-ed into cor_gemm's `cfg`, cor_layernorm's `act` and cor_sam_attention's `variant` (results are identical):
 * walk the output tiles / rows / windows from the LAST to the first. A kernel chain over activations larger than the 256 MB
 * Infinity Cache runs faster when each kernel starts where its producer finished (the producer's last ~200 MB are still cached):
 * the engine alternates the direction along the encoder's chain (measured +2-3 % end to end on SAM-B at batch 32). */
#define COR_ORDER_REVERSE (1 << 30)
#define COR_TOPK_FORCE_LISTS 1 /* cor_similarity_topk flags: per-lane sorted-list kernels only (no threshold-and-append) */
#define COR_TOPK_NO_FALLBACK 2  /* ... : no device-side fallback after a candidate overflow: such queries return index -2 */
#define COR_TOPK_FORCE_GLOBAL_THRESHOLD 8 /* ... : never the two-launch local-threshold path of small shards (A/B partner, tests) */
#define COR_TOPK_WAVE_FINAL 16 /* ... : global-threshold pipeline with the one-wave-per-query selection kernel fed from the records (slower A/B partner, tests) */

enum { COR_F32 = 0, COR_BF16 = 1, COR_F16 = 2 /* gallery storage only */ };
enum { COR_ACT_NONE = 0, COR_ACT_GELU_ERF = 1, COR_ACT_RELU = 2, COR_ACT_SIGMOID = 3, COR_ACT_GELU_TANH = 4 };

int cor_version(void);

/* ---- dense linear algebra -------------------------------------------------------------------------------- */

/* C[M,N] = residual + col_scale * act(A[M,K] . W[N,K]^T + bias)          (nn.Linear / 1x1 conv / patch conv)
 * A, W share `ab_dtype`; bias/col_scale [N] fp32 or NULL; residual fp32 [*, ldr] or NULL, read at row
 * (m % res_row_mod) when res_row_mod > 0 (broadcast of pos_embed over the batch).
 * MFMA path: v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact fp32).
 * ref: every nn.Linear on the path, e.g. lib/sam_model/image_encoder.py:229,239 ; common.py:25-26 ;
 *      PatchEmbed conv image_encoder.py:386-394 (as GEMM over patches) ; neck 1x1 conv :87-92. */
int cor_gemm(const void* A, long lda, const void* W, long ldw, int ab_dtype,
             void* C, long ldc, int c_dtype, int M, int N, int K,
             const float* bias, int act, const float* col_scale,
             const float* residual, long ldr, int res_row_mod, int cfg, void* stream);
/* `cfg` is a PER-CALL kernel choice (no process-global state; safe from several threads / streams): 0 = automatic
 * (13 for bf16 operands from 200 output tiles of 256x256 up, else 2, or 1 when K has a ragged tail); 1: 128x128 register-staged
 * (any K); 2: 128x128 LDS-DMA; 3 / 4: 128x64 / 64x64; 9: 256x128, three LDS buffers; 13: persistent 256x256 ping-pong kernel
 * (bf16 operands; falls back to 2 / 1 where it does not apply); optionally | COR_ORDER_REVERSE. Anything else is COR_EINVAL:
 * the timing probes of the development builds (result-destroying ablation bits, 16x16 MFMA form) are not in this library
 * (they build into tools/probes/libcor_probes.so with -DCOR_PROBES). */

/* y[r,:] = LayerNorm(x[r,:]) * w + b over the last dim, biased variance.
 * ref: nn.LayerNorm (image_encoder.py:169,183), LayerNorm2d common.py:31-43 and mask_adapter.py:226-251 (on
 *      channels-last rows), transformer.py norm1..4. */
int cor_layernorm(const void* x, int x_dtype, void* y, int y_dtype, const float* w, const float* b,
                  int rows, int C, float eps, int act, void* stream);

/* ---- attention -------------------------------------------------------------------------------------------- */

/* out[b,t,h,:] = softmax_k(scale * q[b,t,h,:].k[b,k,h,:]) v ; element (b,t,h,c) of X lives at
 * X + b*x_sb + t*x_st + h*hd + c (strides in elements). hd in {16,32,64,72,80}; bf16 with hd 64 / 72 / 80 and >= 64 queries and
 * keys runs the MFMA flash kernel (72 = SigLIP SO400M/14, the factory's default tower), everything else the row-per-lane kernel.
 * ref: lib/sam_model/transformer.py:218-240 (decoder Attention), SigLIP towers' MHA.
 * PRECONDITION (cor_attention and cor_sam_attention, bf16 MFMA kernels): q, k, v are FINITE. The softmax of those kernels is compiled
 * with -fno-honor-nans (no NaN arises inside: the running maximum starts at -inf, every key tile holds a real key; -inf marks padded
 * key slots and is honoured), so a NaN / Inf operand from upstream (e.g. a bf16 overflow) gives unspecified values in the rows of the
 * (sample, head) that contain it instead of a propagated NaN; other (sample, head) pairs are unaffected (tests/test_gpu_parity.py::
 * test_attention_nonfinite_operands_stay_inside_their_head). The fp32 row-per-lane kernels propagate NaN as IEEE arithmetic does. */
int cor_attention(const void* q, long q_sb, long q_st, const void* k, long k_sb, long k_st,
                  const void* v, long v_sb, long v_st, int dtype,
                  void* out, long o_sb, long o_st, int out_dtype,
                  int B, int H, int Tq, int Tk, int hd, float scale, void* stream);

/* SAM ViTDet attention on the fused qkv activation [B*grid*grid, 3*H*hd] (q|k|v, head-major inside each;
 * hd = 64 for SAM-B/L and 80 for SAM-H (bf16: MFMA flash kernels), 16/32 for reduced test models (row-per-lane kernel)):
 * logits = (q*hd^-0.5).k + q.Rh[qh-kh+S-1] + q.Rw[qw-kw+S-1] (rel-pos from the UNSCALED q).
 * window == 0: global attention over the grid (S = grid). window > 0: non-overlapping window x window tiles of the
 * grid zero-padded bottom/right to a multiple of `window` AFTER norm1, so a padded token's q/k/v equal the qkv
 * bias (`pad_row`, [3*H*hd] in `dtype`); padding/partition/unpartition are folded into addressing.
 * out [B*grid*grid, H*hd]. rel_h / rel_w: [2S-1, hd] fp32.
 * ref: lib/sam_model/image_encoder.py:225-241 (Attention), :244-290 (window partition), :293-362 (rel-pos). */
int cor_sam_attention(const void* qkv, int dtype, void* out, int out_dtype, const void* pad_row,
                      const float* rel_h, const float* rel_w, int B, int H, int hd, int grid, int window, float q_prescale, int variant,
                      void* stream);
/* `q_prescale`: the factor the caller has ALREADY folded into the q third of qkv (and of pad_row): 1 = raw q (always accepted);
 * 0.125 * log2(e) (hd 64: scale * log2 e) lets the bf16 MFMA kernels run the score product directly in the log2 domain with
 * no per-score multiply - the engine folds it into the q rows of the qkv weight / bias at pack time, so it costs nothing at
 * run time and q is rounded to bf16 once. The rel-pos terms are rescaled accordingly inside (they use the unscaled q).
 * `variant` is a PER-CALL kernel choice for the bf16 MFMA path: 0 = default kernels (global: flash_global_pipe, software-
 * pipelined over key tiles, with a pre-scaled q the column bias is the score MFMA's start value; windowed: win_attn, one 7-wave
 * block per (window, head)); 1 = the round-1 chain forms of the same arithmetic (flash_fwd<1> / flash_fwd<2>), kept as the
 * in-process A/B and parity partners (tests, tools/attn_bench.py); 2 = flash_global_pipe with the bias as one fma per score
 * (round 4's default; A/B partner); 4 = flash_global_w64 (global, pre-scaled q only: 64 queries per wave at one wave per SIMD;
 * parity-tested, measured slower: DESIGN 3.2); 2 and 4 select the default kernel where they do not apply (windowed, raw q);
 * optionally | COR_ORDER_REVERSE. Anything else is COR_EINVAL (the timing probe of the global kernel, which writes cycle
 * counters instead of outputs, exists in -DCOR_PROBES builds only: tools/probes/libcor_probes.so, tools/attn_stamps.py). */

/* Which kernel family cor_attention (sam_window = -1) / cor_sam_attention (0 = global, > 0 = windowed) runs for 16-byte-aligned
 * operands of this shape: bf16 with head_dim 64 / 72 / 80 is on the matrix cores. Pure function (no launch). */
enum { COR_KERNEL_ROWLANE = 1, COR_KERNEL_FEWQ = 2, COR_KERNEL_FLASH_MFMA = 3, COR_KERNEL_FLASH_PIPELINED = 4, COR_KERNEL_WINDOW_BLOCK = 5 };
int cor_attention_kernel_id(int dtype, int hd, int Tq, int Tk, int sam_window, int grid);

/* ---- data movement / elementwise --------------------------------------------------------------------------- */

/* Non-overlapping p x p patches of NCHW fp32 images -> rows [B*(H/p)*(W/p), Kpad] (k = c*p*p + dy*p + dx, zero
 * padded to Kpad), i.e. the A operand of the patch-embedding GEMM; floor division: a ragged right/bottom border is
 * dropped as a strided conv does (SO400M/14 at 384 px: 27x27 patches). ref: image_encoder.py:386-394. */
int cor_patchify(const float* img, void* out, int out_dtype, int B, int C, int H, int W, int p, int Kpad, void* stream);

/* 3x3, pad 1 im2col of channels-last tokens [B,H,W,C] -> [B*H*W, 9*C] (k = (ky*3+kx)*C + c).
 * ref: neck conv image_encoder.py:94-100. */
int cor_im2col3x3(const void* x, int dtype, void* out, int B, int H, int W, int C, void* stream);

/* out = a + b[(i) % b_period] elementwise over n elements (b_period = n for a plain add). */
int cor_add(const void* a, int a_dtype, const void* b, int b_dtype, void* out, int out_dtype, long n, long b_period, void* stream);

/* dtype cast / strided row copy: out[r, :C] = in[r, :C]; ld_in == 0 broadcasts one source row. */
int cor_copy_rows(const void* in, long ld_in, int in_dtype, void* out, long ld_out, int out_dtype, int rows, int C, void* stream);

/* [rows, C] channels-last tokens -> NCHW [B, C, HW] (fp32 out) and back. */
int cor_tokens_to_nchw(const void* x, int dtype, float* out, int B, int HW, int C, void* stream);
int cor_nchw_to_tokens(const float* x, void* out, int out_dtype, int B, int HW, int C, void* stream);

/* y = x / max(||x||_2, eps) per row. ref: F.normalize (support_branch.py:85, cir_feature_fuse.py:58, siglip_openclip.py:56). */
int cor_l2norm_rows(const void* x, int x_dtype, void* y, int y_dtype, int rows, int C, float eps, void* stream);

/* out[r,:] = table[ids[r],:] + pos[r % ctx,:]. An id outside [0, vocab) (nn.Embedding raises; a wrong tokenizer) makes the
 * row NaN: never a fault, never plausible garbage. ref: open_clip TextTransformer embedding (siglip_openclip.py:53). */
int cor_embed_tokens(const long long* ids, const float* table, const float* pos, float* out, int rows, int ctx, int D, int vocab, void* stream);

/* ---- support branch (mask adapter, fusion) ------------------------------------------------------------------ */

/* Bilinear resize, align_corners=False, no antialias, fp32 NCHW planes. ref: F.interpolate at mask_adapter.py:20,58,158. */
int cor_bilinear(const float* x, float* out, int planes, int H, int W, int OH, int OW, int clamp01, void* stream);

/* Direct 3x3 stride-2 pad-1 convolution for tiny channel counts, NCHW fp32 in, channels-last fp32 out
 * [B, OH, OW, Cout]. ref: mask_adapter.py:128-137 (mask_downscaling convs). */
int cor_conv3x3s2_small(const float* x, int x_channels_last, const float* w, const float* bias, float* out,
                        int B, int Cin, int Cout, int H, int W, void* stream);

/* Depthwise 7x7 pad 3 on channels-last tokens [B,H,W,C] fp32 -> out (dtype). w_t [49,C] (tap-major transpose of
 * the reference's [C,1,7,7] weight, made once at load), bias [C]. ref: mask_adapter.py:197-199 (ConvNextBlock.dwconv). */
int cor_dwconv7x7(const float* x, const float* w_t, const float* bias, void* out, int out_dtype, int B, int H, int W, int C, void* stream);

/* pooled[b,:] = mean_m sum_p softmax_p(logsigmoid(maps[b,p,m])) feat[b,p,:]   (maps [B,P,M] fp32, feat [B,P,D] fp32)
 * ref: mask_adapter.py:68-79. */
int cor_adapter_pool(const float* maps, const float* feat, float* out, int B, int P, int M, int D, void* stream);

/* pooled[b,:] = sum_p feat[b,p,:]*mask[b,p] / (sum_p mask[b,p] + 1e-8), optional clamp of mask to [0,1] and
 * L2-normalisation. ref: mask_adapter.py:13-25 (MaskedPooling), utils/loss_func.py:35-56 (region embedding;
 * feat there is NCHW: pass feat_nchw=1). */
int cor_masked_pool(const float* feat, int feat_nchw, const float* mask, float* out, int B, int P, int D,
                    int clamp01, int l2norm, void* stream);

/* i' = aI*i ; t' = aT*t  (gates already sigmoid-ed), written into cat [N,2D]; and the final
 * out = normalize(dyn*i' + (1-dyn)*t'). ref: cir_feature_fuse.py:51-58. */
int cor_fuse_gate(const float* img, const float* txt, const float* aI, const float* aT, float* cat, int N, int D, void* stream);
int cor_fuse_mix(const float* cat, const float* dyn, float* out, int N, int D, void* stream);

/* ---- prompt encoder / mask decoder ------------------------------------------------------------------------- */

/* Random-Fourier dense PE as tokens [size*size, 2F] fp32: cat(sin,cos)(2*pi*((2*xy-1) @ G)), G [2,F].
 * ref: my_prompt_encoder.py:62-71,191-211. */
int cor_dense_pe(const float* gauss, float* out, int size, int F, void* stream);

/* Pixel-shuffle of a ConvTranspose2d(k=2,s=2) computed as GEMM: y[B*H*W, 4*Cout] (col = (dy*2+dx)*Cout + co, the
 * order the weight is packed in at load) -> out[B, 2H, 2W, Cout] (+ bias), then optional LayerNorm over Cout
 * (ln_w/ln_b both NULL to skip) and activation.
 * ref: mask_decoder.py:54-60 (output_upscaling). */
int cor_upscale_shuffle(const void* y, int y_dtype, const float* bias, const float* ln_w, const float* ln_b, float eps,
                        int act, void* out, int out_dtype, int B, int H, int W, int Cout, void* stream);

/* Fused second upscaling + hypernetwork product:
 * masks[b,k,2y+dy,2x+dx] = sum_co hyper[b,k,co] * gelu( sum_ci x[b,y,x,ci] * w[ci,co,dy,dx] + bias[co] )
 * x [B,H,W,Cin] (dtype), w [Cin,Cout,2,2] fp32, hyper [B,Kmask,Cout] fp32 -> masks [B,Kmask,2H,2W] fp32.
 * ref: mask_decoder.py:58-59,133-137. */
int cor_upscale_hyper(const void* x, int dtype, const float* w, const float* bias, const float* hyper, long hyper_bs,
                      float* masks, int B, int H, int W, int Cin, int Cout, int Kmask, void* stream);

/* best[b] = argmax_k iou[b, k_off : k_off+Ksel] (first maximum wins, like torch.argmax);
 * hyper_sel[b,:] = hyper[b, k_off+best[b], :]  ([B,Kall,C] -> [B,C]). Selecting the hypernetwork row BEFORE the
 * upscaling means only the chosen mask channel is ever computed / written.
 * ref: sam_with_sup_branch.py:96-100 ; mask_decoder.py:97-102. */
int cor_iou_select(const float* iou, const float* hyper, int B, int Kall, int k_off, int Ksel, int C, long long* best,
                   float* hyper_sel, void* stream);

/* The decoder's five output MLPs in one launch: hyper-network MLP i (i = 0..3) on mask token 1+i -> hyper[b, i, 0:32], IoU head on token 0
 * -> iou[b, 0:4]; each Linear(256,256) ReLU Linear(256,256) ReLU Linear(256, 32 | 4). hs [B,6,256] in `dtype` (COR_F32 | COR_BF16, the
 * weights' type too); w01 [5,2,256,256] (MLP, layer, out, in; MLP 4 = IoU head), b01 [5,2,256] fp32, w2 [132,256] (rows 32 i + c of
 * hyper MLP i, rows 128..131 of the IoU head), b2 [132] fp32. fp32 accumulation over k in order; hidden activations rounded to `dtype`.
 * ref: mask_decoder.py:123-140 (output_hypernetworks_mlps, iou_prediction_head), :147-167 (MLP). */
int cor_decoder_heads(const void* hs, const void* w01, const float* b01, const void* w2, const float* b2, int dtype, float* hyper,
                      float* iou, int B, void* stream);

/* ---- inference harness (mask post-processing, metrics) ---------------------------------------------------------------- */

/* out = (sigmoid(x) - min) / (max - min + 1e-8), min/max per sample. ref: utils/vailder.py:426-430. */
int cor_mask_prob_minmax(const float* logits, float* out, int B, int HW, void* stream);

/* Bilinear resize (half-pixel centres, edge clamp = cv2.INTER_LINEAR) of [B,H,W] probabilities to [B,OH,OW], then
 * (> threshold) ? 255 : 0 as uint8. ref: utils/vailder.py:459-473. */
int cor_resize_binarize(const float* prob, unsigned char* out, int B, int H, int W, int OH, int OW, float threshold, void* stream);

/* The same resize, then (v * 255) truncated to uint8: the soft (grayscale) mask of save_soft_pred_masks.
 * ref: utils/vailder.py:513-656 (:615-621: cv2.resize INTER_LINEAR, (pred * 255).astype(np.uint8)). */
int cor_resize_gray(const float* prob, unsigned char* out, int B, int H, int W, int OH, int OW, void* stream);

/* out[b] = {dice, mae, iou, mdice, miou} of a soft prediction against the ground truth, [B,HW] each.
 * ref: utils/trainer_v3_g.py:381-443 (compute_dice / compute_mae / compute_iou / compute_mdice / compute_miou). */
int cor_mask_metrics(const float* pred, const float* gt, float* out, int B, int HW, float smooth, void* stream);

/* ---- input pre-processing (device side of utils/dataloader.py:266-293) ------------------------------------------------ */

/* Pillow-BILINEAR resize of uint8 images, as torchvision.transforms.Resize performs it on a PIL image (Pillow
 * src/libImaging/Resample.c, 8 bits per channel, antialiased on down-scaling), bit-exact. The caller supplies Pillow's
 * fixed-point tables for one axis: bounds int32[out,2] = (first input index, tap count), kk int32[out,ksize] =
 * int(0.5 + w * 2^22) (cor_amd/preprocess.py:resample_tables). Pass 1: rows. in u8[H,W,C] -> out u8[H,OW,C]; C in {1,3}. */
int cor_resample_rows_u8(const unsigned char* in, unsigned char* out, const int* bounds, const int* kk, int ksize, int H, int W, int C,
                         int OW, void* stream);

/* Pass 2: columns of the uint8 image pass 1 produced, then ToTensor (+ Normalize): in u8[H,W,C] -> out_u8 u8[OH,W,C] (may be
 * NULL) and out_f32 f32[C,OH,W] (may be NULL) = v/255, or (v/255 - mean[c]) / std[c] when mean/std are given (both or
 * neither), in IEEE float32 exactly as torchvision's ToTensor / Normalize compute it. */
int cor_resample_cols_u8(const unsigned char* in, float* out_f32, unsigned char* out_u8, const int* bounds, const int* kk, int ksize, int H,
                         int W, int C, int OH, const float* mean, const float* stdv, void* stream);

/* ---- retrieval ------------------------------------------------------------------------------------------------ */

/* Per query b: the top-k rows g of the gallery shard by score = q[b,:].G[g,:] (fp32 accumulate), ordered by
 * (score desc, index asc); indices are returned as global ids (g + g_offset).
 * Q [Bq,C] fp32, G [Ng,C] in g_dtype (COR_F32 exact chain / COR_BF16 / COR_F16), C <= 256 and C % 16 == 0, k <= 32;
 * missing entries (Ng < k) come back as score -inf, index -1. workspace >= cor_topk_workspace_bytes(Bq,Ng,k).
 * The reference has no gallery/top-k code; the definition follows utils/loss_func.py:84 (cosine of unit vectors).
 * Scores are DEFINED as the fp32 fmaf chain of oracle/c/sim_chain.c (16-bit rows widened exactly), for every gallery dtype: the
 * result is bit-identical to that CPU chain, scores and indices. fp32 shards run the chain kernel. 16-bit shards (C = 256) scan on the
 * matrix cores and re-score a short list with the exact chain: SMALL shards (<= 32 slices of 256..1024 rows per query in one round of
 * blocks, k <= 16: the 8-GPU shard shapes, 256..512 queries x 12.5k rows) in TWO launches with block-local thresholds (sim_block_scan:
 * every block bounds its queries' k-th best score from the maxima of 32 disjoint row classes of its own slice and appends what passes;
 * sim_final_wave: one wave per query selects, re-scores, ranks); everything else by threshold-and-append with a GLOBAL threshold (a strided
 * sample pass bounds each query's k-th best score by 32 super-group maxima, the full MFMA pass ranks them in its prologue and appends the
 * rare score records above the bound, the final pass re-scores the short list: four launches). If a query's candidate list overflows (pathological score distributions, e.g. hundreds of identical rows)
 * it is ranked by an exact brute-force chain pass ON THE DEVICE (no host round trip). `flags` (per call, no process-global state):
 * 0 = default; COR_TOPK_FORCE_LISTS = per-lane sorted-list kernels only; COR_TOPK_NO_FALLBACK = report an overflow as index -2 in every
 * slot of the query instead of falling back (tests); COR_TOPK_FORCE_GLOBAL_THRESHOLD / COR_TOPK_WAVE_FINAL = A/B partners (tests). */
long cor_topk_workspace_bytes(int Bq, int Ng, int k);
int cor_similarity_topk(const float* Q, const void* G, int g_dtype, int Bq, int Ng, int C, int k, long long g_offset,
                        float* out_scores, long long* out_idx, void* workspace, int flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COR_AMD_H */
