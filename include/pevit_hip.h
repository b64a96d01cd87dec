/* pevit_hip.h -- C ABI of the MI355X (gfx950) engine for parameter-efficient fine-tuning of
 * CLIP vision transformers.
 *
 * The reference (eric-ai-lab/PEViT) has no FFI/plugin layer: its hot path sits behind plain
 * PyTorch module methods.  This header declares the drop-in boundary for that path; every
 * entry point names the reference interface it replaces (paths relative to
 * vision_benchmark/evaluation/ of the reference):
 *
 *   pevit_transformer_forward / _backward   Transformer.forward            model.py:1013-1014
 *                                           = 12 x ResidualAttentionBlock.forward  model.py:972-975
 *                                           incl. MultiheadAttention.multi_head_attention_forward
 *                                           (model.py:612-834) and adapter_forward (model.py:563-584;
 *                                           lora_model.py:490-514), and their autograd backward
 *   pevit_visual_forward / _backward        VisionTransformer.forward      model.py:1034-1051
 *                                           (= CLIP.encode_image, model.py:1151-1152)
 *   pevit_head_forward_backward             Classifier.forward tail + CrossEntropyLoss
 *                                           kadaptation_clip.py:128-132,176-185,276,351-352
 *   pevit_sgd_step                          optimizer.step()               kadaptation_clip.py:353,
 *                                           optim/build.py:120-127
 *   pevit_load_block / pevit_load_stem      build_model's load_state_dict  model.py:1247-1250
 *
 * Conventions: extern "C"; every function returns 0 on success and a negative value on error
 * (pevit_last_error() gives the message); nothing throws; no device-memory allocation by any context entry point
 * (the optional "side_stream" knob creates one stream + two events on first use, profiling its
 * events in pevit_profile_begin; the CONTEXT-FREE test entry point pevit_op_gemm allocates a stream-K workspace of its own
 * on first use, once per process).  One process drives one device: the kernel attributes and the CU count are cached
 * per process -- all device memory (weight arena, workspace, parameter and gradient
 * buffers) is owned by the caller and only borrowed; all work is enqueued asynchronously on
 * the caller's hipStream_t (passed as void*); contexts are independent of each other (no
 * mutable process-wide state on the hot path; tuning knobs live in the context) but one context
 * is not thread-safe, matching the reference's single-threaded caller.
 *
 * Row order: the reference keeps activations sequence-first, (N, B, E).  Entry points with
 * "_nbe" arguments take and return exactly that layout; internally rows are batch-major.
 */
#ifndef PEVIT_HIP_H
#define PEVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pevit_ctx pevit_ctx;

enum pevit_method {
    PEVIT_KADAPTATION = 0, /* model.py          : Kronecker rank-32 delta on q and v + shared bias b */
    PEVIT_LORA = 1,        /* lora_model.py     : low-rank A.B on q and v, scale alpha/r            */
    PEVIT_ADAPTER = 2,     /* adapter_model.py  : bottleneck adapter after the MLP                  */
    PEVIT_COMPACTER = 3,   /* compacter_model.py: PHM (n=4) bottleneck adapter after the MLP        */
    PEVIT_NONE = 4         /* frozen tower, no adapter (linear probe)                               */
};

enum pevit_weight_format {
    PEVIT_W_BF16 = 0,     /* frozen block weights rounded to bf16 (reference: fp32 parameters, model.py:1247-1250)    */
    PEVIT_W_FP8_E4M3 = 1, /* OCP e4m3 codes + one power-of-two f32 scale per output channel, packed by
                             pevit_load_block; activations stay bf16, accumulation f32 (BASELINE config 5).  Bit-identical
                             to PEVIT_W_BF16 run on the de-quantised weights.  KAdaptation, LoRA and the frozen tower.  */
    PEVIT_W_FP8_ACT = 3,  /* opt-in, NOT bit-compatible with the bf16 path: PEVIT_W_FP8_E4M3 weights, and in the FORWARD pass the
                             A operands of the four frozen products of a block (LayerNorm outputs, attention output, gelu(h))
                             are e4m3 codes too (unscaled: activations are O(1); saturating at +-448), written by their
                             producers, so that these products run fp8 x fp8 on CDNA4's MX-scaled matrix instruction
                             (v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales: twice the bf16 MFMA rate, half the
                             operand bytes).  The backward pass is that of PEVIT_W_FP8_E4M3 (bf16 activations, saved in
                             bf16).  Deviation from the bf16-on-de-quantised run: profiles/r03_fp8_act.md.                */
    PEVIT_W_F32_VERIFY = 2 /* f32-class VERIFICATION mode, not a production format: weights and every activation kept in
                             f32, the matrix-core contractions run as plain f32 kernels (csrc/verify.hip) inside the same
                             launch sequences, layouts and index arithmetic.  Used by the parity tests to assert the stated
                             gates against the reference's fixtures without bf16 operand rounding; ~20x slower; workspace
                             and arena are twice the size.  All methods.                                                */
};

typedef struct pevit_dims {
    int32_t width;       /* E: visual.conv1.weight.shape[0]                 (model.py:1214) */
    int32_t layers;      /* L                                               (model.py:1215) */
    int32_t patch;       /* P                                               (model.py:1216) */
    int32_t resolution;  /* R = P * grid                                    (model.py:1218) */
    int32_t out_dim;     /* D: visual.proj.shape[1]                                         */
    int32_t method;      /* enum pevit_method                                               */
    int32_t lora_rank;   /* r (reference hard-codes 4: lora_model.py:461)                   */
    int32_t num_classes; /* C: DATASET.NUM_CLASSES          (kadaptation_clip.py:125)       */
    int32_t weight_format; /* enum pevit_weight_format                                      */
} pevit_dims;

const char* pevit_last_error(void);
int pevit_version(void);

int pevit_ctx_create(const pevit_dims* dims, pevit_ctx** out);
void pevit_ctx_destroy(pevit_ctx* ctx);

/* sizes the caller must provide */
size_t pevit_arena_bytes(const pevit_ctx* ctx);                 /* frozen weights, engine layout */
size_t pevit_workspace_bytes(const pevit_ctx* ctx, int batch);  /* activations + scratch          */
/* number of f32 trainable parameters: tower adapters (reference named_parameters() order,
 * README.md:84-87 counts) followed by the head weight (C x D) and bias (C) */
size_t pevit_num_tower_params(const pevit_ctx* ctx);
size_t pevit_num_params(const pevit_ctx* ctx);
/* mask[i] = 1 if parameter i ever receives a gradient; 0 for the v_proj_adapter1_* tensors
 * of KAdaptation, which the reference never uses (model.py:580) and whose .grad stays None */
int pevit_param_grad_mask(const pevit_ctx* ctx, unsigned char* host_mask, size_t n);

int pevit_bind(pevit_ctx* ctx, void* arena, size_t arena_bytes, void* workspace, size_t workspace_bytes,
               int max_batch);
/* flat f32 parameter / gradient / momentum buffers (device, borrowed); gradients are
 * accumulated into `grads`.  `grad_mask` is the device copy of pevit_param_grad_mask (may be
 * NULL = every parameter is updated). */
int pevit_set_params(pevit_ctx* ctx, float* params, float* grads, float* momentum,
                     const unsigned char* grad_mask);

/* frozen weights: device f32 pointers in the OpenAI state-dict layout.  With fp8 weights pevit_load_block uses
 * the bound workspace as packing scratch: call it after pevit_bind, on the stream the context trains on, and not between a
 * forward and its backward (the saved activations are invalidated: the next backward without a new forward fails). */
int pevit_load_block(pevit_ctx* ctx, void* stream, int layer, const float* in_proj_weight,
                     const float* in_proj_bias, const float* out_proj_weight, const float* out_proj_bias,
                     const float* ln_1_weight, const float* ln_1_bias, const float* c_fc_weight,
                     const float* c_fc_bias, const float* c_proj_weight, const float* c_proj_bias,
                     const float* ln_2_weight, const float* ln_2_bias);
int pevit_load_stem(pevit_ctx* ctx, void* stream, const float* conv1_weight, const float* class_embedding,
                    const float* positional_embedding, const float* ln_pre_weight, const float* ln_pre_bias,
                    const float* ln_post_weight, const float* ln_post_bias, const float* proj);
/* Compacter's frozen shared phm_rule (4,4,4) (compacter_model.py:511-519) */
int pevit_load_phm_rule(pevit_ctx* ctx, void* stream, const float* phm_rule);

/* ---- the hot path ------------------------------------------------------------------ */
int pevit_transformer_forward(pevit_ctx* ctx, void* stream, const float* x_nbe, float* y_nbe, int batch,
                              int save_for_backward);
int pevit_transformer_backward(pevit_ctx* ctx, void* stream, const float* dy_nbe, float* dx_nbe_or_null,
                               int batch);
/* blocks [l_lo, l_hi) only -- ResidualAttentionBlock.forward (model.py:972-975) for l_hi = l_lo + 1, MultiheadAttention.forward
 * (model.py:837) being the attention half of it; pevit_transformer_* = the range [0, layers).  Every block keeps its own saved
 * activations: the blocks may be walked one call at a time and differentiated in reverse order (a walk starts at block 0). */
int pevit_blocks_forward(pevit_ctx* ctx, void* stream, const float* x_nbe, float* y_nbe, int batch, int save_for_backward,
                         int l_lo, int l_hi);
int pevit_blocks_backward(pevit_ctx* ctx, void* stream, const float* dy_nbe, float* dx_nbe_or_null, int batch, int l_lo,
                          int l_hi);
int pevit_visual_forward(pevit_ctx* ctx, void* stream, const float* images, float* feat, int batch,
                         int save_for_backward);
int pevit_visual_backward(pevit_ctx* ctx, void* stream, const float* dfeat, int batch);
/* the same backward cut at a block boundary: blocks layer_hi-1 .. layer_lo, whose adapter gradients are complete in
 * the flat buffer when the call's work has run (data parallelism all-reduces them while the next part runs).  The part
 * with layer_hi == layers takes dfeat; parts must be issued top-down and cover [0, layers) exactly once.  The shared
 * phm_rule gradients (KAdaptation) are complete after the part with layer_lo == 0. */
int pevit_visual_backward_part(pevit_ctx* ctx, void* stream, const float* dfeat_or_null, int batch, int layer_hi,
                               int layer_lo);
/* float offset of block `layer`'s parameters in the flat buffer (layer == layers: the head weight) */
size_t pevit_param_layer_offset(const pevit_ctx* ctx, int layer);
/* BatchNorm1d(D, affine=False) -> Linear(D, C) -> mean cross-entropy, forward + backward.
 * bn_training: batch statistics + running-stat update (momentum 0.1); else running stats.
 * Writes logits (B x C), loss (1), dfeat (B x D); accumulates head grads into the flat buffer. */
int pevit_head_forward_backward(pevit_ctx* ctx, void* stream, const float* feat, const int64_t* labels,
                                float* running_mean, float* running_var, int bn_training, float* logits,
                                float* loss, float* dfeat_or_null, int batch);
int pevit_zero_grads(pevit_ctx* ctx, void* stream);
/* flags: bit 0 = first step of the run (momentum buffer := gradient, like torch), bit 1 = Nesterov momentum */
int pevit_sgd_step(pevit_ctx* ctx, void* stream, float lr, float momentum, float weight_decay,
                   float grad_scale, int flags);
/* uint8 pixels (B,3,R,R) instead of preprocessed f32: the dataset transforms of the reference (ToTensor + Normalize with
 * INPUT.MEAN / INPUT.STD: feature.py:537-542, resources/model/vitb32_CLIP.yaml:4-6), x = (u8 / 255 - mean[c]) / std[c], run inside
 * the patch gather -- bit for bit the f32 values `(x.float() / 255 - mean) / std` gives on the host; a quarter of the bytes
 * to upload.  pevit_set_input_norm must have been called on the context. */
int pevit_set_input_norm(pevit_ctx* ctx, const float* mean3, const float* std3);
int pevit_visual_forward_u8(pevit_ctx* ctx, void* stream, const uint8_t* images, float* feat, int batch, int save_for_backward);
int pevit_train_forward_backward_u8(pevit_ctx* ctx, void* stream, const uint8_t* images, const int64_t* labels,
                                    float* running_mean, float* running_var, int bn_training, float* logits,
                                    float* loss, int batch);
/* whole fine-tune step: zero_grad -> forward -> CE -> backward -> (caller all-reduces) -> SGD */
int pevit_train_forward_backward(pevit_ctx* ctx, void* stream, const float* images, const int64_t* labels,
                                 float* running_mean, float* running_var, int bn_training, float* logits,
                                 float* loss, int batch);

/* ---- measurement: HIP events around every MFMA GEMM launch of the context (the dominant kernel
 * family); totals over the GEMM launches recorded between begin and end.  With pevit_tune(ctx, "profile_all", 1) the
 * HBM-bound kernels of the step are bracketed as well: their records carry epi_mnk[0] = 100 + pevit_prof_kind, the row
 * count in epi_mnk[1], flops 0 and their algorithmic bytes; *launches counts every record, the totals only the GEMMs. */
enum pevit_prof_kind {
    PEVIT_PROF_LN_FWD = 0, PEVIT_PROF_LN_BWD = 1, PEVIT_PROF_ATTN_FWD = 2, PEVIT_PROF_ATTN_BWD = 3, PEVIT_PROF_DELTA_ADD = 4,
    PEVIT_PROF_LOWRANK_U = 5, PEVIT_PROF_LOWRANK_GRAD = 6, PEVIT_PROF_IM2COL = 7, PEVIT_PROF_LOWRANK_BWD = 8,
    PEVIT_PROF_ATTN_FWD_DELTA = 9, PEVIT_PROF_ADAPTER_FWD = 10, PEVIT_PROF_ADAPTER_BWD = 11
};
int pevit_profile_begin(pevit_ctx* ctx, int max_launches);
int pevit_profile_end(pevit_ctx* ctx, double* total_ms, double* total_flops, double* total_bytes, int* launches);
/* launch i (0 <= i < *launches of the last pevit_profile_end): its duration, 2*M*N*K and {epilogue, M, N, K} */
int pevit_profile_launch(pevit_ctx* ctx, int i, double* ms, double* flops, int* epi_mnk);
/* ... and the bytes it must move at least: every operand read once, every result written once */
int pevit_profile_launch_bytes(pevit_ctx* ctx, int i, double* bytes);

/* ---- single kernels, exposed for parity tests and profiling ------------------------- */
int pevit_op_gemm(void* stream, int epilogue, const void* A_bf16, int lda, const void* B_bf16, int ldb, int b_rows,
                  int M, int N, int K, const float* bias, const float* resid, int ldr, float* out_f32, int ldo,
                  void* out_bf16, int ldob, void* out2_bf16, int ldob2, const void* aux_bf16, int ldaux,
                  size_t head_stride, int E, int H, int tokens);
/* fp8 weights (PEVIT_W_FP8_E4M3): B = e4m3 codes [b_rows][ldb] as written by pevit_op_quant_fp8 (k-permuted per 128),
 * bscale = per-output-column scale applied to the accumulator (NULL: none), oscale = per-column factor folded into the
 * bf16 output of the DGELU epilogue (NULL: none); K % 128 == 0.  Epilogues 0..5 only. */
int pevit_op_gemm_fp8(void* stream, int epilogue, const void* A_bf16, int lda, const void* B_codes, int ldb, int b_rows,
                      const float* bscale, const float* oscale, int M, int N, int K, const float* bias,
                      const float* resid, int ldr, float* out_f32, int ldo, void* out_bf16, int ldob, void* out2_bf16,
                      int ldob2, const void* aux_bf16, int ldaux, size_t head_stride, int E, int H, int tokens);
/* fp8 x fp8 (PEVIT_W_FP8_ACT): A = UNSCALED saturating e4m3 codes [M][lda] (pevit_op_cast_fp8), B / bscale as above; the
 * matrix-core instruction is v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales.  Epilogues 0, 1, 2, 4; out2_fp8: the
 * EPI_BIAS_GELU activation output is written as e4m3 codes (ldob2 in codes). */
int pevit_op_gemm_f8a(void* stream, int epilogue, const void* A_codes, int lda, const void* B_codes, int ldb, int b_rows,
                      const float* bscale, int M, int N, int K, const float* bias, const float* resid, int ldr,
                      float* out_f32, int ldo, void* out_bf16, int ldob, void* out2, int ldob2, int out2_fp8,
                      size_t head_stride, int E, int H, int tokens);
/* src (rows x cols f32, cols % 128 == 0) -> e4m3 codes without a scale (|x| > 448 saturates), k-permuted per 128 */
int pevit_op_cast_fp8(void* stream, const float* src, void* codes, int rows, int cols);
/* Which kernel family the last pevit_op_gemm / in-step product of this process was launched on (tests, measurements):
 * 1 plain tile, 2 staggered 8-wave tile, 3 k-split tile (alternate k-tiles), 4 phased k-split tile, 5 stream-K, 6 few-row split-K. */
int pevit_debug_last_gemm_path(void);
/* W (rows x cols f32, cols % 128 == 0) -> per-row power-of-two scales 2^ceil(log2(amax/448)), e4m3 codes [rows][cols]
 * and (codes_t != NULL, rows % 128 == 0) the same codes transposed [cols][rows]; both k-permuted per 128 */
int pevit_op_quant_fp8(void* stream, const float* W, int rows, int cols, void* codes, float* scales, void* codes_t);
int pevit_op_dequant_fp8(void* stream, const void* codes, const float* scales, int rows, int cols, float* out_f32);
int pevit_op_ln_fwd(void* stream, const float* x, const float* gamma, const float* beta, int rows, int E,
                    void* y_bf16, float* y_f32, float* mean, float* rstd);
int pevit_op_ln_bwd(void* stream, const float* dy, const float* x, const float* mean, const float* rstd,
                    const float* gamma, const float* dres, float* dx, void* dx_bf16, int rows, int E);
/* as pevit_op_ln_bwd, with a per-column factor on the bf16 copy only (fp8 weights: channel scales of the consuming GEMM) */
int pevit_op_ln_bwd_scaled(void* stream, const float* dy, const float* x, const float* mean, const float* rstd,
                           const float* gamma, const float* dres, float* dx, void* dx_bf16, int rows, int E,
                           const float* bf16_colscale);
int pevit_op_attn_fwd(void* stream, const void* q, const void* k, const void* v, void* out, int ldo, float* lse,
                      int B, int H, int N);
int pevit_op_attn_bwd(void* stream, const void* q, const void* k, const void* v, const void* out, int ldo,
                      const void* dout, int lddo, const float* lse, void* dqkv, int ld, int B, int H, int N);
int pevit_op_cast_bf16(void* stream, const float* src, void* dst, size_t n, float scale);
/* q16: Q as a bf16 panel [E][64] (Q_q | Q_v), the production operand of the forward delta */
int pevit_op_delta_add(void* stream, void* qbuf, void* vbuf, const float* t, const void* q16_bf16, const float* bias,
                       float ascale, int B, int N, int E);
/* delta_add + attn_fwd as ONE launch (attn_delta.hip) for geometries where pevit_op_attn_delta_hpw(B, H, N) > 0 (runs of that
 * many heads own whole reference rows of the raw reshape, model.py:796-799; N <= 64): q and v are rewritten with q + delta,
 * v + delta; bit-identical to pevit_op_delta_add followed by pevit_op_attn_fwd */
int pevit_op_attn_fwd_delta(void* stream, void* q, const void* k, void* v, const float* t, const void* q16_bf16, const float* bias,
                            float ascale, void* out, int ldo, float* lse, int B, int H, int N);
int pevit_op_attn_delta_hpw(int B, int H, int N);
/* measurement only: device buffer of 8 uint64 per workgroup that the next pevit_op_attn_fwd_delta launches fill with s_memtime
 * stamps at their phase boundaries (NULL switches it off) */
int pevit_debug_timeline(void* buf);
/* measurement only: `workgroups` 256-thread workgroups with `lds_bytes` of LDS each that keep their CU slots for `microseconds`
 * on `stream` -- what the kernels of an overlapped RCCL all-reduce do to the one-tile-per-CU GEMMs (scripts/r4_coresidency.py) */
int pevit_debug_occupy(void* stream, int workgroups, int lds_bytes, double microseconds);
int pevit_op_lowrank_u(void* stream, const void* dqkv, int ld, const void* qT, float* u32, void* u_cols, int B,
                       int H, int N, int E);
int pevit_op_lowrank_grad(void* stream, const void* xn, int ldx, const float* u32, const void* dqkv, int ld,
                          const float* t, float* partial, float* dbias_partial, int B, int H, int N, int E);
int pevit_op_lowrank_chunks(int T);
/* post-MLP bottleneck adapters (adapter_model.py:264-282, compacter_model.py:302-308,432-448), one layer:
 * G[e][j] = sum_r X[r][e] Y[r][j] as per-chunk partials [chunks][E][64] (+ column sums of X / Y, may be NULL) */
int pevit_op_tn_chunks(int T);
int pevit_op_tn_gemm64(void* stream, const void* X_bf16, int ldx, const void* Y_bf16, int ldy, float* partial, float* csx,
                       float* csy, int T, int E);
/* LayerNorm backward with trainable affine: dx = dres + LN'(dy); partial[blocks][3][E] = sums of dy*xhat, dy, dres */
int pevit_op_lna_blocks(int rows);
int pevit_op_ln_bwd_affine(void* stream, const float* dy, const float* x, const float* mean, const float* rstd,
                           const float* gamma, const float* dres, float* dx, void* dx_bf16, float* partial, int rows, int E);
/* out0[i] += sum_c partial[c][i]  (i < n), or with three outputs partial[c][3][n] -> out0/1/2; deterministic */
int pevit_op_colsum_reduce(void* stream, const float* partial, int chunks, int n, float* out0, float* out1, float* out2);
/* f32 master parameters -> bf16 panels wd [64][E], wdT [E][64], wu [E][64], wuT [64][E].  Adapter: p0 = down.weight
 * (64,E), p1 = up.weight (E,64).  Compacter: rule (4,4,4), p0..p3 = down.W_left, down.W_right, up.W_left, up.W_right */
int pevit_op_prep_bottleneck(void* stream, int method, const float* rule, const float* p0, const float* p1, const float* p2,
                             const float* p3, void* wd, void* wdT, void* wu, void* wuT, int E);
/* chain rule from the dense panel gradients Gd, Gu ([E][64] each) onto the reference's tensors inside `grads`
 * (accumulating) at float offsets off0.. (Adapter: down.weight, up.weight; Compacter: the four W_left / W_right) */
int pevit_op_chain_bottleneck(void* stream, int method, const float* Gd, const float* Gu, const float* rule,
                              const float* params, float* grads, int E, size_t off0, size_t off1, size_t off2, size_t off3);
/* images (B,3,R,R) f32 -> patches [B*(R/P)^2][Kpad] bf16, k = c*P*P + py*P + px, zero padded (conv1, model.py:1036) */
int pevit_op_im2col(void* stream, const float* images, void* patches_bf16, int B, int R, int P, int Kpad);
int pevit_op_im2col_u8(void* stream, const uint8_t* images, const float* mean3, const float* std3, void* patches_bf16, int B,
                       int R, int P, int Kpad);
/* knobs for A/B measurements, held in the context (ctx == NULL: the process-wide defaults that only the
 * context-free pevit_op_* entry points above use): "gemm_config" (-1 = per-problem heuristic, 0..5 = force a tile
 * configuration: 128x128, 64x128, 64x64 with 4 waves; 256x128, 256x256, 320x256 with 8 waves; 6 = 128x64 with 4 waves),
 * "gemm_persistent", "gemm_big" (0 = never pick the 8-wave tiles), "gemm_big_bias", "gemm_kswitch", "gemm_cfg_longk" /
 * "gemm_cfg_shortk" (configuration of the few-tile problems above / below kswitch), "gemm_ablate" (bit 0 skips the
 * k-loop, bit 1 the epilogue stores, bit 2 the operand stream, bit 3 ds_read + MFMA), "side_stream" (ctx only),
 * "gemm_streamk" (0 = never use the stream-K decomposition of the few-tile long-K products), "gemm_ksplit" (0 = never
 * use the one-tile-per-CU 160x128 / 96x128 k-split tile of the N = E products; 2 = also on problems of several rounds),
 * "gemm_ksplit_stagger" (2 = phased kernel, default; 1 / 0 = the alternate-k-tile kernel with / without the half-iteration
 * offset), "gemm_ksplit_small", "gemm_ksplit_mink", "gemm_kphase_nl" (requests in the LOAD section: 8, or between the
 * MFMAs: 2), "gemm_kz2" (1 = two workgroups per 160x128 tile, half of K each, where those tiles fill at most half the chip;
 * default 0: measured slower, profiles/r05_experiments.md), "gemm_stagger" (0 = legacy 8-wave kernel, 1 = staggered, 2 = also the 256x128 tile), "gemm_band",
 * "gemm_skinny" (0 = never use the few-row split-K kernel), "gemm_skinny_maxm" / "_mink" / "_slices", "gemm_sk_share" /
 * "gemm_sk_band" (with gemm_streamk = 2), "lowrank_xcd", "fused_bottleneck", "profile_all", "fused_attn_delta" (0 = delta_add + attn_fwd as two launches), "fp8_tail" (0 = t = xn P as a launch of its own with fp8 weights), "adapter_fused" (0 = the post-MLP adapter as separate
 * LayerNorm / GEMM launches), "lowrank_combo" (0 = lowrank_u and lowrank_grad as two launches per layer) and
 * "dx_stored" (ctx only);
 * returns 0, or -1 for an unknown key */
int pevit_tune(pevit_ctx* ctx, const char* key, int value);

/* Stream-K GEMMs hand partial tiles from workgroup to workgroup inside one launch; a consumer that waits ~1 s for a
 * partial gives up (the tile is then wrong) and raises an error word instead of hanging the GPU.  Synchronises the
 * stream, returns 1 if that ever happened on this context's workspace (ctx NULL: the workspace of the pevit_op_*
 * entry points) and clears the word, 0 otherwise, -1 on a HIP error.  The reference has no counterpart (its GEMMs are
 * ATen calls, model.py:675,817); the engine checks it after the first step. */
int pevit_streamk_error(pevit_ctx* ctx, void* stream);
/* The same word WITHOUT clearing it, and the number of optimizer updates pevit_sgd_step has withheld on device since the word
 * was raised (the fused SGD kernel leaves parameters and momentum untouched while it is non-zero, so a corrupted tile never
 * reaches them).  The word stays raised -- and every further update withheld -- until pevit_streamk_error clears it: a caller
 * that swallows the error cannot silently train on. */
int pevit_streamk_status(pevit_ctx* ctx, void* stream, unsigned* error_word, unsigned* skipped_updates);

/* ---- data-parallel gradient exchange without a collective-library kernel (the optional allreduce_flat) ----
 * The reference has no multi-process path at all (utils/comm.py:62-150 is dormant); `north_star` asks for an all-reduce of the flat
 * adapter-gradient buffer only.  One pevit_ar per process / GPU: a mailbox in device memory ([2][world][max_floats] + flags,
 * allocated by pevit_ar_create -- the only allocation), exported as an IPC handle (pevit_ar_handle_bytes() bytes, to be carried to
 * the peers by whatever the host has: torch.distributed object collectives, a file, MPI) and opened by every peer with
 * pevit_ar_import.  pevit_allreduce_flat then sums buf[0:n] over the ranks in place: one device-to-device copy of the
 * contribution into every peer's mailbox + one 4-byte flag copy behind it (copy engines over xGMI; no kernel on the sending GPU),
 * and a small local kernel that waits for the peers' flags and adds the contributions in RANK ORDER (identical bits on every
 * rank).  Asynchronous on `stream`; all calls of one pevit_ar must use the same stream, in the same order on every rank.
 * pevit_ar_error: 1 if a reduction ever gave up waiting for a peer (its buffer is then left unreduced); the word stays raised until
 * pevit_ar_reset (round 6: reading it no longer clears it). */
typedef struct pevit_ar pevit_ar;
int pevit_ar_create(pevit_ar** out, int rank, int world, size_t max_floats);
void pevit_ar_destroy(pevit_ar* ar);
int pevit_ar_handle_bytes(void);
int pevit_ar_export(pevit_ar* ar, void* handle_out);
int pevit_ar_import(pevit_ar* ar, int peer, const void* handle);
int pevit_allreduce_flat(pevit_ar* ar, void* stream, float* buf, size_t n);
int pevit_ar_error(pevit_ar* ar, void* stream);       /* 0 fine, 1 a peer never arrived, 2 the ranks passed different sizes */
/* Round 5.  The push is a kernel (stores through the IPC-mapped peer addresses, system-scope fence, one 8-byte release store per
 * peer flag); PEVIT_AR_PUSH=dma selects the copy-engine push of round 4, which an eight-process soak on one device showed to be
 * unreliable (torn / lost flags).  The flag a rank pushes is the pair (epoch, n); ONE workgroup of the reducing launch decides for all of them, so a bucket is
 * reduced everywhere or nowhere; the mailbox is fine-grained memory where the runtime exports such an allocation
 * (pevit_ar_fine_grained; PEVIT_AR_COARSE=1 forces the plain one).  pevit_ar_error_word: the device address of the error word, for
 * pevit_set_external_poison -- pevit_sgd_step then withholds the update of a step whose exchange failed (and writes NaN over that
 * step's loss), as it does for a stream-K hand-off error; the word stays raised until pevit_ar_reset clears it.
 * EXPERIMENTAL: exercised with 2 / 4 / 8 processes on ONE device only; the measured multi-GPU route is RCCL (torch.distributed). */
const unsigned* pevit_ar_error_word(pevit_ar* ar);
/* back to the initial protocol state (epoch 0, flags and error word cleared).  Collective by convention: every rank drains its device
 * and meets the others BEFORE and AFTER this call (dp.FlatAllReduce.resync does both). */
int pevit_ar_reset(pevit_ar* ar, void* stream);
int pevit_ar_fine_grained(pevit_ar* ar);
int pevit_set_external_poison(pevit_ctx* ctx, const unsigned* device_word);
/* Round 5 (data parallelism; no reference counterpart -- /root/reference/vision_benchmark/utils/comm.py is dormant): a hipEvent_t the
 * fused step (pevit_train_forward_backward[_u8]) waits for on its stream AFTER the stem (patch gather, patch embedding, class /
 * position rows, ln_pre: no trainable parameter, no gradient) and BEFORE the adapters are first read and the gradient buffer is
 * cleared.  Recorded by the caller behind the previous step's gradient exchange + pevit_sgd_step, which may then run on another
 * stream, under the next step's stem (engine.HipEngine dp_exchange_mode = "pipelined").  NULL detaches it. */
int pevit_set_step_gate(pevit_ctx* ctx, void* hip_event);

#ifdef __cplusplus
}
#endif
#endif /* PEVIT_HIP_H */
