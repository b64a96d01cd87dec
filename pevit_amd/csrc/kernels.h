// Internal launcher interface between the C-ABI layer (capi.hip) and the kernel files.
#pragma once
#include "common.h"

enum GemmEpilogue {
    EPI_QKV_HEADS = 0,       // +bias, bf16 -> (which, b*H+h, n, d) head layout; cols >= 3E -> f32 t[T][64]
    EPI_BIAS_RESID_F32 = 1,  // out_f32 = acc + bias + resid
    EPI_BIAS_GELU = 2,       // h = acc + bias (bf16, saved) ; g = QuickGELU(h) (bf16)
    EPI_DGELU_BF16 = 3,      // out_bf16 = acc * QuickGELU'(aux)
    EPI_F32 = 4,             // out_f32 = acc
    EPI_BF16 = 5,            // out_bf16 = acc
    EPI_BIAS_BF16 = 6,       // out_bf16 = acc + bias
    EPI_PATCH_EMBED = 7,     // out_f32[b*Ntok+1+g] = acc + pos[1+g]
    EPI_BIAS_RELU_BF16 = 8,  // out_bf16 = relu(acc + bias)
    EPI_BIAS_RESID_KEEP = 9, // out_f32 = acc + bias + resid ; out2_f32 = acc + bias   (post-MLP adapters need h itself)
    EPI_BIAS_GELUNEW = 10,   // a = acc + bias (bf16, saved) ; g = gelu_new(a) (bf16)
    EPI_DRELU_BF16 = 11,     // out_bf16 = acc * (aux > 0)
    EPI_DGELUNEW_BF16 = 12,  // out_bf16 = acc * gelu_new'(aux)
};

struct GemmParams {
    const bf16* A; int lda;
    const void* B; int ldb; int Nb;   // Nb = readable rows of B (>= N); ldb in elements (bf16, or fp8 codes)
    // fp8 B with a bf16 tail (the QKV product of the attention-site adapters with fp8 frozen weights): output columns >= n_fp8 take
    // their B rows from B2 (bf16, pitch ldb2 elements, Nb2 readable rows) -- the 64 trainable panel rows P_q^T | P_v^T -- inside the
    // same launch (gemm8_kernel only: pevit_gemm_mixed_ok).  n_fp8 must be a multiple of the tile width.  B2 == nullptr: off.
    const bf16* B2; int ldb2, Nb2, n_fp8;
    int b_fp8;                        // B holds e4m3 codes, k-permuted per 128 (fp8_kperm), K % 128 == 0
    int a_fp8;                        // A holds e4m3 codes as well (lda in codes): fp8 x fp8 on the MX matrix instruction (needs b_fp8)
    int out2_fp8;                     // EPI_BIAS_GELU: the activation output (outb2, ldob2 in codes) is written as k-permuted e4m3 codes
    const float* bscale;              // fp8 B: per-output-channel (power-of-two) scale, applied to the accumulator
    const float* oscale;              // EPI_DGELU_BF16 only: per-column factor folded into the bf16 output (or null)
    int M, N, K;
    const float* bias;
    const float* resid; int ldr;
    float* outf; int ldo;
    float* outf2; int ldo2;
    bf16* outb; int ldob;
    bf16* outb2; int ldob2;
    const bf16* aux; int ldaux;
    // head-layout epilogue
    size_t head_stride;   // elements between the q, k and v planes
    int E, H, Ntok;
    int dbg;              // measurement only: bit 0 skips the k-loop, bit 1 the epilogue stores, bit 2 the operand stream, bit 3 ds_read + MFMA
    // stream-K workspace (gemm.hip gemm_streamk_kernel): one 128x128 f32 slab and one flag per residency slot, or null
    float* sk_slab; unsigned* sk_flag; int sk_slots;
    int sk_share, sk_band;   // set by the launcher: k-iterations per workgroup, m-tiles per band of the tile walk
    int band;                // set by the launcher: m-tiles per band of gemm.hip tile_origin (0 = TILE_BAND)
    int kz;                  // set by the launcher: K slices per tile across WORKGROUPS (gemm_kphase_kernel<..., KZ = true>: 2), else 0
};

// A/B-measurement knobs.  They live in the context (pevit_tune(ctx, ...)); the single-kernel pevit_op_* entry
// points use one process-wide default instance.
struct GemmTune {
    int config = -1;      // -1: per-problem heuristic, >= 0: force a tile configuration (gemm.hip kConfigs)
    int persistent = 1;
    int ablate = 0;       // GemmParams::dbg
    int kswitch = 2048;   // K from which the few-tile problems use the 128x128 tile instead of 64x128
    int big = 1;          // allow the 8-wave tiles
    int cfg_longk = 0, cfg_shortk = 1;   // tile configuration of the few-tile problems (N = 768 at M = 6400): K >= kswitch / K < kswitch
    int big_bias = 100;   // the 8-wave tile is taken when its stream cost is below big_bias % of the 128x128 tiling's
    int sk_share = 0, sk_band = 0;   // measurement (gemm_streamk = 2): k-iterations per stream-K workgroup, m-tiles per band
    int ksplit = 1;       // N = E long-K products with ~one 160x128 tile per CU: 8-wave tile, two wave groups on alternate k-tiles
    int ksplit_small = 1;     // ... also as a 96x128 tile where that fills the chip and 160x128 does not (M = 3200)
    int ksplit_stagger = 2;   // ... 1: its two wave groups half an iteration apart (alternate k-tiles); 2: phased kernel (groups split each k-tile, 4 stages)
    int skinny = 1;           // few-row long-K products (M <= skinny_maxm, K >= 64 * skinny_mink): K slices, last arriver sums the slabs (gemm_skinny_kernel)
    int skinny_maxm = 128, skinny_mink = 24, skinny_slices = 0;   // skinny_slices > 0: measurement
    int kz2 = 0;              // ... opt-in (round 5, measured slower): two workgroups per 160x128 tile, half of K each, where those tiles fill at most half the chip (M = 3200, N = 768, long K)
    int kphase_nl = 8;        // ... phased kernel: LDS-DMA pieces per k-tile requested in the LOAD section (the rest between the MFMAs)
    int ksplit_mink = 512;    // ... from this K on
    int band = -1;        // >= 0 forces GemmParams::band of the one-round 8-wave launches (measurement); -1 = XCD-aligned
    int stagger = 1;      // 8-wave tiles (bf16 B): the staggered two-group kernel (gemm8_kernel) instead of gemm_kernel
    int streamk = 1;      // few-tile long-K problems: stream-K decomposition of the 128x128 tiling (needs GemmParams::sk_slab)
};
constexpr int PEVIT_SK_SLAB_FLOATS = 128 * 128;   // one partial tile per residency slot
constexpr int PEVIT_SK_MAX_SLOTS = 1024;
int pevit_gemm_last_path();                        // 1 plain tile, 2 staggered 8-wave, 3 k-split (alternate k-tiles), 4 phased k-split, 5 stream-K, 6 few-row split-K
int pevit_gemm_sk_slots();                         // residency slots of the stream-K kernel on this device (2 per CU, multiple of 8)

int pevit_launch_gemm(int epi, const GemmParams& p, const GemmTune& t, hipStream_t stream);
bool pevit_gemm_mixed_ok(const GemmParams& p, const GemmTune& t);   // would this fp8-B problem with a bf16 tail (B2) run on a kernel that supports it?

// ---- norm.hip --------------------------------------------------------------------
// y = LN(x) * gamma + beta over the last dim (eps 1e-5, f32 statistics: model.py:154-160)
int pevit_launch_ln_fwd(const float* x, const float* gamma, const float* beta, int rows, int E,
                        bf16* y_bf16, float* y_f32, float* mean, float* rstd, hipStream_t s,
                        size_t xstride = 0, int f32 = 0, unsigned char* y_fp8 = nullptr);   // y_fp8: k-permuted e4m3 copy [rows][E]
// dx_out = dres + LN-backward(dy)   (gamma/beta frozen: no parameter grads).  dy is f32, or (dy_stored) in the activation
// storage type -- bf16 in production -- when it is the output of a dX GEMM
int pevit_launch_ln_bwd(const void* dy, const float* x, const float* mean, const float* rstd,
                        const float* gamma, const float* dres, float* dx_out, bf16* dx_bf16, int rows, int E,
                        hipStream_t s, size_t xstride = 0, const float* bf16_colscale = nullptr, int f32 = 0, int dy_stored = 0,
                        int res_period = 0,       // res_period > 0: dres is read on rows that are multiples of it only (zero elsewhere)
                        int res16 = 0,            // dres points at bf16 values (may be dx_bf16 itself: in place); dx_out may then be null
                        const float* res_colscale = nullptr);   // res16 + fp8 weights: the power-of-two column scales folded into dres (taken out exactly)

// ---- attention.hip ---------------------------------------------------------------
// q,k,v: (B*H, N, 64) bf16 (q pre-scaled by 1/8, deltas already added); out: rows (b*N+n), cols h*64+d
int pevit_launch_attn_fwd(const bf16* q, const bf16* k, const bf16* v, bf16* out, int ldo,
                          float* lse, int B, int H, int N, hipStream_t s, unsigned char* out_fp8 = nullptr);   // + e4m3 copy, row pitch ldo codes
// dqkv: row layout [T][ld]: cols [0,E) dq, [E,2E) dk, [2E,3E) dv
int pevit_launch_attn_bwd(const bf16* q, const bf16* k, const bf16* v, const bf16* out, int ldo,
                          const bf16* dout, int lddo, const float* lse, bf16* dqkv, int ld,
                          int B, int H, int N, hipStream_t s, int dout_cls_only = 0);   // dout_cls_only (N <= 64): dout is zero except on token 0 of every image; the other rows are not read

// ---- attn_delta.hip (attention-site adapters fused with the attention core, N <= 64) --------------
// heads per workgroup of the fused forms for this geometry, or 0 when there is none (the two-kernel path is used)
int pevit_attn_delta_hpw(int B, int H, int N);
// delta_add + attn_fwd in one launch: q, v (head layout) are rewritten with q + delta, v + delta (saved for backward)
int pevit_launch_attn_fwd_delta(bf16* q, const bf16* k, bf16* v, const float* t, const bf16* q16, const float* bias,
                                float ascale, bf16* out, int ldo, float* lse, int B, int H, int N, hipStream_t s);

void pevit_attn_delta_set_timeline(void* buf);      // measurement only: 8 s_memtime stamps per workgroup of the next fused launches (null = off)

// ---- lowrank.hip -----------------------------------------------------------------
struct AdapterPanels {      // per layer, rewritten every step from the f32 master parameters
    bf16* w_aug_rows;       // &Wqkv_aug[3E][0]  : 64 rows x E  (P_q^T | P_v^T)
    int ldw;
    bf16* wT_aug_cols;      // &WqkvT_aug[0][3E] : E rows, 64 cols (ascale*P_q | ascale*P_v)
    int ldwT;
    float* q32;             // [E][64] f32 : Q_q | Q_v
    bf16* qT;               // [64][E] bf16: Q_q^T ; Q_v^T
    bf16* q16;              // [E][64] bf16: Q_q | Q_v (operand of the forward delta; bf16 storage only)
};
struct LayerStrides { size_t arena_bytes; size_t param_floats; };   // per-layer pointer advance
// KAdaptation: P[:,j] = s_j (x) l_j , Q[:,j] = t_j (x) r_j   (model.py:567-580); all layers
int pevit_launch_prep_kadapt(const float* rule1_l, const float* rule1_r, const float* rule2_l,
                             const float* rule2_r, const float* q_left, const float* q_right,
                             AdapterPanels pan, int E, float ascale, int layers, LayerStrides st, hipStream_t s, int f32 = 0);
// LoRA: P_q = A1q^T, Q_q = A2q (rank r zero-padded to 32)   (lora_model.py:490-514); all layers
int pevit_launch_prep_lora(const float* a1q, const float* a2q, const float* a1v, const float* a2v,
                           int r, AdapterPanels pan, int E, float ascale, int layers, LayerStrides st, hipStream_t s, int f32 = 0);
// q_buf_flat[rr*E+e] += ascale * t[row(rr)][0:32] . Q_q[e] + bias[e]   (and v with cols 32:64)
// rr is the reference's (n*B+b) row index of the raw reshape (model.py:796-799); row(rr)=b*N+n.
// q16: the bf16 panel [E][64] of Q (production operand); q32: the f32 panel (f32 verification mode)
int pevit_launch_delta_add(bf16* qbuf, bf16* vbuf, const float* t, const float* q32, const bf16* q16,
                           const float* bias, float ascale, int B, int N, int E, hipStream_t s, int f32 = 0);
// u[row(rr)][0:32] = dDelta_q[rr] . Q_q ; [32:64] = dDelta_v[rr] . Q_v ; written f32 (u32) and
// bf16 into dqkv[:, 3E:3E+64]
int pevit_launch_lowrank_u(const bf16* dqkv, int ld, const bf16* qT, float* u32, bf16* u_bf16_cols,
                           int B, int H, int N, int E, hipStream_t s);
// partial[chunk][4][E][32]: dP_q, dP_v (= xn^T u), dQ_q, dQ_v (= dDelta^T t_ref); dbias partial[chunk][E]
int pevit_launch_lowrank_grad(const bf16* xn, int ldx, const float* u32, const bf16* dqkv, int ld,
                              const float* t, float* partial, float* dbias_partial, int chunks,
                              int B, int H, int N, int E, hipStream_t s, int xcd_order = 1);   // xcd_order: XCD-contiguous workgroup order (measurement knob)
int pevit_lowrank_chunks(int T);
// u + dQ / d bias of this layer and the deferred dP of the previously processed layer in ONE launch (lowrank_combo_kernel)
int pevit_launch_lowrank_combo(int this_layer, int prev, const bf16* dqkv, int ld, const bf16* qT, float* u32, bf16* ucols, const float* t,
                               float* partial, float* dbias_partial, const bf16* xn_prev, int ldx, const float* u32_prev,
                               float* partial_prev, int B, int H, int N, int E, hipStream_t s);
// reduce the per-chunk partials of all layers and apply the chain rule onto the reference's
// parameter tensors (flat gradient buffer, accumulating)
int pevit_launch_chain_kadapt(const float* partial, size_t partial_layer, const float* dbias_partial, size_t dbias_layer,
                              int chunks, float ascale, int layers, float* G, float* rule_scratch, const float* params,
                              float* grads, size_t p_layer0, size_t p_layer_stride, int E, hipStream_t s);
int pevit_launch_rule_sum(const float* rule_scratch, float* grads, int l_lo, int l_hi, hipStream_t s);
int pevit_launch_chain_lora(const float* partial, size_t partial_layer, int chunks, float ascale, int r, int layers,
                            float* G, float* grads, size_t p_layer0, size_t p_layer_stride, int E, hipStream_t s);

// ---- misc.hip --------------------------------------------------------------------
int pevit_launch_cast_bf16(const float* src, bf16* dst, size_t n, float scale, hipStream_t s, int f32 = 0);
// dst[c][r] = scale(r) * src[r][c]  (bf16 out), used once at load for the backward weights
int pevit_launch_transpose_bf16(const float* src, int rows, int cols, bf16* dst, int ldd,
                                int scaled_rows, float scale, hipStream_t s, int f32 = 0);
int pevit_launch_permute_rows(const float* src, float* dst, int N, int B, int E, int to_internal,
                              hipStream_t s);
int pevit_launch_scale_f32(float* p, size_t n, float scale, hipStream_t s);
int pevit_launch_zero(void* ptr, size_t bytes, hipStream_t s);      // the step's memsets as a kernel (capturable in order into a HIP graph)
int pevit_launch_sgd(float* p, const float* g, float* mom, const unsigned char* has_grad, size_t n,
                     float lr, float momentum, float wd, int first_step, float grad_scale, hipStream_t s,
                     const unsigned* poison = nullptr,    // device word: non-zero = skip the update (stream-K hand-off error)
                     unsigned* skipped = nullptr, const unsigned* poison2 = nullptr, float* loss_slot = nullptr);        // device counter of the updates skipped that way

int pevit_launch_occupy(int blocks, int lds_bytes, double micros, hipStream_t s);   // measurement only (pevit_debug_occupy)

// ---- fp8.hip (e4m3 codes + power-of-two channel scales of the frozen weights) -----------------
int pevit_launch_quant_rows_fp8(const float* W, int rows, int cols, unsigned char* out, int ldo, float* scale, int scaled_rows,
                                float pre, hipStream_t s);
int pevit_launch_quant_transpose_fp8(const float* W, int rows, int cols, const float* scale, unsigned char* outT, int ldo,
                                     int scaled_rows, float pre, hipStream_t s);
int pevit_launch_cast_bf16_cols(const float* src, bf16* dst, size_t rows, int cols, const float* colscale, hipStream_t s);
int pevit_launch_cast_fp8(const float* src, unsigned char* dst, size_t rows, int cols, hipStream_t s);
int pevit_launch_dequant_rows_fp8(const unsigned char* codes, int ldc, const float* scale, int rows, int cols, float* out,
                                  hipStream_t s);

// ---- verify.hip (f32-class verification mode: plain f32 kernels for the matrix-core contractions) -----------------
int pevit_launch_gemm_f32(int epi, const GemmParams& p, hipStream_t s);      // A, B and the bf16-declared buffers hold f32
int pevit_launch_attn_fwd_f32(const float* q, const float* k, const float* v, float* out, int ldo, float* lse, int B, int H, int N,
                              hipStream_t s);
int pevit_launch_attn_bwd_f32(const float* q, const float* k, const float* v, const float* out, int ldo, const float* dout,
                              int lddo, const float* lse, float* dqkv, int ld, int B, int H, int N, hipStream_t s);
int pevit_launch_lowrank_u_f32(const float* dqkv, int ld, const float* q32, float* u32, float* ucols, int B, int H, int N, int E,
                               hipStream_t s);
// G[e][j] = sum_r X[r][e] Y[r][j], partial / column-sum layout of tn_gemm64 (adapter.hip)
int pevit_launch_tn_gemm64_f32(const float* X, int ldx, const float* Y, int ldy, float* partial, float* csx, float* csy, int T, int E,
                               hipStream_t s);
int pevit_launch_lowrank_grad_f32(const float* xn, int ldx, const float* u32, const float* dqkv, int ld, const float* t,
                                  float* partial, float* dbias_partial, int chunks, int B, int H, int N, int E, hipStream_t s);

// ---- stem_head.hip -----------------------------------------------------------------
int pevit_launch_im2col(const float* img, bf16* out, int B, int R, int P, int Kp, hipStream_t s, int f32 = 0);
// uint8 pixels with ToTensor + Normalize folded in: x = (u8 / 255 - mean[c]) / std[c] (feature.py:537-542)
int pevit_launch_im2col_u8(const unsigned char* img, const float* mean3, const float* std3, bf16* out, int B, int R, int P, int Kp,
                           hipStream_t s, int f32 = 0);
int pevit_launch_conv_weight(const float* w, bf16* out, int E, int K, int Kp, hipStream_t s, int f32 = 0);
int pevit_launch_cls_row(const float* cls, const float* pos, float* x, int B, int N, int E, hipStream_t s);
int pevit_launch_head(const float* feat, const int64_t* labels, const float* W, const float* bias, float* gW, float* gb,
                      float* running_mean, float* running_var, int training, float* ybn, float* rstd, float* logits,
                      float* dlogits, float* dybn, float* loss, float* dfeat, int B, int D, int Cc, hipStream_t s, bf16* dfeat_bf16 = nullptr);   // dfeat_bf16: bf16 copy of dfeat written by the BatchNorm backward (no cast launch)

// ---- adapter.hip (post-MLP bottleneck adapters: Adapter, Compacter) -------------------------------
struct BottleneckPanels { bf16* wd; bf16* wdT; bf16* wu; bf16* wuT; };   // [64][E], [E][64], [E][64], [64][E]
int pevit_tn_chunks(int T);
int pevit_lna_blocks(int rows);
int pevit_launch_prep_adapter(const float* w_down, const float* w_up, BottleneckPanels pan, int E, int layers, LayerStrides st,
                              hipStream_t s, int f32 = 0);
int pevit_launch_prep_compacter(const float* rule, const float* dWl, const float* dWr, const float* uWl, const float* uWr,
                                BottleneckPanels pan, int E, int layers, LayerStrides st, hipStream_t s, int f32 = 0);
// G[e][j] = sum_r X[r][e] Y[r][j] (per-chunk partials [chunk][E][64]); optional column sums of X / Y
int pevit_launch_tn_gemm64(const bf16* X, int ldx, const bf16* Y, int ldy, float* partial, float* csx, float* csy, int T, int E,
                           hipStream_t s);
// S = fn(X W1^T [+ b1]) (T x 64, saved) ; out = S W2^T [+ b2 + resid] (f32) in one launch.  mode: 0 forward ReLU, 1 forward
// gelu_new (S_pre = bf16 pre-activation), 2 / 3 backward (fn = multiply by the activation derivative at the saved aux)
int pevit_launch_bottleneck_pair(int mode, const bf16* X, int ldx, const bf16* W1, const float* b1, const bf16* aux, bf16* S_out,
                                 bf16* S_pre, const bf16* W2, const float* b2, const float* resid, float* out, int T, int E,
                                 hipStream_t s);
int pevit_launch_ln_bwd_affine(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                               const float* dres, float* dx, bf16* dx_bf16, float* partial, int rows, int E, hipStream_t s, int f32 = 0);
// ---- adapter_fused.hip: the post-MLP adapter as one launch per direction (bf16 storage, E a multiple of 256) ----
bool pevit_adapter_fused_ok(int E);
int pevit_num_cus();                 // compute units of the current device (cached)
int pevit_adapter_blocks(int T);     // workgroups (= blocks of LayerNorm-affine partials) of the fused backward
// act_kind 0 = ReLU (Adapter), 1 = gelu_new (Compacter).  hraw = c_proj accumulators WITHOUT their bias (f32), bpr = that bias.
int pevit_launch_adapter_fwd(int act_kind, const float* hraw, const float* bpr, const float* x_mid, const float* gamma, const float* beta,
                             const bf16* wd, const float* b_down, const bf16* wu, const float* b_up, bf16* z, float* mean_a,
                             float* rstd_a, bf16* act, bf16* apre, float* x_out, int T, int E, hipStream_t s);
// saved = act (ReLU) / apre (gelu_new); partial: [pevit_lna_blocks(T)][3][E] like ln_bwd_affine
int pevit_launch_adapter_bwd(int act_kind, const bf16* dyb, const float* dres, const bf16* wuT, const bf16* saved, const bf16* wdT,
                             const float* hraw, const float* bpr, const float* mean_a, const float* rstd_a, const float* gamma,
                             bf16* dpre, bf16* dh_bf16, float* partial, int T, int E, hipStream_t s, const bf16* tn_x1 = nullptr,
                             const bf16* tn_y1 = nullptr, float* tn_partial1 = nullptr, const bf16* tn_x2 = nullptr,
                             const bf16* tn_y2 = nullptr, float* tn_partial2 = nullptr, float* tn_csy2 = nullptr,
                             int tn_blocks = 0);      // workgroups of the contraction range, 0 = one per unit pair
int pevit_launch_colsum_reduce(const float* partial, int chunks, int n, float* out, int layers, size_t partial_layer,
                               size_t out_layer, hipStream_t s);
int pevit_launch_colsum_reduce3(const float* partial, int chunks, int n, float* o0, float* o1, float* o2, int layers,
                                size_t partial_layer, size_t out_layer, hipStream_t s);
int pevit_launch_chain_adapter(const float* Gd, const float* Gu, float* g_down, float* g_up, int E, int layers, size_t g_layer,
                               size_t param_layer, hipStream_t s);
int pevit_launch_chain_compacter(const float* Gd, const float* Gu, const float* rule, const float* params, float* grads, int E,
                                 int layers, size_t g_layer, size_t param_layer, size_t off_dWl, size_t off_dWr, size_t off_uWl,
                                 size_t off_uWr, hipStream_t s);
