// C-ABI layer: context, memory layout of the weight arena / workspace, and the launch
// sequences of the fine-tune step.  See include/pevit_hip.h for the boundary contract.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "../../include/pevit_hip.h"
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
extern "C" void pevit_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* pevit_last_error(void) { return g_err; }
extern "C" int pevit_version(void) { return 1; }

#define CHECK(call)            \
    do {                       \
        if ((call) != 0) return -1; \
    } while (0)

namespace {

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    }
};

struct BlockArena {        // byte offsets inside the weight arena, one per layer
    size_t wqkv, wqkvT, wo, woT, wfc, wfcT, wpr, wprT;
    size_t bqkv, bo, bfc, bpr, ln1w, ln1b, ln2w, ln2b;
    size_t q32, qT, q16;
    size_t wd, wdT, wu, wuT;      // post-MLP adapter panels (bf16), rewritten every step
    size_t wpan;                  // fp8 weights only: the 64 adapter rows P_q^T | P_v^T (bf16) of the separate t = xn P product
    size_t sqkv, so, sfc, spr;    // fp8 weights only: per-output-channel scales (f32, powers of two)
};

struct LayerSaved {        // byte offsets inside the workspace, one per layer (kept for backward)
    size_t x_in, x_mid, mean1, rstd1, mean2, rstd2, xn1, qkv, t, lse, attn_out, h;
    size_t hf32, mean_a, rstd_a, z, apre, act;     // post-MLP adapters
};

}  // namespace

static GemmTune g_default_tune;   // used by the context-free single-kernel entry points (pevit_op_*) only

struct pevit_ctx {
    pevit_dims d;
    int E, L, H, N, P, R, D, C, G2, Kpatch;
    int NQ, NQpad;            // 3E+64 and its multiple-of-128 padding
    bool fp8 = false;         // frozen block weights as e4m3 codes + per-channel scales (fp8.hip)
    bool fp8act = false;      // ... and fp8 A operands in the forward frozen products (PEVIT_W_FP8_ACT)
    size_t w_a8 = 0, w_attn8 = 0;   // e4m3 copies of the LayerNorm output / the attention output (fp8act)
    bool f32 = false;         // f32-class verification mode: every bf16-declared buffer holds f32 (verify.hip)
    size_t es = 2;            // bytes per element of those buffers
    float ascale;             // 160 (model.py:564) or alpha/r (lora_model.py:491)
    // arena
    BlockArena* blk = nullptr;
    size_t a_conv, a_cls, a_pos, a_lnpre_w, a_lnpre_b, a_lnpost_w, a_lnpost_b, a_proj, a_projT, a_phm;
    size_t arena_bytes = 0;
    char* arena = nullptr;
    // workspace
    LayerSaved* sav = nullptr;
    size_t w_skflag = 0, w_skslab = 0; int sk_slots = 0;   // stream-K workspace (gemm.hip), sk_slots = 0: disabled
    size_t w_xfinal, w_xn2, w_g, w_dqkv, w_u32, w_u32b, w_dO, w_dh, w_dxn, w_dxa, w_dxb, w_dyb, w_partial, w_dbias;
    size_t w_G, w_rule, partial_layer, dbias_layer;
    size_t w_dpre, w_dpre2, w_dht, w_dhb, w_tnU, w_tnD, w_csx, w_csy, w_lnp, w_Gd, w_Gu, tn_layer, csx_layer, csy_layer, lnp_layer;
    // post-MLP adapter parameter offsets inside one layer's block of the flat buffer (floats)
    size_t o_nw, o_nb, o_dw, o_db, o_uw, o_ub, o_dWl, o_dWr, o_uWl, o_uWr;
    size_t w_patches, w_xpost, w_feat, w_pmean, w_prstd, w_ybn, w_bnrstd, w_logits, w_dlogits, w_dybn, w_dfeat,
        w_dfeatb, w_dxpost;
    size_t ws_bytes_for_max = 0;
    char* ws = nullptr;
    int max_batch = 0;
    // parameters
    float* params = nullptr; float* grads = nullptr; float* mom = nullptr;
    const unsigned char* grad_mask = nullptr;   // device, 1 = parameter receives gradients
    size_t n_tower = 0, n_total = 0;
    size_t p_layer0 = 0, p_layer_stride = 0;     // offsets in floats
    size_t p_head_w = 0, p_head_b = 0;
    float img_mean[3] = {0.f, 0.f, 0.f}, img_std[3] = {1.f, 1.f, 1.f};   // pevit_set_input_norm: preprocessing of uint8 pixels
    bool img_norm_set = false;
    int saved_batch = 0;
    int saved_kind = 0;       // which forward the saved activations belong to: 1 = transformer seam, 2 = visual (class-token pruned)
    // optional per-GEMM timing (HIP events on the caller's stream), see pevit_profile_begin
    bool prof_on = false;
    int prof_all = 0;         // also bracket the HBM-bound kernels (pevit_tune "profile_all")
    int prof_n = 0, prof_cap = 0;
    hipEvent_t* prof_ev = nullptr;      // 2 per launch
    double* prof_flops = nullptr;
    double* prof_bytes = nullptr;       // algorithmic operand + result bytes of each launch
    float* prof_ms = nullptr;           // filled by pevit_profile_end
    int* prof_shape = nullptr;          // epilogue, M, N, K of each launch
    // second stream for work that is off the backward critical path (adapter-gradient contractions); created on
    // first use, so that contexts can still be sized on machines without a GPU
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // A/B-measurement knobs (pevit_tune): per context, so that contexts stay independent of each other
    GemmTune tune;
    hipEvent_t step_gate = nullptr;         // pevit_set_step_gate: the fused step waits for it between the stem and the first block
    bool gate_now = false;                  // (set by train_fb_impl for the forward pass it starts)
    const unsigned* ext_poison = nullptr;   // pevit_set_external_poison: a second error word that withholds the optimizer update (the DP exchange's)
    bool in_fused_step = false;
    const float* dfeatb_of = nullptr;       // the dfeat buffer whose bf16 copy the head's BatchNorm backward has just left in w_dfeatb (consumed by the next visual backward)
    float* last_loss = nullptr;             // where the loss of the step in flight was written (NaN goes there when its update is withheld)
    int gstream16 = 1;        // attention-site adapters, bf16 weights: the residual GRADIENT stream is carried in bf16 only (the copy the dX GEMMs read), LayerNorm backward read-modify-writes it in place: 10 instead of 16 B per element (round 5)
    int dx_stored = 1;        // dX GEMMs hand the LN-input gradient to LayerNorm backward in the activation storage type (bf16)
    int fused_bn = 0;         // post-MLP adapters: down -> activation -> up (and its backward) as one launch each (adapter.hip
                              // bottleneck_pair_kernel): 24.4 + 22.1 us against 22.9 + 19.5 us for the four GEMM launches -- opt-in
    int lowrank_combo = 1;    // attention-site adapters: u + dQ + d bias of a layer and the dP of the layer before it as one launch
    int adapter_tn_fold = 1;  // ... and the two token-contracted weight-gradient products ride in the backward launch (adapter_fused.hip: af_tn_range; > 1: that many workgroups for them)
    int adapter_fused = 1;    // post-MLP adapters: LayerNorm -> down -> activation -> up -> residual (and its backward) as one launch each
    int fp8_tail = 1;         // fp8 weights: t = xn P as the bf16 tail of the QKV launch (0: a separate small product, as before round 4)
    int fused_attn_delta = 1; // delta-add + attention forward as one launch where the geometry allows (attn_delta.hip)
    int lowrank_xcd = 1;      // lowrank_grad: XCD-contiguous workgroup order (+0.2 % per step)
    int side_stream = 0;      // adapter-gradient contractions on a second stream: +0.5 % step throughput, but it slows the GEMMs
                              // it overlaps by 4 %, which blurs the per-kernel roofline measurement: off by default
};

namespace {

inline bool attention_site(const pevit_ctx* c) {
    return c->d.method == PEVIT_KADAPTATION || c->d.method == PEVIT_LORA;
}
inline bool post_mlp(const pevit_ctx* c) { return c->d.method == PEVIT_ADAPTER || c->d.method == PEVIT_COMPACTER; }

void layout_workspace(pevit_ctx* c, int B, LayerSaved* sav, size_t* total, pevit_ctx* fill) {
    Carver cv;
    const size_t T = (size_t)B * c->N, E = c->E, es = c->es;
    size_t o;
    // stream-K hand-off flags (+1 error word) and partial-tile slabs: first, so that their place does not depend on the batch
    o = cv.take((size_t)(PEVIT_SK_MAX_SLOTS + 2) * 4);                        if (fill) fill->w_skflag = o;   // + error word + skipped-update counter
    o = cv.take((size_t)c->sk_slots * PEVIT_SK_SLAB_FLOATS * 4);              if (fill) fill->w_skslab = o;
    for (int l = 0; l < c->L; ++l) {
        LayerSaved s;
        s.x_in = cv.take(T * E * 4);
        s.x_mid = cv.take(T * E * 4);
        s.mean1 = cv.take(T * 4); s.rstd1 = cv.take(T * 4);
        s.mean2 = cv.take(T * 4); s.rstd2 = cv.take(T * 4);
        s.xn1 = cv.take(T * E * es);
        s.qkv = cv.take(3 * T * E * es);
        s.t = cv.take(T * 64 * 4);
        s.lse = cv.take((size_t)B * c->H * c->N * 4);
        s.attn_out = cv.take(T * E * es);
        s.h = cv.take(T * 4 * E * es);
        s.hf32 = s.mean_a = s.rstd_a = s.z = s.apre = s.act = 0;
        if (post_mlp(c)) {
            s.hf32 = cv.take(T * E * 4); s.mean_a = cv.take(T * 4); s.rstd_a = cv.take(T * 4);
            s.z = cv.take(T * E * es); s.apre = cv.take(T * 64 * es); s.act = cv.take(T * 64 * es);
        }
        if (sav) sav[l] = s;
    }
    const int chunks = pevit_lowrank_chunks((int)T);
    o = cv.take(T * E * 4);                 if (fill) fill->w_xfinal = o;
    o = cv.take(T * E * es);                 if (fill) fill->w_xn2 = o;
    o = cv.take(T * 4 * E * es);             if (fill) fill->w_g = o;       // (fp8act: holds gelu(h) as e4m3 codes, half of it used)
    if (c->fp8act) {
        o = cv.take(T * E);                  if (fill) fill->w_a8 = o;
        o = cv.take(T * E);                  if (fill) fill->w_attn8 = o;
    }
    o = cv.take(T * (size_t)c->NQ * es);     if (fill) fill->w_dqkv = o;
    o = cv.take(T * 64 * 4);                if (fill) fill->w_u32 = o;
    o = cv.take(T * 64 * 4);                if (fill) fill->w_u32b = o;      // second u buffer: dP of a layer is taken one launch later (lowrank_combo)
    o = cv.take(T * E * es);                 if (fill) fill->w_dO = o;
    o = cv.take(T * 4 * E * es);             if (fill) fill->w_dh = o;
    o = cv.take(T * E * 4);                 if (fill) fill->w_dxn = o;
    o = cv.take(T * E * 4);                 if (fill) fill->w_dxa = o;
    o = cv.take(T * E * 4);                 if (fill) fill->w_dxb = o;
    o = cv.take(T * E * es);                 if (fill) fill->w_dyb = o;
    // adapter-gradient partials of every layer (reduced once per step, after the layer loop)
    const size_t part_layer = align_up((size_t)chunks * 4 * E * 32 * 4, 256), db_layer = align_up((size_t)chunks * 2 * E * 4, 256);
    o = cv.take(part_layer * c->L);                      if (fill) { fill->w_partial = o; fill->partial_layer = part_layer; }
    o = cv.take(db_layer * c->L);                        if (fill) { fill->w_dbias = o; fill->dbias_layer = db_layer; }
    o = cv.take((size_t)c->L * 4 * E * 32 * 4);          if (fill) fill->w_G = o;
    o = cv.take((size_t)c->L * 4096 * 4);                if (fill) fill->w_rule = o;
    if (post_mlp(c)) {
        const int tch = pevit_tn_chunks((int)T), lnb = std::max(pevit_lna_blocks((int)T), pevit_adapter_blocks((int)T));
        const size_t tn_layer = (size_t)tch * E * 64 * 4, csx_layer = (size_t)tch * E * 4, csy_layer = (size_t)tch * 64 * 4,
                     lnp_layer = (size_t)lnb * 3 * E * 4;
        o = cv.take(T * 64 * es);            if (fill) fill->w_dpre = o;
        o = cv.take(T * 64 * es);            if (fill) fill->w_dpre2 = o;     // d pre alternates: the deferred d W_down product reads the previous one
        o = cv.take(T * E * 4);             if (fill) fill->w_dht = o;
        o = cv.take(T * E * es);             if (fill) fill->w_dhb = o;
        o = cv.take(tn_layer * c->L);       if (fill) { fill->w_tnU = o; fill->tn_layer = tn_layer; }
        o = cv.take(tn_layer * c->L);       if (fill) fill->w_tnD = o;
        o = cv.take(csx_layer * c->L);      if (fill) { fill->w_csx = o; fill->csx_layer = csx_layer; }
        o = cv.take(csy_layer * c->L);      if (fill) { fill->w_csy = o; fill->csy_layer = csy_layer; }
        o = cv.take(lnp_layer * c->L);      if (fill) { fill->w_lnp = o; fill->lnp_layer = lnp_layer; }
        o = cv.take((size_t)c->L * E * 64 * 4);  if (fill) fill->w_Gd = o;
        o = cv.take((size_t)c->L * E * 64 * 4);  if (fill) fill->w_Gu = o;
    }
    const size_t Bz = (size_t)B, D = c->D, Cc = c->C;
    o = cv.take(Bz * c->G2 * (size_t)c->Kpatch * es);      if (fill) fill->w_patches = o;
    o = cv.take(Bz * E * es);      if (fill) fill->w_xpost = o;
    o = cv.take(Bz * D * 4);      if (fill) fill->w_feat = o;
    o = cv.take(Bz * 4);          if (fill) fill->w_pmean = o;
    o = cv.take(Bz * 4);          if (fill) fill->w_prstd = o;
    o = cv.take(Bz * D * 4);      if (fill) fill->w_ybn = o;
    o = cv.take(D * 4);           if (fill) fill->w_bnrstd = o;
    o = cv.take(Bz * Cc * 4);     if (fill) fill->w_logits = o;
    o = cv.take(Bz * Cc * 4);     if (fill) fill->w_dlogits = o;
    o = cv.take(Bz * D * 4);      if (fill) fill->w_dybn = o;
    o = cv.take(Bz * D * 4);      if (fill) fill->w_dfeat = o;
    o = cv.take(Bz * D * es);      if (fill) fill->w_dfeatb = o;
    o = cv.take(Bz * E * 4);      if (fill) fill->w_dxpost = o;
    *total = cv.off;
}

template <typename T>
inline T* at(char* base, size_t off) { return reinterpret_cast<T*>(base + off); }
// advance a bf16-declared pointer by `elems` elements of the context's storage type (bf16, or f32 in verification mode)
inline bf16* eadv(const pevit_ctx* c, const bf16* p, size_t elems) {
    return reinterpret_cast<bf16*>(const_cast<char*>(reinterpret_cast<const char*>(p)) + elems * c->es);
}

}  // namespace

// ------------------------------------------------------------------------------------
extern "C" int pevit_ctx_create(const pevit_dims* dims, pevit_ctx** out) {
    if (!dims || !out) { pevit_set_error("ctx_create: null argument"); return -1; }
    const pevit_dims d = *dims;
    if (d.width <= 0 || d.width % 128 != 0 || d.width > 1024) {
        pevit_set_error("ctx_create: width %d must be a multiple of 128 in (0,1024]", d.width); return -1;
    }
    if (d.layers <= 0 || d.patch <= 0 || d.resolution <= 0 || d.resolution % d.patch != 0) {
        pevit_set_error("ctx_create: bad layers/patch/resolution %d/%d/%d", d.layers, d.patch, d.resolution); return -1;
    }
    if (d.method < 0 || d.method > PEVIT_NONE) { pevit_set_error("ctx_create: unknown method %d", d.method); return -1; }
    if (d.method == PEVIT_LORA && (d.lora_rank < 1 || d.lora_rank > 32)) {
        pevit_set_error("ctx_create: LoRA rank %d outside [1,32]", d.lora_rank); return -1;
    }
    if (d.weight_format != PEVIT_W_BF16 && d.weight_format != PEVIT_W_FP8_E4M3 && d.weight_format != PEVIT_W_F32_VERIFY &&
        d.weight_format != PEVIT_W_FP8_ACT) {
        pevit_set_error("ctx_create: unknown weight_format %d", d.weight_format); return -1;
    }
    if ((d.weight_format == PEVIT_W_FP8_E4M3 || d.weight_format == PEVIT_W_FP8_ACT) && (d.method == PEVIT_ADAPTER || d.method == PEVIT_COMPACTER)) {
        pevit_set_error("ctx_create: fp8 weights are built for the attention-site methods (KAdaptation, LoRA) and the frozen tower"); return -1;
    }
    if (d.out_dim <= 0 || d.out_dim % 8 != 0 || d.num_classes <= 0) {
        pevit_set_error("ctx_create: bad out_dim/num_classes %d/%d", d.out_dim, d.num_classes); return -1;
    }
    pevit_ctx* c = new (std::nothrow) pevit_ctx();
    if (!c) { pevit_set_error("ctx_create: out of host memory"); return -1; }
    c->d = d;
    c->E = d.width; c->L = d.layers; c->H = d.width / 64; c->P = d.patch; c->R = d.resolution;
    const int grid = d.resolution / d.patch;
    c->G2 = grid * grid; c->N = c->G2 + 1; c->D = d.out_dim; c->C = d.num_classes;
    c->Kpatch = (int)align_up((size_t)3 * d.patch * d.patch, 64);
    c->NQ = 3 * c->E + 64; c->NQpad = (int)align_up((size_t)c->NQ, 128);
    c->ascale = d.method == PEVIT_LORA ? 128.0f / (float)d.lora_rank : 160.0f;
    c->fp8act = d.weight_format == PEVIT_W_FP8_ACT;
    c->fp8 = d.weight_format == PEVIT_W_FP8_E4M3 || c->fp8act;
    c->f32 = d.weight_format == PEVIT_W_F32_VERIFY;
    c->es = c->f32 ? 4 : 2;
    c->sk_slots = c->f32 ? 0 : pevit_gemm_sk_slots();
    if (c->N > 288) { pevit_set_error("ctx_create: %d tokens per image exceeds 288", c->N); delete c; return -1; }

    // ---- weight arena -------------------------------------------------------------
    c->blk = new (std::nothrow) BlockArena[c->L];
    c->sav = new (std::nothrow) LayerSaved[c->L];
    if (!c->blk || !c->sav) { pevit_set_error("ctx_create: out of host memory"); pevit_ctx_destroy(c); return -1; }
    Carver cv;
    const size_t E = c->E;
    for (int l = 0; l < c->L; ++l) {
        BlockArena& b = c->blk[l];
        b.wpan = b.sqkv = b.so = b.sfc = b.spr = 0;
        if (c->fp8) {
            // one byte per weight; rows padded to the largest tile (256) so that clamped tile rows stay readable
            const size_t r1 = align_up(E, 256), r3 = align_up(3 * E, 256), r4 = align_up(4 * E, 256);
            b.wqkv = cv.take(r3 * E);
            b.wpan = cv.take(128 * E * 2);
            b.wqkvT = cv.take(E * (size_t)c->NQ * 2);          // QKV backward keeps bf16: its K mixes frozen rows with the adapter panel
            b.wo = cv.take(r1 * E);       b.woT = cv.take(r1 * E);
            b.wfc = cv.take(r4 * E);      b.wfcT = cv.take(r1 * 4 * E);
            b.wpr = cv.take(r1 * 4 * E);  b.wprT = cv.take(r4 * E);
            b.sqkv = cv.take(3 * E * 4); b.so = cv.take(E * 4); b.sfc = cv.take(4 * E * 4); b.spr = cv.take(E * 4);
        } else {
            const size_t es = c->es;
            b.wqkv = cv.take((size_t)c->NQpad * E * es);
            b.wqkvT = cv.take(E * (size_t)c->NQ * es);
            b.wo = cv.take(E * E * es);   b.woT = cv.take(E * E * es);
            b.wfc = cv.take(4 * E * E * es); b.wfcT = cv.take(4 * E * E * es);
            b.wpr = cv.take(4 * E * E * es); b.wprT = cv.take(4 * E * E * es);
        }
        b.bqkv = cv.take(3 * E * 4); b.bo = cv.take(E * 4); b.bfc = cv.take(4 * E * 4); b.bpr = cv.take(E * 4);
        b.ln1w = cv.take(E * 4); b.ln1b = cv.take(E * 4); b.ln2w = cv.take(E * 4); b.ln2b = cv.take(E * 4);
        b.q32 = cv.take(E * 64 * 4); b.qT = cv.take(64 * E * c->es); b.q16 = cv.take(E * 64 * 2);
        b.wd = cv.take(64 * E * c->es); b.wdT = cv.take(64 * E * c->es); b.wu = cv.take(64 * E * c->es); b.wuT = cv.take(64 * E * c->es);
    }
    c->a_conv = cv.take(align_up(E, 128) * (size_t)c->Kpatch * c->es);
    c->a_cls = cv.take(E * 4);
    c->a_pos = cv.take((size_t)c->N * E * 4);
    c->a_lnpre_w = cv.take(E * 4); c->a_lnpre_b = cv.take(E * 4);
    c->a_lnpost_w = cv.take(E * 4); c->a_lnpost_b = cv.take(E * 4);
    c->a_proj = cv.take(align_up((size_t)c->D, 128) * E * c->es);      // [D][E]  (proj^T)
    c->a_projT = cv.take(E * (size_t)c->D * c->es);                   // [E][D]
    c->a_phm = cv.take(64 * 4);
    c->arena_bytes = cv.off;

    // ---- flat trainable parameters (reference named_parameters() order) -----------
    if (d.method == PEVIT_KADAPTATION) {
        c->p_layer0 = 4 * 32 * 32; c->p_layer_stride = 5 * E;
    } else if (d.method == PEVIT_LORA) {
        c->p_layer0 = 0; c->p_layer_stride = 4 * (size_t)d.lora_rank * E;
    } else if (d.method == PEVIT_ADAPTER) {
        // adapter_norm_before.{weight,bias}, adapter_down.1.{weight (64,E), bias}, adapter_up.{weight (E,64), bias}
        c->o_nw = 0; c->o_nb = E; c->o_dw = 2 * E; c->o_db = c->o_dw + 64 * E; c->o_uw = c->o_db + 64;
        c->o_ub = c->o_uw + 64 * E;
        c->p_layer0 = 0; c->p_layer_stride = c->o_ub + E;
    } else if (d.method == PEVIT_COMPACTER) {
        // adapter_norm_before.{weight,bias}, adapter_down.1.{W_left (4,E/4,1), W_right (4,1,16), b (64)},
        // adapter_up.{W_left (4,16,1), W_right (4,1,E/4), b (E)}
        c->o_nw = 0; c->o_nb = E; c->o_dWl = 2 * E; c->o_dWr = c->o_dWl + E; c->o_db = c->o_dWr + 64;
        c->o_uWl = c->o_db + 64; c->o_uWr = c->o_uWl + 64; c->o_ub = c->o_uWr + E;
        c->p_layer0 = 0; c->p_layer_stride = c->o_ub + E;
    } else {
        c->p_layer0 = 0; c->p_layer_stride = 0;
    }
    c->n_tower = c->p_layer0 + c->p_layer_stride * c->L;
    c->p_head_w = c->n_tower;
    c->p_head_b = c->p_head_w + (size_t)c->C * c->D;
    c->n_total = c->p_head_b + c->C;
    *out = c;
    return 0;
}

extern "C" void pevit_ctx_destroy(pevit_ctx* c) {
    if (!c) return;
    for (int i = 0; i < 2 * c->prof_cap; ++i) (void)hipEventDestroy(c->prof_ev[i]);
    delete[] c->prof_ev;
    delete[] c->prof_flops;
    delete[] c->prof_bytes;
    delete[] c->prof_ms;
    delete[] c->prof_shape;
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete[] c->blk;
    delete[] c->sav;
    delete c;
}

extern "C" size_t pevit_arena_bytes(const pevit_ctx* c) { return c ? c->arena_bytes : 0; }
extern "C" size_t pevit_workspace_bytes(const pevit_ctx* c, int batch) {
    if (!c || batch <= 0) return 0;
    size_t total = 0;
    layout_workspace(const_cast<pevit_ctx*>(c), batch, nullptr, &total, nullptr);
    return total;
}
extern "C" size_t pevit_num_tower_params(const pevit_ctx* c) { return c ? c->n_tower : 0; }
extern "C" size_t pevit_num_params(const pevit_ctx* c) { return c ? c->n_total : 0; }
extern "C" size_t pevit_param_layer_offset(const pevit_ctx* c, int layer) {
    if (!c) return 0;
    if (layer >= c->L) return c->n_tower;
    return c->p_layer0 + c->p_layer_stride * (size_t)(layer < 0 ? 0 : layer);
}

extern "C" int pevit_param_grad_mask(const pevit_ctx* c, unsigned char* m, size_t n) {
    if (!c || !m || n != c->n_total) { pevit_set_error("param_grad_mask: size mismatch"); return -1; }
    memset(m, 1, n);
    if (c->d.method == PEVIT_KADAPTATION) {
        const size_t E = c->E;
        for (int l = 0; l < c->L; ++l) {
            const size_t base = c->p_layer0 + c->p_layer_stride * l;
            memset(m + base + 2 * E, 0, 2 * E);       // v_proj_adapter1_left/right
        }
    }
    return 0;
}

extern "C" int pevit_bind(pevit_ctx* c, void* arena, size_t arena_bytes, void* ws, size_t ws_bytes, int max_batch) {
    if (!c || !arena || !ws) { pevit_set_error("bind: null argument"); return -1; }
    if (arena_bytes < c->arena_bytes) { pevit_set_error("bind: arena too small (%zu < %zu)", arena_bytes, c->arena_bytes); return -1; }
    const size_t need = pevit_workspace_bytes(c, max_batch);
    if (ws_bytes < need) { pevit_set_error("bind: workspace too small (%zu < %zu)", ws_bytes, need); return -1; }
    if (((uintptr_t)arena | (uintptr_t)ws) & 255) { pevit_set_error("bind: buffers must be 256-byte aligned"); return -1; }
    c->arena = (char*)arena; c->ws = (char*)ws; c->max_batch = max_batch; c->ws_bytes_for_max = need;
    // the stream-K flags must read 0 before the first launch (every launch leaves them 0 again)
    HIP_OK(hipMemset(c->ws, 0, (size_t)(PEVIT_SK_MAX_SLOTS + 2) * 4));
    return 0;
}

extern "C" int pevit_set_params(pevit_ctx* c, float* params, float* grads, float* mom, const unsigned char* mask) {
    if (!c || !params || !grads) { pevit_set_error("set_params: null argument"); return -1; }
    c->params = params; c->grads = grads; c->mom = mom; c->grad_mask = mask;
    return 0;
}

// ------------------------------------------------------------------------------------
extern "C" int pevit_load_block(pevit_ctx* c, void* stream, int l, const float* in_w, const float* in_b,
                                const float* out_w, const float* out_b, const float* ln1w, const float* ln1b,
                                const float* fc_w, const float* fc_b, const float* pr_w, const float* pr_b,
                                const float* ln2w, const float* ln2b) {
    if (!c || !c->arena) { pevit_set_error("load_block: context not bound"); return -1; }
    if (l < 0 || l >= c->L) { pevit_set_error("load_block: layer %d out of range", l); return -1; }
    hipStream_t s = (hipStream_t)stream;
    const BlockArena& b = c->blk[l];
    const size_t E = c->E;
    char* A = c->arena;
    // the 1/sqrt(head_dim) of model.py:786-787 is folded into the q rows (exact: a power of two)
    if (c->fp8) {
        typedef unsigned char u8;
        const int e = (int)E;
        HIP_OK(hipMemsetAsync(A + b.wqkv, 0, align_up(3 * E, 256) * E, s));
        HIP_OK(hipMemsetAsync(A + b.wpan, 0, 128 * E * 2, s));
        HIP_OK(hipMemsetAsync(A + b.wo, 0, align_up(E, 256) * E, s));
        HIP_OK(hipMemsetAsync(A + b.woT, 0, align_up(E, 256) * E, s));
        HIP_OK(hipMemsetAsync(A + b.wfc, 0, align_up(4 * E, 256) * E, s));
        HIP_OK(hipMemsetAsync(A + b.wfcT, 0, align_up(E, 256) * 4 * E, s));
        HIP_OK(hipMemsetAsync(A + b.wpr, 0, align_up(E, 256) * 4 * E, s));
        HIP_OK(hipMemsetAsync(A + b.wprT, 0, align_up(4 * E, 256) * E, s));
        CHECK(pevit_launch_quant_rows_fp8(in_w, 3 * e, e, at<u8>(A, b.wqkv), e, at<float>(A, b.sqkv), e, 0.125f, s));
        CHECK(pevit_launch_quant_rows_fp8(out_w, e, e, at<u8>(A, b.wo), e, at<float>(A, b.so), 0, 1.0f, s));
        CHECK(pevit_launch_quant_transpose_fp8(out_w, e, e, at<float>(A, b.so), at<u8>(A, b.woT), e, 0, 1.0f, s));
        CHECK(pevit_launch_quant_rows_fp8(fc_w, 4 * e, e, at<u8>(A, b.wfc), e, at<float>(A, b.sfc), 0, 1.0f, s));
        CHECK(pevit_launch_quant_transpose_fp8(fc_w, 4 * e, e, at<float>(A, b.sfc), at<u8>(A, b.wfcT), 4 * e, 0, 1.0f, s));
        CHECK(pevit_launch_quant_rows_fp8(pr_w, e, 4 * e, at<u8>(A, b.wpr), 4 * e, at<float>(A, b.spr), 0, 1.0f, s));
        CHECK(pevit_launch_quant_transpose_fp8(pr_w, e, 4 * e, at<float>(A, b.spr), at<u8>(A, b.wprT), e, 0, 1.0f, s));
        // QKV backward (bf16): the transposed copy holds the DE-QUANTISED weights, exactly representable in bf16
        HIP_OK(hipMemsetAsync(A + b.wqkvT, 0, E * (size_t)c->NQ * 2, s));
        const size_t skip = align_up((size_t)(PEVIT_SK_MAX_SLOTS + 2) * 4, 256);    // the stream-K flags stay zero
        // 3E*E floats of the bound workspace serve as packing scratch: whatever activations a previous forward saved there
        // are overwritten, so a backward through them is refused from here on (saved_batch = 0), and the load must be
        // issued on the stream the engine trains on (include/pevit_hip.h: one stream per context)
        float* tmp = at<float>(c->ws, skip);
        c->saved_batch = 0; c->saved_kind = 0;
        if (skip + (size_t)3 * E * E * 4 > c->ws_bytes_for_max) { pevit_set_error("load_block: workspace too small for the fp8 packing scratch"); return -1; }
        CHECK(pevit_launch_dequant_rows_fp8(at<u8>(A, b.wqkv), e, at<float>(A, b.sqkv), 3 * e, e, tmp, s));
        CHECK(pevit_launch_transpose_bf16(tmp, 3 * e, e, at<bf16>(A, b.wqkvT), c->NQ, 0, 1.0f, s));
    } else {
        const int f = c->f32;
        HIP_OK(hipMemsetAsync(A + b.wqkv, 0, (size_t)c->NQpad * E * c->es, s));
        CHECK(pevit_launch_cast_bf16(in_w, at<bf16>(A, b.wqkv), E * E, 0.125f, s, f));
        CHECK(pevit_launch_cast_bf16(in_w + E * E, eadv(c, at<bf16>(A, b.wqkv), E * E), 2 * E * E, 1.0f, s, f));
        HIP_OK(hipMemsetAsync(A + b.wqkvT, 0, E * (size_t)c->NQ * c->es, s));
        CHECK(pevit_launch_transpose_bf16(in_w, 3 * (int)E, (int)E, at<bf16>(A, b.wqkvT), c->NQ, (int)E, 0.125f, s, f));
        CHECK(pevit_launch_cast_bf16(out_w, at<bf16>(A, b.wo), E * E, 1.0f, s, f));
        CHECK(pevit_launch_transpose_bf16(out_w, (int)E, (int)E, at<bf16>(A, b.woT), (int)E, 0, 1.0f, s, f));
        CHECK(pevit_launch_cast_bf16(fc_w, at<bf16>(A, b.wfc), 4 * E * E, 1.0f, s, f));
        CHECK(pevit_launch_transpose_bf16(fc_w, 4 * (int)E, (int)E, at<bf16>(A, b.wfcT), 4 * (int)E, 0, 1.0f, s, f));
        CHECK(pevit_launch_cast_bf16(pr_w, at<bf16>(A, b.wpr), 4 * E * E, 1.0f, s, f));
        CHECK(pevit_launch_transpose_bf16(pr_w, (int)E, 4 * (int)E, at<bf16>(A, b.wprT), (int)E, 0, 1.0f, s, f));
    }
    // biases and LN affines stay f32; the q third of in_proj_bias carries the same 1/8
    HIP_OK(hipMemcpyAsync(A + b.bqkv, in_b, 3 * E * 4, hipMemcpyDeviceToDevice, s));
    CHECK(pevit_launch_scale_f32(at<float>(A, b.bqkv), E, 0.125f, s));
    HIP_OK(hipMemcpyAsync(A + b.bo, out_b, E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + b.bfc, fc_b, 4 * E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + b.bpr, pr_b, E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + b.ln1w, ln1w, E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + b.ln1b, ln1b, E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + b.ln2w, ln2w, E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + b.ln2b, ln2b, E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemsetAsync(A + b.q32, 0, E * 64 * 4, s));
    HIP_OK(hipMemsetAsync(A + b.qT, 0, 64 * E * c->es, s));
    HIP_OK(hipMemsetAsync(A + b.q16, 0, E * 64 * 2, s));
    return 0;
}

// ------------------------------------------------------------------------------------
namespace {

int check_ready(pevit_ctx* c, int B, const char* who) {
    if (!c || !c->arena || !c->ws) { pevit_set_error("%s: context not bound", who); return -1; }
    if (B <= 0 || B > c->max_batch) { pevit_set_error("%s: batch %d outside [1,%d]", who, B, c->max_batch); return -1; }
    if (c->d.method != PEVIT_NONE && (!c->params || !c->grads)) { pevit_set_error("%s: parameters not set", who); return -1; }
    return 0;
}

AdapterPanels panels(pevit_ctx* c, int l) {
    const BlockArena& b = c->blk[l];
    AdapterPanels p;
    p.w_aug_rows = c->fp8 ? at<bf16>(c->arena, b.wpan) : eadv(c, at<bf16>(c->arena, b.wqkv), (size_t)3 * c->E * c->E);
    p.ldw = c->E;
    p.wT_aug_cols = eadv(c, at<bf16>(c->arena, b.wqkvT), 3 * (size_t)c->E);
    p.ldwT = c->NQ;
    p.q32 = at<float>(c->arena, b.q32);
    p.qT = at<bf16>(c->arena, b.qT);
    p.q16 = at<bf16>(c->arena, b.q16);
    return p;
}

// rebuild the bf16 adapter panels of every layer from the f32 master parameters (one launch)
int prep_adapters(pevit_ctx* c, hipStream_t s) {
    const size_t E = c->E;
    LayerStrides st;
    st.arena_bytes = c->L > 1 ? c->blk[1].wqkv - c->blk[0].wqkv : 0;
    st.param_floats = c->p_layer_stride;
    const float* lp = c->params + c->p_layer0;
    if (c->d.method == PEVIT_KADAPTATION) {
        const float* r = c->params;
        CHECK(pevit_launch_prep_kadapt(r, r + 1024, r + 2048, r + 3072, lp, lp + E, panels(c, 0), c->E, c->ascale, c->L, st, s, c->f32));
    } else if (c->d.method == PEVIT_LORA) {
        const size_t rE = (size_t)c->d.lora_rank * E;
        CHECK(pevit_launch_prep_lora(lp, lp + rE, lp + 2 * rE, lp + 3 * rE, c->d.lora_rank, panels(c, 0), c->E, c->ascale,
                                     c->L, st, s, c->f32));
    } else if (post_mlp(c)) {
        const BlockArena& b0 = c->blk[0];
        BottleneckPanels bp{at<bf16>(c->arena, b0.wd), at<bf16>(c->arena, b0.wdT), at<bf16>(c->arena, b0.wu),
                            at<bf16>(c->arena, b0.wuT)};
        if (c->d.method == PEVIT_ADAPTER)
            CHECK(pevit_launch_prep_adapter(lp + c->o_dw, lp + c->o_uw, bp, c->E, c->L, st, s, c->f32));
        else
            CHECK(pevit_launch_prep_compacter(at<float>(c->arena, c->a_phm), lp + c->o_dWl, lp + c->o_dWr, lp + c->o_uWl,
                                              lp + c->o_uWr, bp, c->E, c->L, st, s, c->f32));
    }
    return 0;
}

// ---- optional per-launch timing: HIP events on the caller's stream around a launch (pevit_profile_begin / _end) ----
// GEMM launches are always recorded while profiling is on; the HBM-bound kernels of the step (LayerNorm, attention, the low-rank
// adapter kernels, ...) only with pevit_tune(ctx, "profile_all", 1), so that the GEMM-family measurement keeps its own cadence.
// Non-GEMM records carry shape[0] = 100 + kind (PEVIT_PROF_* in pevit_hip.h), flops 0 and the algorithmic bytes of the launch.
int prof_open(pevit_ctx* c, hipStream_t s, bool is_gemm) {
    if (!c->prof_on || c->prof_n >= c->prof_cap || (!is_gemm && !c->prof_all)) return -1;
    (void)hipEventRecord(c->prof_ev[2 * c->prof_n], s);
    return c->prof_n;
}
void prof_close(pevit_ctx* c, hipStream_t s, int slot, double flops, double bytes, int s0, int s1, int s2, int s3) {
    if (slot < 0) return;
    (void)hipEventRecord(c->prof_ev[2 * slot + 1], s);
    c->prof_flops[slot] = flops; c->prof_bytes[slot] = bytes;
    int* sh = c->prof_shape + 4 * slot;
    sh[0] = s0; sh[1] = s1; sh[2] = s2; sh[3] = s3;
    c->prof_n = slot + 1;
}
// CHECK() of a non-GEMM launch, bracketed when "profile_all" is on: kind = PEVIT_PROF_*, bytes = what the launch must move
#define PROF(c, s, kind, rows, bytes, call)                                                  \
    do {                                                                                     \
        const int _slot = prof_open(c, s, false);                                            \
        const int _rc = (call);                                                              \
        prof_close(c, s, _slot, 0.0, (double)(bytes), 100 + (kind), (int)(rows), 0, 0);      \
        if (_rc != 0) return -1;                                                             \
    } while (0)

// every GEMM of the step goes through here so that it can be bracketed with HIP events
int gemm(pevit_ctx* c, int epi, const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    if (c->sk_slots && c->ws) {
        p.sk_flag = at<unsigned>(c->ws, c->w_skflag); p.sk_slab = at<float>(c->ws, c->w_skslab); p.sk_slots = c->sk_slots;
    }
    const int slot = prof_open(c, s, true);
    const int rc = c->f32 ? pevit_launch_gemm_f32(epi, p, s) : pevit_launch_gemm(epi, p, c->tune, s);
    if (slot >= 0) {
        // every operand read once, every result written once (the minimum any schedule must move)
        const double mn = (double)p.M * (double)p.N;
        const double bytes = 2.0 * ((double)p.M + (double)p.N) * (double)p.K + (p.bias ? 4.0 * p.N : 0.0) +
                             mn * ((p.resid ? 4.0 : 0.0) + (p.aux ? 2.0 : 0.0) + (p.outf ? 4.0 : 0.0) +
                                   (p.outf2 ? 4.0 : 0.0) + (p.outb ? 2.0 : 0.0) + (p.outb2 ? 2.0 : 0.0));
        prof_close(c, s, slot, 2.0 * (double)p.M * (double)p.N * (double)p.K, bytes, epi, p.M, p.N, p.K);
    }
    return rc;
}

GemmParams gp(const bf16* A, int lda, const bf16* B, int ldb, int Nb, int M, int N, int K) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.Nb = Nb; p.M = M; p.N = N; p.K = K;
    return p;
}

// frozen weight operand of layer-l products: bf16, or e4m3 codes + channel scales (ctx->fp8).  `scale_off` = arena offset
// of the per-output-channel scales for the FORWARD products, 0 for the dX products (their scales ride on the A operand).
GemmParams gpw(const pevit_ctx* c, const bf16* A, int lda, size_t w_off, int ldb, int Nb, int M, int N, int K, size_t scale_off) {
    GemmParams p = gp(A, lda, at<bf16>(c->arena, w_off), ldb, c->fp8 ? (int)align_up((size_t)Nb, 256) : Nb, M, N, K);
    if (c->fp8) {
        p.b_fp8 = 1;
        p.bscale = scale_off ? at<float>(c->arena, scale_off) : nullptr;
    }
    return p;
}

// forward of the L residual blocks on internal (batch-major) rows.  x0 -> sav[0].x_in must
// already hold the input; the output lands in ws + w_xfinal.
// cls_only: the caller consumes only the class token of the last block (VisionTransformer.forward,
// model.py:1046) -- everything of the last block that sits after the attention core is then
// evaluated on the B class-token rows only (identical results, ~6 % fewer FLOPs per step).
int blocks_forward(pevit_ctx* c, hipStream_t s, int B, bool cls_only, int l_lo = 0, int l_hi = -1) {
    if (l_hi < 0) l_hi = c->L;
    const int E = c->E, T = B * c->N, H = c->H, N = c->N;
    cls_only = cls_only && !post_mlp(c);
    char* W = c->ws; char* A = c->arena;
    const bool site = attention_site(c);
    if (site || post_mlp(c)) CHECK(prep_adapters(c, s));
    for (int l = l_lo; l < l_hi; ++l) {
        const BlockArena& b = c->blk[l];
        const LayerSaved& v = c->sav[l];
        float* x_in = at<float>(W, v.x_in);
        float* x_mid = at<float>(W, v.x_mid);
        float* x_out = (l + 1 < c->L) ? at<float>(W, c->sav[l + 1].x_in) : at<float>(W, c->w_xfinal);
        bf16* qkv = at<bf16>(W, v.qkv);
        const size_t plane = (size_t)T * E;
        // x = x + attn(ln_1(x))                                         model.py:973
        unsigned char* a8 = c->fp8act ? at<unsigned char>(W, c->w_a8) : nullptr;
        unsigned char* attn8 = c->fp8act ? at<unsigned char>(W, c->w_attn8) : nullptr;
        PROF(c, s, PEVIT_PROF_LN_FWD, T, (double)T * E * (4 + c->es),
             pevit_launch_ln_fwd(x_in, at<float>(A, b.ln1w), at<float>(A, b.ln1b), T, E, at<bf16>(W, v.xn1), nullptr,
                                 at<float>(W, v.mean1), at<float>(W, v.rstd1), s, 0, c->f32, a8));
        if (!c->fp8) {
            GemmParams p = gp(at<bf16>(W, v.xn1), E, at<bf16>(A, b.wqkv), E, c->NQpad, T, site ? c->NQ : 3 * E, E);
            p.bias = at<float>(A, b.bqkv); p.outb = qkv; p.head_stride = plane; p.outf = at<float>(W, v.t); p.ldo = 64;
            p.E = E; p.H = H; p.Ntok = N;
            CHECK(gemm(c, EPI_QKV_HEADS, p, s));
        } else {
            // fp8 codes for the 3E frozen rows; the 64 trainable adapter rows stay bf16: as the bf16 tail of the same launch where the
            // product runs on the staggered 8-wave kernel (round 4: the separate t = xn P product was the whole 1-2 % by which
            // the fp8 format trailed bf16), as a small product of their own otherwise
            GemmParams p = gpw(c, at<bf16>(W, v.xn1), E, b.wqkv, E, 3 * E, T, 3 * E, E, b.sqkv);
            if (a8) { p.A = reinterpret_cast<const bf16*>(a8); p.a_fp8 = 1; }
            p.bias = at<float>(A, b.bqkv); p.outb = qkv; p.head_stride = plane; p.E = E; p.H = H; p.Ntok = N;
            bool tail = false;
            if (site && !a8 && c->fp8_tail) {
                GemmParams m = p;
                m.N = c->NQ; m.B2 = at<bf16>(A, b.wpan); m.ldb2 = E; m.Nb2 = 128; m.n_fp8 = 3 * E; m.outf = at<float>(W, v.t); m.ldo = 64;
                if (pevit_gemm_mixed_ok(m, c->tune)) { p = m; tail = true; }
            }
            CHECK(gemm(c, EPI_QKV_HEADS, p, s));
            if (site && !tail) {
                GemmParams q = gp(at<bf16>(W, v.xn1), E, at<bf16>(A, b.wpan), E, 128, T, 64, E);
                q.outf = at<float>(W, v.t); q.ldo = 64;
                CHECK(gemm(c, EPI_F32, q, s));
            }
        }
        const float* dbias = nullptr;
        if (c->d.method == PEVIT_KADAPTATION) dbias = c->params + c->p_layer0 + c->p_layer_stride * l + 4 * (size_t)E;
        // delta-add and the attention core as ONE launch where a run of heads owns whole reference rows of the raw reshape
        // (attn_delta.hip: N <= 64; ViT-B/32), otherwise delta_add + attn_fwd
        // ... unless its one-workgroup-per-CU runs leave between a quarter and three quarters of the chip empty (measured at batch
        // 64: 128 runs for 256 CUs, the two kernels are 0.6 % of the step faster; fused_attn_delta = 2 forces the fused form)
        const int ad_hpw = pevit_attn_delta_hpw(B, H, N);
        const int ad_runs = ad_hpw > 0 ? (B * H + ad_hpw - 1) / ad_hpw : 0;
        const bool ad_fill = ad_hpw > 0 && (c->fused_attn_delta > 1 || 4 * ad_runs >= 3 * pevit_num_cus() || 4 * ad_runs <= pevit_num_cus());
        const bool fused_ad = site && c->fused_attn_delta && !c->f32 && !attn8 && ad_fill;
        if (fused_ad) {
            PROF(c, s, PEVIT_PROF_ATTN_FWD_DELTA, T, (double)T * E * (3 + 2 + 1) * 2 + (double)T * 64 * 4 + (double)B * H * N * 4,   // q, k, v in; q', v', out
                 pevit_launch_attn_fwd_delta(qkv, qkv + plane, qkv + 2 * plane, at<float>(W, v.t), at<bf16>(A, b.q16), dbias, c->ascale,
                                             at<bf16>(W, v.attn_out), E, at<float>(W, v.lse), B, H, N, s));
        } else {
        if (site) {
            PROF(c, s, PEVIT_PROF_DELTA_ADD, T, (double)T * E * 4 * c->es + (double)T * 64 * 4,     // q and v read + written, t read
                 pevit_launch_delta_add(qkv, eadv(c, qkv, 2 * plane), at<float>(W, v.t), at<float>(A, b.q32), at<bf16>(A, b.q16), dbias,
                                        c->ascale, B, N, E, s, c->f32));
        }
        if (c->f32)
            CHECK(pevit_launch_attn_fwd_f32((const float*)qkv, (const float*)eadv(c, qkv, plane), (const float*)eadv(c, qkv, 2 * plane),
                                            at<float>(W, v.attn_out), E, at<float>(W, v.lse), B, H, N, s));
        else
            PROF(c, s, PEVIT_PROF_ATTN_FWD, T, (double)T * E * 4 * 2 + (double)B * H * N * 4,
                 pevit_launch_attn_fwd(qkv, qkv + plane, qkv + 2 * plane, at<bf16>(W, v.attn_out), E, at<float>(W, v.lse), B,
                                       H, N, s, attn8));
        }
        // rows of the tail of this block: all T, or (last block, cls_only) the B class-token rows, which
        // sit N*E elements apart in every [T][E] buffer
        const bool cls = cls_only && l == c->L - 1;
        const int R = cls ? B : T;
        const int rs = cls ? N * E : E;            // row stride of [T][E] buffers
        {
            GemmParams p = gpw(c, at<bf16>(W, v.attn_out), rs, b.wo, E, E, R, E, E, b.so);
            if (attn8) { p.A = reinterpret_cast<const bf16*>(attn8); p.a_fp8 = 1; }
            p.bias = at<float>(A, b.bo); p.resid = x_in; p.ldr = rs; p.outf = x_mid; p.ldo = rs;
            CHECK(gemm(c, EPI_BIAS_RESID_F32, p, s));
        }
        // x = x + mlp(ln_2(x))                                          model.py:974
        PROF(c, s, PEVIT_PROF_LN_FWD, R, (double)R * E * (4 + c->es),
             pevit_launch_ln_fwd(x_mid, at<float>(A, b.ln2w), at<float>(A, b.ln2b), R, E, at<bf16>(W, c->w_xn2), nullptr,
                                 at<float>(W, v.mean2), at<float>(W, v.rstd2), s, (size_t)rs, c->f32, a8));
        {
            GemmParams p = gpw(c, at<bf16>(W, c->w_xn2), E, b.wfc, E, 4 * E, R, 4 * E, E, b.sfc);
            if (a8) { p.A = reinterpret_cast<const bf16*>(a8); p.a_fp8 = 1; p.out2_fp8 = 1; }     // gelu(h) leaves as e4m3 codes
            p.bias = at<float>(A, b.bfc); p.outb = at<bf16>(W, v.h); p.ldob = 4 * E; p.outb2 = at<bf16>(W, c->w_g);
            p.ldob2 = 4 * E;
            CHECK(gemm(c, EPI_BIAS_GELU, p, s));
        }
        if (cls) {
            GemmParams p = gpw(c, at<bf16>(W, c->w_g), 4 * E, b.wpr, 4 * E, E, R, E, 4 * E, b.spr);
            if (a8) p.a_fp8 = 1;
            p.bias = at<float>(A, b.bpr); p.resid = x_mid; p.ldr = rs; p.outf = x_out; p.ldo = rs;
            CHECK(gemm(c, EPI_BIAS_RESID_F32, p, s));
            continue;
        }
        if (!post_mlp(c)) {
            GemmParams p = gpw(c, at<bf16>(W, c->w_g), 4 * E, b.wpr, 4 * E, E, T, E, 4 * E, b.spr);
            if (a8) p.a_fp8 = 1;
            p.bias = at<float>(A, b.bpr); p.resid = x_mid; p.ldr = E; p.outf = x_out; p.ldo = E;
            CHECK(gemm(c, EPI_BIAS_RESID_F32, p, s));
        } else {
            // x = x + [h + up(act(down(LN_a(h))))]         adapter_model.py:330-336 / compacter_model.py:497-503
            const float* lp = c->params + c->p_layer0 + c->p_layer_stride * l;
            if (c->adapter_fused && !c->f32 && !c->fused_bn && pevit_adapter_fused_ok(E)) {
                // two launches (adapter_fused.hip): c_proj writes its accumulators once (the bias joins in the adapter kernel), then
                // LayerNorm -> down -> activation -> up -> residual for 32 rows per workgroup
                GemmParams p = gp(at<bf16>(W, c->w_g), 4 * E, at<bf16>(A, b.wpr), 4 * E, E, T, E, 4 * E);
                p.outf = at<float>(W, v.hf32); p.ldo = E;
                CHECK(gemm(c, EPI_F32, p, s));
                PROF(c, s, PEVIT_PROF_ADAPTER_FWD, T, (double)T * E * (4 + 4 + 4 + 2) + (double)T * 64 * 4,
                     pevit_launch_adapter_fwd(c->d.method == PEVIT_ADAPTER ? 0 : 1, at<float>(W, v.hf32), at<float>(A, b.bpr), x_mid, lp + c->o_nw,
                                              lp + c->o_nb, at<bf16>(A, b.wd), lp + c->o_db, at<bf16>(A, b.wu), lp + c->o_ub, at<bf16>(W, v.z),
                                              at<float>(W, v.mean_a), at<float>(W, v.rstd_a), at<bf16>(W, v.act), at<bf16>(W, v.apre), x_out,
                                              T, E, s));
                continue;
            }
            float* ytmp = at<float>(W, c->w_dxn);           // x_mid + h ; scratch that is free during the forward pass
            {
                GemmParams p = gp(at<bf16>(W, c->w_g), 4 * E, at<bf16>(A, b.wpr), 4 * E, E, T, E, 4 * E);
                p.bias = at<float>(A, b.bpr); p.resid = x_mid; p.ldr = E; p.outf = ytmp; p.ldo = E;
                p.outf2 = at<float>(W, v.hf32); p.ldo2 = E;
                CHECK(gemm(c, EPI_BIAS_RESID_KEEP, p, s));
            }
            CHECK(pevit_launch_ln_fwd(at<float>(W, v.hf32), lp + c->o_nw, lp + c->o_nb, T, E, at<bf16>(W, v.z), nullptr,
                                      at<float>(W, v.mean_a), at<float>(W, v.rstd_a), s, 0, c->f32));
            if (c->fused_bn && !c->f32) {
                // down -> activation -> up (+ bias + x_mid + h) in one launch (adapter.hip bottleneck_pair_kernel)
                CHECK(pevit_launch_bottleneck_pair(c->d.method == PEVIT_ADAPTER ? 0 : 1, at<bf16>(W, v.z), E, at<bf16>(A, b.wd), lp + c->o_db,
                                                   nullptr, at<bf16>(W, v.act), at<bf16>(W, v.apre), at<bf16>(A, b.wu), lp + c->o_ub, ytmp,
                                                   x_out, T, E, s));
            } else {
            {
                    GemmParams p = gp(at<bf16>(W, v.z), E, at<bf16>(A, b.wd), E, 64, T, 64, E);
                    p.bias = lp + c->o_db;
                    if (c->d.method == PEVIT_ADAPTER) {
                        p.outb = at<bf16>(W, v.act); p.ldob = 64;
                        CHECK(gemm(c, EPI_BIAS_RELU_BF16, p, s));
                    } else {
                        p.outb = at<bf16>(W, v.apre); p.ldob = 64; p.outb2 = at<bf16>(W, v.act); p.ldob2 = 64;
                        CHECK(gemm(c, EPI_BIAS_GELUNEW, p, s));
                    }
                }
                {
                    GemmParams p = gp(at<bf16>(W, v.act), 64, at<bf16>(A, b.wu), 64, E, T, E, 64);
                    p.bias = lp + c->o_ub; p.resid = ytmp; p.ldr = E; p.outf = x_out; p.ldo = E;
                    CHECK(gemm(c, EPI_BIAS_RESID_F32, p, s));
                }
            }
        }
    }
    return 0;
}

// backward of the blocks.  On entry ws+w_dxa holds dL/dx_final (f32) and ws+w_dyb its bf16 copy.
// On exit ws+w_dxa holds dL/dx_0 if need_dx0.
// cls_only mirrors blocks_forward: on entry only the class-token rows of dxa / dyb are defined (and
// read); dxb and dO must have been zeroed by the caller.
// Layers l_hi-1 .. l_lo are processed (the whole tower: L, 0) and the adapter gradients of exactly these layers are
// reduced and chained onto the reference's tensors at the end -- data parallelism runs the tower in two halves so that
// the all-reduce of the upper half's gradients overlaps the backward of the lower half.
// the residual GRADIENT stream lives in bf16 only (pevit_ctx::gstream16): attention-site adapters, and the post-MLP adapters on
// their fused kernels; bf16 / fp8 weights (never the f32 verification mode)
bool gstream16_on(const pevit_ctx* c) {
    if (!c->gstream16 || c->f32 || !c->dx_stored) return false;
    if (attention_site(c)) return true;
    return post_mlp(c) && c->adapter_fused && !c->fused_bn && pevit_adapter_fused_ok(c->E);
}

int blocks_backward(pevit_ctx* c, hipStream_t s, int B, bool need_dx0, bool cls_only, int l_hi, int l_lo) {
    const int E = c->E, T = B * c->N, H = c->H, N = c->N;
    cls_only = cls_only && !post_mlp(c);
    char* W = c->ws; char* A = c->arena;
    const bool site = attention_site(c);
    const int chunks = pevit_lowrank_chunks(T);
    float* dxa = at<float>(W, c->w_dxa);
    float* dxb = at<float>(W, c->w_dxb);
    bf16* dyb = at<bf16>(W, c->w_dyb);
    float* dxn = at<float>(W, c->w_dxn);
    bf16* dqkv = at<bf16>(W, c->w_dqkv);
    const bool use_side = c->side_stream && site;
    bool side_pending = false;
    const bool combo = c->lowrank_combo && site && !c->f32 && !use_side;
    const bool gs16 = gstream16_on(c);
    int prev_layer = -1, u_par = 0;
    float* u_last = nullptr;
    int tn_pend = -1, tn_par = 0;          // post-MLP adapters: layer whose d W_down product is still owed, and the d pre buffer in turn
    const bf16* tn_pend_dpre = nullptr;
    if (use_side && !c->side) {
        HIP_OK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    for (int l = l_hi - 1; l >= l_lo; --l) {
        const BlockArena& b = c->blk[l];
        const LayerSaved& v = c->sav[l];
        bf16* qkv = at<bf16>(W, v.qkv);
        const size_t plane = (size_t)T * E;
        const bf16* mlp_dy = dyb;          // upstream gradient of the MLP output (bf16)
        if (post_mlp(c)) {
            // out = x_mid + h + up(act(down(LN_a(h)))) :  dx_out (dxa, dyb) flows to x_mid, to h, and into the adapter
            const float* lp = c->params + c->p_layer0 + c->p_layer_stride * l;
            const bool fused_ad = c->adapter_fused && !c->f32 && !c->fused_bn && pevit_adapter_fused_ok(E);
            const bool fold = fused_ad && c->adapter_tn_fold;      // both weight-gradient products inside the backward launch
            bf16* dpre = at<bf16>(W, (fold && tn_par) ? c->w_dpre2 : c->w_dpre);
            // d W_up[e][j] = sum_r dx_out[r][e] act[r][j] ; d b_up = colsum(dx_out)
            if (fold) {}
            else if (c->f32)
                CHECK(pevit_launch_tn_gemm64_f32((const float*)dyb, E, at<float>(W, v.act), 64, at<float>(W, c->w_tnU + (size_t)l * c->tn_layer),
                                                 nullptr, nullptr, T, E, s));
            else
                CHECK(pevit_launch_tn_gemm64(dyb, E, at<bf16>(W, v.act), 64, at<float>(W, c->w_tnU + (size_t)l * c->tn_layer),
                                             nullptr, nullptr, T, E, s));
            if (fused_ad) {
                const int pl = tn_pend >= 0 ? tn_pend : l;       // the layer whose d W_down product this launch carries (if any)
                // d pre, d z and the LayerNorm backward with its affine-gradient column sums in one launch (adapter_fused.hip); the
                // forward pass left the c_proj accumulators WITHOUT their bias in hf32
                PROF(c, s, PEVIT_PROF_ADAPTER_BWD, T, (double)T * E * (2 + 4 + 4 + 2) + (double)T * 64 * 4,
                     pevit_launch_adapter_bwd(c->d.method == PEVIT_ADAPTER ? 0 : 1, dyb, gs16 ? nullptr : dxa, at<bf16>(A, b.wuT),
                                              c->d.method == PEVIT_ADAPTER ? at<bf16>(W, v.act) : at<bf16>(W, v.apre), at<bf16>(A, b.wdT),
                                              at<float>(W, v.hf32), at<float>(A, b.bpr), at<float>(W, v.mean_a), at<float>(W, v.rstd_a),
                                              lp + c->o_nw, dpre, at<bf16>(W, c->w_dhb), at<float>(W, c->w_lnp + (size_t)l * c->lnp_layer), T, E, s,
                                              fold ? dyb : nullptr, at<bf16>(W, v.act), at<float>(W, c->w_tnU + (size_t)l * c->tn_layer),
                                              (fold && tn_pend >= 0) ? at<bf16>(W, c->sav[pl].z) : nullptr, tn_pend_dpre,
                                              at<float>(W, c->w_tnD + (size_t)pl * c->tn_layer), at<float>(W, c->w_csy + (size_t)pl * c->csy_layer),
                                              c->adapter_tn_fold > 1 ? c->adapter_tn_fold : 0));
                if (fold) { tn_pend = l; tn_pend_dpre = dpre; tn_par ^= 1; }
            } else if (c->fused_bn && !c->f32) {
                // d pre = (dx_out W_up) * act'(saved) ; d z = d pre W_down, one launch
                CHECK(pevit_launch_bottleneck_pair(c->d.method == PEVIT_ADAPTER ? 2 : 3, dyb, E, at<bf16>(A, b.wuT), nullptr,
                                                   c->d.method == PEVIT_ADAPTER ? at<bf16>(W, v.act) : at<bf16>(W, v.apre), dpre, nullptr,
                                                   at<bf16>(A, b.wdT), nullptr, nullptr, dxn, T, E, s));
            } else {
            {   // d act = dx_out W_up ; d pre = d act * act'(pre)
                GemmParams p = gp(dyb, E, at<bf16>(A, b.wuT), E, 64, T, 64, E);
                p.outb = dpre; p.ldob = 64; p.ldaux = 64;
                if (c->d.method == PEVIT_ADAPTER) { p.aux = at<bf16>(W, v.act); CHECK(gemm(c, EPI_DRELU_BF16, p, s)); }
                else { p.aux = at<bf16>(W, v.apre); CHECK(gemm(c, EPI_DGELUNEW_BF16, p, s)); }
            }
            {   // d z = d pre W_down
                GemmParams p = gp(dpre, 64, at<bf16>(A, b.wdT), 64, E, T, E, 64);
                p.outf = dxn; p.ldo = E;
                CHECK(gemm(c, EPI_F32, p, s));
            }
            }
            // d W_down[j][e] = sum_r d pre[r][j] z[r][e] ; d b_down = colsum(d pre)
            if (fold) {}
            else if (c->f32)
                CHECK(pevit_launch_tn_gemm64_f32(at<float>(W, v.z), E, (const float*)dpre, 64, at<float>(W, c->w_tnD + (size_t)l * c->tn_layer),
                                                 nullptr, at<float>(W, c->w_csy + (size_t)l * c->csy_layer), T, E, s));
            else
                CHECK(pevit_launch_tn_gemm64(at<bf16>(W, v.z), E, dpre, 64, at<float>(W, c->w_tnD + (size_t)l * c->tn_layer), nullptr,
                                             at<float>(W, c->w_csy + (size_t)l * c->csy_layer), T, E, s));
            // d h = dx_out + LN_a-backward(d z) ; partial sums for d gamma_a, d beta_a
            if (!fused_ad)
                CHECK(pevit_launch_ln_bwd_affine(dxn, at<float>(W, v.hf32), at<float>(W, v.mean_a), at<float>(W, v.rstd_a), lp + c->o_nw,
                                                 dxa, nullptr, at<bf16>(W, c->w_dhb),
                                                 at<float>(W, c->w_lnp + (size_t)l * c->lnp_layer), T, E, s, c->f32));
            mlp_dy = at<bf16>(W, c->w_dhb);
            if (l == 0 && !need_dx0) break;     // nothing trainable below the first block's adapter
        }
        const bool cls = cls_only && l == c->L - 1;
        const int R = cls ? B : T;
        const int rs = cls ? N * E : E;            // row stride of [T][E] buffers
        // ---- MLP branch: d h = (dy W_proj) * gelu'(h) ; d xn2 = d h W_fc
        {
            // fp8: mlp_dy arrives with c_proj's channel scales folded in, and leaves with c_fc's (for the next product)
            GemmParams p = gpw(c, mlp_dy, rs, b.wprT, E, 4 * E, R, 4 * E, E, 0);
            p.aux = at<bf16>(W, v.h); p.ldaux = 4 * E; p.outb = at<bf16>(W, c->w_dh); p.ldob = 4 * E;
            if (c->fp8) p.oscale = at<float>(A, b.sfc);
            CHECK(gemm(c, EPI_DGELU_BF16, p, s));
        }
        {
            // the LN-input gradient leaves the GEMM in the activation storage type (bf16): LayerNorm backward is
            // HBM-bound, and this halves the bytes on both sides of the hand-over
            GemmParams p = gpw(c, at<bf16>(W, c->w_dh), 4 * E, b.wfcT, 4 * E, E, R, E, 4 * E, 0);
            if (c->dx_stored) { p.outb = reinterpret_cast<bf16*>(dxn); p.ldob = E; CHECK(gemm(c, EPI_BF16, p, s)); }
            else { p.outf = dxn; p.ldo = E; CHECK(gemm(c, EPI_F32, p, s)); }
        }
        // fp8: the bf16 copy feeds the out-projection backward, whose contraction runs over out_proj's output channels
        // dy (stored type or f32) + x + residual gradient read, f32 gradient + its stored copy written
        if (gs16)
            PROF(c, s, PEVIT_PROF_LN_BWD, R, (double)R * E * (c->es + 4 + c->es + c->es),
                 pevit_launch_ln_bwd(dxn, at<float>(W, v.x_mid), at<float>(W, v.mean2), at<float>(W, v.rstd2),
                                     at<float>(A, b.ln2w), reinterpret_cast<const float*>(dyb), nullptr, dyb, R, E, s, (size_t)rs,
                                     c->fp8 ? at<float>(A, b.so) : nullptr, 0, 1, 0, 1, c->fp8 ? at<float>(A, b.spr) : nullptr));
        else
        PROF(c, s, PEVIT_PROF_LN_BWD, R, (double)R * E * ((c->dx_stored ? c->es : 4) + 4 + 4 + 4 + c->es),
             pevit_launch_ln_bwd(dxn, at<float>(W, v.x_mid), at<float>(W, v.mean2), at<float>(W, v.rstd2),
                                 at<float>(A, b.ln2w), dxa, dxb, dyb, R, E, s, (size_t)rs,
                                 c->fp8 ? at<float>(A, b.so) : nullptr, c->f32, c->dx_stored));
        // ---- attention branch
        {
            GemmParams p = gpw(c, dyb, rs, b.woT, E, E, R, E, E, 0);
            p.outb = at<bf16>(W, c->w_dO); p.ldob = rs;
            CHECK(gemm(c, EPI_BF16, p, s));
        }
        // dqkv / u32 are about to be overwritten: the previous layer's gradient contraction must have read them
        if (side_pending) { HIP_OK(hipStreamWaitEvent(s, c->ev_join, 0)); side_pending = false; }
        if (c->f32)
            CHECK(pevit_launch_attn_bwd_f32((const float*)qkv, (const float*)eadv(c, qkv, plane), (const float*)eadv(c, qkv, 2 * plane),
                                            at<float>(W, v.attn_out), E, at<float>(W, c->w_dO), E, at<float>(W, v.lse), (float*)dqkv,
                                            c->NQ, B, H, N, s));
        else
            PROF(c, s, PEVIT_PROF_ATTN_BWD, T, (double)T * E * (N <= 64 ? 7 : 8) * 2 + (double)B * H * N * 4,     // q, k, v, (out: N > 64 only), dout in; dq, dk, dv out
                 pevit_launch_attn_bwd(qkv, qkv + plane, qkv + 2 * plane, at<bf16>(W, v.attn_out), E, at<bf16>(W, c->w_dO), E,
                                       at<float>(W, v.lse), dqkv, c->NQ, B, H, N, s, (cls && N <= 64) ? 1 : 0));
        if (site && combo) {
            // u, dQ_q, dQ_v, d bias of this layer and the dP of the layer before it in ONE launch (lowrank.hip lowrank_combo_kernel)
            float* u_cur = at<float>(W, u_par ? c->w_u32b : c->w_u32);
            const LayerSaved* pv = prev_layer >= 0 ? &c->sav[prev_layer] : nullptr;
            PROF(c, s, PEVIT_PROF_LOWRANK_BWD, T, (double)T * E * 3 * 2 + (double)T * 64 * 14 + (double)chunks * 4 * E * 32 * 4,
                 pevit_launch_lowrank_combo(1, pv ? 1 : 0, dqkv, c->NQ, at<bf16>(A, b.qT), u_cur, dqkv + 3 * E, at<float>(W, v.t),
                                            at<float>(W, c->w_partial + (size_t)l * c->partial_layer),
                                            at<float>(W, c->w_dbias + (size_t)l * c->dbias_layer),
                                            pv ? at<bf16>(W, pv->xn1) : nullptr, E, u_last,
                                            pv ? at<float>(W, c->w_partial + (size_t)prev_layer * c->partial_layer) : nullptr, B, H, N, E, s));
            u_last = u_cur; prev_layer = l; u_par ^= 1;
        } else if (site) {
            if (c->f32)
                CHECK(pevit_launch_lowrank_u_f32((const float*)dqkv, c->NQ, at<float>(A, b.q32), at<float>(W, c->w_u32),
                                                 (float*)eadv(c, dqkv, 3 * (size_t)E), B, H, N, E, s));
            else
                PROF(c, s, PEVIT_PROF_LOWRANK_U, T, (double)T * E * 2 * 2 + (double)T * 64 * 6,
                     pevit_launch_lowrank_u(dqkv, c->NQ, at<bf16>(A, b.qT), at<float>(W, c->w_u32), dqkv + 3 * E, B, H, N, E, s));
            // the token-contracted adapter gradients feed nothing before the end of the step: run them beside
            // the QKV-backward GEMM / LayerNorm backward / next layer's MLP GEMMs on the second stream
            hipStream_t gs = s;
            if (use_side) {
                HIP_OK(hipEventRecord(c->ev_fork, s));
                HIP_OK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
                gs = c->side;
            }
            if (c->f32)
                CHECK(pevit_launch_lowrank_grad_f32(at<float>(W, v.xn1), E, at<float>(W, c->w_u32), (const float*)dqkv, c->NQ,
                                                    at<float>(W, v.t), at<float>(W, c->w_partial + (size_t)l * c->partial_layer),
                                                    at<float>(W, c->w_dbias + (size_t)l * c->dbias_layer), chunks, B, H, N, E, gs));
            else
                PROF(c, gs, PEVIT_PROF_LOWRANK_GRAD, T, (double)T * E * 3 * 2 + (double)T * 64 * 8 + (double)chunks * 4 * E * 32 * 4,
                     pevit_launch_lowrank_grad(at<bf16>(W, v.xn1), E, at<float>(W, c->w_u32), dqkv, c->NQ, at<float>(W, v.t),
                                               at<float>(W, c->w_partial + (size_t)l * c->partial_layer),
                                               at<float>(W, c->w_dbias + (size_t)l * c->dbias_layer), chunks, B, H, N, E, gs, c->lowrank_xcd));
            if (use_side) { HIP_OK(hipEventRecord(c->ev_join, c->side)); side_pending = true; }
        }
        if (l > 0 || need_dx0) {
            GemmParams p = gp(dqkv, c->NQ, at<bf16>(A, b.wqkvT), c->NQ, E, T, E, site ? c->NQ : 3 * E);
            if (c->dx_stored) { p.outb = reinterpret_cast<bf16*>(dxn); p.ldob = E; CHECK(gemm(c, EPI_BF16, p, s)); }
            else { p.outf = dxn; p.ldo = E; CHECK(gemm(c, EPI_F32, p, s)); }
            // fp8: this bf16 copy is the upstream gradient of layer l-1's c_proj backward
            if (gs16)      // the f32 copy only where the caller asked for dx (the lowest block walked)
                PROF(c, s, PEVIT_PROF_LN_BWD, T, (double)T * E * (c->es + 4 + c->es + c->es),
                     pevit_launch_ln_bwd(dxn, at<float>(W, v.x_in), at<float>(W, v.mean1), at<float>(W, v.rstd1),
                                         at<float>(A, b.ln1w), reinterpret_cast<const float*>(dyb), (need_dx0 && l == l_lo) ? dxa : nullptr, dyb,
                                         T, E, s, 0, (c->fp8 && l > 0) ? at<float>(A, c->blk[l - 1].spr) : nullptr, 0, 1, cls ? N : 0, 1,
                                         c->fp8 ? at<float>(A, b.so) : nullptr));
            else
            PROF(c, s, PEVIT_PROF_LN_BWD, T, (double)T * E * ((c->dx_stored ? c->es : 4) + 4 + 4 + 4 + c->es),
                 pevit_launch_ln_bwd(dxn, at<float>(W, v.x_in), at<float>(W, v.mean1), at<float>(W, v.rstd1),
                                     at<float>(A, b.ln1w), dxb, dxa, dyb, T, E, s, 0,
                                     (c->fp8 && l > 0) ? at<float>(A, c->blk[l - 1].spr) : nullptr, c->f32, c->dx_stored,
                                     cls ? N : 0));        // last block, class-token pruning: dxb carries a gradient on the class rows only
        }
    }
    if (side_pending) HIP_OK(hipStreamWaitEvent(s, c->ev_join, 0));
    if (tn_pend >= 0)                   // the d W_down product of the last adapter walked
        CHECK(pevit_launch_tn_gemm64(at<bf16>(W, c->sav[tn_pend].z), E, tn_pend_dpre, 64, at<float>(W, c->w_tnD + (size_t)tn_pend * c->tn_layer),
                                     nullptr, at<float>(W, c->w_csy + (size_t)tn_pend * c->csy_layer), T, E, s));
    if (combo && prev_layer >= 0)       // the dP of the last layer walked
        PROF(c, s, PEVIT_PROF_LOWRANK_BWD, T, (double)T * E * 2 + (double)T * 64 * 4,
             pevit_launch_lowrank_combo(0, 1, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, at<bf16>(W, c->sav[prev_layer].xn1), E,
                                        u_last, at<float>(W, c->w_partial + (size_t)prev_layer * c->partial_layer), B, H, N, E, s));
    // adapter gradients of layers [l_lo, l_hi): reduce the partials and chain onto the reference's tensors
    const int nl = l_hi - l_lo;
    const size_t pl0 = c->p_layer0 + c->p_layer_stride * l_lo;          // first float of layer l_lo's parameters
    if (nl <= 0) return 0;
    if (c->d.method == PEVIT_KADAPTATION) {
        CHECK(pevit_launch_chain_kadapt(at<float>(W, c->w_partial + (size_t)l_lo * c->partial_layer), c->partial_layer / 4,
                                        at<float>(W, c->w_dbias + (size_t)l_lo * c->dbias_layer), c->dbias_layer / 4, chunks,
                                        c->ascale, nl, at<float>(W, c->w_G) + (size_t)l_lo * 4 * E * 32,
                                        at<float>(W, c->w_rule) + (size_t)l_lo * 4096, c->params, c->grads, pl0, c->p_layer_stride, E, s));
        // the shared rule factors collect from every layer: each range adds its own layers (top first, one running sum), so a
        // backward that never reaches block 0 keeps its rule contributions and a walk in ranges equals the one-call backward
        CHECK(pevit_launch_rule_sum(at<float>(W, c->w_rule), c->grads, l_lo, l_hi, s));
    } else if (c->d.method == PEVIT_LORA) {
        CHECK(pevit_launch_chain_lora(at<float>(W, c->w_partial + (size_t)l_lo * c->partial_layer), c->partial_layer / 4, chunks,
                                      c->ascale, c->d.lora_rank, nl, at<float>(W, c->w_G) + (size_t)l_lo * 4 * E * 32, c->grads, pl0,
                                      c->p_layer_stride, E, s));
    } else if (post_mlp(c)) {
        const bool fused_lnp = c->adapter_fused && !c->f32 && !c->fused_bn && pevit_adapter_fused_ok(E);
        const int tch = pevit_tn_chunks(T), lnb = fused_lnp ? pevit_adapter_blocks(T) : pevit_lna_blocks(T);
        const size_t ps = c->p_layer_stride, gl = (size_t)E * 64;
        float* g0 = c->grads + pl0;
        float* Gd = at<float>(W, c->w_Gd) + (size_t)l_lo * gl;
        float* Gu = at<float>(W, c->w_Gu) + (size_t)l_lo * gl;
        CHECK(pevit_launch_zero(Gd, (size_t)nl * gl * 4, s));
        CHECK(pevit_launch_zero(Gu, (size_t)nl * gl * 4, s));
        CHECK(pevit_launch_colsum_reduce(at<float>(W, c->w_tnD + (size_t)l_lo * c->tn_layer), tch, (int)gl, Gd, nl, c->tn_layer / 4, gl, s));
        CHECK(pevit_launch_colsum_reduce(at<float>(W, c->w_tnU + (size_t)l_lo * c->tn_layer), tch, (int)gl, Gu, nl, c->tn_layer / 4, gl, s));
        // biases and LayerNorm affine: straight column sums into the flat gradient buffer
        // d b_up from the f32 column sums of the upstream gradient (third plane of the LN partials)
        CHECK(pevit_launch_colsum_reduce(at<float>(W, c->w_csy + (size_t)l_lo * c->csy_layer), tch, 64, g0 + c->o_db, nl,
                                         c->csy_layer / 4, ps, s));
        CHECK(pevit_launch_colsum_reduce3(at<float>(W, c->w_lnp + (size_t)l_lo * c->lnp_layer), lnb, E, g0 + c->o_nw, g0 + c->o_nb,
                                          g0 + c->o_ub, nl, c->lnp_layer / 4, ps, s));
        if (c->d.method == PEVIT_ADAPTER) {
            CHECK(pevit_launch_chain_adapter(Gd, Gu, g0 + c->o_dw, g0 + c->o_uw, E, nl, gl, ps, s));
        } else {
            CHECK(pevit_launch_chain_compacter(Gd, Gu, at<float>(c->arena, c->a_phm), c->params + pl0, g0, E, nl, gl, ps, c->o_dWl,
                                               c->o_dWr, c->o_uWl, c->o_uWr, s));
        }
    }
    return 0;
}

}  // namespace

// Blocks [l_lo, l_hi) of the tower on (N,B,E) activations: ResidualAttentionBlock.forward (model.py:972-975) for one block,
// Transformer.forward (model.py:1013) for all of them.  Every block keeps its own saved activations, so the blocks can be
// walked one call at a time (reference-side code that iterates visual.transformer.resblocks) and differentiated in reverse.
extern "C" int pevit_blocks_forward(pevit_ctx* c, void* stream, const float* x_nbe, float* y_nbe, int B, int save_for_backward,
                                    int l_lo, int l_hi) {
    CHECK(check_ready(c, B, "blocks_forward"));
    if (l_lo < 0 || l_hi > c->L || l_lo >= l_hi) { pevit_set_error("blocks_forward: bad block range [%d, %d)", l_lo, l_hi); return -1; }
    hipStream_t s = (hipStream_t)stream;
    size_t total; layout_workspace(c, B, c->sav, &total, c);
    CHECK(pevit_launch_permute_rows(x_nbe, at<float>(c->ws, c->sav[l_lo].x_in), c->N, B, c->E, 1, s));
    CHECK(blocks_forward(c, s, B, false, l_lo, l_hi));
    const size_t out = l_hi < c->L ? c->sav[l_hi].x_in : c->w_xfinal;
    CHECK(pevit_launch_permute_rows(at<float>(c->ws, out), y_nbe, c->N, B, c->E, 0, s));
    c->saved_batch = save_for_backward ? B : 0; c->saved_kind = 1;
    return 0;
}

extern "C" int pevit_transformer_forward(pevit_ctx* c, void* stream, const float* x_nbe, float* y_nbe, int B,
                                         int save_for_backward) {
    return pevit_blocks_forward(c, stream, x_nbe, y_nbe, B, save_for_backward, 0, c ? c->L : 0);
}

// dy (gradient of the output of block l_hi-1) -> dx (gradient of the input of block l_lo, may be NULL); the adapter gradients of
// exactly these blocks are accumulated into the flat gradient buffer
extern "C" int pevit_blocks_backward(pevit_ctx* c, void* stream, const float* dy_nbe, float* dx_nbe, int B, int l_lo, int l_hi) {
    CHECK(check_ready(c, B, "blocks_backward"));
    if (l_lo < 0 || l_hi > c->L || l_lo >= l_hi) { pevit_set_error("blocks_backward: bad block range [%d, %d)", l_lo, l_hi); return -1; }
    if (c->saved_batch != B || c->saved_kind != 1) {
        pevit_set_error("blocks_backward: the saved activations are not those of a blocks/transformer forward with batch %d "
                        "(saved: batch %d, %s)", B, c->saved_batch, c->saved_kind == 2 ? "visual_forward" : "none");
        return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * c->N * c->E;
    CHECK(pevit_launch_permute_rows(dy_nbe, at<float>(c->ws, c->w_dxa), c->N, B, c->E, 1, s));
    if (c->fp8)
        CHECK(pevit_launch_cast_bf16_cols(at<float>(c->ws, c->w_dxa), at<bf16>(c->ws, c->w_dyb), (size_t)B * c->N, c->E,
                                          at<float>(c->arena, c->blk[l_hi - 1].spr), s));
    else
        CHECK(pevit_launch_cast_bf16(at<float>(c->ws, c->w_dxa), at<bf16>(c->ws, c->w_dyb), n, 1.0f, s, c->f32));
    CHECK(blocks_backward(c, s, B, dx_nbe != nullptr, false, l_hi, l_lo));
    if (dx_nbe) CHECK(pevit_launch_permute_rows(at<float>(c->ws, c->w_dxa), dx_nbe, c->N, B, c->E, 0, s));
    return 0;
}

extern "C" int pevit_transformer_backward(pevit_ctx* c, void* stream, const float* dy_nbe, float* dx_nbe, int B) {
    return pevit_blocks_backward(c, stream, dy_nbe, dx_nbe, B, 0, c ? c->L : 0);
}

extern "C" int pevit_zero_grads(pevit_ctx* c, void* stream) {
    if (!c || !c->grads) { pevit_set_error("zero_grads: parameters not set"); return -1; }
    CHECK(pevit_launch_zero(c->grads, c->n_total * sizeof(float), (hipStream_t)stream));
    return 0;
}

extern "C" int pevit_sgd_step(pevit_ctx* c, void* stream, float lr, float momentum, float wd, float grad_scale,
                              int flags) {
    if (!c || !c->params || !c->grads || !c->mom) { pevit_set_error("sgd_step: parameters/momentum not set"); return -1; }
    // the error word of the workspace bound NOW (a re-bind moves it; without stream-K slots -- f32 verification mode -- there is
    // no hand-off that could fail and nothing to guard)
    unsigned* poison = (c->ws && c->sk_slots) ? at<unsigned>(c->ws, c->w_skflag) + c->sk_slots : nullptr;
    unsigned* skipped = poison ? at<unsigned>(c->ws, c->w_skflag) + PEVIT_SK_MAX_SLOTS + 1 : nullptr;
    float* loss_slot = c->last_loss; c->last_loss = nullptr;
    return pevit_launch_sgd(c->params, c->grads, c->mom, c->grad_mask, c->n_total, lr, momentum, wd, flags,
                            grad_scale, (hipStream_t)stream, poison, skipped, c->ext_poison, (poison || c->ext_poison) ? loss_slot : nullptr);
}

// a device word owned by the caller (e.g. pevit_ar_error_word) that, while non-zero, makes pevit_sgd_step withhold the update --
// the same treatment a stream-K hand-off error gets.  nullptr detaches it.  The word must outlive the context's use of it.
extern "C" int pevit_set_external_poison(pevit_ctx* c, const unsigned* device_word) {
    if (!c) { pevit_set_error("set_external_poison: null context"); return -1; }
    c->ext_poison = device_word;
    return 0;
}

// Round 5, data parallelism: an event the FUSED step (pevit_train_forward_backward[_u8]) waits for on its stream AFTER the stem
// (patch gather, patch embedding, class / position rows, ln_pre -- nothing of which reads a trainable parameter or touches the
// gradient buffer) and BEFORE the first use of the adapters and the clearing of the gradients.  The caller records it behind the
// previous step's gradient exchange + optimizer update, which it may then run on another stream: the exchange's latency and its
// cross-stream hand-overs run under ~70 us of the next step's stem instead of between two steps.  nullptr detaches it.
extern "C" int pevit_set_step_gate(pevit_ctx* c, void* event) {
    if (!c) { pevit_set_error("set_step_gate: null context"); return -1; }
    c->step_gate = (hipEvent_t)event;
    return 0;
}

// ------------------------------------------------------------------------------------
extern "C" int pevit_load_stem(pevit_ctx* c, void* stream, const float* conv_w, const float* cls, const float* pos,
                               const float* lnpre_w, const float* lnpre_b, const float* lnpost_w, const float* lnpost_b,
                               const float* proj) {
    if (!c || !c->arena) { pevit_set_error("load_stem: context not bound"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    char* A = c->arena;
    const size_t E = c->E;
    CHECK(pevit_launch_conv_weight(conv_w, at<bf16>(A, c->a_conv), c->E, 3 * c->P * c->P, c->Kpatch, s, c->f32));
    HIP_OK(hipMemcpyAsync(A + c->a_cls, cls, E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + c->a_pos, pos, (size_t)c->N * E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + c->a_lnpre_w, lnpre_w, E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + c->a_lnpre_b, lnpre_b, E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + c->a_lnpost_w, lnpost_w, E * 4, hipMemcpyDeviceToDevice, s));
    HIP_OK(hipMemcpyAsync(A + c->a_lnpost_b, lnpost_b, E * 4, hipMemcpyDeviceToDevice, s));
    // proj is (E, D): feat = x @ proj  ->  B operand [D][E] = proj^T ; backward uses proj itself [E][D]
    CHECK(pevit_launch_transpose_bf16(proj, c->E, c->D, at<bf16>(A, c->a_proj), c->E, 0, 1.0f, s, c->f32));
    CHECK(pevit_launch_cast_bf16(proj, at<bf16>(A, c->a_projT), E * (size_t)c->D, 1.0f, s, c->f32));
    return 0;
}

extern "C" int pevit_load_phm_rule(pevit_ctx* c, void* stream, const float* phm_rule) {
    if (!c || !c->arena) { pevit_set_error("load_phm_rule: context not bound"); return -1; }
    HIP_OK(hipMemcpyAsync(c->arena + c->a_phm, phm_rule, 64 * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// Preprocessing constants of the uint8 entry points: x = (u8 / 255 - mean[c]) / std[c], the dataset transforms of the reference
// (ToTensor + Normalize(INPUT.MEAN, INPUT.STD), feature.py:537-542; resources/model/vitb32_CLIP.yaml:4-6)
extern "C" int pevit_set_input_norm(pevit_ctx* c, const float* mean3, const float* std3) {
    if (!c || !mean3 || !std3) { pevit_set_error("set_input_norm: null argument"); return -1; }
    for (int i = 0; i < 3; ++i) {
        if (!(std3[i] > 0.f)) { pevit_set_error("set_input_norm: std[%d] = %g must be positive", i, (double)std3[i]); return -1; }
        c->img_mean[i] = mean3[i]; c->img_std[i] = std3[i];
    }
    c->img_norm_set = true;
    return 0;
}

// images (B,3,R,R) f32 -> feat (B,D) f32                               model.py:1034-1051
static int visual_forward_impl(pevit_ctx* c, void* stream, const void* images_any, int u8, float* feat, int B, int save_for_backward);
extern "C" int pevit_visual_forward(pevit_ctx* c, void* stream, const float* images, float* feat, int B,
                                    int save_for_backward) {
    return visual_forward_impl(c, stream, images, 0, feat, B, save_for_backward);
}
// the same from uint8 pixels (B,3,R,R): the reference's ToTensor + Normalize run inside the patch gather (pevit_set_input_norm)
extern "C" int pevit_visual_forward_u8(pevit_ctx* c, void* stream, const uint8_t* images, float* feat, int B,
                                       int save_for_backward) {
    if (c && !c->img_norm_set) { pevit_set_error("visual_forward_u8: call pevit_set_input_norm first"); return -1; }
    return visual_forward_impl(c, stream, images, 1, feat, B, save_for_backward);
}
static int visual_forward_impl(pevit_ctx* c, void* stream, const void* images_any, int u8, float* feat, int B, int save_for_backward) {
    const float* images = (const float*)images_any;
    CHECK(check_ready(c, B, "visual_forward"));
    hipStream_t s = (hipStream_t)stream;
    size_t total; layout_workspace(c, B, c->sav, &total, c);
    char* W = c->ws; char* A = c->arena;
    const int E = c->E, N = c->N, T = B * N;
    float* xpre = at<float>(W, c->w_dxn);               // scratch, free during the forward pass
    if (u8)
        PROF(c, s, PEVIT_PROF_IM2COL, B, (double)B * 3 * c->R * c->R * 1 + (double)B * c->G2 * c->Kpatch * c->es,
             pevit_launch_im2col_u8((const unsigned char*)images_any, c->img_mean, c->img_std, at<bf16>(W, c->w_patches), B, c->R, c->P,
                                    c->Kpatch, s, c->f32));
    else
        PROF(c, s, PEVIT_PROF_IM2COL, B, (double)B * 3 * c->R * c->R * 4 + (double)B * c->G2 * c->Kpatch * c->es,
             pevit_launch_im2col(images, at<bf16>(W, c->w_patches), B, c->R, c->P, c->Kpatch, s, c->f32));
    CHECK(pevit_launch_cls_row(at<float>(A, c->a_cls), at<float>(A, c->a_pos), xpre, B, N, E, s));
    {
        GemmParams p = gp(at<bf16>(W, c->w_patches), c->Kpatch, at<bf16>(A, c->a_conv), c->Kpatch, E, B * c->G2, E, c->Kpatch);
        p.resid = at<float>(A, c->a_pos); p.ldr = E; p.outf = xpre; p.ldo = E; p.Ntok = N;
        CHECK(gemm(c, EPI_PATCH_EMBED, p, s));
    }
    CHECK(pevit_launch_ln_fwd(xpre, at<float>(A, c->a_lnpre_w), at<float>(A, c->a_lnpre_b), T, E, nullptr,
                              at<float>(W, c->sav[0].x_in), nullptr, nullptr, s));
    if (c->gate_now) {                                  // fused step with a gate (pevit_set_step_gate): parameters and gradient buffer from here on
        c->gate_now = false;
        HIP_OK(hipStreamWaitEvent(s, c->step_gate, 0));
        CHECK(pevit_zero_grads(c, stream));
    }
    CHECK(blocks_forward(c, s, B, true));
    // ln_post on the class token of every image (row b*N), then @ proj
    CHECK(pevit_launch_ln_fwd(at<float>(W, c->w_xfinal), at<float>(A, c->a_lnpost_w), at<float>(A, c->a_lnpost_b), B, E,
                              at<bf16>(W, c->w_xpost), nullptr, at<float>(W, c->w_pmean), at<float>(W, c->w_prstd), s,
                              (size_t)N * E, c->f32));
    {
        GemmParams p = gp(at<bf16>(W, c->w_xpost), E, at<bf16>(A, c->a_proj), E, c->D, B, c->D, E);
        p.outf = feat ? feat : at<float>(W, c->w_feat); p.ldo = c->D;
        CHECK(gemm(c, EPI_F32, p, s));
    }
    c->saved_batch = save_for_backward ? B : 0; c->saved_kind = 2;
    return 0;
}

// dfeat (B,D) f32 -> adapter gradients (nothing below the first block is trainable).
// Layers l_hi-1 .. l_lo; the entry work (proj^T, ln_post backward) belongs to the part that starts at L.  Data
// parallelism calls (L, L/2) then (L/2, 0) and all-reduces the first part's gradients while the second runs.
extern "C" int pevit_visual_backward_part(pevit_ctx* c, void* stream, const float* dfeat, int B, int l_hi, int l_lo) {
    CHECK(check_ready(c, B, "visual_backward"));
    if (c->saved_batch != B || c->saved_kind != 2) {
        pevit_set_error("visual_backward: the saved activations are not those of a visual_forward with batch %d (saved: batch %d, %s)",
                        B, c->saved_batch, c->saved_kind == 1 ? "transformer_forward" : "none");
        return -1;
    }
    if (l_lo < 0 || l_hi > c->L || l_lo >= l_hi) { pevit_set_error("visual_backward: bad layer range [%d, %d)", l_lo, l_hi); return -1; }
    if (c->d.method == PEVIT_NONE) return 0;
    hipStream_t s = (hipStream_t)stream;
    char* W = c->ws; char* A = c->arena;
    const int E = c->E, N = c->N, T = B * N;
    const bool cls = !post_mlp(c);
    if (l_hi == c->L) {
        if (!dfeat) { pevit_set_error("visual_backward: dfeat is required for the part that starts at the last block"); return -1; }
        // (the head's BatchNorm backward leaves the bf16 copy of ITS dfeat in w_dfeatb: no cast launch then)
        if (c->dfeatb_of != dfeat) CHECK(pevit_launch_cast_bf16(dfeat, at<bf16>(W, c->w_dfeatb), (size_t)B * c->D, 1.0f, s, c->f32));
        c->dfeatb_of = nullptr;
        {
            GemmParams p = gp(at<bf16>(W, c->w_dfeatb), c->D, at<bf16>(A, c->a_projT), c->D, E, B, E, c->D);
            p.outf = at<float>(W, c->w_dxpost); p.ldo = E;
            CHECK(gemm(c, EPI_F32, p, s));
        }
        // dL/dx_final is zero except on the class-token rows.  With class-token pruning of the last block
        // only those rows of dxa / dyb are ever read; the full-size buffers the last block's attention and
        // LN1 backward consume (dO, dxb) are zeroed instead.
        if (cls) {
            // ... or not read at all: LayerNorm backward takes the residual gradient on the class-token rows only (res_period), and
            // the attention backward for N <= 64 reads dO on token 0 only (dout_cls_only) -- no fill of dxb (19.7 MB) / dO (9.8 MB)
            if (c->f32 || N > 64) CHECK(pevit_launch_zero(W + c->w_dO, (size_t)T * E * c->es, s));
        } else {
            if (!gstream16_on(c)) CHECK(pevit_launch_zero(W + c->w_dxa, (size_t)T * E * 4, s));      // (the bf16 stream never reads the f32 copy)
            CHECK(pevit_launch_zero(W + c->w_dyb, (size_t)T * E * c->es, s));
        }
        CHECK(pevit_launch_ln_bwd(at<float>(W, c->w_dxpost), at<float>(W, c->w_xfinal), at<float>(W, c->w_pmean),
                                  at<float>(W, c->w_prstd), at<float>(A, c->a_lnpost_w), nullptr, at<float>(W, c->w_dxa),
                                  at<bf16>(W, c->w_dyb), B, E, s, (size_t)N * E,
                                  c->fp8 ? at<float>(A, c->blk[c->L - 1].spr) : nullptr, c->f32));
    }
    CHECK(blocks_backward(c, s, B, false, cls, l_hi, l_lo));
    return 0;
}

extern "C" int pevit_visual_backward(pevit_ctx* c, void* stream, const float* dfeat, int B) {
    return pevit_visual_backward_part(c, stream, dfeat, B, c ? c->L : 0, 0);
}

extern "C" int pevit_head_forward_backward(pevit_ctx* c, void* stream, const float* feat, const int64_t* labels,
                                           float* running_mean, float* running_var, int bn_training, float* logits,
                                           float* loss, float* dfeat, int B) {
    if (!c || !c->ws || !c->params || !c->grads) { pevit_set_error("head: context not ready"); return -1; }
    if (B <= 0 || B > c->max_batch) { pevit_set_error("head: batch %d outside [1,%d]", B, c->max_batch); return -1; }
    if (!feat || !running_mean || !running_var || !logits) { pevit_set_error("head: null argument"); return -1; }
    if (labels && !loss) { pevit_set_error("head: labels given but loss is null"); return -1; }
    // torch.nn.BatchNorm1d raises "Expected more than 1 value per channel when training" (the reference's train_one
    // skips such batches, kadaptation_clip.py:341); the batch variance of one sample is 0, never a usable statistic
    if (bn_training && B < 2) { pevit_set_error("head: BatchNorm in training mode needs more than 1 sample per batch (got %d)", B); return -1; }
    hipStream_t s = (hipStream_t)stream;
    char* W = c->ws;
    if (c->saved_batch == 0) { size_t total; layout_workspace(c, B, c->sav, &total, c); }
    if (labels) c->last_loss = loss;
    // only inside the fused step (train_fb_impl): there nobody can touch dfeat between the head and the tower backward
    bf16* dfb = (c->in_fused_step && dfeat && labels && !c->f32) ? at<bf16>(W, c->w_dfeatb) : nullptr;
    c->dfeatb_of = dfb ? dfeat : nullptr;
    return pevit_launch_head(feat, labels, c->params + c->p_head_w, c->params + c->p_head_b,
                             labels ? c->grads + c->p_head_w : nullptr, labels ? c->grads + c->p_head_b : nullptr,
                             running_mean, running_var, bn_training, at<float>(W, c->w_ybn), at<float>(W, c->w_bnrstd),
                             logits, at<float>(W, c->w_dlogits), at<float>(W, c->w_dybn), loss, dfeat, B, c->D, c->C, s, dfb);
}

static int train_fb_impl(pevit_ctx* c, void* stream, const void* images, int u8, const int64_t* labels, float* running_mean,
                         float* running_var, int bn_training, float* logits, float* loss, int B);
extern "C" int pevit_train_forward_backward(pevit_ctx* c, void* stream, const float* images, const int64_t* labels,
                                            float* running_mean, float* running_var, int bn_training, float* logits,
                                            float* loss, int B) {
    return train_fb_impl(c, stream, images, 0, labels, running_mean, running_var, bn_training, logits, loss, B);
}
extern "C" int pevit_train_forward_backward_u8(pevit_ctx* c, void* stream, const uint8_t* images, const int64_t* labels,
                                               float* running_mean, float* running_var, int bn_training, float* logits,
                                               float* loss, int B) {
    if (c && !c->img_norm_set) { pevit_set_error("train_forward_backward_u8: call pevit_set_input_norm first"); return -1; }
    return train_fb_impl(c, stream, images, 1, labels, running_mean, running_var, bn_training, logits, loss, B);
}
static int train_fb_impl(pevit_ctx* c, void* stream, const void* images, int u8, const int64_t* labels, float* running_mean,
                         float* running_var, int bn_training, float* logits, float* loss, int B) {
    CHECK(check_ready(c, B, "train_forward_backward"));
    c->gate_now = c->step_gate != nullptr;              // with a gate the gradients are cleared behind it, inside the forward pass
    if (!c->gate_now) CHECK(pevit_zero_grads(c, stream));
    const int frc = visual_forward_impl(c, stream, images, u8, nullptr, B, 1);
    c->gate_now = false;
    if (frc) return frc;
    float* feat = at<float>(c->ws, c->w_feat);
    float* dfeat = at<float>(c->ws, c->w_dfeat);
    c->in_fused_step = true;
    const int hrc = pevit_head_forward_backward(c, stream, feat, labels, running_mean, running_var, bn_training, logits, loss, dfeat, B);
    c->in_fused_step = false;
    if (hrc) { c->dfeatb_of = nullptr; return hrc; }
    CHECK(pevit_visual_backward(c, stream, dfeat, B));
    return 0;
}

// ------------------------------------------------------------------------------------
// Per-launch timing of the dominant kernel family (the MFMA GEMMs) with HIP events recorded on
// the caller's stream around every GEMM launch of the context.  Events are created here, not in
// the hot path.  pevit_profile_end synchronises the events and returns the totals.
extern "C" int pevit_profile_begin(pevit_ctx* c, int max_launches) {
    if (!c || max_launches <= 0) { pevit_set_error("profile_begin: bad argument"); return -1; }
    if (c->prof_cap < max_launches) {
        for (int i = 0; i < 2 * c->prof_cap; ++i) (void)hipEventDestroy(c->prof_ev[i]);
        delete[] c->prof_ev; delete[] c->prof_flops; delete[] c->prof_bytes; delete[] c->prof_ms; delete[] c->prof_shape;
        c->prof_ev = new (std::nothrow) hipEvent_t[2 * max_launches];
        c->prof_flops = new (std::nothrow) double[max_launches];
        c->prof_bytes = new (std::nothrow) double[max_launches];
        c->prof_ms = new (std::nothrow) float[max_launches];
        c->prof_shape = new (std::nothrow) int[4 * max_launches];
        if (!c->prof_ev || !c->prof_flops || !c->prof_bytes || !c->prof_ms || !c->prof_shape) { pevit_set_error("profile_begin: out of host memory"); return -1; }
        for (int i = 0; i < 2 * max_launches; ++i) HIP_OK(hipEventCreate(&c->prof_ev[i]));
        c->prof_cap = max_launches;
    }
    c->prof_n = 0; c->prof_on = true;
    return 0;
}

extern "C" int pevit_profile_end(pevit_ctx* c, double* total_ms, double* total_flops, double* total_bytes, int* launches) {
    if (!c || !c->prof_on) { pevit_set_error("profile_end: profiling is not active"); return -1; }
    c->prof_on = false;
    double ms = 0.0, fl = 0.0, by = 0.0;
    for (int i = 0; i < c->prof_n; ++i) {
        HIP_OK(hipEventSynchronize(c->prof_ev[2 * i + 1]));
        float t = 0.f;
        HIP_OK(hipEventElapsedTime(&t, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
        c->prof_ms[i] = t;
        if (c->prof_shape[4 * i] < 100) { ms += t; fl += c->prof_flops[i]; by += c->prof_bytes[i]; }     // totals: the GEMM family
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (total_bytes) *total_bytes = by;
    if (launches) *launches = c->prof_n;
    return 0;
}

// launch i of the last begin/end pair: duration, 2*M*N*K, and {epilogue, M, N, K}
extern "C" int pevit_profile_launch(pevit_ctx* c, int i, double* ms, double* flops, int* epi_mnk) {
    if (!c || c->prof_on || i < 0 || i >= c->prof_n) { pevit_set_error("profile_launch: no such recorded launch"); return -1; }
    if (ms) *ms = c->prof_ms[i];
    if (flops) *flops = c->prof_flops[i];
    if (epi_mnk) for (int k = 0; k < 4; ++k) epi_mnk[k] = c->prof_shape[4 * i + k];
    return 0;
}

// ... and the algorithmic bytes of that launch (operands read once + results written once)
extern "C" int pevit_profile_launch_bytes(pevit_ctx* c, int i, double* bytes) {
    if (!c || c->prof_on || i < 0 || i >= c->prof_n || !bytes) { pevit_set_error("profile_launch_bytes: no such recorded launch"); return -1; }
    *bytes = c->prof_bytes[i];
    return 0;
}

// ------------------------------------------------------------------------------------
// single-kernel entry points (parity tests, profiling)
// stream-K workspace of the context-free entry point (tests / microbenchmarks): allocated on first use
static int op_sk_workspace(GemmParams& p) {
    static char* ws = nullptr;
    const int slots = pevit_gemm_sk_slots();
    const size_t flag_bytes = align_up((size_t)(PEVIT_SK_MAX_SLOTS + 1) * 4, 256);
    if (!ws) {
        HIP_OK(hipMalloc((void**)&ws, flag_bytes + (size_t)slots * PEVIT_SK_SLAB_FLOATS * 4));
        HIP_OK(hipMemset(ws, 0, flag_bytes));
    }
    p.sk_flag = reinterpret_cast<unsigned*>(ws); p.sk_slab = reinterpret_cast<float*>(ws + flag_bytes); p.sk_slots = slots;
    return 0;
}

extern "C" int pevit_op_gemm(void* stream, int epi, const void* A, int lda, const void* Bm, int ldb, int b_rows, int M,
                             int N, int K, const float* bias, const float* resid, int ldr, float* outf, int ldo,
                             void* outb, int ldob, void* outb2, int ldob2, const void* aux, int ldaux,
                             size_t head_stride, int E, int H, int tokens) {
    GemmParams p = gp((const bf16*)A, lda, (const bf16*)Bm, ldb, b_rows, M, N, K);
    p.bias = bias; p.resid = resid; p.ldr = ldr; p.outf = outf; p.ldo = ldo; p.outb = (bf16*)outb; p.ldob = ldob;
    p.outb2 = (bf16*)outb2; p.ldob2 = ldob2; p.aux = (const bf16*)aux; p.ldaux = ldaux; p.head_stride = head_stride;
    p.E = E; p.H = H; p.Ntok = tokens;
    if (g_default_tune.streamk) CHECK(op_sk_workspace(p));
    return pevit_launch_gemm(epi, p, g_default_tune, (hipStream_t)stream);
}
// 1 if a stream-K consumer ever gave up waiting for a partial tile (context-free workspace when ctx is null); clears it
extern "C" int pevit_streamk_error(pevit_ctx* c, void* stream) {
    GemmParams p; memset(&p, 0, sizeof(p));
    unsigned* flag = nullptr;
    if (c) { if (!c->ws || !c->sk_slots) return 0; flag = at<unsigned>(c->ws, c->w_skflag) + c->sk_slots; }
    else { if (op_sk_workspace(p)) return -1; flag = p.sk_flag + p.sk_slots; }
    unsigned v = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, flag, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (v) {
        (void)hipMemset(flag, 0, 4);
        if (c) (void)hipMemset(at<unsigned>(c->ws, c->w_skflag) + PEVIT_SK_MAX_SLOTS + 1, 0, 4);    // the skipped-update counter with it
    }
    return v ? 1 : 0;
}
// the same word without clearing it, plus the number of optimizer updates the fused SGD kernel withheld because of it.  A caller
// that wants to go on after the error calls pevit_streamk_error (which clears the word) and knows how many steps it lost.
extern "C" int pevit_streamk_status(pevit_ctx* c, void* stream, unsigned* error_word, unsigned* skipped_updates) {
    if (!c) { pevit_set_error("streamk_status: null context"); return -1; }
    unsigned v[2] = {0, 0};
    if (c->ws && c->sk_slots) {
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
        if (hipMemcpy(&v[0], at<unsigned>(c->ws, c->w_skflag) + c->sk_slots, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (hipMemcpy(&v[1], at<unsigned>(c->ws, c->w_skflag) + PEVIT_SK_MAX_SLOTS + 1, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    if (error_word) *error_word = v[0];
    if (skipped_updates) *skipped_updates = v[1];
    return 0;
}
extern "C" int pevit_op_gemm_fp8(void* stream, int epi, const void* A, int lda, const void* Bcodes, int ldb, int b_rows,
                                 const float* bscale, const float* oscale, int M, int N, int K, const float* bias,
                                 const float* resid, int ldr, float* outf, int ldo, void* outb, int ldob, void* outb2,
                                 int ldob2, const void* aux, int ldaux, size_t head_stride, int E, int H, int tokens) {
    GemmParams p = gp((const bf16*)A, lda, (const bf16*)Bcodes, ldb, b_rows, M, N, K);
    p.b_fp8 = 1; p.bscale = bscale; p.oscale = oscale;
    p.bias = bias; p.resid = resid; p.ldr = ldr; p.outf = outf; p.ldo = ldo; p.outb = (bf16*)outb; p.ldob = ldob;
    p.outb2 = (bf16*)outb2; p.ldob2 = ldob2; p.aux = (const bf16*)aux; p.ldaux = ldaux; p.head_stride = head_stride;
    p.E = E; p.H = H; p.Ntok = tokens;
    return pevit_launch_gemm(epi, p, g_default_tune, (hipStream_t)stream);
}
// fp8 x fp8 form (PEVIT_W_FP8_ACT): A = unscaled e4m3 codes as written by pevit_op_cast_fp8
extern "C" int pevit_op_gemm_f8a(void* stream, int epi, const void* Acodes, int lda, const void* Bcodes, int ldb, int b_rows,
                                 const float* bscale, int M, int N, int K, const float* bias, const float* resid, int ldr,
                                 float* outf, int ldo, void* outb, int ldob, void* outb2, int ldob2, int out2_fp8,
                                 size_t head_stride, int E, int H, int tokens) {
    GemmParams p = gp((const bf16*)Acodes, lda, (const bf16*)Bcodes, ldb, b_rows, M, N, K);
    p.b_fp8 = 1; p.a_fp8 = 1; p.bscale = bscale; p.out2_fp8 = out2_fp8;
    p.bias = bias; p.resid = resid; p.ldr = ldr; p.outf = outf; p.ldo = ldo; p.outb = (bf16*)outb; p.ldob = ldob;
    p.outb2 = (bf16*)outb2; p.ldob2 = ldob2; p.head_stride = head_stride; p.E = E; p.H = H; p.Ntok = tokens;
    return pevit_launch_gemm(epi, p, g_default_tune, (hipStream_t)stream);
}
extern "C" int pevit_op_cast_fp8(void* stream, const float* src, void* codes, int rows, int cols) {
    return pevit_launch_cast_fp8(src, (unsigned char*)codes, (size_t)rows, cols, (hipStream_t)stream);
}
extern "C" int pevit_op_quant_fp8(void* stream, const float* W, int rows, int cols, void* codes, float* scales,
                                  void* codes_t) {
    CHECK(pevit_launch_quant_rows_fp8(W, rows, cols, (unsigned char*)codes, cols, scales, 0, 1.0f, (hipStream_t)stream));
    if (codes_t)
        CHECK(pevit_launch_quant_transpose_fp8(W, rows, cols, scales, (unsigned char*)codes_t, rows, 0, 1.0f, (hipStream_t)stream));
    return 0;
}
extern "C" int pevit_op_dequant_fp8(void* stream, const void* codes, const float* scales, int rows, int cols, float* out) {
    return pevit_launch_dequant_rows_fp8((const unsigned char*)codes, cols, scales, rows, cols, out, (hipStream_t)stream);
}
extern "C" int pevit_op_ln_fwd(void* stream, const float* x, const float* gamma, const float* beta, int rows, int E,
                               void* y_bf16, float* y_f32, float* mean, float* rstd) {
    return pevit_launch_ln_fwd(x, gamma, beta, rows, E, (bf16*)y_bf16, y_f32, mean, rstd, (hipStream_t)stream);
}
extern "C" int pevit_op_ln_bwd(void* stream, const float* dy, const float* x, const float* mean, const float* rstd,
                               const float* gamma, const float* dres, float* dx, void* dx_bf16, int rows, int E) {
    return pevit_launch_ln_bwd(dy, x, mean, rstd, gamma, dres, dx, (bf16*)dx_bf16, rows, E, (hipStream_t)stream);
}
extern "C" int pevit_op_ln_bwd_scaled(void* stream, const float* dy, const float* x, const float* mean, const float* rstd,
                                      const float* gamma, const float* dres, float* dx, void* dx_bf16, int rows, int E,
                                      const float* bf16_colscale) {
    return pevit_launch_ln_bwd(dy, x, mean, rstd, gamma, dres, dx, (bf16*)dx_bf16, rows, E, (hipStream_t)stream, 0, bf16_colscale);
}
extern "C" int pevit_op_attn_fwd(void* stream, const void* q, const void* k, const void* v, void* out, int ldo,
                                 float* lse, int B, int H, int N) {
    return pevit_launch_attn_fwd((const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)out, ldo, lse, B, H, N,
                                 (hipStream_t)stream);
}
extern "C" int pevit_op_attn_bwd(void* stream, const void* q, const void* k, const void* v, const void* out, int ldo,
                                 const void* dout, int lddo, const float* lse, void* dqkv, int ld, int B, int H, int N) {
    return pevit_launch_attn_bwd((const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)out, ldo,
                                 (const bf16*)dout, lddo, lse, (bf16*)dqkv, ld, B, H, N, (hipStream_t)stream);
}
extern "C" int pevit_op_cast_bf16(void* stream, const float* src, void* dst, size_t n, float scale) {
    return pevit_launch_cast_bf16(src, (bf16*)dst, n, scale, (hipStream_t)stream);
}
extern "C" int pevit_op_delta_add(void* stream, void* qbuf, void* vbuf, const float* t, const void* q16,
                                  const float* bias, float ascale, int B, int N, int E) {
    return pevit_launch_delta_add((bf16*)qbuf, (bf16*)vbuf, t, nullptr, (const bf16*)q16, bias, ascale, B, N, E, (hipStream_t)stream);
}
extern "C" int pevit_op_attn_fwd_delta(void* stream, void* q, const void* k, void* v, const float* t, const void* q16, const float* bias,
                                       float ascale, void* out, int ldo, float* lse, int B, int H, int N) {
    return pevit_launch_attn_fwd_delta((bf16*)q, (const bf16*)k, (bf16*)v, t, (const bf16*)q16, bias, ascale, (bf16*)out, ldo, lse, B, H, N,
                                       (hipStream_t)stream);
}
extern "C" int pevit_debug_occupy(void* stream, int workgroups, int lds_bytes, double microseconds) {
    return pevit_launch_occupy(workgroups, lds_bytes, microseconds, (hipStream_t)stream);
}
extern "C" int pevit_debug_timeline(void* buf) { pevit_attn_delta_set_timeline(buf); return 0; }
extern "C" int pevit_op_attn_delta_hpw(int B, int H, int N) { return pevit_attn_delta_hpw(B, H, N); }
extern "C" int pevit_op_lowrank_u(void* stream, const void* dqkv, int ld, const void* qT, float* u32, void* u_cols, int B,
                                  int H, int N, int E) {
    return pevit_launch_lowrank_u((const bf16*)dqkv, ld, (const bf16*)qT, u32, (bf16*)u_cols, B, H, N, E,
                                  (hipStream_t)stream);
}
extern "C" int pevit_op_lowrank_grad(void* stream, const void* xn, int ldx, const float* u32, const void* dqkv, int ld,
                                     const float* t, float* partial, float* dbias_partial, int B, int H, int N, int E) {
    return pevit_launch_lowrank_grad((const bf16*)xn, ldx, u32, (const bf16*)dqkv, ld, t, partial, dbias_partial,
                                     pevit_lowrank_chunks(B * N), B, H, N, E, (hipStream_t)stream);
}
extern "C" int pevit_op_lowrank_chunks(int T) { return pevit_lowrank_chunks(T); }
// ---- post-MLP adapter kernels (adapter.hip), one layer at a time
extern "C" int pevit_op_tn_chunks(int T) { return pevit_tn_chunks(T); }
extern "C" int pevit_op_lna_blocks(int rows) { return pevit_lna_blocks(rows); }
extern "C" int pevit_op_tn_gemm64(void* stream, const void* X, int ldx, const void* Y, int ldy, float* partial, float* csx,
                                  float* csy, int T, int E) {
    return pevit_launch_tn_gemm64((const bf16*)X, ldx, (const bf16*)Y, ldy, partial, csx, csy, T, E, (hipStream_t)stream);
}
extern "C" int pevit_op_ln_bwd_affine(void* stream, const float* dy, const float* x, const float* mean, const float* rstd,
                                      const float* gamma, const float* dres, float* dx, void* dx_bf16, float* partial,
                                      int rows, int E) {
    return pevit_launch_ln_bwd_affine(dy, x, mean, rstd, gamma, dres, dx, (bf16*)dx_bf16, partial, rows, E, (hipStream_t)stream);
}
extern "C" int pevit_op_colsum_reduce(void* stream, const float* partial, int chunks, int n, float* out0, float* out1,
                                      float* out2) {
    if (out1 || out2) {
        if (!out1 || !out2) { pevit_set_error("colsum_reduce: give one output or three"); return -1; }
        return pevit_launch_colsum_reduce3(partial, chunks, n, out0, out1, out2, 1, 0, 0, (hipStream_t)stream);
    }
    return pevit_launch_colsum_reduce(partial, chunks, n, out0, 1, 0, 0, (hipStream_t)stream);
}
extern "C" int pevit_op_prep_bottleneck(void* stream, int method, const float* rule, const float* p0, const float* p1,
                                        const float* p2, const float* p3, void* wd, void* wdT, void* wu, void* wuT, int E) {
    BottleneckPanels pan{(bf16*)wd, (bf16*)wdT, (bf16*)wu, (bf16*)wuT};
    LayerStrides st{0, 0};
    if (method == PEVIT_ADAPTER) return pevit_launch_prep_adapter(p0, p1, pan, E, 1, st, (hipStream_t)stream);
    if (method == PEVIT_COMPACTER) return pevit_launch_prep_compacter(rule, p0, p1, p2, p3, pan, E, 1, st, (hipStream_t)stream);
    pevit_set_error("prep_bottleneck: method %d is not a post-MLP adapter", method);
    return -1;
}
extern "C" int pevit_op_chain_bottleneck(void* stream, int method, const float* Gd, const float* Gu, const float* rule,
                                         const float* params, float* grads, int E, size_t off0, size_t off1, size_t off2,
                                         size_t off3) {
    if (method == PEVIT_ADAPTER)
        return pevit_launch_chain_adapter(Gd, Gu, grads + off0, grads + off1, E, 1, 0, 0, (hipStream_t)stream);
    if (method == PEVIT_COMPACTER)
        return pevit_launch_chain_compacter(Gd, Gu, rule, params, grads, E, 1, 0, 0, off0, off1, off2, off3, (hipStream_t)stream);
    pevit_set_error("chain_bottleneck: method %d is not a post-MLP adapter", method);
    return -1;
}
extern "C" int pevit_op_im2col_u8(void* stream, const uint8_t* images, const float* mean3, const float* std3, void* patches_bf16, int B,
                                  int R, int P, int Kpad) {
    return pevit_launch_im2col_u8(images, mean3, std3, (bf16*)patches_bf16, B, R, P, Kpad, (hipStream_t)stream);
}
extern "C" int pevit_op_im2col(void* stream, const float* images, void* patches_bf16, int B, int R, int P, int Kpad) {
    return pevit_launch_im2col(images, (bf16*)patches_bf16, B, R, P, Kpad, (hipStream_t)stream);
}
extern "C" int pevit_tune(pevit_ctx* c, const char* key, int value) {
    GemmTune& t = c ? c->tune : g_default_tune;
    if (key && !strcmp(key, "gemm_config")) { t.config = value; return 0; }
    if (key && !strcmp(key, "gemm_persistent")) { t.persistent = value; return 0; }
    if (key && !strcmp(key, "gemm_ablate")) { t.ablate = value; return 0; }
    if (key && !strcmp(key, "gemm_kswitch")) { t.kswitch = value; return 0; }
    if (key && !strcmp(key, "gemm_big")) { t.big = value; return 0; }
    if (key && !strcmp(key, "gemm_cfg_longk")) { t.cfg_longk = value; return 0; }
    if (key && !strcmp(key, "gemm_cfg_shortk")) { t.cfg_shortk = value; return 0; }
    if (key && !strcmp(key, "gemm_big_bias")) { t.big_bias = value; return 0; }
    if (key && c && !strcmp(key, "side_stream")) { c->side_stream = value; return 0; }
    if (key && c && !strcmp(key, "fused_bottleneck")) { c->fused_bn = value; return 0; }
    if (key && !strcmp(key, "gemm_streamk")) { t.streamk = value; return 0; }
    if (key && !strcmp(key, "gemm_sk_share")) { t.sk_share = value; return 0; }
    if (key && !strcmp(key, "gemm_sk_band")) { t.sk_band = value; return 0; }
    if (key && !strcmp(key, "gemm_ksplit")) { t.ksplit = value; return 0; }
    if (key && !strcmp(key, "gemm_ksplit_small")) { t.ksplit_small = value; return 0; }
    if (key && !strcmp(key, "gemm_ksplit_stagger")) { t.ksplit_stagger = value; return 0; }
    if (key && !strcmp(key, "gemm_ksplit_mink")) { t.ksplit_mink = value; return 0; }
    if (key && !strcmp(key, "gemm_kphase_nl")) { t.kphase_nl = value; return 0; }
    if (key && !strcmp(key, "gemm_kz2")) { t.kz2 = value; return 0; }
    if (key && !strcmp(key, "gemm_skinny")) { t.skinny = value; return 0; }
    if (key && !strcmp(key, "gemm_skinny_maxm")) { t.skinny_maxm = value; return 0; }
    if (key && !strcmp(key, "gemm_skinny_mink")) { t.skinny_mink = value; return 0; }
    if (key && !strcmp(key, "gemm_skinny_slices")) { t.skinny_slices = value; return 0; }
    if (key && !strcmp(key, "gemm_band")) { t.band = value; return 0; }
    if (key && !strcmp(key, "gemm_stagger")) { t.stagger = value; return 0; }
    if (key && c && !strcmp(key, "dx_stored")) { c->dx_stored = value; return 0; }
    if (key && c && !strcmp(key, "gstream_bf16")) { c->gstream16 = value; return 0; }
    if (key && c && !strcmp(key, "profile_all")) { c->prof_all = value; return 0; }
    if (key && c && !strcmp(key, "fused_attn_delta")) { c->fused_attn_delta = value; return 0; }
    if (key && c && !strcmp(key, "fp8_tail")) { c->fp8_tail = value; return 0; }
    if (key && c && !strcmp(key, "adapter_fused")) { c->adapter_fused = value; return 0; }
    if (key && c && !strcmp(key, "adapter_tn_fold")) { c->adapter_tn_fold = value; return 0; }
    if (key && c && !strcmp(key, "lowrank_combo")) { c->lowrank_combo = value; return 0; }
    if (key && c && !strcmp(key, "lowrank_xcd")) { c->lowrank_xcd = value; return 0; }
    pevit_set_error("tune: unknown key %s", key ? key : "(null)");
    return -1;
}

extern "C" int pevit_debug_last_gemm_path(void) { return pevit_gemm_last_path(); }
