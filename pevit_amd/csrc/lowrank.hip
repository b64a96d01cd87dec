// Rank-R form of the attention-site adapters (KAdaptation and LoRA).
//
// The reference materialises H = sum_i kron(rule_i, W_i) (a 32x768x768 tensor, twice per
// layer: model.py:406-417,575,580) and multiplies x @ H * 160 + b (model.py:584).  With the
// rank-1 factors it actually uses, H == P Q^T with P[:,i] = s_i (x) l_i, Q[:,i] = t_i (x) r_i
// (SURVEY 9.5), so the same delta is  ascale * (x P) Q^T + b  with P,Q in R^{E x 32}.  LoRA
// (lora_model.py:490-514) is the same shape with P = A1^T, Q = A2, rank r <= 32.
//
//   t = xn P            fused into the QKV GEMM as 64 extra output columns (gemm.hip)
//   delta-add           q_flat[i] += ascale * t[row] . Q[e] + b[e]        (this file)
//   u = dDelta Q        16-row MFMA tiles, K = E                          (this file)
//   dxn += ascale*u P^T fused into the QKV-backward GEMM as 64 extra K columns
//   dP = xn^T u, dQ = dDelta^T t, db = colsum(dDelta)   f32 MFMA, contraction over tokens
//   chain rule to the reference's parameter tensors                       (this file)
//
// "flat" addressing reproduces the reference's raw reshape (model.py:796-799):
// the (N,B,E)-contiguous delta buffer is reinterpreted as (B*H, N, 64), i.e. element
// i = rr*E + e of the delta (rr = n*B + b) is added to element i of the head-layout q buffer.
// Internally rows are batch-major (row = b*N + n), hence row(rr) below.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ int row_of_ref(int rr, int B, int N) {
    const int n = rr / B, b = rr - n * B;
    return b * N + n;
}

// ---------------------------------------------------------------------------------
// blockIdx.y = layer: every per-layer pointer advances by a constant stride (LayerStrides)
__device__ __forceinline__ AdapterPanels layer_panels(AdapterPanels pan, LayerStrides st, int l) {
    pan.w_aug_rows = reinterpret_cast<bf16*>(reinterpret_cast<char*>(pan.w_aug_rows) + (size_t)l * st.arena_bytes);
    pan.wT_aug_cols = reinterpret_cast<bf16*>(reinterpret_cast<char*>(pan.wT_aug_cols) + (size_t)l * st.arena_bytes);
    pan.q32 = reinterpret_cast<float*>(reinterpret_cast<char*>(pan.q32) + (size_t)l * st.arena_bytes);
    pan.qT = reinterpret_cast<bf16*>(reinterpret_cast<char*>(pan.qT) + (size_t)l * st.arena_bytes);
    pan.q16 = reinterpret_cast<bf16*>(reinterpret_cast<char*>(pan.q16) + (size_t)l * st.arena_bytes);
    return pan;
}

// Two index orders in one launch (blockIdx.z): z = 0 walks (j, e) with e fastest and writes the panels whose rows run along e
// (P^T rows of Wqkv_aug, Q^T); z = 1 walks (e, j) with j fastest and writes the panels whose rows run along j (the 64 adapter columns
// of Wqkv^T_aug, Q as f32 and bf16).  Written from one order, half of the stores were 2-byte writes 4.7 KB apart (15 us per step).
template <typename ST>
__global__ void prep_kadapt_kernel(const float* __restrict__ rule1_l, const float* __restrict__ rule1_r,
                                   const float* __restrict__ rule2_l, const float* __restrict__ rule2_r,
                                   const float* __restrict__ q_left, const float* __restrict__ q_right,
                                   AdapterPanels pan, int E, float ascale, LayerStrides st) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * 32) return;
    pan = layer_panels(pan, st, blockIdx.y);
    q_left += (size_t)blockIdx.y * st.param_floats; q_right += (size_t)blockIdx.y * st.param_floats;
    const bool by_e = blockIdx.z == 0;
    const int j = by_e ? idx / E : idx & 31, e = by_e ? idx - j * E : idx >> 5;
    const int F = E / 32, a = e / F, kk = e - a * F;
    const float l = q_left[j * F + kk], r = q_right[j * F + kk];
    const float pq = rule1_l[j * 32 + a] * l, pv = rule2_l[j * 32 + a] * l;
    const float qq = rule1_r[j * 32 + a] * r, qv = rule2_r[j * 32 + a] * r;
    if (by_e) {
        st_store<ST>(pan.w_aug_rows, (size_t)j * pan.ldw + e, pq);
        st_store<ST>(pan.w_aug_rows, (size_t)(32 + j) * pan.ldw + e, pv);
        st_store<ST>(pan.qT, (size_t)j * E + e, qq);
        st_store<ST>(pan.qT, (size_t)(32 + j) * E + e, qv);
    } else {
        st_store<ST>(pan.wT_aug_cols, (size_t)e * pan.ldwT + j, ascale * pq);
        st_store<ST>(pan.wT_aug_cols, (size_t)e * pan.ldwT + 32 + j, ascale * pv);
        pan.q32[(size_t)e * 64 + j] = qq;
        pan.q32[(size_t)e * 64 + 32 + j] = qv;
        if constexpr (sizeof(ST) == 2) {       // the forward delta's operand: Q rounded to bf16 like P (rows of Wqkv_aug) and Q^T
            pan.q16[(size_t)e * 64 + j] = f2bf(qq);
            pan.q16[(size_t)e * 64 + 32 + j] = f2bf(qv);
        }
    }
}

template <typename ST>
__global__ void prep_lora_kernel(const float* __restrict__ a1q, const float* __restrict__ a2q,
                                 const float* __restrict__ a1v, const float* __restrict__ a2v, int r,
                                 AdapterPanels pan, int E, float ascale, LayerStrides st) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * 32) return;
    pan = layer_panels(pan, st, blockIdx.y);
    { const size_t o = (size_t)blockIdx.y * st.param_floats; a1q += o; a2q += o; a1v += o; a2v += o; }
    const int j = idx / E, e = idx - j * E;
    const float pq = j < r ? a1q[(size_t)j * E + e] : 0.f, pv = j < r ? a1v[(size_t)j * E + e] : 0.f;
    const float qq = j < r ? a2q[(size_t)e * r + j] : 0.f, qv = j < r ? a2v[(size_t)e * r + j] : 0.f;
    st_store<ST>(pan.w_aug_rows, (size_t)j * pan.ldw + e, pq);
    st_store<ST>(pan.w_aug_rows, (size_t)(32 + j) * pan.ldw + e, pv);
    st_store<ST>(pan.wT_aug_cols, (size_t)e * pan.ldwT + j, ascale * pq);
    st_store<ST>(pan.wT_aug_cols, (size_t)e * pan.ldwT + 32 + j, ascale * pv);
    pan.q32[(size_t)e * 64 + j] = qq;
    pan.q32[(size_t)e * 64 + 32 + j] = qv;
    if constexpr (sizeof(ST) == 2) {           // the forward delta's operand: Q rounded to bf16 like P (rows of Wqkv_aug) and Q^T
        pan.q16[(size_t)e * 64 + j] = f2bf(qq);
        pan.q16[(size_t)e * 64 + 32 + j] = f2bf(qv);
    }
    st_store<ST>(pan.qT, (size_t)j * E + e, qq);
    st_store<ST>(pan.qT, (size_t)(32 + j) * E + e, qv);
}

// ---------------------------------------------------------------------------------
// delta-add on the matrix core.  D[e][rr] = sum_j Q[e][j] t[rr][j] is a K=32 product, exactly one
// v_mfma_f32_16x16x32_bf16.  Production (bf16 storage): Q enters as its bf16 panel q16 -- the same rounding P carries in the
// forward t = xn P product and Q^T in the backward u = dDelta Q -- and the f32 t as bf16 hi + lo parts: Q*t_lo + Q*t_hi, two
// MFMAs per 16x16 tile (round 4; before, Q was split as well: three MFMAs and twice the operand bytes, which is what bounds
// the fused kernel of attn_delta.hip).  f32 verification mode: both operands split, hi*hi + hi*lo + lo*hi (error ~2^-17).
__device__ __forceinline__ void split_bf16v(const float4 a, const float4 b, bf16x8& hi, bf16x8& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = f2bf(v[i]);
        lo[i] = f2bf(v[i] - bf2f(hi[i]));
    }
}
__device__ __forceinline__ void split_bf16(const float* src, bf16x8& hi, bf16x8& lo) {
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = f2bf(v[i]);
        lo[i] = f2bf(v[i] - bf2f(hi[i]));
    }
}

// A wave owns DA_RG groups of 16 reference rows and 64 columns of E (two steps of 32 = two 16x16 tiles whose
// output rows are interleaved, tile 0: e = 8g+r, tile 1: e = 8g+4+r, so that a lane ends up with 8 consecutive e of
// one row -> one 16-byte read-modify-write); the split Q fragments of a step are loaded once and reused for all
// DA_RG row groups (they were 2/3 of this kernel's L2 traffic when every 16-row group re-read them).
#ifndef DA_RG_V
#define DA_RG_V 4
#endif
constexpr int DA_RG = DA_RG_V, DA_COLS = 64;
template <typename ST>
__global__ __launch_bounds__(256) void delta_add_kernel(bf16* qbuf, bf16* vbuf, const float* __restrict__ t,
                                                        const float* __restrict__ q32, const bf16* __restrict__ q16,
                                                        const float* __restrict__ bias, float ascale, int B, int N, int E) {
    constexpr bool Q16 = sizeof(ST) == 2;
    const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int which = blockIdx.y;                  // 0: q, 1: v
    const int T = B * N;
    const int parts = E / DA_COLS;
    const int grp = wg / parts, part = wg - grp * parts;
    const int rr0 = grp * 16 * DA_RG;
    if (rr0 >= T) return;                          // whole wave exits together
    size_t boff[DA_RG];                            // element offset of this lane's row in the q / v buffer
    bf16* const base = which ? vbuf : qbuf;
    bool rok[DA_RG];
    int rrk[DA_RG];
#pragma unroll
    for (int k = 0; k < DA_RG; ++k) {
        const int rr = rr0 + 16 * k + c16;
        rok[k] = rr < T;
        rrk[k] = rok[k] ? rr : T - 1;
        boff[k] = (size_t)rrk[k] * E;
    }
    // the read-modify-write operands come from HBM: request all of them before anything else (the t / Q fragments
    // below are L2 hits and their bf16 split is ALU work that runs under this latency)
    constexpr int NST = DA_COLS / 32;
    typedef typename std::conditional<sizeof(ST) == 2, bf16x8, f32x8>::type raw8;
    raw8 cur[NST][DA_RG];                          // kept raw: converting here would wait for the loads here
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int k = 0; k < DA_RG; ++k)
            cur[st][k] = *reinterpret_cast<const raw8*>(reinterpret_cast<const ST*>(base) + boff[k] + part * DA_COLS + st * 32 + 8 * g);
    // ... and so are the t rows, the Q rows and the bias of BOTH 32-column steps (L2 hits): a load issued after the first step's
    // stores would make hipcc wait with vmcnt(0), i.e. for those stores as well (gfx950 counts them in vmcnt)
    float4 traw[DA_RG][2], qraw[NST][2][2], braw[NST][2];
    bf16x8 qb16[NST][2];
#pragma unroll
    for (int k = 0; k < DA_RG; ++k) {
        const float* src = t + (size_t)row_of_ref(rrk[k], B, N) * 64 + which * 32 + 8 * g;
        traw[k][0] = *reinterpret_cast<const float4*>(src); traw[k][1] = *reinterpret_cast<const float4*>(src + 4);
    }
    const int m = c16;
    // a valid address of >= E floats either way; the value is dropped without a bias
    const float* bsrc = bias ? bias : (Q16 ? reinterpret_cast<const float*>(q16) : q32);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int eb = part * DA_COLS + st * 32;
        const int e_t0 = eb + 8 * (m >> 2) + (m & 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (Q16) {
                qb16[st][h] = load_bf16x8(q16 + (size_t)(e_t0 + 4 * h) * 64 + which * 32 + 8 * g);
            } else {
                const float* src = q32 + (size_t)(e_t0 + 4 * h) * 64 + which * 32 + 8 * g;
                qraw[st][h][0] = *reinterpret_cast<const float4*>(src); qraw[st][h][1] = *reinterpret_cast<const float4*>(src + 4);
            }
        }
        braw[st][0] = *reinterpret_cast<const float4*>(bsrc + eb + 8 * g);
        braw[st][1] = *reinterpret_cast<const float4*>(bsrc + eb + 8 * g + 4);
    }
    // every request above has landed before the first store goes out: an s_waitcnt the compiler SEES (it folds existing ones into
    // its model) -- otherwise, past the first conditional store, it no longer knows which loads are complete and waits with
    // vmcnt(0) before every further store, i.e. for the previous store
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) only
    bf16x8 th[DA_RG], tl[DA_RG];
#pragma unroll
    for (int k = 0; k < DA_RG; ++k) split_bf16v(traw[k][0], traw[k][1], th[k], tl[k]);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int eb = part * DA_COLS + st * 32;
        bf16x8 q0h, q0l, q1h, q1l;
        if constexpr (Q16) {
            q0h = qb16[st][0]; q1h = qb16[st][1]; q0l = q0h; q1l = q1h;      // (the lo parts are not used)
        } else {
            split_bf16v(qraw[st][0][0], qraw[st][0][1], q0h, q0l);
            split_bf16v(qraw[st][1][0], qraw[st][1][1], q1h, q1l);
        }
        float bb[8] = {braw[st][0].x, braw[st][0].y, braw[st][0].z, braw[st][0].w, braw[st][1].x, braw[st][1].y, braw[st][1].z, braw[st][1].w};
        if (!bias) {
#pragma unroll
            for (int i = 0; i < 8; ++i) bb[i] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < DA_RG; ++k) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!Q16) a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0l, th[k], a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0h, tl[k], a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0h, th[k], a0, 0, 0, 0);
            if constexpr (!Q16) a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1l, th[k], a1, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1h, tl[k], a1, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1h, th[k], a1, 0, 0, 0);
            // lane: column rr, rows 4g+r of each tile -> e = eb + 8g + r (tile 0), eb + 8g + 4 + r (tile 1)
            float o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = (float)cur[st][k][r] + ascale * a0[r] + bb[r];
                o[4 + r] = (float)cur[st][k][4 + r] + ascale * a1[r] + bb[4 + r];
            }
            if (rok[k]) {
                st_store8<ST>(base, boff[k] + eb + 8 * g, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// address of the 64-element head row that holds flat elements [rr*E + e0, +64) of dDelta
__device__ __forceinline__ const bf16* ddelta_slab(const bf16* dqkv, int ld, int col0, int rr, int e0, int E,
                                                   int H, int N) {
    const int c = (int)(((long long)rr * E + e0) >> 6);
    const int bh = c / N, n = c - bh * N;
    const int b = bh / H, h = bh - b * H;
    return dqkv + ((size_t)b * N + n) * ld + col0 + h * 64;
}

// u = dDelta . Q : block = 16*LU_RG reference rows, the LU_WAVES waves split E (contraction); LDS reduction.
// Every wave requests all of its dDelta fragments (HBM) up front; each Q fragment (L2) is used by LU_RG row groups.
#ifndef LU_RG
#define LU_RG 2
#endif
#ifndef LU_WAVES
#define LU_WAVES 8
#endif
// body: W waves split E (contraction), RG groups of 16 reference rows; red = W * RG * 4 * 64 * 4 floats of LDS
template <int W, int RG>
__device__ __forceinline__ void lowrank_u_body(float* red_raw, int blk, const bf16* __restrict__ dqkv, int ld,
                                               const bf16* __restrict__ qT, float* __restrict__ u32,
                                               bf16* __restrict__ ucols, int B, int H, int N, int E) {
    float (*red)[RG][4][64][4] = reinterpret_cast<float (*)[RG][4][64][4]>(red_raw);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int T = B * N;
    const int rr0 = blk * 16 * RG;
    int rr[RG];
#pragma unroll
    for (int k = 0; k < RG; ++k) { rr[k] = rr0 + 16 * k + c16; rr[k] = rr[k] < T ? rr[k] : T - 1; }
    f32x4 acc[RG][4];
#pragma unroll
    for (int k = 0; k < RG; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int LU_UNROLL = 3;
    const int steps = E / 32;
    if (wid < W) {
    for (int sb = wid; sb < steps; sb += W * LU_UNROLL) {
        bf16x8 aq[LU_UNROLL][RG], av[LU_UNROLL][RG];
#pragma unroll
        for (int i = 0; i < LU_UNROLL; ++i) {
            const int s = min(sb + W * i, steps - 1);
            const int e0 = (s >> 1) * 64, doff = (s & 1) * 32 + 8 * g;
#pragma unroll
            for (int k = 0; k < RG; ++k) {
                aq[i][k] = load_bf16x8(ddelta_slab(dqkv, ld, 0, rr[k], e0, E, H, N) + doff);
                av[i][k] = load_bf16x8(ddelta_slab(dqkv, ld, 2 * E, rr[k], e0, E, H, N) + doff);
            }
        }
#pragma unroll
        for (int i = 0; i < LU_UNROLL; ++i) {
            const int s = sb + W * i;
            if (s < steps) {
                const int ke = 32 * s + 8 * g;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const bf16x8 bq = load_bf16x8(qT + (size_t)(16 * nt + c16) * E + ke);
                    const bf16x8 bv = load_bf16x8(qT + (size_t)(32 + 16 * nt + c16) * E + ke);
#pragma unroll
                    for (int k = 0; k < RG; ++k) {
                        acc[k][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[i][k], bq, acc[k][nt], 0, 0, 0);
                        acc[k][2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i][k], bv, acc[k][2 + nt], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < RG; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wid][k][i][lane][r] = acc[k][i][r];
    }
    __syncthreads();
    if (wid >= W) return;
    // wave w sums (row group k, tile i) pairs w, w + W, ... in a fixed order
    for (int pi = wid; pi < RG * 4; pi += W) {
        const int k = pi >> 2, i = pi & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) v += red[w][k][i][lane][r];
            const int rro = rr0 + 16 * k + 4 * g + r;
            if (rro < T) {
                const int row = row_of_ref(rro, B, N);
                const int col = (i >> 1) * 32 + (i & 1) * 16 + c16;
                u32[(size_t)row * 64 + col] = v;
                ucols[(size_t)row * ld + col] = f2bf(v);
            }
        }
    }
}

__global__ __launch_bounds__(64 * LU_WAVES) void lowrank_u_kernel(const bf16* __restrict__ dqkv, int ld,
                                                        const bf16* __restrict__ qT, float* __restrict__ u32,
                                                        bf16* __restrict__ ucols, int B, int H, int N, int E) {
    __shared__ float red[LU_WAVES * LU_RG * 4 * 64 * 4];
    lowrank_u_body<LU_WAVES, LU_RG>(red, blockIdx.x, dqkv, ld, qT, u32, ucols, B, H, N, E);
}

// ---------------------------------------------------------------------------------
// Token-contracted products  G = X^T Y  (contraction over the token rows):
//   kind 0: X = xn,  Y = [u_q | u_v]  -> G0 = dP_q, G1 = dP_v
//   kind 1: X = dDelta_q (flat view), Y = t_q  -> G2 = dQ_q  (+ column sums -> d bias)
//   kind 2: X = dDelta_v,             Y = t_v  -> G3 = dQ_v  (+ column sums)
// The MFMA wants the contraction index contiguous per lane, but both operands are stored
// token-major.  One workgroup = (chunk of LG_ROWS tokens, kind, es slabs of 64 columns e).  Both panels go to LDS
// ROW-major (X: 16-byte writes as loaded; Y: f32 -> bf16, 8-byte writes) and every MFMA fragment is a pair of
// ds_read_b64_tr_b16 (the gfx950 transposing read: in a 16-lane group lane 4j+q passes the address of 4 consecutive
// columns of token-row j and lane i receives column i of that 4 x 16 block; scripts/probe_tr_b16.hip), k-slot idx of
// lane group g = token 32ks + 4g + idx (idx < 4) and 32ks + 16 + 4g + idx - 4.  The Y fragments are the same for every
// slab: they are read once into registers, and the X panel of the next slab is in flight (registers) while the current
// one is multiplied.  One deterministic partial per chunk goes to HBM.
#ifndef LG_ES_V
#define LG_ES_V 0        // 0: the launcher picks the slabs per workgroup (lg_pick_es); > 0 pins it (measurement builds)
#endif
constexpr int LG_ROWS = 256;
constexpr int LG_LD = 72;          // row stride (elements) of the LDS tiles: 36 dwords, 8 consecutive rows cover all banks

__device__ __forceinline__ bf16x8 lg_trfrag(const bf16* tile, int ks, int col0, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int m = lane & 15, g = lane >> 4;
    const bf16* src = tile + (32 * ks + 4 * g + (m >> 2)) * LG_LD + col0 + 4 * (m & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(src));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(src + 16 * LG_LD));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
}

// LDS of the body: Xs, Ys (LG_ROWS x LG_LD bf16 each) + 4 x 64 floats
constexpr int LG_LDS_BYTES = 2 * LG_ROWS * LG_LD * 2 + 4 * 64 * 4;
// One (chunk, kind, slab group) workgroup of 256 threads.  kind_lo / nkinds select which kinds a launch (or a block range of the
// combined launch) covers: {0, 3} everything, {1, 2} the two dQ products, {0, 1} dP alone.
__device__ __forceinline__ void lowrank_grad_body(char* smem, int bid, int kind_lo, int nkinds, const bf16* __restrict__ xn, int ldx,
                                                  const float* __restrict__ u32, const bf16* __restrict__ dqkv, int ld,
                                                  const float* __restrict__ t, float* __restrict__ partial,
                                                  float* __restrict__ dbias_partial, int B, int H, int N, int E, int es) {
    bf16* Xs = reinterpret_cast<bf16*>(smem);
    bf16* Ys = Xs + LG_ROWS * LG_LD;
    float (*cs)[64] = reinterpret_cast<float (*)[64]>(smem + 2 * LG_ROWS * LG_LD * 2);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int groups = E / 64 / es;                         // es = 64-column slabs per workgroup
    const int per_chunk = groups * nkinds;
    // the `groups` workgroups of a (chunk, kind) read the same Y rows (64 KB of u / 32 KB of t): the caller hands out consecutive
    // LOGICAL ids on one XCD (workgroup b runs on XCD b % 8), so that its L2 fetches them once instead of up to six L2s once each
    const int chunk = bid / per_chunk, rem = bid - chunk * per_chunk;
    const int kind = kind_lo + rem / groups, eg = rem % groups;
    const int T = B * N;
    const int r0 = chunk * LG_ROWS;
    const int col0 = (kind == 1) ? 0 : 2 * E;
    const int toff = (kind == 1) ? 0 : 32;
    const int c = tid & 7;

    // No load of this kernel sits inside a bounds branch: rows beyond T read row T - 1 and are zeroed by a select.  (`v = 0;
    // if (r < T) v = load` makes hipcc drain with vmcnt(0) right behind every such request -- the 16 Y loads below were 16
    // consecutive memory round trips; scripts/isa_serial_loads.py finds the pattern in a -S listing.)
    bf16x8 xv[LG_ROWS / 32];
    auto load_x = [&](int e0) {
#pragma unroll
        for (int it = 0; it < LG_ROWS / 32; ++it) {
            const int r = r0 + (tid >> 3) + 32 * it;
            const int rc = r < T ? r : T - 1;
            const bf16* src = (kind == 0) ? xn + (size_t)rc * ldx + e0 : ddelta_slab(dqkv, ld, col0, rc, e0, E, H, N);
            xv[it] = load_bf16x8(src + 8 * c);
        }
    };
    auto mask_x = [&]() {                                    // after the loads have been consumed from registers: zero the rows beyond T
#pragma unroll
        for (int it = 0; it < LG_ROWS / 32; ++it)
            if (r0 + (tid >> 3) + 32 * it >= T) xv[it] = zero_bf16x8();
    };
    load_x(eg * es * 64);
    // Y (f32): kind 0 -> 64 columns of u (16 float4 per row); else 32 columns of t (8 float4 per row)
    const int sh = (kind == 0) ? 4 : 3;                     // log2(float4 groups per row)
    const int nY = (kind == 0) ? 16 : 8;                    // loads per thread
    {
        float4 yv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * (it < nY ? it : 0);   // the rounds beyond nY repeat round 0 (L1 hits) and are not stored
            const int y = idx >> sh, c4 = idx & ((1 << sh) - 1);
            const int r = r0 + y, rc = r < T ? r : T - 1;
            const float* src = (kind == 0) ? u32 + (size_t)rc * 64 : t + (size_t)row_of_ref(rc, B, N) * 64 + toff;
            yv[it] = *reinterpret_cast<const float4*>(src + 4 * c4);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if (it < nY) {
                const int idx = tid + 256 * it;
                const int y = idx >> sh, j = 4 * (idx & ((1 << sh) - 1));
                const bool rok = r0 + y < T;
                bf16x4 o;
                o[0] = f2bf(rok ? yv[it].x : 0.f); o[1] = f2bf(rok ? yv[it].y : 0.f);
                o[2] = f2bf(rok ? yv[it].z : 0.f); o[3] = f2bf(rok ? yv[it].w : 0.f);
                *reinterpret_cast<bf16x4*>(Ys + y * LG_LD + j) = o;
            }
        }
    }
    __syncthreads();
    // Y fragments: col j = 16nt + c16, the same for every slab
    const int NT = (kind == 0) ? 4 : 2;
    bf16x8 bfr[LG_ROWS / 32][4];
#pragma unroll
    for (int ks = 0; ks < LG_ROWS / 32; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            bfr[ks][nt] = (nt < NT) ? lg_trfrag(Ys, ks, 16 * nt, lane) : zero_bf16x8();

    const size_t plane = (size_t)E * 32;
    float* base = partial + ((size_t)chunk * 4 + (kind == 0 ? 0 : kind + 1)) * plane;
    for (int sl = 0; sl < es; ++sl) {
        const int e0 = (eg * es + sl) * 64;
        // ---- this slab's X panel: registers -> LDS (row-major), column sums on the way ----
        float colsum[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) colsum[i] = 0.f;
        mask_x();
#pragma unroll
        for (int it = 0; it < LG_ROWS / 32; ++it) {
            const int y = (tid >> 3) + 32 * it;
            *reinterpret_cast<bf16x8*>(Xs + y * LG_LD + 8 * c) = xv[it];
#pragma unroll
            for (int i = 0; i < 8; ++i) colsum[i] += bf2f(xv[it][i]);
        }
        if (kind != 0) {     // column sums: reduce over the 8 row-lanes of this wave, then over waves
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float a = colsum[i];
                a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
                if ((lane >> 3) == 0) cs[wid][8 * c + i] = a;
            }
        }
        __syncthreads();
        if (sl + 1 < es) load_x(e0 + 64);                   // in flight during the MFMAs below
        // ---- MFMA: wave w owns rows e = 16w..16w+15 of the 64 x (64|32) tile ----
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < LG_ROWS / 32; ++ks) {
            const bf16x8 a = lg_trfrag(Xs, ks, 16 * wid, lane);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (nt < NT) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[ks][nt], acc[nt], 0, 0, 0);
        }
        // the next slab's X panel (requested above) is in before this slab's stores go out: the compiler then knows that no load
        // is pending behind them and does not wait for the STORES when the panel is used (vmcnt counts both on gfx950)
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) only
        // C layout: col j = 16nt + c16, rows e = 16w + 4g + reg
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt < NT) {
                float* out = base + (nt >> 1) * plane;           // kind 0: j-tiles 2,3 are the u_v plane
                const int j = 16 * (nt & 1) + c16;
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(e0 + 16 * wid + 4 * g + r) * 32 + j] = acc[nt][r];
            }
        }
        if (kind != 0 && tid < 64)
            dbias_partial[((size_t)chunk * 2 + (kind - 1)) * E + e0 + tid] = cs[0][tid] + cs[1][tid] + cs[2][tid] + cs[3][tid];
        __syncthreads();                                   // Xs and cs are rewritten by the next slab
    }
}

__global__ __launch_bounds__(256, 2) void lowrank_grad_kernel(const bf16* __restrict__ xn, int ldx,
                                                              const float* __restrict__ u32,
                                                              const bf16* __restrict__ dqkv, int ld,
                                                              const float* __restrict__ t, float* __restrict__ partial,
                                                              float* __restrict__ dbias_partial, int B, int H, int N,
                                                              int E, int remap, int es) {
    __shared__ __attribute__((aligned(16))) char smem[LG_LDS_BYTES];
    const int bid = remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    lowrank_grad_body(smem, bid, 0, 3, xn, ldx, u32, dqkv, ld, t, partial, dbias_partial, B, H, N, E, es);
}

// ONE launch per layer for the whole low-rank backward (round 4): block ranges, each padded to a multiple of 8 so that a range's
// logical ids keep their XCD (block b runs on XCD b % 8):
//   [0, nu)            u = dDelta Q of THIS layer (four waves, 16 * LC_URG reference rows per block)
//   [nu, nu + n12)     dQ_q, dQ_v (+ bias column sums) of THIS layer
//   [.., + n0)         dP = xn^T u of the layer processed BEFORE this one (its u is complete: it was written by the previous launch)
// u and the dQ products both read dq / dv while they are hot; dP needs the finished u, hence its one-launch delay (the last layer's
// dP gets a launch of its own).  Replaces lowrank_u + lowrank_grad (24.4 us, two dispatches) per layer.
#ifndef LC_URG_V
#define LC_URG_V 2
#endif
constexpr int LC_UW = 4, LC_URG = LC_URG_V;      // waves and 16-row groups of a u block
// HC, NC: heads and tokens per image as compile-time constants (0 = the runtime arguments): 17.5 -> 16.3 us at H = 12, N = 50 (round 5)
template <int HC, int NC>
__global__ __launch_bounds__(256, 2) void lowrank_combo_kernel(const bf16* __restrict__ dqkv, int ld, const bf16* __restrict__ qT,
                                                               float* __restrict__ u32, bf16* __restrict__ ucols,
                                                               const float* __restrict__ t, float* __restrict__ partial,
                                                               float* __restrict__ dbias_partial,
                                                               const bf16* __restrict__ xn_prev, int ldx, const float* __restrict__ u32_prev,
                                                               float* __restrict__ partial_prev, int nu, int nu_pad, int n12, int n12_pad,
                                                               int n0, int B, int H_rt, int N_rt, int E_rt, int es) {
    const int H = HC ? HC : H_rt, N = NC ? NC : N_rt, E = HC ? 64 * HC : E_rt;
    __shared__ __attribute__((aligned(16))) char smem[LG_LDS_BYTES];
    static_assert(LC_UW * LC_URG * 4 * 64 * 4 * 4 <= LG_LDS_BYTES, "the u reduction fits the gradient body's LDS");
    int b = blockIdx.x;
    if (b < nu_pad) {
        if (b < nu) lowrank_u_body<LC_UW, LC_URG>(reinterpret_cast<float*>(smem), b, dqkv, ld, qT, u32, ucols, B, H, N, E);
        return;
    }
    b -= nu_pad;
    // (one call site for both gradient ranges: three inlined copies of the body crash hipcc's inliner)
    const bool this_layer = b < n12_pad;
    const int local = this_layer ? b : b - n12_pad;
    const int n0_pad = (n0 + 7) & ~7;
    const int bid = xcd_remap(local, this_layer ? n12_pad : n0_pad);
    if (bid >= (this_layer ? n12 : n0)) return;
    lowrank_grad_body(smem, bid, this_layer ? 1 : 0, this_layer ? 2 : 1, xn_prev, ldx, u32_prev, dqkv, ld, t,
                      this_layer ? partial : partial_prev, dbias_partial, B, H, N, E, es);
}

// sum the per-chunk partials of every layer (blockIdx.y): G[l][4][E][32], and the bias gradient
__global__ void lowrank_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ dbias_partial,
                                      int chunks, float* __restrict__ G, float* g_b, int E, size_t partial_layer,
                                      size_t dbias_layer, size_t gb_layer) {
    // Four consecutive elements per thread (16-byte requests; round 5: was one), eight chunks per round (one memory round trip each:
    // 25 chunks = 4 trips).  Fixed summation tree per element -- the same one as before: deterministic, same bits.
    const int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int total = 4 * E * 32, l = blockIdx.y;
    partial += (size_t)l * partial_layer;
    if (idx < total) {
        float4 sacc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        int c = 0;
        for (; c + 7 < chunks; c += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(partial + (size_t)(c + u) * total + idx);
#pragma unroll
            for (int u = 0; u < 8; ++u) { sacc[u].x += v[u].x; sacc[u].y += v[u].y; sacc[u].z += v[u].z; sacc[u].w += v[u].w; }
        }
        {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(partial + (size_t)min(c + u, chunks - 1) * total + idx);    // clamped: no load inside a branch
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c + u < chunks) { sacc[u].x += v[u].x; sacc[u].y += v[u].y; sacc[u].z += v[u].z; sacc[u].w += v[u].w; }
        }
        float4 o;
        o.x = ((sacc[0].x + sacc[1].x) + (sacc[2].x + sacc[3].x)) + ((sacc[4].x + sacc[5].x) + (sacc[6].x + sacc[7].x));
        o.y = ((sacc[0].y + sacc[1].y) + (sacc[2].y + sacc[3].y)) + ((sacc[4].y + sacc[5].y) + (sacc[6].y + sacc[7].y));
        o.z = ((sacc[0].z + sacc[1].z) + (sacc[2].z + sacc[3].z)) + ((sacc[4].z + sacc[5].z) + (sacc[6].z + sacc[7].z));
        o.w = ((sacc[0].w + sacc[1].w) + (sacc[2].w + sacc[3].w)) + ((sacc[4].w + sacc[5].w) + (sacc[6].w + sacc[7].w));
        *reinterpret_cast<float4*>(G + (size_t)l * total + idx) = o;
    }
    // the bias gradient: one column per thread of the LAST blocks of the grid (they have the fewest elements to sum), the chunks'
    // two partial rows requested eight chunks at a time (round 5: the plain loop made every chunk its own memory round trip -- 25
    // of them in a row while the rest of the grid had long finished); summed in chunk order as before: same bits
    const int col = ((int)(gridDim.x - 1 - blockIdx.x) * (int)blockDim.x + (int)threadIdx.x);
    if (g_b && col < E) {
        const float* db = dbias_partial + (size_t)l * dbias_layer;
        float s = 0.f;
        for (int c0 = 0; c0 < chunks; c0 += 8) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = min(c0 + u, chunks - 1);
                a[u] = db[((size_t)c * 2) * E + col]; b[u] = db[((size_t)c * 2 + 1) * E + col];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u < chunks) s += a[u] + b[u];
        }
        g_b[(size_t)l * gb_layer + col] += s;
    }
}

// KAdaptation chain rule, one block per (rank index j, layer).  The per-layer factors
// are written straight into the flat gradient buffer; the contributions to the shared phm_rule
// factors go to rule_scratch[l][4][32][32] and are summed over layers by rule_sum_kernel (a
// fixed-order, deterministic sum).
__global__ __launch_bounds__(256) void chain_kadapt_kernel(const float* __restrict__ G, float ascale,
                                                           const float* __restrict__ rule1_l, const float* __restrict__ rule1_r,
                                                           const float* __restrict__ rule2_l, const float* __restrict__ rule2_r,
                                                           const float* __restrict__ q_left, const float* __restrict__ q_right,
                                                           float* __restrict__ rule_scratch, float* g_q_left,
                                                           float* g_q_right, int E, size_t param_layer) {
    extern __shared__ float gs[];      // [4][E]
    const int j = blockIdx.x, l = blockIdx.y, F = E / 32, tid = threadIdx.x;
    G += (size_t)l * 4 * E * 32;
    q_left += (size_t)l * param_layer; q_right += (size_t)l * param_layer;
    g_q_left += (size_t)l * param_layer; g_q_right += (size_t)l * param_layer;
    float* rs = rule_scratch + (size_t)l * 4096;
    // column j of G into LDS, eight requests per thread in flight (round 5: one per trip made the 12 trips of E = 768 twelve memory
    // round trips: this kernel's 12 us)
    for (int i0 = tid; i0 < 4 * E; i0 += 8 * blockDim.x) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = G[(size_t)min(i0 + u * (int)blockDim.x, 4 * E - 1) * 32 + j];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * (int)blockDim.x < 4 * E) gs[i0 + u * blockDim.x] = v[u];
    }
    __syncthreads();
    const float* G0 = gs; const float* G1 = gs + E; const float* G2 = gs + 2 * E; const float* G3 = gs + 3 * E;
    const float* lf = q_left + j * F; const float* r = q_right + j * F;
    const float* s1 = rule1_l + j * 32; const float* t1 = rule1_r + j * 32;
    const float* s2 = rule2_l + j * 32; const float* t2 = rule2_r + j * 32;
    if (tid < 32) {                                  // d s1[a], d s2[a]
        float a1 = 0.f, a2 = 0.f;
        for (int k = 0; k < F; ++k) { a1 += G0[tid * F + k] * lf[k]; a2 += G1[tid * F + k] * lf[k]; }
        rs[0 * 1024 + j * 32 + tid] = ascale * a1;    // rule1_left
        rs[2 * 1024 + j * 32 + tid] = ascale * a2;    // rule2_left
    } else if (tid < 64) {                           // d t1[c], d t2[c]
        const int c = tid - 32;
        float a1 = 0.f, a2 = 0.f;
        for (int p = 0; p < F; ++p) { a1 += G2[c * F + p] * r[p]; a2 += G3[c * F + p] * r[p]; }
        rs[1 * 1024 + j * 32 + c] = ascale * a1;      // rule1_right
        rs[3 * 1024 + j * 32 + c] = ascale * a2;      // rule2_right
    } else if (tid < 64 + F) {                       // d l[k]  (q and v paths share Wq: SURVEY 9.1)
        const int k = tid - 64;
        float a = 0.f;
        for (int aa = 0; aa < 32; ++aa) a += G0[aa * F + k] * s1[aa] + G1[aa * F + k] * s2[aa];
        g_q_left[j * F + k] += ascale * a;
    } else if (tid < 64 + 2 * F) {                   // d r[p]
        const int p = tid - 64 - F;
        float a = 0.f;
        for (int c = 0; c < 32; ++c) a += G2[c * F + p] * t1[c] + G3[c * F + p] * t2[c];
        g_q_right[j * F + p] += ascale * a;
    }
}

// g_rule[0:4096] += rule_scratch[l][0:4096] for l = l_hi-1 .. l_lo, ONE running sum that starts from the value already in
// g_rule (rule1_left | rule1_right | rule2_left | rule2_right).  A backward over [l_lo, l_hi) adds exactly its own layers, and a
// tower walked in several ranges from the top (data parallelism: (L, L/2) then (L/2, 0); a block-by-block autograd walk)
// performs the same additions in the same order as the one-call backward: bit-identical shared-rule gradients.
__global__ void rule_sum_kernel(const float* __restrict__ rule_scratch, float* g_rule, int l_lo, int l_hi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4096) return;
    float s = g_rule[i];
    // top layer first, one running sum (the order the staged backward relies on); eight layers' values requested per trip
    for (int l0 = l_hi - 1; l0 >= l_lo; l0 -= 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = rule_scratch[(size_t)max(l0 - u, l_lo) * 4096 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (l0 - u >= l_lo) s += v[u];
    }
    g_rule[i] = s;
}

__global__ void chain_lora_kernel(const float* __restrict__ G, float ascale, int r, float* g_a1q, float* g_a2q,
                                  float* g_a1v, float* g_a2v, int E, size_t param_layer) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * r) return;
    G += (size_t)blockIdx.y * 4 * E * 32;
    { const size_t o = (size_t)blockIdx.y * param_layer; g_a1q += o; g_a2q += o; g_a1v += o; g_a2v += o; }
    const int e = idx / r, j = idx - e * r;
    const size_t plane = (size_t)E * 32;
    g_a1q[(size_t)j * E + e] += ascale * G[(size_t)e * 32 + j];
    g_a1v[(size_t)j * E + e] += ascale * G[plane + (size_t)e * 32 + j];
    g_a2q[(size_t)e * r + j] += ascale * G[2 * plane + (size_t)e * 32 + j];
    g_a2v[(size_t)e * r + j] += ascale * G[3 * plane + (size_t)e * 32 + j];
}

}  // namespace

int pevit_launch_prep_kadapt(const float* rule1_l, const float* rule1_r, const float* rule2_l, const float* rule2_r,
                             const float* q_left, const float* q_right, AdapterPanels pan, int E, float ascale,
                             int layers, LayerStrides st, hipStream_t s, int f32) {
    if (E % 32) { pevit_set_error("prep_kadapt: width %d not divisible by phm_dim 32", E); return -1; }
    if (f32) hipLaunchKernelGGL(prep_kadapt_kernel<float>, dim3(ceil_div(E * 32, 256), layers, 2), dim3(256), 0, s, rule1_l, rule1_r,
                                rule2_l, rule2_r, q_left, q_right, pan, E, ascale, st);
    else hipLaunchKernelGGL(prep_kadapt_kernel<bf16>, dim3(ceil_div(E * 32, 256), layers, 2), dim3(256), 0, s, rule1_l, rule1_r,
                            rule2_l, rule2_r, q_left, q_right, pan, E, ascale, st);
    LAUNCH_OK("prep_kadapt_kernel");
    return 0;
}

int pevit_launch_prep_lora(const float* a1q, const float* a2q, const float* a1v, const float* a2v, int r,
                           AdapterPanels pan, int E, float ascale, int layers, LayerStrides st, hipStream_t s, int f32) {
    if (r < 1 || r > 32) { pevit_set_error("prep_lora: rank %d outside [1,32]", r); return -1; }
    if (f32) hipLaunchKernelGGL(prep_lora_kernel<float>, dim3(ceil_div(E * 32, 256), layers), dim3(256), 0, s, a1q, a2q, a1v, a2v, r,
                                pan, E, ascale, st);
    else hipLaunchKernelGGL(prep_lora_kernel<bf16>, dim3(ceil_div(E * 32, 256), layers), dim3(256), 0, s, a1q, a2q, a1v, a2v, r,
                            pan, E, ascale, st);
    LAUNCH_OK("prep_lora_kernel");
    return 0;
}

int pevit_launch_delta_add(bf16* qbuf, bf16* vbuf, const float* t, const float* q32, const bf16* q16, const float* bias, float ascale,
                           int B, int N, int E, hipStream_t s, int f32) {
    if (E % DA_COLS) { pevit_set_error("delta_add: width %d must be a multiple of %d", E, DA_COLS); return -1; }
    const int T = B * N;
    const int waves = ceil_div(T, 16 * DA_RG) * (E / DA_COLS);
    if (f32) hipLaunchKernelGGL(delta_add_kernel<float>, dim3(ceil_div(waves, 4), 2), dim3(256), 0, s, qbuf, vbuf, t, q32, q16, bias, ascale,
                                B, N, E);
    else hipLaunchKernelGGL(delta_add_kernel<bf16>, dim3(ceil_div(waves, 4), 2), dim3(256), 0, s, qbuf, vbuf, t, q32, q16, bias, ascale,
                            B, N, E);
    LAUNCH_OK("delta_add_kernel");
    return 0;
}

int pevit_launch_lowrank_u(const bf16* dqkv, int ld, const bf16* qT, float* u32, bf16* u_bf16_cols, int B, int H,
                           int N, int E, hipStream_t s) {
    const int T = B * N;
    hipLaunchKernelGGL(lowrank_u_kernel, dim3(ceil_div(T, 16 * LU_RG)), dim3(64 * LU_WAVES), 0, s, dqkv, ld, qT, u32, u_bf16_cols, B, H,
                       N, E);
    LAUNCH_OK("lowrank_u_kernel");
    return 0;
}

int pevit_lowrank_chunks(int T) { return ceil_div(T, LG_ROWS); }

// 64-column slabs per gradient workgroup.  Two workgroups fit a CU (LDS), so a launch of <= 2 * CUs blocks runs as ONE wave of
// workgroups; a block costs about (2 + es) slab times (its Y panel and B fragments are loaded once, then es slabs of X), so the
// launch costs  waves * (2 + es).  ViT-B/32, batch 128 (25 chunks, 12 slabs, 200 u blocks): es = 2 -> 650 blocks, two waves,
// 21.4 us;  es = 3 -> 500 blocks, one wave, 17.8 us;  es = 4 -> 425 blocks, one wave of longer blocks, 20.7 us (measured, round 4).
static int lg_pick_es(int E, int chunks, int kinds, int other_blocks) {
    const int slabs = E / 64;
    if (LG_ES_V > 0) return (slabs % LG_ES_V == 0) ? LG_ES_V : 0;
    const int slots = 2 * pevit_num_cus();
    int best = 0, best_cost = 0;
    for (int es = 1; es <= 8; ++es) {
        if (slabs % es) continue;
        const int total = other_blocks + chunks * (slabs / es) * kinds;
        const int cost = ceil_div(total, slots) * (2 + es);
        if (!best || cost < best_cost) { best = es; best_cost = cost; }
    }
    return best;
}

// see lowrank_combo_kernel.  this_layer = 0: only the deferred dP of the previous layer (end of the layer loop); prev = 0: no deferred work.
int pevit_launch_lowrank_combo(int this_layer, int prev, const bf16* dqkv, int ld, const bf16* qT, float* u32, bf16* ucols, const float* t,
                               float* partial, float* dbias_partial, const bf16* xn_prev, int ldx, const float* u32_prev,
                               float* partial_prev, int B, int H, int N, int E, hipStream_t s) {
    const int T = B * N, chunks = ceil_div(T, LG_ROWS);
    if (E % 64) { pevit_set_error("lowrank_combo: width %d must be a multiple of 64", E); return -1; }
    const int nu = this_layer ? ceil_div(T, 16 * LC_URG) : 0;
    const int es = lg_pick_es(E, chunks, (this_layer ? 2 : 0) + (prev ? 1 : 0), nu);
    if (!es) { pevit_set_error("lowrank_combo: no slab grouping for width %d", E); return -1; }
    const int groups = E / 64 / es;
    const int n12 = this_layer ? chunks * groups * 2 : 0, n0 = prev ? chunks * groups : 0;
    const int nu_pad = (nu + 7) & ~7, n12_pad = (n12 + 7) & ~7, n0_pad = (n0 + 7) & ~7;
    if (nu_pad + n12_pad + n0_pad == 0) return 0;
#define LC_GO(HC, NC) hipLaunchKernelGGL((lowrank_combo_kernel<HC, NC>), dim3(nu_pad + n12_pad + n0_pad), dim3(256), 0, s, dqkv, ld, qT, u32, ucols, t, partial, \
                                        dbias_partial, xn_prev, ldx, u32_prev, partial_prev, nu, nu_pad, n12, n12_pad, n0, B, H, N, E, es)
    if (E != 64 * H) LC_GO(0, 0);
    else if (H == 12 && N == 50) LC_GO(12, 50);          // ViT-B/32
    else if (H == 12 && N == 197) LC_GO(12, 197);        // ViT-B/16
    else if (H == 16 && N == 257) LC_GO(16, 257);        // ViT-L/14
    else LC_GO(0, 0);
#undef LC_GO
    LAUNCH_OK("lowrank_combo_kernel");
    return 0;
}

int pevit_launch_lowrank_grad(const bf16* xn, int ldx, const float* u32, const bf16* dqkv, int ld, const float* t,
                              float* partial, float* dbias_partial, int chunks, int B, int H, int N, int E,
                              hipStream_t s, int xcd_order) {
    const int T = B * N;
    if (chunks != ceil_div(T, LG_ROWS)) { pevit_set_error("lowrank_grad: chunks mismatch"); return -1; }
    if (E % 64) { pevit_set_error("lowrank_grad: width %d must be a multiple of 64", E); return -1; }
    const int es = lg_pick_es(E, chunks, 3, 0);
    if (!es) { pevit_set_error("lowrank_grad: no slab grouping for width %d", E); return -1; }
    hipLaunchKernelGGL(lowrank_grad_kernel, dim3(chunks * (E / 64 / es) * 3), dim3(256), 0, s, xn, ldx, u32, dqkv, ld, t,
                       partial, dbias_partial, B, H, N, E, xcd_order, es);
    LAUNCH_OK("lowrank_grad_kernel");
    return 0;
}

// partial: [layers][chunks+? ...] see capi.hip; G: [layers][4][E][32]; rule_scratch: [layers][4096]
int pevit_launch_chain_kadapt(const float* partial, size_t partial_layer, const float* dbias_partial, size_t dbias_layer,
                              int chunks, float ascale, int layers, float* G, float* rule_scratch, const float* params,
                              float* grads, size_t p_layer0, size_t p_layer_stride, int E, hipStream_t s) {
    if (64 + 2 * (E / 32) > 256) { pevit_set_error("chain_kadapt: width %d too large", E); return -1; }
    float* g_b = grads + p_layer0 + 4 * (size_t)E;
    hipLaunchKernelGGL(lowrank_reduce_kernel, dim3(ceil_div(4 * E * 32, 4 * 256), layers), dim3(256), 0, s, partial,
                       dbias_partial, chunks, G, g_b, E, partial_layer, dbias_layer, p_layer_stride);
    LAUNCH_OK("lowrank_reduce_kernel");
    const float* r = params;
    const float* lp = params + p_layer0;
    float* lg = grads + p_layer0;
    hipLaunchKernelGGL(chain_kadapt_kernel, dim3(32, layers), dim3(256), 4 * E * sizeof(float), s, G, ascale, r, r + 1024,
                       r + 2048, r + 3072, lp, lp + E, rule_scratch, lg, lg + E, E, p_layer_stride);
    LAUNCH_OK("chain_kadapt_kernel");
    return 0;
}

// shared phm_rule factors: the contributions of layers [l_lo, l_hi) (rule_scratch is indexed by absolute layer) are added
// to the flat gradient buffer, top layer first, after the chain call of that range
int pevit_launch_rule_sum(const float* rule_scratch, float* grads, int l_lo, int l_hi, hipStream_t s) {
    hipLaunchKernelGGL(rule_sum_kernel, dim3(16), dim3(256), 0, s, rule_scratch, grads, l_lo, l_hi);
    LAUNCH_OK("rule_sum_kernel");
    return 0;
}

int pevit_launch_chain_lora(const float* partial, size_t partial_layer, int chunks, float ascale, int r, int layers,
                            float* G, float* grads, size_t p_layer0, size_t p_layer_stride, int E, hipStream_t s) {
    hipLaunchKernelGGL(lowrank_reduce_kernel, dim3(ceil_div(4 * E * 32, 4 * 256), layers), dim3(256), 0, s, partial,
                       (const float*)nullptr, chunks, G, (float*)nullptr, E, partial_layer, (size_t)0, (size_t)0);
    LAUNCH_OK("lowrank_reduce_kernel");
    const size_t rE = (size_t)r * E;
    float* lg = grads + p_layer0;
    hipLaunchKernelGGL(chain_lora_kernel, dim3(ceil_div(E * r, 256), layers), dim3(256), 0, s, G, ascale, r, lg, lg + rE,
                       lg + 2 * rE, lg + 3 * rE, E, p_layer_stride);
    LAUNCH_OK("chain_lora_kernel");
    return 0;
}
