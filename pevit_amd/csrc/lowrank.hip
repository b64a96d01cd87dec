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
__global__ void prep_kadapt_kernel(const float* __restrict__ rule1_l, const float* __restrict__ rule1_r,
                                   const float* __restrict__ rule2_l, const float* __restrict__ rule2_r,
                                   const float* __restrict__ q_left, const float* __restrict__ q_right,
                                   AdapterPanels pan, int E, float ascale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * 32) return;
    const int j = idx / E, e = idx - j * E;     // e fastest: coalesced row writes
    const int F = E / 32, a = e / F, kk = e - a * F;
    const float l = q_left[j * F + kk], r = q_right[j * F + kk];
    const float pq = rule1_l[j * 32 + a] * l, pv = rule2_l[j * 32 + a] * l;
    const float qq = rule1_r[j * 32 + a] * r, qv = rule2_r[j * 32 + a] * r;
    pan.w_aug_rows[(size_t)j * pan.ldw + e] = f2bf(pq);
    pan.w_aug_rows[(size_t)(32 + j) * pan.ldw + e] = f2bf(pv);
    pan.wT_aug_cols[(size_t)e * pan.ldwT + j] = f2bf(ascale * pq);
    pan.wT_aug_cols[(size_t)e * pan.ldwT + 32 + j] = f2bf(ascale * pv);
    pan.q32[(size_t)e * 64 + j] = qq;
    pan.q32[(size_t)e * 64 + 32 + j] = qv;
    pan.qT[(size_t)j * E + e] = f2bf(qq);
    pan.qT[(size_t)(32 + j) * E + e] = f2bf(qv);
}

__global__ void prep_lora_kernel(const float* __restrict__ a1q, const float* __restrict__ a2q,
                                 const float* __restrict__ a1v, const float* __restrict__ a2v, int r,
                                 AdapterPanels pan, int E, float ascale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * 32) return;
    const int j = idx / E, e = idx - j * E;
    const float pq = j < r ? a1q[(size_t)j * E + e] : 0.f, pv = j < r ? a1v[(size_t)j * E + e] : 0.f;
    const float qq = j < r ? a2q[(size_t)e * r + j] : 0.f, qv = j < r ? a2v[(size_t)e * r + j] : 0.f;
    pan.w_aug_rows[(size_t)j * pan.ldw + e] = f2bf(pq);
    pan.w_aug_rows[(size_t)(32 + j) * pan.ldw + e] = f2bf(pv);
    pan.wT_aug_cols[(size_t)e * pan.ldwT + j] = f2bf(ascale * pq);
    pan.wT_aug_cols[(size_t)e * pan.ldwT + 32 + j] = f2bf(ascale * pv);
    pan.q32[(size_t)e * 64 + j] = qq;
    pan.q32[(size_t)e * 64 + 32 + j] = qv;
    pan.qT[(size_t)j * E + e] = f2bf(qq);
    pan.qT[(size_t)(32 + j) * E + e] = f2bf(qv);
}

// ---------------------------------------------------------------------------------
// delta-add: one wave = 128 consecutive e of ROWS_PER_WAVE reference rows; each lane keeps the
// 2 x 32 panel entries of its two columns in registers, t[row] is wave-uniform (scalar loads).
constexpr int DA_ROWS = 32;
__global__ __launch_bounds__(256) void delta_add_kernel(bf16* qbuf, bf16* vbuf, const float* __restrict__ t,
                                                        const float* __restrict__ q32, const float* __restrict__ bias,
                                                        float ascale, int B, int N, int E) {
    const int lane = threadIdx.x & 63;
    const int wg = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int which = blockIdx.y;                  // 0: q, 1: v
    const int slabs = E / 128;
    const int slab = wg % slabs, rg = wg / slabs;
    const int T = B * N;
    const int e = slab * 128 + lane * 2;
    float q0[32], q1[32];
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
        const float4 a = *reinterpret_cast<const float4*>(q32 + (size_t)e * 64 + which * 32 + j);
        const float4 b = *reinterpret_cast<const float4*>(q32 + (size_t)(e + 1) * 64 + which * 32 + j);
        q0[j] = a.x; q0[j + 1] = a.y; q0[j + 2] = a.z; q0[j + 3] = a.w;
        q1[j] = b.x; q1[j + 1] = b.y; q1[j + 2] = b.z; q1[j + 3] = b.w;
    }
    const float b0 = bias ? bias[e] : 0.f, b1 = bias ? bias[e + 1] : 0.f;
    bf16* buf = which ? vbuf : qbuf;
    const int rr_end = min((rg + 1) * DA_ROWS, T);
    for (int rr = rg * DA_ROWS; rr < rr_end; ++rr) {
        const float* tr = t + (size_t)row_of_ref(rr, B, N) * 64 + which * 32;
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float tv = tr[j];
            d0 = fmaf(tv, q0[j], d0);
            d1 = fmaf(tv, q1[j], d1);
        }
        bf16x2* p = reinterpret_cast<bf16x2*>(buf + (size_t)rr * E + e);
        bf16x2 cur = *p;
        cur[0] = f2bf(bf2f(cur[0]) + ascale * d0 + b0);
        cur[1] = f2bf(bf2f(cur[1]) + ascale * d1 + b1);
        *p = cur;
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

// u = dDelta . Q : block = 16 reference rows, the 4 waves split E; LDS reduction.
__global__ __launch_bounds__(256) void lowrank_u_kernel(const bf16* __restrict__ dqkv, int ld,
                                                        const bf16* __restrict__ qT, float* __restrict__ u32,
                                                        bf16* __restrict__ ucols, int B, int H, int N, int E) {
    __shared__ float red[4][4][64][4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int T = B * N;
    const int rr0 = blockIdx.x * 16;
    int rr = rr0 + c16; rr = rr < T ? rr : T - 1;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int steps = E / 32;
    for (int s = wid; s < steps; s += 4) {
        const int e0 = (s >> 1) * 64, doff = (s & 1) * 32 + 8 * g;
        const bf16x8 aq = load_bf16x8(ddelta_slab(dqkv, ld, 0, rr, e0, E, H, N) + doff);
        const bf16x8 av = load_bf16x8(ddelta_slab(dqkv, ld, 2 * E, rr, e0, E, H, N) + doff);
        const int ke = 32 * s + 8 * g;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const bf16x8 bq = load_bf16x8(qT + (size_t)(16 * nt + c16) * E + ke);
            const bf16x8 bv = load_bf16x8(qT + (size_t)(32 + 16 * nt + c16) * E + ke);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bq, acc[nt], 0, 0, 0);
            acc[2 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[2 + nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wid][i][lane][r] = acc[i][r];
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = red[0][i][lane][r] + red[1][i][lane][r] + red[2][i][lane][r] + red[3][i][lane][r];
                const int rro = rr0 + 4 * g + r;
                if (rro < T) {
                    const int row = row_of_ref(rro, B, N);
                    const int col = (i >> 1) * 32 + (i & 1) * 16 + c16;
                    u32[(size_t)row * 64 + col] = v;
                    ucols[(size_t)row * ld + col] = f2bf(v);
                }
            }
    }
}

// ---------------------------------------------------------------------------------
// Token-contracted products on the f32 matrix core (v_mfma_f32_32x32x2_f32): both operands
// are read "down the rows" with lanes along the contiguous dimension, so no transposes.
//   kind 0: G0 = xn^T u_q, G1 = xn^T u_v      kind 1: G2 = dDq^T t_q (+db)   kind 2: G3 = dDv^T t_v (+db)
constexpr int LG_ROWS = 256;   // rows per chunk
__global__ __launch_bounds__(256) void lowrank_grad_kernel(const bf16* __restrict__ xn, int ldx,
                                                           const float* __restrict__ u32,
                                                           const bf16* __restrict__ dqkv, int ld,
                                                           const float* __restrict__ t, float* __restrict__ partial,
                                                           float* __restrict__ dbias_partial, int B, int H, int N,
                                                           int E) {
    const int lane = threadIdx.x & 63;
    const int wg = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int etiles = E / 32;
    const int per_chunk = etiles * 3;
    const int chunk = wg / per_chunk, rem = wg - chunk * per_chunk;
    const int kind = rem / etiles, et = rem - kind * etiles;
    const int T = B * N;
    const int r_begin = chunk * LG_ROWS;
    if (r_begin >= T) return;
    const int r_end = min(r_begin + LG_ROWS, T);
    const int li = lane & 31, lk = lane >> 5;
    const int e = et * 32 + li;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float colsum = 0.f;
    if (kind == 0) {
#pragma unroll 4
        for (int r = r_begin; r < r_end; r += 2) {
            const int row = r + lk;
            const bool ok = row < r_end;
            const int rs = ok ? row : r_end - 1;
            const float a = ok ? bf2f(xn[(size_t)rs * ldx + e]) : 0.f;
            const float bq = u32[(size_t)rs * 64 + li];
            const float bv = u32[(size_t)rs * 64 + 32 + li];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc1, 0, 0, 0);
        }
    } else {
        const int col0 = (kind == 1) ? 0 : 2 * E;
        const int toff = (kind == 1) ? 0 : 32;
        const int e0 = (e >> 6) << 6, d = e & 63;
#pragma unroll 4
        for (int r = r_begin; r < r_end; r += 2) {
            const int rr = r + lk;
            const bool ok = rr < r_end;
            const int rs = ok ? rr : r_end - 1;
            const float a = ok ? bf2f(ddelta_slab(dqkv, ld, col0, rs, e0, E, H, N)[d]) : 0.f;
            const float bt = t[(size_t)row_of_ref(rs, B, N) * 64 + toff + li];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bt, acc0, 0, 0, 0);
            colsum += a;
        }
    }
    // C layout: col j = lane&31, row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const size_t plane = (size_t)E * 32;
    float* out0 = partial + ((size_t)chunk * 4 + (kind == 0 ? 0 : kind + 1)) * plane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
        out0[(size_t)(et * 32 + i) * 32 + li] = acc0[r];
        if (kind == 0) out0[plane + (size_t)(et * 32 + i) * 32 + li] = acc1[r];
    }
    if (kind != 0) {
        colsum += __shfl_xor(colsum, 32, 64);
        if (lk == 0) dbias_partial[((size_t)chunk * 2 + (kind - 1)) * E + e] = colsum;
    }
}

// sum the per-chunk partials: G[4][E][32], and the bias gradient
__global__ void lowrank_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ dbias_partial,
                                      int chunks, float* __restrict__ G, float* g_b, int E) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 4 * E * 32;
    if (idx < total) {
        float s = 0.f;
        for (int c = 0; c < chunks; ++c) s += partial[(size_t)c * total + idx];
        G[idx] = s;
    }
    if (g_b && idx < E) {
        float s = 0.f;
        for (int c = 0; c < chunks; ++c)
            s += dbias_partial[((size_t)c * 2) * E + idx] + dbias_partial[((size_t)c * 2 + 1) * E + idx];
        g_b[idx] += s;
    }
}

// KAdaptation chain rule, one block per rank index j
__global__ __launch_bounds__(256) void chain_kadapt_kernel(const float* __restrict__ G, float ascale,
                                                           const float* __restrict__ rule1_l, const float* __restrict__ rule1_r,
                                                           const float* __restrict__ rule2_l, const float* __restrict__ rule2_r,
                                                           const float* __restrict__ q_left, const float* __restrict__ q_right,
                                                           float* g_rule1_l, float* g_rule1_r, float* g_rule2_l,
                                                           float* g_rule2_r, float* g_q_left, float* g_q_right, int E) {
    extern __shared__ float gs[];      // [4][E]
    const int j = blockIdx.x, F = E / 32, tid = threadIdx.x;
    for (int i = tid; i < 4 * E; i += blockDim.x) gs[i] = G[(size_t)i * 32 + j];
    __syncthreads();
    const float* G0 = gs; const float* G1 = gs + E; const float* G2 = gs + 2 * E; const float* G3 = gs + 3 * E;
    const float* l = q_left + j * F; const float* r = q_right + j * F;
    const float* s1 = rule1_l + j * 32; const float* t1 = rule1_r + j * 32;
    const float* s2 = rule2_l + j * 32; const float* t2 = rule2_r + j * 32;
    if (tid < 32) {                                  // d s1[a], d s2[a]
        float a1 = 0.f, a2 = 0.f;
        for (int k = 0; k < F; ++k) { a1 += G0[tid * F + k] * l[k]; a2 += G1[tid * F + k] * l[k]; }
        g_rule1_l[j * 32 + tid] += ascale * a1;
        g_rule2_l[j * 32 + tid] += ascale * a2;
    } else if (tid < 64) {                           // d t1[c], d t2[c]
        const int c = tid - 32;
        float a1 = 0.f, a2 = 0.f;
        for (int p = 0; p < F; ++p) { a1 += G2[c * F + p] * r[p]; a2 += G3[c * F + p] * r[p]; }
        g_rule1_r[j * 32 + c] += ascale * a1;
        g_rule2_r[j * 32 + c] += ascale * a2;
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

__global__ void chain_lora_kernel(const float* __restrict__ G, float ascale, int r, float* g_a1q, float* g_a2q,
                                  float* g_a1v, float* g_a2v, int E) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * r) return;
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
                             hipStream_t s) {
    if (E % 32) { pevit_set_error("prep_kadapt: width %d not divisible by phm_dim 32", E); return -1; }
    hipLaunchKernelGGL(prep_kadapt_kernel, dim3(ceil_div(E * 32, 256)), dim3(256), 0, s, rule1_l, rule1_r, rule2_l,
                       rule2_r, q_left, q_right, pan, E, ascale);
    return 0;
}

int pevit_launch_prep_lora(const float* a1q, const float* a2q, const float* a1v, const float* a2v, int r,
                           AdapterPanels pan, int E, float ascale, hipStream_t s) {
    if (r < 1 || r > 32) { pevit_set_error("prep_lora: rank %d outside [1,32]", r); return -1; }
    hipLaunchKernelGGL(prep_lora_kernel, dim3(ceil_div(E * 32, 256)), dim3(256), 0, s, a1q, a2q, a1v, a2v, r, pan, E,
                       ascale);
    return 0;
}

int pevit_launch_delta_add(bf16* qbuf, bf16* vbuf, const float* t, const float* q32, const float* bias, float ascale,
                           int B, int N, int E, hipStream_t s) {
    if (E % 128) { pevit_set_error("delta_add: width %d must be a multiple of 128", E); return -1; }
    const int T = B * N;
    const int waves = (E / 128) * ceil_div(T, DA_ROWS);
    hipLaunchKernelGGL(delta_add_kernel, dim3(ceil_div(waves, 4), 2), dim3(256), 0, s, qbuf, vbuf, t, q32, bias,
                       ascale, B, N, E);
    return 0;
}

int pevit_launch_lowrank_u(const bf16* dqkv, int ld, const bf16* qT, float* u32, bf16* u_bf16_cols, int B, int H,
                           int N, int E, hipStream_t s) {
    const int T = B * N;
    hipLaunchKernelGGL(lowrank_u_kernel, dim3(ceil_div(T, 16)), dim3(256), 0, s, dqkv, ld, qT, u32, u_bf16_cols, B, H,
                       N, E);
    return 0;
}

int pevit_lowrank_chunks(int T) { return ceil_div(T, LG_ROWS); }

int pevit_launch_lowrank_grad(const bf16* xn, int ldx, const float* u32, const bf16* dqkv, int ld, const float* t,
                              float* partial, float* dbias_partial, int chunks, int B, int H, int N, int E,
                              hipStream_t s) {
    const int T = B * N;
    if (chunks != ceil_div(T, LG_ROWS)) { pevit_set_error("lowrank_grad: chunks mismatch"); return -1; }
    const int waves = chunks * (E / 32) * 3;
    hipLaunchKernelGGL(lowrank_grad_kernel, dim3(ceil_div(waves, 4)), dim3(256), 0, s, xn, ldx, u32, dqkv, ld, t,
                       partial, dbias_partial, B, H, N, E);
    return 0;
}

int pevit_launch_chain_kadapt(const float* partial, const float* dbias_partial, int chunks, float ascale,
                              const float* rule1_l, const float* rule1_r, const float* rule2_l, const float* rule2_r,
                              const float* q_left, const float* q_right, float* g_rule1_l, float* g_rule1_r,
                              float* g_rule2_l, float* g_rule2_r, float* g_q_left, float* g_q_right, float* g_b, int E,
                              hipStream_t s) {
    // G lives right behind the partials (the caller sizes the buffer for chunks+1 planes)
    float* G = const_cast<float*>(partial) + (size_t)chunks * 4 * E * 32;
    hipLaunchKernelGGL(lowrank_reduce_kernel, dim3(ceil_div(4 * E * 32, 256)), dim3(256), 0, s, partial, dbias_partial,
                       chunks, G, g_b, E);
    if (64 + 2 * (E / 32) > 256) { pevit_set_error("chain_kadapt: width %d too large", E); return -1; }
    hipLaunchKernelGGL(chain_kadapt_kernel, dim3(32), dim3(256), 4 * E * sizeof(float), s, G, ascale, rule1_l, rule1_r,
                       rule2_l, rule2_r, q_left, q_right, g_rule1_l, g_rule1_r, g_rule2_l, g_rule2_r, g_q_left,
                       g_q_right, E);
    return 0;
}

int pevit_launch_chain_lora(const float* partial, int chunks, float ascale, int r, float* g_a1q, float* g_a2q,
                            float* g_a1v, float* g_a2v, int E, hipStream_t s) {
    float* G = const_cast<float*>(partial) + (size_t)chunks * 4 * E * 32;
    hipLaunchKernelGGL(lowrank_reduce_kernel, dim3(ceil_div(4 * E * 32, 256)), dim3(256), 0, s, partial,
                       (const float*)nullptr, chunks, G, (float*)nullptr, E);
    hipLaunchKernelGGL(chain_lora_kernel, dim3(ceil_div(E * r, 256)), dim3(256), 0, s, G, ascale, r, g_a1q, g_a2q,
                       g_a1v, g_a2v, E);
    return 0;
}
