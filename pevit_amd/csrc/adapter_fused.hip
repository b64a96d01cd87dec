// Post-MLP bottleneck adapters (Adapter: adapter_model.py:264-282,330-336; Compacter: compacter_model.py:432-448,497-503) as ONE
// launch per direction (round 4).
//
//   forward   x_out = x_mid + h + up(act(down(LN_a(h))))        h = c_proj(gelu(c_fc(ln_2 x))) + b
//   before:   c_proj GEMM writing TWO f32 tensors (x_mid + h and h), LayerNorm, down GEMM (+ activation), up GEMM (+ residual):
//             four launches, ~118 MB per layer at B = 128;
//   here:     the c_proj GEMM writes its accumulators once (f32, bias not yet added), and adapter_fwd_kernel takes 32 token rows
//             per workgroup through  h = acc + b  ->  LayerNorm statistics  ->  z (bf16, saved; LDS)  ->  down product on the
//             matrix core (K = E split over the four waves, fixed-order LDS reduction)  ->  activation (saved)  ->  up product
//             ->  x_out = (x_mid + h) + up + b_up,  reading h and x_mid once: two launches, ~89 MB.
//   backward  adapter_bwd_kernel, the same skeleton: dyb rows -> LDS, d pre = (dy Wu) * act' (saved, feeds the d W_down contraction),
//             d z = d pre Wd (f32, stays in LDS), LayerNorm backward with the affine-gradient column sums, d h = dx_out + LN'(d z)
//             as the bf16 operand of the c_proj backward GEMM -- replaces the dReLU GEMM, the d z GEMM (19.7 MB written and read
//             back) and ln_bwd_affine.
// Same rounding points as the separate kernels and as oracle/emul_bf16.py: z, act (and Compacter's pre-activation), d pre in bf16;
// everything else f32.  The token-contracted weight gradients stay with tn_gemm64 (adapter.hip).
#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

// Token rows per workgroup: 32 (one 120 KB workgroup per CU: the phases of the chain -- rows in, statistics, product, activation,
// product, rows out -- follow each other, and all 200 workgroups are in the same phase at the same time).  Round 5 also built 16-row
// workgroups on 16x16x32 matrix-core tiles (AF_ROWS_V = 16: 52 KB forward / 65-76 KB backward, <= 128 VGPRs, two workgroups per CU)
// to run one's memory phases under another's products.  Measured on one box, T = 6400, E = 768: forward 24.9 us against 22.3, backward
// 29.8 against 29.8 -- a launch is ONE round of workgroups that all start together, so two on a CU are in the same phase too, and the
// half-size tiles fetch the 196 KB of adapter weights twice as often.  Kept as a compile-time option; what did pay is below
// (E / 256 as a template parameter, requests issued a phase ahead).
#ifndef AF_ROWS_V
#define AF_ROWS_V 32
#endif
constexpr int AF_ROWS = AF_ROWS_V;
static_assert(AF_ROWS == 16 || AF_ROWS == 32, "16-row (16x16x32 tiles) or 32-row (32x32x16 tiles) workgroups");
#ifndef AF_WAVES_V
#define AF_WAVES_V 8
#endif
constexpr int AF_WAVES = AF_WAVES_V;      // 8: two waves per SIMD (with 4 every phase of the single resident workgroup ran exposed: 33 us)
constexpr int AF_MAXV = 4;           // float4 per lane and row (kernel template parameter NV = E / 256): E <= 1024
constexpr int AF_RPW = AF_ROWS / AF_WAVES;                  // rows a wave owns in the row phases
constexpr int AF_RB = AF_RPW < 4 ? AF_RPW : 4;              // of which this many are in flight together (forward)
constexpr int AF_RBB = AF_ROWS == 16 ? 1 : 4;               // ... (backward: the five column accumulators leave room for one row of a 16-row tile)
constexpr int AF_CPT = AF_ROWS * 64 / (64 * AF_WAVES);      // columns of the ROWS x 64 middle operand per thread
constexpr int AF_REDLD = AF_ROWS == 16 ? 68 : 64;           // row stride of a partial product in LDS (16 rows: padded, see af_first_product)
constexpr int AF_NRED = AF_ROWS == 16 ? 2 : AF_WAVES;       // partial products that meet in LDS
static_assert(AF_RPW % AF_RB == 0 && (AF_CPT == 8 || AF_CPT == 4 || AF_CPT == 2), "row batches; 2, 4 or 8 middle columns per thread");
static_assert(AF_ROWS == 32 || AF_WAVES == 8, "16 rows: four column tiles x two K halves = eight waves");
constexpr int AF_MINW = AF_ROWS == 16 ? 4 : 2;              // waves per SIMD the kernels are compiled for: 16-row tiles want two workgroups per CU (<= 128 VGPRs;
                                                            // the E = 1024 backward does not fit them without scratch and stays at one)

struct AfLds {                       // byte offsets inside the dynamic LDS block (E-dependent)
    int zs, red, as, u, colred, total, total_bwd;
};
__host__ __device__ inline AfLds af_layout(int E) {
    AfLds l;
    const int zs_bytes = AF_ROWS * (E + 8) * 2;                  // bf16 rows, 16 bytes of padding
    const int red_bytes = AF_NRED * AF_ROWS * AF_REDLD * 4;      // partial [ROWS][64] f32 of the K-split product
    const int u_bytes = AF_ROWS * (E + 4) * 4;                   // f32 result of the second product (aliases zs + red)
    l.zs = 0; l.red = zs_bytes; l.u = 0;
    const int front = zs_bytes + red_bytes > u_bytes ? zs_bytes + red_bytes : u_bytes;
    l.as = (front + 15) & ~15;                                   // [32][72] bf16: the 64-wide middle operand
    l.colred = l.as + AF_ROWS * 72 * 2;                          // end of the block (the backward column sums alias U ...
    l.total = l.colred;
    const int cs_bytes = (AF_WAVES - 1) * 3 * E * 4;             // ... and, behind 16-row tiles, run on over the dead middle operand)
    l.total_bwd = l.total > cs_bytes ? l.total : cs_bytes;
    return l;
}

// first product: P[ROWS][64] = X[ROWS][E] (LDS, bf16) . W[64][E]^T (global, bf16), K = E split over waves; the partials meet in
// LDS and thread t sums AF_CPT consecutive columns of one row over the partials in a fixed order.  Returns this thread's sums in v.
//   32 rows: 32x32x16 tiles, two per wave, K split eight ways;
//   16 rows: 16x16x32 tiles, wave (j = wid & 3, kh = wid >> 2) owns columns 16 j .. +15 over K half kh (two partials; rows of a
//            partial 68 floats apart: the four row groups of a fragment store land on four different 16-bank sets).
__device__ __forceinline__ void af_first_product(const bf16* Xs, int ldx, const bf16* __restrict__ W, int E, float* red, int wid,
                                                 int lane, int tid, float (&v)[AF_CPT]) {
    if constexpr (AF_ROWS == 32) {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const int kw = E / AF_WAVES, k0 = wid * kw;
        const int frow = lane & 31, fk = 8 * (lane >> 5);
        for (int ks = 0; ks < kw; ks += 64) {                       // four k-steps per round: the weight fragments of a round requested together
            bf16x8 b[4][2], a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = min(k0 + ks + 16 * s, E - 16) + fk;
#pragma unroll
                for (int j = 0; j < 2; ++j) b[s][j] = load_bf16x8(W + (size_t)(32 * j + frow) * E + k);
                a[s] = *reinterpret_cast<const bf16x8*>(Xs + frow * ldx + k);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (ks + 16 * s < kw) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s][j], acc[j], 0, 0, 0);
                }
            }
        }
        float* mine = red + wid * (AF_ROWS * AF_REDLD);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                mine[row * AF_REDLD + 32 * j + frow] = acc[j][r];
            }
    } else {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const int j = wid & 3, kh = wid >> 2, m = lane & 15, kg = lane >> 4;
        const int kw = E / 2, k0 = kh * kw;
        const bf16* wrow = W + (size_t)(16 * j + m) * E;
        for (int ks = 0; ks < kw; ks += 128) {                      // four 32-wide k-steps per round, their weight fragments requested together
            bf16x8 b[4], a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = min(k0 + ks + 32 * s, E - 32) + 8 * kg;
                b[s] = load_bf16x8(wrow + k);
                a[s] = *reinterpret_cast<const bf16x8*>(Xs + m * ldx + k);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (ks + 32 * s < kw) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s], b[s], acc, 0, 0, 0);
        }
        float* mine = red + kh * (AF_ROWS * AF_REDLD);
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(4 * kg + r) * AF_REDLD + 16 * j + m] = acc[r];
    }
    __syncthreads();
    constexpr int TPR = 64 / AF_CPT;                            // threads per row
    const int row = tid / TPR, c0 = (tid % TPR) * AF_CPT;
#pragma unroll
    for (int i = 0; i < AF_CPT; ++i) v[i] = 0.f;
#pragma unroll
    for (int w = 0; w < AF_NRED; ++w) {
        const float* src = red + w * (AF_ROWS * AF_REDLD) + row * AF_REDLD + c0;
        if constexpr (AF_CPT == 2) {
            const float2 x0 = *reinterpret_cast<const float2*>(src);
            v[0] += x0.x; v[1] += x0.y;
        } else {
#pragma unroll
            for (int q = 0; q < AF_CPT / 4; ++q) {
                const float4 x0 = *reinterpret_cast<const float4*>(src + 4 * q);
                v[4 * q] += x0.x; v[4 * q + 1] += x0.y; v[4 * q + 2] += x0.z; v[4 * q + 3] += x0.w;
            }
        }
    }
}
// AF_CPT consecutive bf16 of row `row` (< T) of a saved [T][64] operand
__device__ __forceinline__ void af_load_mid(const bf16* glob, bf16 (&o)[AF_CPT]) {
    if constexpr (AF_CPT == 8) { const bf16x8 t = load_bf16x8(glob); for (int i = 0; i < 8; ++i) o[i] = t[i]; }
    else if constexpr (AF_CPT == 4) { const bf16x4 t = *reinterpret_cast<const bf16x4*>(glob); for (int i = 0; i < 4; ++i) o[i] = t[i]; }
    else { const bf16x2 t = *reinterpret_cast<const bf16x2*>(glob); o[0] = t[0]; o[1] = t[1]; }
}
__device__ __forceinline__ void af_put_mid(bf16* dst, const bf16 (&o)[AF_CPT]) {
    if constexpr (AF_CPT == 8) { bf16x8 t; for (int i = 0; i < 8; ++i) t[i] = o[i]; *reinterpret_cast<bf16x8*>(dst) = t; }
    else if constexpr (AF_CPT == 4) { bf16x4 t; for (int i = 0; i < 4; ++i) t[i] = o[i]; *reinterpret_cast<bf16x4*>(dst) = t; }
    else { bf16x2 t; t[0] = o[0]; t[1] = o[1]; *reinterpret_cast<bf16x2*>(dst) = t; }
}
// AF_CPT consecutive bf16 of the middle operand: LDS image and (row < T) the saved global copy
__device__ __forceinline__ void af_store_mid(bf16* lds, bf16* glob, bool to_glob, const bf16 (&o)[AF_CPT]) {
    af_put_mid(lds, o);
    if (to_glob) af_put_mid(glob, o);
}

// second product: U[ROWS][E] (LDS, f32, row stride E + 4) = S[ROWS][64] (LDS, bf16, row stride 72) . W[E][64]^T (global, bf16)
__device__ __forceinline__ void af_second_product(const bf16* Ss, const bf16* __restrict__ W, int E, float* U, int wid, int lane) {
    if constexpr (AF_ROWS == 32) {
        const int frow = lane & 31, fk = 8 * (lane >> 5);
        bf16x8 a[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const bf16x8*>(Ss + frow * 72 + 16 * s + fk);
        const int nfrag = E / 32;
        for (int f0 = wid; f0 < nfrag; f0 += 2 * AF_WAVES) {        // two fragments per round: 8 weight requests in flight
            bf16x8 b[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = min(f0 + AF_WAVES * u, nfrag - 1);
#pragma unroll
                for (int s = 0; s < 4; ++s) b[u][s] = load_bf16x8(W + (size_t)(32 * f + frow) * 64 + 16 * s + fk);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int f = f0 + AF_WAVES * u;
                if (f < nfrag) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[u][s], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        U[row * (E + 4) + 32 * f + frow] = acc[r];
                    }
                }
            }
        }
    } else {
        const int m = lane & 15, kg = lane >> 4;
        bf16x8 a[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) a[s] = *reinterpret_cast<const bf16x8*>(Ss + m * 72 + 32 * s + 8 * kg);
        const int nfrag = E / 16;                                   // 16-column tiles: wave w takes w, w + 8, ...
        for (int f0 = wid; f0 < nfrag; f0 += 4 * AF_WAVES) {        // four tiles per round: 8 weight requests in flight
            bf16x8 b[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = min(f0 + AF_WAVES * u, nfrag - 1);
#pragma unroll
                for (int s = 0; s < 2; ++s) b[u][s] = load_bf16x8(W + (size_t)(16 * f + m) * 64 + 32 * s + 8 * kg);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + AF_WAVES * u;
                if (f < nfrag) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s], b[u][s], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) U[(4 * kg + r) * (E + 4) + 16 * f + m] = acc[r];
                }
            }
        }
    }
}

// ACT: 0 = ReLU (Adapter), 1 = gelu_new on the bf16 pre-activation (Compacter)
template <int ACT, int NV>
__global__ __launch_bounds__(64 * AF_WAVES, AF_MINW) void adapter_fwd_kernel(const float* __restrict__ hraw, const float* __restrict__ bpr,
                                                                    const float* __restrict__ x_mid, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, const bf16* __restrict__ wd,
                                                                    const float* __restrict__ b_down, const bf16* __restrict__ wu,
                                                                    const float* __restrict__ b_up, bf16* __restrict__ z,
                                                                    float* __restrict__ mean_a, float* __restrict__ rstd_a,
                                                                    bf16* __restrict__ act, bf16* __restrict__ apre,
                                                                    float* __restrict__ x_out, int T, int E_rt, int rb) {
#ifdef AF_E_RUNTIME                                              // (A/B builds)
    const int E = E_rt;
#else
    constexpr int E = 256 * NV;                                 // (the launcher dispatches on E / 256: every offset and trip count below is a constant)
    (void)E_rt;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AfLds L = af_layout(E);
    bf16* Zs = reinterpret_cast<bf16*>(smem + L.zs);
    float* red = reinterpret_cast<float*>(smem + L.red);
    bf16* As = reinterpret_cast<bf16*>(smem + L.as);
    float* U = reinterpret_cast<float*>(smem + L.u);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r0 = blockIdx.x * rb, rend = min(r0 + rb, T);    // this workgroup owns rows [r0, rend): rb <= 32 of the 32-row tile
    const int ldz = E + 8;
    int cc[NV];                                                 // E = 256 NV: every lane column is a live one
    float4 gm[NV], bt[NV], bp[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + i * 256;
        cc[i] = c;
        gm[i] = *reinterpret_cast<const float4*>(gamma + cc[i]);
        bt[i] = *reinterpret_cast<const float4*>(beta + cc[i]);
        bp[i] = *reinterpret_cast<const float4*>(bpr + cc[i]);
    }
    // ---- phase 1: the wave's 8 rows, 4 at a time (all 8 loads of a batch in flight): h = acc + b, statistics, z, s = x_mid + h
    constexpr int RB = NV == 4 ? 1 : AF_RB;                     // E = 1024: one row in flight keeps the kernel at 128 VGPRs
    float4 s[AF_RPW][NV];
#pragma unroll
    for (int half = 0; half < AF_RPW / RB; ++half) {
        float4 hv[RB][NV], xv[RB][NV];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int row = min(r0 + wid * AF_RPW + half * RB + k, T - 1);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                hv[k][i] = *reinterpret_cast<const float4*>(hraw + (size_t)row * E + cc[i]);
                xv[k][i] = *reinterpret_cast<const float4*>(x_mid + (size_t)row * E + cc[i]);
            }
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int lr = wid * AF_RPW + half * RB + k, row = r0 + lr;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                hv[k][i].x += bp[i].x; hv[k][i].y += bp[i].y; hv[k][i].z += bp[i].z; hv[k][i].w += bp[i].w;
                sum += hv[k][i].x + hv[k][i].y + hv[k][i].z + hv[k][i].w;
            }
            const float mean = wave_sum(sum) / (float)E;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float a = hv[k][i].x - mean, b = hv[k][i].y - mean, c = hv[k][i].z - mean, d = hv[k][i].w - mean;
                q += a * a + b * b + c * c + d * d;
            }
            const float rstd = rsqrtf(wave_sum(q) / (float)E + 1e-5f);
            if (lane == 0 && row < rend) { mean_a[row] = mean; rstd_a[row] = rstd; }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float4 h4 = hv[k][i];
                bf16x4 o;
                o[0] = f2bf((h4.x - mean) * rstd * gm[i].x + bt[i].x); o[1] = f2bf((h4.y - mean) * rstd * gm[i].y + bt[i].y);
                o[2] = f2bf((h4.z - mean) * rstd * gm[i].z + bt[i].z); o[3] = f2bf((h4.w - mean) * rstd * gm[i].w + bt[i].w);
                {
                    *reinterpret_cast<bf16x4*>(Zs + lr * ldz + cc[i]) = o;
                    if (row < rend) *reinterpret_cast<bf16x4*>(z + (size_t)row * E + cc[i]) = o;
                }
                s[half * RB + k][i] = make_float4(h4.x + xv[k][i].x, h4.y + xv[k][i].y, h4.z + xv[k][i].z, h4.w + xv[k][i].w);
            }
        }
    }
    __syncthreads();
    // ---- phase 2: pre = z Wd^T + b_down, activation
    {
        float v[AF_CPT];
        af_first_product(Zs, ldz, wd, E, red, wid, lane, tid, v);
        constexpr int TPR = 64 / AF_CPT;
        const int lr = tid / TPR, c0 = (tid % TPR) * AF_CPT, row = r0 + lr;
        bf16 o[AF_CPT], pre8[AF_CPT];
#pragma unroll
        for (int i = 0; i < AF_CPT; ++i) {
            const float pv = v[i] + b_down[c0 + i];
            if constexpr (ACT == 0) { o[i] = f2bf(fmaxf(pv, 0.0f)); pre8[i] = o[i]; }
            else { pre8[i] = f2bf(pv); o[i] = f2bf(gelu_new_f(bf2f(pre8[i]))); }
        }
        af_store_mid(As + lr * 72 + c0, act + (size_t)min(row, T - 1) * 64 + c0, row < rend, o);
        if constexpr (ACT == 1) {
            if (row < rend) {
                af_put_mid(apre + (size_t)row * 64 + c0, pre8);
            }
        }
    }
    __syncthreads();                                            // As complete; Zs / red are dead: U may overwrite them
    // ---- phase 3: U = act Wu^T
    af_second_product(As, wu, E, U, wid, lane);
    __syncthreads();
    // ---- phase 4: x_out = (x_mid + h) + U + b_up
#pragma unroll
    for (int k = 0; k < AF_RPW; ++k) {
        const int lr = wid * AF_RPW + k, row = r0 + lr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (row < rend) {
                const float4 u = *reinterpret_cast<const float4*>(U + lr * (E + 4) + cc[i]);
                const float4 b = *reinterpret_cast<const float4*>(b_up + cc[i]);
                float4 o;
                o.x = s[k][i].x + (u.x + b.x); o.y = s[k][i].y + (u.y + b.y); o.z = s[k][i].z + (u.z + b.z); o.w = s[k][i].w + (u.w + b.w);
                *reinterpret_cast<float4*>(x_out + (size_t)row * E + cc[i]) = o;
            }
        }
    }
}

// ---- the token-contracted weight gradients in the shadow of the backward workgroups -------------------------------------------
// adapter_bwd_kernel keeps ONE workgroup on 200 of the 256 CUs for ~27 us (T = 6400; its LDS block admits no second one); the two
// G[e][j] = sum_r X[r][e] Y[r][j] products of an adapter (d W_up = dx_out^T act, d W_down = z^T d pre: tn_gemm64_kernel, 8.5 us
// per launch, 24 launches per step) are independent small jobs, so they ride in the SAME launch as extra workgroups, which the
// dispatcher places on the idle CUs:
//   d W_up   of THIS layer (its operands exist before the launch), and
//   d W_down of the layer processed BEFORE this one (its d pre was written by the previous launch; d pre alternates between two
//            buffers so that this launch's backward workgroups do not overwrite it; the last layer's product gets a launch of its own).
// A workgroup of the range takes two (256-row chunk, 64-column slab) units at a time, one per half of its eight waves.  The arithmetic and
// its ORDER are those of tn_gemm64_kernel (adapter.hip) -- MFMA k-steps 0..7 of the chunk in sequence into one accumulator, column
// sums per thread over the eight row batches in sequence, then the same shuffles -- hence the same bits; the chunk passes through
// LDS as two 128-row halves so that both units fit the block.
constexpr int TNH_ROWS = 128, TNH_LD = 72, TN_CHUNK = 256;
constexpr int TNH_BYTES = 2 * TNH_ROWS * TNH_LD * 2 + 4 * 64 * 4;
static_assert(AF_WAVES == 8, "two four-wave halves per workgroup of the contraction range");

struct AfTn {
    const bf16* X1; const bf16* Y1; float* P1; int n1;                     // this layer: X = dx_out (bf16), Y = saved activation
    const bf16* X2; const bf16* Y2; float* P2; float* csy2; int n2;        // previous launch's layer: X = z, Y = d pre (+ its column sums)
};

__device__ __forceinline__ bf16x8 af_trfrag(const bf16* tile, int ks, int col0, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int m = lane & 15, g = lane >> 4;
    const bf16* src = tile + (32 * ks + 4 * g + (m >> 2)) * TNH_LD + col0 + 4 * (m & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(src));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(src + 16 * TNH_LD));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
}

// The contraction range: workgroup blk of nblk walks the unit pairs blk, blk + nblk, ... (its two halves take one unit each), with
// sixteen 16-byte requests per thread in flight per unit.  The launcher's default is one workgroup per pair (a single trip); fewer,
// walking workgroups measured slower (see the launcher), and since round 5 they no longer request the next unit under the current one:
// those 64 registers are what keeps the launch at <= 128 VGPRs, i.e. two workgroups per CU.
// Every thread of the workgroup executes the same barriers (the trip count is the workgroup's; a half without a unit left works
// on a clamped one and stores nothing).
__device__ __forceinline__ void af_tn_range(char* smem, int blk, int nblk, const AfTn& tn, int T, int E) {
    const int half = threadIdx.x >> 8, t = threadIdx.x & 255;
    char* lds = smem + half * TNH_BYTES;
    bf16* Xs = reinterpret_cast<bf16*>(lds);
    bf16* Ys = Xs + TNH_ROWS * TNH_LD;
    float (*cs)[64] = reinterpret_cast<float (*)[64]>(lds + 2 * TNH_ROWS * TNH_LD * 2);
    const int lane = t & 63, w = t >> 6, g = lane >> 4, c16 = lane & 15, c = t & 7;
    const int eslabs = E / 64, nu = tn.n1 + tn.n2;
    const int trips = (nu - 2 * blk + 2 * nblk - 1) / (2 * nblk);
    bf16x8 xv[8], yv[8];
    // rows beyond T read row T - 1 (zeroed when used): no load sits inside a bounds branch
    auto request = [&](int u, bf16x8 (&x)[8], bf16x8 (&y)[8]) {
        const int uc = min(u, nu - 1);
        const bool first = uc < tn.n1;
        const int unit = first ? uc : uc - tn.n1;
        const bf16* X = first ? tn.X1 : tn.X2;
        const bf16* Y = first ? tn.Y1 : tn.Y2;
        const int chunk = unit / eslabs, e0 = (unit - chunk * eslabs) * 64, r0 = chunk * TN_CHUNK;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = min(r0 + (t >> 3) + 32 * it, T - 1);
            x[it] = load_bf16x8(X + (size_t)r * E + e0 + 8 * c);
            y[it] = load_bf16x8(Y + (size_t)r * 64 + 8 * c);
        }
    };
    for (int k = 0; k < trips; ++k) {
        const int u = 2 * (blk + k * nblk) + half;
        const bool valid = u < nu;
        const int uc = min(u, nu - 1);
        const bool first = uc < tn.n1;
        const int unit = first ? uc : uc - tn.n1;
        float* partial = first ? tn.P1 : tn.P2;
        float* csy = first ? nullptr : tn.csy2;
        const int chunk = unit / eslabs, e0 = (unit - chunk * eslabs) * 64, r0 = chunk * TN_CHUNK;
        request(u, xv, yv);
        if (k) __syncthreads();                              // the previous unit's fragments and column sums have been read
        float sy[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sy[i] = 0.f;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (r0 + (t >> 3) + 32 * it >= T) { xv[it] = zero_bf16x8(); yv[it] = zero_bf16x8(); }
#pragma unroll
            for (int i = 0; i < 8; ++i) sy[i] += bf2f(yv[it][i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float b = sy[i];
            b += __shfl_xor(b, 8, 64); b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if ((lane >> 3) == 0) cs[w][8 * c + i] = b;
        }
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if (ph) __syncthreads();                         // the first half's fragments have been read
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int y = (t >> 3) + 32 * it;
                *reinterpret_cast<bf16x8*>(Xs + y * TNH_LD + 8 * c) = xv[4 * ph + it];
                *reinterpret_cast<bf16x8*>(Ys + y * TNH_LD + 8 * c) = yv[4 * ph + it];
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 a = af_trfrag(Xs, ks, 16 * w, lane);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, af_trfrag(Ys, ks, 16 * nt, lane), acc[nt], 0, 0, 0);
            }
        }
        float csv = 0.f;
        if (t < 64) csv = cs[0][t] + cs[1][t] + cs[2][t] + cs[3][t];
        if (valid) {
            float* out = partial + (size_t)chunk * E * 64;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(e0 + 16 * w + 4 * g + r) * 64 + 16 * nt + c16] = acc[nt][r];
            if (csy && e0 == 0 && t < 64) csy[(size_t)chunk * 64 + t] = csv;
        }
    }
}


// backward.  dyb: bf16 copy of dx_out (the operand of the products), dres: dx_out itself (f32).  Outputs: dpre (bf16 [T][64], feeds
// the d W_down contraction), dh_bf16 = bf16(dres + LN_a'(d z)) (operand of the c_proj backward GEMM), partial[block][3][E] =
// column sums of dz*xhat (d gamma), dz (d beta), dres (d b_up) over the block's rows (layout of ln_bwd_affine_kernel).
// RES16 (round 5, bf16 gradient stream): dx_out exists in bf16 only -- the residual and the d b_up column sums take it from dyb.
template <int ACT, bool RES16, int NV>
__global__ __launch_bounds__(64 * AF_WAVES, (NV == 4 ? 2 : AF_MINW)) void adapter_bwd_kernel(const bf16* __restrict__ dyb, const float* __restrict__ dres,
                                                                    const bf16* __restrict__ wuT, const bf16* __restrict__ saved,
                                                                    const bf16* __restrict__ wdT, const float* __restrict__ hraw,
                                                                    const float* __restrict__ bpr, const float* __restrict__ mean_a,
                                                                    const float* __restrict__ rstd_a, const float* __restrict__ gamma,
                                                                    bf16* __restrict__ dpre, bf16* __restrict__ dh_bf16,
                                                                    float* __restrict__ partial, int T, int E_rt, int rb, int nb,
                                                                    AfTn tn) {
#ifdef AF_E_RUNTIME
    const int E = E_rt;
#else
    constexpr int E = 256 * NV;
    (void)E_rt;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= nb) {                                 // the contraction range (see af_tn_range)
        af_tn_range(smem, (int)blockIdx.x - nb, (int)gridDim.x - nb, tn, T, E);
        return;
    }
    const AfLds L = af_layout(E);
    bf16* Ys = reinterpret_cast<bf16*>(smem + L.zs);
    float* red = reinterpret_cast<float*>(smem + L.red);
    bf16* Ds = reinterpret_cast<bf16*>(smem + L.as);
    float* U = reinterpret_cast<float*>(smem + L.u);
    float* colred = reinterpret_cast<float*>(smem + L.u);       // aliases U after the row walk
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r0 = blockIdx.x * rb, rend = min(r0 + rb, T);    // this workgroup owns rows [r0, rend): rb <= 32 of the 32-row tile
    const int ldz = E + 8;
    // ---- phase 1: dyb rows -> LDS (16-byte pieces; rows beyond T read row T - 1, their results are never stored)
    {
        const int ppr = E / 8;
        for (int idx = tid; idx < AF_ROWS * ppr; idx += 64 * AF_WAVES) {
            const int lr = idx / ppr, c = idx - lr * ppr;
            const int row = min(r0 + lr, T - 1);
            *reinterpret_cast<bf16x8*>(Ys + lr * ldz + 8 * c) = load_bf16x8(dyb + (size_t)row * E + 8 * c);
        }
    }
    __syncthreads();
    // ---- phase 2: d pre = (dy Wu) * act'
    {
        float v[AF_CPT];
        constexpr int TPR = 64 / AF_CPT;
        const int lr = tid / TPR, c0 = (tid % TPR) * AF_CPT, row = r0 + lr;
        bf16 sv[AF_CPT], o[AF_CPT];
        af_load_mid(saved + (size_t)min(row, T - 1) * 64 + c0, sv);      // requested ahead of the product
        af_first_product(Ys, ldz, wuT, E, red, wid, lane, tid, v);
#pragma unroll
        for (int i = 0; i < AF_CPT; ++i) {
            const float h = bf2f(sv[i]);
            o[i] = f2bf(ACT == 0 ? (h > 0.f ? v[i] : 0.f) : v[i] * gelu_new_grad_f(h));
        }
        af_store_mid(Ds + lr * 72 + c0, dpre + (size_t)min(row, T - 1) * 64 + c0, row < rend, o);
    }
    __syncthreads();
    // ---- phase 3: d z = d pre Wd
    af_second_product(Ds, wdT, E, U, wid, lane);
    __syncthreads();
    // ---- phase 4: LayerNorm backward of the wave's rows; column sums
    int cc[NV];                                                 // E = 256 NV: every lane column is a live one
    float4 gv[NV], bp[NV], ag[NV], ab[NV], ar[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + i * 256;
        cc[i] = c;
        gv[i] = *reinterpret_cast<const float4*>(gamma + cc[i]);
        bp[i] = *reinterpret_cast<const float4*>(bpr + cc[i]);
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = ag[i]; ar[i] = ag[i];
    }
#pragma unroll
    for (int half = 0; half < AF_RPW / AF_RBB; ++half) {
        float4 xv[AF_RBB][NV], rv[AF_RBB][NV];
        float mu[AF_RBB], rs[AF_RBB];
#pragma unroll
        for (int k = 0; k < AF_RBB; ++k) {
            const int row = min(r0 + wid * AF_RPW + half * AF_RBB + k, T - 1);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                xv[k][i] = *reinterpret_cast<const float4*>(hraw + (size_t)row * E + cc[i]);
                if constexpr (RES16) {
                    const bf16x4 r16 = *reinterpret_cast<const bf16x4*>(dyb + (size_t)row * E + cc[i]);
                    rv[k][i] = make_float4(bf2f(r16[0]), bf2f(r16[1]), bf2f(r16[2]), bf2f(r16[3]));
                } else rv[k][i] = *reinterpret_cast<const float4*>(dres + (size_t)row * E + cc[i]);
            }
            mu[k] = mean_a[row]; rs[k] = rstd_a[row];
        }
#pragma unroll
        for (int k = 0; k < AF_RBB; ++k) {
            const int lr = wid * AF_RPW + half * AF_RBB + k, row = r0 + lr;
            const bool rok = row < rend;
            const float mean = mu[k], rstd = rs[k];
            float4 d[NV], xh[NV], gd[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                d[i] = *reinterpret_cast<const float4*>(U + lr * (E + 4) + cc[i]);
                const float4 x4 = make_float4(xv[k][i].x + bp[i].x, xv[k][i].y + bp[i].y, xv[k][i].z + bp[i].z, xv[k][i].w + bp[i].w);
                xh[i] = make_float4((x4.x - mean) * rstd, (x4.y - mean) * rstd, (x4.z - mean) * rstd, (x4.w - mean) * rstd);
                if (rok) {
                    ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
                    ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
                    ar[i].x += rv[k][i].x; ar[i].y += rv[k][i].y; ar[i].z += rv[k][i].z; ar[i].w += rv[k][i].w;
                }
                gd[i] = make_float4(d[i].x * gv[i].x, d[i].y * gv[i].y, d[i].z * gv[i].z, d[i].w * gv[i].w);
                s1 += gd[i].x + gd[i].y + gd[i].z + gd[i].w;
                s2 += gd[i].x * xh[i].x + gd[i].y * xh[i].y + gd[i].z * xh[i].z + gd[i].w * xh[i].w;
            }
            const float m1 = wave_sum(s1) / (float)E, m2 = wave_sum(s2) / (float)E;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (rok) {
                    bf16x4 o;
                    o[0] = f2bf(rstd * (gd[i].x - m1 - xh[i].x * m2) + rv[k][i].x); o[1] = f2bf(rstd * (gd[i].y - m1 - xh[i].y * m2) + rv[k][i].y);
                    o[2] = f2bf(rstd * (gd[i].z - m1 - xh[i].z * m2) + rv[k][i].z); o[3] = f2bf(rstd * (gd[i].w - m1 - xh[i].w * m2) + rv[k][i].w);
                    *reinterpret_cast<bf16x4*>(dh_bf16 + (size_t)row * E + cc[i]) = o;
                }
            }
        }
    }
    // column sums of the block: waves 1..3 through LDS (over U, which every wave has finished reading), wave 0 adds them in order
    __syncthreads();
    if (wid > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            {
                *reinterpret_cast<float4*>(colred + ((size_t)(wid - 1) * 3 + 0) * E + cc[i]) = ag[i];
                *reinterpret_cast<float4*>(colred + ((size_t)(wid - 1) * 3 + 1) * E + cc[i]) = ab[i];
                *reinterpret_cast<float4*>(colred + ((size_t)(wid - 1) * 3 + 2) * E + cc[i]) = ar[i];
            }
        }
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            {
                float4 g = ag[i], b = ab[i], r = ar[i];
                for (int w = 0; w < AF_WAVES - 1; ++w) {
                    const float4 g2 = *reinterpret_cast<const float4*>(colred + ((size_t)w * 3 + 0) * E + cc[i]);
                    const float4 b2 = *reinterpret_cast<const float4*>(colred + ((size_t)w * 3 + 1) * E + cc[i]);
                    const float4 r2 = *reinterpret_cast<const float4*>(colred + ((size_t)w * 3 + 2) * E + cc[i]);
                    g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
                    b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
                    r.x += r2.x; r.y += r2.y; r.z += r2.z; r.w += r2.w;
                }
                *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + 0) * E + cc[i]) = g;
                *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + 1) * E + cc[i]) = b;
                *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + 2) * E + cc[i]) = r;
            }
        }
    }
}

int af_check(int T, int E, const char* who) {
    if (T <= 0 || !pevit_adapter_fused_ok(E)) { pevit_set_error("%s: unsupported shape T=%d E=%d (E a multiple of 256, <= 1024)", who, T, E); return -1; }
    return 0;
}
template <typename K>
int af_attr(K kern, int bytes, const char* who) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        pevit_set_error("%s: cannot reserve %d bytes of LDS", who, bytes); return -1;
    }
    return 0;
}

}  // namespace

bool pevit_adapter_fused_ok(int E) { return E % 256 == 0 && E <= 256 * AF_MAXV && (E / AF_WAVES) % 16 == 0; }   // (K split: 32-row tiles E / 8 in 16-steps, 16-row tiles E / 2 in 32-steps)

// rows per workgroup.  (Measured and not kept: 25 valid rows per 32-row tile, which spreads T = 6400 over 256 workgroups instead of
// 200 -- 31.1 / 31.8 us against 29.2 / 30.4: the kernels are bound by their chain of phases, not by bytes per CU.)
static int af_rows_per_wg(int T) { (void)T; return AF_ROWS; }
int pevit_adapter_blocks(int T) { return ceil_div(T, af_rows_per_wg(T)); }

int pevit_launch_adapter_fwd(int act_kind, const float* hraw, const float* bpr, const float* x_mid, const float* gamma, const float* beta,
                             const bf16* wd, const float* b_down, const bf16* wu, const float* b_up, bf16* z, float* mean_a,
                             float* rstd_a, bf16* act, bf16* apre, float* x_out, int T, int E, hipStream_t s) {
    if (af_check(T, E, "adapter_fwd")) return -1;
    const AfLds L = af_layout(E);
    const int lds = L.colred;
    const int rb = af_rows_per_wg(T);
    const dim3 grid(ceil_div(T, rb)), block(64 * AF_WAVES);
    static bool attr[2][AF_MAXV] = {};
    auto go = [&](auto kern, int slot, int nv) -> int {
        if (!attr[slot][nv - 1]) { if (af_attr(kern, 160 * 1024, "adapter_fwd")) return -1; attr[slot][nv - 1] = true; }
        hipLaunchKernelGGL(kern, grid, block, lds, s, hraw, bpr, x_mid, gamma, beta, wd, b_down, wu, b_up, z, mean_a, rstd_a, act, apre, x_out, T, E, rb);
        return 0;
    };
    int rc = -1;
    switch (4 * (act_kind != 0) + E / 256 - 1) {
        case 0: rc = go(adapter_fwd_kernel<0, 1>, 0, 1); break;
        case 1: rc = go(adapter_fwd_kernel<0, 2>, 0, 2); break;
        case 2: rc = go(adapter_fwd_kernel<0, 3>, 0, 3); break;
        case 3: rc = go(adapter_fwd_kernel<0, 4>, 0, 4); break;
        case 4: rc = go(adapter_fwd_kernel<1, 1>, 1, 1); break;
        case 5: rc = go(adapter_fwd_kernel<1, 2>, 1, 2); break;
        case 6: rc = go(adapter_fwd_kernel<1, 3>, 1, 3); break;
        case 7: rc = go(adapter_fwd_kernel<1, 4>, 1, 4); break;
    }
    if (rc) return rc;
    LAUNCH_OK("adapter_fwd_kernel");
    return 0;
}

// tn_*: the contraction range of the launch (see af_tn_range); tn_x1 = nullptr / tn_x2 = nullptr leave a product out.
//   product 1: partial1[chunk][E][64] = sum over the chunk's rows of x1[r][e] y1[r][j]             (x1: [T][E] bf16, y1: [T][64] bf16)
//   product 2: the same for x2, y2 -> partial2, and csy2[chunk][64] = column sums of y2
int pevit_launch_adapter_bwd(int act_kind, const bf16* dyb, const float* dres, const bf16* wuT, const bf16* saved, const bf16* wdT,
                             const float* hraw, const float* bpr, const float* mean_a, const float* rstd_a, const float* gamma,
                             bf16* dpre, bf16* dh_bf16, float* partial, int T, int E, hipStream_t s, const bf16* tn_x1,
                             const bf16* tn_y1, float* tn_partial1, const bf16* tn_x2, const bf16* tn_y2, float* tn_partial2,
                             float* tn_csy2, int tn_blocks) {
    if (af_check(T, E, "adapter_bwd")) return -1;
    if (tn_x2 && tn_y2 == dpre) { pevit_set_error("adapter_bwd: the deferred contraction reads the d pre buffer this launch writes"); return -1; }
    const AfLds L = af_layout(E);
    const int rb = af_rows_per_wg(T);
    const int nb = ceil_div(T, rb), units = ceil_div(T, TN_CHUNK) * (E / 64);
    AfTn tn;
    tn.X1 = tn_x1; tn.Y1 = tn_y1; tn.P1 = tn_partial1; tn.n1 = tn_x1 ? units : 0;
    tn.X2 = tn_x2; tn.Y2 = tn_y2; tn.P2 = tn_partial2; tn.csy2 = tn_csy2; tn.n2 = tn_x2 ? units : 0;
    // workgroups of the contraction range: one per unit pair.  (Measured, Adapter ViT-B/32 batch 128, same box: no folding 4.92 ms per
    // step; 40 / 80 persistent workgroups walking the pairs 4.96; 112: 4.78; one per pair (300): 4.72 -- a unit is a latency chain
    // (requests -> LDS -> matrix core -> stores), so more of them in flight beats fewer, longer-lived ones.)
    const int ntn = min(ceil_div(tn.n1 + tn.n2, 2), tn_blocks > 0 ? tn_blocks : (1 << 30));
    const int lds = ntn ? max(L.total_bwd, 2 * TNH_BYTES) : L.total_bwd;
    const dim3 grid(nb + ntn), block(64 * AF_WAVES);
    static bool attr[4][AF_MAXV] = {};
    // dres == nullptr: the bf16 gradient stream (round 5) -- dx_out is read from dyb wherever the f32 copy was
    auto go = [&](auto kern, int slot, int nv) -> int {
        if (!attr[slot][nv - 1]) { if (af_attr(kern, 160 * 1024, "adapter_bwd")) return -1; attr[slot][nv - 1] = true; }
        hipLaunchKernelGGL(kern, grid, block, lds, s, dyb, dres, wuT, saved, wdT, hraw, bpr, mean_a, rstd_a, gamma, dpre, dh_bf16, partial, T, E, rb, nb, tn);
        return 0;
    };
    int rc = -1;
    const int slot = 2 * (act_kind != 0) + (dres ? 0 : 1);
#define AF_BWD_CASE(A, R, N) case 4 * (2 * A + R) + N - 1: rc = go(adapter_bwd_kernel<A, R != 0, N>, 2 * A + R, N); break;
    switch (4 * slot + E / 256 - 1) {
        AF_BWD_CASE(0, 0, 1) AF_BWD_CASE(0, 0, 2) AF_BWD_CASE(0, 0, 3) AF_BWD_CASE(0, 0, 4)
        AF_BWD_CASE(0, 1, 1) AF_BWD_CASE(0, 1, 2) AF_BWD_CASE(0, 1, 3) AF_BWD_CASE(0, 1, 4)
        AF_BWD_CASE(1, 0, 1) AF_BWD_CASE(1, 0, 2) AF_BWD_CASE(1, 0, 3) AF_BWD_CASE(1, 0, 4)
        AF_BWD_CASE(1, 1, 1) AF_BWD_CASE(1, 1, 2) AF_BWD_CASE(1, 1, 3) AF_BWD_CASE(1, 1, 4)
    }
#undef AF_BWD_CASE
    if (rc) return rc;
    LAUNCH_OK("adapter_bwd_kernel");
    return 0;
}
