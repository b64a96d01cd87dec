// Post-MLP bottleneck adapters: Adapter (adapter_model.py:204-336) and Compacter
// (compacter_model.py:196-308,356-503).
//
//   x <- x + [ h + up(act(down(LN_a(h)))) ]          h = mlp(ln_2(x))
//
// (the reference's Adapter evaluates mlp(ln_2(x)) twice -- adapter_model.py:333 -- with identical
// values; here h is computed once and its gradient is the sum of both uses).
// Adapter : down = Linear(E,64), act = ReLU,      up = Linear(64,E)
// Compacter: down/up = PHMLinear, H = sum_{i<4} kron(rule_i (4x4), Wl_i Wr_i), act = gelu_new; the
//            shared rule (4,4,4) is frozen (its name lacks "compacter": compacter_clip.py:122).
// Both reduce to two dense E x 64 panels, so they share every kernel: the contractions are the MFMA
// GEMM (gemm.hip, epilogues BIAS_RELU / BIAS_GELUNEW / DRELU / DGELUNEW), and this file holds
//   * the panel builders (f32 master parameters -> bf16 panels in both orientations, every step),
//   * the token-contracted weight gradients  G = X^T Y  (LDS-transposed panels + bf16 MFMA),
//   * LayerNorm backward with gradients for the trainable affine (adapter_norm_before),
//   * the chain rule from the dense panels back to the reference's parameter tensors.
#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

constexpr int TG_ROWS = 256;
constexpr int TG_LD = 72;           // row stride (elements) of the row-major LDS tiles
// MFMA fragment of the TRANSPOSE of a row-major LDS tile [token][TG_LD] by ds_read_b64_tr_b16 (see lowrank.hip lg_trfrag):
// lane (m, g) gets column col0 + m for the tokens 32ks + 4g + 0..3 and 32ks + 16 + 4g + 0..3
__device__ __forceinline__ bf16x8 tg_trfrag(const bf16* tile, int ks, int col0, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int m = lane & 15, g = lane >> 4;
    const bf16* src = tile + (32 * ks + 4 * g + (m >> 2)) * TG_LD + col0 + 4 * (m & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(src));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(src + 16 * TG_LD));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
}

// ---------------------------------------------------------------------------------------------
// panels: wd [64][E] (B operand of the down GEMM), wdT [E][64] (d z GEMM), wu [E][64] (up GEMM),
// wuT [64][E] (d act GEMM).  blockIdx.y = layer.
template <typename ST>
__global__ void prep_adapter_kernel(const float* __restrict__ w_down, const float* __restrict__ w_up, BottleneckPanels pan,
                                    int E, LayerStrides st) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 64 * E) return;
    const size_t lo = (size_t)blockIdx.y;
    w_down += lo * st.param_floats; w_up += lo * st.param_floats;
    char* base = reinterpret_cast<char*>(pan.wd) + lo * st.arena_bytes;
    bf16* wd = reinterpret_cast<bf16*>(base);
    bf16* wdT = reinterpret_cast<bf16*>(base + ((char*)pan.wdT - (char*)pan.wd));
    bf16* wu = reinterpret_cast<bf16*>(base + ((char*)pan.wu - (char*)pan.wd));
    bf16* wuT = reinterpret_cast<bf16*>(base + ((char*)pan.wuT - (char*)pan.wd));
    const int j = idx / E, e = idx - j * E;
    const float d = w_down[(size_t)j * E + e];        // (64, E)
    const float u = w_up[(size_t)e * 64 + j];         // (E, 64)
    st_store<ST>(wd, (size_t)j * E + e, d);
    st_store<ST>(wdT, (size_t)e * 64 + j, d);
    st_store<ST>(wu, (size_t)e * 64 + j, u);
    st_store<ST>(wuT, (size_t)j * E + e, u);
}

// Compacter: H_down[a*Fi+k][c*16+p] = sum_i rule[i][a][c] Wl[i][k] Wr[i][p]   (Fi = E/4)
//            H_up  [a*16+k][c*Fi+p] = sum_i rule[i][a][c] Ul[i][k] Ur[i][p]
// effective dense weights: w_down[j][e] = H_down[e][j], w_up[e][j] = H_up[j][e].
template <typename ST>
__global__ void prep_compacter_kernel(const float* __restrict__ rule, const float* __restrict__ dWl,
                                      const float* __restrict__ dWr, const float* __restrict__ uWl,
                                      const float* __restrict__ uWr, BottleneckPanels pan, int E, LayerStrides st) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 64 * E) return;
    const size_t lo = (size_t)blockIdx.y;
    dWl += lo * st.param_floats; dWr += lo * st.param_floats; uWl += lo * st.param_floats; uWr += lo * st.param_floats;
    char* base = reinterpret_cast<char*>(pan.wd) + lo * st.arena_bytes;
    bf16* wd = reinterpret_cast<bf16*>(base);
    bf16* wdT = reinterpret_cast<bf16*>(base + ((char*)pan.wdT - (char*)pan.wd));
    bf16* wu = reinterpret_cast<bf16*>(base + ((char*)pan.wu - (char*)pan.wd));
    bf16* wuT = reinterpret_cast<bf16*>(base + ((char*)pan.wuT - (char*)pan.wd));
    const int Fi = E / 4;
    const int j = idx / E, e = idx - j * E;
    {   // down: e = a*Fi + k, j = c*16 + p
        const int a = e / Fi, k = e - a * Fi, c = j >> 4, p = j & 15;
        float h = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) h += rule[i * 16 + a * 4 + c] * dWl[i * Fi + k] * dWr[i * 16 + p];
        st_store<ST>(wd, (size_t)j * E + e, h);
        st_store<ST>(wdT, (size_t)e * 64 + j, h);
    }
    {   // up: j = a*16 + k, e = c*Fi + p
        const int a = j >> 4, k = j & 15, c = e / Fi, p = e - c * Fi;
        float h = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) h += rule[i * 16 + a * 4 + c] * uWl[i * 16 + k] * uWr[i * Fi + p];
        st_store<ST>(wu, (size_t)e * 64 + j, h);
        st_store<ST>(wuT, (size_t)j * E + e, h);
    }
}

// ---------------------------------------------------------------------------------------------
// G[e][j] = sum_r X[r][e] Y[r][j]   X: bf16 [T][ldx] (64-column slab per workgroup), Y: bf16 [T][ldy]
// (64 columns).  Workgroup = (chunk of TG_ROWS rows, 64 columns e).  Same structure as
// lowrank_grad_kernel: coalesced 16-byte loads, ROW-major LDS images (16-byte writes), both MFMA operands by the
// transposing LDS read.
// partial[chunk][E][64]; optional column sums of X -> csx[chunk][E], of Y -> csy[chunk][64].
__global__ __launch_bounds__(256) void tn_gemm64_kernel(const bf16* __restrict__ X, int ldx, const bf16* __restrict__ Y,
                                                        int ldy, float* __restrict__ partial, float* __restrict__ csx,
                                                        float* __restrict__ csy, int T, int E) {
    __shared__ __attribute__((aligned(16))) bf16 Xs[TG_ROWS * TG_LD];
    __shared__ __attribute__((aligned(16))) bf16 Ys[TG_ROWS * TG_LD];
    __shared__ float cs[2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int eslabs = E / 64;
    const int chunk = blockIdx.x / eslabs, e0 = (blockIdx.x - chunk * eslabs) * 64;
    const int r0 = chunk * TG_ROWS;
    const int c = tid & 7;
    bf16x8 xv[TG_ROWS / 32], yv[TG_ROWS / 32];
#pragma unroll
    for (int it = 0; it < TG_ROWS / 32; ++it) {
        const int r = r0 + (tid >> 3) + 32 * it;
        xv[it] = zero_bf16x8(); yv[it] = zero_bf16x8();
        if (r < T) {
            xv[it] = load_bf16x8(X + (size_t)r * ldx + e0 + 8 * c);
            yv[it] = load_bf16x8(Y + (size_t)r * ldy + 8 * c);
        }
    }
    float sx[8], sy[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sx[i] = 0.f; sy[i] = 0.f; }
#pragma unroll
    for (int it = 0; it < TG_ROWS / 32; ++it) {
        const int y = (tid >> 3) + 32 * it;
        *reinterpret_cast<bf16x8*>(Xs + y * TG_LD + 8 * c) = xv[it];
        *reinterpret_cast<bf16x8*>(Ys + y * TG_LD + 8 * c) = yv[it];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sx[i] += bf2f(xv[it][i]); sy[i] += bf2f(yv[it][i]); }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = sx[i], b = sy[i];
        a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 8, 64); b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
        if ((lane >> 3) == 0) { cs[0][wid][8 * c + i] = a; cs[1][wid][8 * c + i] = b; }
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < TG_ROWS / 32; ++ks) {
        const bf16x8 a = tg_trfrag(Xs, ks, 16 * wid, lane);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, tg_trfrag(Ys, ks, 16 * nt, lane), acc[nt], 0, 0, 0);
    }
    float* out = partial + (size_t)chunk * E * 64;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(e0 + 16 * wid + 4 * g + r) * 64 + 16 * nt + c16] = acc[nt][r];
    if (tid < 64) {
        if (csx) csx[(size_t)chunk * E + e0 + tid] = cs[0][0][tid] + cs[0][1][tid] + cs[0][2][tid] + cs[0][3][tid];
        if (csy && e0 == 0) csy[(size_t)chunk * 64 + tid] = cs[1][0][tid] + cs[1][1][tid] + cs[1][2][tid] + cs[1][3][tid];
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward with trainable affine: dx (+ dres) as ln_bwd, plus per-block partial sums of
// d gamma = sum_r dy*xhat, d beta = sum_r dy, and the f32 column sums of dres (the gradient of the
// adapter's up-projection bias: summing its bf16 copy instead loses the cancellation-heavy sum).
// Block = 32 rows = 8 waves x 4 rows (1600 waves at T = 6400: the 64-row / 4-wave form left 156 CUs idle and
// ran 4x slower than the plain LayerNorm backward); partial[block][3][E].
constexpr int LNA_ROWS = 32;
constexpr int LNA_WAVES = 8;
constexpr int LNA_MAXV = 4;
template <typename ST>
__global__ __launch_bounds__(64 * LNA_WAVES) void ln_bwd_affine_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            const float* __restrict__ gamma, const float* dres, float* dx,
                                                            bf16* __restrict__ dx_bf16, float* __restrict__ partial, int rows,
                                                            int E) {
    extern __shared__ float red[];     // [LNA_WAVES - 1][3][E]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float4 ag[LNA_MAXV], ab[LNA_MAXV], ar[LNA_MAXV];      // sums of dy*xhat, dy, dres
#pragma unroll
    for (int i = 0; i < LNA_MAXV; ++i) {
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = make_float4(0.f, 0.f, 0.f, 0.f); ar[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // A wave walks its LNA_ROWS / LNA_WAVES rows with the NEXT row's operands already requested (processed one after the other
    // each row paid a full HBM round trip before its reductions could start: 27.8 us for 88 MB in step).  The walk is unrolled and
    // no load sits inside a bounds branch: rows beyond `rows` / columns beyond E read a clamped address and are masked, gamma is
    // loaded once, the next row's mean / rstd travel with its operands, and the next row is IN before this row's (conditional)
    // stores go out -- otherwise hipcc waits with vmcnt(0) at the next use, which on gfx950 also waits for those stores.
    constexpr int RPW = LNA_ROWS / LNA_WAVES;
    const float* rsrc = dres ? dres : x;
    const bool has_res = dres != nullptr;
    bool cok[LNA_MAXV];
    int cc[LNA_MAXV];
    float4 gv[LNA_MAXV];
#pragma unroll
    for (int i = 0; i < LNA_MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        cok[i] = c < E; cc[i] = cok[i] ? c : 0;
        gv[i] = *reinterpret_cast<const float4*>(gamma + cc[i]);
    }
    float4 nd[LNA_MAXV], nx[LNA_MAXV], nr[LNA_MAXV];
    float nmean, nrstd;
    auto request = [&](int row) {                 // row already clamped into [0, rows)
        const size_t base = (size_t)row * E;
#pragma unroll
        for (int i = 0; i < LNA_MAXV; ++i) {
            nd[i] = *reinterpret_cast<const float4*>(dy + base + cc[i]);
            nx[i] = *reinterpret_cast<const float4*>(x + base + cc[i]);
            nr[i] = *reinterpret_cast<const float4*>(rsrc + base + cc[i]);
        }
        nmean = mean_in[row]; nrstd = rstd_in[row];
    };
    const int row_first = blockIdx.x * LNA_ROWS + wid;
    request(min(row_first, rows - 1));
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int row = row_first + k * LNA_WAVES;
        const bool rok = row < rows;
        const float mean = nmean, rstd = nrstd;
        const size_t base = (size_t)row * E;
        float4 cd[LNA_MAXV], cx[LNA_MAXV], cr[LNA_MAXV];
#pragma unroll
        for (int i = 0; i < LNA_MAXV; ++i) { cd[i] = nd[i]; cx[i] = nx[i]; cr[i] = has_res ? nr[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
        if (k + 1 < RPW) request(min(row + LNA_WAVES, rows - 1));
        float4 gd[LNA_MAXV], xh[LNA_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LNA_MAXV; ++i) {
            const bool on = rok && cok[i];
            const float4 d = cd[i];
            const float4 xv = cx[i];
            const float4 g = gv[i];
            xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            if (on) {
                ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
                ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
                ar[i].x += cr[i].x; ar[i].y += cr[i].y; ar[i].z += cr[i].z; ar[i].w += cr[i].w;
            }
            gd[i] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
            s1 += cok[i] ? gd[i].x + gd[i].y + gd[i].z + gd[i].w : 0.f;
            s2 += cok[i] ? gd[i].x * xh[i].x + gd[i].y * xh[i].y + gd[i].z * xh[i].z + gd[i].w * xh[i].w : 0.f;
        }
        const float m1 = wave_sum(s1) / (float)E, m2 = wave_sum(s2) / (float)E;
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), visible to the compiler: the next row is in, nothing is pending behind the stores
#pragma unroll
        for (int i = 0; i < LNA_MAXV; ++i) {
            if (rok && cok[i]) {
                const int c = lane * 4 + i * 256;
                float4 o;
                o.x = rstd * (gd[i].x - m1 - xh[i].x * m2); o.y = rstd * (gd[i].y - m1 - xh[i].y * m2);
                o.z = rstd * (gd[i].z - m1 - xh[i].z * m2); o.w = rstd * (gd[i].w - m1 - xh[i].w * m2);
                o.x += cr[i].x; o.y += cr[i].y; o.z += cr[i].z; o.w += cr[i].w;
                if (dx) *reinterpret_cast<float4*>(dx + base + c) = o;       // (the engine consumes only the stored-type copy)
                if (dx_bf16) st_store4<ST>(dx_bf16, base + c, o.x, o.y, o.z, o.w);
            }
        }
    }
    // combine the waves' column sums (fixed order), write this block's partial [3][E]
    if (wid > 0) {
#pragma unroll
        for (int i = 0; i < LNA_MAXV; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < E) {
                *reinterpret_cast<float4*>(red + ((size_t)(wid - 1) * 3 + 0) * E + c) = ag[i];
                *reinterpret_cast<float4*>(red + ((size_t)(wid - 1) * 3 + 1) * E + c) = ab[i];
                *reinterpret_cast<float4*>(red + ((size_t)(wid - 1) * 3 + 2) * E + c) = ar[i];
            }
        }
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int i = 0; i < LNA_MAXV; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < E) {
                float4 g = ag[i], b = ab[i], r = ar[i];
                for (int w = 0; w < LNA_WAVES - 1; ++w) {
                    const float4 g2 = *reinterpret_cast<const float4*>(red + ((size_t)w * 3 + 0) * E + c);
                    const float4 b2 = *reinterpret_cast<const float4*>(red + ((size_t)w * 3 + 1) * E + c);
                    const float4 r2 = *reinterpret_cast<const float4*>(red + ((size_t)w * 3 + 2) * E + c);
                    g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
                    b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
                    r.x += r2.x; r.y += r2.y; r.z += r2.z; r.w += r2.w;
                }
                *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + 0) * E + c) = g;
                *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + 1) * E + c) = b;
                *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + 2) * E + c) = r;
            }
        }
    }
}

// out[l*out_layer + i] += sum_c partial[l*partial_layer + c*n + i].  Block = 64 columns x 4 chunk groups
// (group g sums chunks c = g, g+4, ... in order; the 4 group sums are combined in a fixed order): deterministic.
constexpr int CR_COLS = 64, CR_GROUPS = 4;
__device__ __forceinline__ float colsum_groups(const float* __restrict__ p, int chunks, size_t stride, int col, int g,
                                               float (*red)[CR_COLS]) {
    float s0 = 0.f, s1 = 0.f;
    int c = g;
    for (; c + CR_GROUPS < chunks; c += 2 * CR_GROUPS) { s0 += p[(size_t)c * stride + col]; s1 += p[(size_t)(c + CR_GROUPS) * stride + col]; }
    if (c < chunks) s0 += p[(size_t)c * stride + col];
    red[g][threadIdx.x & (CR_COLS - 1)] = s0 + s1;
    __syncthreads();
    return (red[0][threadIdx.x & (CR_COLS - 1)] + red[1][threadIdx.x & (CR_COLS - 1)]) +
           (red[2][threadIdx.x & (CR_COLS - 1)] + red[3][threadIdx.x & (CR_COLS - 1)]);
}

__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ partial, int chunks, int n, float* out,
                                                            size_t partial_layer, size_t out_layer) {
    __shared__ float red[CR_GROUPS][CR_COLS];
    const int i = blockIdx.x * CR_COLS + (threadIdx.x & (CR_COLS - 1)), g = threadIdx.x >> 6;
    const int col = i < n ? i : n - 1;
    const float s = colsum_groups(partial + (size_t)blockIdx.y * partial_layer, chunks, (size_t)n, col, g, red);
    if (g == 0 && i < n) out[(size_t)blockIdx.y * out_layer + i] += s;
}

// partial[l][block][3][n] -> out0/out1/out2[l*out_layer + i] += sum over blocks
__global__ __launch_bounds__(256) void colsum_reduce3_kernel(const float* __restrict__ partial, int chunks, int n, float* o0, float* o1,
                                                             float* o2, size_t partial_layer, size_t out_layer) {
    __shared__ float red[CR_GROUPS][CR_COLS];
    const int i = blockIdx.x * CR_COLS + (threadIdx.x & (CR_COLS - 1)), g = threadIdx.x >> 6;
    const int col = i < 3 * n ? i : 3 * n - 1;
    const float s = colsum_groups(partial + (size_t)blockIdx.y * partial_layer, chunks, (size_t)3 * n, col, g, red);
    if (g == 0 && i < 3 * n) {
        const int which = i / n, e = i - which * n;
        float* o = which == 0 ? o0 : (which == 1 ? o1 : o2);
        o[(size_t)blockIdx.y * out_layer + e] += s;
    }
}

// Adapter chain: g_down[j][e] += G_down[e][j] ; g_up[e][j] += G_up[e][j]   (G already chunk-reduced)
__global__ void chain_adapter_kernel(const float* __restrict__ Gd, const float* __restrict__ Gu, float* g_down, float* g_up,
                                     int E, size_t g_layer, size_t param_layer) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 64 * E) return;
    const size_t l = blockIdx.y;
    const int e = idx >> 6, j = idx & 63;
    g_down[l * param_layer + (size_t)j * E + e] += Gd[l * g_layer + idx];
    g_up[l * param_layer + idx] += Gu[l * g_layer + idx];
}

// Compacter chain: dH -> d W_left / d W_right of both PHM layers (compacter_model.py:302-308 differentiated).
//   dH_down[e][j] = Gd[e][j]  (e = a*Fi+k, j = c*16+p) ; dH_up[j][e] = Gu[e][j] (j = a*16+k, e = c*Fi+p)
//   0: d dWl[i][k] = sum_{a,c,p} Gd[a*Fi+k][c*16+p] rule[i][a][c] dWr[i][p]      3: d uWr[i][p] = sum_{a,c,k} Gu[c*Fi+p][a*16+k] rule[i][a][c] uWl[i][k]
//   1: d dWr[i][o] = sum_{a,c,k} Gd[a*Fi+k][c*16+o] rule[i][a][c] dWl[i][k]      2: d uWl[i][o] = sum_{a,c,p} Gu[c*Fi+p][a*16+o] rule[i][a][c] uWr[i][p]
// One 16-wave block per (tensor, layer), all four PHM slices i at once.  A wave reads whole 64-float rows of G (lane = column:
// 256 coalesced bytes per request, every row fetched once per block) -- tensors 0 / 3 contract a row against a per-lane
// coefficient and reduce it over the wave; tensors 1 / 2 accumulate per lane over the rows and meet through LDS.  Every output is
// a fixed-order sum: deterministic.  (Round 4; the one-thread-per-output form walked 256 strided loads in sequence: 43 us.)
constexpr int CC_WAVES = 16;
__global__ __launch_bounds__(64 * CC_WAVES) void chain_compacter_kernel(const float* __restrict__ Gd, const float* __restrict__ Gu,
                                                              const float* __restrict__ rule, const float* __restrict__ params,
                                                              float* grads, int E, size_t g_layer, size_t param_layer,
                                                              size_t off_dWl, size_t off_dWr, size_t off_uWl, size_t off_uWr) {
    __shared__ float part[CC_WAVES][4][16];
    const size_t l = blockIdx.y;
    const int which = blockIdx.x;
    const int Fi = E / 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 4, lo = lane & 15;
    const float* P = params + l * param_layer; float* Gp = grads + l * param_layer;
    const bool down = which < 2;
    const float* G = (down ? Gd : Gu) + l * g_layer;
    if (which == 0 || which == 3) {
        // long output index q (k resp. p), short contracted index lo (p resp. k), rule index pair (x, hi): rows x*Fi + q, x = 0..3
        const float* vshort = P + (which == 0 ? off_dWr : off_uWl);            // [i][16]
        float m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float r = (which == 0) ? rule[i * 16 + x * 4 + hi] : rule[i * 16 + hi * 4 + x];      // rule[i][a][c]
                m[i][x] = r * vshort[i * 16 + lo];
            }
        float* out = Gp + (which == 0 ? off_dWl : off_uWr);
        for (int q0 = w; q0 < Fi; q0 += 2 * CC_WAVES) {
            float xr[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = min(q0 + u * CC_WAVES, Fi - 1);
#pragma unroll
                for (int x = 0; x < 4; ++x) xr[u][x] = G[(size_t)(x * Fi + q) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = q0 + u * CC_WAVES;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = xr[u][0] * m[i][0] + xr[u][1] * m[i][1] + xr[u][2] * m[i][2] + xr[u][3] * m[i][3];
                    v = wave_sum(v);
                    if (lane == 0 && q < Fi) out[(size_t)i * Fi + q] += v;
                }
            }
        }
    } else {
        // 16 outputs lo per slice; lane column = (hi, lo); rows x*Fi + q contracted with rule[i][.][.] (pair (x, hi)) and vlong[i][q]
        const float* vlong = P + (which == 1 ? off_dWl : off_uWr);             // [i][Fi]
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int x = 0; x < 4; ++x) {
            float r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = (which == 1) ? rule[i * 16 + x * 4 + hi] : rule[i * 16 + hi * 4 + x];
            for (int q0 = w; q0 < Fi; q0 += 4 * CC_WAVES) {
                float xr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xr[u] = G[(size_t)(x * Fi + min(q0 + u * CC_WAVES, Fi - 1)) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + u * CC_WAVES;
                    if (q < Fi) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] += xr[u] * r[i] * vlong[(size_t)i * Fi + q];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = acc[i];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane < 16) part[w][i][lane] = v;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int i = threadIdx.x >> 4, o = threadIdx.x & 15;
            float t = 0.f;
            for (int ww = 0; ww < CC_WAVES; ++ww) t += part[ww][i][o];
            Gp[(which == 1 ? off_dWr : off_uWl) + (size_t)i * 16 + o] += t;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Bottleneck pair in ONE launch:  S = fn(X W1^T [+ b1])  (T x 64, K = E),  out = S W2^T [+ b2 + resid]  (T x E, K = 64).
//   forward  (adapter_model.py:270-272 / compacter_model.py:437-443): X = z = LN_a(h), W1 = down, fn = ReLU | gelu_new,
//            S = act (saved; Compacter also saves the bf16 pre-activation), W2 = up, resid = x_mid + h  -> block output (f32)
//   backward: X = d out (bf16), W1 = up^T, fn = multiply by act'(saved), S = d pre (saved for the weight gradients),
//            W2 = down^T -> d z (f32)
// As two GEMM launches these were N = 64 and K = 64 products on 64x64 tiles (100 workgroups, ~1 % MFMA utilisation,
// 12.6 + 10.3 us forward, 10.9 + 8.6 us backward per layer in step); here a workgroup owns 32 token rows: phase 1 on four
// waves = 2 column fragments x 2 halves of K (fragment-shaped global loads: X is read once, W1 comes from L2), the halves
// meet through LDS, S goes to LDS (and HBM); phase 2: every wave multiplies S by six 32-row panels of W2.
// Same rounding points as the two-launch form (S rounded to bf16 once, f32 accumulation).
enum { BN_FWD_RELU = 0, BN_FWD_GELUNEW = 1, BN_BWD_RELU = 2, BN_BWD_GELUNEW = 3 };
constexpr int BNK_ROWS = 32, BNK_LDS = 72;      // token rows per workgroup; row stride (elements) of S in LDS

template <int MODE>
__global__ __launch_bounds__(256) void bottleneck_pair_kernel(const bf16* __restrict__ X, int ldx, const bf16* __restrict__ W1,
                                                              const float* __restrict__ b1, const bf16* __restrict__ aux,
                                                              bf16* __restrict__ S_out, bf16* __restrict__ S_pre,
                                                              const bf16* __restrict__ W2, const float* __restrict__ b2,
                                                              const float* __restrict__ resid, float* __restrict__ out, int T, int E) {
    __shared__ __attribute__((aligned(16))) float red[2][16][64];
    __shared__ __attribute__((aligned(16))) bf16 Ss[BNK_ROWS * BNK_LDS];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r0 = blockIdx.x * BNK_ROWS;
    const int frow = lane & 31, fhalf = lane >> 5;
    // ---- phase 1: wave = (column fragment nf, K half kh)
    {
        const int nf = wid & 1, kh = wid >> 1;
        const int row = min(r0 + frow, T - 1);
        const bf16* xa = X + (size_t)row * ldx + kh * (E / 2) + 8 * fhalf;
        const bf16* wb = W1 + (size_t)(nf * 32 + frow) * E + kh * (E / 2) + 8 * fhalf;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int steps = E / 32;                       // 16-deep k-steps per half
#pragma unroll 8
        for (int ks = 0; ks < steps; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_bf16x8(xa + 16 * ks), load_bf16x8(wb + 16 * ks), acc, 0, 0, 0);
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[nf][r][lane] = acc[r];
        }
        __syncthreads();
        if (kh == 0) {
            const int col = nf * 32 + frow;
            const float bias = (MODE == BN_FWD_RELU || MODE == BN_FWD_GELUNEW) ? b1[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * fhalf, grow = r0 + lr;
                float v = acc[r] + red[nf][r][lane] + bias;
                bf16 sv;
                if constexpr (MODE == BN_FWD_RELU) {
                    sv = f2bf(fmaxf(v, 0.f));
                } else if constexpr (MODE == BN_FWD_GELUNEW) {
                    const bf16 pre = f2bf(v);
                    if (grow < T) S_pre[(size_t)grow * 64 + col] = pre;
                    sv = f2bf(gelu_new_f(bf2f(pre)));
                } else {
                    const float a = grow < T ? bf2f(aux[(size_t)grow * 64 + col]) : 0.f;
                    if constexpr (MODE == BN_BWD_RELU) sv = f2bf(a > 0.f ? v : 0.f);
                    else sv = f2bf(v * gelu_new_grad_f(a));
                }
                Ss[lr * BNK_LDS + col] = sv;
                if (grow < T) S_out[(size_t)grow * 64 + col] = sv;
            }
        }
        __syncthreads();
    }
    // ---- phase 2: out[32 x E] = S[32 x 64] W2^T, wave w takes the 32-column panels w, w + 4, ... in batches of FB: all
    // loads of a batch (W2 fragments, residual values) are issued before its first store -- a load after a store makes hipcc
    // wait with vmcnt(0), i.e. for the store (the first version of this kernel ran 54 us that way)
    constexpr int FB = 4;
    bf16x8 sa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa[ks] = *reinterpret_cast<const bf16x8*>(Ss + frow * BNK_LDS + 16 * ks + 8 * fhalf);
    const int nfr = E / 32;
    for (int f0 = wid; f0 < nfr; f0 += 4 * FB) {
        f32x16 acc[FB];
        float rv[FB][16], bias[FB];
        bf16x8 wf[FB][4];
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            const int f = min(f0 + 4 * i, nfr - 1), col = f * 32 + frow;
            const bf16* wb = W2 + (size_t)col * 64 + 8 * fhalf;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[i][ks] = load_bf16x8(wb + 16 * ks);
            bias[i] = b2 ? b2[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int grow = min(r0 + (r & 3) + 8 * (r >> 2) + 4 * fhalf, T - 1);
                rv[i][r] = resid ? resid[(size_t)grow * E + col] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[ks], wf[i][ks], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            const int f = f0 + 4 * i, col = f * 32 + frow;
            if (f < nfr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int grow = r0 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    if (grow < T) out[(size_t)grow * E + col] = acc[i][r] + bias[i] + rv[i][r];
                }
            }
        }
    }
}

}  // namespace

int pevit_tn_chunks(int T) { return ceil_div(T, TG_ROWS); }
int pevit_lna_blocks(int rows) { return ceil_div(rows, LNA_ROWS); }

int pevit_launch_prep_adapter(const float* w_down, const float* w_up, BottleneckPanels pan, int E, int layers, LayerStrides st,
                              hipStream_t s, int f32) {
    if (f32) hipLaunchKernelGGL(prep_adapter_kernel<float>, dim3(ceil_div(64 * E, 256), layers), dim3(256), 0, s, w_down, w_up, pan, E, st);
    else hipLaunchKernelGGL(prep_adapter_kernel<bf16>, dim3(ceil_div(64 * E, 256), layers), dim3(256), 0, s, w_down, w_up, pan, E, st);
    LAUNCH_OK("prep_adapter_kernel");
    return 0;
}
int pevit_launch_prep_compacter(const float* rule, const float* dWl, const float* dWr, const float* uWl, const float* uWr,
                                BottleneckPanels pan, int E, int layers, LayerStrides st, hipStream_t s, int f32) {
    if (E % 4) { pevit_set_error("prep_compacter: width %d not divisible by 4", E); return -1; }
    if (f32) hipLaunchKernelGGL(prep_compacter_kernel<float>, dim3(ceil_div(64 * E, 256), layers), dim3(256), 0, s, rule, dWl, dWr, uWl, uWr,
                                pan, E, st);
    else hipLaunchKernelGGL(prep_compacter_kernel<bf16>, dim3(ceil_div(64 * E, 256), layers), dim3(256), 0, s, rule, dWl, dWr, uWl, uWr,
                            pan, E, st);
    LAUNCH_OK("prep_compacter_kernel");
    return 0;
}
int pevit_launch_tn_gemm64(const bf16* X, int ldx, const bf16* Y, int ldy, float* partial, float* csx, float* csy, int T, int E,
                           hipStream_t s) {
    if (E % 64 || (ldx % 8) || (ldy % 8)) { pevit_set_error("tn_gemm64: bad shape E=%d ldx=%d ldy=%d", E, ldx, ldy); return -1; }
    hipLaunchKernelGGL(tn_gemm64_kernel, dim3(ceil_div(T, TG_ROWS) * (E / 64)), dim3(256), 0, s, X, ldx, Y, ldy, partial, csx,
                       csy, T, E);
    LAUNCH_OK("tn_gemm64_kernel");
    return 0;
}
int pevit_launch_ln_bwd_affine(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                               const float* dres, float* dx, bf16* dx_bf16, float* partial, int rows, int E, hipStream_t s, int f32) {
    if (E % 4 || E > 256 * LNA_MAXV) { pevit_set_error("ln_bwd_affine: unsupported width %d", E); return -1; }
    const size_t lds = (size_t)(LNA_WAVES - 1) * 3 * E * sizeof(float);      // 64.5 KiB at E = 768
    static bool attr_set = false;
    if (!attr_set) {
        const int maxlds = (LNA_WAVES - 1) * 3 * 256 * LNA_MAXV * (int)sizeof(float);
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_affine_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_affine_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
        attr_set = true;
    }
    if (f32) hipLaunchKernelGGL(ln_bwd_affine_kernel<float>, dim3(ceil_div(rows, LNA_ROWS)), dim3(64 * LNA_WAVES), lds, s, dy, x, mean, rstd,
                                gamma, dres, dx, dx_bf16, partial, rows, E);
    else hipLaunchKernelGGL(ln_bwd_affine_kernel<bf16>, dim3(ceil_div(rows, LNA_ROWS)), dim3(64 * LNA_WAVES), lds, s, dy, x, mean, rstd,
                            gamma, dres, dx, dx_bf16, partial, rows, E);
    LAUNCH_OK("ln_bwd_affine_kernel");
    return 0;
}
int pevit_launch_colsum_reduce(const float* partial, int chunks, int n, float* out, int layers, size_t partial_layer,
                               size_t out_layer, hipStream_t s) {
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(ceil_div(n, CR_COLS), layers), dim3(256), 0, s, partial, chunks, n, out,
                       partial_layer, out_layer);
    LAUNCH_OK("colsum_reduce_kernel");
    return 0;
}
int pevit_launch_colsum_reduce3(const float* partial, int chunks, int n, float* o0, float* o1, float* o2, int layers,
                                size_t partial_layer, size_t out_layer, hipStream_t s) {
    hipLaunchKernelGGL(colsum_reduce3_kernel, dim3(ceil_div(3 * n, CR_COLS), layers), dim3(256), 0, s, partial, chunks, n, o0, o1, o2,
                       partial_layer, out_layer);
    LAUNCH_OK("colsum_reduce3_kernel");
    return 0;
}
int pevit_launch_chain_adapter(const float* Gd, const float* Gu, float* g_down, float* g_up, int E, int layers, size_t g_layer,
                               size_t param_layer, hipStream_t s) {
    hipLaunchKernelGGL(chain_adapter_kernel, dim3(ceil_div(64 * E, 256), layers), dim3(256), 0, s, Gd, Gu, g_down, g_up, E,
                       g_layer, param_layer);
    LAUNCH_OK("chain_adapter_kernel");
    return 0;
}
int pevit_launch_chain_compacter(const float* Gd, const float* Gu, const float* rule, const float* params, float* grads, int E,
                                 int layers, size_t g_layer, size_t param_layer, size_t off_dWl, size_t off_dWr, size_t off_uWl,
                                 size_t off_uWr, hipStream_t s) {
    hipLaunchKernelGGL(chain_compacter_kernel, dim3(4, layers), dim3(64 * CC_WAVES), 0, s, Gd, Gu, rule, params, grads, E, g_layer,
                       param_layer, off_dWl, off_dWr, off_uWl, off_uWr);
    LAUNCH_OK("chain_compacter_kernel");
    return 0;
}

int pevit_launch_bottleneck_pair(int mode, const bf16* X, int ldx, const bf16* W1, const float* b1, const bf16* aux, bf16* S_out,
                                 bf16* S_pre, const bf16* W2, const float* b2, const float* resid, float* out, int T, int E,
                                 hipStream_t s) {
    if (E % 64 != 0 || T <= 0) { pevit_set_error("bottleneck_pair: bad shape T=%d E=%d", T, E); return -1; }
    const dim3 grid(ceil_div(T, BNK_ROWS)), block(256);
    switch (mode) {
        case BN_FWD_RELU: hipLaunchKernelGGL(bottleneck_pair_kernel<BN_FWD_RELU>, grid, block, 0, s, X, ldx, W1, b1, aux, S_out, S_pre, W2, b2, resid, out, T, E); break;
        case BN_FWD_GELUNEW: hipLaunchKernelGGL(bottleneck_pair_kernel<BN_FWD_GELUNEW>, grid, block, 0, s, X, ldx, W1, b1, aux, S_out, S_pre, W2, b2, resid, out, T, E); break;
        case BN_BWD_RELU: hipLaunchKernelGGL(bottleneck_pair_kernel<BN_BWD_RELU>, grid, block, 0, s, X, ldx, W1, b1, aux, S_out, S_pre, W2, b2, resid, out, T, E); break;
        case BN_BWD_GELUNEW: hipLaunchKernelGGL(bottleneck_pair_kernel<BN_BWD_GELUNEW>, grid, block, 0, s, X, ldx, W1, b1, aux, S_out, S_pre, W2, b2, resid, out, T, E); break;
        default: pevit_set_error("bottleneck_pair: unknown mode %d", mode); return -1;
    }
    LAUNCH_OK("bottleneck_pair_kernel");
    return 0;
}
