// f32-class VERIFICATION kernels (weight_format PEVIT_W_F32_VERIFY).
//
// BASELINE.md states the bf16 gates (logits <= 2e-2, gradients <= 5e-2 rel-L2) for parity with the reference.  On the
// randomly initialised, 12-24 layer towers of the fixtures, bf16 OPERAND ROUNDING ALONE moves logits by 3-10 % and
// individual gradients by up to 50 % (the f32 oracle with nothing changed but its contraction operands rounded to bf16
// shows the same spread, tests/test_gpu_tower.py) -- chaotic amplification, not a kernel property, and no
// bf16-operand engine can meet the stated gates on those inputs.  This mode separates "rounding" from "bug": the SAME
// launch sequences, memory layouts, index arithmetic (raw-reshape scramble, head layout, class-token pruning, chain
// rules, flat parameter buffer) and f32 kernels (LayerNorm, head, optimizer, chain rules) run with every activation and
// weight kept in f32, and the matrix-core contractions replaced by the plain f32 kernels of this file.  In this mode the
// stated gates are asserted against the reference's fixtures as written (tests/test_gpu_verify.py).  It is a test mode:
// ~20x slower, never benchmarked.
//
// Kernels: naive tiled f32 GEMM with the production epilogues (gemm_epilogue.h), attention forward / backward with f32
// softmax statistics, and the token-contracted low-rank adapter products.  All sums run in a fixed order (deterministic).
#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] B[N,K]^T, f32 operands.  64x64 tile, 256 threads, thread = 2 rows x 8 columns.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int BK = 32;
    __shared__ float As[64][BK + 1];
    __shared__ float Bs[64][BK + 1];
    const float* A = reinterpret_cast<const float*>(p.A);
    const float* B = reinterpret_cast<const float*>(p.B);
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int t = threadIdx.x, r0 = (t >> 3) * 2, c0 = (t & 7) * 8;
    float acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += BK) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = t + 256 * it, r = idx >> 5, k = idx & 31;
            int ar = m0 + r; ar = ar < p.M ? ar : p.M - 1;
            int br = n0 + r; br = br < p.Nb ? br : p.Nb - 1;
            As[r][k] = A[(size_t)ar * p.lda + k0 + k];
            Bs[r][k] = B[(size_t)br * p.ldb + k0 + k];
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < BK; ++k) {
            const float a0 = As[r0][k], a1 = As[r0 + 1][k];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float b = Bs[c0 + j][k];
                acc[0][j] = fmaf(a0, b, acc[0][j]);
                acc[1][j] = fmaf(a1, b, acc[1][j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + r0 + i, col = n0 + c0;
        if (row < p.M && col < p.N) epilogue_store<EPI, float>(p, row, col, acc[i]);
    }
}

template <int EPI>
int launch_f32(const GemmParams& p, hipStream_t s) {
    hipLaunchKernelGGL(gemm_f32_kernel<EPI>, dim3(ceil_div(p.N, 64), ceil_div(p.M, 64)), dim3(256), 0, s, p);
    LAUNCH_OK("gemm_f32_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// attention, one (batch, head) per workgroup, q/k/v (B*H, N, 64) f32, out / dout rows (b*N+n), cols h*64+d
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, float* __restrict__ out, int ldo,
                                                           float* __restrict__ lse, int H, int N) {
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const float* kh = k + (size_t)bh * N * 64;
    const float* vh = v + (size_t)bh * N * 64;
    for (int i = threadIdx.x; i < N; i += 256) {
        float qi[64];
#pragma unroll
        for (int d = 0; d < 64; ++d) qi[d] = q[((size_t)bh * N + i) * 64 + d];
        float m = -3.0e38f;
        for (int j = 0; j < N; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) s = fmaf(qi[d], kh[(size_t)j * 64 + d], s);
            m = fmaxf(m, s);
        }
        float l = 0.f, o[64];
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] = 0.f;
        for (int j = 0; j < N; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) s = fmaf(qi[d], kh[(size_t)j * 64 + d], s);
            const float pj = expf(s - m);
            l += pj;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = fmaf(pj, vh[(size_t)j * 64 + d], o[d]);
        }
        const float inv = 1.0f / l;
        float* dst = out + ((size_t)b * N + i) * ldo + h * 64;
#pragma unroll
        for (int d = 0; d < 64; ++d) dst[d] = o[d] * inv;
        lse[(size_t)bh * N + i] = m + logf(l);
    }
}

__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ out, int ldo,
                                                           const float* __restrict__ dout, int lddo,
                                                           const float* __restrict__ lse, float* __restrict__ dqkv, int ld,
                                                           int H, int N) {
    __shared__ float del[320];
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H, E = H * 64;
    const float* qh = q + (size_t)bh * N * 64;
    const float* kh = k + (size_t)bh * N * 64;
    const float* vh = v + (size_t)bh * N * 64;
    const float* oh = out + (size_t)b * N * ldo + h * 64;
    const float* doh = dout + (size_t)b * N * lddo + h * 64;
    const float* lh = lse + (size_t)bh * N;
    for (int i = threadIdx.x; i < N; i += 256) {
        float s = 0.f;
        for (int d = 0; d < 64; ++d) s = fmaf(doh[(size_t)i * lddo + d], oh[(size_t)i * ldo + d], s);
        del[i] = s;
    }
    __syncthreads();
    // dQ_i = sum_j p_ij (dP_ij - delta_i) k_j
    for (int i = threadIdx.x; i < N; i += 256) {
        float qi[64], doi[64], acc[64];
#pragma unroll
        for (int d = 0; d < 64; ++d) { qi[d] = qh[(size_t)i * 64 + d]; doi[d] = doh[(size_t)i * lddo + d]; acc[d] = 0.f; }
        const float li = lh[i], di = del[i];
        for (int j = 0; j < N; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) { s = fmaf(qi[d], kh[(size_t)j * 64 + d], s); dp = fmaf(doi[d], vh[(size_t)j * 64 + d], dp); }
            const float ds = expf(s - li) * (dp - di);
#pragma unroll
            for (int d = 0; d < 64; ++d) acc[d] = fmaf(ds, kh[(size_t)j * 64 + d], acc[d]);
        }
        float* dst = dqkv + ((size_t)b * N + i) * ld + h * 64;
#pragma unroll
        for (int d = 0; d < 64; ++d) dst[d] = acc[d];
    }
    // dK_j = sum_i ds_ij q_i ; dV_j = sum_i p_ij dO_i
    for (int j = threadIdx.x; j < N; j += 256) {
        float kj[64], vj[64], ak[64], av[64];
#pragma unroll
        for (int d = 0; d < 64; ++d) { kj[d] = kh[(size_t)j * 64 + d]; vj[d] = vh[(size_t)j * 64 + d]; ak[d] = 0.f; av[d] = 0.f; }
        for (int i = 0; i < N; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) { s = fmaf(qh[(size_t)i * 64 + d], kj[d], s); dp = fmaf(doh[(size_t)i * lddo + d], vj[d], dp); }
            const float pij = expf(s - lh[i]);
            const float ds = pij * (dp - del[i]);
#pragma unroll
            for (int d = 0; d < 64; ++d) {
                ak[d] = fmaf(ds, qh[(size_t)i * 64 + d], ak[d]);
                av[d] = fmaf(pij, doh[(size_t)i * lddo + d], av[d]);
            }
        }
        float* dst = dqkv + ((size_t)b * N + j) * ld + h * 64;
#pragma unroll
        for (int d = 0; d < 64; ++d) { dst[E + d] = ak[d]; dst[2 * E + d] = av[d]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// low-rank adapter products (lowrank.hip states the math and the flat addressing)
__device__ __forceinline__ int vrow_of_ref(int rr, int B, int N) {
    const int n = rr / B, b = rr - n * B;
    return b * N + n;
}
__device__ __forceinline__ const float* vslab(const float* dqkv, int ld, int col0, int rr, int e0, int E, int H, int N) {
    const int c = (int)(((long long)rr * E + e0) >> 6);
    const int bh = c / N, n = c - bh * N;
    const int b = bh / H, h = bh - b * H;
    return dqkv + ((size_t)b * N + n) * ld + col0 + h * 64;
}

// u[row(rr)][j] = sum_e dDelta_{q|v}[rr][e] Q32[e][j]    thread = (rr, j)
__global__ void lowrank_u_f32_kernel(const float* __restrict__ dqkv, int ld, const float* __restrict__ q32, float* __restrict__ u32,
                                     float* __restrict__ ucols, int B, int H, int N, int E) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int T = B * N;
    if (idx >= T * 64) return;
    const int rr = idx >> 6, j = idx & 63;
    const int col0 = j < 32 ? 0 : 2 * E;
    float s = 0.f;
    for (int e0 = 0; e0 < E; e0 += 64) {
        const float* slab = vslab(dqkv, ld, col0, rr, e0, E, H, N);
        for (int d = 0; d < 64; ++d) s = fmaf(slab[d], q32[(size_t)(e0 + d) * 64 + j], s);
    }
    const int row = vrow_of_ref(rr, B, N);
    u32[(size_t)row * 64 + j] = s;
    ucols[(size_t)row * ld + j] = s;
}

// same partial layout as lowrank_grad_kernel: partial[chunk][4][E][32], dbias_partial[chunk][2][E]
constexpr int VG_ROWS = 256;       // == LG_ROWS of lowrank.hip (pevit_lowrank_chunks)
__global__ __launch_bounds__(256) void lowrank_grad_f32_kernel(const float* __restrict__ xn, int ldx, const float* __restrict__ u32,
                                                               const float* __restrict__ dqkv, int ld, const float* __restrict__ t,
                                                               float* __restrict__ partial, float* __restrict__ dbias_partial,
                                                               int B, int H, int N, int E) {
    const int epairs = E / 64, per_chunk = epairs * 3;
    const int chunk = blockIdx.x / per_chunk, rem = blockIdx.x - chunk * per_chunk;
    const int kind = rem / epairs, e0 = (rem - kind * epairs) * 64;
    const int T = B * N, r0 = chunk * VG_ROWS, r1 = min(T, r0 + VG_ROWS);
    const size_t plane = (size_t)E * 32;
    const int tid = threadIdx.x;
    if (kind == 0) {        // G0/G1[e][j] = sum_r xn[r][e] u[r][j | 32+j]
        for (int o = tid; o < 64 * 64; o += 256) {
            const int e = e0 + (o >> 6), jj = o & 63;
            float s = 0.f;
            for (int r = r0; r < r1; ++r) s = fmaf(xn[(size_t)r * ldx + e], u32[(size_t)r * 64 + jj], s);
            partial[((size_t)chunk * 4 + (jj >> 5)) * plane + (size_t)e * 32 + (jj & 31)] = s;
        }
    } else {                // G2/G3[e][j] = sum_rr dDelta[rr][e] t[row(rr)][j (+32)] ; d bias = column sums of dDelta
        const int col0 = kind == 1 ? 0 : 2 * E, toff = kind == 1 ? 0 : 32;
        for (int o = tid; o < 64 * 32; o += 256) {
            const int d = o >> 5, j = o & 31;
            float s = 0.f;
            for (int rr = r0; rr < r1; ++rr)
                s = fmaf(vslab(dqkv, ld, col0, rr, e0, E, H, N)[d], t[(size_t)vrow_of_ref(rr, B, N) * 64 + toff + j], s);
            partial[((size_t)chunk * 4 + kind + 1) * plane + (size_t)(e0 + d) * 32 + j] = s;
        }
        if (tid < 64) {
            float s = 0.f;
            for (int rr = r0; rr < r1; ++rr) s += vslab(dqkv, ld, col0, rr, e0, E, H, N)[tid];
            dbias_partial[((size_t)chunk * 2 + (kind - 1)) * E + e0 + tid] = s;
        }
    }
}

// post-MLP adapters: partial[chunk][E][64], csx[chunk][E], csy[chunk][64] exactly as tn_gemm64_kernel writes them
constexpr int VT_ROWS = 256;       // == TG_ROWS of adapter.hip (pevit_tn_chunks)
__global__ __launch_bounds__(256) void tn_gemm64_f32_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ Y, int ldy,
                                                            float* __restrict__ partial, float* __restrict__ csx,
                                                            float* __restrict__ csy, int T, int E) {
    const int eslabs = E / 64;
    const int chunk = blockIdx.x / eslabs, e0 = (blockIdx.x - chunk * eslabs) * 64;
    const int r0 = chunk * VT_ROWS, r1 = min(T, r0 + VT_ROWS), tid = threadIdx.x;
    for (int o = tid; o < 64 * 64; o += 256) {
        const int e = e0 + (o >> 6), j = o & 63;
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s = fmaf(X[(size_t)r * ldx + e], Y[(size_t)r * ldy + j], s);
        partial[(size_t)chunk * E * 64 + (size_t)e * 64 + j] = s;
    }
    if (tid < 64) {
        if (csx) { float s = 0.f; for (int r = r0; r < r1; ++r) s += X[(size_t)r * ldx + e0 + tid]; csx[(size_t)chunk * E + e0 + tid] = s; }
        if (csy && e0 == 0) { float s = 0.f; for (int r = r0; r < r1; ++r) s += Y[(size_t)r * ldy + tid]; csy[(size_t)chunk * 64 + tid] = s; }
    }
}

}  // namespace

int pevit_launch_tn_gemm64_f32(const float* X, int ldx, const float* Y, int ldy, float* partial, float* csx, float* csy, int T, int E,
                               hipStream_t s) {
    if (E % 64) { pevit_set_error("tn_gemm64 (f32 verification): bad width %d", E); return -1; }
    hipLaunchKernelGGL(tn_gemm64_f32_kernel, dim3(ceil_div(T, VT_ROWS) * (E / 64)), dim3(256), 0, s, X, ldx, Y, ldy, partial, csx, csy, T, E);
    LAUNCH_OK("tn_gemm64_f32_kernel");
    return 0;
}

int pevit_launch_gemm_f32(int epi, const GemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.K % 32 || p.N % 8) {
        pevit_set_error("gemm (f32 verification): bad problem M=%d N=%d K=%d", p.M, p.N, p.K); return -1;
    }
    switch (epi) {
        case EPI_QKV_HEADS: return launch_f32<EPI_QKV_HEADS>(p, s);
        case EPI_BIAS_RESID_F32: return launch_f32<EPI_BIAS_RESID_F32>(p, s);
        case EPI_BIAS_GELU: return launch_f32<EPI_BIAS_GELU>(p, s);
        case EPI_DGELU_BF16: return launch_f32<EPI_DGELU_BF16>(p, s);
        case EPI_F32: return launch_f32<EPI_F32>(p, s);
        case EPI_BF16: return launch_f32<EPI_BF16>(p, s);
        case EPI_BIAS_BF16: return launch_f32<EPI_BIAS_BF16>(p, s);
        case EPI_PATCH_EMBED: return launch_f32<EPI_PATCH_EMBED>(p, s);
        case EPI_BIAS_RELU_BF16: return launch_f32<EPI_BIAS_RELU_BF16>(p, s);
        case EPI_BIAS_RESID_KEEP: return launch_f32<EPI_BIAS_RESID_KEEP>(p, s);
        case EPI_BIAS_GELUNEW: return launch_f32<EPI_BIAS_GELUNEW>(p, s);
        case EPI_DRELU_BF16: return launch_f32<EPI_DRELU_BF16>(p, s);
        case EPI_DGELUNEW_BF16: return launch_f32<EPI_DGELUNEW_BF16>(p, s);
    }
    pevit_set_error("gemm (f32 verification): unknown epilogue %d", epi);
    return -1;
}

int pevit_launch_attn_fwd_f32(const float* q, const float* k, const float* v, float* out, int ldo, float* lse, int B, int H, int N,
                              hipStream_t s) {
    if (N < 1 || N > 320) { pevit_set_error("attn_fwd (f32 verification): N=%d outside [1,320]", N); return -1; }
    hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(B * H), dim3(256), 0, s, q, k, v, out, ldo, lse, H, N);
    LAUNCH_OK("attn_fwd_f32_kernel");
    return 0;
}

int pevit_launch_attn_bwd_f32(const float* q, const float* k, const float* v, const float* out, int ldo, const float* dout,
                              int lddo, const float* lse, float* dqkv, int ld, int B, int H, int N, hipStream_t s) {
    if (N < 1 || N > 320) { pevit_set_error("attn_bwd (f32 verification): N=%d outside [1,320]", N); return -1; }
    hipLaunchKernelGGL(attn_bwd_f32_kernel, dim3(B * H), dim3(256), 0, s, q, k, v, out, ldo, dout, lddo, lse, dqkv, ld, H, N);
    LAUNCH_OK("attn_bwd_f32_kernel");
    return 0;
}

int pevit_launch_lowrank_u_f32(const float* dqkv, int ld, const float* q32, float* u32, float* ucols, int B, int H, int N, int E,
                               hipStream_t s) {
    hipLaunchKernelGGL(lowrank_u_f32_kernel, dim3(ceil_div(B * N * 64, 256)), dim3(256), 0, s, dqkv, ld, q32, u32, ucols, B, H, N, E);
    LAUNCH_OK("lowrank_u_f32_kernel");
    return 0;
}

int pevit_launch_lowrank_grad_f32(const float* xn, int ldx, const float* u32, const float* dqkv, int ld, const float* t,
                                  float* partial, float* dbias_partial, int chunks, int B, int H, int N, int E, hipStream_t s) {
    if (chunks != ceil_div(B * N, VG_ROWS)) { pevit_set_error("lowrank_grad (f32 verification): chunks mismatch"); return -1; }
    hipLaunchKernelGGL(lowrank_grad_f32_kernel, dim3(chunks * (E / 64) * 3), dim3(256), 0, s, xn, ldx, u32, dqkv, ld, t, partial,
                       dbias_partial, B, H, N, E);
    LAUNCH_OK("lowrank_grad_f32_kernel");
    return 0;
}
