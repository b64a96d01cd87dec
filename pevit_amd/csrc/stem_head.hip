// Stem (patch embedding) and classifier head of the fine-tune step.
//
// Stem, reference model.py:1035-1040: conv2d(k=s=P, no bias) == GEMM over im2col patches;
// class token + positional embedding; ln_pre.  Only the im2col gather and the class-token row
// live here, the contraction itself is the MFMA GEMM (EPI_PATCH_EMBED adds the positional rows).
//
// Head, reference kadaptation_clip.py:128-132,176-185,276: BatchNorm1d(D, affine=False) ->
// Linear(D, C) -> CrossEntropyLoss(mean), plus its backward.  B x D x C is tiny (128x512x100),
// so these are plain f32 VALU kernels: they are launch-latency, not throughput, work.
#include "common.h"
#include "kernels.h"

namespace {

// patches[(b*G2 + gy*G + gx)][c*P*P + i*P + j] = img[b][c][gy*P+i][gx*P+j]   (bf16, zero padded to Kp)
template <typename ST>
__global__ void im2col_kernel(const float* __restrict__ img, bf16* __restrict__ out, int B, int R, int P, int Kp) {
    const int G = R / P, G2 = G * G, K = 3 * P * P;
    const size_t total = (size_t)B * G2 * (Kp / 2);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k2 = (int)(idx % (Kp / 2));
        const size_t row = idx / (Kp / 2);
        const int b = (int)(row / G2), gidx = (int)(row - (size_t)b * G2);
        const int gy = gidx / G, gx = gidx - gy * G;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = 2 * k2 + u;
            float v = 0.f;
            if (k < K) {
                const int c = k / (P * P), rem = k - c * P * P;
                const int i = rem / P, j = rem - i * P;
                v = img[(((size_t)b * 3 + c) * R + gy * P + i) * R + gx * P + j];
            }
            st_store<ST>(out, row * Kp + 2 * k2 + u, v);
        }
    }
}

// P % 8 == 0 (patch 16 / 32): 8 consecutive k = 8 consecutive pixels of one image row -> two float4 loads, one 16-byte
// (bf16) store per thread; the generic kernel above moved 2 elements per thread with a division chain each and ran at
// half the HBM rate (55 us for 115 MB at ViT-B/32, B = 128).
template <typename ST>
__global__ void im2col8_kernel(const float* __restrict__ img, bf16* __restrict__ out, int B, int R, int P, int Kp) {
    const int G = R / P, G2 = G * G, K = 3 * P * P, K8 = Kp / 8;
    const size_t total = (size_t)B * G2 * K8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % K8) * 8;
        const size_t row = idx / K8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (k < K) {
            const int b = (int)(row / G2), gidx = (int)(row - (size_t)b * G2);
            const int gy = gidx / G, gx = gidx - gy * G;
            const int c = k / (P * P), rem = k - c * P * P;
            const int i = rem / P, j = rem - i * P;
            const float* src = img + (((size_t)b * 3 + c) * R + gy * P + i) * R + gx * P + j;
            const float4 a = *reinterpret_cast<const float4*>(src), d = *reinterpret_cast<const float4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
        }
        st_store4<ST>(out, row * Kp + k, v[0], v[1], v[2], v[3]);
        st_store4<ST>(out, row * Kp + k + 4, v[4], v[5], v[6], v[7]);
    }
}

// The same gather from uint8 pixels with the reference's preprocessing folded in: ToTensor + Normalize of the dataset transforms
// (feature.py:537-542, vitb32_CLIP.yaml INPUT.MEAN / STD), x = (u8 / 255 - mean[c]) / std[c] in f32 with correctly rounded
// divisions -- bit for bit what `(x.float() / 255.0 - mean) / std` gives on the host -- then the bf16 rounding of the patches.
// A quarter of the bytes to upload and to read here (8 pixels = one 8-byte load).  norm = {mean[3], std[3]}.
struct PixelNorm { float mean[3], stdv[3]; };
template <typename ST>
__global__ void im2col8_u8_kernel(const unsigned char* __restrict__ img, bf16* __restrict__ out, int B, int R, int P, int Kp,
                                  PixelNorm nm) {
    const int G = R / P, G2 = G * G, K = 3 * P * P, K8 = Kp / 8;
    const size_t total = (size_t)B * G2 * K8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % K8) * 8;
        const size_t row = idx / K8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (k < K) {
            const int b = (int)(row / G2), gidx = (int)(row - (size_t)b * G2);
            const int gy = gidx / G, gx = gidx - gy * G;
            const int c = k / (P * P), rem = k - c * P * P;
            const int i = rem / P, j = rem - i * P;
            const uint2 raw = *reinterpret_cast<const uint2*>(img + (((size_t)b * 3 + c) * R + gy * P + i) * R + gx * P + j);
            const float m = nm.mean[c], sd = nm.stdv[c];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned px = ((u < 4 ? raw.x : raw.y) >> (8 * (u & 3))) & 255u;
                v[u] = ((float)px / 255.0f - m) / sd;
            }
        }
        st_store4<ST>(out, row * Kp + k, v[0], v[1], v[2], v[3]);
        st_store4<ST>(out, row * Kp + k + 4, v[4], v[5], v[6], v[7]);
    }
}
template <typename ST>
__global__ void im2col_u8_kernel(const unsigned char* __restrict__ img, bf16* __restrict__ out, int B, int R, int P, int Kp, PixelNorm nm) {
    const int G = R / P, G2 = G * G, K = 3 * P * P;
    const size_t total = (size_t)B * G2 * (Kp / 2);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k2 = (int)(idx % (Kp / 2));
        const size_t row = idx / (Kp / 2);
        const int b = (int)(row / G2), gidx = (int)(row - (size_t)b * G2);
        const int gy = gidx / G, gx = gidx - gy * G;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = 2 * k2 + u;
            float v = 0.f;
            if (k < K) {
                const int c = k / (P * P), rem = k - c * P * P;
                const int i = rem / P, j = rem - i * P;
                v = ((float)img[(((size_t)b * 3 + c) * R + gy * P + i) * R + gx * P + j] / 255.0f - nm.mean[c]) / nm.stdv[c];
            }
            st_store<ST>(out, row * Kp + 2 * k2 + u, v);
        }
    }
}

// conv1.weight (E, 3, P, P) f32 -> [E][Kp] bf16 zero padded
template <typename ST>
__global__ void conv_weight_kernel(const float* __restrict__ w, bf16* __restrict__ out, int E, int K, int Kp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)E * Kp) return;
    const int e = (int)(idx / Kp), k = (int)(idx - (size_t)e * Kp);
    st_store<ST>(out, idx, k < K ? w[(size_t)e * K + k] : 0.f);
}

// x[b*N + 0][:] = class_embedding + positional_embedding[0]
__global__ void cls_row_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x,
                               int B, int N, int E) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < E; e += blockDim.x) x[(size_t)b * N * E + e] = cls[e] + pos[e];
}

// ---- head -------------------------------------------------------------------------
// BatchNorm1d(D, affine=False): one workgroup per 16 features; thread = (feature f = tid & 15, row group g = tid >> 4),
// the 16 row groups split the batch and combine through LDS in a fixed order (D/16 workgroups, B/16 loads deep: the
// 64-feature form had 8 workgroups walking 32 dependent loads each and took 17 us for 128 x 512 floats).
__device__ __forceinline__ float bn_reduce16(float v, float (*red)[17], int f, int g) {
    red[g][f] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][f];
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(256) void bn_fwd_kernel(const float* __restrict__ feat, float* __restrict__ y,
                                                     float* __restrict__ rstd_out, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, int training, int B, int D) {
    __shared__ float red[16][17];
    const int f = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int d = blockIdx.x * 16 + f;
    const bool ok = d < D;
    float mean, var;
    if (training) {
        float s = 0.f;
        for (int b = g; b < B; b += 16) s += ok ? feat[(size_t)b * D + d] : 0.f;
        mean = bn_reduce16(s, red, f, g) / (float)B;
        float q = 0.f;
        for (int b = g; b < B; b += 16) { const float t = ok ? feat[(size_t)b * D + d] - mean : 0.f; q += t * t; }
        const float qq = bn_reduce16(q, red, f, g);
        var = qq / (float)B;                                   // biased: used for normalisation
        if (g == 0 && ok) {
            const float unbiased = B > 1 ? qq / (float)(B - 1) : var;
            running_mean[d] = 0.9f * running_mean[d] + 0.1f * mean;
            running_var[d] = 0.9f * running_var[d] + 0.1f * unbiased;
        }
    } else {
        mean = ok ? running_mean[d] : 0.f; var = ok ? running_var[d] : 1.f;
    }
    const float rstd = rsqrtf(var + 1e-5f);
    if (g == 0 && ok) rstd_out[d] = rstd;
    if (ok) for (int b = g; b < B; b += 16) y[(size_t)b * D + d] = (feat[(size_t)b * D + d] - mean) * rstd;
}

// dfeat = rstd * (dy - mean_b(dy) - yhat * mean_b(dy*yhat))   (training) ;  rstd * dy (eval)
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yhat,
                                                     const float* __restrict__ rstd, float* __restrict__ dfeat,
                                                     int training, int B, int D, bf16* __restrict__ dfeat_b16) {
    __shared__ float red[16][17];
    const int f = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int d = blockIdx.x * 16 + f;
    const bool ok = d < D;
    float m1 = 0.f, m2 = 0.f;
    if (training) {
        float a = 0.f, c = 0.f;
        for (int b = g; b < B; b += 16)
            if (ok) { const float gg = dy[(size_t)b * D + d]; a += gg; c += gg * yhat[(size_t)b * D + d]; }
        m1 = bn_reduce16(a, red, f, g) / (float)B;
        m2 = bn_reduce16(c, red, f, g) / (float)B;
    }
    if (!ok) return;
    const float r = rstd[d];
    for (int b = g; b < B; b += 16) {
        const float gg = dy[(size_t)b * D + d];
        const float o = training ? r * (gg - m1 - yhat[(size_t)b * D + d] * m2) : r * gg;
        dfeat[(size_t)b * D + d] = o;
        if (dfeat_b16) dfeat_b16[(size_t)b * D + d] = f2bf(o);      // the operand of the projection backward (round 5: no cast launch)
    }
}

// Small dense products of the head on the f32 matrix core (exact f32, like the reference's head):
//   C[i][j] (+)= sum_k A[i*sAi + k*sAk] * B[k*sBk + j*sBj]  (+ bias[j]) ; optional colsum of A.
// One workgroup per 32x32 output tile; its 4 waves split K and combine through LDS.
struct SmallGemm {
    const float* A; long sAi, sAk;
    const float* B; long sBk, sBj;
    float* C; long ldc;
    const float* bias;       // per column j, or null
    float* rowsumA;          // rowsumA[i] += sum_k A(i,k)   (written by the j-tile 0 workgroups), or null
    int M, N, K, accumulate;
};
__global__ __launch_bounds__(256) void small_gemm_kernel(SmallGemm p) {
    __shared__ float red[3][64][17];
    __shared__ float rs[3][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 31, lk = lane >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int i = i0 + li, j = j0 + li;
    const bool iok = i < p.M, jok = j < p.N;
    const float* ap = p.A + (long)(iok ? i : 0) * p.sAi;
    const float* bp = p.B + (long)(jok ? j : 0) * p.sBj;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float rsum = 0.f;
    // chunks of 8 k: lane (row, half) takes k = 8c + 4 half + u of chunk c for its u-th MFMA (the contraction order is free as long
    // as A and B agree); an operand that is contiguous along k is read with one 16-byte load per chunk (the logits product, K = 512:
    // both; the d-feature product: dlogits).  With 4-byte loads 2 KiB apart the logits product took 22 us of the head's 60.
    const bool va = p.sAk == 1 && (p.sAi & 3) == 0 && (reinterpret_cast<size_t>(p.A) & 15) == 0;
    const bool vb = p.sBk == 1 && (p.sBj & 3) == 0 && (reinterpret_cast<size_t>(p.B) & 15) == 0;
    const int nchunks = (va || vb) ? p.K >> 3 : 0;
    constexpr int CU = 4;       // chunks requested ahead of their MFMAs (8 and 16 pairs below measured no faster: 16.1 / 11.6 / 7.4 us vs 16.3 / 12.2 / 6.2)
    for (int cb = wid; cb < nchunks; cb += 4 * CU) {
        float4 a4[CU], b4[CU];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int c = cb + 4 * u;
            const bool cok = c < nchunks;
            const long k = 8 * (cok ? c : 0) + 4 * lk;
            a4[u] = b4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cok && iok) {
                if (va) a4[u] = *reinterpret_cast<const float4*>(ap + k);
                else a4[u] = make_float4(ap[k * p.sAk], ap[(k + 1) * p.sAk], ap[(k + 2) * p.sAk], ap[(k + 3) * p.sAk]);
            }
            if (cok && jok) {
                if (vb) b4[u] = *reinterpret_cast<const float4*>(bp + k);
                else b4[u] = make_float4(bp[k * p.sBk], bp[(k + 1) * p.sBk], bp[(k + 2) * p.sBk], bp[(k + 3) * p.sBk]);
            }
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u].x, b4[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u].y, b4[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u].z, b4[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u].w, b4[u].w, acc, 0, 0, 0);
            rsum += (a4[u].x + a4[u].y) + (a4[u].z + a4[u].w);
        }
    }
    // the rest (everything when neither operand is contiguous along k): wave w takes k-pairs w, w+4, ... ; 8 pairs are loaded ahead
    const int kbase = nchunks * 8;
    const int npairs = (p.K - kbase + 1) / 2;
    constexpr int PU = 8;
    for (int pb = wid; pb < npairs; pb += 4 * PU) {
        float a[PU], b[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int k = kbase + 2 * (pb + 4 * u) + lk;
            const bool kok = (pb + 4 * u) < npairs && k < p.K;
            a[u] = (kok && iok) ? ap[(long)k * p.sAk] : 0.f;
            b[u] = (kok && jok) ? bp[(long)k * p.sBk] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
            rsum += a[u];
        }
    }
    if (wid > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wid - 1][lane][r] = acc[r];
        rs[wid - 1][lane] = rsum;
    }
    __syncthreads();
    if (wid != 0) return;
    // the bias and (when accumulating) all 16 old values are requested before the first store, from clamped addresses: loaded one
    // by one inside the bounds branch they were 16 consecutive load -> wait -> store round trips
    const float* bsrc = p.bias ? p.bias : p.C;
    const float bv0 = bsrc[jok ? j : 0];
    const float bv = p.bias ? bv0 : 0.f;
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oi = min(i0 + (r & 3) + 8 * (r >> 2) + 4 * lk, p.M - 1);
        old[r] = p.C[(long)oi * p.ldc + (jok ? j : 0)];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), visible to the compiler: nothing is pending behind the stores below
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = acc[r] + red[0][lane][r] + red[1][lane][r] + red[2][lane][r];
        const int oi = i0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (oi < p.M && jok) p.C[(long)oi * p.ldc + j] = (p.accumulate ? old[r] : 0.f) + v + bv;
    }
    if (p.rowsumA && blockIdx.x == 0) {
        rsum += rs[0][lane] + rs[1][lane] + rs[2][lane];
        rsum += __shfl_xor(rsum, 32, 64);
        if (lk == 0 && iok) p.rowsumA[i] += rsum;
    }
}

// torch.nn.CrossEntropyLoss(reduction="mean") semantics for class-index targets: rows whose target equals the default
// ignore_index (-100) contribute nothing and the mean runs over the remaining rows; any other target outside [0, C) is an
// error in torch (device assert) and poisons the loss with NaN here instead of reading out of bounds.
constexpr int CE_IGNORE = -100;
__device__ __forceinline__ float ce_valid_rows(const int64_t* __restrict__ labels, int B, int lane) {
    float n = 0.f;
    for (int b = lane; b < B; b += 64) n += labels[b] != CE_IGNORE ? 1.f : 0.f;
    return wave_sum(n);
}

// one wave per row: log-softmax, row loss, dlogits = (softmax - onehot) / (number of non-ignored rows)
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                 float* __restrict__ dlogits, float* __restrict__ rowloss, int B,
                                                 int Cc) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float nvalid = ce_valid_rows(labels, B, lane);
    const float* lr = logits + (size_t)b * Cc;
    float m = -3.0e38f;
    for (int c = lane; c < Cc; c += 64) m = fmaxf(m, lr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < Cc; c += 64) s += __expf(lr[c] - m);
    s = wave_sum(s);
    const float lse = m + __logf(s);
    const int64_t lab64 = labels[b];
    const bool ignored = lab64 == CE_IGNORE, bad = !ignored && (lab64 < 0 || lab64 >= Cc);
    const int lab = (int)lab64;
    const float nanv = __builtin_nanf("");
    for (int c = lane; c < Cc; c += 64) {
        const float pr = __expf(lr[c] - lse);
        const float g = (pr - (c == lab ? 1.f : 0.f)) / nvalid;
        dlogits[(size_t)b * Cc + c] = ignored ? 0.f : (bad ? nanv : g);
    }
    if (lane == 0) rowloss[b] = ignored ? 0.f : (bad ? nanv : lse - lr[lab]);
}

// ce_kernel + loss_mean_kernel as ONE workgroup (round 5: one launch less on the class-token tail): wave w takes rows w, w + 16, ...
// with exactly ce_kernel's arithmetic, then wave 0 forms the mean with exactly loss_mean_kernel's summation order.
__global__ __launch_bounds__(1024) void ce_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                       float* __restrict__ dlogits, float* __restrict__ rowloss,
                                                       float* __restrict__ loss, int B, int Cc) {
    __shared__ float rl_s[2048];                      // the row losses, for the mean behind the barrier (no trip through memory)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float nvalid = ce_valid_rows(labels, B, lane);
    const float nanv = __builtin_nanf("");
    if (Cc <= 128) {
        // Up to 128 classes: a lane holds columns lane and lane + 64.  The rows of a wave go EIGHT at a time with every logit and
        // label of the batch requested before the first reduction (round 5: with one row per trip each row was its own memory
        // round trip -- 20.5 us for 128 x 100 logits; the arithmetic and its order are unchanged, hence the same bits).
        constexpr int RU = 8;
        const int c0 = min(lane, Cc - 1), c1 = min(lane + 64, Cc - 1);
        const bool ok0 = lane < Cc, ok1 = lane + 64 < Cc;
        for (int b0 = wid; b0 < B; b0 += 16 * RU) {
            float v0[RU], v1[RU];
            int64_t lb[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int b = min(b0 + 16 * u, B - 1);
                v0[u] = logits[(size_t)b * Cc + c0];
                v1[u] = logits[(size_t)b * Cc + c1];
                lb[u] = labels[b];
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int b = b0 + 16 * u;
                if (b >= B) break;                         // wave-uniform
                float m = -3.0e38f;
                if (ok0) m = fmaxf(m, v0[u]);
                if (ok1) m = fmaxf(m, v1[u]);
                m = wave_max(m);
                float s = 0.f;
                if (ok0) s += __expf(v0[u] - m);
                if (ok1) s += __expf(v1[u] - m);
                s = wave_sum(s);
                const float lse = m + __logf(s);
                const int64_t lab64 = lb[u];
                const bool ignored = lab64 == CE_IGNORE, bad = !ignored && (lab64 < 0 || lab64 >= Cc);
                const int lab = (int)lab64;
                if (ok0) {
                    const float g = (__expf(v0[u] - lse) - (lane == lab ? 1.f : 0.f)) / nvalid;
                    dlogits[(size_t)b * Cc + lane] = ignored ? 0.f : (bad ? nanv : g);
                }
                if (ok1) {
                    const float g = (__expf(v1[u] - lse) - (lane + 64 == lab ? 1.f : 0.f)) / nvalid;
                    dlogits[(size_t)b * Cc + lane + 64] = ignored ? 0.f : (bad ? nanv : g);
                }
                // the label's logit: held by lane lab % 64 (no load behind the reductions)
                const float at = __shfl((lab & 64) ? v1[u] : v0[u], lab & 63, 64);
                if (lane == 0) { const float rl = ignored ? 0.f : (bad ? nanv : lse - at); rowloss[b] = rl; rl_s[b] = rl; }
            }
        }
    } else
    for (int b = wid; b < B; b += 16) {
        const float* lr = logits + (size_t)b * Cc;
        float m = -3.0e38f;
        for (int c = lane; c < Cc; c += 64) m = fmaxf(m, lr[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < Cc; c += 64) s += __expf(lr[c] - m);
        s = wave_sum(s);
        const float lse = m + __logf(s);
        const int64_t lab64 = labels[b];
        const bool ignored = lab64 == CE_IGNORE, bad = !ignored && (lab64 < 0 || lab64 >= Cc);
        const int lab = (int)lab64;
        for (int c = lane; c < Cc; c += 64) {
            const float pr = __expf(lr[c] - lse);
            const float g = (pr - (c == lab ? 1.f : 0.f)) / nvalid;
            dlogits[(size_t)b * Cc + c] = ignored ? 0.f : (bad ? nanv : g);
        }
        if (lane == 0) { const float rl = ignored ? 0.f : (bad ? nanv : lse - lr[lab]); rowloss[b] = rl; rl_s[b] = rl; }
    }
    __syncthreads();
    if (wid == 0) {
        float s = 0.f;
        for (int b = lane; b < B; b += 64) s += rl_s[b];
        s = wave_sum(s);
        if (lane == 0) loss[0] = s / nvalid;
    }
}

// loss = sum(rowloss) / (number of non-ignored rows): one wave, fixed summation order
__global__ void loss_mean_kernel(const float* __restrict__ rowloss, const int64_t* __restrict__ labels,
                                 float* __restrict__ loss, int B) {
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) s += rowloss[b];
    s = wave_sum(s);
    const float nvalid = ce_valid_rows(labels, B, threadIdx.x);
    if (threadIdx.x == 0) loss[0] = s / nvalid;
}

}  // namespace

int pevit_launch_im2col(const float* img, bf16* out, int B, int R, int P, int Kp, hipStream_t s, int f32) {
    if (P % 8 == 0 && R % 4 == 0) {
        const size_t total8 = (size_t)B * (R / P) * (R / P) * (Kp / 8);
        const int blocks8 = (int)((total8 + 255) / 256 > 16384 ? 16384 : (total8 + 255) / 256);
        if (f32) hipLaunchKernelGGL(im2col8_kernel<float>, dim3(blocks8), dim3(256), 0, s, img, out, B, R, P, Kp);
        else hipLaunchKernelGGL(im2col8_kernel<bf16>, dim3(blocks8), dim3(256), 0, s, img, out, B, R, P, Kp);
        LAUNCH_OK("im2col8_kernel");
        return 0;
    }
    const size_t total = (size_t)B * (R / P) * (R / P) * (Kp / 2);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    if (f32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(blocks), dim3(256), 0, s, img, out, B, R, P, Kp);
    else hipLaunchKernelGGL(im2col_kernel<bf16>, dim3(blocks), dim3(256), 0, s, img, out, B, R, P, Kp);
    LAUNCH_OK("im2col_kernel");
    return 0;
}
int pevit_launch_im2col_u8(const unsigned char* img, const float* mean3, const float* std3, bf16* out, int B, int R, int P, int Kp,
                           hipStream_t s, int f32) {
    PixelNorm nm;
    for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3[c]; nm.stdv[c] = std3[c]; }
    if (P % 8 == 0 && R % 8 == 0) {
        const size_t total8 = (size_t)B * (R / P) * (R / P) * (Kp / 8);
        const int blocks8 = (int)((total8 + 255) / 256 > 16384 ? 16384 : (total8 + 255) / 256);
        if (f32) hipLaunchKernelGGL(im2col8_u8_kernel<float>, dim3(blocks8), dim3(256), 0, s, img, out, B, R, P, Kp, nm);
        else hipLaunchKernelGGL(im2col8_u8_kernel<bf16>, dim3(blocks8), dim3(256), 0, s, img, out, B, R, P, Kp, nm);
        LAUNCH_OK("im2col8_u8_kernel");
        return 0;
    }
    const size_t total = (size_t)B * (R / P) * (R / P) * (Kp / 2);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    if (f32) hipLaunchKernelGGL(im2col_u8_kernel<float>, dim3(blocks), dim3(256), 0, s, img, out, B, R, P, Kp, nm);
    else hipLaunchKernelGGL(im2col_u8_kernel<bf16>, dim3(blocks), dim3(256), 0, s, img, out, B, R, P, Kp, nm);
    LAUNCH_OK("im2col_u8_kernel");
    return 0;
}
int pevit_launch_conv_weight(const float* w, bf16* out, int E, int K, int Kp, hipStream_t s, int f32) {
    const dim3 grid((unsigned)(((size_t)E * Kp + 255) / 256));
    if (f32) hipLaunchKernelGGL(conv_weight_kernel<float>, grid, dim3(256), 0, s, w, out, E, K, Kp);
    else hipLaunchKernelGGL(conv_weight_kernel<bf16>, grid, dim3(256), 0, s, w, out, E, K, Kp);
    LAUNCH_OK("conv_weight_kernel");
    return 0;
}
int pevit_launch_cls_row(const float* cls, const float* pos, float* x, int B, int N, int E, hipStream_t s) {
    hipLaunchKernelGGL(cls_row_kernel, dim3(B), dim3(256), 0, s, cls, pos, x, B, N, E);
    LAUNCH_OK("cls_row_kernel");
    return 0;
}
int pevit_launch_head(const float* feat, const int64_t* labels, const float* W, const float* bias, float* gW, float* gb,
                      float* running_mean, float* running_var, int training, float* ybn, float* rstd, float* logits,
                      float* dlogits, float* dybn, float* loss, float* dfeat, int B, int D, int Cc, hipStream_t s, bf16* dfeat_b16) {
    hipLaunchKernelGGL(bn_fwd_kernel, dim3(ceil_div(D, 16)), dim3(256), 0, s, feat, ybn, rstd, running_mean, running_var,
                       training, B, D);
    LAUNCH_OK("bn_fwd_kernel");
    {   // logits[b][c] = ybn[b] . W[c] + bias[c]
        SmallGemm g{ybn, D, 1, W, 1, D, logits, Cc, bias, nullptr, B, Cc, D, 0};
        hipLaunchKernelGGL(small_gemm_kernel, dim3(ceil_div(Cc, 32), ceil_div(B, 32)), dim3(256), 0, s, g);
        LAUNCH_OK("small_gemm_kernel");
    }
    if (!labels) return 0;
    float* rowloss = dybn;      // dybn is written later (by the dgrad product); reuse its head as scratch
    if (B <= 2048) {        // one workgroup: cross entropy of every row + the mean (bit-identical to the two kernels below)
        hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(1024), 0, s, logits, labels, dlogits, rowloss, loss, B, Cc);
        LAUNCH_OK("ce_loss_kernel");
    } else {
        hipLaunchKernelGGL(ce_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, s, logits, labels, dlogits, rowloss, B, Cc);
        LAUNCH_OK("ce_kernel");
        hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(64), 0, s, rowloss, labels, loss, B);
        LAUNCH_OK("loss_mean_kernel");
    }
    if (gW) {   // gW[c][d] += sum_b dl[b][c] ybn[b][d] ; gb[c] += sum_b dl[b][c]
        SmallGemm g{dlogits, 1, Cc, ybn, D, 1, gW, D, nullptr, gb, Cc, D, B, 1};
        hipLaunchKernelGGL(small_gemm_kernel, dim3(ceil_div(D, 32), ceil_div(Cc, 32)), dim3(256), 0, s, g);
        LAUNCH_OK("small_gemm_kernel");
    }
    if (dfeat) {   // dybn[b][d] = sum_c dl[b][c] W[c][d]
        SmallGemm g{dlogits, Cc, 1, W, D, 1, dybn, D, nullptr, nullptr, B, D, Cc, 0};
        hipLaunchKernelGGL(small_gemm_kernel, dim3(ceil_div(D, 32), ceil_div(B, 32)), dim3(256), 0, s, g);
        LAUNCH_OK("small_gemm_kernel");
        hipLaunchKernelGGL(bn_bwd_kernel, dim3(ceil_div(D, 16)), dim3(256), 0, s, dybn, ybn, rstd, dfeat, training, B, D, dfeat_b16);
        LAUNCH_OK("bn_bwd_kernel");
    }
    return 0;
}
