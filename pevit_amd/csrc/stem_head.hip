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
__global__ void im2col_kernel(const float* __restrict__ img, bf16* __restrict__ out, int B, int R, int P, int Kp) {
    const int G = R / P, G2 = G * G, K = 3 * P * P;
    const size_t total = (size_t)B * G2 * (Kp / 2);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k2 = (int)(idx % (Kp / 2));
        const size_t row = idx / (Kp / 2);
        const int b = (int)(row / G2), gidx = (int)(row - (size_t)b * G2);
        const int gy = gidx / G, gx = gidx - gy * G;
        bf16x2 o;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = 2 * k2 + u;
            float v = 0.f;
            if (k < K) {
                const int c = k / (P * P), rem = k - c * P * P;
                const int i = rem / P, j = rem - i * P;
                v = img[(((size_t)b * 3 + c) * R + gy * P + i) * R + gx * P + j];
            }
            o[u] = f2bf(v);
        }
        *reinterpret_cast<bf16x2*>(out + row * Kp + 2 * k2) = o;
    }
}

// conv1.weight (E, 3, P, P) f32 -> [E][Kp] bf16 zero padded
__global__ void conv_weight_kernel(const float* __restrict__ w, bf16* __restrict__ out, int E, int K, int Kp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)E * Kp) return;
    const int e = (int)(idx / Kp), k = (int)(idx - (size_t)e * Kp);
    out[idx] = f2bf(k < K ? w[(size_t)e * K + k] : 0.f);
}

// x[b*N + 0][:] = class_embedding + positional_embedding[0]
__global__ void cls_row_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x,
                               int B, int N, int E) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < E; e += blockDim.x) x[(size_t)b * N * E + e] = cls[e] + pos[e];
}

// ---- head -------------------------------------------------------------------------
// one thread per feature d: batch statistics (training) or running statistics (eval)
__global__ void bn_fwd_kernel(const float* __restrict__ feat, float* __restrict__ y, float* __restrict__ rstd_out,
                              float* running_mean, float* running_var, int training, int B, int D) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    float mean, var;
    if (training) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += feat[(size_t)b * D + d];
        mean = s / (float)B;
        float q = 0.f;
        for (int b = 0; b < B; ++b) { const float t = feat[(size_t)b * D + d] - mean; q += t * t; }
        var = q / (float)B;                                   // biased: used for normalisation
        const float unbiased = B > 1 ? q / (float)(B - 1) : var;
        running_mean[d] = 0.9f * running_mean[d] + 0.1f * mean;
        running_var[d] = 0.9f * running_var[d] + 0.1f * unbiased;
    } else {
        mean = running_mean[d]; var = running_var[d];
    }
    const float rstd = rsqrtf(var + 1e-5f);
    rstd_out[d] = rstd;
    for (int b = 0; b < B; ++b) y[(size_t)b * D + d] = (feat[(size_t)b * D + d] - mean) * rstd;
}

// logits[b][c] = y[b] . W[c] + bias[c]
__global__ void linear_fwd_kernel(const float* __restrict__ y, const float* __restrict__ W, const float* __restrict__ bias,
                                  float* __restrict__ logits, int B, int D, int Cc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Cc) return;
    const int b = idx / Cc, c = idx - b * Cc;
    const float* yr = y + (size_t)b * D; const float* wr = W + (size_t)c * D;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc = fmaf(yr[d], wr[d], acc);
    logits[idx] = acc + bias[c];
}

// single block: per-row log-softmax, mean loss, dlogits = (softmax - onehot)/B
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                 float* __restrict__ dlogits, float* __restrict__ loss, int B, int Cc) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc = 0.f;
    for (int b = wid; b < B; b += 4) {
        const float* lr = logits + (size_t)b * Cc;
        float m = -3.0e38f;
        for (int c = lane; c < Cc; c += 64) m = fmaxf(m, lr[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < Cc; c += 64) s += __expf(lr[c] - m);
        s = wave_sum(s);
        const float lse = m + __logf(s);
        const int lab = (int)labels[b];
        for (int c = lane; c < Cc; c += 64) {
            const float p = __expf(lr[c] - lse);
            dlogits[(size_t)b * Cc + c] = (p - (c == lab ? 1.f : 0.f)) / (float)B;
        }
        if (lane == 0) acc += lse - lr[lab];
    }
    if (lane == 0) part[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (part[0] + part[1] + part[2] + part[3]) / (float)B;
}

// gW[c][d] += sum_b dl[b][c] y[b][d] ; gb[c] += sum_b dl[b][c]
__global__ void linear_wgrad_kernel(const float* __restrict__ dl, const float* __restrict__ y, float* gW, float* gb,
                                    int B, int D, int Cc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < Cc * D) {
        const int c = idx / D, d = idx - c * D;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc = fmaf(dl[(size_t)b * Cc + c], y[(size_t)b * D + d], acc);
        gW[idx] += acc;
    }
    if (idx < Cc) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dl[(size_t)b * Cc + idx];
        gb[idx] += acc;
    }
}

// dy[b][d] = sum_c dl[b][c] W[c][d]
__global__ void linear_dgrad_kernel(const float* __restrict__ dl, const float* __restrict__ W, float* __restrict__ dy,
                                    int B, int D, int Cc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, d = idx - b * D;
    float acc = 0.f;
    for (int c = 0; c < Cc; ++c) acc = fmaf(dl[(size_t)b * Cc + c], W[(size_t)c * D + d], acc);
    dy[idx] = acc;
}

// dfeat = rstd * (dy - mean_b(dy) - yhat * mean_b(dy*yhat))   (training) ;  rstd * dy (eval)
__global__ void bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yhat, const float* __restrict__ rstd,
                              float* __restrict__ dfeat, int training, int B, int D) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const float r = rstd[d];
    float m1 = 0.f, m2 = 0.f;
    if (training) {
        for (int b = 0; b < B; ++b) { const float g = dy[(size_t)b * D + d]; m1 += g; m2 += g * yhat[(size_t)b * D + d]; }
        m1 /= (float)B; m2 /= (float)B;
    }
    for (int b = 0; b < B; ++b) {
        const float g = dy[(size_t)b * D + d];
        dfeat[(size_t)b * D + d] = training ? r * (g - m1 - yhat[(size_t)b * D + d] * m2) : r * g;
    }
}

}  // namespace

int pevit_launch_im2col(const float* img, bf16* out, int B, int R, int P, int Kp, hipStream_t s) {
    const size_t total = (size_t)B * (R / P) * (R / P) * (Kp / 2);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(im2col_kernel, dim3(blocks), dim3(256), 0, s, img, out, B, R, P, Kp);
    return 0;
}
int pevit_launch_conv_weight(const float* w, bf16* out, int E, int K, int Kp, hipStream_t s) {
    hipLaunchKernelGGL(conv_weight_kernel, dim3((unsigned)(((size_t)E * Kp + 255) / 256)), dim3(256), 0, s, w, out, E, K, Kp);
    return 0;
}
int pevit_launch_cls_row(const float* cls, const float* pos, float* x, int B, int N, int E, hipStream_t s) {
    hipLaunchKernelGGL(cls_row_kernel, dim3(B), dim3(256), 0, s, cls, pos, x, B, N, E);
    return 0;
}
int pevit_launch_head(const float* feat, const int64_t* labels, const float* W, const float* bias, float* gW, float* gb,
                      float* running_mean, float* running_var, int training, float* ybn, float* rstd, float* logits,
                      float* dlogits, float* dybn, float* loss, float* dfeat, int B, int D, int Cc, hipStream_t s) {
    hipLaunchKernelGGL(bn_fwd_kernel, dim3(ceil_div(D, 64)), dim3(64), 0, s, feat, ybn, rstd, running_mean, running_var,
                       training, B, D);
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(ceil_div(B * Cc, 256)), dim3(256), 0, s, ybn, W, bias, logits, B, D, Cc);
    if (!labels) return 0;
    hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(256), 0, s, logits, labels, dlogits, loss, B, Cc);
    if (gW) hipLaunchKernelGGL(linear_wgrad_kernel, dim3(ceil_div(Cc * D, 256)), dim3(256), 0, s, dlogits, ybn, gW, gb, B, D, Cc);
    if (dfeat) {
        hipLaunchKernelGGL(linear_dgrad_kernel, dim3(ceil_div(B * D, 256)), dim3(256), 0, s, dlogits, W, dybn, B, D, Cc);
        hipLaunchKernelGGL(bn_bwd_kernel, dim3(ceil_div(D, 64)), dim3(64), 0, s, dybn, ybn, rstd, dfeat, training, B, D);
    }
    return 0;
}
