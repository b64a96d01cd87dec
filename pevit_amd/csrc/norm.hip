// LayerNorm forward / backward for the frozen ln_1 / ln_2 / ln_pre / ln_post of the CLIP
// vision tower (reference: model.py:154-160 -- statistics in f32, eps = 1e-5).
// One 64-lane wavefront per row; the row lives in registers (E <= 1024), reductions are
// wave shuffles, every global access is a 16-byte-per-lane coalesced segment.  These kernels
// are pure HBM streaming: ~ (4+2) B/elem forward, (4+4+4+4) B/elem backward.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAXV = 4;   // float4 per lane -> E <= 1024

template <typename ST>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int rows, int E, size_t xstride,
                                                     bf16* __restrict__ yb, float* __restrict__ yf,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     unsigned char* __restrict__ y8) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * xstride;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < E) {
            v[i] = *reinterpret_cast<const float4*>(xr + c);
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < E) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + cc * cc + d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)E + 1e-5f);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < E) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            const float4 b = *reinterpret_cast<const float4*>(beta + c);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            if (yb) st_store4<ST>(yb, (size_t)row * E + c, o.x, o.y, o.z, o.w);
            if (yf) *reinterpret_cast<float4*>(yf + (size_t)row * E + c) = o;
            if (y8) {       // e4m3 copy for the fp8 x fp8 products: 4 consecutive channels stay contiguous under fp8_kperm
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(o.x, -448.f), 448.f), fminf(fmaxf(o.y, -448.f), 448.f), 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(o.z, -448.f), 448.f), fminf(fmaxf(o.w, -448.f), 448.f), w, true);
                *reinterpret_cast<int*>(y8 + (size_t)row * E + fp8_kperm(c)) = w;
            }
        }
    }
}

// dx = dres + rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat))
// DYT: storage of the upstream gradient dy -- f32, or the activation storage type ST when it comes straight out of a
// dX GEMM (bf16 in production: one rounding, half the bytes on both sides of this HBM-bound kernel)
template <typename ST, typename DYT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_, const float* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const float* dres,
                                                     float* dx, bf16* __restrict__ dx_bf16, int rows, int E, size_t xstride,
                                                     const float* __restrict__ bscale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float mean = mean_in[row], rstd = rstd_in[row];
    const size_t base = (size_t)row * E;          // dy rows are always compact
    const size_t xb = (size_t)row * xstride;      // x, dres, dx, dx_bf16 share the row stride
    float4 gd[MAXV], xh[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < E) {
            float4 d;
            if constexpr (sizeof(DYT) == 2) {
                const bf16x4 d4 = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(dy_) + base + c);
                d = make_float4(bf2f(d4[0]), bf2f(d4[1]), bf2f(d4[2]), bf2f(d4[3]));
            } else {
                d = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + base + c);
            }
            const float4 xv = *reinterpret_cast<const float4*>(x + xb + c);
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            gd[i] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
            xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            s1 += gd[i].x + gd[i].y + gd[i].z + gd[i].w;
            s2 += gd[i].x * xh[i].x + gd[i].y * xh[i].y + gd[i].z * xh[i].z + gd[i].w * xh[i].w;
        }
    }
    const float m1 = wave_sum(s1) / (float)E;
    const float m2 = wave_sum(s2) / (float)E;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < E) {
            float4 o;
            o.x = rstd * (gd[i].x - m1 - xh[i].x * m2);
            o.y = rstd * (gd[i].y - m1 - xh[i].y * m2);
            o.z = rstd * (gd[i].z - m1 - xh[i].z * m2);
            o.w = rstd * (gd[i].w - m1 - xh[i].w * m2);
            if (dres) {
                const float4 r = *reinterpret_cast<const float4*>(dres + xb + c);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            *reinterpret_cast<float4*>(dx + xb + c) = o;
            if (dx_bf16) {
                // fp8 weights: the consuming GEMM contracts over these columns; their power-of-two channel
                // scales are folded into its bf16 operand here (exact), the f32 stream stays unscaled
                if (bscale) {
                    const float4 sc = *reinterpret_cast<const float4*>(bscale + c);
                    o.x *= sc.x; o.y *= sc.y; o.z *= sc.z; o.w *= sc.w;
                }
                st_store4<ST>(dx_bf16, xb + c, o.x, o.y, o.z, o.w);
            }
        }
    }
}

}  // namespace

int pevit_launch_ln_fwd(const float* x, const float* gamma, const float* beta, int rows, int E, bf16* y_bf16,
                        float* y_f32, float* mean, float* rstd, hipStream_t s, size_t xstride, int f32, unsigned char* y_fp8) {
    if (xstride == 0) xstride = (size_t)E;
    if (E % 4 != 0 || E > 256 * MAXV) { pevit_set_error("ln_fwd: unsupported width %d", E); return -1; }
    if (y_fp8 && E % 128 != 0) { pevit_set_error("ln_fwd: the fp8 copy needs a width that is a multiple of 128"); return -1; }
    if (rows <= 0) return 0;
    if (f32) hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, x, gamma, beta, rows, E, xstride,
                                y_bf16, y_f32, mean, rstd, y_fp8);
    else hipLaunchKernelGGL(ln_fwd_kernel<bf16>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, x, gamma, beta, rows, E, xstride,
                            y_bf16, y_f32, mean, rstd, y_fp8);
    LAUNCH_OK("ln_fwd_kernel");
    return 0;
}

int pevit_launch_ln_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, float* dx_out, bf16* dx_bf16, int rows, int E, hipStream_t s,
                        size_t xstride, const float* bf16_colscale, int f32, int dy_stored) {
    if (xstride == 0) xstride = (size_t)E;
    if (E % 4 != 0 || E > 256 * MAXV) { pevit_set_error("ln_bwd: unsupported width %d", E); return -1; }
    if (rows <= 0) return 0;
    const dim3 grid(ceil_div(rows, 4));
    if (f32) hipLaunchKernelGGL((ln_bwd_kernel<float, float>), grid, dim3(256), 0, s, dy, x, mean, rstd, gamma, dres,
                                dx_out, dx_bf16, rows, E, xstride, bf16_colscale);
    else if (dy_stored) hipLaunchKernelGGL((ln_bwd_kernel<bf16, bf16>), grid, dim3(256), 0, s, dy, x, mean, rstd, gamma, dres,
                                           dx_out, dx_bf16, rows, E, xstride, bf16_colscale);
    else hipLaunchKernelGGL((ln_bwd_kernel<bf16, float>), grid, dim3(256), 0, s, dy, x, mean, rstd, gamma, dres,
                            dx_out, dx_bf16, rows, E, xstride, bf16_colscale);
    LAUNCH_OK("ln_bwd_kernel");
    return 0;
}
