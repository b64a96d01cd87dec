// LayerNorm forward / backward for the frozen ln_1 / ln_2 / ln_pre / ln_post of the CLIP
// vision tower (reference: model.py:154-160 -- statistics in f32, eps = 1e-5).
// One 64-lane wavefront per row; the row lives in registers (E <= 1024), reductions are
// wave shuffles, every global access is a 16-byte-per-lane coalesced segment.  These kernels
// are pure HBM streaming: ~ (4+2) B/elem forward, (4+4+4+4) B/elem backward.
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAXV = 4;   // float4 per lane -> E <= 1024

// NV = E / 256 when E is a multiple of 256 (round 5: no column predicates, no masked fourth group at E = 768), 0 = any E <= 1024
template <typename ST, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int rows, int E, size_t xstride,
                                                     bf16* __restrict__ yb, float* __restrict__ yf,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     unsigned char* __restrict__ y8) {
    constexpr int NVV = NV > 0 ? NV : MAXV;
    if constexpr (NV > 0) E = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * xstride;
    // Every load of the row -- x, gamma, beta -- is requested before the first reduction, and none of them sits inside a bounds
    // branch (columns beyond E read column 0 and are masked): a value loaded inside a branch makes hipcc wait with vmcnt(0) at its
    // first use after the join, and on gfx950 that also waits for the STORES issued so far -- the store loop below then paid one
    // store latency per 256 columns (profiles/NOTES_gemm.md (r03_gemm_experiments) section 3 has the same finding for the GEMM epilogues).
    float4 v[NVV], gm[NVV], bt[NVV];
    bool ok[NVV];
#pragma unroll
    for (int i = 0; i < NVV; ++i) {
        const int c = lane * 4 + i * 256;
        ok[i] = NV > 0 || c < E;
        const int cc = ok[i] ? c : 0;
        v[i] = *reinterpret_cast<const float4*>(xr + cc);
        gm[i] = *reinterpret_cast<const float4*>(gamma + cc);
        bt[i] = *reinterpret_cast<const float4*>(beta + cc);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NVV; ++i) s += ok[i] ? v[i].x + v[i].y + v[i].z + v[i].w : 0.f;
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NVV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += ok[i] ? a * a + b * b + cc * cc + d * d : 0.f;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)E + 1e-5f);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), visible to the compiler: every load is in before the first (conditional) store
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NVV; ++i) {
        const int c = lane * 4 + i * 256;
        if (ok[i]) {
            const float4 g = gm[i], b = bt[i];
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
// RES16 (round 5): the residual gradient arrives in the activation storage type (the bf16 copy of the gradient stream that the
// dX GEMMs read anyway) instead of f32, and dx may be null -- the stream is then carried in bf16 only, read-modify-write in place
// (dres == dx_bf16: every lane has all its loads in registers before its first store): 10 instead of 16 bytes per element.
// SCL: power-of-two channel scales may be present (fp8 weights: bscale on the outgoing bf16 copy, rscale on the incoming bf16 residual);
// false: neither is, and their loads are not issued (they used to re-read gamma as a stand-in: 6 of a lane's 18 requests at E = 768)
template <typename ST, typename DYT, bool RES16, int NV, bool SCL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_, const float* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const float* dres,
                                                     float* dx, bf16* dx_bf16, int rows, int E, size_t xstride,
                                                     const float* __restrict__ bscale, int res_period,
                                                     const float* __restrict__ rscale) {
    constexpr int NVV = NV > 0 ? NV : MAXV;
    if constexpr (NV > 0) E = 256 * NV;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float mean = mean_in[row], rstd = rstd_in[row];
    const size_t base = (size_t)row * E;          // dy rows are always compact
    const size_t xb = (size_t)row * xstride;      // x, dres, dx, dx_bf16 share the row stride
    // All loads -- dy, x, gamma, the residual gradient, the fp8 channel scales -- are requested up front and none sits inside a
    // bounds branch (columns beyond E read column 0 and are masked; a missing residual gradient reads x and the value is dropped):
    // with the loads inside `if (c < E)` hipcc waited for each 256-column group before requesting the next (four memory round
    // trips per row), and the residual gradient, loaded between the stores, cost a fifth plus one store latency per group.
    float4 gd[NVV], xh[NVV], rs[NVV], sc[NVV], xv[NVV], gv[NVV];
    typename std::conditional<sizeof(DYT) == 2, bf16x4, float4>::type dv[NVV];
    bf16x4 rs16[NVV];
    float4 ru[NVV];        // RES16 with fp8 weights: the power-of-two channel scales folded into the incoming bf16 stream (taken out again, exactly)
    const float* usrc = (RES16 && SCL && rscale) ? rscale : gamma;
    bool ok[NVV];
    // res_period > 0: the residual gradient is zero except on rows that are multiples of res_period (the class-token rows of the
    // last block, whose upstream gradient exists on those rows only): the other rows read nothing of it (row 0 stands in, dropped)
    const float* rsrc = (dres && !RES16) ? dres : x;
    const bf16* rsrc16 = reinterpret_cast<const bf16*>(dres);
    const bool has_res = dres != nullptr && (res_period <= 0 || row % res_period == 0);
    const size_t rb = has_res ? xb : 0;
    const bool scaled = SCL && dx_bf16 && bscale;
    const float* ssrc = scaled ? bscale : gamma;
#pragma unroll
    for (int i = 0; i < NVV; ++i) {
        const int c = lane * 4 + i * 256;
        ok[i] = NV > 0 || c < E;
        const int cc = ok[i] ? c : 0;
        if constexpr (sizeof(DYT) == 2) dv[i] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(dy_) + base + cc);
        else dv[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + base + cc);
        xv[i] = *reinterpret_cast<const float4*>(x + xb + cc);
        gv[i] = *reinterpret_cast<const float4*>(gamma + cc);
        if constexpr (RES16) { rs16[i] = *reinterpret_cast<const bf16x4*>(rsrc16 + rb + cc); if constexpr (SCL) ru[i] = *reinterpret_cast<const float4*>(usrc + cc); }
        else rs[i] = *reinterpret_cast<const float4*>(rsrc + rb + cc);
        if constexpr (SCL) sc[i] = *reinterpret_cast<const float4*>(ssrc + cc);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NVV; ++i) {
        float4 d;
        if constexpr (sizeof(DYT) == 2) d = make_float4(bf2f(dv[i][0]), bf2f(dv[i][1]), bf2f(dv[i][2]), bf2f(dv[i][3]));
        else d = dv[i];
        const float4 g = gv[i];
        gd[i] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
        xh[i] = make_float4((xv[i].x - mean) * rstd, (xv[i].y - mean) * rstd, (xv[i].z - mean) * rstd, (xv[i].w - mean) * rstd);
        s1 += ok[i] ? gd[i].x + gd[i].y + gd[i].z + gd[i].w : 0.f;
        s2 += ok[i] ? gd[i].x * xh[i].x + gd[i].y * xh[i].y + gd[i].z * xh[i].z + gd[i].w * xh[i].w : 0.f;
        if constexpr (RES16) {
            rs[i] = make_float4(bf2f(rs16[i][0]), bf2f(rs16[i][1]), bf2f(rs16[i][2]), bf2f(rs16[i][3]));
            if (SCL && rscale) {   // 1 / 2^k, exact: (254 << 23) - bits(2^k)
                rs[i].x *= __int_as_float(0x7F000000 - __float_as_int(ru[i].x)); rs[i].y *= __int_as_float(0x7F000000 - __float_as_int(ru[i].y));
                rs[i].z *= __int_as_float(0x7F000000 - __float_as_int(ru[i].z)); rs[i].w *= __int_as_float(0x7F000000 - __float_as_int(ru[i].w));
            }
        }
        rs[i] = has_res ? rs[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (SCL) sc[i] = scaled ? sc[i] : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    const float m1 = wave_sum(s1) / (float)E;
    const float m2 = wave_sum(s2) / (float)E;
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), visible to the compiler: every load is in before the first (conditional) store
#pragma unroll
    for (int i = 0; i < NVV; ++i) {
        const int c = lane * 4 + i * 256;
        if (ok[i]) {
            float4 o;
            o.x = rstd * (gd[i].x - m1 - xh[i].x * m2);
            o.y = rstd * (gd[i].y - m1 - xh[i].y * m2);
            o.z = rstd * (gd[i].z - m1 - xh[i].z * m2);
            o.w = rstd * (gd[i].w - m1 - xh[i].w * m2);
            o.x += rs[i].x; o.y += rs[i].y; o.z += rs[i].z; o.w += rs[i].w;       // zeros without a residual gradient
            if (!RES16 || dx) *reinterpret_cast<float4*>(dx + xb + c) = o;
            if (dx_bf16) {
                // fp8 weights: the consuming GEMM contracts over these columns; their power-of-two channel
                // scales are folded into its bf16 operand here (exact: ones otherwise), the f32 stream stays unscaled
                if constexpr (SCL) { o.x *= sc[i].x; o.y *= sc[i].y; o.z *= sc[i].z; o.w *= sc[i].w; }
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
    // The E / 256 instances are NOT used by the forward kernel: measured on one box (scripts/experiments/gpu_r5_ln.sh) the generic
    // predicated form is the faster one here (7.2 against 7.5 us at E = 768, 10.3 against 10.9 at E = 1024) while the backward kernel
    // gains 14 % from them (11.0 -> 9.5 us); LN_FWD_NV builds them for A/B runs.
#ifdef LN_FWD_NV_ON
    const int nv = E % 256 == 0 ? E / 256 : 0;
#else
    const int nv = 0;
#endif
#define LN_FWD_GO(ST, N) hipLaunchKernelGGL((ln_fwd_kernel<ST, N>), dim3(ceil_div(rows, 4)), dim3(256), 0, s, x, gamma, beta, rows, E, xstride, y_bf16, y_f32, mean, rstd, y_fp8)
#define LN_FWD_NV(ST) switch (nv) { case 1: LN_FWD_GO(ST, 1); break; case 2: LN_FWD_GO(ST, 2); break; case 3: LN_FWD_GO(ST, 3); break; \
                                    case 4: LN_FWD_GO(ST, 4); break; default: LN_FWD_GO(ST, 0); }
    if (f32) { LN_FWD_NV(float) } else { LN_FWD_NV(bf16) }
#undef LN_FWD_NV
#undef LN_FWD_GO
    LAUNCH_OK("ln_fwd_kernel");
    return 0;
}

int pevit_launch_ln_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, float* dx_out, bf16* dx_bf16, int rows, int E, hipStream_t s,
                        size_t xstride, const float* bf16_colscale, int f32, int dy_stored, int res_period, int res16,
                        const float* res_colscale) {
    if (xstride == 0) xstride = (size_t)E;
    if (E % 4 != 0 || E > 256 * MAXV) { pevit_set_error("ln_bwd: unsupported width %d", E); return -1; }
    if (rows <= 0) return 0;
    const dim3 grid(ceil_div(rows, 4));
    if (res16 && (f32 || !dy_stored || !dx_bf16)) { pevit_set_error("ln_bwd: the bf16 residual form needs bf16 storage on both sides"); return -1; }
    if (res16 && !dres) { pevit_set_error("ln_bwd: the bf16 residual form reads its residual gradient from dres (null)"); return -1; }
#ifdef LN_NV_OFF
    const int nv = 0;                          // (A/B builds: the generic kernels)
#else
    const int nv = E % 256 == 0 ? E / 256 : 0;
#endif
    const bool scl = bf16_colscale != nullptr || res_colscale != nullptr;
#define LN_BWD_GO(ST, DYT, R, N) do { if (scl) hipLaunchKernelGGL((ln_bwd_kernel<ST, DYT, R, N, true>), grid, dim3(256), 0, s, dy, x, mean, rstd, gamma, dres, \
                                                    dx_out, dx_bf16, rows, E, xstride, bf16_colscale, res_period, res_colscale); \
                                      else hipLaunchKernelGGL((ln_bwd_kernel<ST, DYT, R, N, false>), grid, dim3(256), 0, s, dy, x, mean, rstd, gamma, dres, \
                                                    dx_out, dx_bf16, rows, E, xstride, bf16_colscale, res_period, res_colscale); } while (0)
#define LN_BWD_NV(ST, DYT, R) switch (nv) { case 1: LN_BWD_GO(ST, DYT, R, 1); break; case 2: LN_BWD_GO(ST, DYT, R, 2); break; \
                                            case 3: LN_BWD_GO(ST, DYT, R, 3); break; case 4: LN_BWD_GO(ST, DYT, R, 4); break; default: LN_BWD_GO(ST, DYT, R, 0); }
    if (res16) { LN_BWD_NV(bf16, bf16, true) }
    else if (f32) { LN_BWD_NV(float, float, false) }
    else if (dy_stored) { LN_BWD_NV(bf16, bf16, false) }
    else { LN_BWD_NV(bf16, float, false) }
#undef LN_BWD_NV
#undef LN_BWD_GO
    LAUNCH_OK("ln_bwd_kernel");
    return 0;
}
