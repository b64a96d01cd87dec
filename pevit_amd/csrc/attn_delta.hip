// Attention-site adapters fused with the attention core (KAdaptation / LoRA, N <= 64 tokens per image).
//
// The reference adds the adapter delta to q and v through a RAW reshape (model.py:796-799): element
// i = rr*E + e of the (N,B,E)-contiguous delta lands on flat element i of the head-layout buffer (B*H, N, 64).  In units
// of 64-element head rows: head row c = bh*N + n receives block j = c % H of reference row rr = c / H.  A run of
// HPW consecutive heads whose first head row c0 = bh0*N is a multiple of H therefore owns the COMPLETE reference rows
// [c0/H, (c0 + HPW*N)/H): with N = 50, H = 12 six heads are exactly 25 reference rows.  One workgroup per such run:
//
//   forward  (attn_fwd_delta_kernel):  q, k, v of the run -> LDS (one round trip), the delta
//            ascale * t[row(rr)] . Q[e] + b[e]  of its reference rows on the matrix core (Q as its bf16 panel, t split hi + lo, the same
//            products in the same order as lowrank.hip delta_add_kernel) added to q and v IN LDS, q' and v' written back
//            for the backward pass, softmax(q' k^T) v' per head from LDS.  Replaces delta_add + attn_fwd: q and v are not
//            read-modified-written through HBM (39 MB per layer at B = 128) and one dispatch per layer disappears.
//
// Results are bit-identical to the two-kernel path (tests/test_gpu_ops2.py): same roundings (q' = bf16(q + delta)), same
// MFMA order per 16-query tile.
#include "common.h"
#include "kernels.h"

#ifndef AFD_LDR
#define AFD_LDR 72          // LDS row stride (elements): 36 dwords, 16-byte aligned rows
#endif

namespace {

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 ldsfrag(const bf16* base, int row, int s, int g) {
    return *reinterpret_cast<const bf16x8*>(base + row * AFD_LDR + 32 * s + 8 * g);
}
// lowrank.hip split_bf16v: f32 -> bf16 hi + bf16 lo
__device__ __forceinline__ void split8(const float4 a, const float4 b, bf16x8& hi, bf16x8& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = f2bf(v[i]);
        lo[i] = f2bf(v[i] - bf2f(hi[i]));
    }
}
// attention.hip tfrag_tr with this file's row stride
__device__ __forceinline__ bf16x8 tfrag(const bf16* Ys, int dt, int s, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int m = lane & 15, g = lane >> 4;
    const bf16* src = Ys + (32 * s + 4 * g + (m >> 2)) * AFD_LDR + 16 * (m & 3) + 4 * dt;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(src));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(src + 16 * AFD_LDR));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
}
__device__ __forceinline__ void store16o(bf16* dst, const f32x4 o[4], float scale) {
    bf16x8 a, b;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[dt * 4 + r] = f2bf(o[dt][r] * scale);
            b[dt * 4 + r] = f2bf(o[dt + 2][r] * scale);
        }
    store_bf16x8(dst, a);
    store_bf16x8(dst + 8, b);
}

// HPW heads per workgroup, NW waves, PER 32-column delta steps per wave (E / 32 <= PER * NW / 2).  LDS: q, k, v of the run as [(HPW-1)*N + max(N + 16, 64)][AFD_LDR] bf16 each (the rows behind the last
// head are zero: the 64-row key / value tiles of a head run over into the next head's rows, which are finite and meet
// probabilities that are exactly 0).
// HC, NC: heads and tokens per image as compile-time constants (0 = the runtime arguments).  Round 5: 20.5 -> 19.7 us at H = 12,
// N = 50 (scripts/experiments/gpu_r5_fixn.sh) -- bounds, row strides and the head / row arithmetic fold.
template <int HPW, int NW, int PER, int HC, int NC>
__global__ __launch_bounds__(64 * NW) void attn_fwd_delta_kernel(bf16* __restrict__ q, const bf16* __restrict__ k, bf16* __restrict__ v,
                                                                 const float* __restrict__ t, const bf16* __restrict__ q16,
                                                                 const float* __restrict__ bias, float ascale,
                                                                 bf16* __restrict__ out, int ldo, float* __restrict__ lse,
                                                                 int B, int H_rt, int N_rt, unsigned long long* __restrict__ tl) {
    const int N = NC ? NC : N_rt, H = HC ? HC : H_rt;
    constexpr int NT = 64 * NW, LDR = AFD_LDR;
    // measurement only (pevit_debug_timeline): s_memtime of wave 0 of every workgroup at the phase boundaries
    auto stamp = [&](int i) { if (tl && threadIdx.x == 0) tl[(size_t)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int E = H * 64, T = B * N, BH = B * H;
    const int bh0 = blockIdx.x * HPW;
    const int nh = min(HPW, BH - bh0);
    const int rows = nh * N;                              // head rows of this run
    const int prow = (HPW - 1) * N + (N + 16 > 64 ? N + 16 : 64);   // LDS rows per tensor: the last head's 64-row tile stays inside
    bf16* Qs = reinterpret_cast<bf16*>(smem);
    bf16* Ks = Qs + prow * LDR;
    bf16* Vs = Ks + prow * LDR;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const size_t goff = (size_t)bh0 * N * 64;             // the run is contiguous in the head-layout buffers
    const long long c0 = (long long)bh0 * N;              // first head row of the run (global index)
    const int rr0 = (int)(c0 / H), jofs = (int)(c0 - (long long)rr0 * H);   // first reference row, block offset inside it

    // ---- every global request of the first phase up front (clamped addresses, no load inside a bounds branch) ----
    constexpr int PIECES = (HPW * 64 * 8 + NT - 1) / NT;  // 16-byte pieces per thread and tensor (N <= 64)
    // Request order = the order of use: q, v and the delta operands first, k LAST -- the delta phase does not touch k, whose 38 KB
    // (of this workgroup's 219 KB) may still be on their way while q and v are staged and the delta runs; k goes to LDS behind it.
    bf16x8 rq[PIECES], rk[PIECES], rv[PIECES];
    const int npc = rows * 8;
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
        const int idx = min(tid + NT * it, npc - 1);
        rq[it] = load_bf16x8(q + goff + (size_t)idx * 8);
        rv[it] = load_bf16x8(v + goff + (size_t)idx * 8);
    }
    // delta operands.  The first NW/2 waves work on q, the others on v; a wave owns the 32-column steps st = wq, wq + NW/2, ...
    // (PER of them) and both 16-row groups of reference rows.  Per step: the Q rows of the two interleaved 16x16 tiles (A
    // operand of the MFMA; tile 0: e = eb + 8*(m>>2) + (m&3), tile 1: + 4); per wave: the t rows (B operand, column
    // rr = rr0 + 16k + c16).  ALL of them are requested here, beside the q / k / v requests: one L2 round trip under the HBM one
    // (requested step by step inside the loop, the four dependent round trips were 8 of this kernel's 24 us).
    constexpr int HW = NW / 2;
    const int steps = E / 32;                             // per tensor (<= PER * HW)
    const int which = wid >= HW ? 1 : 0, wq = wid - which * HW;
    const float* bsrc = bias ? bias : reinterpret_cast<const float*>(q16);   // a valid address of >= E floats either way; the value is dropped without a bias
    bf16x8 qa[PER][2];
    float4 ta[2][2];
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
        const int rr = min(rr0 + 16 * kq + c16, T - 1);
        const int n = rr / B, b = rr - n * B;
        const float* src = t + ((size_t)b * N + n) * 64 + which * 32 + 8 * g;
        ta[kq][0] = *reinterpret_cast<const float4*>(src); ta[kq][1] = *reinterpret_cast<const float4*>(src + 4);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int st = min(wq + HW * i, steps - 1);
        const int e_t0 = st * 32 + 8 * (c16 >> 2) + (c16 & 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) qa[i][h] = load_bf16x8(q16 + (size_t)(e_t0 + 4 * h) * 64 + which * 32 + 8 * g);
    }
    // the bias of this lane's 8 columns of every step: with the other requests (behind the staging it was one more L2 round trip
    // in front of the delta phase)
    float4 ba[PER][2];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int st = min(wq + HW * i, steps - 1);
        ba[i][0] = *reinterpret_cast<const float4*>(bsrc + st * 32 + 8 * g);
        ba[i][1] = *reinterpret_cast<const float4*>(bsrc + st * 32 + 8 * g + 4);
    }
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
        const int idx = min(tid + NT * it, npc - 1);
        rk[it] = load_bf16x8(k + goff + (size_t)idx * 8);
    }
    if (tl) { __builtin_amdgcn_s_waitcnt(0x0F70); stamp(1); }       // every request has landed
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
        const int idx = tid + NT * it;
        if (idx < prow * 8) {
            const int y = idx >> 3, c = idx & 7;
            const bool live = idx < npc;
            *reinterpret_cast<bf16x8*>(Qs + y * LDR + 8 * c) = live ? rq[it] : zero_bf16x8();
            *reinterpret_cast<bf16x8*>(Vs + y * LDR + 8 * c) = live ? rv[it] : zero_bf16x8();
        }
    }
    // rows [PIECES*NT/8, prow) (only when the pieces do not cover the zero pad)
    for (int idx = tid + NT * PIECES; idx < prow * 8; idx += NT) {
        const int y = idx >> 3, c = idx & 7;
        *reinterpret_cast<bf16x8*>(Qs + y * LDR + 8 * c) = zero_bf16x8();
        *reinterpret_cast<bf16x8*>(Ks + y * LDR + 8 * c) = zero_bf16x8();
        *reinterpret_cast<bf16x8*>(Vs + y * LDR + 8 * c) = zero_bf16x8();
    }
    __syncthreads();
    stamp(2);

    // ---- delta: q_s[c][d] = bf16(q_s + ascale * t[row(rr)] . Q[e] + b[e]),  c = H*(rr - rr0) + e/64 - jofs, d = e % 64 ----
    {
        bf16* Xs = which ? Vs : Qs;
        bf16x8 th[2], tlo[2];
        split8(ta[0][0], ta[0][1], th[0], tlo[0]);
        split8(ta[1][0], ta[1][1], th[1], tlo[1]);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int st = wq + HW * i;                   // wave-uniform
            if (st < steps) {
                const int eb = st * 32;
                const bf16x8 q0h = qa[i][0], q1h = qa[i][1];
                float bb[8] = {ba[i][0].x, ba[i][0].y, ba[i][0].z, ba[i][0].w, ba[i][1].x, ba[i][1].y, ba[i][1].z, ba[i][1].w};
                if (!bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bb[e] = 0.f;
                }
                const int jblk = (eb + 8 * g) >> 6, d0 = (eb + 8 * g) & 63;
#pragma unroll
                for (int kq = 0; kq < 2; ++kq) {
                    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                    a0 = mfma16(q0h, tlo[kq], a0); a0 = mfma16(q0h, th[kq], a0);      // lowrank.hip delta_add_kernel<bf16>: Q*t_lo + Q*t_hi
                    a1 = mfma16(q1h, tlo[kq], a1); a1 = mfma16(q1h, th[kq], a1);
                    const int rr = rr0 + 16 * kq + c16;
                    const int cl = H * (16 * kq + c16) + jblk - jofs;        // local head row of this lane's 8 elements
                    if (rr < T && cl >= 0 && cl < rows) {
                        bf16* dst = Xs + cl * LDR + d0;
                        const bf16x8 cur = *reinterpret_cast<const bf16x8*>(dst);
                        bf16x8 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o[r] = f2bf((float)cur[r] + ascale * a0[r] + bb[r]);
                            o[4 + r] = f2bf((float)cur[4 + r] + ascale * a1[r] + bb[4 + r]);
                        }
                        *reinterpret_cast<bf16x8*>(dst) = o;
                    }
                }
            }
        }
    }
    stamp(3);
    // ---- k to LDS (requested first thing, used by the attention only) ----
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
        const int idx = tid + NT * it;
        if (idx < prow * 8) {
            const int y = idx >> 3, c = idx & 7;
            *reinterpret_cast<bf16x8*>(Ks + y * LDR + 8 * c) = idx < npc ? rk[it] : zero_bf16x8();
        }
    }
    __syncthreads();
    stamp(4);

    // ---- q' and v' back to HBM (the backward pass recomputes the scores from them); the stores drain under the attention ----
    for (int idx = tid; idx < npc; idx += NT) {
        const int y = idx >> 3, c = idx & 7;
        store_bf16x8(q + goff + (size_t)idx * 8, *reinterpret_cast<const bf16x8*>(Qs + y * LDR + 8 * c));
        store_bf16x8(v + goff + (size_t)idx * 8, *reinterpret_cast<const bf16x8*>(Vs + y * LDR + 8 * c));
    }

    stamp(5);
    // ---- softmax(q' k^T) v' : one (head, 16-query tile) per task, body of attention.hip attn_fwd_kernel<2, .> ----
    const int nxt = (N + 15) >> 4;
    for (int task = wid; task < nh * nxt; task += NW) {
        const int hh = task / nxt, xt = task - hh * nxt;
        const int bh = bh0 + hh, b = bh / H, h = bh - b * H;
        const bf16* Qh = Qs + hh * N * LDR;
        const bf16* Kh = Ks + hh * N * LDR;
        const bf16* Vh = Vs + hh * N * LDR;
        const int xq = 16 * xt + c16;
        const int xs = xq < N ? xq : N - 1;
        const bf16x8 qf0 = ldsfrag(Qh, xs, 0, g), qf1 = ldsfrag(Qh, xs, 1, g);
        f32x4 z[4];
        float m = -3.0e38f;
#pragma unroll
        for (int yt = 0; yt < 4; ++yt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = mfma16(ldsfrag(Kh, 16 * yt + c16, 0, g), qf0, acc);
            acc = mfma16(ldsfrag(Kh, 16 * yt + c16, 1, g), qf1, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * yt + 4 * g + r;
                acc[r] = key < N ? acc[r] : -3.0e38f;
                m = fmaxf(m, acc[r]);
            }
            z[yt] = acc;
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(z[2 * s + half][r] - m);
                    const bf16 pb = f2bf(p);
                    l += bf2f(pb);
                    pf[s][half * 4 + r] = pb;
                }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) o[dt] = mfma16(tfrag(Vh, dt, s, lane), pf[s], o[dt]);
        }
        if (xq < N) {
            store16o(out + ((size_t)b * N + xq) * ldo + h * 64 + 16 * g, o, 1.0f / l);
            if (g == 0) lse[(size_t)bh * N + xq] = m + __logf(l);
        }
    }
    stamp(6);
    if (tl) { __builtin_amdgcn_s_waitcnt(0x0F70); stamp(7); }       // this wave's stores are out
}

unsigned long long* g_timeline = nullptr;
}  // namespace

void pevit_attn_delta_set_timeline(void* buf) { g_timeline = (unsigned long long*)buf; }

// heads per workgroup for which a run of heads owns whole reference rows and fills the chip evenly; 0 = no fused form
int pevit_attn_delta_hpw(int B, int H, int N) {
    if (N > 64 || H > 12) return 0;                       // widths up to 768: four 32-column delta steps per wave
    const int hpw = 6;
    if ((hpw * N) % H) return 0;                          // runs must start on a reference-row boundary (the backward pass relies on it)
    if (hpw * N / H + 2 > 32) return 0;                   // two 16-row groups of reference rows per run
    return hpw;
}

int pevit_launch_attn_fwd_delta(bf16* q, const bf16* k, bf16* v, const float* t, const bf16* q16, const float* bias,
                                float ascale, bf16* out, int ldo, float* lse, int B, int H, int N, hipStream_t s) {
    constexpr int HPW = 6, NW = 12, PER = 4;
    if (pevit_attn_delta_hpw(B, H, N) != HPW) { pevit_set_error("attn_fwd_delta: no fused form for H=%d N=%d", H, N); return -1; }
    if (ldo % 8) { pevit_set_error("attn_fwd_delta: ldo must be a multiple of 8"); return -1; }
    const int bytes = 3 * ((HPW - 1) * N + (N + 16 > 64 ? N + 16 : 64)) * AFD_LDR * 2;
    static_assert(PER * (NW / 2) * 32 >= 768, "delta steps of the widest supported tower");
    if (bytes > 160 * 1024) { pevit_set_error("attn_fwd_delta: %d bytes of LDS", bytes); return -1; }
    static bool attr[2] = {false, false};
    auto go = [&](auto kern, int slot) -> int {
        if (!attr[slot]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                pevit_set_error("attn_fwd_delta: cannot reserve LDS"); return -1;
            }
            attr[slot] = true;
        }
        hipLaunchKernelGGL(kern, dim3(ceil_div(B * H, HPW)), dim3(64 * NW), bytes, s, q, k, v, t, q16, bias, ascale, out, ldo, lse, B, H, N, g_timeline);
        return 0;
    };
    const int rc = (H == 12 && N == 50) ? go(attn_fwd_delta_kernel<HPW, NW, PER, 12, 50>, 1) : go(attn_fwd_delta_kernel<HPW, NW, PER, 0, 0>, 0);
    if (rc) return rc;
    LAUNCH_OK("attn_fwd_delta_kernel");
    return 0;
}
