// Softmax attention core (forward and backward) for one (batch, head) per workgroup.
//
// Reference math: model.py:806-812  (bmm(q,k^T) -> softmax(-1) -> bmm(w,v)), with q already
// scaled by 1/sqrt(64) and q,v already carrying the adapter deltas (model.py:786-799).
// CLIP ViTs have N = 50 / 197 / 257 tokens and head_dim 64, so one head's Q,K,V (<= 33 KB each
// in bf16) fit in a single workgroup's LDS: no online-softmax rescaling is needed, the whole
// score row of a 16-query tile lives in registers.
//
// MFMA formulation (v_mfma_f32_16x16x32_bf16, D[i][j] = sum_k A[i][k] B[k][j]):
//   Z^T[y][x] = sum_d Y[y][d] X[x][d]        A = row fragment of Y, B = row fragment of X
//       -> lane l holds column x = l&15 and rows y = 4*(l>>4)+reg : a whole softmax row
//          (all y for one x) sits in ONE 16-lane column, so row max / sum are register
//          reductions plus two shuffles.
//   O^T[d][x] = sum_y Yt[d][y] W[y][x]       A = fragment of the TRANSPOSED Y (LDS), B = W
//       -> W (P or dS) is fed straight from the Z^T accumulator registers: because the MFMA
//          contraction order is arbitrary, k-slot (j, j+4) of lane group g is defined as
//          y = 32s + 4g + j and 32s + 16 + 4g + j, exactly what the accumulators hold.
//   The rows of O^T are permuted (d = 16*(m>>2) + 4*dt + (m&3)) so that each lane ends up with
//   16 consecutive d of one token: 32-byte stores, 128 contiguous bytes per token.
// Forward:  x = queries, y = keys  : Z = S, W = P, Yt = V^T                      -> O
// Backward pass A: x = queries, y = keys: Z1 = S, Z2 = dP, W = dS, Yt = K^T      -> dQ
// Backward pass B: x = keys, y = queries: Z1 = S, Z2 = dP, W = dS|P, Yt = Q^T|dO^T -> dK, dV
// (S and dP are recomputed in each pass; 14 small MFMAs per 16x16 cell instead of 10, in
// exchange for no cross-wave reductions and no atomics.)
#include "common.h"
#include "kernels.h"

// LDS tiles are row-major [tokens][LD], in one of two layouts chosen per kernel instance (template parameter LD):
//   LD = 64: unpadded 128-byte rows, the 16-byte chunks of a row XOR-swizzled by the row (chunk c of row r sits at c ^ ((r >> 1) & 7));
//   LD = 80: rows padded to 160 bytes, no swizzle (rounds 1-4).
// Round 5 (PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.44-0.62 on every attention kernel; model: scripts/lds_bank_model2.py):
// with padded rows the ds_read_b128 row fragments are conflict free but every ds_read_b64_tr_b16 transposed fragment is 4-way
// conflicted -- its 32 lanes address 8 rows x 4 pieces 32 bytes apart, and 40-dword rows put all of them on 16 of the 64 banks.
// The swizzled layout keeps row fragments and staging writes conflict free and brings the transposed reads to 2-way (the minimum
// for 8-byte pieces in 16-byte-aligned rows); tiles are 20 % smaller.  It costs address arithmetic: a padded address is base +
// immediate, a swizzled one needs an XOR per fragment.  Measured on one box (profiles/r05_experiments.md): LDS-active cycles -31 %
// (backward, N = 257) / -42 % (forward) as modelled, kernel time -6 % / -3 % there, -3 % at N = 50, forward -6 % at N = 197 -- but
// the N = 197 backward (8 waves, two workgroups per CU, VALU-bound) LOSES 12 % (71 -> 80 us; with padding AND swizzle 84 us:
// it is the arithmetic).  So: swizzled everywhere except that one instance.
#ifndef ATT_LD_FWD
#define ATT_LD_FWD 64
#endif
#ifndef ATT_LD_BWD_SMALL
#define ATT_LD_BWD_SMALL 64
#endif
#ifndef ATT_LD_BWD_MID
#define ATT_LD_BWD_MID 80
#endif
#ifndef ATT_LD_BWD_BIG
#define ATT_LD_BWD_BIG 64
#endif

// Round 5 measured the token count N as a compile-time constant of these kernels (scripts/experiments/gpu_r5_fixn.sh, _fixn2.sh):
// the large-N FORWARD kernel gains 10-12 % (31.0 -> 28.0 us at N = 197, 32.2 -> 28.2 at N = 257) and has such instances (template
// parameter NC; 0 = the runtime argument); the backward kernel does not (-2 % at N = 50, +5 % / +1.5 % at 197 / 257).
namespace {

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// 16-byte row fragment: row `row` of a row-major [*, stride] bf16 matrix, elements 32*s+8*g..+7
__device__ __forceinline__ bf16x8 rowfrag(const bf16* base, size_t stride, int row, int s, int g) {
    return load_bf16x8(base + (size_t)row * stride + 32 * s + 8 * g);
}

// 16-byte row fragment of an LDS tile (swizzled chunks): row `row`, elements 32*s+8*g..+7
template <int LD>
__device__ __forceinline__ int att_swz(int r) { return LD == 64 ? (r >> 1) & 7 : 0; }
template <int LD>
__device__ __forceinline__ bf16x8 ldsfrag(const bf16* Ys, int row, int s, int g) {
    return *reinterpret_cast<const bf16x8*>(Ys + row * LD + 8 * ((4 * s + g) ^ att_swz<LD>(row)));
}
// the same from either side: an LDS tile (LDS = true) or a row-major global matrix
template <bool LDS, int LD>
__device__ __forceinline__ bf16x8 xfrag(const bf16* base, size_t stride, int row, int s, int g) {
    if constexpr (LDS) return ldsfrag<LD>(base, row, s, g);
    else return rowfrag(base, stride, row, s, g);
}
// 16-byte piece `c` of row `y` of an LDS tile (staging writes)
template <int LD>
__device__ __forceinline__ bf16x8* ldschunk(bf16* Ys, int y, int c) {
    return reinterpret_cast<bf16x8*>(Ys + y * LD + 8 * (c ^ att_swz<LD>(y)));
}

// Fragment of the TRANSPOSE of a row-major tile Ys[y][LDR] (LDS), output row m -> d = 16*(m>>2) + 4*dt + (m&3), with gfx950's
// transposing read: in each 16-lane
// group, lane 4j+q passes the address of 4 consecutive d of token-row j, and lane i receives, as element j, element i&3
// of the piece addressed by lane 4j + (i>>2) (measured: scripts/probe_tr_b16.hip).  Lane 4j+q therefore points at
// Ys[32s + 4g + j][16q + 4dt ..+3], and lane m ends up with d = 16*(m>>2) + 4*dt + (m&3) for the tokens 32s+4g+0..3
// (second read: +16 tokens) -- no transposed copy in LDS, no scattered 2-byte writes.
template <int LDR>
__device__ __forceinline__ bf16x8 tfrag_tr(const bf16* Ys, int dt, int s, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int m = lane & 15, g = lane >> 4;
    const int R = 32 * s + 4 * g + (m >> 2);                     // (row R + 16 has the same swizzle)
    const bf16* src = Ys + R * LDR + 8 * ((2 * (m & 3) + (dt >> 1)) ^ att_swz<LDR>(R)) + 4 * (dt & 1);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(src));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(src + 16 * LDR));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
}

__device__ __forceinline__ void store16(bf16* dst, const f32x4 o[4], float scale) {
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

// ------------------------------------------------------------------------------------
template <int KT32, int NW, int LDK, int NC>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                       const bf16* __restrict__ v, bf16* __restrict__ out, int ldo,
                                                       float* __restrict__ lse, int H, int N_rt, unsigned char* __restrict__ out8) {
    const int N = NC ? NC : N_rt;
    constexpr int NPAD = 32 * KT32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Ks = reinterpret_cast<bf16*>(smem);                 // [NPAD][LDK] row-major (rows beyond N: whatever row N - 1 holds)
    bf16* Vs = Ks + NPAD * LDK;                               // [NPAD][LDK] row-major, rows beyond N zero
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const bf16* qh = q + (size_t)bh * N * 64;
    const bf16* kh = k + (size_t)bh * N * 64;
    const bf16* vh = v + (size_t)bh * N * 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;

    // K and V both stay row-major (16-byte LDS writes); V^T fragments come from the transposing read (tfrag_tr).
    // One round trip to HBM for everything the first query tile needs: all K / V pieces of this thread AND its Q fragments are
    // requested before the first LDS write, and no load sits inside a bounds branch (rows beyond N read row N - 1; V's are
    // zeroed by a select) -- a branch around a load makes hipcc drain behind it, one round trip per piece.
    const int nxt = (N + 15) >> 4;
    bf16x8 qf[2];
    {
        const int xs0 = min(16 * wid + c16, N - 1);
        qf[0] = rowfrag(qh, 64, xs0, 0, g);
        qf[1] = rowfrag(qh, 64, xs0, 1, g);
    }
    constexpr int IT = (NPAD * 8 + 64 * NW - 1) / (64 * NW);
    if constexpr (IT <= 6) {
        bf16x8 kk[IT], vv[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = threadIdx.x + 64 * NW * it, y = idx >> 3, c = idx & 7;
            const int ys = y < N ? y : N - 1;
            kk[it] = load_bf16x8(kh + (size_t)ys * 64 + 8 * c);
            vv[it] = load_bf16x8(vh + (size_t)ys * 64 + 8 * c);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = threadIdx.x + 64 * NW * it, y = idx >> 3, c = idx & 7;
            if (idx < NPAD * 8) {
                *ldschunk<LDK>(Ks, y, c) = kk[it];
                *ldschunk<LDK>(Vs, y, c) = y < N ? vv[it] : zero_bf16x8();
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < NPAD * 8; idx += 64 * NW) {
            const int y = idx >> 3, c = idx & 7;
            const int ys = y < N ? y : N - 1;
            const bf16x8 kk = load_bf16x8(kh + (size_t)ys * 64 + 8 * c);
            const bf16x8 vv = load_bf16x8(vh + (size_t)ys * 64 + 8 * c);
            *ldschunk<LDK>(Ks, y, c) = kk;
            *ldschunk<LDK>(Vs, y, c) = y < N ? vv : zero_bf16x8();
        }
    }
    __syncthreads();

    for (int xt = wid; xt < nxt; xt += NW) {
        const int xq = 16 * xt + c16;                 // this lane's query (column)
        const int xs = xq < N ? xq : N - 1;
        if (xt != wid) {
            qf[0] = rowfrag(qh, 64, xs, 0, g);
            qf[1] = rowfrag(qh, 64, xs, 1, g);
        }
        f32x4 z[2 * KT32];
        float m = -3.0e38f;
#pragma unroll
        for (int yt = 0; yt < 2 * KT32; ++yt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = mfma16(ldsfrag<LDK>(Ks, 16 * yt + c16, 0, g), qf[0], acc);
            acc = mfma16(ldsfrag<LDK>(Ks, 16 * yt + c16, 1, g), qf[1], acc);
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
        bf16x8 pf[KT32];
#pragma unroll
        for (int s = 0; s < KT32; ++s)
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(z[2 * s + half][r] - m);
                    // the row sum uses the bf16-rounded probabilities that the PV product sees
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
            for (int s = 0; s < KT32; ++s) o[dt] = mfma16(tfrag_tr<LDK>(Vs, dt, s, lane), pf[s], o[dt]);
        }
        if (xq < N) {
            store16(out + ((size_t)b * N + xq) * ldo + h * 64 + 16 * g, o, 1.0f / l);
            if (out8) {      // e4m3 copy (k-permuted) for the fp8 x fp8 out-projection
                const float inv = 1.0f / l;
                float lo8[8], hi8[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { lo8[r] = o[0][r] * inv; lo8[4 + r] = o[1][r] * inv; hi8[r] = o[2][r] * inv; hi8[4 + r] = o[3][r] * inv; }
                unsigned char* rb = out8 + ((size_t)b * N + xq) * ldo;
                store8_fp8(rb, h * 64 + 16 * g, lo8);
                store8_fp8(rb, h * 64 + 16 * g + 8, hi8);
            }
            if (g == 0) lse[(size_t)bh * N + xq] = m + __logf(l);
        }
    }
}

// ------------------------------------------------------------------------------------
// rows [0, NPAD) of a row-major LDS tile <- src rows (stride elements apart), zero beyond N; 16-byte loads and writes
// Four pieces per thread are requested before the first LDS write and none inside a bounds branch (rows beyond N read row N - 1
// and are zeroed by a select): with `v = 0; if (y < N) v = load` hipcc drains behind every request, one round trip per piece.
template <int LDR>
__device__ __forceinline__ void stage_rows(bf16* dst, const bf16* src, size_t stride, int N, int NPAD) {
    const int total = NPAD * 8, step = blockDim.x;
    for (int idx0 = threadIdx.x; idx0 < total; idx0 += 4 * step) {
        bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = min(idx0 + u * step, total - 1);
            const int y = idx >> 3, c = idx & 7;
            v[u] = load_bf16x8(src + (size_t)min(y, N - 1) * stride + 8 * c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = idx0 + u * step;
            const int y = idx >> 3, c = idx & 7;
            if (idx < total) *ldschunk<LDR>(dst, y, c) = y < N ? v[u] : zero_bf16x8();
        }
    }
}

// ALL4 (N <= 64): Q, K, V, dO of the head are LDS-resident for both passes.  Otherwise the tiles of the loop side only:
// K, V during pass A, then Q, dO in the same LDS during pass B; the 16-row x-side fragments come from global memory once
// per tile.  Either way every fragment inside the pass loops is an LDS read: row fragments ds_read_b128, transposed
// ones ds_read_b64_tr_b16 (tfrag_tr) -- no transposed copies, no dependent global loads in the loops.
#ifndef ATT_BWD_MINW
#define ATT_BWD_MINW 3     // waves per SIMD the N <= 64 backward kernel is compiled for: 158 VGPRs, three 41 KB workgroups per CU (23.5 -> 21.6 us in step; 4 spills)
#endif
template <int KT32, bool ALL4, int NW, int LDR>
__global__ __launch_bounds__(64 * NW, (ALL4 ? ATT_BWD_MINW : 1)) void attn_bwd_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                          const bf16* __restrict__ v, const bf16* __restrict__ out,
                                                          int ldo, const bf16* __restrict__ dout, int lddo,
                                                          const float* __restrict__ lse, bf16* __restrict__ dqkv, int ld,
                                                          int H, int N_rt, int dout_cls) {
    const int N = N_rt;
    constexpr int NPAD = 32 * KT32;
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [lse][delta][tile 0][tile 1]([tile 2][tile 3])
    float* lse_s = reinterpret_cast<float*>(smem);
    float* del_s = lse_s + NPAD;
    bf16* T0 = reinterpret_cast<bf16*>(del_s + NPAD);
    bf16* T1 = T0 + NPAD * LDR;
    bf16* Qs = ALL4 ? T0 : T0;                 // pass B (and, with ALL4, pass A's x side)
    bf16* Ks = ALL4 ? T1 : T0;                 // pass A
    bf16* Vs = ALL4 ? T1 + NPAD * LDR : T1;    // pass A
    bf16* dOs = ALL4 ? T1 + 2 * NPAD * LDR : T1;   // pass B
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H, E = H * 64;
    const bf16* qh = q + (size_t)bh * N * 64;
    const bf16* kh = k + (size_t)bh * N * 64;
    const bf16* vh = v + (size_t)bh * N * 64;
    const bf16* oh = out + (size_t)b * N * ldo + h * 64;       // row y at oh + y*ldo
    const bf16* doh = dout + (size_t)b * N * lddo + h * 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;

    if constexpr (ALL4) {
        // One round trip to HBM: the four tensors of both loop iterations are requested before the first LDS write.  Padded rows
        // are zero.  delta[y] = sum_keys P[y][key] dP[y][key] (== sum_d dO[y][d] O[y][d]) comes out of pass A, where a lane holds
        // P and dP of its query in f32 anyway: the attention output is not read at all (9.8 of this kernel's 79 MB at ViT-B/32,
        // batch 128), and delta no longer carries the bf16 rounding of the stored output.
        constexpr int IT = NPAD * 8 / NT;
        // (no load inside a bounds branch: rows beyond N read row N - 1 and are zeroed by a select afterwards -- with
        // `v = 0; if (y < N) v = load` hipcc drains behind every group of requests: one round trip per `it`, and one more for lse)
        bf16x8 vq[IT], vk[IT], vv[IT], vd[IT];
        float vl[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = threadIdx.x + NT * it, y = idx >> 3, c = idx & 7;
            const int yc = y < N ? y : N - 1;
            vq[it] = load_bf16x8(qh + (size_t)yc * 64 + 8 * c);
            vk[it] = load_bf16x8(kh + (size_t)yc * 64 + 8 * c);
            vv[it] = load_bf16x8(vh + (size_t)yc * 64 + 8 * c);
            vd[it] = load_bf16x8(doh + (size_t)(dout_cls ? 0 : yc) * lddo + 8 * c);     // dout_cls: only the class-token row carries a gradient
            vl[it] = lse[(size_t)bh * N + yc];
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = threadIdx.x + NT * it, y = idx >> 3, c = idx & 7;
            if (y >= N) { vq[it] = zero_bf16x8(); vk[it] = zero_bf16x8(); vv[it] = zero_bf16x8(); vd[it] = zero_bf16x8(); vl[it] = 0.f; }
            if (dout_cls && y != 0) vd[it] = zero_bf16x8();
            *ldschunk<LDR>(Qs, y, c) = vq[it];
            *ldschunk<LDR>(Ks, y, c) = vk[it];
            *ldschunk<LDR>(Vs, y, c) = vv[it];
            *ldschunk<LDR>(dOs, y, c) = vd[it];
            if (c == 0) {
                del_s[y] = 0.f;                 // rows < N: written by pass A; the pad rows stay 0 (pass B multiplies them by p = 0)
                lse_s[y] = vl[it];
            }
        }
    } else {
        stage_rows<LDR>(Ks, kh, 64, N, NPAD);
        stage_rows<LDR>(Vs, vh, 64, N, NPAD);
        // delta[y] = sum_d dO[y][d] * O[y][d]   (== sum_keys P*dP), 8 lanes per row
        for (int idx0 = threadIdx.x; idx0 < NPAD * 8; idx0 += 2 * NT) {     // two pieces per round, clamped rows, no load in a branch
            bf16x8 a[2], o[2];
            float ls[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = min(idx0 + u * NT, NPAD * 8 - 1);
                const int y = min(idx >> 3, N - 1), c = idx & 7;
                a[u] = load_bf16x8(doh + (size_t)y * lddo + 8 * c);
                o[u] = load_bf16x8(oh + (size_t)y * ldo + 8 * c);
                ls[u] = lse[(size_t)bh * N + y];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = idx0 + u * NT;
                const int y = idx >> 3, c = idx & 7;
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += bf2f(a[u][i]) * bf2f(o[u][i]);
                acc += __shfl_xor(acc, 1, 64);
                acc += __shfl_xor(acc, 2, 64);
                acc += __shfl_xor(acc, 4, 64);
                if (c == 0 && idx < NPAD * 8) {
                    del_s[y] = y < N ? acc : 0.f;
                    lse_s[y] = y < N ? ls[u] : 0.f;
                }
            }
        }
    }
    __syncthreads();

    const int ntile = (N + 15) >> 4;
    // One tile more than a whole number of rounds (ViT-L/14: N = 257 -> 17 tiles on 16 waves, i.e. TWO rounds for 1.06 rounds of
    // work; a 9-wave build of this kernel ran exactly as long as the 16-wave one): the last tile is then shared by the waves --
    // wave w takes the 32-key (pass B: 32-query) chunk w of it, the partial sums meet in LDS scratch behind the tiles
    // (launch_bwd reserves COOP_BYTES) and wave 0 stores them.  Round 5.
    const bool coop = !ALL4 && (ntile % NW) == 1 && ntile > NW && KT32 <= NW;
    const int nmain = coop ? ntile - 1 : ntile;
    float* coop_s = reinterpret_cast<float*>(smem + 2 * NPAD * 4 + 2 * NPAD * LDR * 2);      // [KT32][64 lanes][16] f32
    // the y-chunks [s0, s1) of one x-tile of pass A: dQ^T partial sums in o
    auto passA_chunks = [&](const bf16x8 (&x1)[2], const bf16x8 (&x2)[2], float lse_x, float del_x, int s0, int s1, f32x4 (&o)[4]) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll(KT32 <= 2 ? KT32 : 1)
        for (int s = s0; s < s1; ++s) {
            bf16x8 dsb;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int yr = 32 * s + 16 * half + c16;       // padded rows of the LDS tiles are zero
                f32x4 z1 = {0.f, 0.f, 0.f, 0.f}, z2 = {0.f, 0.f, 0.f, 0.f};
                z1 = mfma16(ldsfrag<LDR>(Ks, yr, 0, g), x1[0], z1);
                z1 = mfma16(ldsfrag<LDR>(Ks, yr, 1, g), x1[1], z1);
                z2 = mfma16(ldsfrag<LDR>(Vs, yr, 0, g), x2[0], z2);
                z2 = mfma16(ldsfrag<LDR>(Vs, yr, 1, g), x2[1], z2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 32 * s + 16 * half + 4 * g + r;
                    const float p = key < N ? __expf(z1[r] - lse_x) : 0.f;
                    dsb[half * 4 + r] = f2bf(p * (z2[r] - del_x));
                }
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(tfrag_tr<LDR>(Ks, dt, s, lane), dsb, o[dt]);
        }
    };
    // partial sums of the shared tile: every wave with a chunk leaves 16 floats per lane, wave 0 adds them in chunk order
    auto coop_put = [&](const f32x4 (&o)[4]) {
        if (wid < KT32) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(coop_s + ((size_t)wid * 64 + lane) * 16 + 4 * dt) = o[dt];
        }
    };
    auto coop_sum = [&](f32x4 (&o)[4]) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < KT32; ++w)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(coop_s + ((size_t)w * 64 + lane) * 16 + 4 * dt);
                o[dt][0] += v[0]; o[dt][1] += v[1]; o[dt][2] += v[2]; o[dt][3] += v[3];
            }
    };
    // ---------------- pass A: x = queries, y = keys -> dQ -----------------------------
    {
        const bf16* Qr = ALL4 ? Qs : qh;   const size_t qst = ALL4 ? LDR : 64;
        const bf16* Dr = ALL4 ? dOs : doh; const size_t dst_ = ALL4 ? LDR : (size_t)lddo;
        for (int xt = wid; xt < nmain; xt += NW) {
            const int xq = 16 * xt + c16;
            const int xs = xq < N ? xq : N - 1;
            bf16x8 x1[2], x2[2];
            x1[0] = xfrag<ALL4, LDR>(Qr, qst, xs, 0, g);  x1[1] = xfrag<ALL4, LDR>(Qr, qst, xs, 1, g);
            x2[0] = xfrag<ALL4, LDR>(Dr, dst_, xs, 0, g); x2[1] = xfrag<ALL4, LDR>(Dr, dst_, xs, 1, g);
            const float lse_x = lse_s[xs];
            f32x4 o[4];
            if constexpr (ALL4) {
                // all keys of this lane's query first (P and dP in f32), delta = sum P dP over the lane's keys and the four lane
                // groups, then dS and the dQ products
                f32x4 zp[KT32][2], zd[KT32][2];
                float dsum = 0.f;
#pragma unroll
                for (int s = 0; s < KT32; ++s)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int yr = 32 * s + 16 * half + c16;       // padded rows of the LDS tiles are zero
                        f32x4 z1 = {0.f, 0.f, 0.f, 0.f}, z2 = {0.f, 0.f, 0.f, 0.f};
                        z1 = mfma16(ldsfrag<LDR>(Ks, yr, 0, g), x1[0], z1);
                        z1 = mfma16(ldsfrag<LDR>(Ks, yr, 1, g), x1[1], z1);
                        z2 = mfma16(ldsfrag<LDR>(Vs, yr, 0, g), x2[0], z2);
                        z2 = mfma16(ldsfrag<LDR>(Vs, yr, 1, g), x2[1], z2);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = 32 * s + 16 * half + 4 * g + r;
                            const float p = key < N ? __expf(z1[r] - lse_x) : 0.f;
                            z1[r] = p;
                            dsum += p * z2[r];
                        }
                        zp[s][half] = z1; zd[s][half] = z2;
                    }
                dsum += __shfl_xor(dsum, 16, 64);
                dsum += __shfl_xor(dsum, 32, 64);
                if (g == 0 && xq < N) del_s[xq] = dsum;            // pass B reads it behind the barrier below
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KT32; ++s) {
                    bf16x8 dsb;
#pragma unroll
                    for (int half = 0; half < 2; ++half)
#pragma unroll
                        for (int r = 0; r < 4; ++r) dsb[half * 4 + r] = f2bf(zp[s][half][r] * (zd[s][half][r] - dsum));
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(tfrag_tr<LDR>(Ks, dt, s, lane), dsb, o[dt]);
                }
            } else {
            passA_chunks(x1, x2, lse_x, del_s[xs], 0, KT32, o);
            }
            if (xq < N) store16(dqkv + ((size_t)b * N + xq) * ld + h * 64 + 16 * g, o, 1.0f);
        }
        if constexpr (!ALL4) {
            if (coop) {                                    // workgroup-uniform
                const int xt = ntile - 1, xq = 16 * xt + c16, xs = xq < N ? xq : N - 1;
                f32x4 o[4];
                if (wid < KT32) {
                    bf16x8 x1[2], x2[2];
                    x1[0] = xfrag<ALL4, LDR>(Qr, qst, xs, 0, g);  x1[1] = xfrag<ALL4, LDR>(Qr, qst, xs, 1, g);
                    x2[0] = xfrag<ALL4, LDR>(Dr, dst_, xs, 0, g); x2[1] = xfrag<ALL4, LDR>(Dr, dst_, xs, 1, g);
                    passA_chunks(x1, x2, lse_s[xs], del_s[xs], wid, wid + 1, o);
                }
                coop_put(o);
                __syncthreads();
                if (wid == 0) {
                    coop_sum(o);
                    if (xq < N) store16(dqkv + ((size_t)b * N + xq) * ld + h * 64 + 16 * g, o, 1.0f);
                }
            }
        }
    }
    if constexpr (ALL4) {
        // delta of every query is in LDS before pass B reads it.  An LDS-only barrier: __syncthreads() would also wait (vmcnt(0))
        // for the dQ stores of pass A -- one store round trip per workgroup (tests/test_isa_hygiene.py)
        // (the barrier intrinsic carries no memory semantics and a fence would bring the vmcnt(0) back: the compiler-only memory
        // clobbers on both sides keep the del_s stores above and the del_s loads below from being moved across it)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!ALL4) {
        __syncthreads();                       // every wave is done with K, V
        stage_rows<LDR>(Qs, qh, 64, N, NPAD);
        stage_rows<LDR>(dOs, doh, (size_t)lddo, N, NPAD);
        __syncthreads();
    }
    // ---------------- pass B: x = keys, y = queries -> dK, dV -------------------------
    {
        const bf16* Kr = ALL4 ? Ks : kh;   const size_t kst = ALL4 ? LDR : 64;
        const bf16* Vr = ALL4 ? Vs : vh;   const size_t vst = ALL4 ? LDR : 64;
        // the y-chunks [s0, s1) of one x-tile of pass B: dK^T, dV^T partial sums
        auto passB_chunks = [&](const bf16x8 (&x1)[2], const bf16x8 (&x2)[2], int s0, int s1, f32x4 (&ok)[4], f32x4 (&ov)[4]) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { ok[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ov[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll(KT32 <= 2 ? KT32 : 1)
            for (int s = s0; s < s1; ++s) {
                bf16x8 dsb, pb;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int yt = 2 * s + half;
                    const int yr = 16 * yt + c16;
                    f32x4 z1 = {0.f, 0.f, 0.f, 0.f}, z2 = {0.f, 0.f, 0.f, 0.f};
                    z1 = mfma16(ldsfrag<LDR>(Qs, yr, 0, g), x1[0], z1);
                    z1 = mfma16(ldsfrag<LDR>(Qs, yr, 1, g), x1[1], z1);
                    z2 = mfma16(ldsfrag<LDR>(dOs, yr, 0, g), x2[0], z2);
                    z2 = mfma16(ldsfrag<LDR>(dOs, yr, 1, g), x2[1], z2);
                    const f32x4 lse_y = *reinterpret_cast<const f32x4*>(lse_s + 16 * yt + 4 * g);
                    const f32x4 del_y = *reinterpret_cast<const f32x4*>(del_s + 16 * yt + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qy = 16 * yt + 4 * g + r;
                        const float p = qy < N ? __expf(z1[r] - lse_y[r]) : 0.f;
                        pb[half * 4 + r] = f2bf(p);
                        dsb[half * 4 + r] = f2bf(p * (z2[r] - del_y[r]));
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    ok[dt] = mfma16(tfrag_tr<LDR>(Qs, dt, s, lane), dsb, ok[dt]);
                    ov[dt] = mfma16(tfrag_tr<LDR>(dOs, dt, s, lane), pb, ov[dt]);
                }
            }
        };
        for (int xt = wid; xt < nmain; xt += NW) {
            const int xk = 16 * xt + c16;
            const int xs = xk < N ? xk : N - 1;
            bf16x8 x1[2], x2[2];
            x1[0] = xfrag<ALL4, LDR>(Kr, kst, xs, 0, g); x1[1] = xfrag<ALL4, LDR>(Kr, kst, xs, 1, g);
            x2[0] = xfrag<ALL4, LDR>(Vr, vst, xs, 0, g); x2[1] = xfrag<ALL4, LDR>(Vr, vst, xs, 1, g);
            f32x4 ok[4], ov[4];
            passB_chunks(x1, x2, 0, KT32, ok, ov);
            if (xk < N) {
                bf16* dst = dqkv + ((size_t)b * N + xk) * ld + h * 64 + 16 * g;
                store16(dst + E, ok, 1.0f);
                store16(dst + 2 * E, ov, 1.0f);
            }
        }
        if constexpr (!ALL4) {
            if (coop) {
                const int xt = ntile - 1, xk = 16 * xt + c16, xs = xk < N ? xk : N - 1;
                f32x4 ok[4], ov[4];
                if (wid < KT32) {
                    bf16x8 x1[2], x2[2];
                    x1[0] = xfrag<ALL4, LDR>(Kr, kst, xs, 0, g); x1[1] = xfrag<ALL4, LDR>(Kr, kst, xs, 1, g);
                    x2[0] = xfrag<ALL4, LDR>(Vr, vst, xs, 0, g); x2[1] = xfrag<ALL4, LDR>(Vr, vst, xs, 1, g);
                    passB_chunks(x1, x2, wid, wid + 1, ok, ov);
                }
                __syncthreads();                               // pass A's reduction has read the scratch
                coop_put(ok);
                __syncthreads();
                bf16* dst = dqkv + ((size_t)b * N + xk) * ld + h * 64 + 16 * g;
                if (wid == 0) { f32x4 t[4]; coop_sum(t); if (xk < N) store16(dst + E, t, 1.0f); }
                __syncthreads();
                coop_put(ov);
                __syncthreads();
                if (wid == 0) { f32x4 t[4]; coop_sum(t); if (xk < N) store16(dst + 2 * E, t, 1.0f); }
            }
        }
    }
}

template <int KT32, int NW, int LD, int NC = 0>
int launch_fwd(const bf16* q, const bf16* k, const bf16* v, bf16* out, int ldo, float* lse, int B, int H, int N,
               hipStream_t s, unsigned char* out8) {
    constexpr int NPAD = 32 * KT32;
    const int bytes = 2 * NPAD * LD * 2;
    static bool attr = false;
    if (!attr && bytes > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<KT32, NW, LD, NC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
            pevit_set_error("attn_fwd: cannot reserve %d bytes of LDS", bytes); return -1;
        }
        attr = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<KT32, NW, LD, NC>), dim3(B * H), dim3(64 * NW), bytes, s, q, k, v, out, ldo, lse, H, N, out8);
    LAUNCH_OK("attn_fwd_kernel");
    return 0;
}

template <int KT32, bool ALL4, int NW, int LD>
int launch_bwd(const bf16* q, const bf16* k, const bf16* v, const bf16* out, int ldo, const bf16* dout, int lddo,
               const float* lse, bf16* dqkv, int ld, int B, int H, int N, hipStream_t s, int dout_cls = 0) {
    constexpr int NPAD = 32 * KT32;
    // + the scratch of the shared last tile (attn_bwd_kernel `coop`: same condition), [KT32][64][16] f32 behind the two tiles -- only
    // where it is used: at N = 197 (13 tiles on 8 waves: no shared tile) the 74 KB block must keep admitting two workgroups per CU
    const int ntile = (N + 15) >> 4;
    const bool coop = !ALL4 && (ntile % NW) == 1 && ntile > NW && KT32 <= NW;
    const int bytes = 2 * NPAD * 4 + (ALL4 ? 4 : 2) * NPAD * LD * 2 + (coop ? KT32 * 64 * 16 * 4 : 0);
    constexpr int bytes_max = 2 * NPAD * 4 + (ALL4 ? 4 : 2) * NPAD * LD * 2 + (ALL4 ? 0 : KT32 * 64 * 16 * 4);   // the LIMIT set once; a launch passes what it uses
    static bool attr = false;
    if (!attr && bytes_max > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<KT32, ALL4, NW, LD>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes_max) != hipSuccess) {
            pevit_set_error("attn_bwd: cannot reserve %d bytes of LDS", bytes_max); return -1;
        }
        attr = true;
    }
    hipLaunchKernelGGL((attn_bwd_kernel<KT32, ALL4, NW, LD>), dim3(B * H), dim3(64 * NW), bytes, s, q, k, v, out, ldo, dout, lddo, lse,
                       dqkv, ld, H, N, ALL4 ? dout_cls : 0);
    LAUNCH_OK("attn_bwd_kernel");
    return 0;
}

}  // namespace

// waves per workgroup of the large-N variants (one or two workgroups per CU: the 13 / 17 sixteen-row tiles of a head are
// spread over that many waves)
#ifndef ATT_NW_BIG
#define ATT_NW_BIG 16
#endif
#ifndef ATT_NW_MID
#define ATT_NW_MID 8
#endif
#ifndef ATT_NW_BIG_F
#define ATT_NW_BIG_F 8
#endif
#ifndef ATT_NW_MID_F
#define ATT_NW_MID_F 4
#endif

int pevit_launch_attn_fwd(const bf16* q, const bf16* k, const bf16* v, bf16* out, int ldo, float* lse, int B, int H,
                          int N, hipStream_t s, unsigned char* out8) {
    if (N < 1 || N > 288) { pevit_set_error("attn_fwd: tokens per image N=%d outside [1,288]", N); return -1; }
    if (ldo % 8) { pevit_set_error("attn_fwd: ldo must be a multiple of 8"); return -1; }
    if (N <= 64) return launch_fwd<2, 4, ATT_LD_FWD>(q, k, v, out, ldo, lse, B, H, N, s, out8);
    if (N == 197) return launch_fwd<7, ATT_NW_MID_F, ATT_LD_FWD, 197>(q, k, v, out, ldo, lse, B, H, N, s, out8);      // ViT-B/16
    if (N == 257) return launch_fwd<9, ATT_NW_BIG_F, ATT_LD_FWD, 257>(q, k, v, out, ldo, lse, B, H, N, s, out8);      // ViT-L/14
    if (N <= 224) return launch_fwd<7, ATT_NW_MID_F, ATT_LD_FWD>(q, k, v, out, ldo, lse, B, H, N, s, out8);
    return launch_fwd<9, ATT_NW_BIG_F, ATT_LD_FWD>(q, k, v, out, ldo, lse, B, H, N, s, out8);
}

int pevit_launch_attn_bwd(const bf16* q, const bf16* k, const bf16* v, const bf16* out, int ldo, const bf16* dout,
                          int lddo, const float* lse, bf16* dqkv, int ld, int B, int H, int N, hipStream_t s, int dout_cls_only) {
    if (N < 1 || N > 288) { pevit_set_error("attn_bwd: tokens per image N=%d outside [1,288]", N); return -1; }
    if ((ldo % 8) || (lddo % 8) || (ld % 8)) { pevit_set_error("attn_bwd: leading dims must be multiples of 8"); return -1; }
    // N <= 64: all four operands LDS-resident; above, the loop side of each pass
    if (N <= 64) return launch_bwd<2, true, 4, ATT_LD_BWD_SMALL>(q, k, v, out, ldo, dout, lddo, lse, dqkv, ld, B, H, N, s, dout_cls_only);
    if (dout_cls_only) { pevit_set_error("attn_bwd: dout_cls_only needs N <= 64 (N = %d)", N); return -1; }
    if (N <= 224) return launch_bwd<7, false, ATT_NW_MID, ATT_LD_BWD_MID>(q, k, v, out, ldo, dout, lddo, lse, dqkv, ld, B, H, N, s);   // 74 KB (padded rows): two workgroups per CU
    return launch_bwd<9, false, ATT_NW_BIG, ATT_LD_BWD_BIG>(q, k, v, out, ldo, dout, lddo, lse, dqkv, ld, B, H, N, s);           // 76 KB of tiles + 37 KB shared-tile scratch: one per CU
}
