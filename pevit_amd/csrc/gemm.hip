// bf16 x bf16 -> f32 "NT" GEMM on the CDNA4 matrix cores with fused epilogues.
//
//   C[M,N] = A[M,K] * B[N,K]^T          A, B row-major, K contiguous in both
//
// This one kernel family carries every dense contraction of the ViT block that the
// reference issues as aten::linear / mm: QKV projection (model.py:675),
// out-projection (:816), c_fc / c_proj (:959-961) and their dX-only backward forms
// (the backbone is frozen, so no dW GEMMs exist).  Weights are stored [out][in] exactly
// as the OpenAI checkpoint has them, which is already the K-contiguous "B^T" layout the
// MFMA B-fragment wants; the backward GEMMs use a transposed copy made once at load.
//
// Tile: 128x128x64 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave, built from
// v_mfma_f32_32x32x16_bf16).  Operands are staged HBM->LDS with 16-byte LDS-DMA
// (global_load_lds), double buffered, XOR-swizzled on the *source* side so that the
// ds_read_b128 fragment reads are bank-conflict free (cdna guide T2 / rule 21).
// Epilogue: accumulators are transposed through LDS so that every global access is a
// full 16-byte-per-lane row segment.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 KiB
constexpr int A_BYTES = BM * BK * 2;              // 16 KiB

__device__ __forceinline__ float sigmoidf_fast(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int row, int col, float v[8]) {
    // row < M and col < N (col multiple of 8) are guaranteed by the caller.
    if constexpr (EPI == EPI_QKV_HEADS) {
        const int E3 = 3 * p.E;
        if (col < E3) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            const int which = col / p.E, ce = col - which * p.E;
            const int h = ce >> 6, d = ce & 63;
            const int b = row / p.Ntok, n = row - b * p.Ntok;
            bf16* base = p.outb + (size_t)which * p.head_stride;
            bf16* dst = base + ((size_t)(b * p.H + h) * p.Ntok + n) * 64 + d;
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
            store_bf16x8(dst, o);
        } else {
            float* dst = p.outf + (size_t)row * p.ldo + (col - E3);
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    } else if constexpr (EPI == EPI_BIAS_RESID_F32) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        const float* r = p.resid + (size_t)row * p.ldr + col;
        const float4 r0 = *reinterpret_cast<const float4*>(r);
        const float4 r1 = *reinterpret_cast<const float4*>(r + 4);
        float* dst = p.outf + (size_t)row * p.ldo + col;
        *reinterpret_cast<float4*>(dst) =
            make_float4(v[0] + b0.x + r0.x, v[1] + b0.y + r0.y, v[2] + b0.z + r0.z, v[3] + b0.w + r0.w);
        *reinterpret_cast<float4*>(dst + 4) =
            make_float4(v[4] + b1.x + r1.x, v[5] + b1.y + r1.y, v[6] + b1.z + r1.z, v[7] + b1.w + r1.w);
    } else if constexpr (EPI == EPI_BIAS_GELU) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        bf16x8 h, g;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h[i] = f2bf(v[i]);
            // QuickGELU (model.py:163-165) evaluated on the bf16-rounded pre-activation that
            // the backward pass will see, so fwd and bwd agree on the same h.
            const float hv = bf2f(h[i]);
            g[i] = f2bf(hv * sigmoidf_fast(1.702f * hv));
        }
        store_bf16x8(p.outb + (size_t)row * p.ldob + col, h);
        store_bf16x8(p.outb2 + (size_t)row * p.ldob2 + col, g);
    } else if constexpr (EPI == EPI_DGELU_BF16) {
        const bf16x8 h = load_bf16x8(p.aux + (size_t)row * p.ldaux + col);
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float hv = bf2f(h[i]);
            const float s = sigmoidf_fast(1.702f * hv);
            o[i] = f2bf(v[i] * (s * (1.0f + 1.702f * hv * (1.0f - s))));
        }
        store_bf16x8(p.outb + (size_t)row * p.ldob + col, o);
    } else if constexpr (EPI == EPI_F32) {
        float* dst = p.outf + (size_t)row * p.ldo + col;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else if constexpr (EPI == EPI_BF16) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
        store_bf16x8(p.outb + (size_t)row * p.ldob + col, o);
    } else if constexpr (EPI == EPI_BIAS_BF16) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        bf16x8 o;
        o[0] = f2bf(v[0] + b0.x); o[1] = f2bf(v[1] + b0.y); o[2] = f2bf(v[2] + b0.z); o[3] = f2bf(v[3] + b0.w);
        o[4] = f2bf(v[4] + b1.x); o[5] = f2bf(v[5] + b1.y); o[6] = f2bf(v[6] + b1.z); o[7] = f2bf(v[7] + b1.w);
        store_bf16x8(p.outb + (size_t)row * p.ldob + col, o);
    } else if constexpr (EPI == EPI_PATCH_EMBED) {
        // row = b*G2 + g (patch index), output row = b*Ntok + 1 + g ; + positional embedding
        const int G2 = p.Ntok - 1;
        const int b = row / G2, g = row - b * G2;
        const float* pos = p.resid + (size_t)(1 + g) * p.ldr + col;
        const float4 r0 = *reinterpret_cast<const float4*>(pos);
        const float4 r1 = *reinterpret_cast<const float4*>(pos + 4);
        float* dst = p.outf + ((size_t)b * p.Ntok + 1 + g) * p.ldo + col;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0] + r0.x, v[1] + r0.y, v[2] + r0.z, v[3] + r0.w);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4] + r1.x, v[5] + r1.y, v[6] + r1.z, v[7] + r1.w);
    } else if constexpr (EPI == EPI_BIAS_RELU_BF16) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = f2bf(fmaxf(v[i], 0.0f));
        store_bf16x8(p.outb + (size_t)row * p.ldob + col, o);
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid >> 1, wn = wid & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA loader addressing --------------------------------------------------
    // Wave w, piece i (0..3) fills rows R = (w*4+i)*8 .. +7 of the 128x64 tile, one 1 KiB
    // lane-linear block; lane l lands at row R+(l>>3), physical 16-byte chunk l&7, and
    // fetches the logical chunk (l&7) ^ swz(row) of that row from HBM.
    const bf16* a_src[4];
    const bf16* b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wid * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
        int br = n0 + row; br = br < p.Nb ? br : p.Nb - 1;
        a_src[i] = p.A + (size_t)ar * p.lda + chunk * 8;
        b_src[i] = p.B + (size_t)br * p.ldb + chunk * 8;
    }
    auto issue_tile = [&](int kt, int stage) {
        char* sa = smem + stage * STAGE_BYTES + (wid * 4) * 1024;
        char* sb = sa + A_BYTES;
        const int koff = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(a_src[i] + koff, sa + i * 1024);
            glds16(b_src[i] + koff, sb + i * 1024);
        }
    };

    // ---- fragment read addressing ---------------------------------------------------
    // 32x32x16 bf16: lane l holds row (l&31), k = 8*(l>>5)..+7 of the 16-wide k-step.
    const int frow = lane & 31;
    const int fswz = (frow >> 1) & 7;
    const int fhalf = lane >> 5;
    int a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_off[i] = (wm * 64 + i * 32 + frow) * 128;
        b_off[i] = A_BYTES + (wn * 64 + i * 32 + frow) * 128;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.K / BK;
    issue_tile(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue_tile(kt + 1, (kt + 1) & 1);
        const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + coff);
                bfr[i] = *reinterpret_cast<const bf16x8*>(st + b_off[i] + coff);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: C fragment -> LDS (per-wave 64x64 f32) -> row segments ----------
    __syncthreads();
    float* cw = reinterpret_cast<float*>(smem + wid * 16384);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = j * 32 + (lane & 31);
                cw[row * 64 + col] = acc[i][j][r];
            }
    // same-wave LDS ops are ordered; make the compiler wait for the writes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int ncols = (EPI == EPI_QKV_HEADS) ? p.N : p.N;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int lr = pass * 8 + (lane >> 3);
        const int lc = (lane & 7) * 8;
        const int row = m0 + wm * 64 + lr;
        const int col = n0 + wn * 64 + lc;
        if (row < p.M && col < ncols) {
            float v[8];
            const float4 x0 = *reinterpret_cast<const float4*>(cw + lr * 64 + lc);
            const float4 x1 = *reinterpret_cast<const float4*>(cw + lr * 64 + lc + 4);
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w;
            v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
            epilogue_store<EPI>(p, row, col, v);
        }
    }
}

template <int EPI>
int launch_epi(const GemmParams& p, hipStream_t stream) {
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_nt_kernel<EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES) != hipSuccess) {
            pevit_set_error("hipFuncSetAttribute(gemm, %d) failed", EPI);
            return -1;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_bf16_nt_kernel<EPI>, dim3(tiles), dim3(256), 2 * STAGE_BYTES, stream, p);
    return 0;
}

}  // namespace

int pevit_launch_gemm(int epi, const GemmParams& p, hipStream_t stream) {
    if (p.K % BK != 0 || p.K <= 0) { pevit_set_error("gemm: K=%d must be a positive multiple of %d", p.K, BK); return -1; }
    if (p.N % 8 != 0) { pevit_set_error("gemm: N=%d must be a multiple of 8", p.N); return -1; }
    if (p.M <= 0 || p.N <= 0) { pevit_set_error("gemm: empty problem M=%d N=%d", p.M, p.N); return -1; }
    if ((p.lda % 8) || (p.ldb % 8)) { pevit_set_error("gemm: lda/ldb must be multiples of 8"); return -1; }
    switch (epi) {
        case EPI_QKV_HEADS: return launch_epi<EPI_QKV_HEADS>(p, stream);
        case EPI_BIAS_RESID_F32: return launch_epi<EPI_BIAS_RESID_F32>(p, stream);
        case EPI_BIAS_GELU: return launch_epi<EPI_BIAS_GELU>(p, stream);
        case EPI_DGELU_BF16: return launch_epi<EPI_DGELU_BF16>(p, stream);
        case EPI_F32: return launch_epi<EPI_F32>(p, stream);
        case EPI_BF16: return launch_epi<EPI_BF16>(p, stream);
        case EPI_BIAS_BF16: return launch_epi<EPI_BIAS_BF16>(p, stream);
        case EPI_PATCH_EMBED: return launch_epi<EPI_PATCH_EMBED>(p, stream);
        case EPI_BIAS_RELU_BF16: return launch_epi<EPI_BIAS_RELU_BF16>(p, stream);
    }
    pevit_set_error("gemm: unknown epilogue %d", epi);
    return -1;
}
