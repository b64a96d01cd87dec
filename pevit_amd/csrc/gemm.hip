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
// Workgroup = 256 threads = 4 waves as 2x2; a BM x BN x BK tile is built from
// v_mfma_f32_32x32x16_bf16.  Operands are staged HBM->LDS with 16-byte LDS-DMA
// (global_load_lds), double buffered, XOR-swizzled on the *source* side so that the
// ds_read_b128 fragment reads are bank-conflict free (cdna guide T2 / rule 21).
// Epilogue: accumulators are transposed through LDS, one 32x32 fragment per wave at a time, so
// that global accesses are 16/32-byte-per-lane row segments.
//
// The fine-tune step's GEMMs are SMALL (M = B*N = 6400 rows): a 128x128 tiling gives only 300
// workgroups for the N=768 products, fewer than the 512 slots of 256 CUs x 2, and measured
// per-workgroup speed does not depend on how many workgroups share a CU (one wave per SIMD is
// latency-bound on its own ds_read -> MFMA chain).  So the tile shape is chosen per problem to
// put >= ~3 workgroups on every CU: see pick_config().
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TILE_BAND = 6;
int g_gemm_config = -1;    // -1: heuristic ; >= 0: force a tile configuration (A/B measurements)
int g_gemm_persistent = 1;
int g_gemm_hoist = 1;       // hoist all fragment reads of a k-tile ahead of its MFMAs
int g_gemm_ablate = 0;      // measurement only (GemmParams::dbg)
int g_gemm_kswitch = 2048;  // K from which the few-tile problems use the 128x128 tile instead of 64x128
int g_gemm_dephase = 0;     // x 512 clk start delay of the second half of the grid (0: off; helps back-to-back microbenchmarks by 7 %, costs 2 % inside the step)

int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

__device__ __forceinline__ float sigmoidf_fast(float x) { return 1.0f / (1.0f + __expf(-x)); }
// transformers' "gelu_new" (compacter_model.py:8,172): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
__device__ __forceinline__ float gelu_new_f(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float gelu_new_grad_f(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float th = tanhf(u);
    return 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
}

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
    } else if constexpr (EPI == EPI_BIAS_RESID_KEEP) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        float* d2 = p.outf2 + (size_t)row * p.ldo2 + col;
        *reinterpret_cast<float4*>(d2) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(d2 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        const float* r = p.resid + (size_t)row * p.ldr + col;
        const float4 r0 = *reinterpret_cast<const float4*>(r);
        const float4 r1 = *reinterpret_cast<const float4*>(r + 4);
        float* dst = p.outf + (size_t)row * p.ldo + col;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0] + r0.x, v[1] + r0.y, v[2] + r0.z, v[3] + r0.w);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4] + r1.x, v[5] + r1.y, v[6] + r1.z, v[7] + r1.w);
    } else if constexpr (EPI == EPI_BIAS_GELUNEW) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        bf16x8 a, g;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a[i] = f2bf(v[i]);
            g[i] = f2bf(gelu_new_f(bf2f(a[i])));
        }
        store_bf16x8(p.outb + (size_t)row * p.ldob + col, a);
        store_bf16x8(p.outb2 + (size_t)row * p.ldob2 + col, g);
    } else if constexpr (EPI == EPI_DRELU_BF16) {
        const bf16x8 a = load_bf16x8(p.aux + (size_t)row * p.ldaux + col);
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = f2bf(bf2f(a[i]) > 0.f ? v[i] : 0.f);
        store_bf16x8(p.outb + (size_t)row * p.ldob + col, o);
    } else if constexpr (EPI == EPI_DGELUNEW_BF16) {
        const bf16x8 a = load_bf16x8(p.aux + (size_t)row * p.ldaux + col);
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i] * gelu_new_grad_f(bf2f(a[i])));
        store_bf16x8(p.outb + (size_t)row * p.ldob + col, o);
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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// tile coordinates of this workgroup.  Order: XCD-contiguous (xcd_remap), and inside that a band
// of TILE_BAND m-tiles is walked n-major, so the tiles an XCD has in flight share TILE_BAND
// A-panels and only a few B-panels (a 128x768 bf16 panel is 192 KiB; the XCD's L2 is 4 MiB).
template <int BM, int BN>
__device__ __forceinline__ void tile_origin(const GemmParams& p, int tile, int& m0, int& n0) {
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int t = xcd_remap(tile, tiles_m * tiles_n);
    const int band = t / (TILE_BAND * tiles_n), within = t - band * (TILE_BAND * tiles_n);
    const int mb = min(TILE_BAND, tiles_m - band * TILE_BAND);
    const int tn = within / mb, tm = band * TILE_BAND + (within - tn * mb);
    m0 = tm * BM; n0 = tn * BN;
}

// Persistent form: gridDim.x workgroups walk the tiles (tile = blockIdx.x, += gridDim.x; gridDim.x
// is a multiple of 8 so a workgroup's tiles stay on one XCD range).  The first k-tile of the NEXT
// output tile is requested (LDS-DMA into stage 0) before the epilogue of the current one runs out of
// stage 1, so the HBM/L2 latency of the prologue -- one of only 12 k-iterations when K = 768 -- is
// hidden behind the epilogue's LDS transposes and global stores.
template <int EPI, int BM, int BN, int BK, int MINB, int SCHED>
__global__ __launch_bounds__(256, MINB) void gemm_bf16_nt_kernel(GemmParams p, int ntiles) {
    constexpr int WM = BM / 64, WN = BN / 64;           // 32x32 fragments per wave (m, n)
    constexpr int ROWB = BK * 2;                        // bytes per tile row
    constexpr int CH = BK / 8;                          // 16-byte chunks per row
    constexpr int RPP = 1024 / ROWB;                    // rows per 1 KiB LDS-DMA piece
    constexpr int SWZ_SHIFT = (ROWB == 128) ? 1 : 2;    // rows per 256-byte LDS bank row: 2 or 4
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM / RPP / 4, PB = BN / RPP / 4; // pieces per wave
    constexpr int KS = BK / 16;                         // MFMA k-steps per k-tile
    static_assert(PA >= 1 && PB >= 1, "tile too small for 4 loader waves");
    static_assert(STAGE_BYTES >= 16384, "the epilogue borrows 16 KiB of stage 1");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid >> 1, wn = wid & 1;

    // LDS-DMA: a piece is RPP rows x ROWB bytes = 1 KiB, lane-linear; lane l lands at row
    // R + l/CH, physical chunk l%CH, and fetches the logical chunk (l%CH) ^ swz(row) from HBM.
    const bf16* a_src[PA];
    const bf16* b_src[PB];
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = (wid * PA + i) * RPP + lane / CH;
            const int chunk = (lane % CH) ^ ((row >> SWZ_SHIFT) & (CH - 1));
            int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
            a_src[i] = p.A + (size_t)ar * p.lda + chunk * 8;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int row = (wid * PB + i) * RPP + lane / CH;
            const int chunk = (lane % CH) ^ ((row >> SWZ_SHIFT) & (CH - 1));
            int br = n0 + row; br = br < p.Nb ? br : p.Nb - 1;
            b_src[i] = p.B + (size_t)br * p.ldb + chunk * 8;
        }
    };
    auto issue_tile = [&](int kt, int stage) {
        char* sa = smem + stage * STAGE_BYTES + (wid * PA) * 1024;
        char* sb = smem + stage * STAGE_BYTES + A_BYTES + (wid * PB) * 1024;
        const int koff = kt * BK;
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(a_src[i] + koff, sa + i * 1024);
#pragma unroll
        for (int i = 0; i < PB; ++i) glds16(b_src[i] + koff, sb + i * 1024);
    };
    // 32x32x16 bf16 fragment: lane l holds row (l&31), k = 8*(l>>5)..+7 of the 16-wide k-step.
    const int frow = lane & 31, fswz = (frow >> SWZ_SHIFT) & (CH - 1), fhalf = lane >> 5;
    int a_off[WM], b_off[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a_off[i] = (wm * (BM / 2) + i * 32 + frow) * ROWB;
#pragma unroll
    for (int i = 0; i < WN; ++i) b_off[i] = A_BYTES + (wn * (BN / 2) + i * 32 + frow) * ROWB;

    const int nk = (p.dbg & 1) ? 0 : p.K / BK;
    int tile = blockIdx.x;
    // De-phase the two workgroups that share a CU: they run tiles of equal length, so left alone both are in
    // their k-loop (operand stream) and then both in their epilogue (store burst) at the same time.  Holding
    // the second half of the grid back by ~half an epilogue makes one's stores overlap the other's loads
    // (measured -7 % on the step's GEMMs, scripts/phase_gemm.py).
    if (p.dephase > 0 && blockIdx.x >= (gridDim.x >> 1))
        for (int i = 0; i < p.dephase; ++i) __builtin_amdgcn_s_sleep(8);
    int m0, n0;
    tile_origin<BM, BN>(p, tile, m0, n0);
    set_sources(m0, n0);
    constexpr bool ONE_STAGE = (SCHED == 3);    // single LDS stage (32 KiB): no overlap inside the workgroup, 4 workgroups per CU
    if constexpr (!ONE_STAGE) issue_tile(0, 0);
    while (true) {
        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        if constexpr (SCHED == 4) {
            // Register-resident schedule (the 128x128 edition of gemm256_kernel): all fragments of a k-tile are
            // pulled into registers, the stage is released with a barrier and immediately re-targeted by the
            // LDS-DMA of k-tile kt+2, so two k-tiles are in flight while k-tile kt is multiplied.
            constexpr int G = PA + PB;
            __syncthreads();                            // previous epilogue no longer uses stage 1 as scratch
            if (nk > 1) issue_tile(1, 1);
            if (nk > 1) wait_vmcnt<G>(); else wait_vmcnt<0>();      // k-tile 0 (and, in order, the epilogue's stores)
            __builtin_amdgcn_s_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                const char* st = smem + (kt & 1) * STAGE_BYTES;
                bf16x8 af[KS][WM], bfr[KS][WN];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
#pragma unroll
                    for (int i = 0; i < WM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + coff);
#pragma unroll
                    for (int j = 0; j < WN; ++j) bfr[ks][j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + coff);
                }
#pragma unroll
                for (int ks = 0; ks < KS / 2; ++ks)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();           // nobody reads this stage any more
                if (kt + 2 < nk) issue_tile(kt + 2, kt & 1);
#pragma unroll
                for (int ks = KS / 2; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
                if (kt + 1 < nk) {
                    if (kt + 2 < nk) wait_vmcnt<G>(); else wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                }
            }
        } else
        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (ONE_STAGE) {
                __syncthreads();                       // the stage is no longer being read
                issue_tile(kt, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (kt + 1 < nk) issue_tile(kt + 1, (kt + 1) & 1);
            }
            const char* st = smem + (ONE_STAGE ? 0 : (kt & 1)) * STAGE_BYTES;
            if constexpr (SCHED == 0 || SCHED == 3) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
                    bf16x8 af[WM], bfr[WN];
#pragma unroll
                    for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + coff);
#pragma unroll
                    for (int j = 0; j < WN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + coff);
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            } else {
                // all fragments of the k-tile are read up front (KS*(WM+WN) ds_read_b128), then the
                // MFMAs run back to back; SCHED 2 additionally raises the wave priority for them.
                bf16x8 af[KS][WM], bfr[KS][WN];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
#pragma unroll
                    for (int i = 0; i < WM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + coff);
#pragma unroll
                    for (int j = 0; j < WN; ++j) bfr[ks][j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + coff);
                }
                if constexpr (SCHED == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
                if constexpr (SCHED == 2) __builtin_amdgcn_s_setprio(0);
            }
        }
        // both stages are idle after this barrier: stage 0 receives the next tile's first k-tile while
        // the epilogue transposes through (this wave's 4 KiB of) stage 1
        __syncthreads();
        const int cm0 = m0, cn0 = n0;
        const int next = tile + gridDim.x;
        if (next < ntiles) {
            tile_origin<BM, BN>(p, next, m0, n0);
            set_sources(m0, n0);
            if constexpr (!ONE_STAGE) issue_tile(0, 0);
        }
        float* cw = reinterpret_cast<float*>(smem + (ONE_STAGE ? 0 : STAGE_BYTES) + wid * 4096);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    cw[row * 32 + (lane & 31)] = acc[i][j][r];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int lr = pass * 16 + (lane >> 2);
                    const int lc = (lane & 3) * 8;
                    const int row = cm0 + wm * (BM / 2) + i * 32 + lr;
                    const int col = cn0 + wn * (BN / 2) + j * 32 + lc;
                    const float4 x0 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc);
                    const float4 x1 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc + 4);
                    if (row < p.M && col < p.N && !(p.dbg & 2)) {
                        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        epilogue_store<EPI>(p, row, col, v);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        if (next >= ntiles) break;
        tile = next;
    }
}

// ------------------------------------------------------------------------------------------------
// Ring form.  Measured on MI355X (scripts/ablate_gemm.py, profiles/r01_gemm_ablation.txt): the k-loop of
// the kernel above runs at the same per-workgroup speed whatever shares its CU, and every tile shape with
// the same number of operand bytes in flight per CU lands at the same TFLOP/s -- it is bound by the
// round trip of the ONE k-tile it keeps in flight, not by MFMA, LDS or L2 bandwidth.  This form keeps
// S-1 k-tiles (BK = 32, 16 KiB each) in flight per workgroup in an S-deep LDS ring, with counted
// s_waitcnt vmcnt (never 0 in steady state) and a raw s_barrier per k-tile, and treats the k-tiles of all
// the output tiles a persistent workgroup owns as ONE stream: the loads of the next output tile are
// already in flight while the current one runs its epilogue.
//
// Ordering rules relied on (MI355X_MICROARCH.md item 7 / cdna guide "8-phase" notes):
//   RAW  a ds_read of ring slot g sees the LDS-DMA data once the issuing waves waited for it with a
//        counted vmcnt AND the reader passed a barrier after that wait  (wait -> s_barrier -> ds_read);
//   WAR  slot (g-1) % S is re-targeted by DMA only after the barrier of iteration g, which every wave
//        reaches after its ds_reads of iteration g-1 returned (they feed its MFMAs);
//   vmcnt decrements in issue order on gfx9-family parts (loads, LDS-DMA and stores share the counter),
//        so "at most N outstanding" means "all but the youngest N completed": the epilogue's global
//        loads / stores sit between DMA groups in that order and only make the waits conservative.

template <int EPI, int BM, int BK, int S, int WROWS>
__global__ __launch_bounds__(256) void gemm_ring_kernel(GemmParams p, int ntiles) {
    constexpr int BN = 128, WCOLS = 4 / WROWS;
    constexpr int WTM = BM / WROWS, WTN = BN / WCOLS;            // wave tile
    constexpr int WM = WTM / 32, WN = WTN / 32;                 // 32x32 fragments per wave
    constexpr int ROWB = BK * 2, CH = BK / 8, RPP = 1024 / ROWB;
    constexpr int SWZ_SHIFT = (ROWB == 128) ? 1 : 2;
    constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int PA = BM / RPP / 4, PB = BN / RPP / 4;         // 1 KiB pieces per wave
    constexpr int G = PA + PB;                                  // LDS-DMA instructions per wave per k-tile
    constexpr int KS = BK / 16;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && (BM / RPP) % 4 == 0, "tile shape");
    static_assert(S >= 3 && (S - 2) * G <= 63 && STAGE_BYTES >= 16384, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / WCOLS, wn = wid % WCOLS;
    const int nk = (p.dbg & 1) ? 1 : p.K / BK;

    // ---- producer state: position of the next k-tile to request in this workgroup's stream
    int itile = blockIdx.x, ikt = 0, islot = 0;
    const bf16* a_src[PA];
    const bf16* b_src[PB];
    auto set_sources = [&](int tile) {
        int m0, n0;
        tile_origin<BM, BN>(p, tile, m0, n0);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = (wid * PA + i) * RPP + lane / CH;
            const int chunk = (lane % CH) ^ ((row >> SWZ_SHIFT) & (CH - 1));
            int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
            a_src[i] = p.A + (size_t)ar * p.lda + chunk * 8;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int row = (wid * PB + i) * RPP + lane / CH;
            const int chunk = (lane % CH) ^ ((row >> SWZ_SHIFT) & (CH - 1));
            int br = n0 + row; br = br < p.Nb ? br : p.Nb - 1;
            b_src[i] = p.B + (size_t)br * p.ldb + chunk * 8;
        }
    };
    auto issue_next = [&]() {
        if (itile >= ntiles) return;
        char* sa = smem + islot * STAGE_BYTES + (wid * PA) * 1024;
        char* sb = smem + islot * STAGE_BYTES + A_BYTES + (wid * PB) * 1024;
        const int koff = ikt * BK;
        if (!(p.dbg & 4)) {
#pragma unroll
            for (int i = 0; i < PA; ++i) glds16(a_src[i] + koff, sa + i * 1024);
#pragma unroll
            for (int i = 0; i < PB; ++i) glds16(b_src[i] + koff, sb + i * 1024);
        }
        islot = (islot + 1 == S) ? 0 : islot + 1;
        if (++ikt == nk) {
            ikt = 0;
            itile += gridDim.x;
            if (itile < ntiles) set_sources(itile);
        }
    };

    // ---- consumer state
    const int frow = lane & 31, fswz = (frow >> SWZ_SHIFT) & (CH - 1), fhalf = lane >> 5;
    int a_off[WM], b_off[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a_off[i] = (wm * WTM + i * 32 + frow) * ROWB;
#pragma unroll
    for (int j = 0; j < WN; ++j) b_off[j] = A_BYTES + (wn * WTN + j * 32 + frow) * ROWB;

    set_sources(itile);
#pragma unroll
    for (int i = 0; i < S - 1; ++i) issue_next();

    int cslot = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        for (int kt = 0; kt < nk; ++kt) {
            // the oldest group in flight is the k-tile about to be consumed; while the stream still has
            // k-tiles to request exactly S-1 groups are in flight here, afterwards fewer (drain).
            if (itile < ntiles) wait_vmcnt<(S - 2) * G>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            issue_next();
            const char* st = smem + cslot * STAGE_BYTES;
            cslot = (cslot + 1 == S) ? 0 : cslot + 1;
            if (p.dbg & 8) continue;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
                bf16x8 af[WM], bfr[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + coff);
#pragma unroll
                for (int j = 0; j < WN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + coff);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        // ---- epilogue through the ring slot that was consumed last: it is not a DMA target before the
        // next iteration's barrier.  One barrier so that no wave is still reading it.
        __builtin_amdgcn_s_barrier();
        int cm0, cn0;
        tile_origin<BM, BN>(p, tile, cm0, cn0);
        const int eslot = (cslot == 0) ? S - 1 : cslot - 1;
        float* cw = reinterpret_cast<float*>(smem + eslot * STAGE_BYTES + wid * 4096);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    cw[row * 32 + (lane & 31)] = acc[i][j][r];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int lr = pass * 16 + (lane >> 2);
                    const int lc = (lane & 3) * 8;
                    const int row = cm0 + wm * WTM + i * 32 + lr;
                    const int col = cn0 + wn * WTN + j * 32 + lc;
                    const float4 x0 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc);
                    const float4 x1 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc + 4);
                    if (row < p.M && col < p.N && !(p.dbg & 2)) {
                        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        epilogue_store<EPI>(p, row, col, v);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
    }
}

// 0 (default): two-stage kernel everywhere.  1: 160x128x64 three-stage ring (108 KiB, one workgroup per CU) for
// problems whose 128x128 tiling has more tiles than CUs while the 160-row tiling fits in one round (M = 6400,
// N = 768: 240 tiles).  2: force that ring.  5: the 128x128x32 five-stage ring.  Both rings are bit-identical to
// the two-stage kernel and measured SLOWER on the step's shapes (c_proj 63 vs 53 us; 6.32 vs 5.93 ms per step):
// kept as measurement variants, see profiles/r01_l2_fetch_bound.md.
int g_gemm_ring = 0;

template <int EPI, int BM, int BK, int S, int WROWS>
int launch_ring(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = S * (BM + 128) * BK * 2;
    auto kern = gemm_ring_kernel<EPI, BM, BK, S, WROWS>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            pevit_set_error("hipFuncSetAttribute(gemm ring epi %d BM %d) failed", EPI, BM);
            return -1;
        }
        attr_set = true;
    }
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, 128);
    const int slots = num_cus() * ((160 * 1024) / lds);
    hipLaunchKernelGGL(kern, dim3(tiles < slots ? tiles : slots), dim3(256), lds, stream, p, tiles);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// 256x256x64 form for problems that tile into at most ~one workgroup per CU (QKV at B=128: 25 x 10 tiles).
// Half the operand bytes per flop of the 128x128 kernel, which is what the L2 -> LDS stream bounds
// (profiles/r01_l2_fetch_bound.md).  One 512-thread workgroup per CU, waves 2 (M) x 4 (N), wave tile 128 x 64
// = 4 x 2 fragments of 32x32 (128 accumulator registers).  The operands of a whole k-tile are pulled from LDS
// into registers (16 A + 8 B fragments = 96 registers) and the MFMAs run from registers, so an LDS buffer is
// free again as soon as every wave has read it: with only two 64 KiB buffers TWO k-tiles of LDS-DMA stay in
// flight (k-tile t+1 landing, t+2 just requested) while k-tile t is being multiplied, with a counted vmcnt:
//     read k-tile t -> lgkmcnt(0), barrier -> request k-tile t+2 into the buffer just read -> MFMAs of t
//     -> vmcnt(8) (k-tile t+1 landed, t+2 may fly) -> barrier
// LDS image per buffer: A rows 0..255 then B rows 0..255, 128-byte rows, same source-side XOR swizzle as above.
template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(GemmParams p, int ntiles) {
    constexpr int BM = 256, BN = 256, BK = 64, ROWB = 128, CH = 8;
    constexpr int BUF_BYTES = (BM + BN) * ROWB;                 // 64 KiB
    constexpr int G = 8;                                        // LDS-DMA instructions per wave per k-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int nk = (p.dbg & 1) ? 0 : p.K / BK;
    const int frow = lane & 31, fswz = (frow >> 1) & (CH - 1), fhalf = lane >> 5;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int m0, n0;
        tile_origin<BM, BN>(p, tile, m0, n0);
        // a 64 KiB buffer is 64 pieces of 1 KiB (8 rows x 128 B); wave w loads pieces w, w+8, ... (4 of A, 4 of B)
        const bf16* src[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int piece = wid + 8 * i;                      // 0..31 A, 32..63 B
            const int row = (piece & 31) * 8 + (lane >> 3);     // row inside the 256-row panel
            const int chunk = (lane & 7) ^ ((row >> 1) & (CH - 1));
            if (piece < 32) {
                int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
                src[i] = p.A + (size_t)ar * p.lda + chunk * 8;
            } else {
                int br = n0 + row; br = br < p.Nb ? br : p.Nb - 1;
                src[i] = p.B + (size_t)br * p.ldb + chunk * 8;
            }
        }
        auto issue_tile = [&](int kt, int buf) {
            char* base = smem + buf * BUF_BYTES;
            const int koff = kt * BK;
#pragma unroll
            for (int i = 0; i < G; ++i) glds16(src[i] + koff, base + (wid + 8 * i) * 1024);
        };
        int a_off[4], b_off[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a_off[i] = (wm * 128 + i * 32 + frow) * ROWB;
#pragma unroll
        for (int j = 0; j < 2; ++j) b_off[j] = BM * ROWB + (wn * 64 + j * 32 + frow) * ROWB;

        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        if (nk > 0) issue_tile(0, 0);
        if (nk > 1) { issue_tile(1, 1); wait_vmcnt<G>(); } else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + (kt & 1) * BUF_BYTES;
            bf16x8 af[4][4], bfr[4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + coff);
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[ks][j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + coff);
            }
            // first half of the products while the second half of the fragments is still arriving
            if (p.dbg & 32) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
            if (p.dbg & 32) __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // nobody reads this buffer any more
            if (kt + 2 < nk) issue_tile(kt + 2, kt & 1);
            if (p.dbg & 32) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 2; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
            if (p.dbg & 32) __builtin_amdgcn_s_setprio(0);
            if (kt + 1 < nk) {
                if (kt + 2 < nk) wait_vmcnt<G>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();                   // k-tile kt+1 has landed for every wave
            }
        }
        // ---- epilogue: both buffers are idle (the last k-tile was read, nothing is in flight)
        __builtin_amdgcn_s_barrier();
        float* cw = reinterpret_cast<float*>(smem + wid * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    cw[row * 32 + (lane & 31)] = acc[i][j][r];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int lr = pass * 16 + (lane >> 2);
                    const int lc = (lane & 3) * 8;
                    const int row = m0 + wm * 128 + i * 32 + lr;
                    const int col = n0 + wn * 64 + j * 32 + lc;
                    const float4 x0 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc);
                    const float4 x1 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc + 4);
                    if (row < p.M && col < p.N && !(p.dbg & 2)) {
                        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        epilogue_store<EPI>(p, row, col, v);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        __builtin_amdgcn_s_barrier();                           // scratch is about to become a DMA target again
    }
}

int g_gemm_256 = 0;     // 0: never, 1: when the 256x256 tiling has at most one tile per CU and more than half of them, 2: always

template <int EPI>
int launch_256(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = 2 * 512 * 128;
    auto kern = gemm256_kernel<EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            pevit_set_error("hipFuncSetAttribute(gemm256 epi %d) failed", EPI);
            return -1;
        }
        attr_set = true;
    }
    const int tiles = ceil_div(p.M, 256) * ceil_div(p.N, 256);
    const int slots = num_cus();
    hipLaunchKernelGGL(kern, dim3(tiles < slots ? tiles : slots), dim3(512), lds, stream, p, tiles);
    return 0;
}

struct TileConfig { int bm, bn, bk, minb, sched; };
constexpr TileConfig kConfigs[] = {
    {128, 128, 64, 2, 0},   // 0: 64 KiB LDS, 2 workgroups / CU
    {128, 128, 32, 4, 0},   // 1: 32 KiB LDS, 4 workgroups / CU
    {128, 64, 64, 3, 0},    // 2: 48 KiB LDS, 3 workgroups / CU
    {64, 64, 64, 4, 0},     // 3: 32 KiB LDS, 4 workgroups / CU
    {64, 128, 64, 3, 0},    // 4: 48 KiB LDS, 3 workgroups / CU
    {128, 128, 64, 2, 1},   // 5: as 0, fragments hoisted
    {128, 128, 64, 2, 2},   // 6: as 5 + s_setprio around the MFMA block
    {64, 128, 64, 3, 1},    // 7: as 4, fragments hoisted
    {128, 128, 64, 4, 3},   // 8: ONE 32 KiB stage, 4 workgroups / CU: overlap comes from the other workgroups only
    {128, 128, 64, 2, 4},   // 9: register-resident k-tile, two k-tiles of LDS-DMA in flight per workgroup
};
constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);

template <int EPI, int CFG>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr TileConfig c = kConfigs[CFG];
    constexpr int stage = (c.bm + c.bn) * c.bk * 2;
    constexpr int nstage = c.sched == 3 ? 1 : 2;
    constexpr int lds = nstage * stage > 16384 ? nstage * stage : 16384;
    auto kern = gemm_bf16_nt_kernel<EPI, c.bm, c.bn, c.bk, c.minb, c.sched>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            pevit_set_error("hipFuncSetAttribute(gemm epi %d cfg %d) failed", EPI, CFG);
            return -1;
        }
        attr_set = true;
    }
    const int tiles = ceil_div(p.M, c.bm) * ceil_div(p.N, c.bn);
    // persistent grid: one workgroup per residency slot (a multiple of 8 keeps XCD affinity), or one
    // per tile when the tiles do not even fill the slots
    int grid = tiles;
    if (g_gemm_persistent) {
        const int slots = num_cus() * c.minb;
        if (tiles > slots) grid = slots;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p, tiles);
    return 0;
}

// Tile shape per problem.  Measured on MI355X (scripts/bench_gemm.py): with M = 6400 the large-N
// products want the 128x128 tile, the N = 768 products want 4x as many, smaller workgroups.
int pick_config(const GemmParams& p) {
    if (g_gemm_config >= 0 && g_gemm_config < kNumConfigs) return g_gemm_config;
    if (p.N <= 64) return 3;                      // bottleneck products: 64-wide tiles
    const long t128 = (long)ceil_div(p.M, 128) * ceil_div(p.N, 128);
    const bool hoist = g_gemm_hoist != 0;
    if (t128 >= 700 || p.K >= g_gemm_kswitch) return hoist ? 5 : 0;
    return hoist ? 7 : 4;
}

template <int EPI>
int launch_epi(const GemmParams& p, hipStream_t stream) {
    if (g_gemm_256 && g_gemm_config < 0 && p.N > 64) {
        const long t256 = (long)ceil_div(p.M, 256) * ceil_div(p.N, 256);
        if (g_gemm_256 == 2 || (t256 <= num_cus() && 2 * t256 > num_cus())) return launch_256<EPI>(p, stream);
    }
    if (g_gemm_ring && g_gemm_config < 0 && p.N > 64) {
        if (g_gemm_ring == 5) return launch_ring<EPI, 128, 32, 5, 2>(p, stream);
        const long t128 = (long)ceil_div(p.M, 128) * ceil_div(p.N, 128), t160 = (long)ceil_div(p.M, 160) * ceil_div(p.N, 128);
        if (g_gemm_ring == 2 || (t128 > num_cus() && t160 <= num_cus())) return launch_ring<EPI, 160, 64, 3, 1>(p, stream);
    }
    switch (pick_config(p)) {
        case 0: return launch_cfg<EPI, 0>(p, stream);
        case 1: return launch_cfg<EPI, 1>(p, stream);
        case 2: return launch_cfg<EPI, 2>(p, stream);
        case 3: return launch_cfg<EPI, 3>(p, stream);
        case 4: return launch_cfg<EPI, 4>(p, stream);
        case 5: return launch_cfg<EPI, 5>(p, stream);
        case 6: return launch_cfg<EPI, 6>(p, stream);
        case 8: return launch_cfg<EPI, 8>(p, stream);
        case 9: return launch_cfg<EPI, 9>(p, stream);
        default: return launch_cfg<EPI, 7>(p, stream);
    }
}

}  // namespace

int pevit_gemm_set_variant(int v) { const int old = g_gemm_config; g_gemm_config = v; return old; }
int pevit_gemm_set_hoist(int v) { const int old = g_gemm_hoist; g_gemm_hoist = v; return old; }
int pevit_gemm_set_ablate(int v) { const int old = g_gemm_ablate; g_gemm_ablate = v; return old; }
int pevit_gemm_set_dephase(int v) { const int old = g_gemm_dephase; g_gemm_dephase = v; return old; }
int pevit_gemm_set_256(int v) { const int old = g_gemm_256; g_gemm_256 = v; return old; }
int pevit_gemm_set_kswitch(int v) { const int old = g_gemm_kswitch; g_gemm_kswitch = v; return old; }
int pevit_gemm_set_ring(int v) { const int old = g_gemm_ring; g_gemm_ring = v; return old; }
int pevit_gemm_set_persistent(int v) { const int old = g_gemm_persistent; g_gemm_persistent = v; return old; }

int pevit_launch_gemm(int epi, const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    p.dbg = g_gemm_ablate;
    p.dephase = g_gemm_dephase;
    if (p.K % 64 != 0 || p.K <= 0) { pevit_set_error("gemm: K=%d must be a positive multiple of 64", p.K); return -1; }
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
        case EPI_BIAS_RESID_KEEP: return launch_epi<EPI_BIAS_RESID_KEEP>(p, stream);
        case EPI_BIAS_GELUNEW: return launch_epi<EPI_BIAS_GELUNEW>(p, stream);
        case EPI_DRELU_BF16: return launch_epi<EPI_DRELU_BF16>(p, stream);
        case EPI_DGELUNEW_BF16: return launch_epi<EPI_DGELUNEW_BF16>(p, stream);
    }
    pevit_set_error("gemm: unknown epilogue %d", epi);
    return -1;
}
