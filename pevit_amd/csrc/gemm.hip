// bf16 x {bf16 | fp8-e4m3} -> f32 "NT" GEMM on the CDNA4 matrix cores with fused epilogues.
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
// One kernel template, two workgroup geometries:
//   * 4 waves (2x2), tiles 64x64 .. 128x128, 2-4 workgroups per CU: the many-small-tiles regime
//     (N = 768 products of the ViT-B step: 300-600 tiles for 256 CUs);
//   * 8 waves (2x4 / 4x2), tiles 256x128 .. 320x256, ONE workgroup per CU: every k-tile moves
//     (BM+BN)*128 bytes through the CU's L1 for 2*BM*BN*64 flops, half the bytes per flop of the
//     128x128 tile.  The L2 -> LDS operand stream (~23 B/clk/CU, profiles/NOTES_gemm.md (r01_l2_fetch_bound)) is
//     what bounds these GEMMs, so the large tile is the lever wherever the tile count still fills
//     the chip (N >= 2304 at M = 6400).
// Operands are staged HBM->LDS with 16-byte LDS-DMA (global_load_lds), double buffered,
// XOR-swizzled on the *source* side so that the ds_read_b128 fragment reads are bank-conflict
// free (cdna guide T2 / rule 21).  v_mfma_f32_32x32x16_bf16 throughout.
//
// fp8 weights (BASELINE config 5): B holds OCP e4m3 codes with one power-of-two scale per output
// channel.  A k-tile of B is then 64 BYTES per row, half a cache line, so B is staged in
// 128-byte rows = TWO k-tiles at a time (every other iteration; same piece geometry, swizzle and
// LDS footprint as the bf16 path, half the bytes per flop).  The codes are stored k-permuted
// inside every 128-element group (fp8_kperm below) so that one lane's fragments of a k-tile are
// 32 contiguous bytes: two ds_read_b128, then v_cvt_scalef32_pk_bf16_fp8 (exact: e4m3 is a
// subset of bf16) feeds the same bf16 MFMA.  The channel scale multiplies the f32 accumulator in
// the epilogue, which is exact for powers of two: the fp8 path is BIT-IDENTICAL to the bf16 path
// run on the de-quantised weights (tests/test_gpu_fp8.py).
//
// Epilogue: accumulators are transposed through LDS, one 32x32 fragment per wave at a time, so
// that global accesses are 16/32-byte-per-lane row segments.
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace {
thread_local int g_last_path = 0;   // which kernel family the last pevit_launch_gemm OF THIS THREAD took (pevit_debug_last_gemm_path; tests): contexts driven from different threads do not share it

// measurement bits of GemmParams::dbg; -DGEMM_NO_DBG compiles them out
#ifdef GEMM_NO_DBG
#define GEMM_DBG(p, bit) false
#else
#define GEMM_DBG(p, bit) ((p).dbg & (bit))
#endif

#ifndef GEMM8_ABLATE
#define GEMM8_ABLATE 0   // measurement builds of gemm8_kernel (scripts/build_variants.sh): 4 = no operand stream, 8 = no ds_read / MFMA
#endif

constexpr int TILE_BAND = 6;

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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// tile coordinates of this workgroup.  Order: XCD-contiguous (xcd_remap), and inside that a band
// of TILE_BAND m-tiles is walked n-major, so the tiles an XCD has in flight share TILE_BAND
// A-panels and only a few B-panels (a 128x768 bf16 panel is 192 KiB; the XCD's L2 is 4 MiB).
// GemmParams::band overrides the band height: a one-round launch whose tiles split evenly over the 8 XCDs gives every XCD
// whole bands, so that no A panel is fetched by two private L2s (xcd_band below).
template <int BM, int BN>
__device__ __forceinline__ void tile_origin(const GemmParams& p, int tile, int& m0, int& n0) {
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int t = xcd_remap(tile, tiles_m * tiles_n);
    const int tb = p.band > 0 ? p.band : TILE_BAND;
    const int band = t / (tb * tiles_n), within = t - band * (tb * tiles_n);
    const int mb = min(tb, tiles_m - band * tb);
    const int tn = within / mb, tm = band * tb + (within - tn * mb);
    m0 = tm * BM; n0 = tn * BN;
}

// 8 fp8 codes (two dwords) -> 8 bf16, exact
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(int lo, int hi) {
    const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false);
    const bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false);
    const bf16x2 d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    bf16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1]; o[4] = c[0]; o[5] = c[1]; o[6] = d[0]; o[7] = d[1];
    return o;
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// Persistent form: gridDim.x workgroups walk the tiles (tile = blockIdx.x, += gridDim.x; gridDim.x
// is a multiple of 8 so a workgroup's tiles stay on one XCD range).  The first k-tile of the NEXT
// output tile is requested (LDS-DMA into stage 0) before the epilogue of the current one runs out of
// stage 1, so the HBM/L2 latency of the prologue -- one of only 12 k-iterations when K = 768 -- is
// hidden behind the epilogue's LDS transposes and global stores.
//
// WGM x WGN waves, each owning WM x WN fragments of 32x32.  MINW = waves per SIMD the register
// allocation must allow (= workgroups per CU x waves per workgroup / 4).
// LDS: [A stage 0][B stage 0][A stage 1][B stage 1], rows of 128 bytes (64 bf16 / 128 fp8).
template <int EPI, int WGM, int WGN, int WM, int WN, int MINW, bool HOIST, bool BF8, bool SPREAD, bool PIPE>
__global__ __launch_bounds__(WGM * WGN * 64, MINW) void gemm_kernel(GemmParams p, int ntiles) {
    constexpr int NW = WGM * WGN;
    constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32, BK = 64;
    constexpr int ROWB = 128, CH = 8, RPP = 8;          // bytes per LDS row, 16-byte chunks per row, rows per 1 KiB piece
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW;   // 1 KiB pieces per wave
    constexpr int KS = BK / 16;                         // MFMA k-steps per k-tile
    static_assert(PA * RPP * NW == BM && PB * RPP * NW == BN, "tile rows must split evenly over the loader waves");
    static_assert(STAGE_BYTES >= NW * 4096, "the epilogue borrows 4 KiB per wave of stage 1");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / WGN, wn = wid % WGN;

    // LDS-DMA: a piece is 8 rows x 128 bytes = 1 KiB, lane-linear; lane l lands at row R + l/8,
    // physical chunk l%8, and fetches the logical chunk (l%8) ^ swz(row) from HBM.
    const char* a_src[PA];
    const char* b_src[PB];
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = (wid * PA + i) * RPP + lane / CH;
            const int chunk = (lane % CH) ^ ((row >> 1) & (CH - 1));
            int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
            a_src[i] = reinterpret_cast<const char*>(p.A + (size_t)ar * p.lda) + chunk * 16;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int row = (wid * PB + i) * RPP + lane / CH;
            const int chunk = (lane % CH) ^ ((row >> 1) & (CH - 1));
            int br = n0 + row; br = br < p.Nb ? br : p.Nb - 1;
            b_src[i] = reinterpret_cast<const char*>(p.B) + (size_t)br * p.ldb * (BF8 ? 1 : 2) + chunk * 16;
        }
    };
    // k-tile kt of A (and of a bf16 B) -> stage kt&1.  fp8 B: rows of 128 codes = k-tiles 2j, 2j+1 -> B stage j&1,
    // requested together with the even k-tile of A.  Piece q of this wave: q < PA -> A, else B.
    auto issue_piece = [&](int kt, int q) {
        if (GEMM_DBG(p, 4)) return;                              // measurement only: no operand stream
        if (q < PA) {
            glds16(a_src[q] + kt * 128, smem + (kt & 1) * STAGE_BYTES + (wid * PA + q) * 1024);
        } else if constexpr (BF8) {
            if (!(kt & 1))
                glds16(b_src[q - PA] + (kt >> 1) * 128, smem + ((kt >> 1) & 1) * STAGE_BYTES + A_BYTES + (wid * PB + q - PA) * 1024);
        } else {
            glds16(b_src[q - PA] + kt * 128, smem + (kt & 1) * STAGE_BYTES + A_BYTES + (wid * PB + q - PA) * 1024);
        }
    };
    auto issue_tile = [&](int kt) {
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) issue_piece(kt, q);
    };
    // 32x32x16 bf16 fragment: lane l holds row (l&31), k = 8*(l>>5)..+7 of the 16-wide k-step.
    const int frow = lane & 31, fswz = (frow >> 1) & (CH - 1), fhalf = lane >> 5;
    int a_off[WM], b_off[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a_off[i] = (wm * WM * 32 + i * 32 + frow) * ROWB;
#pragma unroll
    for (int j = 0; j < WN; ++j) b_off[j] = A_BYTES + (wn * WN * 32 + j * 32 + frow) * ROWB;

    const int nk = GEMM_DBG(p, 1) ? 0 : p.K / BK;
    int tile = blockIdx.x;
    int m0, n0;
    tile_origin<BM, BN>(p, tile, m0, n0);
    set_sources(m0, n0);
    if (nk > 0) issue_tile(0);
    while (true) {
        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        // One k-tile.  The LDS-DMA requests of k-tile kt+1 are SPREAD between the MFMAs of k-tile kt (one 1 KiB piece
        // per group of WN MFMAs): a wave issues in order, and a CU accepts only ~64 outstanding 128-byte requests, so
        // a burst of PA+PB requests at the top of the iteration parks every wave in its issue slot until the burst has
        // drained -- the k-loop then runs load + compute instead of max(load, compute) (measured: 320x256 tile,
        // 5200 clk per k-tile against 2560 of MFMA and ~3100 of stream; profiles/NOTES_gemm.md (r02_gemm_experiments)).
        auto k_tile = [&](int kt, auto issue_next) {
            constexpr bool ISSUE = decltype(issue_next)::value;
            constexpr int NG = KS * WM, NP = PA + PB;       // MFMA groups per k-tile, pieces per wave
            // ... and only between the groups of the FIRST HALF of the k-tile: a request issued late in the iteration is
            // still in flight at the top of the next one (the windows 3 ... 10 of 20 groups all measure +1.0-1.4 % per step
            // over the full spread; profiles/NOTES_gemm.md (r02_gemm_experiments) section 16)
            constexpr int SPREAD_NG = NG / 2 > 0 ? NG / 2 : 1;
#ifdef GEMM_STREAM_FREE   // measurement build only (scripts/build_variants.sh): the stream-only ablation issues without waiting
            if (!GEMM_DBG(p, 8)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#endif
            if constexpr (ISSUE && !SPREAD) issue_tile(kt + 1);
            const char* sa = smem + (kt & 1) * STAGE_BYTES;
            const char* sb = smem + ((BF8 ? (kt >> 1) : kt) & 1) * STAGE_BYTES;
            if (GEMM_DBG(p, 8)) {                                // measurement only: operand stream without ds_read / MFMA
                if constexpr (ISSUE) issue_tile(kt + 1);
                return;
            }
            // fp8: this lane's 32 codes of the k-tile = logical chunks 4*par + 2*fhalf + {0, 1} of its row
            i32x4 braw[WN][2];
            if constexpr (BF8) {
                const int c0 = (kt & 1) * 4 + fhalf * 2;
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    braw[j][0] = *reinterpret_cast<const i32x4*>(sb + b_off[j] + ((c0 ^ fswz) << 4));
                    braw[j][1] = *reinterpret_cast<const i32x4*>(sb + b_off[j] + (((c0 + 1) ^ fswz) << 4));
                }
            }
            if constexpr (HOIST) {
                // all fragments of the k-tile are read up front (KS*(WM+WN) ds_read_b128), then the MFMAs run back to back
                bf16x8 af[KS][WM], bfr[KS][WN];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
#pragma unroll
                    for (int i = 0; i < WM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(sa + a_off[i] + coff);
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        if constexpr (BF8) bfr[ks][j] = fp8x8_to_bf16(braw[j][ks >> 1][(ks & 1) * 2], braw[j][ks >> 1][(ks & 1) * 2 + 1]);
                        else bfr[ks][j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j] + coff);
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < WM; ++i) {
                        if constexpr (ISSUE && SPREAD) {
#pragma unroll
                            for (int q = 0; q < NP; ++q)
                                if (q * SPREAD_NG / NP == ks * WM + i) issue_piece(kt + 1, q);
                        }
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
                    bf16x8 af[WM], bfr[WN];
#pragma unroll
                    for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + a_off[i] + coff);
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        if constexpr (BF8) bfr[j] = fp8x8_to_bf16(braw[j][ks >> 1][(ks & 1) * 2], braw[j][ks >> 1][(ks & 1) * 2 + 1]);
                        else bfr[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j] + coff);
                    }
#pragma unroll
                    for (int i = 0; i < WM; ++i) {
                        if constexpr (ISSUE && SPREAD) {
#pragma unroll
                            for (int q = 0; q < NP; ++q)
                                if (q * SPREAD_NG / NP == ks * WM + i) issue_piece(kt + 1, q);
                        }
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        };
        if constexpr (PIPE) {
            // Software-pipelined k-loop (bf16 B, hoisted fragments): the fragments of k-tile t+1 are read from LDS into a
            // SECOND register set while the MFMAs of k-tile t run from the first, so neither the ds_read latency nor the
            // LDS bandwidth of a k-tile sits in front of its MFMAs any more -- measured "compute only" the plain loop
            // reaches 39 % of the MFMA peak on the step's shapes, its waves alternate between a read phase and an MFMA
            // phase in lock step with the workgroup barrier.  Because a k-tile's LDS stage is free as soon as its
            // fragments are in registers, the LDS-DMA of k-tile t+2 is requested at the top of iteration t: two k-tiles
            // of operand stream in flight with two stages.
            //   top of iteration t:  DMA(t+1) landed (vmcnt), own reads(t) landed (lgkmcnt) -> barrier
            //                        -> request DMA(t+2) into stage t&1 -> reads(t+1) -> MFMA(t)
            static_assert(HOIST && !BF8, "the pipelined loop is built for hoisted bf16 fragments");
            constexpr int G = PA + PB;
            bf16x8 afA[KS][WM], bfA[KS][WN], afB[KS][WM], bfB[KS][WN];
            auto read_frags = [&](int kt, bf16x8 (&af)[KS][WM], bf16x8 (&bf)[KS][WN]) {
                const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
#pragma unroll
                    for (int i = 0; i < WM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + coff);
#pragma unroll
                    for (int j = 0; j < WN; ++j) bf[ks][j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + coff);
                }
            };
            auto mfma_all = [&](bf16x8 (&af)[KS][WM], bf16x8 (&bf)[KS][WN]) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
            };
            auto step = [&](int t, bf16x8 (&caf)[KS][WM], bf16x8 (&cbf)[KS][WN], bf16x8 (&naf)[KS][WM], bf16x8 (&nbf)[KS][WN]) {
                // DMA(t+1) and this wave's reads(t) have landed.  The builtin (not inline asm) so that the compiler's own
                // wait-count bookkeeping knows it: otherwise it protects the MFMAs of k-tile t with an lgkmcnt(0) placed
                // AFTER the reads of k-tile t+1 were issued, which serialises them again
                __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0) expcnt(7) lgkmcnt(0)
                __builtin_amdgcn_s_barrier();                                    // ... for every wave: stage t&1 is free
                if (t + 2 < nk) issue_tile(t + 2);
                if (t + 1 < nk) read_frags(t + 1, naf, nbf);
                mfma_all(caf, cbf);
            };
            if (nk > 0) {
                __syncthreads();                                // the previous epilogue no longer uses stage 1 as scratch
                if (nk > 1) { issue_tile(1); wait_vmcnt<G>(); } else wait_vmcnt<0>();    // k-tile 0 landed; k-tile 1 may fly
                __builtin_amdgcn_s_barrier();
                read_frags(0, afA, bfA);
                for (int kt = 0; kt < nk; kt += 2) {
                    step(kt, afA, bfA, afB, bfB);
                    if (kt + 1 < nk) step(kt + 1, afB, bfB, afA, bfA);
                }
            }
        } else {
            for (int kt = 0; kt + 1 < nk; ++kt) k_tile(kt, std::true_type{});
            if (nk > 0) k_tile(nk - 1, std::false_type{});
        }
        // both stages are idle after this barrier: stage 0 receives the next tile's first k-tile while
        // the epilogue transposes through (this wave's 4 KiB of) stage 1
        __syncthreads();
        const int cm0 = m0, cn0 = n0;
        const int next = tile + gridDim.x;
        if (next < ntiles) {
            tile_origin<BM, BN>(p, next, m0, n0);
            set_sources(m0, n0);
            if (nk > 0) issue_tile(0);
        }
        float* cw = reinterpret_cast<float*>(smem + STAGE_BYTES + wid * 4096);
        // column constants (bias; dGELU / fp8: channel scales) of this lane's 8 columns per fragment column j, loaded ONCE, and the
        // per-element operand (residual, positional embedding, saved activation) requested one fragment AHEAD, both from clamped
        // addresses: no load between two stores, none inside a bounds branch (gemm_epilogue.h, epilogue_store_full)
        float cc[WN][8], bs[WN][8];
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = min(cn0 + wn * WN * 32 + j * 32 + (lane & 3) * 8, p.N - 8);
            epi_cols<EPI>(p, col, cc[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) bs[j][e] = 1.f;
            if constexpr (BF8) {
                if (p.bscale) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.bscale + col), b1 = *reinterpret_cast<const float4*>(p.bscale + col + 4);
                    bs[j][0] = b0.x; bs[j][1] = b0.y; bs[j][2] = b0.z; bs[j][3] = b0.w; bs[j][4] = b1.x; bs[j][5] = b1.y; bs[j][6] = b1.z; bs[j][7] = b1.w;
                }
            }
        }
        constexpr bool OPND = epi_reads_resid<EPI> || epi_reads_aux<EPI>;
        EpiOperand opc[2], opn[2];
        auto prefetch = [&](int f, EpiOperand (&o)[2]) {
            if constexpr (OPND) {
                const int i = f / WN, j = f - i * WN;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int row = min(cm0 + wm * WM * 32 + i * 32 + pass * 16 + (lane >> 2), p.M - 1);
                    const int col = min(cn0 + wn * WN * 32 + j * 32 + (lane & 3) * 8, p.N - 8);
                    epi_prefetch<EPI, bf16>(p, row, col, o[pass]);
                }
            }
        };
        prefetch(0, opc);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    cw[row * 32 + (lane & 31)] = acc[i][j][r];
                }
                if (i * WN + j + 1 < WM * WN) prefetch(i * WN + j + 1, opn);      // before this fragment's stores
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int lr = pass * 16 + (lane >> 2);
                    const int lc = (lane & 3) * 8;
                    const int row = cm0 + wm * WM * 32 + i * 32 + lr;
                    const int col = cn0 + wn * WN * 32 + j * 32 + lc;
                    const float4 x0 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc);
                    const float4 x1 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc + 4);
                    float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    if constexpr (BF8) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= bs[j][e];
                    }
                    if (row < p.M && col < p.N && !GEMM_DBG(p, 2)) epilogue_store_full<EPI, bf16>(p, row, col, v, cc[j], opc[pass]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if constexpr (OPND) { opc[0] = opn[0]; opc[1] = opn[1]; }
            }
        if (next >= ntiles) break;
        tile = next;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 8-wave tiles (2 x 4 waves of WM x 2 fragments: 256x256 / 320x256, one workgroup per CU, bf16 B) with a STAGGERED
// two-group schedule (round 3).  In gemm_kernel above hipcc serialises every pair of MFMAs behind a ds_read_b128 it has just
// issued (one fragment register set, re-used: the 160 accumulator registers leave room for nothing else), all eight waves
// cross the k-tile barrier together, read together and compute together: "compute only" that loop keeps the matrix pipe 46 %
// busy (profiles/NOTES_gemm.md (r02_gemm_experiments) section 1).  Here a k-tile is cut into phases of KSP k-steps; in a phase a wave
//     LOAD : reads ALL its fragments of the phase (KSP * (WM + 2) ds_read_b128), requests its share of the next k-tile
//            (LDS-DMA), waits for its reads                                          -> s_barrier
//     MFMA : KSP * WM * 2 MFMAs back to back at raised priority                      -> s_barrier
// and the upper half of the workgroup (waves 4-7, the partners of waves 0-3 on the four SIMDs) runs ONE barrier behind the
// lower half: between two barriers one wave of every SIMD is in its MFMA section while its partner reads and requests, so the
// matrix pipe always has a wave whose operands are already in registers (cdna guide 5.5 T3/T5: role split + setprio).
// LDS hazards, by barrier count (2 stages; interval numbering: group 0 LOADs in even intervals, group 1 in odd ones):
//   * WAR: every LOAD section waits for its own ds_reads BEFORE its barrier, so a stage is free for the DMA of the k-tile
//     after next as soon as the last reader has passed that barrier;
//   * RAW: a wave's requests for k-tile t+1 go out in its LOAD sections of k-tile t (never in the last one) and every wave
//     waits for its own requests before the barrier that precedes group 0's first LOAD of k-tile t+1 (group 0: after its last
//     MFMA section of k-tile t; group 1: at the end of its last LOAD section).
// Operands are requested with buffer_load ... lds: one SGPR descriptor per operand, a 32-bit byte offset per 1 KiB piece and
// the k-tile offset in an SGPR, instead of nine 64-bit pointers per lane (which spilled: 256 VGPRs, 44 bytes of scratch).
// Same MFMA order per accumulator as gemm_kernel: bit-identical results.
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// 16 bytes per lane, buffer form of the LDS-DMA: descriptor {base, bytes} + per-lane byte offset + scalar byte offset.
// Kept in a __device__ function: written inline in the kernel (or in a lambda of it) the HOST pass of hipcc silently drops the
// whole kernel stub (a deferred diagnostic on the builtin's operands that is never printed) and the library fails to load
// with an undefined symbol.
__device__ __forceinline__ void buffer_lds16(const void* base, int bytes, char* lds_wave_base, int voff, int soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)lds_wave_base, 16, voff, soff, 0, 0);
}

// OPS: 0 = bf16 x bf16; 1 = bf16 x fp8 weights converted to bf16 (bit-identical to 0 on the de-quantised weights);
// 2 = fp8 x fp8 on the MX-scaled matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales: twice the
// bf16 rate, half the operand bytes): A holds e4m3 codes too, written k-permuted like the weights by its producer; a
// k-tile is then 128 codes = two 64-wide MFMA k-steps in the same 128-byte LDS rows.
template <int EPI, int WGM, int WGN, int WM, int WN, int KSP, int OPS>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmParams p, int ntiles) {
    constexpr int NW = 8;
    constexpr bool BF8 = OPS == 1, F8A = OPS == 2;
    static_assert(WGM * WGN == NW && (WN == 1 || WN == 2), "eight waves; one or two fragment columns per wave");
    constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32, BK = F8A ? 128 : 64, KSTEP = F8A ? 64 : 16;
    constexpr int ROWB = 128, CH = 8, RPP = 8;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA_T = BM / RPP, NPT = (BM + BN) / RPP, NP = (NPT + NW - 1) / NW;   // A pieces, pieces per k-tile, per wave
    constexpr int KS = BK / KSTEP, NPH = KS / KSP;       // MFMA k-steps per k-tile, phases per k-tile
    constexpr int ISSUE_PH = NPH > 1 ? NPH - 1 : 1;     // phases whose LOAD section carries LDS-DMA requests
    static_assert(BM % RPP == 0 && BN % RPP == 0 && NPH * KSP == KS, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wid >> 2, wm = wid / WGN, wn = wid % WGN;      // stagger groups: waves 0-3 / 4-7, whatever the wave grid

    // (rows - 1) * pitch + K elements are readable; the launcher has checked that both fit 31 bits
    // fp8 B (BF8): rows of 128 codes = TWO k-tiles, requested with the even k-tile into B stage (kt >> 1) & 1 (as in gemm_kernel);
    // the conversion to bf16 sits in the LOAD section, i.e. beside the partner wave's MFMA section
    const int a_bytes = ((p.M - 1) * p.lda + p.K) * (F8A ? 1 : 2), b_bytes = ((p.Nb - 1) * p.ldb + p.K) * (BF8 || F8A ? 1 : 2);
    // fp8 B with a bf16 tail (GemmParams::B2): the column tiles at n0 >= n_fp8 stream bf16 rows of B2, one k-tile per stage like
    // A and like the bf16 kernel (OPS = 0: same requests, same fragment reads, same MFMA order -- bit-identical to it)
    // (B2 is addressed through B's descriptor, as the byte offset b2_off from B: a second descriptor selected per tile made hipcc
    // keep both in scratch memory -- 296 bytes per lane -- and every fp8 product of ViT-L/14 lost 12 %)
    const int n_fp8 = p.n_fp8, Nb2 = p.Nb2, ldb2 = p.ldb2;
    const bool mixed = BF8 && EPI == EPI_QKV_HEADS && p.B2 != nullptr;
    const int b2_off = mixed ? (int)(reinterpret_cast<const char*>(p.B2) - reinterpret_cast<const char*>(p.B)) : 0;
    const int bdesc_bytes = mixed ? b2_off + ((Nb2 - 1) * ldb2 + p.K) * 2 : b_bytes;
    bool panel = false;                                       // the tile whose sources are set is a bf16-tail tile (workgroup-uniform)
    // 1 KiB pieces (8 rows x 128 bytes) of a k-tile: [0, PA_T) = A, then B; wave w requests the pieces w, w + 8, ... (a tile
    // whose piece count is no multiple of 8 -- 160x256: 52 -- leaves the last round to the first waves)
    int voff[NP];
    auto set_sources = [&](int m0, int n0) {
        if constexpr (BF8) panel = mixed && n0 >= n_fp8;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = i * NW + wid;                       // wave-uniform
            const bool is_a = q < PA_T;
            const int rb = is_a ? q : q - PA_T;
            const int row = rb * RPP + lane / CH;
            const int chunk = (lane % CH) ^ ((row >> 1) & (CH - 1));
            int r = (is_a ? m0 : n0) + row;
            const int lim = is_a ? p.M : p.Nb;
            r = r < lim ? r : lim - 1;
            voff[i] = r * (is_a ? p.lda * (F8A ? 1 : 2) : p.ldb * (BF8 || F8A ? 1 : 2)) + chunk * 16;
            if constexpr (BF8) {
                if (panel && !is_a) {
                    const int r2 = min(n0 - n_fp8 + row, Nb2 - 1);
                    voff[i] = b2_off + r2 * ldb2 * 2 + chunk * 16;
                }
            }
        }
    };
    auto issue_piece = [&](int kt, int i) {
#if GEMM8_ABLATE != 4   // measurement builds: 4 = no operand stream, 8 = no ds_read / MFMA
        const int q = i * NW + wid;
        if ((NPT % NW) && q >= NPT) return;
        char* dst = smem + (kt & 1) * STAGE_BYTES + q * 1024;
        if (q < PA_T) buffer_lds16(p.A, a_bytes, dst, voff[i], kt * 128);
        else if constexpr (!BF8) buffer_lds16(p.B, b_bytes, dst, voff[i], kt * 128);
        else {
            const int kk = panel ? kt : (kt >> 1);          // bf16 tail: a k-tile per stage; fp8 rows: two k-tiles, requested with the even one
            if (panel || !(kt & 1)) buffer_lds16(p.B, bdesc_bytes, smem + (kk & 1) * STAGE_BYTES + q * 1024, voff[i], kk * 128);
        }
#endif
    };
    const int frow = lane & 31, fswz = (frow >> 1) & (CH - 1), fhalf = lane >> 5;
    const int a_base = (wm * WM * 32 + frow) * ROWB, b_base = A_BYTES + (wn * WN * 32 + frow) * ROWB;
    int coff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) coff[ks] = ((ks * 2 + fhalf) ^ fswz) << 4;

    const int nk = GEMM_DBG(p, 1) ? 0 : p.K / BK;
    int tile = blockIdx.x;
    int m0, n0;
    tile_origin<BM, BN>(p, tile, m0, n0);
    set_sources(m0, n0);
    if (nk > 0) {
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_piece(0, q);
    }
    while (true) {
        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        // k-tile 0 has landed for every wave, and the previous epilogue no longer uses stage 1 as scratch
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (grp) __builtin_amdgcn_s_barrier();                      // the upper half runs one barrier behind
        // (Tried: one dword per 128-byte line of the dGELU epilogue's saved-activation block requested here, so that the
        // lines travel to the L2 under the k-loop.  In step the kernel got SLOWER, 42.4 vs 38.4 us: the requests compete
        // with the operand stream and the lines are evicted again before the epilogue; profiles/r03_gemm_epilogues.md.)
        // the k-loop, instantiated twice for fp8 B: PANEL = this tile streams the bf16 tail (B2) and reads bf16 fragments, exactly
        // the OPS = 0 loop; otherwise the fp8 loop -- chosen once per tile, no format test inside the loop (with the test inside,
        // every fp8 product of ViT-L/14 lost 12 %)
        auto kloop = [&](auto panel_c) __attribute__((always_inline)) {
        constexpr bool PANEL = decltype(panel_c)::value;
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + (kt & 1) * STAGE_BYTES;
            const bool more = kt + 1 < nk;
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                // ---- LOAD
                typedef typename std::conditional<F8A, i32x8, bf16x8>::type frag_t;
                frag_t af[KSP][WM], bfr[KSP][WN];
#if GEMM8_ABLATE != 8
                if constexpr (F8A) {
                    // lane (row, half): the 32 codes of k-step ks = chunks 4*ks + 2*half + {0, 1} of its row (k-permuted storage)
#pragma unroll
                    for (int s = 0; s < KSP; ++s) {
                        const int c0 = (ph * KSP + s) * 4 + fhalf * 2;
                        const int o0 = (c0 ^ fswz) << 4, o1 = ((c0 + 1) ^ fswz) << 4;
#pragma unroll
                        for (int j = 0; j < WN; ++j) {
                            const i32x4 lo = *reinterpret_cast<const i32x4*>(st + b_base + j * 32 * ROWB + o0);
                            const i32x4 hi = *reinterpret_cast<const i32x4*>(st + b_base + j * 32 * ROWB + o1);
                            bfr[s][j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        }
#pragma unroll
                        for (int i = 0; i < WM; ++i) {
                            const i32x4 lo = *reinterpret_cast<const i32x4*>(st + a_base + i * 32 * ROWB + o0);
                            const i32x4 hi = *reinterpret_cast<const i32x4*>(st + a_base + i * 32 * ROWB + o1);
                            af[s][i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        }
                    }
                } else {
                // fp8 B (round 5): the 8 codes of this lane for k-step ks are read where they are converted, one 8-byte read per
                // fragment column and k-step.  (Rounds 3-4 fetched the lane's 32 codes of the whole k-tile in phase 0 and kept them:
                // 16 more live registers pushed the 320x256 / 256x256 kernels to 256 VGPRs + 64 bytes of scratch in the k-loop, and
                // the fp8 form of every K = E product ran 4-7 us behind its bf16 form at ViT-L/14.)
#pragma unroll
                for (int s = 0; s < KSP; ++s) {
                    const int ks = ph * KSP + s;
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        if constexpr (BF8 && !PANEL) {
                            typedef __attribute__((ext_vector_type(2))) int i32x2;
                            const char* sb = smem + ((kt >> 1) & 1) * STAGE_BYTES;
                            const int c0 = (kt & 1) * 4 + fhalf * 2 + (ks >> 1);
                            const i32x2 raw = *reinterpret_cast<const i32x2*>(sb + b_base + j * 32 * ROWB + ((c0 ^ fswz) << 4) + (ks & 1) * 8);
                            bfr[s][j] = fp8x8_to_bf16(raw[0], raw[1]);
                        }
                        else bfr[s][j] = *reinterpret_cast<const bf16x8*>(st + b_base + j * 32 * ROWB + coff[ks]);
                    }
#pragma unroll
                    for (int i = 0; i < WM; ++i) af[s][i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 32 * ROWB + coff[ph * KSP + s]);
                }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (ph < ISSUE_PH && more) {
#pragma unroll
                    for (int q = ph * NP / ISSUE_PH; q < (ph + 1) * NP / ISSUE_PH; ++q) issue_piece(kt + 1, q);
                }
                if (ph == NPH - 1 && grp) wait_vmcnt<0>();
                __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this wave's reads are in registers
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- MFMA
#if GEMM8_ABLATE != 8
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < KSP; ++s)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                        {
                            if constexpr (F8A) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[s][i], bfr[s][j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i], bfr[s][j], acc[i][j], 0, 0, 0);
                        }
                __builtin_amdgcn_s_setprio(0);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (ph == NPH - 1 && !grp) wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
        }
        };
        if constexpr (BF8 && EPI == EPI_QKV_HEADS) { if (panel) kloop(std::true_type{}); else kloop(std::false_type{}); }   // only the QKV product has a tail
        else kloop(std::false_type{});
        if (!grp) __builtin_amdgcn_s_barrier();                     // ... and the lower half waits for it here
        // both stages are idle: stage 0 receives the next tile's first k-tile while the epilogue transposes through
        // (this wave's 8 KiB of) stage 1
        const int cm0 = m0, cn0 = n0;
        const bool scaled = (BF8 || F8A) && p.bscale && !panel;   // the bf16 tail carries no channel scales (captured before the next tile's sources are set)
        const int next = tile + gridDim.x;
        if (next < ntiles) {
            tile_origin<BM, BN>(p, next, m0, n0);
            set_sources(m0, n0);
            if (nk > 0) {
#pragma unroll
                for (int q = 0; q < NP; ++q) issue_piece(0, q);
            }
        }
        // Epilogue: a fragment ROW (32 x 64 = both fragments j) goes through this wave's 8 KiB of stage 1 and comes back as
        // whole output rows: 8 lanes cover the 64 columns of a row = one 128-byte line of a bf16 output (two of an f32 one),
        // 8 rows per pass.  (gemm_kernel's 32 x 32 transposes write half lines, 16 rows at a time.)  16-byte chunk c of row r
        // sits at chunk c ^ (r & 1): the ds_read_b128 lane groups then touch every bank once.
        // FAST path (this wave's 32*WM x 64 block lies inside the matrix, epilogue with column constants only): no branch
        // and no load between the stores -- the column constants are loaded once, the saved activation of the dGELU
        // epilogue one fragment row ahead -- so that every s_waitcnt the compiler places is an exact vmcnt(N).  With a
        // bounds branch around each access it falls back to vmcnt(0) before every use of a loaded value, and on gfx950
        // that also waits for all STORES issued so far: the dGELU epilogue then ran one store latency per pass, 24 us for
        // 78 MB in step (profiles/r03_gemm_epilogues.md).
        constexpr int COLS = WN * 32, LPR = COLS / 8, RPASS = 64 / LPR, NPASS = 32 / RPASS;   // lanes per row, rows per pass, passes
        float* cw = reinterpret_cast<float*>(smem + STAGE_BYTES + wid * (32 * COLS * 4));
        // 32 x COLS floats per wave behind stage 0: stage 1, and beyond it where a stage is smaller than that (launch_big8 sizes the LDS)
        const int erow = lane / LPR, ec8 = lane % LPR;                 // pass-local row, 8-column group of this lane
        const int gr0 = cm0 + wm * WM * 32, gc0 = cn0 + wn * WN * 32, gc = gc0 + ec8 * 8;
        auto to_lds = [&](int i) {
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int col = j * 32 + (lane & 31);
                    cw[row * COLS + ((((col >> 2) ^ (row & 1)) << 2) | (col & 3))] = acc[i][j][r];
                }
        };
        auto from_lds = [&](int ps, float (&v)[8]) {
            const int lr = ps * RPASS + erow;
            const float4 x0 = *reinterpret_cast<const float4*>(cw + lr * COLS + (((2 * ec8) ^ (lr & 1)) << 2));
            const float4 x1 = *reinterpret_cast<const float4*>(cw + lr * COLS + (((2 * ec8 + 1) ^ (lr & 1)) << 2));
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        };
        const bool inside = gr0 + WM * 32 <= p.M && gc0 + WN * 32 <= p.N;        // wave-uniform
        if (gc0 >= p.N || gr0 >= p.M || GEMM_DBG(p, 2)) {
            // nothing of this wave's block is inside the matrix (the 64 adapter columns of the QKV product fill a quarter
            // of their 256-column tile), or a measurement run without stores
        } else if (epi_has_pre<EPI> && inside) {
            if constexpr (epi_has_pre<EPI>) {
                float cc[8], bsc[8];
                epi_load_cols<EPI>(p, gc, cc);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsc[e] = 1.0f;
                if constexpr (BF8 || F8A) {
                    if (scaled) {
                        const float4 b0 = *reinterpret_cast<const float4*>(p.bscale + gc), b1 = *reinterpret_cast<const float4*>(p.bscale + gc + 4);
                        bsc[0] = b0.x; bsc[1] = b0.y; bsc[2] = b0.z; bsc[3] = b0.w; bsc[4] = b1.x; bsc[5] = b1.y; bsc[6] = b1.z; bsc[7] = b1.w;
                    }
                }
                bf16x8 aux_nxt[NPASS];
                auto load_aux = [&](int i) {
                    if constexpr (epi_reads_aux<EPI>) {
#pragma unroll
                        for (int ps = 0; ps < NPASS; ++ps)
                            aux_nxt[ps] = load_bf16x8(p.aux + (size_t)(gr0 + i * 32 + ps * RPASS + erow) * p.ldaux + gc);
                    }
                };
                load_aux(0);
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    to_lds(i);
                    float h[NPASS][8];
                    if constexpr (epi_reads_aux<EPI>) {
#pragma unroll
                        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
                            for (int e = 0; e < 8; ++e) h[ps][e] = bf2f(aux_nxt[ps][e]);
                        if (i + 1 < WM) load_aux(i + 1);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) {
                        float v[8];
                        from_lds(ps, v);
                        if constexpr (BF8 || F8A) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] *= bsc[e];
                        }
                        epilogue_store_pre<EPI, bf16>(p, gr0 + i * 32 + ps * RPASS + erow, gc, v, cc, h[ps]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                to_lds(i);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int row = gr0 + i * 32 + ps * RPASS + erow;
                    float v[8];
                    from_lds(ps, v);
                    if (row < p.M && gc < p.N) {
                        if constexpr (BF8 || F8A) { if (scaled) mul8(v, p.bscale + gc); }
                        epilogue_store<EPI, bf16>(p, row, gc, v);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (next >= ntiles) break;
        tile = next;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stream-K form of the 128x128 tile (4 waves, 2 workgroups per CU, bf16 B) for the few-tile long-K products: at ViT-B/32,
// B = 128 the N = E products have 300 tiles for 512 residency slots, so 44 CUs stream two whole tiles while 212 stream one,
// and the kernel takes as long as the 44 (54 us; a problem whose 512 tiles carry 28 k-iterations each, the same work per
// slot, takes 33 us -- scripts/gpu_streamk_bound.py).  Here the tiles x k-iterations space is cut into gridDim.x equal
// ranges, one per resident workgroup, walked in increasing order.  A range that starts inside a tile (k0 > 0) computes that
// tile's tail FIRST and publishes the partial accumulators (one f32 slab per workgroup); the workgroup that holds the
// tile's first iterations has them LAST in its own range, so by the time it has finished and polls, the partials it needs
// were published long ago: no dependency chains, no stalls beyond the hand-off itself.  It adds the slabs in ascending
// order (a fixed order: results are deterministic), runs the epilogue and clears the flags for the next launch.
//
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): write-through (sc1) 16-byte slab stores -> every wave drains
// vmcnt -> barrier -> relaxed agent-scope flag store; consumer: relaxed agent-scope poll by one lane (bounded: a lost
// producer turns into a wrong tile and a raised sk_flag[slots] error word, never into a hang) -> agent acquire -> barrier ->
// plain loads.  All gridDim.x workgroups must be resident at once: the launcher sizes the grid to the 2-per-CU residency.
__device__ __forceinline__ void sk_tile_origin(const GemmParams& p, int t, int& m0, int& n0) {
    const int tiles_n = (p.N + 127) / 128, tiles_m = (p.M + 127) / 128, tb = p.sk_band;
    const int band = t / (tb * tiles_n), within = t - band * (tb * tiles_n);
    const int mb = min(tb, tiles_m - band * tb);
    const int tn = within / mb, tm = band * tb + (within - tn * mb);
    m0 = tm * 128; n0 = tn * 128;
}

__device__ __forceinline__ void sk_store16_sc1(float* dst, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_streamk_kernel(GemmParams p, int ntiles) {
    constexpr int NW = 4, WGN = 2, WM = 2, WN = 2, BM = 128, BN = 128, BK = 64;
    constexpr int ROWB = 128, CH = 8, RPP = 8;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW, KS = BK / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const char* a_src[PA];
    const char* b_src[PB];
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = (wid * PA + i) * RPP + lane / CH;
            const int chunk = (lane % CH) ^ ((row >> 1) & (CH - 1));
            int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
            a_src[i] = reinterpret_cast<const char*>(p.A + (size_t)ar * p.lda) + chunk * 16;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int row = (wid * PB + i) * RPP + lane / CH;
            const int chunk = (lane % CH) ^ ((row >> 1) & (CH - 1));
            int br = n0 + row; br = br < p.Nb ? br : p.Nb - 1;
            b_src[i] = reinterpret_cast<const char*>(p.B) + (size_t)br * p.ldb * 2 + chunk * 16;
        }
    };
    auto issue_tile = [&](int kt, int st) {      // k-tile kt -> LDS stage st
#pragma unroll
        for (int q = 0; q < PA; ++q) glds16(a_src[q] + kt * 128, smem + st * STAGE_BYTES + (wid * PA + q) * 1024);
#pragma unroll
        for (int q = 0; q < PB; ++q) glds16(b_src[q] + kt * 128, smem + st * STAGE_BYTES + A_BYTES + (wid * PB + q) * 1024);
    };
    const int frow = lane & 31, fswz = (frow >> 1) & (CH - 1), fhalf = lane >> 5;
    int a_off[WM], b_off[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a_off[i] = (wm * WM * 32 + i * 32 + frow) * ROWB;
#pragma unroll
    for (int j = 0; j < WN; ++j) b_off[j] = A_BYTES + (wn * WN * 32 + j * 32 + frow) * ROWB;

    const int nk = p.K / BK;
    const int slots = gridDim.x;
    // block b runs on XCD b % 8: position = XCD-major, so that an XCD's workgroups walk consecutive tiles
    const int pos = (blockIdx.x & 7) * (slots >> 3) + (blockIdx.x >> 3);
    const long long total = (long long)ntiles * nk;
    auto range_start = [&](int q) -> long long { return min(total, (long long)q * p.sk_share); };
    long long it = range_start(pos);
    const long long it_end = range_start(pos + 1);
    if (it >= it_end) return;
    int tile = (int)(it / nk), k0 = (int)(it - (long long)tile * nk);
    int m0, n0;
    sk_tile_origin(p, tile, m0, n0);
    set_sources(m0, n0);
    issue_tile(k0, 0);
    while (true) {
        const int k1 = (int)min((long long)nk, k0 + (it_end - it));     // this segment: k-tiles [k0, k1) of `tile`
        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        for (int kt = k0; kt < k1; ++kt) {
            const int st = (kt - k0) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < k1) issue_tile(kt + 1, st ^ 1);
            const char* sa = smem + st * STAGE_BYTES;
            bf16x8 af[KS][WM], bfr[KS][WN];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
#pragma unroll
                for (int i = 0; i < WM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(sa + a_off[i] + coff);
#pragma unroll
                for (int j = 0; j < WN; ++j) bfr[ks][j] = *reinterpret_cast<const bf16x8*>(sa + b_off[j] + coff);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                   // both stages idle
        it += k1 - k0;
        const int ctile = tile, ck0 = k0, cm0 = m0, cn0 = n0;
        const bool more = it < it_end;
        if (more) {                                        // the rest of the range starts the next tile at k = 0
            tile = ctile + 1; k0 = 0;
            sk_tile_origin(p, tile, m0, n0);
            set_sources(m0, n0);
            issue_tile(0, 0);                              // lands in stage 0 while stage 1 serves the epilogue
        }
        if (ck0 > 0) {
            // ---- tail of a tile another workgroup started: publish the partial accumulators
            if (p.dbg & 16) { if (!more) break; continue; }   // measurement only: no hand-off
            float* slab = p.sk_slab + (size_t)pos * (BM * BN);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        sk_store16_sc1(slab + ((((wid * 4 + i * 2 + j) * 4 + q) * 64 + lane) << 2), v);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(p.sk_flag + pos, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (k1 < nk && !(p.dbg & 16)) {
                // ---- head of a tile: the workgroups pos+1, ... computed the rest, early in their ranges
                const long long tile_end = (long long)(ctile + 1) * nk;
                int np = 0;
                while (pos + 1 + np < slots && range_start(pos + 1 + np) < tile_end) ++np;
                if (threadIdx.x == 0) {
                    for (int pp = pos + 1; pp <= pos + np; ++pp) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(p.sk_flag + pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > (1u << 20)) {   // ~1 s: give up loudly instead of hanging the GPU
                                __hip_atomic_store(p.sk_flag + p.sk_slots, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                for (int pp = pos + 1; pp <= pos + np; ++pp) {
                    const float* slab = p.sk_slab + (size_t)pp * (BM * BN);
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + ((((wid * 4 + i * 2 + j) * 4 + q) * 64 + lane) << 2));
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[i][j][4 * q + r] += v[r];
                            }
                }
                __syncthreads();
                if (threadIdx.x == 0)
                    for (int pp = pos + 1; pp <= pos + np; ++pp)
                        __hip_atomic_store(p.sk_flag + pp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            float* cw = reinterpret_cast<float*>(smem + STAGE_BYTES + wid * 4096);
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
                        const int row = cm0 + wm * WM * 32 + i * 32 + lr;
                        const int col = cn0 + wn * WN * 32 + j * 32 + lc;
                        const float4 x0 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc);
                        const float4 x1 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc + 4);
                        if (row < p.M && col < p.N) {
                            float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                            epilogue_store<EPI, bf16>(p, row, col, v);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
        }
        if (!more) break;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Few-row long-K products (M <= 128, K >= 1536: c_proj forward and c_fc backward on the class-token rows of the last block):
// 6 tiles of 128x128 for 256 CUs.  Stream-K gives such a tile two workgroups that walk 24 k-tiles each behind a two-stage
// ring (31 us for 0.6 GFLOP), and is switched off under data parallelism (its consumers WAIT for their producers, which must
// therefore be resident).  Here: 128x64 tiles, the K range cut into a few slices of ~12 k-tiles, one workgroup per
// (tile, slice) with a FOUR-stage ring and exact vmcnt waits; the 128x64 f32 partial goes to a slab (write-through stores) and
// the workgroup that draws the LAST ticket of its tile adds the slabs in slice order -- a fixed order, so the result does not
// depend on who arrives last -- and runs the epilogue.  Nobody waits for anybody: no residency requirement.
// Bounds (measured, profiles/NOTES_gemm.md (r03_gemm_experiments) 5c): one CU moves ~60 GB/s, so both the k-walk of a workgroup and the
// slabs its tile's finisher re-reads must stay at a few hundred KB: 16 slices of 3 k-tiles made the finisher read 1 MB (27 us).
// Slabs and ticket words are the stream-K workspace (GemmParams::sk_slab / sk_flag; tickets return to 0).
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_skinny_kernel(GemmParams p, int tiles_n, int nslices, int ksl) {
    constexpr int NW = 4, WN = 2, BM = 128, BN = 64, BK = 64, S = 4;
    constexpr int ROWB = 128, CH = 8, RPP = 8;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW, KS = BK / 16, NPW = PA + PB;     // pieces per wave per k-tile
    constexpr int SLAB = BM * BN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned ticket;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave w: rows 32w .. 32w+31, all 64 columns
    const int tile = blockIdx.x / nslices, slice = blockIdx.x - tile * nslices;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = p.K / BK;
    const int k0 = slice * ksl, n = min(nk, k0 + ksl) - k0;               // this slice: k-tiles [k0, k0 + n)
    const char* a_src[PA];
    const char* b_src[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wid * PA + i) * RPP + lane / CH;
        const int chunk = (lane % CH) ^ ((row >> 1) & (CH - 1));
        int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
        a_src[i] = reinterpret_cast<const char*>(p.A + (size_t)ar * p.lda) + chunk * 16 + (size_t)k0 * 128;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = (wid * PB + i) * RPP + lane / CH;
        const int chunk = (lane % CH) ^ ((row >> 1) & (CH - 1));
        int br = n0 + row; br = br < p.Nb ? br : p.Nb - 1;
        b_src[i] = reinterpret_cast<const char*>(p.B) + (size_t)br * p.ldb * 2 + chunk * 16 + (size_t)k0 * 128;
    }
    auto issue_tile = [&](int kt) {
        char* st = smem + (kt & (S - 1)) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(a_src[i] + kt * 128, st + (wid * PA + i) * 1024);
#pragma unroll
        for (int i = 0; i < PB; ++i) glds16(b_src[i] + kt * 128, st + A_BYTES + (wid * PB + i) * 1024);
    };
#pragma unroll
    for (int kt = 0; kt < S - 1; ++kt)
        if (kt < n) issue_tile(kt);
    const int frow = lane & 31, fswz = (frow >> 1) & (CH - 1), fhalf = lane >> 5;
    f32x16 acc[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    for (int kt = 0; kt < n; ++kt) {
        // k-tile kt has landed when at most the younger requested ones (up to kt + S - 2) are in flight
        const int after = min(n - 1, kt + S - 2) - kt;
        if (after >= 2) wait_vmcnt<2 * NPW>(); else if (after == 1) wait_vmcnt<NPW>(); else wait_vmcnt<0>();
        __syncthreads();                                      // ... for every wave; and stage (kt - 1) % S is free again
        if (kt + S - 1 < n) issue_tile(kt + S - 1);
        const char* sa = smem + (kt & (S - 1)) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(sa + (wid * 32 + frow) * ROWB + coff);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(sa + A_BYTES + (j * 32 + frow) * ROWB + coff);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[j], 0, 0, 0);
            }
        }
    }
    if (nslices > 1) {
        // ---- publish this slice's partial, draw a ticket; the last arriver of the tile sums all slabs in slice order
        float* slab = p.sk_slab + (size_t)blockIdx.x * SLAB;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
                sk_store16_sc1(slab + ((((wid * WN + j) * 4 + q) * 64 + lane) << 2), v);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(p.sk_flag + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (ticket != (unsigned)(nslices - 1)) return;        // block-uniform
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0) __hip_atomic_store(p.sk_flag + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        f32x4 part[6][WN][4];                                 // all slabs requested before the first add
        const float* base = p.sk_slab + (size_t)tile * nslices * SLAB;
#pragma unroll
        for (int sl = 0; sl < 6; ++sl)
            if (sl < nslices) {
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        part[sl][j][q] = *reinterpret_cast<const f32x4*>(base + (size_t)sl * SLAB + ((((wid * WN + j) * 4 + q) * 64 + lane) << 2));
            }
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
        for (int sl = 0; sl < 6; ++sl)
            if (sl < nslices) {
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[j][4 * q + r] += part[sl][j][q][r];
            }
    }
    __syncthreads();                                          // every wave is past its LDS reads: stage 0 serves as epilogue scratch
    float* cw = reinterpret_cast<float*>(smem + wid * 4096);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            cw[row * 32 + (lane & 31)] = acc[j][r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int lr = pass * 16 + (lane >> 2);
            const int lc = (lane & 3) * 8;
            const int row = m0 + wid * 32 + lr;
            const int col = n0 + j * 32 + lc;
            const float4 x0 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc);
            const float4 x1 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc + 4);
            if (row < p.M && col < p.N) {
                float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                epilogue_store<EPI, bf16>(p, row, col, v);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// The end of a k-split tile, shared by gemm_ksplit_kernel and gemm_kphase_kernel: the two groups' partial sums meet through
// LDS (every wave is past its last k-tile and has no request in flight): fragment f = i * WN + j is FINISHED by group f % 2,
// which receives the other group's partial at hand[((gw * NF + f) * 16 + r) * 64 + lane] and runs that fragment's epilogue.
// SCRATCH_OFF: byte offset of the 4 KiB-per-wave epilogue scratch, behind the hand-off area.
// KZ (round 5): the tile's K range is split over TWO workgroups (slice z = 0, 1).  After the groups' hand-off each wave holds the
// workgroup's sums of the fragments it finishes; it publishes them in the stream-K workspace (slab tile * 2 + z, write-through),
// the workgroup draws a ticket, and the one that arrives LAST adds the other's slab and runs the epilogue -- nobody waits for
// anybody (no residency requirement), and with two slices the sum does not depend on who was last (a + b = b + a).  Returns false
// in the workgroup that arrived first (it has nothing left to do for this tile).
template <int EPI, int WM, int WN, int NWG, int NOWN, int SCRATCH_OFF, bool KZ = false>
__device__ __forceinline__ bool ksplit_finish(const GemmParams& p, char* smem, f32x16 (&acc)[WM][WN], float4 (&rpre)[NOWN][2][2],
                                              float (&bpre)[WN][8], int m0, int n0, int grp, int gw, int wm, int wn, int wid, int lane,
                                              int tile = 0, int z = 0) {
    constexpr bool PRE = (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_BIAS_RESID_KEEP);
    constexpr int NF = WM * WN;
    constexpr int NOWNF = (NF + 1) / 2;                          // fragments a wave finishes at most
    constexpr int SLABF = 2 * NWG * NOWNF * 16 * 64;            // floats a workgroup publishes
    float* hand = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
            if (((i * WN + j) & 1) != grp) {
#pragma unroll
                for (int r = 0; r < 16; ++r) hand[((gw * NF + i * WN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
            }
    __syncthreads();
    if constexpr (KZ) {
        __shared__ unsigned kz_ticket;
        float* mine = p.sk_slab + (size_t)(tile * 2 + z) * SLABF + (size_t)wid * NOWNF * 1024;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (((i * WN + j) & 1) != grp) continue;
                const int k = (i * WN + j) >> 1;                  // fragment 2k + grp is this wave's k-th (a compile-time index)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += hand[((gw * NF + i * WN + j) * 16 + r) * 64 + lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    sk_store16_sc1(mine + ((k * 4 + q) * 64 + lane) * 4, v);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) kz_ticket = __hip_atomic_fetch_add(p.sk_flag + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (kz_ticket != 1u) return false;                      // workgroup-uniform: the other slice finishes the tile
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0) __hip_atomic_store(p.sk_flag + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        const float* other = p.sk_slab + (size_t)(tile * 2 + (1 - z)) * SLABF + (size_t)wid * NOWNF * 1024;
        f32x4 part[NOWNF][4];                                    // every piece requested before the first add
#pragma unroll
        for (int kk = 0; kk < NOWNF; ++kk)
#pragma unroll
            for (int q = 0; q < 4; ++q) part[kk][q] = *reinterpret_cast<const f32x4*>(other + ((kk * 4 + q) * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (((i * WN + j) & 1) != grp) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][4 * q + r] += part[(i * WN + j) >> 1][q][r];
            }
    }
    static_assert(SCRATCH_OFF >= NWG * NF * 16 * 64 * 4, "epilogue scratch must not overlap the hand-off area");
    // 4 KiB of epilogue scratch per wave behind the hand-off area (launch_ksplit sizes the LDS for it)
    float* cw = reinterpret_cast<float*>(smem + SCRATCH_OFF + wid * 4096);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (((i * WN + j) & 1) != grp) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if constexpr (KZ) cw[row * 32 + (lane & 31)] = acc[i][j][r];      // (the other group's share went in above)
                else cw[row * 32 + (lane & 31)] = acc[i][j][r] + hand[((gw * NF + i * WN + j) * 16 + r) * 64 + lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int lr = pass * 16 + (lane >> 2);
                const int lc = (lane & 3) * 8;
                const int row = m0 + wm * WM * 32 + i * 32 + lr;
                const int col = n0 + wn * WN * 32 + j * 32 + lc;
                const float4 x0 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc);
                const float4 x1 = *reinterpret_cast<const float4*>(cw + lr * 32 + lc + 4);
                if (row < p.M && col < p.N) {
                    float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    if constexpr (PRE) {          // epilogue_store<EPI_BIAS_RESID_*> with the residual and the bias already in registers
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bpre[j][e];
                        if constexpr (EPI == EPI_BIAS_RESID_KEEP) store8f(p.outf2 + (size_t)row * p.ldo2 + col, v);
                        const float4 r0 = rpre[(i * WN + j) >> 1][pass][0], r1 = rpre[(i * WN + j) >> 1][pass][1];
                        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                        v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                        store8f(p.outf + (size_t)row * p.ldo + col, v);
                    } else {
                        epilogue_store<EPI, bf16>(p, row, col, v);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// One-tile-per-CU form for the N = E products: a 160x128 tile gives M = 6400, N = 768 exactly 240 tiles for 256 CUs (the
// vendor library picks the same shape class for these problems: MT128x160 / MT160x128, profiles/NOTES_gemm.md (r02_vendor_gemm_shapes)),
// so every CU streams the same 288 rows per k-tile -- 10 % fewer bytes than two 128x128 halves and no imbalance.  With one
// 4-wave workgroup per CU that tile ran at ~19 B/clk/CU of operand stream whatever the depth of the LDS ring (2, 3 or 4
// stages: 45.8 / 45.8 / 45.4 us for c_proj; profiles/NOTES_gemm.md (r02_gemm_experiments) section 15): the stream rate depends on how
// many waves issue requests.  Hence:
// The same tile with TWO wave groups that take alternate k-tiles (k-tile 2i -> group 0, 2i+1 -> group 1): twice the waves
// issue LDS-DMA requests and twice the bytes are in flight per CU, which is what the L2 -> LDS stream rate depends on
// (4-wave tiles measure ~19-25 B/clk/CU, 8-wave tiles 32; profiles/NOTES_gemm.md (r02_gemm_experiments) section 12), without shrinking the
// tile or the per-wave fragment block.  Each group double-buffers its own k-tiles (4 LDS stages in all); one workgroup
// barrier per pair of k-tiles; at the end the two partial sums (even k-tiles, odd k-tiles) meet through LDS, every 32x32
// fragment being finished and stored by one of the two groups.
template <int EPI, int WGM, int WGN, int WM, int WN, bool STAG>
__global__ __launch_bounds__(2 * WGM * WGN * 64, 1) void gemm_ksplit_kernel(GemmParams p, int ntiles) {
    constexpr int NWG = WGM * WGN;                         // waves per group
    constexpr int BM = WGM * WM * 32, BN = WGN * WN * 32, BK = 64;
    constexpr int ROWB = 128, CH = 8, RPP = 8;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA = BM / RPP / NWG, PB = BN / RPP / NWG, KS = BK / 16;
    static_assert(PA * RPP * NWG == BM && PB * RPP * NWG == BN, "tile rows must split evenly over a group's loader waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wid / NWG, gw = wid - grp * NWG;
    const int wm = gw / WGN, wn = gw % WGN;
    const int frow = lane & 31, fswz = (frow >> 1) & (CH - 1), fhalf = lane >> 5;
    int a_off[WM], b_off[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a_off[i] = (wm * WM * 32 + i * 32 + frow) * ROWB;
#pragma unroll
    for (int j = 0; j < WN; ++j) b_off[j] = A_BYTES + (wn * WN * 32 + j * 32 + frow) * ROWB;
    char* gbase = smem + grp * 2 * STAGE_BYTES;            // this group's two stages
    const int nk = p.K / BK;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {      // one tile per workgroup in the measured cases; more when tiles > CUs
    int m0, n0;
    tile_origin<BM, BN>(p, tile, m0, n0);
    const char* a_src[PA];
    const char* b_src[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (gw * PA + i) * RPP + lane / CH;
        const int chunk = (lane % CH) ^ ((row >> 1) & (CH - 1));
        int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
        a_src[i] = reinterpret_cast<const char*>(p.A + (size_t)ar * p.lda) + chunk * 16;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = (gw * PB + i) * RPP + lane / CH;
        const int chunk = (lane % CH) ^ ((row >> 1) & (CH - 1));
        int br = n0 + row; br = br < p.Nb ? br : p.Nb - 1;
        b_src[i] = reinterpret_cast<const char*>(p.B) + (size_t)br * p.ldb * 2 + chunk * 16;
    }
    auto issue_tile = [&](int kt, int st) {
#pragma unroll
        for (int q = 0; q < PA; ++q) glds16(a_src[q] + kt * 128, gbase + st * STAGE_BYTES + (gw * PA + q) * 1024);
#pragma unroll
        for (int q = 0; q < PB; ++q) glds16(b_src[q] + kt * 128, gbase + st * STAGE_BYTES + A_BYTES + (gw * PB + q) * 1024);
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    if (grp < nk) issue_tile(grp, 0);
    // residual epilogues: this wave's residual values (the fragments its group will finish) are requested now, so that the
    // epilogue -- which every CU reaches at the same moment with one tile per CU -- only has its stores left
    constexpr bool PRE = (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_BIAS_RESID_KEEP);
    constexpr int NOWN = PRE ? (WM * WN + 1) / 2 : 1;
    float4 rpre[NOWN][2][2];
    // ... and the bias of this lane's 8 columns per fragment column j: a load between the stores of the epilogue makes
    // hipcc wait with vmcnt(0), which on gfx950 waits for the STORES issued so far as well (one store latency per pass)
    float bpre[WN][8];
    if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int col = n0 + wn * WN * 32 + j * 32 + (lane & 3) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) bpre[j][e] = 0.f;
            if (col < p.N) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
                bpre[j][0] = b0.x; bpre[j][1] = b0.y; bpre[j][2] = b0.z; bpre[j][3] = b0.w;
                bpre[j][4] = b1.x; bpre[j][5] = b1.y; bpre[j][6] = b1.z; bpre[j][7] = b1.w;
            }
        }
    }
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (((i * WN + j) & 1) != grp) continue;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int row = m0 + wm * WM * 32 + i * 32 + pass * 16 + (lane >> 2);
                    const int col = n0 + wn * WN * 32 + j * 32 + (lane & 3) * 8;
                    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
                    if (row < p.M && col < p.N) {
                        const float* src = p.resid + (size_t)row * p.ldr + col;
                        r0 = *reinterpret_cast<const float4*>(src); r1 = *reinterpret_cast<const float4*>(src + 4);
                    }
                    rpre[(i * WN + j) >> 1][pass][0] = r0; rpre[(i * WN + j) >> 1][pass][1] = r1;
                }
            }
    }
    const int iters = (nk + 1) >> 1;
    // STAG: group 1 runs ONE barrier behind group 0 (two barriers per pair of k-tiles).  A group requests its next k-tile
    // right after its own barrier and needs it two barriers later, so the requests of the two groups reach the L2 half an
    // iteration apart and one group's are in flight while the other's are issued: the operand stream -- what bounds these
    // products, 1.77 MB per CU at <= 32 B/clk -- no longer drains to empty at every iteration boundary (unstaggered, every
    // wave waits for its k-tile, crosses the barrier, and only then are the next requests issued: one L2 latency per
    // iteration with nothing in flight).  Also the two waves of a SIMD no longer want the matrix pipe at the same time.
    if (STAG && grp) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        const int kt = 2 * it + grp;
        wait_vmcnt<0>();                                   // this wave's pieces of k-tile kt have landed
        if constexpr (STAG) __builtin_amdgcn_s_barrier();  // ... and those of the other waves of the group; the group's other stage is free
        else __syncthreads();                              // both groups' k-tiles have landed; the other stages are free
        if (kt + 2 < nk) issue_tile(kt + 2, (it + 1) & 1);
        if (kt < nk) {
            const char* sa = gbase + (it & 1) * STAGE_BYTES;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int coff = (((ks * 2 + fhalf) ^ fswz) << 4);
                bf16x8 af[WM], bfr[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + a_off[i] + coff);
#pragma unroll
                for (int j = 0; j < WN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sa + b_off[j] + coff);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr (STAG) {
            __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): the stage this group just read may be requested again
            __builtin_amdgcn_s_barrier();                  // (the other group's top-of-iteration barrier)
        }
    }
    if (STAG && !grp) __builtin_amdgcn_s_barrier();
    __syncthreads();
    ksplit_finish<EPI, WM, WN, NWG, NOWN, 3 * STAGE_BYTES>(p, smem, acc, rpre, bpre, m0, n0, grp, gw, wm, wn, wid, lane);
    __syncthreads();                                       // the LDS scratch is free before the next tile's first k-tile lands in it
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same tile (32*WM x 128, two groups of 1 x 4 waves, each wave a column of WM fragments) with the PHASED schedule of
// gemm8_kernel (round 3).  gemm_ksplit_kernel gives each group whole k-tiles and lets hipcc schedule the k-tile: a wave requests
// its 9 pieces (~100 cycles of issue each), then reads and multiplies fragment by fragment behind lgkmcnt(0) waits, then waits
// for its requests with vmcnt(0): an iteration is ~3,900 cycles for 1,280 cycles of MFMAs per SIMD (matrix pipe 19-21 % busy by
// the counters, profiles/archive/r03_mfma_utilisation.md), and at most one k-tile per group is in flight.  Here
//   * the groups split every k-tile: group g multiplies its k-steps 2g, 2g+1 (both groups read the same LDS stage);
//   * a wave's k-tile is ONE phase: LOAD (all 2*(WM+1) fragments, its share of the k-tile S-1 ahead by LDS-DMA, lgkmcnt(0))
//     -> barrier -> MFMA (2*WM back to back at raised priority) -> barrier, group 1 one barrier behind group 0, so that on
//     every SIMD one wave multiplies while its partner reads and requests;
//   * S = 4 stages of one k-tile: requests are issued three k-tiles (six barrier intervals) before their data is read, and the
//     wait for them is an exact vmcnt(pieces of the two younger k-tiles) -- never a drain.
// Hazards, by barrier interval (group 0: LOAD(kt) in interval 2kt, MFMA(kt) in 2kt+1; group 1: one later):
//   * WAR: stage (kt-1) % S is requested again from interval 2kt on; its last readers (group 1, LOAD(kt-1), interval 2kt-1)
//     waited for their ds_reads before the barrier that ends that interval;
//   * RAW: k-tile kt+1 is first read in interval 2kt+2; every wave waits for ITS pieces of k-tile kt+1 before the barrier that
//     ends interval 2kt+1 (group 0: after MFMA(kt); group 1: at the end of LOAD(kt)), having issued up to k-tile kt+S-1.
// Same MFMA order per accumulator and the same final sum (even-steps partial + odd-steps partial differs from the alternate-
// k-tile split of gemm_ksplit_kernel: results agree to f32 rounding, not bit for bit; tests/test_gpu_ops.py).
// KZ (round 5): two workgroups per tile, each half of the k-tiles (work items [0, ntiles): slice 0, [ntiles, 2 ntiles): slice 1);
// they meet in ksplit_finish<..., KZ>.  For the long-K N = E products at M = 3200 (batch 64), where 160x128 tiles fill 120 CUs and
// 96x128 tiles (204) stream 224 rows per k-tile for 96 rows of output: two slices of the larger tile put 240 workgroups on the
// chip with half the k-walk each.  MEASURED (batch 64, same box, scripts/experiments/gpu_r5_kz.sh): 31.8 / 35.7 us against 29.0 /
// 31.5 us for the 96x128 tiles (step 3.34 against 3.03 ms): the k-walk does halve, but the meeting of the two slices -- 96 KB
// written through, a ticket, 96 KB read back by another CU -- costs 9-11 us, more than the walk saves; as everywhere on this part,
// a dependency through memory inside a kernel costs what a kernel boundary costs.  Off by default (gemm_kz2 = 1 enables it).
template <int EPI, int WM, int NLREQ, bool KZ = false>
__global__ __launch_bounds__(512, 2) void gemm_kphase_kernel(GemmParams p, int ntiles) {
    constexpr int NW = 8, NWG = 4, WN = 1, S = 4;
    constexpr int BM = WM * 32, BN = 128, BK = 64;
    constexpr int ROWB = 128, CH = 8, RPP = 8;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PA_T = BM / RPP, NPT = (BM + BN) / RPP, NP = (NPT + NW - 1) / NW, NFULL = NPT % NW ? NPT % NW : NW;
    constexpr int NL = NLREQ < NP ? NLREQ : NP;            // pieces requested in the LOAD section; the others between the MFMAs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wid >> 2, gw = wid & 3, wm = 0, wn = gw;
    const bool big = wid < NFULL;                           // this wave requests NP pieces per k-tile (the others NP - 1)
    const int a_bytes = ((p.M - 1) * p.lda + p.K) * 2, b_bytes = ((p.Nb - 1) * p.ldb + p.K) * 2;
    const int frow = lane & 31, fswz = (frow >> 1) & (CH - 1), fhalf = lane >> 5;
    const int a_base = frow * ROWB, b_base = A_BYTES + (wn * 32 + frow) * ROWB;
    int coff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) coff[s] = ((((grp * 2 + s) * 2 + fhalf) ^ fswz) << 4);
    const int nk_all = p.K / BK;
    // outstanding requests this wave may leave when it needs k-tile t: its pieces of the k-tiles issued after t (at most S - 2)
    // tail / first k-tile: everything this wave has requested so far (drains; at most S - 1 times per tile)
  for (int work = blockIdx.x; work < (KZ ? 2 * ntiles : ntiles); work += gridDim.x) {
    const int z = (KZ && work >= ntiles) ? 1 : 0, tile = work - z * ntiles;
    const int k0 = KZ ? z * (nk_all / 2) : 0;                 // this slice: k-tiles [k0, k0 + nk)
    const int nk = KZ ? (z ? nk_all - nk_all / 2 : nk_all / 2) : nk_all;
    int m0, n0;
    tile_origin<BM, BN>(p, tile, m0, n0);
    int voff[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int q = i * NW + wid;                           // wave-uniform
        const bool is_a = q < PA_T;
        const int rb = is_a ? q : q - PA_T;
        const int row = rb * RPP + lane / CH;
        const int chunk = (lane % CH) ^ ((row >> 1) & (CH - 1));
        int r = (is_a ? m0 : n0) + row;
        const int lim = is_a ? p.M : p.Nb;
        r = r < lim ? r : lim - 1;
        voff[i] = r * (is_a ? p.lda : p.ldb) * 2 + chunk * 16 + k0 * 128;
    }
    auto issue_piece = [&](int kt, int i) {
        char* st = smem + (kt & (S - 1)) * STAGE_BYTES;
        const int q = i * NW + wid;
        const bool is_a = q < PA_T;                         // wave-uniform: a scalar select of the descriptor, no branch
        if (i < NP - 1 || NFULL == NW || big)               // only the last round can be short
            buffer_lds16(is_a ? (const void*)p.A : (const void*)p.B, is_a ? a_bytes : b_bytes, st + q * 1024, voff[i], kt * 128);
    };
    auto issue_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NP; ++i) issue_piece(kt, i);
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.0f;
    // residual epilogues: the bias of this lane's 8 columns and the residual values of the fragments this wave's group will
    // finish are requested FIRST (older than every operand request: the exact vmcnt waits below then cover them), so that the
    // epilogue -- which every CU reaches at the same moment with one tile per CU -- only has its stores left
    constexpr bool PRE = (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_BIAS_RESID_KEEP);
    constexpr int NOWN = PRE ? (WM * WN + 1) / 2 : 1;
    float4 rpre[NOWN][2][2];
    float bpre[WN][8];
    if constexpr (PRE) {
        // clamped addresses instead of bounds branches (a value loaded inside a branch makes hipcc drain with vmcnt(0) at the
        // join); rows / columns outside the matrix are never stored
        const int col = min(n0 + wn * 32 + (lane & 3) * 8, p.N - 8);
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        bpre[0][0] = b0.x; bpre[0][1] = b0.y; bpre[0][2] = b0.z; bpre[0][3] = b0.w;
        bpre[0][4] = b1.x; bpre[0][5] = b1.y; bpre[0][6] = b1.z; bpre[0][7] = b1.w;
#pragma unroll
        for (int k = 0; k < NOWN; ++k) {
            const int i = min(2 * k + grp, WM - 1);           // the k-th fragment this group finishes (a harmless repeat past the end)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int row = min(m0 + i * 32 + pass * 16 + (lane >> 2), p.M - 1);
                const float* src = p.resid + (size_t)row * p.ldr + col;
                rpre[k][pass][0] = *reinterpret_cast<const float4*>(src);
                rpre[k][pass][1] = *reinterpret_cast<const float4*>(src + 4);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
        if (t < nk) issue_tile(t);
    if (nk >= S - 1) { if (big) wait_vmcnt<2 * NP>(); else wait_vmcnt<2 * NP - 2>(); }     // k-tile 0 has landed; 1 and 2 may fly
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();                      // group 1 runs one barrier behind
    // one k-tile of this wave; STEADY: a k-tile is requested and two younger ones stay in flight (kt + S - 1 < nk).
    // Request order of a wave: L(0) M(0) L(1) M(1) ... (L(k): NL pieces of k-tile k + S - 1 in LOAD(k), M(k): the rest between
    // the MFMAs of k).  Group 0 waits for k-tile kt + 1 (= L, M of kt - 2) after MFMA(kt): 2 k-tiles' pieces are younger;
    // group 1 at the end of LOAD(kt): one k-tile's pieces + NL are younger.  Outside the steady state the waits drain.
    // past the last request (kt + S - 1 >= nk): k-tile kt + 1 with only k-tile kt + 2 (all of it requested long ago) younger
    auto tail_wait = [&](int kt) {
        if (kt + 2 < nk) { if (big) wait_vmcnt<NP>(); else wait_vmcnt<NP - 1>(); }
        else if (kt + 1 < nk) wait_vmcnt<0>();
    };
    auto ktile = [&](int kt, auto steady) {
        constexpr bool STEADY = decltype(steady)::value;
        const char* st = smem + (kt & (S - 1)) * STAGE_BYTES;
        const bool req = STEADY || kt + S - 1 < nk;
        // ---- LOAD
        bf16x8 af[2][WM], bfr[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bfr[s] = *reinterpret_cast<const bf16x8*>(st + b_base + coff[s]);
#pragma unroll
            for (int i = 0; i < WM; ++i) af[s][i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 32 * ROWB + coff[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (req) {
#pragma unroll
            for (int i = 0; i < NL; ++i) issue_piece(kt + S - 1, i);
        }
        if (grp) {
            if constexpr (STEADY) { if (big) wait_vmcnt<NP + NL>(); else wait_vmcnt<NP - 1 + (NL < NP ? NL : NP - 1)>(); }
            else tail_wait(kt);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): this wave's reads are in registers
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i], bfr[s], acc[i][0], 0, 0, 0);
                // one request behind every second MFMA
                const int m = s * WM + i;
                if ((m & 1) && NL + (m >> 1) < NP && req) issue_piece(kt + S - 1, NL + (m >> 1));
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!grp) {
            if constexpr (STEADY) { if (big) wait_vmcnt<2 * NP>(); else wait_vmcnt<2 * NP - 2>(); }
            else tail_wait(kt);
        }
        __builtin_amdgcn_s_barrier();
    };
    int kt = 0;
    for (; kt + S - 1 < nk; ++kt) ktile(kt, std::true_type{});
    for (; kt < nk; ++kt) ktile(kt, std::false_type{});
    if (!grp) __builtin_amdgcn_s_barrier();
    wait_vmcnt<0>();
    __syncthreads();
    constexpr int SCRATCH = 3 * STAGE_BYTES > NWG * WM * 4096 ? 3 * STAGE_BYTES : NWG * WM * 4096;
    ksplit_finish<EPI, WM, WN, NWG, NOWN, SCRATCH, KZ>(p, smem, acc, rpre, bpre, m0, n0, grp, gw, wm, wn, wid, lane, tile, z);
    __syncthreads();                                       // the LDS scratch is free before the next tile's first k-tile lands in it
  }
}

// band height (m-tiles) that hands every XCD whole bands: tiles / 8 consecutive tile indices per XCD (xcd_remap) = k bands of
// tiles_n * band tiles.  0 = keep TILE_BAND (several rounds, or no even split).
int xcd_band(int tiles, int tiles_n, int grid, const GemmTune& t) {
    if (t.band >= 0) return t.band;
    if (tiles > grid || tiles % 8 || (tiles / 8) % tiles_n) return 0;
    const int per_xcd_m = tiles / 8 / tiles_n;          // m-tiles per XCD
    for (int b = min(per_xcd_m, 8); b >= 1; --b)
        if (per_xcd_m % b == 0) return b;
    return 0;
}

template <int EPI, int WGM, int WGN, int WM, int WN, bool STAG>
int launch_ksplit(const GemmParams& p, const GemmTune& t, hipStream_t stream) {
    constexpr int bm = WGM * WM * 32, bn = WGN * WN * 32, stage = (bm + bn) * 128;
    constexpr int scratch_end = 3 * stage + 2 * WGM * WGN * 4096;
    constexpr int lds = 4 * stage > scratch_end ? 4 * stage : scratch_end;
    auto kern = gemm_ksplit_kernel<EPI, WGM, WGN, WM, WN, STAG>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            pevit_set_error("hipFuncSetAttribute(k-split gemm epi %d) failed", EPI);
            return -1;
        }
        attr_set = true;
    }
    const int tiles = ceil_div(p.M, bm) * ceil_div(p.N, bn);
    const int grid = min(tiles, num_cus() & ~7);
    GemmParams pb = p;
    pb.band = xcd_band(tiles, ceil_div(p.N, bn), grid, t);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(2 * WGM * WGN * 64), lds, stream, pb, tiles);
    LAUNCH_OK("gemm (k-split)");
    g_last_path = 3;
    return 0;
}

// floats of stream-K workspace a two-slice launch needs: two slabs per tile (ksplit_finish: 8 waves x ceil(WM / 2) fragments x 1024)
constexpr long kz_slab_floats(int wm) { return 2L * 8 * ((wm + 1) / 2) * 1024; }

template <int EPI, int WM, int NL, bool KZ = false>
int launch_kphase(const GemmParams& p, const GemmTune& t, hipStream_t stream) {
    constexpr int bm = WM * 32, bn = 128, stage = (bm + bn) * 128;
    constexpr int scratch = 3 * stage > 4 * WM * 4096 ? 3 * stage : 4 * WM * 4096;
    constexpr int lds = 4 * stage > scratch + 8 * 4096 ? 4 * stage : scratch + 8 * 4096;
    auto kern = gemm_kphase_kernel<EPI, WM, NL, KZ>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            pevit_set_error("hipFuncSetAttribute(phased k-split gemm epi %d) failed", EPI);
            return -1;
        }
        attr_set = true;
    }
    const long long a = ((long long)p.M * p.lda + p.K) * 2, b = ((long long)p.Nb * p.ldb + p.K) * 2;
    if (a >= (1LL << 31) || b >= (1LL << 31)) return launch_ksplit<EPI, 1, 4, WM, 1, true>(p, t, stream);   // 31-bit buffer offsets
    const int tiles = ceil_div(p.M, bm) * ceil_div(p.N, bn);
    const int grid = min(KZ ? 2 * tiles : tiles, num_cus() & ~7);
    GemmParams pb = p;
    pb.band = xcd_band(tiles, ceil_div(p.N, bn), min(tiles, grid), t);
    pb.kz = KZ ? 2 : 0;
    if (KZ && (!p.sk_slab || !p.sk_flag || tiles > p.sk_slots || tiles * kz_slab_floats(WM) > (long)p.sk_slots * PEVIT_SK_SLAB_FLOATS)) {
        pevit_set_error("gemm (two K slices per tile): the stream-K workspace is missing or too small for %d tiles", tiles); return -1;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, pb, tiles);
    LAUNCH_OK("gemm (phased k-split)");
    g_last_path = KZ ? 7 : 4;
    return 0;
}

// wgm x wgn waves of wm x wn fragments; wgs = workgroups per CU the LDS and registers are sized for
struct TileConfig { int wgm, wgn, wm, wn, wgs; bool hoist, spread, pipe; };
constexpr TileConfig kConfigs[] = {
    // 4-wave tiles request the next k-tile in one burst: with 2-4 workgroups per CU another workgroup computes while
    // this one sits in its issue slots (in the step: burst 5.59 ms, spread 5.75 ms; profiles/NOTES_gemm.md (r02_gemm_experiments) section 2).  The 8-wave tiles (one workgroup per
    // CU) spread the requests between their MFMAs.
    {2, 2, 2, 2, 2, true, false, false},   // 0: 128x128, 4 waves, 64 KiB, 2 workgroups / CU
    {2, 2, 1, 2, 3, true, false, false},   // 1:  64x128, 4 waves, 48 KiB, 3 workgroups / CU
    {2, 2, 1, 1, 4, false, false, false},  // 2:  64x64,  4 waves, 32 KiB, 4 workgroups / CU (bottleneck products, N <= 64)
    {4, 2, 2, 2, 1, true, true, false},    // 3: 256x128, 8 waves, 96 KiB, 1 workgroup / CU
    {2, 4, 4, 2, 1, false, true, false},   // 4: 256x256, 8 waves, 128 KiB
    {2, 4, 5, 2, 1, false, true, false},   // 5: 320x256, 8 waves, 144 KiB
    {2, 2, 2, 1, 3, true, false, false},   // 6: 128x64,  4 waves, 48 KiB, 3 workgroups / CU
    // software-pipelined twins of 0 and 1 (register double buffer, two k-tiles of LDS-DMA in flight): bit-identical
    // results, measured SLOWER on every shape of the step (c_proj 62.8 vs 54.9 us, 5.48 vs 5.45 ms per step) -- the
    // k-loop is bound by the operand stream, not by the ds_read -> MFMA latency the pipelining removes
    // (profiles/NOTES_gemm.md (r02_gemm_experiments) section 3).  Opt-in through gemm_config / gemm_cfg_longk for measurements only.
    {2, 2, 2, 2, 2, true, false, true},    // 7
    {2, 2, 1, 2, 2, true, false, true},    // 8
};
constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);
constexpr int cfg_bm(int c) { return kConfigs[c].wgm * kConfigs[c].wm * 32; }
constexpr int cfg_bn(int c) { return kConfigs[c].wgn * kConfigs[c].wn * 32; }

template <int EPI, int CFG, bool BF8>
int launch_cfg(const GemmParams& p, const GemmTune& t, hipStream_t stream) {
    constexpr TileConfig c = kConfigs[CFG];
    constexpr int nw = c.wgm * c.wgn, bm = cfg_bm(CFG), bn = cfg_bn(CFG);
    constexpr int lds = 2 * (bm + bn) * 128;
    constexpr int minw = c.wgs * nw / 4;
    auto kern = gemm_kernel<EPI, c.wgm, c.wgn, c.wm, c.wn, minw, c.hoist, BF8, c.spread, c.pipe && !BF8>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            pevit_set_error("hipFuncSetAttribute(gemm epi %d cfg %d) failed", EPI, CFG);
            return -1;
        }
        attr_set = true;
    }
    const int tiles = ceil_div(p.M, bm) * ceil_div(p.N, bn);
    // persistent grid: one workgroup per residency slot (a multiple of 8 keeps XCD affinity), or one
    // per tile when the tiles do not even fill the slots
    int grid = tiles;
    if (t.persistent) {
        const int slots = num_cus() * c.wgs;
        if (tiles > slots) grid = slots;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), lds, stream, p, tiles);
    LAUNCH_OK("gemm");
    g_last_path = 1;
    return 0;
}

// Tile shape per problem.  Measured on MI355X (scripts/bench_gemm.py, profiles/NOTES_gemm.md (r02_gemm_experiments)).
// The 8-wave tiles win when their tiling still gives (almost) every CU one tile per round;
// the N = 768 products of the ViT-B step (75 tiles of 256x256) stay on the 4-wave tiles.
// configuration 9 = 160x256 on 1 x 8 waves: only the staggered kernel has it (gemm8_kernel; `allow9`).  It is what fills the
// chip at the reference's own batch of 64 (M = 3200: 240 tiles for c_fc / dGELU where 320x256 has 120).
constexpr int CFG_160x256 = 9;
int pick_config(const GemmParams& p, const GemmTune& t, bool allow9) {
    if (t.config >= 0 && t.config < kNumConfigs) return t.config;
    if (t.config == CFG_160x256 && allow9) return CFG_160x256;
    if (p.N <= 64) return 2;
    const int cus = num_cus();
    if (t.big) {
        // rounds of the persistent loop and the fraction of the last round that is filled
        const int cand[4] = {5, 4, 3, CFG_160x256};
        int best = -1; double best_cost = 1e30;
        for (int i = 0; i < (allow9 ? 4 : 3); ++i) {
            const int c = cand[i];
            const int bm = c == 5 ? 320 : c == CFG_160x256 ? 160 : 256, bn = c == 3 ? 128 : 256;
            const long tiles = (long)ceil_div(p.M, bm) * ceil_div(p.N, bn);
            if (tiles * 10 < (long)cus * 7) continue;                 // fewer than 0.7 tiles per CU: leave to the small tiles
            const long rounds = (tiles + cus - 1) / cus;
            // per-CU operand bytes per k-tile (the L1-fill stream) x rounds ~ time of the k-loop
            const double cost = (double)rounds * (bm + bn);
            if (cost < best_cost) { best_cost = cost; best = c; }
        }
        if (best >= 0) {
            // the 128x128 tiling for comparison: 2 workgroups per CU, each streaming 256 rows per k-tile
            const long t128 = (long)ceil_div(p.M, 128) * ceil_div(p.N, 128);
            const double cost128 = (double)((t128 + cus - 1) / cus) * 256.0;
            if (best_cost < cost128 * t.big_bias / 100.0) return best;
        }
    }
    const long t128 = (long)ceil_div(p.M, 128) * ceil_div(p.N, 128);
    if (t128 >= 700) return 0;
    // few-tile problems (the N = E products): 128x128 only when K is long AND its tiles fit the 2 x CUs residency slots in
    // one round -- 300 tiles at ViT-B/32 B=128; at 520 (ViT-L/14, B=32) or 594 (ViT-B/16, B=64) tiles the handful of
    // second-round tiles doubles the kernel time and 64x128 is 1.3-2.7 % faster per step (profiles/NOTES_gemm.md (r02_gemm_experiments) 9)
    return (p.K >= t.kswitch && t128 <= 2L * cus) ? t.cfg_longk : t.cfg_shortk;
}

// Stream-K share (k-iterations per workgroup), 0 = keep the plain tiling.  Measured on the N = 768 products of ViT-B/32
// (scripts/gpu_streamk_sweep.py, profiles/NOTES_gemm.md (r02_gemm_experiments) section 13): the k-loops only get faster when workgroups
// that share an A panel -- TILE_BAND tiles = TILE_BAND * nk iterations apart in the walk -- stay in PHASE (same k at the
// same time), i.e. when the share divides TILE_BAND * nk: otherwise every workgroup streams its own k-slices through the
// XCD's 4 MiB L2 and the kernel becomes fabric-bound (63-75 us against 54 plain).  A share of at least nk / 2 keeps the
// hand-offs at one partial per tile.  K = 3072, 300 tiles: share 32 (450 workgroups), 54.8 -> 43.0 us.  Problems with MORE
// tiles than slots (ViT-B/16 at B = 64: 594, ViT-L/14 at B = 32: 520) were tried with the band chosen along with the share
// (band 7 / share 56, band 9 / share 72): +0.6 % and -3.2 % per step, so they keep the 64x128 tiling.
struct SkPlan { int share, band; };
SkPlan sk_plan(long tiles, int nk, int slots) {
    const long total = tiles * nk;
    const int lo = (int)max((total + slots - 1) / slots, (long)(nk + 1) / 2);
    for (int s = lo; 4 * s <= 3 * nk; ++s)
        if ((TILE_BAND * nk) % s == 0) return SkPlan{s, TILE_BAND};
    return SkPlan{0, 0};
}

// stream-K applies where the heuristic takes the 128x128 tile for a long-K problem with fewer tiles than residency slots
// (the N = E products of ViT-B/32), bf16 B, the five epilogues those products use (c_proj forward with and without the kept MLP output, the
// dX products in f32 / bf16, the patch embedding)
SkPlan streamk_plan(const GemmParams& p, const GemmTune& t, int cfg) {
    const SkPlan none{0, 0};
    if (!t.streamk || !p.sk_slab || !p.sk_flag || !t.persistent) return none;
    const int tiles_m = ceil_div(p.M, 128);
    const long tiles = (long)tiles_m * ceil_div(p.N, 128);
    const int slots = min(pevit_gemm_sk_slots(), p.sk_slots);
    const int nk = p.K / 64;
    if (slots < 8) return none;
    if (t.streamk == 2)        // measurement only
        return (t.sk_share && (tiles * nk + t.sk_share - 1) / t.sk_share <= slots) ? SkPlan{t.sk_share, t.sk_band ? t.sk_band : TILE_BAND} : none;
    if (t.config >= 0 || cfg != 0 || t.ablate || nk < 16 || tiles >= slots) return none;
    const SkPlan pl = sk_plan(tiles, nk, slots);
    if (pl.share) return pl;
    return t.streamk == 3 ? SkPlan{nk, TILE_BAND} : none;     // 3: the whole-tile walk through this kernel (measurement)
}

template <int EPI>
int launch_streamk(const GemmParams& p_in, SkPlan plan, hipStream_t stream) {
    const int share = plan.share;
    constexpr int lds = 2 * (128 + 128) * 128;
    auto kern = gemm_streamk_kernel<EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            pevit_set_error("hipFuncSetAttribute(stream-K gemm epi %d) failed", EPI);
            return -1;
        }
        attr_set = true;
    }
    GemmParams p = p_in;
    p.sk_share = share; p.sk_band = plan.band;
    const int tiles = ceil_div(p.M, 128) * ceil_div(p.N, 128);
    // every workgroup must be resident: the grid never exceeds the 2-per-CU slots (streamk_share checked that)
    const int grid = (int)(((long)tiles * (p.K / 64) + share - 1) / share + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p, tiles);
    LAUNCH_OK("gemm (stream-K)");
    g_last_path = 5;
    return 0;
}

// few-row long-K products on gemm_skinny_kernel: 2-6 slices of ~12 k-tiles, tiles x slices within the slab slots
struct SkinnyPlan { int nslices, ksl; };
SkinnyPlan skinny_plan(const GemmParams& p, const GemmTune& t) {
    const SkinnyPlan none{0, 0};
    if (!t.skinny || t.config >= 0 || t.ablate || p.b_fp8 || p.a_fp8 || !p.sk_slab || !p.sk_flag) return none;
    const int nk = p.K / 64;
    if (p.M > t.skinny_maxm || p.N < 64 || nk < t.skinny_mink) return none;
    const long tiles = (long)ceil_div(p.M, 128) * ceil_div(p.N, 64);
    const int slots = min(pevit_gemm_sk_slots(), p.sk_slots);       // a slot holds 128x128 floats: two of these slabs
    const int nslices = t.skinny_slices > 0 ? min(6, max(1, t.skinny_slices)) : nk < 24 ? 1 : min(6, max(2, (nk + 6) / 12));
    if (tiles * nslices > 2L * slots || tiles > slots) return none;  // tickets: one word per tile
    return SkinnyPlan{nslices, ceil_div(nk, nslices)};
}

template <int EPI>
int launch_skinny(const GemmParams& p, SkinnyPlan plan, hipStream_t stream) {
    constexpr int lds = 4 * (128 + 64) * 128;
    auto kern = gemm_skinny_kernel<EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            pevit_set_error("hipFuncSetAttribute(few-row gemm epi %d) failed", EPI);
            return -1;
        }
        attr_set = true;
    }
    const int tiles_n = ceil_div(p.N, 64), tiles = ceil_div(p.M, 128) * tiles_n;
    hipLaunchKernelGGL(kern, dim3(tiles * plan.nslices), dim3(256), lds, stream, p, tiles_n, plan.nslices, plan.ksl);
    LAUNCH_OK("gemm (few rows)");
    g_last_path = 6;
    return 0;
}

// the 160x128 two-group tile: where the heuristic takes a 4-wave tile for a long-K problem whose 160x128 tiling gives (almost)
// every CU exactly one tile -- the N = E products of ViT-B/32 at B = 128 (240 tiles)
// returns the tile height in fragments: 5 (160x128), 3 (96x128: M = 3200, the reference's own batch of 64, gives 204 tiles where
// 160x128 gives 120), or 0 = not this kernel
int use_ksplit(const GemmParams& p, const GemmTune& t, int cfg) {
    if (!t.ksplit || t.config >= 0 || (cfg != 0 && cfg != 1) || t.ablate || p.K < t.ksplit_mink) return 0;
    const int cus = num_cus() & ~7;
    const int cand[2] = {5, 3};
    for (int i = 0; i < (t.ksplit_small ? 2 : 1); ++i) {
        const long tiles = (long)ceil_div(p.M, cand[i] * 32) * ceil_div(p.N, 128);
        const long rounds = (tiles + cus - 1) / cus;
        if (t.ksplit == 1 && rounds > 1) continue;          // 2: also problems of several rounds (measurement)
        if (4 * tiles >= 3 * rounds * cus) return cand[i];  // the last round at least 3/4 full on average
    }
    return 0;
}

// two workgroups per 160x128 tile, half of K each (gemm_kphase_kernel<..., KZ>): where those tiles fill at most half the chip and
// each slice still walks >= 16 k-tiles (M = 3200, N = 768, K >= 2048: three products per layer at batch 64)
bool use_kz2(const GemmParams& p, const GemmTune& t) {
    if (!t.kz2 || !p.sk_slab || !p.sk_flag) return false;
    const long tiles = (long)ceil_div(p.M, 160) * ceil_div(p.N, 128);
    const long long a = ((long long)p.M * p.lda + p.K) * 2, b = ((long long)p.Nb * p.ldb + p.K) * 2;
    return 2 * tiles <= (num_cus() & ~7) && 4 * tiles >= (num_cus() & ~7) && p.K / 64 >= 32 && tiles <= p.sk_slots &&
           tiles * kz_slab_floats(5) <= (long)p.sk_slots * PEVIT_SK_SLAB_FLOATS && a < (1LL << 31) && b < (1LL << 31);
}

// the staggered 8-wave kernel: 256x256 (configuration 4) and 320x256 (5), bf16 B, operands addressable with 31-bit byte offsets
template <int EPI, int WGM, int WGN, int WM, int WN, int KSP, int OPS>
int launch_big8(const GemmParams& p, const GemmTune& t, hipStream_t stream) {
    constexpr int bm = WGM * WM * 32, bn = WGN * WN * 32, stage = (bm + bn) * 128, scratch = 8 * 32 * WN * 32 * 4;
    constexpr int lds = stage + (stage > scratch ? stage : scratch);
    auto kern = gemm8_kernel<EPI, WGM, WGN, WM, WN, KSP, OPS>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            pevit_set_error("hipFuncSetAttribute(staggered gemm epi %d) failed", EPI);
            return -1;
        }
        attr_set = true;
    }
    const int tiles = ceil_div(p.M, bm) * ceil_div(p.N, bn);
    int grid = tiles;
    if (t.persistent && tiles > num_cus()) grid = num_cus();
    GemmParams pb = p;
    pb.band = xcd_band(tiles, ceil_div(p.N, bn), grid, t);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, pb, tiles);
    LAUNCH_OK("gemm (staggered 8-wave)");
    g_last_path = 2;
    return 0;
}
bool big8_ok(const GemmParams& p, const GemmTune& t) {
    if (!t.stagger || (t.ablate & 12)) return false;
    const long long a = ((long long)p.M * p.lda + p.K) * 2, b = ((long long)p.Nb * p.ldb + p.K) * (p.b_fp8 ? 1 : 2);
    return a < (1LL << 31) && b < (1LL << 31);
}

template <int EPI, bool BF8>
int launch_epi(const GemmParams& p, const GemmTune& t, hipStream_t stream) {
    if constexpr (!BF8 && (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_F32 || EPI == EPI_BF16 || EPI == EPI_BIAS_RESID_KEEP ||
                           EPI == EPI_BIAS_GELU || EPI == EPI_DGELU_BF16)) {
        const SkinnyPlan sp = skinny_plan(p, t);
        if (sp.nslices) return launch_skinny<EPI>(p, sp, stream);
    }
    const bool stag = big8_ok(p, t);
    const int cfg = pick_config(p, t, stag);
    if (stag) {
        if (cfg == 5) return launch_big8<EPI, 2, 4, 5, 2, 1, BF8 ? 1 : 0>(p, t, stream);
        if (cfg == 4) return launch_big8<EPI, 2, 4, 4, 2, 1, BF8 ? 1 : 0>(p, t, stream);
        if (cfg == CFG_160x256) return launch_big8<EPI, 1, 8, 5, 1, 1, BF8 ? 1 : 0>(p, t, stream);
        if (cfg == 3 && t.stagger >= 2) return launch_big8<EPI, 4, 2, 2, 2, 2, BF8 ? 1 : 0>(p, t, stream);   // 256x128, two k-steps per phase (measurement)
    }
    if constexpr (!BF8 && (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_F32 || EPI == EPI_BF16 || EPI == EPI_BIAS_RESID_KEEP || EPI == EPI_PATCH_EMBED)) {
        const int kwm = use_ksplit(p, t, cfg);
        if (kwm && t.ksplit_stagger == 2 && use_kz2(p, t)) return launch_kphase<EPI, 5, 8, true>(p, t, stream);
        if (kwm && t.ksplit_stagger == 2) {
            // requests between the MFMAs (kphase_nl = 2: two in LOAD, the rest behind every second MFMA) win 3 % back to back and
            // lose 0.9 % in the step (28.14 k vs 27.89 k images/s, two pairs; profiles/NOTES_gemm.md (r03_gemm_experiments) section 8)
            if (kwm == 5) return t.kphase_nl <= 2 ? launch_kphase<EPI, 5, 2>(p, t, stream) : launch_kphase<EPI, 5, 8>(p, t, stream);
            return t.kphase_nl <= 2 ? launch_kphase<EPI, 3, 2>(p, t, stream) : launch_kphase<EPI, 3, 8>(p, t, stream);
        }
        if (kwm == 5) return t.ksplit_stagger ? launch_ksplit<EPI, 1, 4, 5, 1, true>(p, t, stream) : launch_ksplit<EPI, 1, 4, 5, 1, false>(p, t, stream);
        if (kwm == 3) return launch_ksplit<EPI, 1, 4, 3, 1, true>(p, t, stream);
        const SkPlan plan = streamk_plan(p, t, cfg);
        if (plan.share) return launch_streamk<EPI>(p, plan, stream);
    }
    switch (cfg) {
        case 0: return launch_cfg<EPI, 0, BF8>(p, t, stream);
        case 1: return launch_cfg<EPI, 1, BF8>(p, t, stream);
        case 2: return launch_cfg<EPI, 2, BF8>(p, t, stream);
        case 3: return launch_cfg<EPI, 3, BF8>(p, t, stream);
        case 4: return launch_cfg<EPI, 4, BF8>(p, t, stream);
        case 6: return launch_cfg<EPI, 6, BF8>(p, t, stream);
        case 7: return launch_cfg<EPI, 7, BF8>(p, t, stream);
        case 8: return launch_cfg<EPI, 8, BF8>(p, t, stream);
        default: return launch_cfg<EPI, 5, BF8>(p, t, stream);
    }
}

// fp8 x fp8 (GemmParams::a_fp8): always on the staggered kernel; the 8-wave tile whose rounds x bytes-per-k-tile is smallest
template <int EPI>
int launch_f8a(const GemmParams& p, const GemmTune& t, hipStream_t stream) {
    const long long a = (long long)p.M * p.lda + p.K, b = (long long)p.Nb * p.ldb + p.K;
    if (a >= (1LL << 31) || b >= (1LL << 31)) { pevit_set_error("gemm (fp8 x fp8): operands beyond 2 GiB"); return -1; }
    // 160x256 on 1 x 8 waves only: a fragment of the 64-deep instruction is 8 registers per lane, and with two fragment columns
    // per wave (320x256 / 256x256: 160 / 128 accumulator registers + 56 / 48 of fragments) hipcc spills 166 / 26 registers
    return launch_big8<EPI, 1, 8, 5, 1, 1, 2>(p, t, stream);
}

}  // namespace

// the bf16 tail of an fp8-B problem (GemmParams::B2) is implemented by gemm8_kernel<.., OPS = 1> only: true when launch_epi would
// take one of its 256-column tiles for this problem and the tail starts on a tile boundary
bool pevit_gemm_mixed_ok(const GemmParams& p, const GemmTune& t) {
    if (!p.b_fp8 || p.a_fp8 || !p.B2 || p.n_fp8 % 256 || p.K % 128) return false;       // (EPI_QKV_HEADS only: pevit_launch_gemm checks)
    if (!big8_ok(p, t)) return false;
    // the tail is addressed through B's buffer descriptor: it must lie behind B, within 31 bits of it
    const long long off = reinterpret_cast<const char*>(p.B2) - reinterpret_cast<const char*>(p.B);
    if (off <= 0 || off + ((long long)p.Nb2 * p.ldb2 + p.K) * 2 >= (1LL << 31)) return false;
    const int cfg = pick_config(p, t, true);
    return cfg == 5 || cfg == 4 || cfg == CFG_160x256;
}

int pevit_num_cus() { return num_cus(); }
int pevit_gemm_sk_slots() { return min(2 * num_cus(), PEVIT_SK_MAX_SLOTS) & ~7; }
int pevit_gemm_last_path() { return g_last_path; }

int pevit_launch_gemm(int epi, const GemmParams& p_in, const GemmTune& t, hipStream_t stream) {
    GemmParams p = p_in;
    p.dbg = t.ablate;
    if (p.K % 64 != 0 || p.K <= 0) { pevit_set_error("gemm: K=%d must be a positive multiple of 64", p.K); return -1; }
    if (p.N % 8 != 0) { pevit_set_error("gemm: N=%d must be a multiple of 8", p.N); return -1; }
    if (p.M <= 0 || p.N <= 0) { pevit_set_error("gemm: empty problem M=%d N=%d", p.M, p.N); return -1; }
    if ((!p.a_fp8 && (p.lda % 8)) || (p.ldb % (p.b_fp8 ? 16 : 8))) { pevit_set_error("gemm: lda/ldb must be multiples of 8 (16 for fp8 B)"); return -1; }
    if (p.a_fp8) {
        if (!p.b_fp8 || p.K % 128 != 0 || (p.lda % 16)) { pevit_set_error("gemm: fp8 A needs fp8 B, K %% 128 == 0 and lda %% 16 == 0"); return -1; }
        // the forward frozen products of the block (opt-in weight format "fp8-act")
        switch (epi) {
            case EPI_QKV_HEADS: return launch_f8a<EPI_QKV_HEADS>(p, t, stream);
            case EPI_BIAS_RESID_F32: return launch_f8a<EPI_BIAS_RESID_F32>(p, t, stream);
            case EPI_BIAS_GELU: return launch_f8a<EPI_BIAS_GELU>(p, t, stream);
            case EPI_F32: return launch_f8a<EPI_F32>(p, t, stream);
        }
        pevit_set_error("gemm: epilogue %d has no fp8 x fp8 form", epi);
        return -1;
    }
    if (p.B2 && (epi != EPI_QKV_HEADS || !pevit_gemm_mixed_ok(p, t))) { pevit_set_error("gemm: this problem has no kernel with a bf16 tail (ask pevit_gemm_mixed_ok first)"); return -1; }
    if (p.b_fp8) {
        if (p.K % 128 != 0) { pevit_set_error("gemm: fp8 B needs K=%d to be a multiple of 128", p.K); return -1; }
        // the frozen-weight products of the block and their dX forms
        switch (epi) {
            case EPI_QKV_HEADS: return launch_epi<EPI_QKV_HEADS, true>(p, t, stream);
            case EPI_BIAS_RESID_F32: return launch_epi<EPI_BIAS_RESID_F32, true>(p, t, stream);
            case EPI_BIAS_GELU: return launch_epi<EPI_BIAS_GELU, true>(p, t, stream);
            case EPI_DGELU_BF16: return launch_epi<EPI_DGELU_BF16, true>(p, t, stream);
            case EPI_F32: return launch_epi<EPI_F32, true>(p, t, stream);
            case EPI_BF16: return launch_epi<EPI_BF16, true>(p, t, stream);
        }
        pevit_set_error("gemm: epilogue %d has no fp8-weight form", epi);
        return -1;
    }
    switch (epi) {
        case EPI_QKV_HEADS: return launch_epi<EPI_QKV_HEADS, false>(p, t, stream);
        case EPI_BIAS_RESID_F32: return launch_epi<EPI_BIAS_RESID_F32, false>(p, t, stream);
        case EPI_BIAS_GELU: return launch_epi<EPI_BIAS_GELU, false>(p, t, stream);
        case EPI_DGELU_BF16: return launch_epi<EPI_DGELU_BF16, false>(p, t, stream);
        case EPI_F32: return launch_epi<EPI_F32, false>(p, t, stream);
        case EPI_BF16: return launch_epi<EPI_BF16, false>(p, t, stream);
        case EPI_BIAS_BF16: return launch_epi<EPI_BIAS_BF16, false>(p, t, stream);
        case EPI_PATCH_EMBED: return launch_epi<EPI_PATCH_EMBED, false>(p, t, stream);
        case EPI_BIAS_RELU_BF16: return launch_epi<EPI_BIAS_RELU_BF16, false>(p, t, stream);
        case EPI_BIAS_RESID_KEEP: return launch_epi<EPI_BIAS_RESID_KEEP, false>(p, t, stream);
        case EPI_BIAS_GELUNEW: return launch_epi<EPI_BIAS_GELUNEW, false>(p, t, stream);
        case EPI_DRELU_BF16: return launch_epi<EPI_DRELU_BF16, false>(p, t, stream);
        case EPI_DGELUNEW_BF16: return launch_epi<EPI_DGELUNEW_BF16, false>(p, t, stream);
    }
    pevit_set_error("gemm: unknown epilogue %d", epi);
    return -1;
}
