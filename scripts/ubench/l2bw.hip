// L2 -> CU load throughput on gfx950 as a function of access shape, staging path and depth.
// Mimics the operand stream of a tiled GEMM: a workgroup (4 waves) walks the K extent of a 128-row panel
// (row stride ld bytes) in slabs of SLAB bytes per row; many workgroups share a panel (L2 hits).
//   hipcc --offload-arch=gfx950 -O3 -o l2bw l2bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LPR lanes per row (16 B each) -> SLAB = LPR*16 bytes per row per step; a wave covers 32 rows per step.
template <int LPR, bool GLDS, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ base, size_t ld, int panels, int kbytes, int steps,
                                                      unsigned* sink, int lds_pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RPI = 64 / LPR;            // rows per instruction
    constexpr int NI = 32 / RPI;             // instructions per step per wave
    constexpr int SLAB = LPR * 16;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int panel = blockIdx.x % panels;
    const char* src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = panel * 128 + wid * 32 + i * RPI + lane / LPR;
        src[i] = base + (size_t)row * ld + (lane % LPR) * 16;
    }
    u32x4 acc = {0, 0, 0, 0};
    int koff = (blockIdx.x / panels) * SLAB % kbytes;      // workgroups of one panel start at different k
    // ring of DEPTH steps; each step lands in its own LDS slot (GLDS) or register set (plain)
    u32x4 r[DEPTH][NI];
    auto issue = [&](int slot) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if constexpr (GLDS) {
                char* dst = smem + ((slot * 4 + wid) * NI + i) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + koff),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else {
                r[slot][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src[i] + koff));
            }
        }
        koff += SLAB; if (koff >= kbytes) koff -= kbytes;
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d);
    for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            issue((d + DEPTH - 1) % DEPTH);
            wait_vm<NI * (DEPTH - 1)>();
            if constexpr (GLDS) {
                // touch the landed slot so that the data path is complete (one ds_read per lane)
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + ((d * 4 + wid) * NI) * 1024 + lane * 16);
                acc ^= v;
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i) acc ^= r[d][i];
            }
        }
    }
    wait_vm<0>();
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int LPR, bool GLDS, int DEPTH>
double run(const char* buf, size_t ld, int panels, int kbytes, int wgs, int steps, unsigned* sink, int lds_bytes) {
    auto kern = stream_kernel<LPR, GLDS, DEPTH>;
    if (GLDS && DEPTH * 4 * (LPR / 2) * 1024 > lds_bytes) return 0.0;     // does not fit this occupancy
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds_bytes, 0, buf, ld, panels, kbytes, steps, sink, 0);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds_bytes, 0, buf, ld, panels, kbytes, steps, sink, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * steps * 128.0 * (LPR * 16) * reps;
    return bytes / (ms * 1e-3) / 1e12;       // TB/s
}


// ---- GEMM-like operand stream: workgroup (tm, tn) streams A panel tm and B panel tn along K ------------
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
// LPR: lanes per row (4 -> BK 32, 8 -> BK 64).  MODE bit0: barrier per step; bit1: skip A; bit2: skip B; bit3: no xcd remap
template <int LPR, int DEPTH, int MODE>
__global__ __launch_bounds__((MODE & 16) ? 512 : 256) void gemm_stream_kernel(const char* __restrict__ A, const char* __restrict__ B, size_t ld, int tiles_m,
                                                           int tiles_n, int kbytes, unsigned* sink, int band) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RPI = 64 / LPR, NI = 32 / RPI, SLAB = LPR * 16;
    const int lane = threadIdx.x & 63, wid8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = wid8 & 3, grp = wid8 >> 2;
    const int ntiles = tiles_m * tiles_n;
    const int bid = (MODE & 32) ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;     // NB: halves of a tile land on different XCDs
    const int bid2 = (MODE & 64) ? (int)((blockIdx.x & 7) + ((blockIdx.x >> 4) << 3)) : bid;   // 64: halves on the same XCD
    const int t = (MODE & 8) ? bid2 : xcd_remap(bid2, ntiles);
    const int bnd = t / (band * tiles_n), within = t - bnd * (band * tiles_n);
    const int mb = min(band, tiles_m - bnd * band);
    const int tn = within / mb, tm = bnd * band + (within - tn * mb);
    const char* a_src[NI];
    const char* b_src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = wid * 32 + i * RPI + lane / LPR;
        a_src[i] = A + (size_t)(tm * 128 + r) * ld + (lane % LPR) * 16;
        b_src[i] = B + (size_t)(tn * 128 + r) * ld + (lane % LPR) * 16;
    }
    constexpr int PER = ((MODE & 2) ? 0 : NI) + ((MODE & 4) ? 0 : NI);
    if (MODE & (16 | 32 | 64)) kbytes >>= 1;
    int koff = grp * kbytes;
    if (MODE & 32) koff = (blockIdx.x & 1) * kbytes;
    if (MODE & 64) koff = ((blockIdx.x >> 3) & 1) * kbytes;
    char* const smem_g = smem + grp * (DEPTH * 4 * 2 * NI) * 1024;
    u32x4 acc = {0, 0, 0, 0};
    auto issue = [&](int slot) {
        char* dst = smem_g + ((slot * 4 + wid) * 2 * NI) * 1024;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (!(MODE & 2))
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            if (!(MODE & 4))
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + koff),
                                                 (__attribute__((address_space(3))) void*)(dst + (NI + i) * 1024), 16, 0, 0);
        }
        koff += SLAB;
    };
    const int steps = kbytes / SLAB;
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d);
    for (int s = 0; s + DEPTH <= steps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (s + d + DEPTH - 1 < steps) { issue((d + DEPTH - 1) % DEPTH); wait_vm<PER * (DEPTH - 1)>(); } else wait_vm<0>();
            if (MODE & 1) __builtin_amdgcn_s_barrier();
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem_g + ((d * 4 + wid) * 2 * NI) * 1024 + lane * 16);
            acc ^= v;
        }
    }
    wait_vm<0>();
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int LPR, int DEPTH, int MODE>
void run_gemm_like(const char* A, const char* B, size_t ld, int tiles_m, int tiles_n, int kbytes, unsigned* sink, int band, const char* what) {
    auto kern = gemm_stream_kernel<LPR, DEPTH, MODE>;
    const int lds = DEPTH * 4 * 2 * (LPR / 2) * 1024 * ((MODE & 16) ? 2 : 1);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wgs = tiles_m * tiles_n * ((MODE & (32 | 64)) ? 2 : 1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3((MODE & 16) ? 512 : 256), lds, 0, A, B, ld, tiles_m, tiles_n, kbytes, sink, band);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3((MODE & 16) ? 512 : 256), lds, 0, A, B, ld, tiles_m, tiles_n, kbytes, sink, band);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const int panels = ((MODE & 2) ? 0 : 1) + ((MODE & 4) ? 0 : 1);
    const double bytes = (double)tiles_m * tiles_n * panels * 128.0 * kbytes * reps;
    printf("  %-58s %7.1f us/launch %6.2f TB/s\n", what, ms * 1e3 / reps, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

// ---- co-resident pairing experiment: 512 workgroups (2 per CU); do two workgroups on one CU that stream the SAME A
// panel (and different B panels) get the second copy from L1?  MODE 0: pairs by blockIdx (b, b+256 share A);
// MODE 1: pairs by hardware CU id (s_getreg HW_ID / XCC_ID + atomics); MODE 2: no sharing (all panels distinct).
template <int PMODE>
__global__ __launch_bounds__(256) void pair_stream_kernel(const char* __restrict__ A, const char* __restrict__ B, size_t ld, int kbytes,
                                                          unsigned* sink, int* cnt, int* dense, int* ncu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LPR = 8, DEPTH = 2, RPI = 64 / LPR, NI = 32 / RPI, SLAB = LPR * 16;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ int s_pair, s_slot;
    if (threadIdx.x == 0) {
        if (PMODE == 1) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // XCC_ID
            const int key = (int)(((xcc & 15) << 8) | ((hw >> 8) & 0xff));                 // xcc | se,sh,cu
            const int slot = atomicAdd(&cnt[key], 1);
            if (slot == 0) { const int d = atomicAdd(ncu, 1); atomicExch(&dense[key], d); }
            int d;
            while ((d = atomicAdd(&dense[key], 0)) < 0) __builtin_amdgcn_s_sleep(1);
            s_pair = d; s_slot = slot;
        } else if (PMODE == 0) {
            s_pair = blockIdx.x & 255; s_slot = blockIdx.x >> 8;
        } else {
            s_pair = blockIdx.x; s_slot = 0;
        }
    }
    __syncthreads();
    const int pair = s_pair, slot = s_slot;
    // pair p -> A panel p % 50 ; B panel: (2 * (p / 50) + slot) ... distinct B panels per slot
    const int tm = (PMODE == 2) ? (blockIdx.x % 50) : (pair % 50);
    const int tn = (PMODE == 2) ? (blockIdx.x / 50) : (2 * (pair / 50) + (slot & 1) + 12 * (slot >> 1));
    const char* a_src[NI];
    const char* b_src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = wid * 32 + i * RPI + lane / LPR;
        a_src[i] = A + (size_t)(tm * 128 + r) * ld + (lane % LPR) * 16;
        b_src[i] = B + (size_t)((tn % 24) * 128 + r) * ld + (lane % LPR) * 16;
    }
    int koff = 0;
    u32x4 acc = {0, 0, 0, 0};
    auto issue = [&](int s) {
        char* dst = smem + ((s * 4 + wid) * 2 * NI) * 1024;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + koff),
                                             (__attribute__((address_space(3))) void*)(dst + (NI + i) * 1024), 16, 0, 0);
        }
        koff += SLAB;
    };
    const int steps = kbytes / SLAB;
    issue(0);
    for (int s = 0; s + DEPTH <= steps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (s + d + 1 < steps) { issue((d + 1) % DEPTH); wait_vm<2 * NI>(); } else wait_vm<0>();
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + ((d * 4 + wid) * 2 * NI) * 1024 + lane * 16);
            acc ^= v;
        }
    }
    wait_vm<0>();
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    if (PMODE == 1 && threadIdx.x == 0 && slot > 1) atomicAdd(&sink[1], 1u);      // CUs that got more than 2 workgroups
}

template <int PMODE>
void run_pair(const char* A, const char* B, size_t ld, int kbytes, unsigned* sink, const char* what) {
    int *cnt, *dense, *ncu;
    CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMalloc(&dense, 4096 * 4)); CK(hipMalloc(&ncu, 4));
    auto kern = pair_stream_kernel<PMODE>;
    const int lds = 64 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float total = 0;
    const int reps = 10;
    for (int w = 0; w < reps + 2; ++w) {
        CK(hipMemsetAsync(cnt, 0, 4096 * 4)); CK(hipMemsetAsync(dense, 0xff, 4096 * 4)); CK(hipMemsetAsync(ncu, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(512), dim3(256), lds, 0, A, B, ld, kbytes, sink, cnt, dense, ncu);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (w >= 2) total += ms;
    }
    int h_ncu = 0; unsigned h_sink[2] = {0, 0};
    CK(hipMemcpy(&h_ncu, ncu, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_sink, sink, 8, hipMemcpyDeviceToHost));
    const double bytes = 512.0 * 2 * 128.0 * kbytes * reps;
    printf("  %-52s %7.1f us/launch %6.2f TB/s   (distinct CUs seen %d, extra slots %u)\n", what, total * 1e3 / reps,
           bytes / (total * 1e-3) / 1e12, h_ncu, h_sink[1]);
    fflush(stdout);
}

// ---- 256x256 tiles: one 8-wave workgroup per CU streams a 256-row A panel and a 256-row B panel over a K range
__global__ __launch_bounds__(512) void big_tile_stream_kernel(const char* __restrict__ A, const char* __restrict__ B, size_t ld,
                                                              int tiles_m, int tiles_n, int ksplit, int kbytes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LPR = 8, RPI = 8, NI = 4, SLAB = 128;           // per wave and step: 32 rows of A and 32 rows of B
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x, tile = item / ksplit, kpart = item - tile * ksplit;
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    const char* a_src[NI];
    const char* b_src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = wid * 32 + i * RPI + lane / LPR;
        a_src[i] = A + (size_t)(tm * 256 + r) * ld + (lane % LPR) * 16;
        b_src[i] = B + (size_t)(tn * 256 + r) * ld + (lane % LPR) * 16;
    }
    const int kspan = kbytes / ksplit;
    int koff = kpart * kspan;
    u32x4 acc = {0, 0, 0, 0};
    auto issue = [&](int s) {
        char* dst = smem + ((s * 8 + wid) * 2 * NI) * 1024;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + koff),
                                             (__attribute__((address_space(3))) void*)(dst + (NI + i) * 1024), 16, 0, 0);
        }
        koff += SLAB;
    };
    const int steps = kspan / SLAB;
    issue(0);
    for (int s = 0; s + 2 <= steps; s += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            if (s + d + 1 < steps) { issue((d + 1) % 2); wait_vm<2 * NI>(); } else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + ((d * 8 + wid) * 2 * NI) * 1024 + lane * 16);
            acc ^= v;
        }
    }
    wait_vm<0>();
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

void run_big(const char* A, const char* B, size_t ld, int tiles_m, int tiles_n, int ksplit, int kbytes, unsigned* sink, const char* what) {
    const int lds = 2 * 8 * 8 * 1024;     // 128 KiB: one workgroup per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(big_tile_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wgs = tiles_m * tiles_n * ksplit;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(big_tile_stream_kernel, dim3(wgs), dim3(512), lds, 0, A, B, ld, tiles_m, tiles_n, ksplit, kbytes, sink);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(big_tile_stream_kernel, dim3(wgs), dim3(512), lds, 0, A, B, ld, tiles_m, tiles_n, ksplit, kbytes, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)tiles_m * tiles_n * 512.0 * kbytes * reps;
    printf("  %-64s %4d WGs %7.1f us/launch %6.2f TB/s\n", what, wgs, ms * 1e3 / reps, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const bool pmc = argc > 1;
    const int K = 3072;                       // bf16 elements per row
    const size_t ld = (size_t)K * 2;
    const int rows = 6400;
    char* buf; unsigned* sink;
    CK(hipMalloc(&buf, (size_t)rows * ld)); CK(hipMemset(buf, 1, (size_t)rows * ld)); CK(hipMalloc(&sink, 4));
    const int kbytes = K * 2;
    {
        char* Bm; CK(hipMalloc(&Bm, (size_t)3072 * ld)); CK(hipMemset(Bm, 2, (size_t)3072 * ld));
        printf("GEMM-like stream, M=6400 (50 row panels), K=%d, 128x128 tiles\n", K);
        if (argc > 1 && argv[1][0] == 'b') {     // 256x256 tiles
            run_big(buf, Bm, ld, 25, 9, 1, 1536, sink, "qkv   N=2304 K=768 : 256x256 tiles");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, 18, 1536, sink, 6, "qkv   N=2304 K=768 : 128x128 tiles");
            run_big(buf, Bm, ld, 25, 12, 1, 1536, sink, "c_fc  N=3072 K=768 : 256x256 tiles");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, 24, 1536, sink, 6, "c_fc  N=3072 K=768 : 128x128 tiles");
            run_big(buf, Bm, ld, 25, 3, 3, 6144, sink, "c_proj N=768 K=3072: 256x256 tiles, split-K 3");
            run_big(buf, Bm, ld, 25, 3, 1, 6144, sink, "c_proj N=768 K=3072: 256x256 tiles, no split");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, 6, 6144, sink, 6, "c_proj N=768 K=3072: 128x128 tiles");
            run_big(buf, Bm, ld, 25, 3, 3, 4608, sink, "dqkv  N=768 K=2304: 256x256 tiles, split-K 3");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, 6, 4608, sink, 6, "dqkv  N=768 K=2304: 128x128 tiles");
            run_big(buf, Bm, ld, 25, 3, 3, 1536, sink, "out   N=768 K=768 : 256x256 tiles, split-K 3");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, 6, 1536, sink, 6, "out   N=768 K=768 : 128x128 tiles");
            return 0;
        }
        if (argc > 1 && argv[1][0] == 'p') {     // co-resident pairing
            CK(hipMemset(sink, 0, 4));
            unsigned* sink2; CK(hipMalloc(&sink2, 8)); CK(hipMemset(sink2, 0, 8));
            for (int kb : {1536, 6144}) {
                printf(" K extent %d B\n", kb);
                run_pair<2>(buf, Bm, ld, kb, sink2, "512 WGs, tiles in plain order (A shared by chance)");
                run_pair<0>(buf, Bm, ld, kb, sink2, "512 WGs, blocks b and b+256 share the A panel");
                run_pair<1>(buf, Bm, ld, kb, sink2, "512 WGs, the workgroups of one CU share the A panel");
            }
            return 0;
        }
        if (argc > 1 && argv[1][0] == 'k') {     // split-K across workgroups
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, 6, kbytes, sink, 6, "tn=6: one WG per tile");
            run_gemm_like<8, 2, 32>(buf, Bm, ld, 50, 6, kbytes, sink, 6, "tn=6: two WGs per tile (K halves), different XCDs");
            run_gemm_like<8, 2, 64>(buf, Bm, ld, 50, 6, kbytes, sink, 6, "tn=6: two WGs per tile (K halves), same XCD");
            run_gemm_like<8, 2, 16>(buf, Bm, ld, 50, 6, kbytes, sink, 6, "tn=6: one 8-wave WG per tile (K halves)");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, 12, kbytes, sink, 6, "tn=12: one WG per tile");
            run_gemm_like<8, 2, 64>(buf, Bm, ld, 50, 12, kbytes, sink, 6, "tn=12: two WGs per tile (K halves), same XCD");
            return 0;
        }
        if (argc > 1 && argv[1][0] == 's') {     // row-stride sweep
            for (int pad : {0, 64, 128, 256, 384, 1024, 2048 + 128}) {
                const size_t ldp = ld + pad;
                char* Ap; char* Bp;
                CK(hipMalloc(&Ap, (size_t)6400 * ldp)); CK(hipMemset(Ap, 1, (size_t)6400 * ldp));
                CK(hipMalloc(&Bp, (size_t)3072 * ldp)); CK(hipMemset(Bp, 2, (size_t)3072 * ldp));
                char name[128];
                snprintf(name, sizeof name, "row stride %zu B, tn=6", ldp);
                run_gemm_like<8, 2, 0>(Ap, Bp, ldp, 50, 6, kbytes, sink, 6, name);
                snprintf(name, sizeof name, "row stride %zu B, tn=24", ldp);
                run_gemm_like<8, 2, 0>(Ap, Bp, ldp, 50, 24, kbytes, sink, 6, name);
                CK(hipFree(Ap)); CK(hipFree(Bp));
            }
            return 0;
        }
        if (pmc) {     // two kernels only, for counter passes
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, 6, kbytes, sink, 6, "gemm-like tn=6 BK64 depth2");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, 24, kbytes, sink, 6, "gemm-like tn=24 BK64 depth2");
            double t = run<8, true, 4>(buf, ld, 50, kbytes, 256, 768, sink, 80 * 1024);
            printf("independent panels (50), glds 128B rows depth4: %.2f TB/s\n", t);
            t = run<8, true, 4>(buf, ld, 8, kbytes, 256, 768, sink, 80 * 1024);
            printf("shared panels (8), glds 128B rows depth4: %.2f TB/s\n", t);
            return 0;
        }
        for (int tn : {6, 24}) {
            printf(" tiles_n = %d (N = %d): %d workgroups\n", tn, tn * 128, 50 * tn);
            run_gemm_like<4, 4, 1>(buf, Bm, ld, 50, tn, kbytes, sink, 6, "BK32 depth4 barrier, A+B");
            run_gemm_like<4, 4, 0>(buf, Bm, ld, 50, tn, kbytes, sink, 6, "BK32 depth4 no barrier, A+B");
            run_gemm_like<8, 2, 1>(buf, Bm, ld, 50, tn, kbytes, sink, 6, "BK64 depth2 barrier, A+B");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, tn, kbytes, sink, 6, "BK64 depth2 no barrier, A+B");
            run_gemm_like<8, 4, 0>(buf, Bm, ld, 50, tn, kbytes, sink, 6, "BK64 depth4 no barrier, A+B");
            run_gemm_like<8, 2, 16>(buf, Bm, ld, 50, tn, kbytes, sink, 6, "BK64 depth2 no barrier, A+B, 8 waves (K split in WG)");
            run_gemm_like<8, 3, 16>(buf, Bm, ld, 50, tn, kbytes, sink, 6, "BK64 depth3 no barrier, A+B, 8 waves (K split in WG)");
            run_gemm_like<8, 2, 4>(buf, Bm, ld, 50, tn, kbytes, sink, 6, "BK64 depth2 no barrier, A only");
            run_gemm_like<8, 2, 2>(buf, Bm, ld, 50, tn, kbytes, sink, 6, "BK64 depth2 no barrier, B only");
            run_gemm_like<8, 2, 8>(buf, Bm, ld, 50, tn, kbytes, sink, 6, "BK64 depth2 no barrier, A+B, no XCD remap");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, tn, kbytes, sink, 2, "BK64 depth2 no barrier, A+B, band 2");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, tn, kbytes, sink, 50, "BK64 depth2 no barrier, A+B, band 50 (n-major)");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, tn, kbytes, sink, 1, "BK64 depth2 no barrier, A+B, band 1 (m-major)");
        }
        for (int kb : {1536}) {
            printf(" K extent %d B (K=768), tiles_n 24\n", kb);
            run_gemm_like<8, 2, 1>(buf, Bm, ld, 50, 24, kb, sink, 6, "BK64 depth2 barrier, A+B");
            run_gemm_like<8, 2, 0>(buf, Bm, ld, 50, 24, kb, sink, 6, "BK64 depth2 no barrier, A+B");
        }
        return 0;
    }
    printf("panel rows 128, row stride %zu B, K extent %d B; TB/s aggregate (and B/clk/CU at 2.1 GHz x 256 CU)\n", ld, kbytes);
    struct Cfg { const char* name; int wgs; int lds; } cfgs[] = {{"1 WG/CU", 256, 81920 * 2 - 1024}, {"2 WG/CU", 512, 80 * 1024}, {"4 WG/CU", 1024, 40 * 1024}};
    for (int panels : {1, 8, 50}) {
        for (auto& c : cfgs) {
            const int steps = 1536;
            double t;
            auto show = [&](const char* what, double tbs) { printf("  panels %2d %-8s %-26s %6.2f TB/s  %5.1f B/clk/CU\n", panels, c.name, what, tbs, tbs * 1e12 / 256 / 2.1e9); };
            t = run<4, true, 4>(buf, ld, panels, kbytes, c.wgs, steps, sink, c.lds);   show("glds  64B rows depth4", t);
            t = run<8, true, 4>(buf, ld, panels, kbytes, c.wgs, steps / 2, sink, c.lds); show("glds 128B rows depth4", t);
            t = run<16, true, 2>(buf, ld, panels, kbytes, c.wgs, steps / 4, sink, c.lds); show("glds 256B rows depth2", t);
            t = run<8, true, 2>(buf, ld, panels, kbytes, c.wgs, steps / 2, sink, c.lds); show("glds 128B rows depth2", t);
            t = run<8, true, 8>(buf, ld, panels, kbytes, c.wgs, steps / 2, sink, c.lds); show("glds 128B rows depth8", t);
            t = run<4, false, 4>(buf, ld, panels, kbytes, c.wgs, steps, sink, c.lds);  show("regs  64B rows depth4", t);
            t = run<8, false, 4>(buf, ld, panels, kbytes, c.wgs, steps / 2, sink, c.lds); show("regs 128B rows depth4", t);
            t = run<16, false, 2>(buf, ld, panels, kbytes, c.wgs, steps / 4, sink, c.lds); show("regs 256B rows depth2", t);
            fflush(stdout);
        }
    }
    return 0;
}
