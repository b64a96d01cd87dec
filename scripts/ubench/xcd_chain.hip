// A chain of small dependent stages (the class-token rows of the last block + projection + head: ~20 launches on 128 rows, ~200 us
// of a 4.4 ms step): 16 launches against ONE persistent kernel whose workgroups all sit on one XCD and meet at L2-atomic barriers.
//   stage s: panel_{s+1}[128][768] (bf16) = f(panel_s (all of it, written by the other workgroups), W_s (cold, 16 cols x 768 per unit))
//   48 units of 16 output columns per stage; per unit: 196 KB of panel (L2) + 24 KB of weights (HBM) in, 4 KB out.
// A: one launch per stage, 48 workgroups (anywhere on the chip).   B: 256 workgroups launched, those with XCC_ID == 0 stay (32 on this
// part), unit u of a stage goes to participant u % P; barrier = vmcnt(0) -> __syncthreads -> atomic add (agent scope, relaxed) -> spin;
// the panel is read with sc1 loads (another CU's write-through L1 is not a problem, a stale line in MY L1 would be).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/xcd_chain.hip -o /tmp/xcd_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int M = 128, E = 768, UNITS = E / 16, STAGES = 16;

__device__ __forceinline__ u32x4 load16_sc1(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float bits_sum(u32x4 v) { return (float)((v.x ^ v.y ^ v.z ^ v.w) & 0xff); }

// one unit: every thread reads its share of the panel (128 x 768 bf16 = 12288 16-byte pieces / 256 threads = 48 pieces) and of the
// unit's weights (16 x 768 bf16 = 1536 pieces / 256 = 6), reduces, writes 128 x 16 bf16 (256 pieces: one per thread)
template <bool SC1>
__device__ __forceinline__ void unit(const unsigned short* __restrict__ panel, const unsigned short* __restrict__ w, unsigned short* out, int u) {
    const int tid = threadIdx.x;
    float acc = 0.f;
    u32x4 wv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wv[i] = *reinterpret_cast<const u32x4*>(w + ((size_t)u * 1536 + tid + 256 * i) * 8);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        u32x4 pv[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const void* src = panel + ((size_t)tid + 256 * (half * 24 + i)) * 8;
            pv[i] = SC1 ? load16_sc1(src) : *reinterpret_cast<const u32x4*>(src);
        }
        if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 24; ++i) acc += bits_sum(pv[i]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc += bits_sum(wv[i]);
    const unsigned short o = (unsigned short)((unsigned)acc & 0x3f80);
    // row tid / 2, columns 16 u + 8 (tid & 1) .. + 7
    const unsigned oo = (unsigned)o | ((unsigned)o << 16);
    u32x4 ov = {oo, oo, oo, oo};
    unsigned short* dst = out + (size_t)(tid >> 1) * E + 16 * u + 8 * (tid & 1);
    if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(ov) : "memory");
    else *reinterpret_cast<u32x4*>(dst) = ov;
}

__global__ __launch_bounds__(256) void stage_kernel(const unsigned short* panel, const unsigned short* w, unsigned short* out) {
    unit<false>(panel, w, out, blockIdx.x);
}

__global__ __launch_bounds__(256) void chain_kernel(unsigned short* p0, unsigned short* p1, const unsigned short* w, unsigned* sync, int want_xcc) {
    __shared__ int me, P;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;
    // registration: everybody reports; the workgroups of XCD want_xcc draw an index and wait until all have reported
    if (threadIdx.x == 0) {
        int idx = -1;
        if ((int)xcc == want_xcc) idx = (int)__hip_atomic_fetch_add(sync + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        me = idx;
        if (idx >= 0) {
            while (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(2);
            P = (int)__hip_atomic_load(sync + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (me < 0) return;
    const int p = P;
    for (int s = 0; s < STAGES; ++s) {
        const unsigned short* src = (s & 1) ? p1 : p0;
        unsigned short* dst = (s & 1) ? p0 : p1;
        for (int u = me; u < UNITS; u += p) unit<true>(src, w + (size_t)s * UNITS * 1536 * 8, dst, u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(sync + 2 + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(sync + 2 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (me == 0 && threadIdx.x == 0) sync[40] = (unsigned)p;
}

int main() {
    unsigned short *p0, *p1, *w; unsigned* sync;
    const size_t wbytes = (size_t)STAGES * UNITS * 1536 * 16;          // 16 x 1.2 MB
    hipMalloc(&p0, M * E * 2); hipMalloc(&p1, M * E * 2); hipMalloc(&w, wbytes); hipMalloc(&sync, 64 * 4);
    hipMemset(p0, 1, M * E * 2); hipMemset(p1, 1, M * E * 2); hipMemset(w, 1, wbytes);
    // something to flush the caches between repetitions (the real step touches > 2 GB between two visits of these weights)
    char* junk; hipMalloc(&junk, 512 << 20);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f, sum = 0.f; const int reps = 20;
        for (int r = 0; r < reps; ++r) {
            hipMemsetAsync(junk, r, 512 << 20, s);
            hipMemsetAsync(sync, 0, 64 * 4, s);
            hipEventRecord(e0, s);
            if (mode == 0) {
                for (int st = 0; st < STAGES; ++st)
                    hipLaunchKernelGGL(stage_kernel, dim3(UNITS), dim3(256), 0, s, (st & 1) ? p1 : p0, w + (size_t)st * UNITS * 1536 * 8, (st & 1) ? p0 : p1);
            } else {
                hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(256), 0, s, p0, p1, w, sync, 0);
            }
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r >= 3) { sum += ms; if (ms < best) best = ms; }
        }
        unsigned h[64]; hipMemcpy(h, sync, 64 * 4, hipMemcpyDeviceToHost);
        printf("%s: %d stages  mean %.1f us  best %.1f us  (%.2f us per stage)%s\n", mode ? "one persistent kernel on XCD 0" : "one launch per stage (48 workgroups)",
               STAGES, sum / (reps - 3) * 1e3, best * 1e3, sum / (reps - 3) * 1e3 / STAGES, mode ? "" : "");
        if (mode) printf("  participants on XCD 0: %u\n", h[40]);
    }
    return 0;
}
