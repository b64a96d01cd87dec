// Probe (round 5, measurement only -- not part of libpevit_hip.so): a kernel that pulls a buffer into the L2 of EVERY XCD.
// Workgroup b runs on XCD b % 8 (the round-robin the GEMM tile maps already rely on); the per_xcd workgroups of an XCD share the
// buffer in 4 KB stripes, 16 bytes per lane.  Used by scripts/r5_l2_prefetch.py to see whether a side stream that touches the NEXT
// product's weight panel while the current product runs shortens the next product (its first touches per XCD become L2 hits).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/probes/l2_touch.hip -o scripts/probes/libl2touch.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void l2_touch_kernel(const uint4* __restrict__ p, size_t n16, int per_xcd, unsigned* sink) {
    const int xcd_slot = blockIdx.x >> 3;                 // which of the XCD's workgroups
    uint4 acc = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)xcd_slot * 256 + threadIdx.x; i < n16; i += (size_t)per_xcd * 256) {
        const uint4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && sink) *sink = 1u;      // keeps the loads alive
}

extern "C" int probe_l2_touch(void* stream, const void* ptr, size_t bytes, int per_xcd, void* sink) {
    if (per_xcd < 1) per_xcd = 1;
    hipLaunchKernelGGL(l2_touch_kernel, dim3(8 * per_xcd), dim3(256), 0, (hipStream_t)stream, (const uint4*)ptr, bytes / 16, per_xcd, (unsigned*)sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
