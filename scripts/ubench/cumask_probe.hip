// Where do the workgroups of a CU-masked stream land?  hipExtStreamCreateWithCUMask(mask) + a kernel that records XCC_ID / HW_ID.
// build: hipcc --offload-arch=gfx950 -O2 scripts/ubench/cumask_probe.hip -o gpurun_out/cumask_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // XCC_ID
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}
static void run(const char* name, const std::vector<uint32_t>& mask, int wgs) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    unsigned* d; hipMalloc(&d, wgs * 8); hipMemset(d, 0xff, wgs * 8);
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(64), 0, s, d, 2000);      // 20 us at 100 MHz: all resident at once
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * wgs); hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::map<unsigned, int>> per;      // xcc -> (se, sh, cu) -> count
    for (int i = 0; i < wgs; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per[xcc][(se << 8) | (sh << 4) | cu]++;
    }
    printf("%s: %d workgroups\n", name, wgs);
    for (auto& x : per) {
        printf("  xcc %u: %zu distinct CUs:", x.first, x.second.size());
        for (auto& c : x.second) printf(" se%u.sh%u.cu%u x%d", c.first >> 8, (c.first >> 4) & 1, c.first & 15, c.second);
        printf("\n");
    }
    hipFree(d); hipStreamDestroy(s);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
    std::vector<uint32_t> all(8, 0xffffffffu), top16(8, 0), lin(8, 0), low16(8, 0), one(8, 0);
    top16[7] = 0xffff0000u;                     // bits 240..255
    low16[0] = 0x0000ffffu;                     // bits 0..15
    for (int w = 0; w < 8; ++w) lin[w] = 0xc0000000u;       // bits 32w+30, 32w+31
    one[0] = 1u;
    run("all", all, 512);
    run("top16 (bits 240..255)", top16, 64);
    run("low16 (bits 0..15)", low16, 64);
    run("two per 32-bit word (bits 32w+30, 32w+31)", lin, 64);
    run("bit 0 only", one, 16);
    return 0;
}
