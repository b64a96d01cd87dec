// What ds_read_b64_tr_b16 returns: LDS holds u16[i] = i, every lane passes its own element offset; prints the 4 values per lane.
// (on the GPU box)  hipcc --offload-arch=gfx950 scripts/probe_tr_b16.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(const int* addr_of_lane, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int a = addr_of_lane[threadIdx.x];
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)r[j];
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h_addr[l] = 4 * l;                       // lane l -> its own 4 consecutive elements
            if (pat == 1) h_addr[l] = (l & 15) * 64 + (l >> 4) * 4;   // row (l&15) of a [16][64] matrix, 4 cols per 16-lane group
            if (pat == 2) h_addr[l] = (l & 3) * 4 + ((l >> 2) & 3) * 64 + (l >> 4) * 256;
        }
        hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d_addr, d_out);
        hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d : %4d %4d %4d %4d\n", l, h_addr[l], h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
    }
    return 0;
}
