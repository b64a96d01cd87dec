// Small HBM-streaming helpers: weight conversion at load time, (N,B,E)<->(B,N,E) row
// permutes at the API seam, and the fused SGD(momentum) update over the flat parameter
// buffer (reference: optim/build.py:120-127 -> torch.optim.SGD, nesterov=False).
#include "common.h"
#include "kernels.h"

namespace {

template <typename ST>
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t n, float scale) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        st_store4<ST>(dst, i, v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t j = n & ~(size_t)3; j < n; ++j) st_store<ST>(dst, j, src[j] * scale);
}

// dst[c*ldd + r] = src[r*cols + c] * (r < scaled_rows ? scale : 1)
template <typename ST>
__global__ void transpose_bf16_kernel(const float* __restrict__ src, int rows, int cols, bf16* __restrict__ dst,
                                      int ldd, int scaled_rows, float scale) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] * (r < scaled_rows ? scale : 1.0f) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) st_store<ST>(dst, (size_t)c * ldd + r, tile[tx][i]);
    }
}

// to_internal: dst[(b*N+n)*E + e] = src[(n*B+b)*E + e] ; else the inverse
__global__ void permute_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int B, int E,
                                    int to_internal) {
    const int row = blockIdx.x;              // destination row
    int srow;
    if (to_internal) { const int b = row / N, n = row - b * N; srow = n * B + b; }
    else             { const int n = row / B, b = row - n * B; srow = b * N + n; }
    const float4* s = reinterpret_cast<const float4*>(src + (size_t)srow * E);
    float4* d = reinterpret_cast<float4*>(dst + (size_t)row * E);
    for (int i = threadIdx.x; i < E / 4; i += blockDim.x) d[i] = s[i];
}

__global__ void scale_f32_kernel(float* p, size_t n, float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] *= scale;
}

// zero-fill (16 bytes per lane, grid-stride): the step's own memsets.  hipMemsetAsync would do the same work, but as a memset NODE
// of a captured HIP graph it did not keep its place among the kernels of the step (engine.capture_train_step: the gradient
// buffer was not clear when the chain kernels accumulated into it from the second replay on); a kernel node does.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* __restrict__ p, size_t n16, unsigned char* __restrict__ tail, int ntail) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom,
                           const unsigned char* __restrict__ has_grad, size_t n, float lr, float momentum, float wd,
                           int flags, float grad_scale, const unsigned* __restrict__ poison, unsigned* skipped,
                           const unsigned* __restrict__ poison2, float* loss_slot) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // a stream-K hand-off that timed out in this step's GEMMs left a wrong tile behind and raised the context's error
    // word (gemm.hip): the update is skipped, parameters and momentum stay as they were, and the host finds the word at
    // its next check (engine.check_streamk) -- corrupted gradients never reach the parameters
    // poison2: an error word owned by somebody else (pevit_set_external_poison: the gradient exchange of data parallelism,
    // allreduce.hip) -- a bucket that was never reduced must not reach the parameters either
    if ((poison && *poison) || (poison2 && *poison2)) {
        if (i == 0 && skipped) atomicAdd(skipped, 1u);      // how many updates were withheld: reported with the error
        // ... and the loss of this step reads NaN wherever the caller had it written: logs that are read before the host's next
        // check of the error word (engine.STREAMK_CHECK_EVERY) show the failure instead of a number from a corrupted step
        if (i == 0 && loss_slot) *loss_slot = __builtin_nanf("");
        return;
    }
    if (has_grad && !has_grad[i]) return;       // torch skips parameters whose .grad is None
    const bool first_step = flags & 1, nesterov = flags & 2;
    float d = g[i] * grad_scale + wd * p[i];
    const float buf = first_step ? d : momentum * mom[i] + d;
    mom[i] = buf;
    p[i] -= lr * (nesterov ? d + momentum * buf : buf);      // torch.optim.SGD: grad.add(buf, alpha=momentum)
}

// measurement only: `blocks` workgroups that hold their CU slots for `micros` microseconds (wall_clock64: 100 MHz), optionally with
// `lds` bytes of LDS each -- a stand-in for the RCCL kernels of an overlapped all-reduce (scripts/r4_coresidency.py)
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ char occ_lds[];
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        acc += 1;
    }
    if (sink && acc == 0xffffffffu) { occ_lds[threadIdx.x] = 1; sink[0] = acc + occ_lds[0]; }
}

}  // namespace

int pevit_launch_occupy(int blocks, int lds_bytes, double micros, hipStream_t s) {
    if (blocks <= 0) return 0;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            pevit_set_error("occupy: cannot raise the LDS limit"); return -1;
        }
        attr = true;
    }
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, s, (unsigned long long)(micros * 100.0), (unsigned*)nullptr);
    LAUNCH_OK("occupy_kernel");
    return 0;
}

int pevit_launch_cast_bf16(const float* src, bf16* dst, size_t n, float scale, hipStream_t s, int f32) {
    if (n == 0) return 0;
    const int blocks = (int)((n / 4 + 255) / 256);
    const dim3 grid(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
    if (f32) hipLaunchKernelGGL(cast_bf16_kernel<float>, grid, dim3(256), 0, s, src, dst, n, scale);
    else hipLaunchKernelGGL(cast_bf16_kernel<bf16>, grid, dim3(256), 0, s, src, dst, n, scale);
    LAUNCH_OK("cast_bf16_kernel");
    return 0;
}

int pevit_launch_transpose_bf16(const float* src, int rows, int cols, bf16* dst, int ldd, int scaled_rows, float scale,
                                hipStream_t s, int f32) {
    const dim3 grid(ceil_div(cols, 32), ceil_div(rows, 32));
    if (f32) hipLaunchKernelGGL(transpose_bf16_kernel<float>, grid, dim3(256), 0, s, src, rows, cols, dst, ldd, scaled_rows, scale);
    else hipLaunchKernelGGL(transpose_bf16_kernel<bf16>, grid, dim3(256), 0, s, src, rows, cols, dst, ldd, scaled_rows, scale);
    LAUNCH_OK("transpose_bf16_kernel");
    return 0;
}

int pevit_launch_permute_rows(const float* src, float* dst, int N, int B, int E, int to_internal, hipStream_t s) {
    if (E % 4) { pevit_set_error("permute_rows: width %d must be a multiple of 4", E); return -1; }
    hipLaunchKernelGGL(permute_rows_kernel, dim3(N * B), dim3(192), 0, s, src, dst, N, B, E, to_internal);
    LAUNCH_OK("permute_rows_kernel");
    return 0;
}

int pevit_launch_zero(void* ptr, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    if (reinterpret_cast<size_t>(ptr) & 15) { pevit_set_error("zero: the buffer must be 16-byte aligned"); return -1; }
    const size_t n16 = bytes / 16;
    const int ntail = (int)(bytes - n16 * 16);
    size_t blocks = (n16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(ptr), n16,
                       reinterpret_cast<unsigned char*>(ptr) + n16 * 16, ntail);
    LAUNCH_OK("zero_fill_kernel");
    return 0;
}

int pevit_launch_scale_f32(float* p, size_t n, float scale, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(scale_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, scale);
    LAUNCH_OK("scale_f32_kernel");
    return 0;
}

int pevit_launch_sgd(float* p, const float* g, float* mom, const unsigned char* has_grad, size_t n, float lr,
                     float momentum, float wd, int first_step, float grad_scale, hipStream_t s, const unsigned* poison,
                     unsigned* skipped, const unsigned* poison2, float* loss_slot) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, mom, has_grad, n, lr,
                       momentum, wd, first_step, grad_scale, poison, skipped, poison2, loss_slot);
    LAUNCH_OK("sgd_kernel");
    return 0;
}
