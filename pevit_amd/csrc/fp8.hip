// fp8 (OCP e4m3fn) packing of the frozen backbone weights -- BASELINE config 5 (ViT-L/14, "fp8 weights").
//
// Format: one POWER-OF-TWO scale per output channel (row of the [out][in] weight),
//     s[r] = 2^ceil(log2(amax_r / 448)),      code[r][c] = e4m3(W[r][c] / s[r])        (RNE, never saturates)
// so that de-quantisation s[r] * code is exact in bf16 (e4m3's 3 mantissa bits are a subset of bf16's 7 and a
// power of two only shifts the exponent), the scale commutes with every rounding in the GEMM, and the fp8 engine is
// bit-identical to the bf16 engine run on the de-quantised weights (tests/test_gpu_fp8.py).  A floating-point format
// loses nothing from a power-of-two scale: the relative step is 2^-3 .. 2^-4 at every magnitude down to the
// subnormal range (2^-9 * s).
//
// Storage: codes are k-permuted inside every group of 128 input channels (fp8_kperm) so that the 32 codes one MFMA lane
// needs of a 64-wide k-tile are contiguous (gemm.hip).  The backward GEMMs (dX = dY W) use the TRANSPOSED code matrix;
// there the contraction runs over the output channels, whose scales are folded into dY by its producer (exact again).
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float pow2_scale(float amax) {
    if (!(amax > 0.f)) return 1.0f;
    int e;
    const float m = frexpf(amax * (1.0f / 448.0f), &e);       // amax/448 = m * 2^e, m in [0.5, 1)
    return ldexpf(1.0f, m > 0.5f ? e : e - 1);                  // 2^ceil(log2(amax/448))
}

__device__ __forceinline__ unsigned char to_e4m3(float x) {
    return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(x, 0.0f, 0, false) & 0xff);
}

// one workgroup per row: scale[r] and the permuted codes of row r.  rows < scaled_rows carry `pre` (the 1/sqrt(64) folded
// into the q rows of in_proj, model.py:786-787)
__global__ __launch_bounds__(256) void quant_rows_kernel(const float* __restrict__ W, int cols, unsigned char* __restrict__ out,
                                                         int ldo, float* __restrict__ scale, int scaled_rows, float pre) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const float f = r < scaled_rows ? pre : 1.0f;
    const float* w = W + (size_t)r * cols;
    float amax = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) amax = fmaxf(amax, fabsf(w[c] * f));
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = pow2_scale(amax);
    if (threadIdx.x == 0) scale[r] = s;
    const float inv = 1.0f / s;                                 // exact: s is a power of two
    for (int c = threadIdx.x; c < cols; c += 256) out[(size_t)r * ldo + fp8_kperm(c)] = to_e4m3(w[c] * f * inv);
}

// outT[c][perm(r)] = e4m3(W[r][c] * pre(r) / scale[r]) : the same codes, transposed, permuted along r
__global__ void quant_transpose_kernel(const float* __restrict__ W, int rows, int cols, const float* __restrict__ scale,
                                       unsigned char* __restrict__ outT, int ldo, int scaled_rows, float pre) {
    __shared__ unsigned char tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        unsigned char v = 0;
        if (r < rows && c < cols) v = to_e4m3(W[(size_t)r * cols + c] * (r < scaled_rows ? pre : 1.0f) / scale[r]);
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) outT[(size_t)c * ldo + fp8_kperm(r)] = tile[tx][i];
    }
}

// dst[r][c] = bf16(src[r][c] * colscale[c])
__global__ void cast_bf16_cols_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t rows, int cols,
                                      const float* __restrict__ colscale) {
    const size_t n4 = rows * (size_t)cols / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % (size_t)cols);
        const float4 v = *reinterpret_cast<const float4*>(src + i * 4);
        const float4 s = *reinterpret_cast<const float4*>(colscale + c);
        bf16x4 o;
        o[0] = f2bf(v.x * s.x); o[1] = f2bf(v.y * s.y); o[2] = f2bf(v.z * s.z); o[3] = f2bf(v.w * s.w);
        *reinterpret_cast<bf16x4*>(dst + i * 4) = o;
    }
}

// out[r][c] = scale[r] * e4m3_decode(code[r][perm(c)])   (tests / checkpoint export: the weights the engine computes with)
__global__ void dequant_rows_kernel(const unsigned char* __restrict__ codes, int ldc, const float* __restrict__ scale, int cols,
                                    float* __restrict__ out) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const int code = codes[(size_t)r * ldc + fp8_kperm(c)];
        const f32x2 v = __builtin_amdgcn_cvt_pk_f32_fp8(code, false);
        out[(size_t)r * cols + c] = v[0] * scale[r];
    }
}

}  // namespace

int pevit_launch_quant_rows_fp8(const float* W, int rows, int cols, unsigned char* out, int ldo, float* scale, int scaled_rows,
                                float pre, hipStream_t s) {
    if (cols % 128 || ldo % 128) { pevit_set_error("quant_rows_fp8: %d input channels (ld %d) must be multiples of 128", cols, ldo); return -1; }
    hipLaunchKernelGGL(quant_rows_kernel, dim3(rows), dim3(256), 0, s, W, cols, out, ldo, scale, scaled_rows, pre);
    LAUNCH_OK("quant_rows_kernel");
    return 0;
}

int pevit_launch_quant_transpose_fp8(const float* W, int rows, int cols, const float* scale, unsigned char* outT, int ldo,
                                     int scaled_rows, float pre, hipStream_t s) {
    if (rows % 128 || ldo % 128) { pevit_set_error("quant_transpose_fp8: %d output channels (ld %d) must be multiples of 128", rows, ldo); return -1; }
    hipLaunchKernelGGL(quant_transpose_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32)), dim3(256), 0, s, W, rows, cols, scale,
                       outT, ldo, scaled_rows, pre);
    LAUNCH_OK("quant_transpose_kernel");
    return 0;
}

// unscaled, saturating e4m3 codes of an activation matrix [rows][cols], k-permuted per 128 like the weights (the layout the
// fp8 x fp8 products read their A operand in; in the step the producers write it themselves)
__global__ void cast_fp8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t rows, int cols) {
    const size_t n8 = rows * (size_t)cols / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (cols / 8);
        const int c = (int)(i - r * (cols / 8)) * 8;
        const float4 a = *reinterpret_cast<const float4*>(src + r * cols + c), b = *reinterpret_cast<const float4*>(src + r * cols + c + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        store8_fp8(dst + r * cols, c, v);
    }
}
int pevit_launch_cast_fp8(const float* src, unsigned char* dst, size_t rows, int cols, hipStream_t s) {
    if (cols % 128) { pevit_set_error("cast_fp8: width %d must be a multiple of 128", cols); return -1; }
    if (rows == 0) return 0;
    const size_t n8 = rows * (size_t)cols / 8;
    const int blocks = (int)((n8 + 255) / 256);
    hipLaunchKernelGGL(cast_fp8_kernel, dim3(blocks > 4096 ? 4096 : blocks), dim3(256), 0, s, src, dst, rows, cols);
    LAUNCH_OK("cast_fp8_kernel");
    return 0;
}

int pevit_launch_cast_bf16_cols(const float* src, bf16* dst, size_t rows, int cols, const float* colscale, hipStream_t s) {
    if (cols % 4) { pevit_set_error("cast_bf16_cols: width %d must be a multiple of 4", cols); return -1; }
    if (rows == 0) return 0;
    const size_t n4 = rows * (size_t)cols / 4;
    const int blocks = (int)((n4 + 255) / 256);
    hipLaunchKernelGGL(cast_bf16_cols_kernel, dim3(blocks > 4096 ? 4096 : blocks), dim3(256), 0, s, src, dst, rows, cols, colscale);
    LAUNCH_OK("cast_bf16_cols_kernel");
    return 0;
}

int pevit_launch_dequant_rows_fp8(const unsigned char* codes, int ldc, const float* scale, int rows, int cols, float* out,
                                  hipStream_t s) {
    hipLaunchKernelGGL(dequant_rows_kernel, dim3(rows), dim3(256), 0, s, codes, ldc, scale, cols, out);
    LAUNCH_OK("dequant_rows_kernel");
    return 0;
}
