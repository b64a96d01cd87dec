// Shared device helpers for the pevit_amd gfx950 kernels.
// CDNA4 only: 64-wide wavefronts, MFMA bf16 (f32 accumulate), LDS-DMA staging.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define PEVIT_WAVE 64

#define HIP_OK(expr)                                                         \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) {                                              \
            pevit_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,     \
                            hipGetErrorString(_e));                          \
            return -1;                                                       \
        }                                                                    \
    } while (0)

extern "C" void pevit_set_error(const char* fmt, ...);

// after every kernel launch: a failed launch (bad grid, LDS over-subscription, ...) must not return 0
// through the C ABI.  hipGetLastError also clears the sticky error.
#define LAUNCH_OK(what)                                                               \
    do {                                                                              \
        hipError_t _e = hipGetLastError();                                            \
        if (_e != hipSuccess) {                                                       \
            pevit_set_error("%s: kernel launch failed: %s", what, hipGetErrorString(_e)); \
            return -1;                                                                \
        }                                                                             \
    } while (0)

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ bf16x8 load_bf16x8(const bf16* p) {
    return *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ void store_bf16x8(bf16* p, bf16x8 v) {
    *reinterpret_cast<bf16x8*>(p) = v;
}
__device__ __forceinline__ bf16x8 zero_bf16x8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
    return z;
}

// Activation / weight storage is bf16 in production and f32 in the f32-class verification mode (weight_format
// PEVIT_W_F32_VERIFY); such buffers are declared bf16* everywhere and accessed through these helpers in the kernels
// that serve both modes (ST = bf16 or float, offsets in elements of ST).
template <typename ST> __device__ __forceinline__ void st_store(bf16* base, size_t off, float v) {
    if constexpr (sizeof(ST) == 2) base[off] = f2bf(v); else reinterpret_cast<float*>(base)[off] = v;
}
template <typename ST> __device__ __forceinline__ float st_load(const bf16* base, size_t off) {
    if constexpr (sizeof(ST) == 2) return bf2f(base[off]); else return reinterpret_cast<const float*>(base)[off];
}
template <typename ST> __device__ __forceinline__ void st_store4(bf16* base, size_t off, float a, float b, float c, float d) {
    if constexpr (sizeof(ST) == 2) {
        bf16x4 o; o[0] = f2bf(a); o[1] = f2bf(b); o[2] = f2bf(c); o[3] = f2bf(d);
        *reinterpret_cast<bf16x4*>(base + off) = o;
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = make_float4(a, b, c, d);
    }
}

// 8 consecutive elements: one 16-byte store in bf16 storage (off must be a multiple of 8), two in f32 storage
template <typename ST> __device__ __forceinline__ void st_store8(bf16* base, size_t off, const float (&v)[8]) {
    if constexpr (sizeof(ST) == 2) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
        *reinterpret_cast<bf16x8*>(base + off) = o;
    } else {
        float* f = reinterpret_cast<float*>(base) + off;
        *reinterpret_cast<float4*>(f) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(f + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// Asynchronous global -> LDS copy, 16 bytes per lane.  The LDS destination is
// wave-uniform: the hardware writes lane l's 16 bytes at lds_wave_base + 16*l.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware, bijective block-id remap (8 XCDs, block b is dispatched to XCD b%8):
// gives every XCD a contiguous run of tile indices so that neighbouring tiles
// (which share an A row-panel) hit the same private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// fp8 (OCP e4m3) operands of the MX matrix instruction are stored k-permuted inside every group of 128 channels, so that the 32
// codes one MFMA lane needs of a 64-wide k-step are contiguous: [k-step of 64][lane half][16-group][8] (fp8.hip, gemm.hip)
__host__ __device__ __forceinline__ int fp8_kperm(int k) {
    const int grp = k & ~127, kk = k & 127;
    const int par = kk >> 6, ks = (kk >> 4) & 3, half = (kk >> 3) & 1, j = kk & 7;
    return grp + par * 64 + half * 32 + ks * 8 + j;
}
// 8 consecutive channels k0..k0+7 (k0 % 8 == 0) -> 8 contiguous e4m3 codes at fp8_kperm(k0); values beyond +-448 saturate
__device__ __forceinline__ void store8_fp8(unsigned char* row_base, int k0, const float v[8]) {
    float c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = fminf(fmaxf(v[i], -448.0f), 448.0f);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
    *reinterpret_cast<int2*>(row_base + fp8_kperm(k0)) = make_int2(lo, hi);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
