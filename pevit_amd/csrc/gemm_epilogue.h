// Fused GEMM epilogues, shared by the MFMA kernel (gemm.hip, bf16 storage) and the f32-class verification GEMM
// (verify.hip, f32 storage): identical arithmetic, the storage type of the bf16-declared buffers is the template
// parameter ST.
#pragma once
#include "common.h"
#include "kernels.h"

// 1 / (1 + e^-x) with the hardware exp2 and reciprocal (1 ulp each; the results are rounded to bf16): the IEEE division
// hipcc emits for 1.0f / x is ~10 VALU instructions and made the GELU epilogues VALU-bound (18 instructions per element)
__device__ __forceinline__ float sigmoidf_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
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

__device__ __forceinline__ void add8(float v[8], const float* src) {
    const float4 b0 = *reinterpret_cast<const float4*>(src);
    const float4 b1 = *reinterpret_cast<const float4*>(src + 4);
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
}
__device__ __forceinline__ void mul8(float v[8], const float* src) {
    const float4 b0 = *reinterpret_cast<const float4*>(src);
    const float4 b1 = *reinterpret_cast<const float4*>(src + 4);
    v[0] *= b0.x; v[1] *= b0.y; v[2] *= b0.z; v[3] *= b0.w;
    v[4] *= b1.x; v[5] *= b1.y; v[6] *= b1.z; v[7] *= b1.w;
}
#ifndef PEVIT_NT_STORES
#define PEVIT_NT_STORES 0      // measurement builds (scripts/build_variants.sh): 1 = bf16 epilogue stores non-temporal, 2 = f32 ones
#endif
__device__ __forceinline__ void store8f(float* dst, const float v[8]) {
#if PEVIT_NT_STORES & 2
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f32x4v{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4v*>(dst));
    __builtin_nontemporal_store(f32x4v{v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4v*>(dst + 4));
#else
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
#endif
}
// activation storage: bf16 (production) or f32 (the f32-class verification mode, PEVIT_W_F32_VERIFY); pointers are
// declared bf16* throughout and reinterpreted here, element offsets are in elements of ST
template <typename ST> __device__ __forceinline__ void store8s(bf16* base, size_t off, const float v[8]) {
    if constexpr (sizeof(ST) == 2) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
#if PEVIT_NT_STORES & 1
        __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(base + off));
#else
        store_bf16x8(base + off, o);
#endif
    } else {
        store8f(reinterpret_cast<float*>(base) + off, v);
    }
}
template <typename ST> __device__ __forceinline__ void load8s(const bf16* base, size_t off, float v[8]) {
    if constexpr (sizeof(ST) == 2) {
        const bf16x8 h = load_bf16x8(base + off);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = bf2f(h[i]);
    } else {
        const float* p = reinterpret_cast<const float*>(base) + off;
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}
// the value a stored activation will be read back as (bf16 rounding in production, identity in f32 mode)
template <typename ST> __device__ __forceinline__ float as_stored(float x) {
    if constexpr (sizeof(ST) == 2) return bf2f(f2bf(x)); else return x;
}

// epilogues that read a saved activation (GemmParams::aux) at the output position
template <int EPI> constexpr bool epi_reads_aux = (EPI == EPI_DGELU_BF16 || EPI == EPI_DRELU_BF16 || EPI == EPI_DGELUNEW_BF16);
// epilogues whose only other operand is a per-column constant (bias / channel scale): the staggered 8-wave kernel loads
// it ONCE per lane and runs these epilogues branch-free (epilogue_store_pre)
template <int EPI> constexpr bool epi_has_pre = (EPI == EPI_QKV_HEADS || EPI == EPI_BIAS_GELU || EPI == EPI_DGELU_BF16);

// column constants of the 8 columns col..col+7: the bias; for dGELU the channel scale folded into the output (1 when absent)
template <int EPI>
__device__ __forceinline__ void epi_load_cols(const GemmParams& p, int col, float c[8]) {
    const float* src = nullptr;
    float fill = 0.0f;
    if constexpr (EPI == EPI_QKV_HEADS) { if (col < 3 * p.E) src = p.bias + col; }
    else if constexpr (EPI == EPI_BIAS_GELU) src = p.bias + col;
    else if constexpr (EPI == EPI_DGELU_BF16) { fill = 1.0f; if (p.oscale) src = p.oscale + col; }
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = fill;
    if (src) {
        const float4 b0 = *reinterpret_cast<const float4*>(src), b1 = *reinterpret_cast<const float4*>(src + 4);
        c[0] = b0.x; c[1] = b0.y; c[2] = b0.z; c[3] = b0.w; c[4] = b1.x; c[5] = b1.y; c[6] = b1.z; c[7] = b1.w;
    }
}

// the epi_has_pre epilogues with their column constants (and, for dGELU, the saved activation h) already in registers: no
// load, no data-dependent branch around a memory operation -- so that hipcc can count its s_waitcnt vmcnt(N) exactly
// instead of falling back to vmcnt(0), which on gfx950 also waits for every STORE issued so far
template <int EPI, typename ST>
__device__ __forceinline__ void epilogue_store_pre(const GemmParams& p, int row, int col, float v[8], const float c[8], const float h[8]) {
    static_assert(epi_has_pre<EPI>, "epilogue with further operands");
    if constexpr (EPI == EPI_QKV_HEADS) {
        const int E3 = 3 * p.E;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += c[i];
        if (col < E3) {
            const int which = col / p.E, ce = col - which * p.E;
            const int hh = ce >> 6, d = ce & 63;
            const int b = row / p.Ntok, n = row - b * p.Ntok;
            store8s<ST>(p.outb, (size_t)which * p.head_stride + ((size_t)(b * p.H + hh) * p.Ntok + n) * 64 + d, v);
        } else {
            store8f(p.outf + (size_t)row * p.ldo + (col - E3), v);
        }
    } else if constexpr (EPI == EPI_BIAS_GELU) {
        float g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // QuickGELU (model.py:163-165) evaluated on the pre-activation AS STORED (bf16-rounded in production), which
            // is what the backward pass will see, so fwd and bwd agree on the same h.
            v[i] = as_stored<ST>(v[i] + c[i]);
            g[i] = v[i] * sigmoidf_fast(1.702f * v[i]);
        }
        // the saved pre-activation is not read again before the backward pass: non-temporal store (+0.7 % per step)
        if constexpr (sizeof(ST) == 2) {
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
            __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p.outb + (size_t)row * p.ldob + col));
        } else {
            store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
        }
        if (p.out2_fp8) store8_fp8(reinterpret_cast<unsigned char*>(p.outb2) + (size_t)row * p.ldob2, col, g);   // consumer: fp8 x fp8 c_proj
        else store8s<ST>(p.outb2, (size_t)row * p.ldob2 + col, g);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float hv = h[i];
            const float s = sigmoidf_fast(1.702f * hv);
            // fp8 weights: the consumer (c_fc backward) contracts over these columns, whose power-of-two channel scales
            // are folded into its A operand here (exact; c = 1 otherwise)
            v[i] = v[i] * (s * (1.0f + 1.702f * hv * (1.0f - s))) * c[i];
        }
        store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
    }
}

// ... the remaining aux epilogues (bottleneck adapters), activation in registers
template <int EPI, typename ST>
__device__ __forceinline__ void epilogue_store_aux(const GemmParams& p, int row, int col, float v[8], const float h[8]) {
    static_assert(epi_reads_aux<EPI>, "epilogue without an aux operand");
    if constexpr (EPI == EPI_DGELU_BF16) {
        float c[8];
        epi_load_cols<EPI>(p, col, c);
        epilogue_store_pre<EPI, ST>(p, row, col, v, c, h);
        return;
    } else if constexpr (EPI == EPI_DRELU_BF16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = h[i] > 0.f ? v[i] : 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * gelu_new_grad_f(h[i]);
    }
    store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
}

template <int EPI, typename ST>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int row, int col, float v[8]) {
    // row < M and col < N (col multiple of 8) are guaranteed by the caller.
    if constexpr (EPI == EPI_QKV_HEADS || EPI == EPI_BIAS_GELU) {
        float c[8];
        epi_load_cols<EPI>(p, col, c);
        epilogue_store_pre<EPI, ST>(p, row, col, v, c, c);
    } else if constexpr (EPI == EPI_BIAS_RESID_F32) {
        add8(v, p.bias + col);
        add8(v, p.resid + (size_t)row * p.ldr + col);
        store8f(p.outf + (size_t)row * p.ldo + col, v);
    } else if constexpr (epi_reads_aux<EPI>) {
        float h[8];
        load8s<ST>(p.aux, (size_t)row * p.ldaux + col, h);
        epilogue_store_aux<EPI, ST>(p, row, col, v, h);
    } else if constexpr (EPI == EPI_F32) {
        store8f(p.outf + (size_t)row * p.ldo + col, v);
    } else if constexpr (EPI == EPI_BF16) {
        store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
    } else if constexpr (EPI == EPI_BIAS_BF16) {
        add8(v, p.bias + col);
        store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
    } else if constexpr (EPI == EPI_PATCH_EMBED) {
        // row = b*G2 + g (patch index), output row = b*Ntok + 1 + g ; + positional embedding
        const int G2 = p.Ntok - 1;
        const int b = row / G2, g = row - b * G2;
        add8(v, p.resid + (size_t)(1 + g) * p.ldr + col);
        store8f(p.outf + ((size_t)b * p.Ntok + 1 + g) * p.ldo + col, v);
    } else if constexpr (EPI == EPI_BIAS_RESID_KEEP) {
        add8(v, p.bias + col);
        store8f(p.outf2 + (size_t)row * p.ldo2 + col, v);
        add8(v, p.resid + (size_t)row * p.ldr + col);
        store8f(p.outf + (size_t)row * p.ldo + col, v);
    } else if constexpr (EPI == EPI_BIAS_GELUNEW) {
        add8(v, p.bias + col);
        float g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = as_stored<ST>(v[i]);
            g[i] = gelu_new_f(v[i]);
        }
        store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
        store8s<ST>(p.outb2, (size_t)row * p.ldob2 + col, g);
    } else if constexpr (EPI == EPI_BIAS_RELU_BF16) {
        add8(v, p.bias + col);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.0f);
        store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Load-free form of EVERY epilogue, for kernels that walk a tile fragment by fragment (gemm_kernel): the per-column
// constants (epi_cols: bias / channel scale) are loaded once per tile, the per-element operand (f32 residual, positional
// embedding, saved activation) is requested one fragment AHEAD from clamped addresses (epi_prefetch) -- so no load sits between
// two stores or inside a bounds branch, and every s_waitcnt hipcc places is an exact vmcnt(N) over loads that are OLDER than the
// stores in flight.  (A load issued after a store is only consumed once that store has been acknowledged: gfx950 counts both
// in vmcnt, in order.  On these tiles that was one store latency per 16-row pass.)
template <int EPI> constexpr bool epi_reads_resid = (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_BIAS_RESID_KEEP || EPI == EPI_PATCH_EMBED);
template <int EPI> constexpr bool epi_has_bias = (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID_F32 || EPI == EPI_BIAS_RESID_KEEP ||
                                                  EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_GELUNEW || EPI == EPI_BIAS_RELU_BF16);
struct EpiOperand { float4 r0, r1; bf16x8 a; };

// column constants of columns col..col+7 (col < N): bias where the epilogue has one, the dGELU channel scale, else unused
template <int EPI>
__device__ __forceinline__ void epi_cols(const GemmParams& p, int col, float c[8]) {
    if constexpr (epi_has_pre<EPI>) epi_load_cols<EPI>(p, col, c);
    else if constexpr (epi_has_bias<EPI>) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col), b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        c[0] = b0.x; c[1] = b0.y; c[2] = b0.z; c[3] = b0.w; c[4] = b1.x; c[5] = b1.y; c[6] = b1.z; c[7] = b1.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = 0.f;
    }
}

// the per-element operand at (row, col), both already clamped into the matrix: unconditional loads
template <int EPI, typename ST>
__device__ __forceinline__ void epi_prefetch(const GemmParams& p, int row, int col, EpiOperand& o) {
    if constexpr (EPI == EPI_PATCH_EMBED) {
        const int G2 = p.Ntok - 1;
        const int g = row - (row / G2) * G2;
        const float* src = p.resid + (size_t)(1 + g) * p.ldr + col;
        o.r0 = *reinterpret_cast<const float4*>(src); o.r1 = *reinterpret_cast<const float4*>(src + 4);
    } else if constexpr (epi_reads_resid<EPI>) {
        const float* src = p.resid + (size_t)row * p.ldr + col;
        o.r0 = *reinterpret_cast<const float4*>(src); o.r1 = *reinterpret_cast<const float4*>(src + 4);
    } else if constexpr (epi_reads_aux<EPI>) {
        static_assert(sizeof(ST) == 2, "bf16 activation storage");
        o.a = load_bf16x8(p.aux + (size_t)row * p.ldaux + col);
    }
}

// row < M and col < N (col multiple of 8) are guaranteed by the caller; c = epi_cols, o = epi_prefetch of this position
template <int EPI, typename ST>
__device__ __forceinline__ void epilogue_store_full(const GemmParams& p, int row, int col, float v[8], const float c[8], const EpiOperand& o) {
    if constexpr (EPI == EPI_QKV_HEADS || EPI == EPI_BIAS_GELU) {
        epilogue_store_pre<EPI, ST>(p, row, col, v, c, c);
    } else if constexpr (epi_reads_aux<EPI>) {
        float h[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = bf2f(o.a[i]);
        if constexpr (EPI == EPI_DGELU_BF16) epilogue_store_pre<EPI, ST>(p, row, col, v, c, h);
        else epilogue_store_aux<EPI, ST>(p, row, col, v, h);
    } else if constexpr (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_BIAS_RESID_KEEP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += c[i];
        if constexpr (EPI == EPI_BIAS_RESID_KEEP) store8f(p.outf2 + (size_t)row * p.ldo2 + col, v);
        v[0] += o.r0.x; v[1] += o.r0.y; v[2] += o.r0.z; v[3] += o.r0.w; v[4] += o.r1.x; v[5] += o.r1.y; v[6] += o.r1.z; v[7] += o.r1.w;
        store8f(p.outf + (size_t)row * p.ldo + col, v);
    } else if constexpr (EPI == EPI_PATCH_EMBED) {
        const int G2 = p.Ntok - 1;
        const int b = row / G2, g = row - b * G2;
        v[0] += o.r0.x; v[1] += o.r0.y; v[2] += o.r0.z; v[3] += o.r0.w; v[4] += o.r1.x; v[5] += o.r1.y; v[6] += o.r1.z; v[7] += o.r1.w;
        store8f(p.outf + ((size_t)b * p.Ntok + 1 + g) * p.ldo + col, v);
    } else if constexpr (EPI == EPI_F32) {
        store8f(p.outf + (size_t)row * p.ldo + col, v);
    } else if constexpr (EPI == EPI_BF16) {
        store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
    } else if constexpr (EPI == EPI_BIAS_BF16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += c[i];
        store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
    } else if constexpr (EPI == EPI_BIAS_GELUNEW) {
        float g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = as_stored<ST>(v[i] + c[i]);
            g[i] = gelu_new_f(v[i]);
        }
        store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
        store8s<ST>(p.outb2, (size_t)row * p.ldob2 + col, g);
    } else if constexpr (EPI == EPI_BIAS_RELU_BF16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i] + c[i], 0.0f);
        store8s<ST>(p.outb, (size_t)row * p.ldob + col, v);
    }
}
