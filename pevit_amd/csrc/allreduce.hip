// pevit_allreduce_flat: the gradient exchange of data-parallel fine-tuning WITHOUT a collective-library kernel (SURVEY 8b
// "optional allreduce_flat", SURVEY section 5 "one-shot P2P push + local reduce").
//
// The exchanged buffer is tiny (KAdaptation ViT-B/32 + 100-class head: 101,476 floats = 406 KB, in three buckets), so a ring is all
// latency.  One-shot form, one process per GPU on one node:
//   * every rank owns a MAILBOX in its HBM: [2 parities][world][capacity] floats + [2][world] flag words, exported once through
//     hipIpcGetMemHandle and opened by every peer (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0);
//   * all-reduce number e (parity e & 1) of rank r:  ONE kernel (ar_push_kernel, round 5) stores the contribution into every
//     peer's mail[par][r] through the IPC-mapped addresses, fences at system scope and lets its last workgroup write every
//     peer's flag[par][r] = (e, n) with a single 8-byte release store.  (Round 4 pushed with the copy engines -- for every peer
//     hipMemcpyAsync(data) then hipMemcpyAsync(flag), "no compute unit involved".  Kept behind PEVIT_AR_PUSH=dma, but NOT
//     reliable: with eight processes on one device the 8-byte flag was seen torn and, in a 200-round soak, three reductions of one
//     rank never saw a peer's flag at all; the kernel form ran 400 rounds x 8 ranks clean.)
//   * a small local kernel waits (bounded) until flag[par][p] == e for every peer and overwrites buf with
//     sum_{p = 0 .. world-1} contribution_p  IN RANK ORDER -- the same order on every rank, so the replicas stay bit-identical,
//     and for world = 2 bit-identical to any other all-reduce (a two-term f32 sum has one value).  Peer data and flags are read
//     with system-scope loads (they were written by another device; this GPU's L2 may hold lines of the previous use).
//   * two parities: a rank that is through all-reduce e may push e+1 while a slower peer still reads e; it cannot reach e+2
//     before that peer has pushed e+1, i.e. after the peer finished reading parity e & 1.
// Round 5: the mailbox is allocated FINE-GRAINED where the runtime allows it (hipExtMallocWithFlags; peer DMA
// writes into coarse-grained, L2-cached memory are not guaranteed to become visible to spinning loads on a multi-XCD part) and every
// reading thread issues a system-scope acquire behind the barrier; a flag is the pair (epoch, n) so that ranks that disagree on
// the bucket size raise an error instead of summing garbage; ONE workgroup decides whether the launch reduces or gives up
// (block 0 polls the peers, the others read its verdict), so a bucket is reduced everywhere or nowhere; the host epoch advances
// only when every enqueue succeeded; the error word can be handed to the context (pevit_set_external_poison) so that the fused
// SGD kernel withholds the update of a step whose exchange failed, exactly as for a stream-K hand-off.
// EXPERIMENTAL: never run across two devices (no multi-GPU box in the build environment); bench.py's default exchange is RCCL.
// What has run: two / four / eight processes sharing ONE device (tests/test_gpu_allreduce.py) -- IPC export / open, the push, the flag
// protocol, the reduction order, three buckets per step inside engine.forward_backward_dp.  NOT measured: two GPUs (xGMI).
#include "../../include/pevit_hip.h"
#include "common.h"
#include "kernels.h"

#include <stdlib.h>
#include <string.h>

#include <new>

constexpr int AR_MAX_WORLD_C = 16;
namespace {
constexpr int AR_MAX_WORLD = AR_MAX_WORLD_C;
constexpr int AR_STAGE = 8;                 // ring of device words holding the epoch number a flag copy reads from
}  // namespace

struct pevit_ar {
    int rank = 0, world = 1;
    size_t cap = 0;                          // floats per contribution
    char* base = nullptr;                    // own allocation: mail | flags | stage | err
    char* peer[AR_MAX_WORLD] = {};           // IPC-opened allocations of the peers (own entry = base)
    bool opened[AR_MAX_WORLD] = {};
    unsigned epoch = 0;
    size_t off_flags = 0, off_stage = 0, off_err = 0, off_dec = 0, off_ticket = 0, bytes = 0;
    bool fine = false;                       // the allocation is fine-grained (hipExtMallocWithFlags)
    bool dma = false;                        // PEVIT_AR_PUSH=dma: push with the copy engines (round 4) instead of ar_push_kernel
};

namespace {

inline size_t mail_off(const pevit_ar* a, int par, int src) { return ((size_t)par * a->world + src) * a->cap * sizeof(float); }
inline size_t flag_off(const pevit_ar* a, int par, int src) { return a->off_flags + ((size_t)par * a->world + src) * sizeof(unsigned long long); }

// The push as a kernel (round 5, the default): grid (slices, world - 1); block (x, k) stores slice x of the contribution into the
// mailbox of peer (rank + 1 + k) % world through its IPC-mapped address; every block fences at system scope and takes a ticket; the
// LAST block of the launch (all data of all peers is then visible system-wide) writes the (epoch, n) flag of every peer with ONE
// 8-byte system-scope release store each.  This is the textbook release/acquire hand-off; the copy-engine form below it
// (PEVIT_AR_PUSH=dma) relies on two stream-ordered device-to-device copies landing in order and on an 8-byte copy being atomic --
// with eight processes on one device the flag pair was seen torn, and one sum in 1,600 was wrong.
struct ArPeers { char* base[AR_MAX_WORLD_C]; };
__global__ __launch_bounds__(256) void ar_push_kernel(const float* __restrict__ buf, size_t n, ArPeers peers, size_t mail_byte_off,
                                                      size_t flag_byte_off, int rank, int world, unsigned epoch, unsigned* ticket) {
    const int p = (rank + 1 + (int)blockIdx.y) % world;
    float* dst = reinterpret_cast<float*>(peers.base[p] + mail_byte_off);
    const size_t per = 2048, lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    if (((reinterpret_cast<uintptr_t>(buf) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const size_t lo4 = lo / 4, hi4 = hi / 4;               // lo is a multiple of 2048
        for (size_t i = lo4 + threadIdx.x; i < hi4; i += blockDim.x)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(buf)[i];
        for (size_t i = hi4 * 4 + threadIdx.x; i < hi; i += blockDim.x) dst[i] = buf[i];
    } else {
        for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) dst[i] = buf[i];
    }
    __threadfence_system();                                    // this thread's stores are visible to every agent
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == total - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // ready for the next launch (same stream: ordered)
            __threadfence_system();
            const unsigned long long flag = ((unsigned long long)(unsigned)n << 32) | epoch;
            for (int q = 0; q < world; ++q)
                if (q != rank)
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(peers.base[q] + flag_byte_off), flag, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// buf[i] = sum over ranks (rank order) of: own contribution (buf itself) / the peers' pushed copies in the local mailbox
__global__ __launch_bounds__(256) void ar_reduce_kernel(float* __restrict__ buf, size_t n, const float* mail, const unsigned long long* flags,
                                                        int rank, int world, size_t cap, unsigned epoch, unsigned* err,
                                                        long long spin_limit, unsigned* decision) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int good = 1;
        if (blockIdx.x == 0) {
            // the verdict of the whole launch: every peer's flag carries (this epoch, this n)
            const unsigned long long want = ((unsigned long long)(unsigned)n << 32) | epoch;
            unsigned why = 1u;
            for (int p = 0; p < world && good; ++p) {
                if (p == rank) continue;
                long long spins = 0, torn = 0;
                for (;;) {
                    const unsigned long long f = __hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (f == want) break;
                    // this epoch with another n: either the 8-byte flag copy is landing dword by dword (seen with eight processes on
                    // one device: the copy path does not write the pair atomically) -- it completes within microseconds -- or the
                    // ranks really disagree: then it stays that way, and the wait ends after a grace period instead of the full bound
                    if ((unsigned)(f & 0xffffffffu) == epoch && ++torn > 200000) { good = 0; why = 2u; break; }
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > spin_limit) { good = 0; break; }
                }
            }
            if (!good) atomicExch(err, why);  // 1: a peer never arrived, 2: the ranks disagree on the size; buf stays as it is
            __hip_atomic_store(decision, (epoch << 1) | (unsigned)good, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // block 0 of a grid is dispatched no later than any other block of it, so this wait cannot starve it
            long long spins = 0;
            unsigned d;
            while (((d = __hip_atomic_load(decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 1) != epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > 2 * spin_limit) { d = epoch << 1; atomicExch(err, 1u); break; }
            }
            good = (int)(d & 1u);
        }
        ok = good;
    }
    __syncthreads();
    if (!ok) return;
    // the contributions were written by another device (or its copy engine): nothing this CU or this XCD's L2 holds of them is valid
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const bool pair = i + 1 < n;
    float s0 = 0.f, s1 = 0.f;
    for (int p = 0; p < world; ++p) {
        float a, b = 0.f;
        if (p == rank) {
            a = buf[i]; if (pair) b = buf[i + 1];
        } else {
            const float* src = mail + (size_t)p * cap + i;
            if (pair && ((reinterpret_cast<uintptr_t>(src) & 7) == 0)) {
                const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_SYSTEM);
                a = __uint_as_float((unsigned)(v & 0xffffffffu)); b = __uint_as_float((unsigned)(v >> 32));
            } else {
                a = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                if (pair) b = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(src + 1), __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_SYSTEM));
            }
        }
        s0 += a; s1 += b;                    // p = 0, 1, ..., world-1: one order everywhere
    }
    buf[i] = s0; if (pair) buf[i + 1] = s1;
}

}  // namespace

extern "C" int pevit_ar_create(pevit_ar** out, int rank, int world, size_t max_floats) {
    if (!out || world < 1 || world > AR_MAX_WORLD || rank < 0 || rank >= world || max_floats == 0) {
        pevit_set_error("ar_create: bad argument (rank %d, world %d, capacity %zu; world <= %d)", rank, world, max_floats, AR_MAX_WORLD);
        return -1;
    }
    pevit_ar* a = new (std::nothrow) pevit_ar();
    if (!a) { pevit_set_error("ar_create: out of host memory"); return -1; }
    a->rank = rank; a->world = world; a->cap = (max_floats + 63) & ~(size_t)63;
    a->off_flags = align_up(2 * (size_t)world * a->cap * sizeof(float), 256);
    a->off_stage = align_up(a->off_flags + 2 * (size_t)world * sizeof(unsigned long long), 256);
    a->off_err = a->off_stage + AR_STAGE * 256;
    a->off_dec = a->off_err + 256;
    a->off_ticket = a->off_dec + 256;
    a->bytes = a->off_ticket + 256;
    { const char* pm = getenv("PEVIT_AR_PUSH"); a->dma = pm && !strcmp(pm, "dma"); }
    // fine-grained first (peer writes become visible without relying on this GPU's L2 being bypassed); PEVIT_AR_COARSE=1 or a
    // runtime that refuses falls back to the plain allocation
    const char* coarse = getenv("PEVIT_AR_COARSE");
    if (!(coarse && coarse[0] == '1') && hipExtMallocWithFlags((void**)&a->base, a->bytes, hipDeviceMallocFinegrained) == hipSuccess) a->fine = true;
    else {
        (void)hipGetLastError();
        a->base = nullptr;
        if (hipMalloc((void**)&a->base, a->bytes) != hipSuccess) { pevit_set_error("ar_create: hipMalloc of %zu bytes failed", a->bytes); delete a; return -1; }
    }
    if (hipMemset(a->base, 0, a->bytes) != hipSuccess) { pevit_set_error("ar_create: hipMemset failed"); (void)hipFree(a->base); delete a; return -1; }
    a->peer[rank] = a->base;
    *out = a;
    return 0;
}

extern "C" void pevit_ar_destroy(pevit_ar* a) {
    if (!a) return;
    for (int p = 0; p < a->world; ++p)
        if (a->opened[p]) (void)hipIpcCloseMemHandle(a->peer[p]);
    if (a->base) (void)hipFree(a->base);
    delete a;
}

extern "C" int pevit_ar_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int pevit_ar_export(pevit_ar* a, void* handle_out) {
    if (!a || !handle_out) { pevit_set_error("ar_export: null argument"); return -1; }
    hipIpcMemHandle_t h;
    if (a->fine && hipIpcGetMemHandle(&h, a->base) != hipSuccess) {
        // this runtime does not export fine-grained allocations: take the plain one (nothing has been pushed yet)
        (void)hipGetLastError();
        (void)hipFree(a->base);
        a->base = nullptr; a->fine = false;
        HIP_OK(hipMalloc((void**)&a->base, a->bytes));
        HIP_OK(hipMemset(a->base, 0, a->bytes));
        a->peer[a->rank] = a->base;
    }
    HIP_OK(hipIpcGetMemHandle(&h, a->base));
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

extern "C" int pevit_ar_import(pevit_ar* a, int peer, const void* handle) {
    if (!a || !handle || peer < 0 || peer >= a->world) { pevit_set_error("ar_import: bad argument"); return -1; }
    if (peer == a->rank) return 0;
    if (a->opened[peer]) { pevit_set_error("ar_import: peer %d already opened", peer); return -1; }
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    HIP_OK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    a->peer[peer] = (char*)p; a->opened[peer] = true;
    return 0;
}

// in place: buf[0:n] <- sum over the ranks of their buf[0:n], identical bits on every rank.  Asynchronous on `stream`.
extern "C" int pevit_allreduce_flat(pevit_ar* a, void* stream, float* buf, size_t n) {
    if (!a || !buf) { pevit_set_error("allreduce_flat: null argument"); return -1; }
    if (n == 0 || a->world == 1) return 0;
    if (n > a->cap) { pevit_set_error("allreduce_flat: %zu floats exceed the mailbox capacity %zu", n, a->cap); return -1; }
    for (int p = 0; p < a->world; ++p)
        if (!a->peer[p]) { pevit_set_error("allreduce_flat: peer %d has not been imported", p); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (n >> 32) { pevit_set_error("allreduce_flat: %zu floats do not fit the 32-bit size field of the flag", n); return -1; }
    const unsigned e = a->epoch + 1;                            // committed below, once every enqueue has succeeded
    const int par = (int)(e & 1u);
    if (!a->dma) {
        ArPeers peers;
        for (int p = 0; p < AR_MAX_WORLD; ++p) peers.base[p] = p < a->world ? a->peer[p] : nullptr;
        const unsigned slices = (unsigned)((n + 2047) / 2048);
        hipLaunchKernelGGL(ar_push_kernel, dim3(slices, a->world - 1), dim3(256), 0, s, buf, n, peers, mail_off(a, par, a->rank),
                           flag_off(a, par, a->rank), a->rank, a->world, e, reinterpret_cast<unsigned*>(a->base + a->off_ticket));
        LAUNCH_OK("ar_push_kernel");
    } else {
    unsigned* stage = reinterpret_cast<unsigned*>(a->base + a->off_stage + (size_t)(e % AR_STAGE) * 256);
    HIP_OK(hipMemsetD32Async((hipDeviceptr_t)stage, (int)e, 1, s));                     // flag = (n << 32) | epoch
    HIP_OK(hipMemsetD32Async((hipDeviceptr_t)(stage + 1), (int)(unsigned)n, 1, s));
    for (int k = 1; k < a->world; ++k) {                       // start with the right-hand neighbour: the pushes of the ranks spread over the links
        const int p = (a->rank + k) % a->world;
        HIP_OK(hipMemcpyAsync(a->peer[p] + mail_off(a, par, a->rank), buf, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIP_OK(hipMemcpyAsync(a->peer[p] + flag_off(a, par, a->rank), stage, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    }
    }
    const unsigned blocks = (unsigned)((n + 511) / 512);
    hipLaunchKernelGGL(ar_reduce_kernel, dim3(blocks), dim3(256), 0, s, buf, n,
                       reinterpret_cast<const float*>(a->base + mail_off(a, par, 0)),
                       reinterpret_cast<const unsigned long long*>(a->base + flag_off(a, par, 0)), a->rank, a->world, a->cap, e,
                       reinterpret_cast<unsigned*>(a->base + a->off_err), (long long)8000000,       // a few seconds of polling
                       reinterpret_cast<unsigned*>(a->base + a->off_dec) + par);
    LAUNCH_OK("ar_reduce_kernel");
    a->epoch = e;
    return 0;
}

// After an error the ranks' epochs may be out of step for good (one rank refused a call the others made, a peer died and was
// replaced): every rank drains its device and meets the others (the caller's job: a barrier on both sides of this call), then
// calls this -- flags, verdict words, error word and the host epoch return to their initial state.
extern "C" int pevit_ar_reset(pevit_ar* a, void* stream) {
    if (!a) { pevit_set_error("ar_reset: null argument"); return -1; }
    HIP_OK(hipMemsetAsync(a->base + a->off_flags, 0, a->bytes - a->off_flags, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    a->epoch = 0;
    return 0;
}

// device address of the error word (0 = fine, 1 = a peer never arrived, 2 = the ranks passed different sizes): hand it to
// pevit_set_external_poison so that the fused SGD kernel withholds the update of a step whose exchange failed
extern "C" const unsigned* pevit_ar_error_word(pevit_ar* a) { return a ? reinterpret_cast<const unsigned*>(a->base + a->off_err) : nullptr; }
extern "C" int pevit_ar_fine_grained(pevit_ar* a) { return a && a->fine ? 1 : 0; }

// 1 / 2 if a reduction ever gave up waiting for a peer / saw another size (synchronises the stream), 0 otherwise.  The word is NOT
// cleared (round 6): while it is raised the fused SGD kernel withholds every update (pevit_set_external_poison), and only
// pevit_ar_reset -- a collective: all ranks back to the initial protocol state -- lowers it, so that a caller who merely catches the
// exception cannot resume updating next to peers in another state
extern "C" int pevit_ar_error(pevit_ar* a, void* stream) {
    if (!a) return -1;
    unsigned v = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, a->base + a->off_err, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}
