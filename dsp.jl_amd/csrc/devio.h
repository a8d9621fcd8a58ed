// Device-side streaming I/O through buffer descriptors (SRSRC): 32-bit per-lane byte offsets, hardware
// bounds checking.  Reads past the end of a signal (and, via an out-of-range sentinel offset, before its
// start) return 0 and stores there are dropped -- which is exactly the zero padding the reference does by
// hand for the first / last overlap-save blocks (Filters/filt.jl:505-510) and frame tails.
//
// Descriptors are built from wave-uniform values passed through readfirstlane so hipcc keeps them in SGPRs
// (no waterfall loops, cdna guide T20).  A descriptor addresses at most 2 GiB; callers re-base it per unit of
// work, so a 4 GiB+ column is fine.
#pragma once

#include <hip/hip_runtime.h>

#include "fft_lds.h"

namespace mdsp {
namespace io {

using fft::cx;

constexpr int OOB = (int)0x80000000;  // byte offset that is out of range for every descriptor built here

// Cache policy (the `aux` immediate of the buffer instructions: bit 0 sc0, bit 1 nt, bit 4 sc1) of the streaming loads / stores.
#ifndef MDSP_IO_AUX_LOAD
#define MDSP_IO_AUX_LOAD 0
#endif
#ifndef MDSP_IO_AUX_STORE
#define MDSP_IO_AUX_STORE 0
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long long bytes) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    long long nb = bytes < 0 ? 0 : (bytes > 0x7ffffff0ll ? 0x7ffffff0ll : bytes);
    const unsigned n = __builtin_amdgcn_readfirstlane((unsigned)nb);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)n, 0x00020000);
}

template <typename T> struct Ld;
template <> struct Ld<float> {
    static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t r, int off) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, MDSP_IO_AUX_LOAD));
    }
    static __device__ __forceinline__ void store(float v, __amdgpu_buffer_rsrc_t r, int off) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, MDSP_IO_AUX_STORE);
    }
};
template <> struct Ld<double> {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ double load(__amdgpu_buffer_rsrc_t r, int off) {
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, MDSP_IO_AUX_LOAD);
        return __hiloint2double((int)v.y, (int)v.x);
    }
    static __device__ __forceinline__ void store(double d, __amdgpu_buffer_rsrc_t r, int off) {
        u2 v;
        v.x = (unsigned)__double2loint(d);
        v.y = (unsigned)__double2hiint(d);
        __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, MDSP_IO_AUX_STORE);
    }
};
template <> struct Ld<cx<float>> {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ cx<float> load(__amdgpu_buffer_rsrc_t r, int off) {
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, MDSP_IO_AUX_LOAD);
        return {__uint_as_float(v.x), __uint_as_float(v.y)};
    }
    static __device__ __forceinline__ void store(cx<float> c, __amdgpu_buffer_rsrc_t r, int off) {
        u2 v;
        v.x = __float_as_uint(c.x);
        v.y = __float_as_uint(c.y);
        __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, MDSP_IO_AUX_STORE);
    }
};
template <> struct Ld<cx<double>> {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ cx<double> load(__amdgpu_buffer_rsrc_t r, int off) {
        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, MDSP_IO_AUX_LOAD);
        return {__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z)};
    }
    static __device__ __forceinline__ void store(cx<double> c, __amdgpu_buffer_rsrc_t r, int off) {
        u4 v;
        v.x = (unsigned)__double2loint(c.x);
        v.y = (unsigned)__double2hiint(c.x);
        v.z = (unsigned)__double2loint(c.y);
        v.w = (unsigned)__double2hiint(c.y);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, MDSP_IO_AUX_STORE);
    }
};

// Strided unit I/O: thread t of a T-thread transform owns elements i = t + T*e, e = 0..E-1, of a window that
// starts `lead` elements BEFORE the first addressable element of the descriptor (lead >= 0, wave-uniform):
//   * elements with i < lead do not exist (zero padding in front of x / discarded aliased outputs);
//   * the descriptor is based at window element `lead`... no: it is based at window element 0, i.e. up to `lead`
//     elements before the array.  Those addresses are never touched: whole-element groups below `lead` are
//     skipped by wave-uniform branches and the one straddling group uses the OOB sentinel per lane.
//   * past-the-end elements are handled by the descriptor's num_records (reads 0 / store dropped).
// All per-element offsets are one VGPR (t*SZ) plus immediates; nothing per-element is loop-invariant state.
template <typename T, int E, int TT> __device__ __forceinline__ void load_window(T (&out)[E], __amdgpu_buffer_rsrc_t r, int lead, int t) {
    constexpr int SZ = (int)sizeof(T);
    int off = t * SZ;
    asm volatile("" : "+v"(off));  // keep LICM from materialising E offset VGPRs outside the persistent loop
    if (lead == 0) {               // wave-uniform; the steady state: one VGPR offset + immediates, no per-lane tests
#pragma unroll
        for (int e = 0; e < E; ++e) out[e] = Ld<T>::load(r, off + TT * e * SZ);
    } else {                       // leading blocks only: lanes in front of the signal read through the OOB sentinel (-> 0)
#pragma unroll
        for (int e = 0; e < E; ++e) out[e] = Ld<T>::load(r, (t + TT * e) < lead ? OOB : off + TT * e * SZ);
    }
}

// elements e = E0 .. E-1 only (the tail of a window whose head is already in registers)
template <typename T, int E, int TT, int E0> __device__ __forceinline__ void load_window_tail(T (&out)[E], __amdgpu_buffer_rsrc_t r, int t) {
    constexpr int SZ = (int)sizeof(T);
    int off = t * SZ;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int e = E0; e < E; ++e) out[e] = Ld<T>::load(r, off + TT * e * SZ);
}

// Stores of elements ES .. E-1 of a window whose first `lead` elements are dropped, with ES = lead div TT known at compile time: elements above
// ES are plain stores (one VGPR offset + immediates), element ES drops its lanes t < lead - ES TT through the OOB sentinel, elements below ES
// issue nothing at all.
template <typename T, int E, int TT, int ES, typename F> __device__ __forceinline__ void store_window_from(F&& get, __amdgpu_buffer_rsrc_t w, int rem, int t, int off) {
    constexpr int SZ = (int)sizeof(T);
    Ld<T>::store(get(ES), w, t < rem ? OOB : off + TT * ES * SZ);
#pragma unroll
    for (int e = ES + 1; e < E; ++e) Ld<T>::store(get(e), w, off + TT * e * SZ);
}
template <typename T, int E, int TT, int ES, typename F> __device__ __forceinline__ void store_window_switch(F&& get, __amdgpu_buffer_rsrc_t w, int es, int rem, int t, int off) {
    if constexpr (ES < E) {
        if (es == ES) store_window_from<T, E, TT, ES>(get, w, rem, t, off);
        else store_window_switch<T, E, TT, ES + 1>(get, w, es, rem, t, off);
    }
}

#ifndef MDSP_IO_STORE_SWITCH
#define MDSP_IO_STORE_SWITCH 1   // 0: the round-2 form (one compare + select per element and store)
#endif
template <typename T, int E, int TT, typename F> __device__ __forceinline__ void store_window(F&& get, __amdgpu_buffer_rsrc_t w, int lead, int t) {
    constexpr int SZ = (int)sizeof(T);
    int off = t * SZ;
    asm volatile("" : "+v"(off));
    if (lead == 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) Ld<T>::store(get(e), w, off + TT * e * SZ);
    } else if (MDSP_IO_STORE_SWITCH && (TT & (TT - 1)) == 0) {
        // `lead` is wave-uniform (and loop-invariant in the overlap-save kernels: nb - 1): which elements lie wholly below it is a scalar
        // decision, so the per-element compare + select of the branch-free form (2 VALU operations per store, 8 % of the overlap-save
        // kernel's vector instructions) shrinks to ONE select for the element that straddles `lead`, and the dropped elements cost no
        // VMEM slot either.  The chain of wave-uniform compares runs on the scalar unit.
        const int es = __builtin_amdgcn_readfirstlane(lead / TT), rem = __builtin_amdgcn_readfirstlane(lead - (lead / TT) * TT);
        store_window_switch<T, E, TT, 0>(get, w, es, rem, t, off);   // es >= E: the whole window is dropped
    } else {                       // branch-free: lanes below `lead` store through the OOB sentinel (dropped)
#pragma unroll
        for (int e = 0; e < E; ++e) Ld<T>::store(get(e), w, (t + TT * e) < lead ? OOB : off + TT * e * SZ);
    }
}

// ---- LDS-DMA (buffer_load ... lds): global memory straight into LDS, no VGPRs, no ds_write pass ---------------------------------------
// Hand-issued: the compiler's wait-count bookkeeping would make every later ds_read wait for a DMA it knows about (it cannot tell LDS regions
// apart), which defeats a prefetch.  M0 carries the LDS address and belongs to the compiler, so it is saved and restored inside the statement.
// The issuing wave orders its own LDS reads behind a DMA with s_waitcnt vmcnt (DMA operations retire in issue order with the wave's other
// vector-memory operations); other waves additionally need a barrier.
typedef int dma_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_i4 dma_rsrc(const void* base, long long bytes) {   // raw buffer: nothing past `bytes` is moved
    const unsigned long long p = (unsigned long long)base;
    const long long nb = bytes < 0 ? 0 : (bytes > 0x7ffffff0ll ? 0x7ffffff0ll : bytes);
    return dma_i4{(int)__builtin_amdgcn_readfirstlane((unsigned)p), (int)(__builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) & 0xffffu),
                  (int)__builtin_amdgcn_readfirstlane((unsigned)nb), 0x00020000};
}
// 256 consecutive dwords, 16 bytes per lane: byte offset voff (+ lane * 16 inside it) -> lds_byte + 16 lane
__device__ __forceinline__ void dma256(dma_i4 rsrc, unsigned lds_byte, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(rsrc) : "memory");
}
// 64 consecutive dwords, 4 bytes per lane
__device__ __forceinline__ void dma64(dma_i4 rsrc, unsigned lds_byte, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(rsrc) : "memory");
}
template <typename T> __device__ __forceinline__ unsigned lds_byte_address(T* p) {   // LDS byte address of a __shared__ object
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) T*)p;
}

}  // namespace io
}  // namespace mdsp
