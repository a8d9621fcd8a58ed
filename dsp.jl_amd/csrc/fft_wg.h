// Device-side driver for the workgroup FFT of fft_lds.h: runs the P passes with the inter-pass LDS exchanges and
// the synchronisation they need.
//
//  * T == 64 (one wavefront per transform): no s_barrier at all -- DS operations of one wave execute in issue
//    order, so a scheduling fence is enough.
//  * T > 64: one __syncthreads() per exchange when the LDS region is double-buffered (NBUF == 2; exchange k uses
//    buffer k & 1, so a buffer is only rewritten after a barrier that every reader of its previous contents has
//    passed), two per exchange with a single buffer.
#pragma once

#include <hip/hip_runtime.h>

#include "fft_lds.h"

#ifndef MDSP_FFT_SETPRIO
#define MDSP_FFT_SETPRIO 0   // 1: raise the wave priority across every inter-pass exchange (tools: build.py --tag prio --cflags -DMDSP_FFT_SETPRIO=1)
#endif

namespace mdsp {
namespace fft {

template <int T> __device__ __forceinline__ void wg_sync() {
    if constexpr (T <= 64) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
}

template <typename C, int PADSHIFT, int NBUF> constexpr int wg_lds_elems() { return NBUF * lds_elems<C::N, PADSHIFT>(); }

// XBASE: index (mod NBUF) of the buffer used by this transform's first exchange.
template <typename C, int DIR, int TWMODE, int PADSHIFT, int NBUF, int XBASE, int PASS = 0, int PERMUTE = false, typename R>
__device__ __forceinline__ void wg_fft(cx<R> (&v)[C::E], int t, const cx<R> (&tw)[C::NTW > 0 ? C::NTW : 1], const cx<R>* table, cx<R>* lds) {
    constexpr int BUF = (XBASE + PASS) % NBUF;
    cx<R>* region = lds + BUF * lds_elems<C::N, PADSHIFT>();
    pass_compute<C, DIR, PASS, TWMODE, PADSHIFT, PERMUTE>(v, t, tw, table, region);
    if constexpr (PASS < C::P - 1) {
        // PERMUTE == 2: the last exchange stays inside each wavefront (fft_lds.h) -- a scheduling fence instead of an s_barrier.  That mode
        // needs NBUF == 2 (exchange 0 in buffer 0, the private one in buffer 1) and the CALLER puts one workgroup barrier between two
        // transforms (buffer 0 is rewritten by the next transform's first pass).
        constexpr bool PRIVATE = PERMUTE == 2 && PASS == C::P - 2;
        static_assert(PERMUTE != 2 || (NBUF == 2 && lds_elems<C::N, PADSHIFT>() >= 4 * 4 * 272), "wave-private exchange uses the second LDS buffer (pad shift 4)");
#if MDSP_FFT_SETPRIO
        __builtin_amdgcn_s_setprio(1);   // experiment: the exchange (latency chain: barrier, LDS reads) outranks the partner wave's butterflies
#endif
        if constexpr (PRIVATE) wg_sync<64>();
        else wg_sync<C::T>();
        pass_reload<C, PADSHIFT, PASS + 1, PERMUTE>(v, t, region);
        if constexpr (NBUF == 1) wg_sync<C::T>();
#if MDSP_FFT_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        wg_fft<C, DIR, TWMODE, PADSHIFT, NBUF, XBASE, PASS + 1, PERMUTE>(v, t, tw, table, lds);
    }
}
// lane-permuted transform (fft_lds.h "lane permutations"): thread t holds X[io_lane<C, true>(t) + T*e] before and after
template <typename C, int DIR, int TWMODE, int PADSHIFT, int NBUF, int XBASE, typename R>
__device__ __forceinline__ void wg_fft_perm(cx<R> (&v)[C::E], int t, const cx<R> (&tw)[C::NTW > 0 ? C::NTW : 1], const cx<R>* table, cx<R>* lds) {
    wg_fft<C, DIR, TWMODE, PADSHIFT, NBUF, XBASE, 0, true>(v, t, tw, table, lds);
}

// Twiddle source setup for a kernel: registers (loaded once per persistent workgroup), an LDS-resident table shared
// by the workgroup's transform slots, or the global root table.  Returns the pointer pass_compute() should use.
template <typename C, int TWMODE, int PERMUTE = false, typename R>
__device__ __forceinline__ const cx<R>* wg_twiddle_setup(cx<R> (&tw)[C::NTW > 0 ? C::NTW : 1], cx<R>* twl, int t, int slot, const cx<R>* table) {
    static_assert(!PERMUTE || TWMODE == TW_REG || TWMODE == TW_GLOBAL, "lane permutations are wired for register / global twiddles");
    if constexpr (TWMODE == TW_REG) {
        load_twiddles<C, R, 1, TW_REG, PERMUTE>(tw, t, table);
        return table;
    } else if constexpr (TWMODE == TW_LDS) {
        if (slot == 0) fill_lds_twiddles<C, R>(twl, t, table);
        __syncthreads();
        return twl;
    } else if constexpr (TWMODE == TW_HYB) {
        load_twiddles<C, R, 1, TW_HYB>(tw, t, table);
        if (slot == 0) fill_lds_twiddles<C, R, 1, TW_HYB>(twl, t, table);
        __syncthreads();
        return twl;
    } else {
        return table;
    }
}

// Number of exchanges one transform performs.
template <typename C> constexpr int wg_exchanges() { return C::P - 1; }

}  // namespace fft
}  // namespace mdsp
