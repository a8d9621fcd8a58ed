// Internal helpers shared by the translation units of libmi355dsp.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_complex.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mi355dsp.h"

namespace mdsp {

// ---------------------------------------------------------------- error reporting (thread-local message)
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

#define MDSP_FAIL(code, ...) return ::mdsp::set_error((code), __VA_ARGS__)
#define MDSP_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t mdsp_e_ = (expr);                                                                     \
        if (mdsp_e_ != hipSuccess)                                                                       \
            return ::mdsp::set_error(mdsp_e_ == hipErrorOutOfMemory ? MDSP_ERR_NOMEM : MDSP_ERR_DEVICE,  \
                                     "%s failed: %s (%s:%d)", #expr, hipGetErrorString(mdsp_e_),         \
                                     __FILE__, __LINE__);                                                \
    } while (0)
#define MDSP_TRY(expr)                 \
    do {                               \
        int mdsp_s_ = (expr);          \
        if (mdsp_s_ != MDSP_OK)        \
            return mdsp_s_;            \
    } while (0)
// kernel launch check: launch errors only (no sync)
#define MDSP_LAUNCH_CHECK() MDSP_HIP(hipGetLastError())

// ---------------------------------------------------------------- element types
inline bool dtype_valid(int dt) { return dt >= MDSP_F32 && dt <= MDSP_C64; }
inline bool dtype_is_complex(int dt) { return dt == MDSP_C32 || dt == MDSP_C64; }
inline bool dtype_is_double(int dt) { return dt == MDSP_F64 || dt == MDSP_C64; }
inline size_t dtype_size(int dt) {
    switch (dt) {
        case MDSP_F32: return 4;
        case MDSP_F64: return 8;
        case MDSP_C32: return 8;
        default: return 16;
    }
}
inline int dtype_real_of(int dt) { return dtype_is_double(dt) ? MDSP_F64 : MDSP_F32; }
inline int dtype_complex_of(int dt) { return dtype_is_double(dt) ? MDSP_C64 : MDSP_C32; }

template <typename R> struct cplx { R x, y; };
using cf32 = cplx<float>;
using cf64 = cplx<double>;

// ---------------------------------------------------------------- device buffer with RAII
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // grow-only reservation
    int reserve(size_t n) {
        if (n <= bytes) return MDSP_OK;
        release();
        if (n == 0) return MDSP_OK;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            return set_error(MDSP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
        return MDSP_OK;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// device properties (cached per process)
int device_cu_count();

}  // namespace mdsp
