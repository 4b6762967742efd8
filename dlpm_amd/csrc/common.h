// common.h -- shared helpers for libdlpm_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/dlpm_amd.h"

namespace dlpm {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(dlpm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define DLPM_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            ::dlpm::set_error(__VA_ARGS__);       \
            return DLPM_ERR_ARG;                  \
        }                                         \
    } while (0)

#define DLPM_HIP(expr)                                                                        \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::dlpm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                      \
            return DLPM_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define DLPM_LAUNCH_CHECK() DLPM_HIP(hipGetLastError())

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: set it once per (device, kernel), under a lock
// (a process may drive several GPUs, and two host threads may launch a kernel for the first time together).
int ensure_dynamic_lds(const void *kernel, int bytes);

// t <- t - 1 on the device step counter (noise.hip): what DLPM_UPD_ADVANCE does after an update
int launch_step_advance(int32_t *t_dev, hipStream_t st);

// Optional per-launch timing (dlpm_prof_enable): brackets one launch with HIP events on its stream.
bool prof_enabled();
bool prof_detail();   // DLPM_PROF_DETAIL=1: one class per distinct launch shape
struct ProfScope {
    std::string name;
    double flops, bytes;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(const char *n, double fl, double by, hipStream_t s);
    ~ProfScope();
};


// Developer instrumentation, compiled in only with -DDLPM_PHASE_TIMING (DLPM_BUILD_DEFS): kernels that carry a
// ConvLaunch add clock64() deltas of their phases (thread 0 of every workgroup) into this device buffer.
#ifdef DLPM_PHASE_TIMING
unsigned long long *phase_buffer();   // 32 counters, zero-initialised device memory (allocated on first use)
#ifdef DLPM_PHASE_DEFER
// Round 6: the SAME counters, but nothing touches memory until the kernel's last instruction.  The in-place form below issues one
// 64-bit atomicAdd per marker from thread 0 of EVERY workgroup onto the same two cache lines; on gfx9 an atomic without return counts
// on vmcnt like a store, so the next s_waitcnt vmcnt(0) of that wave (the epilogue's bias / residual loads, the loop's first operand
// wait) waits for the atomic's round trip -- and with all 256 CUs in lockstep those atomics queue up behind each other at the one
// address.  That queue, not the arithmetic, was the "epilogue that costs 4x more on 256 workgroups than on 16"
// (profiles/r06/epilogue_lockstep/).  Here the deltas stay in scalar registers (s_memtime is a scalar instruction, the arithmetic
// is wave-uniform) and DLPM_PHASE_FLUSH adds them at the very end.  Markers i, i+1, i+2 of a kernel map to slots (i & 3).
#define DLPM_PHASE_DECL long long _pt = clock64(); unsigned long long _pd0 = 0, _pd1 = 0, _pd2 = 0
#define DLPM_PHASE(p, i)                                                                              \
    do {                                                                                              \
        const long long _n = clock64();                                                               \
        const unsigned long long _d = (unsigned long long)(_n - _pt);                                 \
        if (((i) & 3) == 0) _pd0 += _d; else if (((i) & 3) == 1) _pd1 += _d; else _pd2 += _d;          \
        _pt = _n;                                                                                     \
    } while (0)
#define DLPM_PHASE_FLUSH(p, base)                                                                     \
    do {                                                                                              \
        if ((p).phase && threadIdx.x == 0) {                                                          \
            atomicAdd((p).phase + (base), _pd0);                                                      \
            atomicAdd((p).phase + (base) + 1, _pd1);                                                  \
            atomicAdd((p).phase + (base) + 2, _pd2);                                                  \
        }                                                                                             \
    } while (0)
#else
#define DLPM_PHASE_DECL long long _pt = clock64()
#define DLPM_PHASE(p, i)                                                                              \
    do {                                                                                              \
        if ((p).phase && threadIdx.x == 0) {                                                          \
            const long long _n = clock64();                                                           \
            atomicAdd((p).phase + (i), (unsigned long long)(_n - _pt));                               \
            _pt = _n;                                                                                 \
        }                                                                                             \
    } while (0)
#define DLPM_PHASE_FLUSH(p, base)
#endif
#else
#define DLPM_PHASE_DECL
#define DLPM_PHASE(p, i)
#define DLPM_PHASE_FLUSH(p, base)
#endif

}  // namespace dlpm
