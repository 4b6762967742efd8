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
#define DLPM_PHASE_DECL long long _pt = clock64()
#define DLPM_PHASE(p, i)                                                                              \
    do {                                                                                              \
        if ((p).phase && threadIdx.x == 0) {                                                          \
            const long long _n = clock64();                                                           \
            atomicAdd((p).phase + (i), (unsigned long long)(_n - _pt));                               \
            _pt = _n;                                                                                 \
        }                                                                                             \
    } while (0)
#else
#define DLPM_PHASE_DECL
#define DLPM_PHASE(p, i)
#endif

}  // namespace dlpm
