// common.h -- shared helpers of libampnet_hip.so (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/ampnet_hip.h"

namespace ampnet {

constexpr int WAVE = 64;

// thread-local error text returned by ampnet_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AMPNET_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return AMPNET_OK;
}

#define AMPNET_REQUIRE(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) return ::ampnet::fail(AMPNET_E_ARG, __VA_ARGS__); \
    } while (0)

// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg): off by default,
// when on every instrumented launch is bracketed by two events; ampnet_profile_read() synchronises and sums.
struct ProfScope {
    ProfScope(const char *name, double flops, double bytes, hipStream_t st);
    ~ProfScope();
    int slot;
    hipStream_t st;
};

// process-wide MFMA operand precision (ampnet_set_matrix_precision)
int matrix_precision();

// forward-workspace precision tags (core.hip): set by a train-mode forward, checked by its backward
void ws_tag_set(const void *ws, int mode);
int ws_tag_check(const void *ws, const char *who);

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace ampnet
