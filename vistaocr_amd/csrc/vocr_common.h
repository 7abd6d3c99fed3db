// Shared device/host helpers for libvocr (gfx950 only: 64-lane waves, f32 MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vocr.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VOCR_WAVE 64

// Experiment switches (A/B of kernel variants, tile shapes, launch plans).  The shipped library has ONE code path per shape: every
// switch is its default, a compile-time constant.  A `-DVOCR_EXPERIMENTS` build (scripts/_lib_ab.py) reads them from the environment.
// The two runtime knobs an operator may need stay outside this macro and are documented in include/vocr.h: VOCR_LSTM_SWEEP=step
// (one launch per time step: several processes sharing one GPU) and VOCR_LSTM_WRITE_THROUGH=1.
#ifdef VOCR_EXPERIMENTS
#include <stdlib.h>
#define VOCR_EXPERIMENT_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define VOCR_EXPERIMENT_INT(name, dflt) (dflt)
#endif

void vocr_set_error(const char* fmt, ...);

#define VOCR_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            vocr_set_error(__VA_ARGS__);          \
            return VOCR_EINVAL;                   \
        }                                         \
    } while (0)

#define VOCR_CHECK_LAUNCH(name)                                                       \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            vocr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return VOCR_ELAUNCH;                                                      \
        }                                                                             \
    } while (0)

static inline int vocr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
