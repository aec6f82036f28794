// Shared helpers of the gfx950 kernels (wave = 64 lanes, f32 MFMA 32x32x2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/bigc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// D(32x32) += A(32x2) * B(2x32); lane l supplies A[l&31][l>>5] and B[l>>5][l&31];
// result register r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

int bigc_fail(int code, const char* fmt, ...);
#define BIGC_CHECK_ARG(cond, ...) do { if (!(cond)) return bigc_fail(BIGC_ERR_ARG, __VA_ARGS__); } while (0)
#define BIGC_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "%s launch: %s", name, hipGetErrorString(e_)); } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
