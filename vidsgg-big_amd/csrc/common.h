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

// two f32 -> packed f16 pair, ROUND TO NEAREST (v_cvt_pk_f16_f32).  The split-operand kernels use it for both halves of
// x = h + r 2^-11: with the round-toward-zero pack (v_cvt_pkrtz) both halves were biased towards zero (2^-22 relative per
// operand, always the same sign), which showed up as a 3x larger distance of the front-end GEMMs to a float64 evaluation than
// torch's own f32 GEMM has (benchmarks/accuracy_stages.py); rounded to nearest the pair carries 2^-24 and no bias.
typedef __fp16 bigc_fp16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bigc_fp16x2 bigc_pk_rtn(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const h2 v = __builtin_convertvector(f2{a, b}, h2);
    return *(const bigc_fp16x2*)&v;
}

int bigc_fail(int code, const char* fmt, ...);
#define BIGC_CHECK_ARG(cond, ...) do { if (!(cond)) return bigc_fail(BIGC_ERR_ARG, __VA_ARGS__); } while (0)
#define BIGC_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "%s launch: %s", name, hipGetErrorString(e_)); } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// one weight-gradient product dW (N, K) += dY^T (M, N) X (M, K), db += column sums of dY (backward.hip: bigc_linear_dw_group)
struct BigcDwJob { const float* dY; int ldy; const float* X; int ldx; float* dW; int ldw; float* db; int M, N, K; };
int bigc_linear_dw_group(void* stream, const BigcDwJob* jobs, int n);

// ---- dropout (training paths) ---------------------------------------------------------------------------------------------
// Counter-based masks: element `idx` of dropout site `site` is kept iff hash(seed, site, idx) >= p * 2^32; kept values are
// scaled by 1 / (1 - p).  The forward and the backward regenerate the same mask, nothing is stored.
// (torch's Philox stream cannot be reproduced outside torch; parity with the reference is defined with dropout off,
// tests/test_hip_dropout.py checks this generator against an identical numpy implementation.)
struct DropSpec { uint32_t thresh; float scale; uint32_t seed; uint32_t site; };

static inline DropSpec make_drop(float p, uint32_t seed, uint32_t site) {
    DropSpec d;
    if (!(p > 0.f)) { d.thresh = 0; d.scale = 1.f; d.seed = 0; d.site = 0; return d; }
    double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    d.scale = 1.0f / (1.0f - p);
    d.seed = seed; d.site = site;
    return d;
}

__device__ __forceinline__ bool drop_keep(const DropSpec& d, unsigned long long idx) {
    uint32_t x = (uint32_t)idx * 0x9E3779B1u ^ (uint32_t)(idx >> 32) * 0x85EBCA77u ^ d.seed * 0xC2B2AE3Du ^ d.site * 0x27D4EB2Fu;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x >= d.thresh;
}
// index of an attention probability: (global query row, head, key index inside the sequence)
__device__ __forceinline__ unsigned long long drop_att_idx(int q_row, int n_head, int h, int key) {
    return (((unsigned long long)q_row * (unsigned)n_head + (unsigned)h) << 12) + (unsigned)key;
}
