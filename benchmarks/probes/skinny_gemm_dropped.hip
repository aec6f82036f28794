// MEASURED AND DROPPED (round 2): kept as a record, not built into the library.
// A streaming kernel for the grounding scorer's N = K = 128 GEMMs.  It was correct (10 op tests against float64 incl. ragged
// depth-wise convolutions and the exact-path fallback, and the grounding goldens), but not faster than the tiled kernel it was meant
// to replace: per launch on the 240 k-row activations, plain 50 vs 52 us, k = 7 with residual 183 vs 221 us, k = 3 without residual
// 131 vs 105 us; grounding leg 5.05 vs 4.82 ms.  One wave per SIMD (139 KB of LDS per 4-wave workgroup) serialises everything, and
// a block of 32 rows costs ~3000 VALU instructions (address arithmetic, clamps, selects, the split, the epilogue) -- 23 per element
// slot, 40 us of VALU issue per launch chip-wide -- against 3 us of MFMA: a version that wins needs packed arithmetic, scalar base
// addresses and two waves per SIMD (16-row MFMA blocks so that eight wave corners fit beside the weights).
//
// Streaming GEMM for the grounding scorer's d = 128 layers: C = act(A' W^T + bias) + residual with N = K = 128, where A' is A
// itself or a depth-wise convolution of A along the rows of each sequence (kernel 3 or 7, zero padded) -- every
// DepthWiseSeparableConv1d, out_proj and fc of the combined QANet encoder and of the three heads (models/grd_model_v5.py:36-56,
// 110-137, 182-193) on the (sum of n_v T_v) x 128 activations, ~240 k rows for 8 videos.
//
// Why a kernel of its own.  At 32 flop per byte these GEMMs are bound by memory, not by the matrix cores (11 us of MFMA work
// against 35-60 us of traffic per launch), and the tiled kernel (linear_split.hip: 128 x 128 tiles, 4 K slabs, 228-256 + 128-167
// registers in its depth-wise mode = ONE 4-wave workgroup per CU) spends its time in pipeline fill and drain: 85 us per launch,
// a third of the rate at which a LayerNorm streams the same tensors.  Here nothing is tiled over K or N:
//   * the whole weight matrix (128 x 128) is split once per workgroup into f16 head / residual planes in LDS (70 KB) and stays;
//   * every wave streams blocks of 32 rows on its own -- no workgroup barrier after the weights are in place.  A lane owns two
//     CHANNELS of all 32 rows (one coalesced 512-byte row per load instruction, 64 registers), so the depth-wise convolution
//     along the rows is register arithmetic with the lane's own 2 x ks tap weights; the next block's rows are requested before
//     the MFMAs of the current one (16 KB in flight per wave);
//   * the convolved block goes, split into f16 pairs, through a wave-private LDS corner (17 KB) into MFMA fragment order
//     (lane = row, 8 consecutive k), and 96 v_mfma_f32_32x32x16_f16 produce the 32 x 128 result: a lane of the result owns one
//     feature and 16 rows, so every store instruction writes two full 128-byte lines.
// A block whose operand leaves the range of the split (|x| >= 4096, NaN, Inf; or such a weight) is recomputed by the same wave
// on the exact f32 MFMA from an f32 image of the block in the same LDS corner.
// Split format as in gemm_ps.hip: X = x 2^E, h = f16(X), r = f16(X - h) unscaled, one accumulator, 3 MFMAs per product.
// Blocks overlap by ks - 1 rows (a block of 32 loaded rows yields 32 - (ks - 1) output rows): the halo is re-read from L2.
// Roofline: HBM (bytes of A, residual and C once each).
#include "linear_internal.h"
#include "prof.h"
#include <stdlib.h>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int SK_N = 128, SK_K = 128;
constexpr int SK_WP = 272;                        // bytes of a weight row in LDS: 128 f16 + 16 (fragment reads of 32 rows hit 16 x 4 banks)
constexpr int SK_PLANE = SK_N * SK_WP;            // one weight plane (heads or residuals)
constexpr int SK_AP = 272;                        // bytes of an activation row in a wave's corner (f16 planes)
constexpr int SK_APF = 528;                       // ... and of its f32 image (exact path)
constexpr int SK_CORNER = 2 * 32 * SK_AP;         // 17408 bytes >= 32 * SK_APF = 16896
constexpr int SK_WAVES = 4;
constexpr float SK_ACT_SCALE = (float)(1 << PS_ACT_EXP);
constexpr float SK_LIMIT = 32768.0f;

struct SkArgs {
    const float* A; long lda; const float* W; int ldw; const float* bias; const float* res; long ldr; float* C; long ldc;
    int M, act, n_blocks;
    const float* dw_w; const float* dw_b; int dw_T; const int2* dw_info;
};

__device__ __forceinline__ void sk_split2(float x0, float x1, unsigned& h, unsigned& r) {
    const f32x2 v = {x0, x1};
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const f32x2 d = v - __builtin_convertvector(hh, f32x2);
    const f16x2 rr = __builtin_convertvector(d, f16x2);
    h = *(const unsigned*)&hh;
    r = *(const unsigned*)&rr;
}

template <int DWK>
__global__ __launch_bounds__(64 * SK_WAVES, 1) void skinny_gemm_kernel(const SkArgs p) {
    constexpr int H = DWK / 2, S = 32 - 2 * H;           // halo rows on each side, output rows per block
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SK_PLANE + SK_WAVES * SK_CORNER + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    unsigned char* const wh = smem;
    unsigned char* const wr = smem + SK_PLANE;
    unsigned char* const corner = smem + 2 * SK_PLANE + wave * SK_CORNER;
    float* const red = (float*)(smem + 2 * SK_PLANE + SK_WAVES * SK_CORNER);

    // ---- the weights: largest magnitude -> exponent, then the split planes
    constexpr int WQ = SK_N * SK_K / 4 / (64 * SK_WAVES);          // quads per thread
    float amax = 0.f;
    bool wnan = false;
#pragma unroll 4
    for (int i = 0; i < WQ; ++i) {
        const int idx = tid + 64 * SK_WAVES * i, n = idx >> 5, q = idx & 31;
        const f32x4 w4 = *(const f32x4*)(p.W + (size_t)n * p.ldw + 4 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c) { amax = fmaxf(amax, fabsf(w4[c])); wnan = wnan || !(fabsf(w4[c]) < INFINITY); }
    }
    if (wnan) amax = INFINITY;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = red[0];
#pragma unroll
    for (int i = 1; i < SK_WAVES; ++i) amax = fmaxf(amax, red[i]);
    const bool wbad = !(amax < INFINITY);
    int e = 0;
    if (amax > 0.f && !wbad) { int ex; frexpf(amax, &ex); e = 14 - ex; }
    e = max(-60, min(60, e));
    const float wsc = ldexpf(1.0f, e), osc = ldexpf(1.0f, -e) / SK_ACT_SCALE;
#pragma unroll 4
    for (int i = 0; i < WQ; ++i) {
        const int idx = tid + 64 * SK_WAVES * i, n = idx >> 5, q = idx & 31;
        const f32x4 w4 = *(const f32x4*)(p.W + (size_t)n * p.ldw + 4 * q) * wsc;
        unsigned h0, h1, r0, r1;
        sk_split2(w4[0], w4[1], h0, r0);
        sk_split2(w4[2], w4[3], h1, r1);
        *(u32x2*)(wh + n * SK_WP + q * 8) = u32x2{h0, h1};
        *(u32x2*)(wr + n * SK_WP + q * 8) = u32x2{r0, r1};
    }
    __syncthreads();

    // ---- per-lane constants: the two channels 2 lane, 2 lane + 1
    float dwt[DWK > 0 ? DWK : 1][2], dwb[2] = {0.f, 0.f};
    if constexpr (DWK > 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int j = 0; j < DWK; ++j) dwt[j][c] = p.dw_w[(size_t)(2 * lane + c) * DWK + j];
            dwb[c] = p.dw_b[2 * lane + c];
        }
    }
    float bv[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bv[ni] = p.bias ? p.bias[32 * ni + l31] : 0.f;
    const bool relu = p.act == BIGC_ACT_RELU, sigm = p.act == BIGC_ACT_SIGMOID;

    // ---- the stream of row blocks
    const int stride = gridDim.x * SK_WAVES;
    int b = blockIdx.x * SK_WAVES + wave;
    f32x2 raw[32];
    int2 inf = {0, 1};
    auto load_block = [&](int blk) {
        const long r0 = (long)blk * S - H;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const long row = min(max(r0 + r, 0L), (long)p.M - 1);
            raw[r] = *(const f32x2*)(p.A + row * p.lda + 2 * lane);
        }
        if (DWK > 0) {
            const long row = min(max(r0 + l31, 0L), (long)p.M - 1);
            if (p.dw_info) inf = p.dw_info[row];
            else { inf.x = (int)(row % p.dw_T); inf.y = p.dw_T; }
        }
    };
    if (b < p.n_blocks) load_block(b);
    for (; b < p.n_blocks; b += stride) {
        const long r0 = (long)b * S - H;
        // -- depth-wise convolution in place (output row r of the block = rows r - H .. r + H of the block), tap validity from the
        // OUTPUT row's position in its sequence: bit r of vmask[j] = tap j of block row r lies inside the sequence
        float cv[32][2];
        if constexpr (DWK > 0) {
            unsigned long long vmask[DWK];
#pragma unroll
            for (int j = 0; j < DWK; ++j) {
                const int tt = inf.x + j - H;
                vmask[j] = __builtin_amdgcn_ballot_w64(hi == 0 && tt >= 0 && tt < inf.y);
            }
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                if (r < H || r >= 32 - H) { cv[r][0] = 0.f; cv[r][1] = 0.f; continue; }
                float a0 = dwb[0], a1 = dwb[1];
#pragma unroll
                for (int j = 0; j < DWK; ++j) {
                    const bool ok = (vmask[j] >> r) & 1ull;                       // wave-uniform
                    a0 = ok ? a0 + dwt[j][0] * raw[r + j - H][0] : a0;
                    a1 = ok ? a1 + dwt[j][1] * raw[r + j - H][1] : a1;
                }
                cv[r][0] = a0; cv[r][1] = a1;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 32; ++r) { cv[r][0] = raw[r][0]; cv[r][1] = raw[r][1]; }
        }
        // -- range of the operand; the split image (or, out of range, the f32 image) into the wave's corner
        float bmax = 0.f, nanacc = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) { bmax = fmaxf(bmax, fmaxf(fabsf(cv[r][0]), fabsf(cv[r][1]))); nanacc += (cv[r][0] + cv[r][1]) * 0.f; }
        const bool bad = __builtin_amdgcn_ballot_w64(!(bmax * SK_ACT_SCALE < SK_LIMIT) || !(nanacc == 0.f)) != 0 || wbad;
        if (!bad) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                unsigned h, rr;
                sk_split2(cv[r][0] * SK_ACT_SCALE, cv[r][1] * SK_ACT_SCALE, h, rr);
                *(unsigned*)(corner + r * SK_AP + 4 * lane) = h;
                *(unsigned*)(corner + 32 * SK_AP + r * SK_AP + 4 * lane) = rr;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 32; ++r) *(f32x2*)(corner + r * SK_APF + 8 * lane) = f32x2{cv[r][0], cv[r][1]};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // -- the next block's rows, in flight during the MFMAs and the stores of this one
        if (b + stride < p.n_blocks) load_block(b + stride);

        // -- the residual rows of this block, requested now: they land while the MFMAs run (one wave per SIMD: nothing else would hide them)
        float rv[16][4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long rc = min(max(r0 + mfma_row(r, hi), 0L), (long)p.M - 1);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) rv[r][ni] = p.res ? p.res[rc * p.ldr + 32 * ni + l31] : 0.f;
        }
        f32x16 acc[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
        float scale = osc;
        if (!bad) {
            // fragments of step s + 1 are read while the twelve MFMAs of step s run; term-major order, so that consecutive MFMAs
            // never accumulate into the same block
            struct Frag { f16x8 ah, ar, bh[4], br[4]; };
            Frag F[2];
            auto rd = [&](int s_, Frag& f) {
                f.ah = *(const f16x8*)(corner + l31 * SK_AP + 32 * s_ + 16 * hi);
                f.ar = *(const f16x8*)(corner + 32 * SK_AP + l31 * SK_AP + 32 * s_ + 16 * hi);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    f.bh[ni] = *(const f16x8*)(wh + (32 * ni + l31) * SK_WP + 32 * s_ + 16 * hi);
                    f.br[ni] = *(const f16x8*)(wr + (32 * ni + l31) * SK_WP + 32 * s_ + 16 * hi);
                }
            };
            rd(0, F[0]);
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_) {
                const Frag& f = F[s_ & 1];
                if (s_ < 7) rd(s_ + 1, F[(s_ + 1) & 1]);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah, f.bh[ni], acc[ni], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah, f.br[ni], acc[ni], 0, 0, 0);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ar, f.bh[ni], acc[ni], 0, 0, 0);
            }
        } else {
            // exact f32 MFMA (32x32x2): lane (l31, hi) supplies A[l31][k + hi] and W[n][k + hi]
            scale = 1.0f;
            for (int k = 0; k < SK_K; k += 2) {
                const float av = *(const float*)(corner + l31 * SK_APF + 4 * (k + hi));
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ni] = mfma32(av, p.W[(size_t)(32 * ni + l31) * p.ldw + k + hi], acc[ni]);
            }
        }
        // -- epilogue: lane = feature 32 ni + l31, register r = block row mfma_row(r, hi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int br_ = mfma_row(r, hi);
            const long row = r0 + br_;
            const bool ok = br_ >= H && br_ < 32 - H && row < p.M;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                float v = acc[ni][r] * scale + bv[ni];
                if (relu) v = fmaxf(v, 0.f);
                if (sigm) v = 1.0f / (1.0f + expf(-v));
                v += rv[r][ni];
                if (ok) p.C[row * p.ldc + 32 * ni + l31] = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

bool bigc_skinny_supported(int M, int N, int K, int ks) {
    static const bool off = getenv("BIGC_SKINNY") && atoi(getenv("BIGC_SKINNY")) == 0;
    return !off && !bigc_linear_get_exact() && !bigc_linear_get_fast() && N == SK_N && K == SK_K && M >= 1024 && (ks == 0 || ks == 3 || ks == 7);
}

int bigc_skinny_gemm(void* stream, const float* A, long lda, const float* dw_w, const float* dw_b, int ks, int T, const int2* row_info,
                     const float* W, int ldw, const float* bias, const float* res, long ldr, float* C, long ldc, int M, int act) {
    BIGC_CHECK_ARG(A && W && C && M >= 0, "skinny_gemm: null pointer");
    BIGC_CHECK_ARG(ks == 0 || ks == 3 || ks == 7, "skinny_gemm: ks=%d", ks);
    BIGC_CHECK_ARG(ks == 0 || (dw_w && dw_b && (row_info || T > 0)), "skinny_gemm: depth-wise weights / sequence length");
    BIGC_CHECK_ARG(lda % 2 == 0 && ldw % 4 == 0 && ((uintptr_t)A & 7) == 0 && ((uintptr_t)W & 15) == 0, "skinny_gemm: alignment");
    if (M == 0) return BIGC_OK;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "skinny_gemm: no device");
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
    }
    SkArgs p;
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.res = res; p.ldr = ldr; p.C = C; p.ldc = ldc; p.M = M; p.act = act;
    p.dw_w = dw_w; p.dw_b = dw_b; p.dw_T = T > 0 ? T : 1; p.dw_info = row_info;
    const int S = 32 - (ks > 0 ? ks - 1 : 0);
    p.n_blocks = cdiv(M, S);
    const int grid = p.n_blocks >= n_cu * SK_WAVES ? n_cu : cdiv(p.n_blocks, SK_WAVES);
    ProfScope ps(stream, FAM_LINEAR, 2.0 * M * (double)SK_N * SK_K + 2.0 * M * (double)SK_K * ks,
                 4.0 * ((double)M * SK_K + (double)SK_N * SK_K + (double)M * SK_N * (res ? 2 : 1)));
    hipStream_t hs = (hipStream_t)stream;
    if (ks == 0) hipLaunchKernelGGL(skinny_gemm_kernel<0>, dim3(grid), dim3(64 * SK_WAVES), 0, hs, p);
    else if (ks == 3) hipLaunchKernelGGL(skinny_gemm_kernel<3>, dim3(grid), dim3(64 * SK_WAVES), 0, hs, p);
    else hipLaunchKernelGGL(skinny_gemm_kernel<7>, dim3(grid), dim3(64 * SK_WAVES), 0, hs, p);
    BIGC_LAUNCH_CHECK("skinny_gemm");
    return BIGC_OK;
}

// Raw entry point (tests, benchmarks): C (M, 128) = act(dwconv_ks(A) W^T + bias) + residual, A (M, 128), W (128, 128); ks = 0: no
// convolution.  Sequences are T rows each, or given per row by row_info = (position, length) pairs.
extern "C" int bigc_skinny_gemm_f32(void* stream, const float* A, long lda, const float* dw_w, const float* dw_b, int ks, int T, const int32_t* row_info,
                                    const float* W, int ldw, const float* bias, const float* residual, long ldr, float* C, long ldc, int M, int act) {
    return bigc_skinny_gemm(stream, A, lda, dw_w, dw_b, ks, T, (const int2*)row_info, W, ldw, bias, residual, ldr, C, ldc, M, act);
}
