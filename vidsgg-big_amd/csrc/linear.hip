// The "linear" entry point: C = act(A' W^T + bias) + residual, with an optional segmented row gather on A (plain row
// gather, or the 3-tap implicit GEMM of the stride-2 temporal conv).  Replaces torch.nn.Linear / nn.Conv1d(k3,s2) at every
// call site of the BIG hot path (reference: models/model_0v10.py:290-309,446-458; models/grd_model_v5.py:171-174).
// By default the work goes to the split-operand f16 MFMA kernel (linear_split.hip); this file holds the dispatch and the
// EXACT f32-MFMA kernel (bigc_linear_set_exact(1) / BIGC_LINEAR_EXACT=1):
//
// Tiling (gfx950): 128x128 output tile per 256-thread workgroup, 4 waves as 2x2, each wave
// a 64x64 patch = 2x2 v_mfma_f32_32x32x2_f32 accumulators (64 acc VGPRs). K is walked in
// slabs of 32: both operands are K-contiguous in HBM (activations row-major, torch weights
// (N,K) row-major), so a slab row is one 128-byte line, loaded as 8 x dwordx4 by 8
// consecutive lanes, staged through a double-buffered LDS image with a 36-float row pitch
// (ds_read_b128 conflict-free: 16 lanes with distinct row mod 16 cover all 64 banks).
// A lane reads 4 consecutive k for its row with one ds_read_b128; lanes 0-31 take k..k+3 and
// lanes 32-63 take k+4..k+7 of each group of 8, which only permutes the order of the K
// summation (the MFMA pairs k with k+4).
// Roofline: f32 MFMA 157.3 TFLOP/s; 2*M*N*K flops per launch.
#include "linear_internal.h"
#include "prof.h"
#include <stdlib.h>
#include <atomic>
#include <string.h>

namespace {

constexpr int BM = 128;
constexpr int SEG_ALIGN = 32;        // gather segments are multiples of this many columns

// NT = number of 32-column MFMA tiles per wave: NT=2 -> 128x128 block tile, 2 workgroups per CU;
// NT=4 -> 128x256 block tile (wave patch 64x128, 128 accumulator VGPRs), one workgroup per CU, i.e. one
// wave per SIMD issuing 128 MFMAs between barriers (used when N >= 256: halves the barriers per flop
// and the A re-reads).
//
// BK = K-slab depth: 32 (2 workgroups per CU, 73.7 KB of LDS each) or 16 (pitch 20 floats, 40 KB -> 4 workgroups per CU:
// twice as many independent waves per SIMD to cover each other's barrier waits).
template <bool GATHER, int NT, int BK, int OCC>
__global__ __launch_bounds__(256, OCC) void linear_f32_kernel(const LinearArgs p) {
    constexpr int PITCH = BK + 4;                    // 36 / 20 floats: odd number of 16-byte granules -> conflict-free b128
    constexpr int BNT = 64 * NT;                     // block tile width
    constexpr int TPR = BK / 4;                      // threads per slab row (one float4 each)
    constexpr int RP = 256 / TPR;                    // rows loaded per pass
    constexpr int NA = BM / RP;                      // float4 loads of A per thread per slab
    constexpr int NB = BNT / RP;                     // float4 loads of W per thread per slab
    __shared__ __attribute__((aligned(16))) float smem[2][(BM + BNT) * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, l31 = lane & 31;

    // XCD-aware tile order: workgroup b lands on XCD b%8 (observed dispatch), so give each XCD
    // whole M-tiles and walk their N-tiles back to back: the A slab is then re-read from that
    // XCD's L2 instead of HBM, and W (shared by everyone) stays L2/MALL resident.
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int tm = (j / p.tiles_n) * 8 + xcd, tn = j % p.tiles_n;
    if (tm >= p.tiles_m) return;
    const int m0 = tm * BM, n0 = tn * BNT;

    const int lr = tid / TPR, lc = (tid % TPR) * 4;
    int arow[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int m = m0 + lr + RP * i;
        arow[i] = (m < p.M) ? m : -1;
    }
    f32x4 ra[NA], rb[NB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int cur_seg = -1;

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK, kc = k0 + lc;
        const bool kval = kc < p.K;
        int acol = kc;
        if (GATHER) {
            const int seg = k0 / p.seg_len;
            acol = kc - seg * p.seg_len;
            if (seg != cur_seg) {
                cur_seg = seg;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    int m = m0 + lr + RP * i;
                    arow[i] = p.a_idx[(size_t)min(m, p.M - 1) * p.nseg + seg];
                    if (m >= p.M) arow[i] = -1;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
            ra[i] = (arow[i] >= 0 && kval) ? *(const f32x4*)(p.A + (size_t)arow[i] * p.lda + acol) : zero4;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int n = n0 + lr + RP * i;
            rb[i] = (n < p.N && kval) ? *(const f32x4*)(p.W + (size_t)n * p.ldw + kc) : zero4;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *(f32x4*)(&smem[buf][(lr + RP * i) * PITCH + lc]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *(f32x4*)(&smem[buf][(BM + lr + RP * i) * PITCH + lc]) = rb[i];
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) load_tiles(kt + 1);
        const float* as = &smem[buf][(wm * 64 + l31) * PITCH + 4 * hi];
        const float* bs = &smem[buf][(BM + wn * 32 * NT + l31) * PITCH + 4 * hi];
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f32x4 af[2], bf[NT];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi] = *(const f32x4*)(as + mi * 32 * PITCH + 8 * c);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) bf[ni] = *(const f32x4*)(bs + ni * 32 * PITCH + 8 * c);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma32(af[mi][q], bf[ni][q], acc[mi][ni]);
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lanes 0-31 of a register hold 32 consecutive columns of one row (128-byte stores)
    // The residual reads use clamped (always valid) addresses so that all loads of a 32-row slice
    // are issued back to back before the first use (guarded loads serialise: load, wait, add, store).
    int colv[NT];
    float bv[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        colv[ni] = n0 + wn * 32 * NT + ni * 32 + l31;
        bv[ni] = p.bias ? p.bias[min(colv[ni], p.N - 1)] : 0.f;
    }
    const bool relu = p.act == BIGC_ACT_RELU, sigm = p.act == BIGC_ACT_SIGMOID;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int rbase = m0 + wm * 64 + mi * 32;
#pragma unroll
        for (int nh = 0; nh < NT; nh += 2) {
            f32x16 rv[2];
            if (p.res) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const float* rp = p.res + min(colv[nh + ni], p.N - 1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[ni][r] = rp[(size_t)min(rbase + mfma_row(r, hi), p.M - 1) * p.ldr];
                }
            } else {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[ni][r] = 0.f;
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if (colv[nh + ni] >= p.N) continue;
                float* cp = p.C + colv[nh + ni];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + mfma_row(r, hi);
                    float v = acc[mi][nh + ni][r] + bv[nh + ni];
                    if (relu) v = fmaxf(v, 0.f);
                    if (sigm) v = 1.0f / (1.0f + expf(-v));
                    v += rv[ni][r];
                    if (row < p.M) cp[(size_t)row * p.ldc] = v;
                }
            }
        }
    }
}

// K = 8 (fc_bbox2enti.0, models/model_0v10.py:290-294: the 8 box coordinates -> dim_enti): 16 flops per output element,
// so the kernel is bound by writing C (4 N bytes per row against 32 bytes read).  Plain f32 FMAs: a thread keeps the
// weights of four adjacent output columns in registers and walks down the rows of its block; a wave writes 1 KB
// contiguous per row.  RL = 256 / (N / 4) rows are in flight per pass.
constexpr int SK_ROWS = 128;     // rows per workgroup
__global__ __launch_bounds__(256) void linear_k8_kernel(const LinearArgs p) {
    const int cg = p.N >> 2, c4 = threadIdx.x % cg, rl = threadIdx.x / cg, nrl = 256 / cg;
    f32x4 w0[4], w1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float* wp = p.W + (size_t)(4 * c4 + j) * p.ldw;
        w0[j] = *(const f32x4*)wp;
        w1[j] = *(const f32x4*)(wp + 4);
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *(const f32x4*)(p.bias + 4 * c4);
    const bool relu = p.act == BIGC_ACT_RELU, sigm = p.act == BIGC_ACT_SIGMOID;
    const int r0 = blockIdx.x * SK_ROWS, r1 = min(r0 + SK_ROWS, p.M);
    bool oob = false;
    // (no unroll request: the hr8 epilogue's lane exchange is a convergent operation, the compiler keeps this loop rolled)
    for (int r = r0 + rl; r < r1; r += nrl) {
        const float* ap = p.A + (size_t)r * p.lda;
        const f32x4 a0 = *(const f32x4*)ap, a1 = *(const f32x4*)(ap + 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = a0[0] * w0[j][0];
            v = fmaf(a0[1], w0[j][1], v); v = fmaf(a0[2], w0[j][2], v); v = fmaf(a0[3], w0[j][3], v);
            v = fmaf(a1[0], w1[j][0], v); v = fmaf(a1[1], w1[j][1], v); v = fmaf(a1[2], w1[j][2], v); v = fmaf(a1[3], w1[j][3], v);
            v += bv[j];
            if (relu) v = fmaxf(v, 0.f);
            if (sigm) v = 1.0f / (1.0f + expf(-v));
            o[j] = v;
        }
        if (p.res) o += *(const f32x4*)(p.res + (size_t)r * p.ldr + 4 * c4);
        if (p.C) *(f32x4*)(p.C + (size_t)r * p.ldc + 4 * c4) = o;
        if (p.Cs) {
            // split form "hr8" (gemm_ps.hip): a group is 8 columns = this lane's four and its neighbour's (c4 ^ 1, the adjacent lane)
            const f32x4 x = o * (float)(1 << PS_ACT_EXP);
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = __shfl_xor(x[j], 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) oob = oob || !(fabsf(x[j]) < 32768.0f);
            // the even lane of a pair stores the group's 16 bytes of heads, the odd lane its 16 bytes of residuals: every lane one
            // 16-byte store, contiguous across the wave (before: the even lanes stored both halves, 16 bytes at a stride of 32)
            {
                typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                typedef float f2_t __attribute__((ext_vector_type(2)));
                const bool odd = (c4 & 1) != 0;
                const f32x4 lo = odd ? y : x, hi4 = odd ? x : y;          // the group's eight columns in order
                uint4 out;
                unsigned* op = (unsigned*)&out;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f2_t v2 = q < 2 ? f2_t{lo[2 * q], lo[2 * q + 1]} : f2_t{hi4[2 * q - 4], hi4[2 * q - 3]};
                    const h2_t h2 = __builtin_convertvector(v2, h2_t);
                    const h2_t r2 = __builtin_convertvector(v2 - __builtin_convertvector(h2, f2_t), h2_t);
                    op[q] = odd ? *(const unsigned*)&r2 : *(const unsigned*)&h2;
                }
                *(uint4*)(p.Cs + (size_t)r * p.ldcs + (size_t)c4 * 16) = out;
            }
        }
    }
    if (p.Cs && oob && p.range_flag) atomicOr(p.range_flag, 1);
}

}  // namespace

// Arithmetic mode.  The PROCESS default (0: split-operand f16 MFMA kernels, 1: exact f32 MFMA kernels; initialised from
// BIGC_LINEAR_EXACT) applies to the raw kernel entry points and to every model that has no mode of its own.  A model handle can
// carry its own (bigc_set_mode / grd_set_mode): its entry points open a ModeScope, a thread-local override for the duration of the
// call, so two models in one process -- or one model per thread and GPU, as the reference's DataParallel runs them
// (utils/DataParallel.py:68) -- do not share the switch.
static std::atomic<int> g_exact_mode{(getenv("BIGC_LINEAR_EXACT") && atoi(getenv("BIGC_LINEAR_EXACT")) != 0) ? 1 : 0};
thread_local int bigc_mode_exact_override = -1, bigc_mode_fast_override = -1;
extern "C" int bigc_linear_set_exact(int exact) { return g_exact_mode.exchange(exact ? 1 : 0); }
extern "C" int bigc_linear_get_exact(void) { return bigc_mode_exact_override >= 0 ? bigc_mode_exact_override : g_exact_mode.load(); }
// Opt-in reduced-precision modes of the split-operand GEMM kernels (linear_split.hip, gemm_ps.hip), the "bf16 fast mode" of SURVEY
// section 7 step 10 / BASELINE configs[1], gated on Recall@50 parity instead of the 1e-4 logit tolerance:
//   1 (BIGC_FAST=1 / f16)   every operand rounded to one f16 (11-bit mantissa), ONE v_mfma_f32_32x32x16_f16 per product, f32 accumulation
//   2 (BIGC_FAST=2 / bf16)  the literal bf16 arithmetic: operands rounded to one bf16 (8-bit mantissa), v_mfma_f32_32x32x16_bf16; the model
//                           forwards then do not chain GEMMs through the split form (it carries f16 heads), every GEMM runs on linear_split
// Never the default; initialised from BIGC_FAST.
static int fast_from_env() {
    const char* e = getenv("BIGC_FAST");
    if (!e) return 0;
    if (!strcmp(e, "bf16")) return 2;
    if (!strcmp(e, "f16")) return 1;
    const int v = atoi(e);
    return v == 2 ? 2 : (v != 0 ? 1 : 0);
}
static std::atomic<int> g_fast_mode{fast_from_env()};
extern "C" int bigc_linear_set_fast(int fast) { return g_fast_mode.exchange(fast == 2 ? 2 : (fast ? 1 : 0)); }
extern "C" int bigc_linear_get_fast(void) {
    if (bigc_linear_get_exact()) return 0;
    return bigc_mode_fast_override >= 0 ? bigc_mode_fast_override : g_fast_mode.load();
}

// Accumulation policy of the model forwards' GEMMs on pre-split operands (csrc/model.hip ps_accum, csrc/gemm_ps.hip PsGemm.accum):
//   0 plain    one f32 accumulator per element, tile sizes chosen for speed (rounds 2 - 3 x as coarsely as a blocked CPU GEMM at K >= 1024)
//   1 k1024    blocked accumulation for the K >= 1024 GEMMs only        2 front      ... for the GEMMs with M >= 100 000 rows only
//   3 nodecoder ... for every GEMM but the decoder's                    4 blocked    ... for every GEMM (DEFAULT: every benchmark video within
//   1e-4 of a float64 evaluation).  Initial value from BIGC_ACCUM (the names above).
static int accum_from_env() {
    const char* e = getenv("BIGC_ACCUM");
    if (!e) return 4;
    return !strcmp(e, "plain") ? 0 : !strcmp(e, "k1024") ? 1 : !strcmp(e, "front") ? 2 : !strcmp(e, "nodecoder") ? 3 : 4;
}
static std::atomic<int> g_accum_policy{accum_from_env()};
extern "C" int bigc_set_accum_policy(int policy) { return g_accum_policy.exchange(policy < 0 ? 0 : (policy > 4 ? 4 : policy)); }
extern "C" int bigc_get_accum_policy(void) { return g_accum_policy.load(); }

int bigc_linear_hr8out(void* stream, const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldr,
                       float* C, int ldc, void* Cs, long ldcs, int M, int N, int K, int act, int* range_flag) {
    BIGC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && A && W && Cs, "linear_hr8out: bad argument");
    BIGC_CHECK_ARG(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && N % 8 == 0 && ldcs % 4 == 0 && ((uintptr_t)Cs & 15) == 0 && ((uintptr_t)A & 15) == 0 &&
                   ((uintptr_t)W & 15) == 0, "linear_hr8out: alignment (N=%d must be a multiple of 8)", N);
    BIGC_CHECK_ARG(bigc_linear_get_exact() == 0, "linear_hr8out: not available in exact mode");
    if (M == 0) return BIGC_OK;
    LinearArgs p;
    p.A = A; p.lda = lda; p.a_idx = nullptr; p.nseg = 1; p.seg_len = K;
    p.W = W; p.ldw = ldw; p.bias = bias; p.res = res; p.ldr = ldr; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.act = act;
    p.tiles_m = cdiv(M, BM); p.tiles_n = 0;
    p.Ws = nullptr; p.ldws = 0; p.wbad = nullptr; p.zero_row = nullptr;
    p.Cs = (unsigned char*)Cs; p.ldcs = ldcs * 4; p.range_flag = range_flag;
    const double flops = 2.0 * M * (double)N * K, bytes = 4.0 * ((double)M * K + (double)N * K + (double)M * N * ((res ? 1 : 0) + (C ? 2 : 1)));
    return bigc_linear_split_launch((hipStream_t)stream, p, flops, bytes);
}

int bigc_linear_hr8out_rows(void* stream, const uint64_t* row_ptr, const float* W, int ldw, const float* bias, void* Cs, long ldcs, int M, int N, int K,
                            int act, int* range_flag) {
    BIGC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && row_ptr && W && Cs, "linear_hr8out_rows: bad argument");
    BIGC_CHECK_ARG(K % 4 == 0 && ldw % 4 == 0 && N % 8 == 0 && ldcs % 4 == 0 && ((uintptr_t)Cs & 15) == 0 && ((uintptr_t)W & 15) == 0,
                   "linear_hr8out_rows: alignment (N=%d must be a multiple of 8)", N);
    BIGC_CHECK_ARG(bigc_linear_get_exact() == 0, "linear_hr8out_rows: not available in exact mode");
    if (M == 0) return BIGC_OK;
    LinearArgs p;
    p.A = nullptr; p.lda = 0; p.a_idx = nullptr; p.nseg = 1; p.seg_len = K; p.a_rowptr = row_ptr;
    p.W = W; p.ldw = ldw; p.bias = bias; p.res = nullptr; p.ldr = 0; p.C = nullptr; p.ldc = 0;
    p.M = M; p.N = N; p.K = K; p.act = act;
    p.tiles_m = cdiv(M, BM); p.tiles_n = 0;
    p.Ws = nullptr; p.ldws = 0; p.wbad = nullptr; p.zero_row = nullptr;
    p.Cs = (unsigned char*)Cs; p.ldcs = ldcs * 4; p.range_flag = range_flag;
    const double flops = 2.0 * M * (double)N * K, bytes = 4.0 * ((double)M * K + (double)N * K + (double)M * N);
    return bigc_linear_split_launch((hipStream_t)stream, p, flops, bytes);
}

// C (f32) = act(A W^T + bias) with the rows of A given by address: the first visual-feature layer of the TRAINING forward read in place (train.hip)
int bigc_linear_rows_f32(void* stream, const uint64_t* row_ptr, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K, int act) {
    BIGC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && row_ptr && W && C, "linear_rows: bad argument");
    BIGC_CHECK_ARG(K % 4 == 0 && ldw % 4 == 0 && ldc % 4 == 0 && ldc >= N && ((uintptr_t)C & 15) == 0 && ((uintptr_t)W & 15) == 0, "linear_rows: alignment");
    BIGC_CHECK_ARG(bigc_linear_get_exact() == 0, "linear_rows: not available in exact mode");
    if (M == 0) return BIGC_OK;
    LinearArgs p;
    p.A = nullptr; p.lda = 0; p.a_idx = nullptr; p.nseg = 1; p.seg_len = K; p.a_rowptr = row_ptr;
    p.W = W; p.ldw = ldw; p.bias = bias; p.res = nullptr; p.ldr = 0; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.act = act;
    p.tiles_m = cdiv(M, BM); p.tiles_n = 0;
    p.Ws = nullptr; p.ldws = 0; p.wbad = nullptr; p.zero_row = nullptr;
    const double flops = 2.0 * M * (double)N * K, bytes = 4.0 * ((double)M * K + (double)N * K + (double)M * N);
    return bigc_linear_split_launch((hipStream_t)stream, p, flops, bytes);
}

int bigc_linear_k8_hr8out(void* stream, const float* A, const float* W, const float* bias, void* Cs, long ldcs, int M, int N, int act, int* range_flag) {
    BIGC_CHECK_ARG(A && W && Cs && M >= 0 && N % 8 == 0 && N <= 1024 && 256 % (N / 4) == 0 && ((uintptr_t)Cs & 15) == 0 && ldcs % 4 == 0 &&
                   (!bias || ((uintptr_t)bias & 15) == 0), "linear_k8_hr8out: bad argument");
    if (M == 0) return BIGC_OK;
    LinearArgs p;
    p.A = A; p.lda = 8; p.a_idx = nullptr; p.nseg = 1; p.seg_len = 8;
    p.W = W; p.ldw = 8; p.bias = bias; p.res = nullptr; p.ldr = 0; p.C = nullptr; p.ldc = 0;
    p.M = M; p.N = N; p.K = 8; p.act = act; p.tiles_m = 0; p.tiles_n = 0;
    p.Ws = nullptr; p.ldws = 0; p.wbad = nullptr; p.zero_row = nullptr;
    p.Cs = (unsigned char*)Cs; p.ldcs = ldcs * 4; p.range_flag = range_flag;
    ProfScope ps(stream, FAM_LINEAR_AUX, 2.0 * M * (double)N * 8, 4.0 * ((double)M * 8 + (double)M * N));
    hipLaunchKernelGGL(linear_k8_kernel, dim3(cdiv(M, SK_ROWS)), dim3(256), 0, (hipStream_t)stream, p);
    BIGC_LAUNCH_CHECK("linear_k8 (hr8)");
    return BIGC_OK;
}

extern "C" int bigc_linear_hr8out_f32(void* stream, const float* A, int lda, const float* W, int ldw, const float* bias, const float* residual, int ldr,
                                      float* C, int ldc, void* Cs, long ldcs_bytes, int M, int N, int K, int act, int* range_flag) {
    BIGC_CHECK_ARG(ldcs_bytes % 16 == 0, "linear_hr8out: ldcs_bytes must be a multiple of 16");
    if (K == 8 && !C && !residual && lda == 8 && ldw == 8)
        return bigc_linear_k8_hr8out(stream, A, W, bias, Cs, ldcs_bytes / 4, M, N, act, range_flag);
    return bigc_linear_hr8out(stream, A, lda, W, ldw, bias, residual, ldr, C, ldc, Cs, ldcs_bytes / 4, M, N, K, act, range_flag);
}

extern "C" int bigc_relu_bwd_f32(void* stream, float* dY, const float* Y, long n);      // backward.hip

int bigc_linear_relu_masked(void* stream, const float* A, int lda, const float* W, int ldw, const float* relu_out, int ldm, float* C, int ldc,
                            int M, int N, int K) {
    BIGC_CHECK_ARG(A && W && relu_out && C && M >= 0 && N > 0 && K > 0, "linear_relu_masked: bad argument");
    if (M == 0) return BIGC_OK;
    // exact mode, K = 8, odd layouts: the plain product, then the mask as a pass of its own (needs dense rows)
    if (bigc_linear_get_exact() || K == 8 || K % 4 || lda % 4 || ldw % 4) {
        BIGC_CHECK_ARG(ldc == N && ldm == N, "linear_relu_masked: the two-pass form needs dense rows");
        int rc = bigc_linear_f32(stream, A, lda, nullptr, 1, K, W, ldw, nullptr, nullptr, 0, C, ldc, M, N, K, BIGC_ACT_NONE);
        if (rc != BIGC_OK) return rc;
        return bigc_relu_bwd_f32(stream, C, relu_out, (long)M * N);
    }
    BIGC_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "linear_relu_masked: A and W must be 16-byte aligned");
    LinearArgs p;
    p.A = A; p.lda = lda; p.a_idx = nullptr; p.nseg = 1; p.seg_len = K;
    p.W = W; p.ldw = ldw; p.bias = nullptr; p.res = relu_out; p.ldr = ldm; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.act = BIGC_ACT_RELU_MASK;
    p.tiles_m = cdiv(M, BM); p.tiles_n = 0;
    p.Ws = nullptr; p.ldws = 0; p.wbad = nullptr; p.zero_row = nullptr;
    return bigc_linear_split_launch((hipStream_t)stream, p, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + 2.0 * M * N));
}

bool bigc_linear_pre8_supported(int N, int K) {
    // Opt-in (BIGC_PRE8_FUSED=1).  Measured on the 256-video step: hb (1.19 GB written + read) disappears and so does the 0.29 ms
    // K = 8 kernel, but the 64 extra FMAs per staged quad sit in the MFMA-bound loop of a power-limited kernel: the fused GEMM
    // takes 1.2 ms longer (22.17 vs 21.37 ms per step).  Traffic is not what this step waits for.
    static const bool on = getenv("BIGC_PRE8_FUSED") && atoi(getenv("BIGC_PRE8_FUSED")) == 1;
    return on && bigc_linear_get_exact() == 0 && K % 32 == 0;
}

int bigc_linear_pre8(void* stream, const float* x8, const float* W0, const float* b0, const float* W, int ldw, const float* bias,
                     float* C, int ldc, int M, int N, int K, int act) {
    BIGC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && x8 && W0 && b0 && W && C, "linear_pre8: bad argument");
    BIGC_CHECK_ARG(bigc_linear_pre8_supported(N, K), "linear_pre8: unsupported shape / mode (K=%d)", K);
    BIGC_CHECK_ARG(ldw % 4 == 0 && ((uintptr_t)x8 & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)W0 & 15) == 0 && ((uintptr_t)b0 & 15) == 0,
                   "linear_pre8: alignment");
    if (M == 0) return BIGC_OK;
    LinearArgs p;
    p.A = x8; p.lda = 8; p.a_idx = nullptr; p.nseg = 1; p.seg_len = K;      // (A itself is not read: the staged operand comes from pre_*)
    p.W = W; p.ldw = ldw; p.bias = bias; p.res = nullptr; p.ldr = 0; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.act = act;
    p.tiles_m = cdiv(M, BM); p.tiles_n = 0;
    p.Ws = nullptr; p.ldws = 0; p.wbad = nullptr; p.zero_row = nullptr;
    p.pre_x = x8; p.pre_w = W0; p.pre_b = b0;
    const double flops = 2.0 * M * (double)N * K + 2.0 * M * (double)K * 8, bytes = 4.0 * ((double)M * 8 + (double)N * K + (double)M * N);
    return bigc_linear_split_launch((hipStream_t)stream, p, flops, bytes);
}

bool bigc_linear_dwsep_supported(int N, int K, int ks) {
    static const bool off = getenv("BIGC_DWSEP_FUSED") && atoi(getenv("BIGC_DWSEP_FUSED")) == 0;
    return !off && bigc_linear_get_exact() == 0 && K % 32 == 0 && (ks == 3 || ks == 7);
}

int bigc_linear_dwsep(void* stream, const float* X, int ldx, const float* dw_w, const float* dw_b, int ks, int T, const int2* row_info,
                      const float* W, int ldw, const float* bias, const float* res, int ldr, float* C, int ldc, int M, int N, int K, int act) {
    BIGC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && X && dw_w && dw_b && W && C, "linear_dwsep: bad argument");
    BIGC_CHECK_ARG(bigc_linear_dwsep_supported(N, K, ks), "linear_dwsep: unsupported shape / mode (K=%d ks=%d)", K, ks);
    BIGC_CHECK_ARG(ldx % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)dw_w & 15) == 0 &&
                   ((uintptr_t)dw_b & 15) == 0, "linear_dwsep: alignment");
    BIGC_CHECK_ARG(row_info || T > 0, "linear_dwsep: sequence length");
    if (M == 0) return BIGC_OK;
    LinearArgs p;
    p.A = X; p.lda = ldx; p.a_idx = nullptr; p.nseg = 1; p.seg_len = K;
    p.W = W; p.ldw = ldw; p.bias = bias; p.res = res; p.ldr = ldr; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.act = act;
    p.tiles_m = cdiv(M, BM); p.tiles_n = 0;
    p.Ws = nullptr; p.ldws = 0; p.wbad = nullptr; p.zero_row = nullptr;
    p.dw_w = dw_w; p.dw_b = dw_b; p.dw_ks = ks; p.dw_T = T; p.dw_info = row_info;
    const double flops = 2.0 * M * (double)N * K + 2.0 * M * (double)K * ks, bytes = 4.0 * ((double)M * K + (double)N * K + (double)M * N * (res ? 2 : 1));
    return bigc_linear_split_launch((hipStream_t)stream, p, flops, bytes);
}

extern "C" int bigc_linear_f32(void* stream, const float* A, int lda, const int32_t* a_idx, int nseg, int seg_len,
                               const float* W, int ldw, const float* bias, const float* residual, int ldr,
                               float* C, int ldc, int M, int N, int K, int act) {
    BIGC_CHECK_ARG(M >= 0 && N > 0 && K > 0, "linear: bad shape M=%d N=%d K=%d", M, N, K);
    if (M == 0) return BIGC_OK;
    BIGC_CHECK_ARG(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, "linear: K, lda, ldw must be multiples of 4 (K=%d lda=%d ldw=%d)", K, lda, ldw);
    BIGC_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "linear: A and W must be 16-byte aligned");
    BIGC_CHECK_ARG(A && W && C, "linear: null pointer");
    if (a_idx) {
        BIGC_CHECK_ARG(nseg >= 1 && seg_len > 0 && nseg * seg_len == K, "linear: K=%d != nseg*seg_len=%d*%d", K, nseg, seg_len);
        BIGC_CHECK_ARG(nseg == 1 || seg_len % SEG_ALIGN == 0, "linear: seg_len=%d must be a multiple of %d", seg_len, SEG_ALIGN);
    }
    LinearArgs p;
    p.A = A; p.lda = lda; p.a_idx = a_idx; p.nseg = a_idx ? nseg : 1; p.seg_len = a_idx ? seg_len : K;
    p.W = W; p.ldw = ldw; p.bias = bias; p.res = residual; p.ldr = ldr; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.act = act;
    p.tiles_m = cdiv(M, BM); p.tiles_n = 0;
    hipStream_t s = (hipStream_t)stream;
    const double flops = 2.0 * M * (double)N * K, bytes = 4.0 * ((double)M * K + (double)N * K + (double)M * N * (residual ? 2 : 1));
    static const int force_nt = getenv("BIGC_LINEAR_NT") ? atoi(getenv("BIGC_LINEAR_NT")) : 0;
    static const int force_bk = getenv("BIGC_LINEAR_BK") ? atoi(getenv("BIGC_LINEAR_BK")) : 0;
    const bool exact_only = bigc_linear_get_exact() != 0;
    p.Ws = nullptr; p.ldws = 0; p.wbad = nullptr; p.zero_row = nullptr;
    // K = 8 with a vector-friendly layout: the store-bound FMA kernel (both modes; exact f32)
    if (K == 8 && !a_idx && N % 4 == 0 && N <= 1024 && 256 % (N / 4) == 0 && ldc % 4 == 0 && ((uintptr_t)C & 15) == 0 &&
        (!bias || ((uintptr_t)bias & 15) == 0) && (!residual || (ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0))) {
        ProfScope ps(stream, FAM_LINEAR_AUX, flops, bytes);
        hipLaunchKernelGGL(linear_k8_kernel, dim3(cdiv(M, SK_ROWS)), dim3(256), 0, s, p);
        BIGC_LAUNCH_CHECK("linear_k8");
        return BIGC_OK;
    }
    if (!exact_only) return bigc_linear_split_launch(s, p, flops, bytes);   // f16 MFMAs on split operands, in-kernel range fallback
    ProfScope ps(stream, FAM_LINEAR, flops, bytes);
    // measured on MI355X (benchmarks/linear_shapes.py): NT=2 119 TFLOP/s aggregate, NT=4 112 -> NT=2 by default
    const int nt = (force_nt == 4) ? 4 : 2;
    p.tiles_n = cdiv(N, 64 * nt);
    const int grid = cdiv(p.tiles_m, 8) * 8 * p.tiles_n;
    if (nt == 4) {
        if (a_idx) hipLaunchKernelGGL((linear_f32_kernel<true, 4, 32, 1>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((linear_f32_kernel<false, 4, 32, 1>), dim3(grid), dim3(256), 0, s, p);
    } else if (force_bk == 32) {
        if (a_idx) hipLaunchKernelGGL((linear_f32_kernel<true, 2, 32, 2>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((linear_f32_kernel<false, 2, 32, 2>), dim3(grid), dim3(256), 0, s, p);
    } else {
        // K slabs of 16, three workgroups per CU: 122.8 vs 119.2 TFLOP/s aggregate for slabs of 32 with two
        if (a_idx) hipLaunchKernelGGL((linear_f32_kernel<true, 2, 16, 3>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((linear_f32_kernel<false, 2, 16, 3>), dim3(grid), dim3(256), 0, s, p);
    }
    BIGC_LAUNCH_CHECK("linear_f32");
    return BIGC_OK;
}
