// The three prediction heads of the grounding scorer (models/grd_model_v5.py:182-193, applied at :370-372) as ONE kernel: each head is
//     4 x [ DepthWiseSeparableConv1d(d, d, k = 3) + ReLU ]  ->  DepthWiseSeparableConv1d(d, c_out, k = 3)
// over the clip axis of every (query, video) sequence of the combined encoder's output.  Layer by layer (csrc/grounding.hip: run_head)
// that is 15 GEMM launches per batch that each read and write the whole (rows, 128) activation -- HBM-bound at 32 flop per byte, 1.4 of
// the grounding leg's 4.9 ms.  Here a workgroup keeps a tile of 64 consecutive rows in LDS through all five layers of one head: the
// activation is read from HBM once per head, the intermediate layers never leave the CU, only the (rows, c_out) result is written.
//
// A tile covers rows [g0, g0 + 64) of the row-major activation (sequences back to back); a k = 3 convolution needs its two neighbours
// inside the same sequence, so after five layers rows 5 .. 58 of the tile are exact and are the tile's output (54 rows per workgroup;
// the rows at the rim are recomputed by the neighbouring tiles).  Rows outside a sequence contribute zeros, as the reference's padding.
//
// Range: the split form holds |x| < 32768 (finite).  A tile that met an activation outside it raises a per-launch flag, a point-wise
// weight outside it the flag word behind the prepared weights; grd_head_exact_kernel, launched right behind and gated on the two flags,
// then recomputes the launch through all five layers with plain f32 FMAs from the f32 weights (same role as the in-kernel f32 recompute
// of csrc/linear_split.hip, on which the layer-by-layer path this kernel replaces relies).  NaN / Inf then propagate as in the reference.
//
// Arithmetic: the point-wise layers run on the f16 matrix cores with exactly split f32 operands, three MFMAs per product in two
// accumulators (the scheme of csrc/linear_split.hip); the depth-wise taps are f32 FMAs.  The weights are the MFMA's A operand: a wave
// owns 32 output features and one of the tile's two 32-row blocks and holds its slice of the layer's weight (pre-split fragments,
// 64 registers) during the layer's MFMAs; an output lane owns one data row.
#include "grounding_internal.h"
#include "prof.h"
#include <stdlib.h>
#include <atomic>

typedef _Float16 a16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 a16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HD_D = 128, HD_CR = 64, HD_HALO = 5, HD_TR = HD_CR - 2 * HD_HALO;
constexpr int HD_XP = 132;                  // pitch (floats) of the f32 activation tile
constexpr int HD_AP = 272;                  // pitch (bytes) of a row of a split plane: 128 f16 + 16
constexpr float HD_RS = 2048.0f, HD_RI = 1.0f / 2048.0f;

__device__ __forceinline__ void hd_split8(const float (&x)[8], a16x8& h, a16x8& r) {
    union { bigc_fp16x2 p[4]; a16x8 v; } hh, rr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        hh.p[j] = bigc_pk_rtn(a, b);
        rr.p[j] = bigc_pk_rtn((a - (float)hh.p[j][0]) * HD_RS, (b - (float)hh.p[j][1]) * HD_RS);
    }
    h = hh.v;
    r = rr.v;
}

struct HeadChainHead { const float* dw_w[5]; const float* dw_b[5]; const float* pw_w[5]; const float* pw_b[5]; float* out; int c_out; int act; };
struct HeadChainArgs { const float* x; long rows; int T; const int2* info; const a16x8* frag; const int* frag_bad; int* act_bad; int run_id; HeadChainHead h[3]; };
constexpr float HD_LIM = 32768.0f;

// The point-wise weights in the order the kernel consumes them: per (head, layer, wave, k16 step, plane) 64 lanes x 16 bytes -- lane
// (l31, hi) holds W[32 wave + l31][16 s + 8 hi .. + 7] as f16 heads (plane 0) or scaled residuals (plane 1); rows past the layer's
// width are zeros.  Written once per weights epoch: a wave then fetches its slice with fully coalesced 16-byte loads and no VALU work
// (fetched as f32 rows and split per workgroup it was 32 scattered lines per load instruction, 128 KB per workgroup and layer, and
// the chain kernel ran at a tenth of its MFMA time).
constexpr long HD_FRAG_ELEMS = 3L * 5 * 4 * 8 * 2 * 64;            // a16x8 elements
__global__ __launch_bounds__(64) void grd_head_prep_kernel(const HeadChainArgs a, a16x8* frag, int* frag_bad) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    const int s = blockIdx.x & 7, wave = (blockIdx.x >> 3) & 3, layer = (blockIdx.x >> 5) % 5, head = blockIdx.x / 160;
    const HeadChainHead& H = a.h[head];
    const int n_out = layer < 4 ? HD_D : H.c_out, n = wave * 32 + l31;
    float x8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x8[j] = n < n_out ? H.pw_w[layer][(size_t)n * HD_D + 16 * s + 8 * hi + j] : 0.f;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) bad = bad || !(fabsf(x8[j]) < HD_LIM);
    if (bad) atomicOr(frag_bad, 1);             // (cleared by the launcher before this kernel)
    a16x8 h, r;
    hd_split8(x8, h, r);
    a16x8* dst = frag + ((((long)(head * 5 + layer) * 4 + wave) * 8 + s) * 2) * 64 + lane;
    dst[0] = h;
    dst[64] = r;
}

// EIGHT waves per 64-row tile: wave = (feature block, 32-row block), one accumulator pair per wave, four rows per 32-lane group in the
// depth-wise stage, 128 registers -- four waves per SIMD.  (Round 5; the four-wave form with two 32-row blocks per wave and the next
// layer's weights requested a layer ahead ran 670 against 639 us on the same box; with ONE four-wave workgroup per CU it was 1023 us.
// What bounds the kernel is LDS traffic next to the MFMAs: every wave reads the whole tile's operand planes for its 32 features,
// 240 KB per tile and layer = 1900 clocks of the LDS pipe beside 1536 of the matrix pipe.)
__global__ __launch_bounds__(512, 4) void grd_head_chain_kernel(const HeadChainArgs a) {
    __shared__ __attribute__((aligned(16))) float X[HD_CR * HD_XP];
    __shared__ __attribute__((aligned(16))) unsigned char Ah[HD_CR * HD_AP];
    __shared__ __attribute__((aligned(16))) unsigned char Ar[HD_CR * HD_AP];
    __shared__ int2 rinfo[HD_CR];
    const HeadChainHead& H = a.h[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int fb = wave & 3, rbw = wave >> 2;               // this wave's 32 output features and 32 rows in the point-wise stage
    const long g0 = (long)blockIdx.x * HD_TR - HD_HALO;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 512 * i, row = idx >> 5, c4 = idx & 31;
        const long g = g0 + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (g >= 0 && g < a.rows) v = *(const f32x4*)(a.x + g * HD_D + 4 * c4);
        *(f32x4*)(X + row * HD_XP + 4 * c4) = v;
    }
    bool bad = false;
    if (tid < HD_CR) {
        const long g = g0 + tid;
        int2 ri = {0, 0};
        if (g >= 0 && g < a.rows) ri = a.info ? a.info[g] : int2{(int)(g % a.T), a.T};
        rinfo[tid] = ri;
    }
    __syncthreads();

    const int cq = tid & 31, rg = tid >> 5;                 // depth-wise stage: channels 4 cq .. + 3 of rows 4 rg .. + 3
    auto load_w = [&](int layer, a16x8 (&w)[16]) {
        const a16x8* wf = a.frag + ((((long)(blockIdx.y * 5 + layer) * 4 + fb) * 8) * 2) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = wf[i * 64];
    };
#pragma unroll 1
    for (int layer = 0; layer < 5; ++layer) {
        const int n_out = layer < 4 ? HD_D : H.c_out;
        const bool mm_wave = fb * 32 < n_out;
        {
            const float* dw = H.dw_w[layer] + (size_t)4 * cq * 3;
            f32x4 w0, w1, w2;
#pragma unroll
            for (int c = 0; c < 4; ++c) { w0[c] = dw[3 * c]; w1[c] = dw[3 * c + 1]; w2[c] = dw[3 * c + 2]; }
            const f32x4 bb = *(const f32x4*)(H.dw_b[layer] + 4 * cq);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 xm = rg > 0 ? *(const f32x4*)(X + (4 * rg - 1) * HD_XP + 4 * cq) : zero;
            f32x4 x0 = *(const f32x4*)(X + (4 * rg) * HD_XP + 4 * cq);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = 4 * rg + rr;
                const f32x4 xp = row + 1 < HD_CR ? *(const f32x4*)(X + (row + 1) * HD_XP + 4 * cq) : zero;
                const int2 ri = rinfo[row];
                const f32x4 lo = ri.x > 0 ? xm : zero, up = ri.x + 1 < ri.y ? xp : zero;
                f32x4 v = w0 * lo + w1 * x0 + w2 * up + bb;
                if (ri.y == 0) v = zero;
                bad = bad || !(fabsf(v[0]) < HD_LIM) || !(fabsf(v[1]) < HD_LIM) || !(fabsf(v[2]) < HD_LIM) || !(fabsf(v[3]) < HD_LIM);
                const bigc_fp16x2 h01 = bigc_pk_rtn(v[0], v[1]), h23 = bigc_pk_rtn(v[2], v[3]);
                const bigc_fp16x2 r01 = bigc_pk_rtn((v[0] - (float)h01[0]) * HD_RS, (v[1] - (float)h01[1]) * HD_RS);
                const bigc_fp16x2 r23 = bigc_pk_rtn((v[2] - (float)h23[0]) * HD_RS, (v[3] - (float)h23[1]) * HD_RS);
                union { bigc_fp16x2 p[2]; unsigned long long u; } ph, pr;
                ph.p[0] = h01; ph.p[1] = h23; pr.p[0] = r01; pr.p[1] = r23;
                *(unsigned long long*)(Ah + row * HD_AP + 8 * cq) = ph.u;
                *(unsigned long long*)(Ar + row * HD_AP + 8 * cq) = pr.u;
                xm = x0; x0 = xp;
            }
        }
        __syncthreads();
        if (mm_wave) {
            // (requested here, not a layer ahead: 64 registers that must not live through the depth-wise stage at 128 per wave; the other
            // three waves of the SIMD cover the L2 round trip)
            a16x8 wcur[16];
            load_w(layer, wcur);
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const a16x8 bh = *(const a16x8*)(Ah + (rbw * 32 + l31) * HD_AP + 16 * hi + 32 * s);
                const a16x8 br = *(const a16x8*)(Ar + (rbw * 32 + l31) * HD_AP + 16 * hi + 32 * s);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[2 * s], bh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[2 * s], br, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[2 * s + 1], bh, acc1, 0, 0, 0);
            }
            const int row = rbw * 32 + l31;
            if (layer < 4) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int nn0 = fb * 32 + 8 * r4 + 4 * hi;
                    const f32x4 bias = *(const f32x4*)(H.pw_b[layer] + nn0);
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf((acc0[4 * r4 + c] + acc1[4 * r4 + c] * HD_RI) + bias[c], 0.f);
                    *(f32x4*)(X + row * HD_XP + nn0) = v;
                }
            } else {
                const long g = g0 + row;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nn = fb * 32 + mfma_row(r, hi);
                    if (nn >= n_out) continue;
                    float v = (acc0[r] + acc1[r] * HD_RI) + H.pw_b[layer][nn];
                    if (row >= HD_HALO && row < HD_CR - HD_HALO && g >= 0 && g < a.rows) {
                        if (H.act == BIGC_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                        H.out[g * H.c_out + nn] = v;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (bad) atomicMax(a.act_bad, a.run_id);
}

// Range fallback of the kernel above, launched behind it with the same grid: returns at once unless an activation (flag raised by the chain
// kernel) or a point-wise weight (flag raised when the weights were prepared) was outside the split form; then EVERY tile is recomputed
// through all five layers with plain f32 FMAs from the f32 weights, so that NaN / Inf / huge values come out as the reference's f32
// arithmetic gives them.
__global__ __launch_bounds__(256) void grd_head_exact_kernel(const HeadChainArgs a) {
    if (a.act_bad[0] != a.run_id && a.frag_bad[0] == 0) return;
    __shared__ __attribute__((aligned(16))) float X[HD_CR * HD_XP];
    __shared__ __attribute__((aligned(16))) float Dw[HD_CR * HD_XP];              // the depth-wise result of the current layer
    __shared__ int2 rinfo[HD_CR];
    const HeadChainHead& H = a.h[blockIdx.y];
    const int tid = threadIdx.x, cq = tid & 31, rg = tid >> 5;
    const long g0 = (long)blockIdx.x * HD_TR - HD_HALO;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i, row = idx >> 5, c4 = idx & 31;
        const long g = g0 + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (g >= 0 && g < a.rows) v = *(const f32x4*)(a.x + g * HD_D + 4 * c4);
        *(f32x4*)(X + row * HD_XP + 4 * c4) = v;
    }
    if (tid < HD_CR) {
        const long g = g0 + tid;
        int2 ri = {0, 0};
        if (g >= 0 && g < a.rows) ri = a.info ? a.info[g] : int2{(int)(g % a.T), a.T};
        rinfo[tid] = ri;
    }
    __syncthreads();
    for (int layer = 0; layer < 5; ++layer) {
        const int n_out = layer < 4 ? HD_D : H.c_out;
        {
            const float* dw = H.dw_w[layer] + (size_t)4 * cq * 3;
            f32x4 w0, w1, w2;
            for (int c = 0; c < 4; ++c) { w0[c] = dw[3 * c]; w1[c] = dw[3 * c + 1]; w2[c] = dw[3 * c + 2]; }
            const f32x4 bb = *(const f32x4*)(H.dw_b[layer] + 4 * cq);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            for (int rr = 0; rr < 8; ++rr) {
                const int row = 8 * rg + rr;
                const int2 ri = rinfo[row];
                const f32x4 lo = (row > 0 && ri.x > 0) ? *(const f32x4*)(X + (row - 1) * HD_XP + 4 * cq) : zero;
                const f32x4 up = (row + 1 < HD_CR && ri.x + 1 < ri.y) ? *(const f32x4*)(X + (row + 1) * HD_XP + 4 * cq) : zero;
                f32x4 v = w0 * lo + w1 * *(const f32x4*)(X + row * HD_XP + 4 * cq) + w2 * up + bb;
                if (ri.y == 0) v = zero;
                *(f32x4*)(Dw + row * HD_XP + 4 * cq) = v;
            }
        }
        __syncthreads();
        {
            const int row = tid & 63;                        // a wave shares the output feature (its weight row is wave-uniform)
            const long g = g0 + row;
            for (int nn = tid >> 6; nn < n_out; nn += 4) {
                const float* wr = H.pw_w[layer] + (size_t)nn * HD_D;
                float acc = 0.f;
                for (int k = 0; k < HD_D; ++k) acc = fmaf(Dw[row * HD_XP + k], wr[k], acc);
                float v = acc + H.pw_b[layer][nn];
                if (layer < 4) {
                    X[row * HD_XP + nn] = v < 0.f ? 0.f : v;             // (a NaN stays a NaN, as torch's relu)
                } else if (row >= HD_HALO && row < HD_CR - HD_HALO && g >= 0 && g < a.rows) {
                    if (H.act == BIGC_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                    H.out[g * H.c_out + nn] = v;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

bool grd_head_chain_supported(int d, const Head& a, const Head& b, const Head& c) {
    static const bool off = getenv("BIGC_HEAD_CHAIN") && atoi(getenv("BIGC_HEAD_CHAIN")) == 0;
    return !off && d == HD_D && a.c_out <= 128 && b.c_out <= 128 && c.c_out <= 128 && !bigc_linear_get_exact() && !bigc_linear_get_fast();
}

// heads[i] -> outs[i] (rows, heads[i].c_out) with activation acts[i]; x (rows, 128); sequences: uniform length T, or info[row] = (position, length)
size_t grd_head_frag_bytes() { return (size_t)HD_FRAG_ELEMS * sizeof(a16x8) + 256; }      // + the weights' range flag word behind the fragments

// `frag`: grd_head_frag_bytes() of device memory owned by the model; `refresh`: rebuild it (the weights changed since it was written)
int grd_head_chain(hipStream_t s, const float* x, long rows, int T, const int2* info, const Head* const heads[3], float* const outs[3], const int acts[3],
                   void* frag, bool refresh) {
    if (rows <= 0) return BIGC_OK;
    HeadChainArgs a;
    a.x = x; a.rows = rows; a.T = T > 0 ? T : 1; a.info = info; a.frag = (const a16x8*)frag;
    int* frag_bad = (int*)((char*)frag + (size_t)HD_FRAG_ELEMS * sizeof(a16x8));
    a.frag_bad = frag_bad;
    a.act_bad = frag_bad + 1;
    static std::atomic<int> g_run{0};
    a.run_id = g_run.fetch_add(1, std::memory_order_relaxed) + 1;
    for (int i = 0; i < 3; ++i) {
        for (int l = 0; l < 5; ++l) {
            a.h[i].dw_w[l] = heads[i]->layer[l].dw_w; a.h[i].dw_b[l] = heads[i]->layer[l].dw_b;
            a.h[i].pw_w[l] = heads[i]->layer[l].pw.w; a.h[i].pw_b[l] = heads[i]->layer[l].pw.b;
        }
        a.h[i].out = outs[i]; a.h[i].c_out = heads[i]->c_out; a.h[i].act = acts[i];
    }
    if (refresh) {
        if (hipMemsetAsync(frag_bad, 0, 2 * sizeof(int), s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "grd_head_chain: memset failed");      // (both flag words)
        hipLaunchKernelGGL(grd_head_prep_kernel, dim3(3 * 5 * 4 * 8), dim3(64), 0, s, a, (a16x8*)frag, frag_bad);
        BIGC_LAUNCH_CHECK("grd_head_prep");
    }
    const long tiles = (rows + HD_TR - 1) / HD_TR;
    double flops = 0.0;
    for (int i = 0; i < 3; ++i) flops += 2.0 * rows * HD_D * (4.0 * HD_D + heads[i]->c_out);
    ProfScope ps((void*)s, FAM_GROUNDING, flops, 4.0 * rows * (3.0 * HD_D + heads[0]->c_out + heads[1]->c_out + heads[2]->c_out));
    hipLaunchKernelGGL(grd_head_chain_kernel, dim3((unsigned)tiles, 3), dim3(512), 0, s, a);
    BIGC_LAUNCH_CHECK("grd_head_chain");
    hipLaunchKernelGGL(grd_head_exact_kernel, dim3((unsigned)tiles, 3), dim3(256), 0, s, a);       // gated: returns at once when no flag is up
    BIGC_LAUNCH_CHECK("grd_head_exact");
    return BIGC_OK;
}
