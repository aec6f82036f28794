// A QANet encoder layer of the grounding scorer (models/grd_model_v5.py:110-137, eval mode) in TWO kernels around its attention core:
//
//   grd_enc_chain   out = x + pos;  4 x [ out = relu(point_wise(depth_wise_k(LayerNorm_i(out)))) + out ];  ln = LayerNorm_4(out)
//                   (:124-131 and the normalisation in front of the attention, :132)       -> `out` and `ln`, (rows, 128) each
//   grd_attn_tail   r = out_proj(att) + out;  y = relu(fc(LayerNorm_e(r))) + r                (:133-136)        -> y (rows, 128)
//
// Layer by layer (csrc/grounding.hip: qanet) the first part is 1 + 4 x 2 + 1 launches that each read and write the whole (rows, 128)
// activation -- 10 passes of 123 MB each way on the combined encoder of the benchmark's 241 k rows, 0.65 of the leg's 3.3 ms -- and the
// second 3.  Here a workgroup keeps a tile of 64 consecutive rows in LDS through all four convolution layers: the activation is read once
// (1.6 times with the halo), the normalised and the convolved tensors never exist in memory, `out` and `ln` are written once.
//
// Tile and halo: rows [g0, g0 + 64) of the row-major activation (sequences back to back).  A k-tap depth-wise convolution needs k / 2
// neighbours on either side inside the same sequence, so after four layers rows 4 (k / 2) .. 63 - 4 (k / 2) of the tile are exact
// (40 rows for k = 7, 56 for k = 3) and are the tile's output; the rim is recomputed by the neighbouring tiles.  Taps outside a
// sequence contribute zeros, as the reference's padding (:44).  The tail is row-local: 64 of 64 rows.
//
// Arithmetic, range flag and fallback are those of csrc/grounding_heads.hip: point-wise layers on the f16 matrix cores with exactly
// split f32 operands (three MFMAs per product, two accumulators, weights = the MFMA's A operand in fragment order, prepared once per
// weights epoch), LayerNorm statistics two-pass in f32 (four lanes per row, 32 channels each, summed by DPP quad permutes), depth-wise taps f32 FMAs in ascending tap order.  A tile that meets a value outside the split
// form's range (|v| >= 32768, NaN) raises a flag; the EXACT instantiation, launched right behind and gated on the flags, then recomputes
// the launch with plain f32 FMAs from the f32 weights, so that NaN / Inf come out as the reference's arithmetic gives them.
#include "grounding_internal.h"
#include "prof.h"
#include <stdlib.h>
#include <atomic>

typedef _Float16 a16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int EC_D = 128, EC_CR = 64;
constexpr int EC_XP = 132;                  // pitch (floats) of the f32 activation tile
constexpr int EC_AP = 272;                  // pitch (bytes) of a row of a split plane: 128 f16 + 16
constexpr int EC_LAYERS = 6;                // 4 point-wise convolutions, out_proj, fc
constexpr float EC_RS = 2048.0f, EC_RI = 1.0f / 2048.0f, EC_LIM = 32768.0f;
constexpr long EC_FRAG_ELEMS = (long)EC_LAYERS * 4 * 8 * 2 * 64;            // a16x8 elements per encoder

struct EncChainArgs {
    const float* x; const float* pos; const float* res; long rows; int T; const int2* info;
    const a16x8* frag; const int* frag_bad; int* act_bad; int run_id;     // act_bad holds the id of the last launch that met an out-of-range operand (no memset per launch)
    const float* dwt; int ks;                             // depth-wise weights as (layer, tap (8), channel): prepared with the fragments
    const float* ln_g[6]; const float* ln_b[6];           // [0..3] in front of convolution i, [4] in front of the attention, [5] in front of fc
    const float* dw_w[4]; const float* dw_b[4];
    const float* pw_w[EC_LAYERS]; const float* pw_b[EC_LAYERS];
    float* out; float* ln_out; float eps;
};

__device__ __forceinline__ bool ec_bad4(const f32x4 v) {
    return !(fabsf(v[0]) < EC_LIM) || !(fabsf(v[1]) < EC_LIM) || !(fabsf(v[2]) < EC_LIM) || !(fabsf(v[3]) < EC_LIM);
}

// four values -> 8 bytes of f16 heads and 8 bytes of f16 residuals (x 2048) at columns 4 cq .. + 3 of row `row` of the two planes
__device__ __forceinline__ void ec_store_split(const f32x4 v, unsigned char* Ah, unsigned char* Ar, int row, int cq) {
    const bigc_fp16x2 h01 = bigc_pk_rtn(v[0], v[1]), h23 = bigc_pk_rtn(v[2], v[3]);
    const bigc_fp16x2 r01 = bigc_pk_rtn((v[0] - (float)h01[0]) * EC_RS, (v[1] - (float)h01[1]) * EC_RS);
    const bigc_fp16x2 r23 = bigc_pk_rtn((v[2] - (float)h23[0]) * EC_RS, (v[3] - (float)h23[1]) * EC_RS);
    union { bigc_fp16x2 p[2]; unsigned long long u; } ph, pr;
    ph.p[0] = h01; ph.p[1] = h23; pr.p[0] = r01; pr.p[1] = r23;
    *(unsigned long long*)(Ah + row * EC_AP + 8 * cq) = ph.u;
    *(unsigned long long*)(Ar + row * EC_AP + 8 * cq) = pr.u;
}

// The point-wise weights in the order the kernels consume them (as grd_head_prep_kernel): per (layer, wave, k16 step, plane) 64 lanes x
// 16 bytes -- lane (l31, hi) holds W[32 wave + l31][16 s + 8 hi .. + 7] as f16 heads (plane 0) or scaled residuals (plane 1).
__global__ __launch_bounds__(64) void grd_enc_prep_kernel(const EncChainArgs a, a16x8* frag, int* frag_bad) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    const int s = blockIdx.x & 7, wave = (blockIdx.x >> 3) & 3, layer = blockIdx.x >> 5;
    if (layer >= EC_LAYERS) {                   // four more workgroups: the depth-wise weights (channel, tap) -> (tap, channel), a 16-byte load per tap and lane
        const int dl = blockIdx.x - EC_LAYERS * 32;
        float* dwt = (float*)a.dwt + (size_t)dl * 8 * EC_D;
        for (int i = lane; i < 8 * EC_D; i += 64) {
            const int j = i / EC_D, c = i - j * EC_D;
            dwt[i] = j < a.ks ? a.dw_w[dl][(size_t)c * a.ks + j] : 0.f;
        }
        return;
    }
    const int n = wave * 32 + l31;
    float x8[8];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x8[j] = a.pw_w[layer][(size_t)n * EC_D + 16 * s + 8 * hi + j]; bad = bad || !(fabsf(x8[j]) < EC_LIM); }
    if (bad) atomicOr(frag_bad, 1);             // (cleared by the launcher before this kernel)
    union { bigc_fp16x2 p[4]; a16x8 v; } hh, rr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float u = x8[2 * j], v = x8[2 * j + 1];
        hh.p[j] = bigc_pk_rtn(u, v);
        rr.p[j] = bigc_pk_rtn((u - (float)hh.p[j][0]) * EC_RS, (v - (float)hh.p[j][1]) * EC_RS);
    }
    a16x8* dst = frag + ((((long)layer * 4 + wave) * 8 + s) * 2) * 64 + lane;
    dst[0] = hh.v;
    dst[64] = rr.v;
}

// this wave's slice of a layer's point-wise weight, already split: 16 coalesced 16-byte loads per lane
__device__ __forceinline__ void ec_load_w(const a16x8* frag, int layer, int wave, int lane, a16x8 (&w)[16]) {
    const a16x8* wf = frag + (((long)layer * 4 + wave) * 8 * 2) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = wf[i * 64];
}

// acc[i] (feature 32 fb + mfma_row(r, hi), data row 32 (rb0 + i) + l31) = sum_k W[feature][k] act[row][k]: NRB 32-row blocks in one sweep over k
template <int NRB>
__device__ __forceinline__ void ec_mm(const unsigned char* Ah, const unsigned char* Ar, const a16x8 (&w)[16], int rb0, int l31, int hi,
                                      f32x16 (&acc0)[NRB], f32x16 (&acc1)[NRB]) {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[rb][r] = 0.f; acc1[rb][r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const a16x8 bh = *(const a16x8*)(Ah + ((rb0 + rb) * 32 + l31) * EC_AP + 16 * hi + 32 * s);
            const a16x8 br = *(const a16x8*)(Ar + ((rb0 + rb) * 32 + l31) * EC_AP + 16 * hi + 32 * s);
            acc0[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * s], bh, acc0[rb], 0, 0, 0);
            acc1[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * s], br, acc1[rb], 0, 0, 0);
            acc1[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * s + 1], bh, acc1[rb], 0, 0, 0);
        }
    }
}

struct EcSmem {
    float X[EC_CR * EC_XP];                           // the running activation (`out` of the reference), f32
    unsigned char A2[2 * EC_CR * EC_AP];              // split planes of the current GEMM operand; EXACT: the same operand in f32
    int2 rinfo[EC_CR];                                // (position, length of the sequence) of a tile row; length 0 = outside the tensor
    float2 stat[EC_CR];                               // (mean, 1 / std) of a tile row for the current LayerNorm
};

// sum over the four lanes of a quad, in every lane (two DPP quad permutes: no LDS traffic, no bpermute latency)
__device__ __forceinline__ float ec_quad_sum(float t) {
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));     // lanes [1, 0, 3, 2]
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, true));     // lanes [2, 3, 0, 1]
    return t;
}

// LayerNorm statistics of the tile's 64 rows -> S.stat: four threads per row, 32 channels each, two-pass in f32 (mean, then the squared
// deviations from it), biased variance, eps inside the root (torch.nn.LayerNorm)
__device__ __forceinline__ void ec_stats(EcSmem& S, int tid, float eps) {
    if (tid >= 4 * EC_CR) return;                 // (eight-wave workgroups: the first four waves)
    const int row = tid >> 2, q = tid & 3;
    const float* xp = S.X + row * EC_XP + 32 * q;
    f32x4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = *(const f32x4*)(xp + 4 * i); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    const float mean = ec_quad_sum(s) / (float)EC_D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float t = v[i][c] - mean; ss += t * t; }
    const float rstd = 1.0f / sqrtf(ec_quad_sum(ss) / (float)EC_D + eps);
    if (q == 0) S.stat[row] = make_float2(mean, rstd);
}

// one point-wise layer on the tile: S.A2 (operand) x W[layer] -> epilogue(row, first feature nn0 of a quad, the four results + bias).
// NW = 4 waves: a wave owns 32 features and both 32-row blocks, and `next` >= 0 requests the wave's weight slice of that layer into `wcur`
// right behind the MFMAs (it lands while the epilogue and the next layer's row-wise stages run: with two waves per SIMD nothing else would
// hide an L2 round trip per layer).  NW = 8 waves: a wave owns 32 features and ONE 32-row block and fetches its slice where it uses it
// (64 registers that must not live through the row-wise stages at 128 per wave; the other three waves of the SIMD cover the round trip).
template <bool EXACT, int NW, typename Epi>
__device__ __forceinline__ void ec_pointwise(const EncChainArgs& a, EcSmem& S, int layer, a16x8 (&wcur)[16], int next, int tid, Epi epi) {
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    constexpr int NRB = 8 / NW;
    const int fb = wave & 3, rb0 = NW == 8 ? wave >> 2 : 0;
    if (!EXACT) {
        unsigned char* Ah = S.A2;
        unsigned char* Ar = S.A2 + EC_CR * EC_AP;
        f32x16 acc0[NRB], acc1[NRB];
        if (NW == 8) {
            a16x8 w[16];
            ec_load_w(a.frag, layer, fb, lane, w);
            ec_mm<NRB>(Ah, Ar, w, rb0, l31, hi, acc0, acc1);
        } else {
            ec_mm<NRB>(Ah, Ar, wcur, rb0, l31, hi, acc0, acc1);
            if (next >= 0) ec_load_w(a.frag, next, fb, lane, wcur);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int nn0 = fb * 32 + 8 * r4 + 4 * hi;
            const f32x4 bias = *(const f32x4*)(a.pw_b[layer] + nn0);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = (acc0[rb][4 * r4 + c] + acc1[rb][4 * r4 + c] * EC_RI) + bias[c];
                epi((rb0 + rb) * 32 + l31, nn0, v);
            }
        }
    } else {
        const float* Dw = (const float*)S.A2;
        const int row = tid & 63;                         // a wave shares the output features (its weight rows are wave-uniform)
        for (int nn0 = 4 * (tid >> 6); nn0 < EC_D; nn0 += 4 * NW) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* wr = a.pw_w[layer] + (size_t)(nn0 + c) * EC_D;
                float acc = 0.f;
                for (int k = 0; k < EC_D; ++k) acc = fmaf(Dw[row * EC_XP + k], wr[k], acc);
                v[c] = acc + a.pw_b[layer][nn0 + c];
            }
            epi(row, nn0, v);
        }
    }
}

__device__ __forceinline__ f32x4 ec_relu(const f32x4 v) {                    // (a NaN stays a NaN, as torch's relu)
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = v[c] < 0.f ? 0.f : v[c];
    return o;
}

template <int KS, bool EXACT, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void grd_enc_chain_kernel(const EncChainArgs a) {
    constexpr int NG = 2 * NW, RPG = EC_CR / NG;           // 32-lane groups of the row-wise stages; consecutive rows per group
    if (EXACT && a.act_bad[0] != a.run_id && a.frag_bad[0] == 0) return;
    constexpr int H = KS / 2, HALO = 4 * H, TR = EC_CR - 2 * HALO;
    __shared__ __attribute__((aligned(16))) EcSmem S;
    unsigned char* Ah = S.A2;
    unsigned char* Ar = S.A2 + EC_CR * EC_AP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cq = tid & 31, rg = tid >> 5;                 // row-wise stages: channels 4 cq .. + 3 of rows RPG rg .. + RPG - 1
    const long g0 = (long)blockIdx.x * TR - HALO;

    // ---- the tile: x + pos[position in the sequence] (:124), rows outside the tensor are zeros of length 0
#pragma unroll
    for (int i = 0; i < RPG; ++i) {
        const int row = rg + NG * i;
        const long g = g0 + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int2 ri = {0, 0};
        if (g >= 0 && g < a.rows) {
            ri = a.info ? a.info[g] : int2{(int)(g % a.T), a.T};
            v = *(const f32x4*)(a.x + g * EC_D + 4 * cq) + *(const f32x4*)(a.pos + (size_t)ri.x * EC_D + 4 * cq);
        }
        *(f32x4*)(S.X + row * EC_XP + 4 * cq) = v;
        if (cq == 0) S.rinfo[row] = ri;
    }
    __syncthreads();

    a16x8 wcur[16];
    if (!EXACT && NW == 4) ec_load_w(a.frag, 0, wave, lane, wcur);
    bool bad = false;
#pragma unroll 1
    for (int layer = 0; layer < 4; ++layer) {
        f32x4 wt[KS];                                        // (requested ahead of the statistics: in flight across them and the barrier)
#pragma unroll
        for (int j = 0; j < KS; ++j) wt[j] = *(const f32x4*)(a.dwt + ((size_t)layer * 8 + j) * EC_D + 4 * cq);
        const f32x4 bb = *(const f32x4*)(a.dw_b[layer] + 4 * cq);
        const f32x4 gg = *(const f32x4*)(a.ln_g[layer] + 4 * cq), be = *(const f32x4*)(a.ln_b[layer] + 4 * cq);
        ec_stats(S, tid, a.eps);
        __syncthreads();
        // ---- LayerNorm -> depth-wise k taps along the rows of a sequence, zero padded -> the point-wise GEMM's operand
        {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            auto ln_row = [&](int r) {
                if (r < 0 || r >= EC_CR) return zero;            // (only the rim rows see these)
                const f32x4 v = *(const f32x4*)(S.X + r * EC_XP + 4 * cq);
                const float2 st = S.stat[r];
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (v[c] - st.x) * st.y * gg[c] + be[c];
                return o;
            };
            f32x4 win[KS];                                    // win[j] = normalised row (row + j - H)
#pragma unroll
            for (int j = 1; j < KS; ++j) win[j] = ln_row(RPG * rg + j - 1 - H);
#pragma unroll
            for (int rr = 0; rr < RPG; ++rr) {
                const int row = RPG * rg + rr;
#pragma unroll
                for (int j = 0; j + 1 < KS; ++j) win[j] = win[j + 1];
                win[KS - 1] = ln_row(row + H);
                const int2 ri = S.rinfo[row];
                f32x4 acc = bb;
                // (explicit FMAs in both branches: a row's result must not depend on which rows share its wave -- batched == one video per call)
                if (__all(ri.x >= H && ri.x + H < ri.y)) {          // every tap of every row of this wave inside its sequence
#pragma unroll
                    for (int j = 0; j < KS; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = fmaf(wt[j][c], win[j][c], acc[c]);
                } else {
#pragma unroll
                    for (int j = 0; j < KS; ++j) {
                        const int tt = ri.x + j - H;
                        const bool in = tt >= 0 && tt < ri.y;
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = in ? fmaf(wt[j][c], win[j][c], acc[c]) : acc[c];
                    }
                }
                if (ri.y == 0) acc = zero;
                if (!EXACT) {
                    bad = bad || ec_bad4(acc);
                    ec_store_split(acc, Ah, Ar, row, cq);
                } else {
                    *(f32x4*)((float*)S.A2 + row * EC_XP + 4 * cq) = acc;
                }
            }
        }
        __syncthreads();
        ec_pointwise<EXACT, NW>(a, S, layer, wcur, layer < 3 ? layer + 1 : -1, tid, [&](int row, int nn0, const f32x4 v) {
            f32x4* xp = (f32x4*)(S.X + row * EC_XP + nn0);
            *xp = ec_relu(v) + *xp;                              // relu(conv) + res  (:130-131)
        });
        __syncthreads();
    }

    // ---- the tile's exact rows: `out` and LayerNorm_4(out) (the attention's input, :132)
    ec_stats(S, tid, a.eps);
    __syncthreads();
    {
        const f32x4 gg = *(const f32x4*)(a.ln_g[4] + 4 * cq), be = *(const f32x4*)(a.ln_b[4] + 4 * cq);
#pragma unroll
        for (int rr = 0; rr < RPG; ++rr) {
            const int row = rg + NG * rr;
            const long g = g0 + row;
            if (row < HALO || row >= EC_CR - HALO || g < 0 || g >= a.rows) continue;
            const f32x4 v = *(const f32x4*)(S.X + row * EC_XP + 4 * cq);
            const float2 st = S.stat[row];
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (v[c] - st.x) * st.y * gg[c] + be[c];
            *(f32x4*)(a.out + g * EC_D + 4 * cq) = v;
            *(f32x4*)(a.ln_out + g * EC_D + 4 * cq) = o;
        }
    }
    if (!EXACT && bad) atomicMax(a.act_bad, a.run_id);
}

// r = out_proj(att) + res;  y = relu(fc(LayerNorm_e(r))) + r      (x = att, res = the chain's `out`, out = y; row-local)
template <bool EXACT, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void grd_attn_tail_kernel(const EncChainArgs a) {
    constexpr int NG = 2 * NW, RPG = EC_CR / NG;
    if (EXACT && a.act_bad[0] != a.run_id && a.frag_bad[0] == 0) return;
    __shared__ __attribute__((aligned(16))) EcSmem S;
    unsigned char* Ah = S.A2;
    unsigned char* Ar = S.A2 + EC_CR * EC_AP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cq = tid & 31, rg = tid >> 5;
    const long g0 = (long)blockIdx.x * EC_CR;
    a16x8 wcur[16];
    if (!EXACT && NW == 4) ec_load_w(a.frag, 4, wave, lane, wcur);
    bool bad = false;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RPG; ++i) {
        const int row = rg + NG * i;
        const long g = g0 + row;
        f32x4 v = zero, r = zero;
        if (g < a.rows) { v = *(const f32x4*)(a.x + g * EC_D + 4 * cq); r = *(const f32x4*)(a.res + g * EC_D + 4 * cq); }
        *(f32x4*)(S.X + row * EC_XP + 4 * cq) = r;
        if (!EXACT) {
            bad = bad || ec_bad4(v);
            ec_store_split(v, Ah, Ar, row, cq);
        } else {
            *(f32x4*)((float*)S.A2 + row * EC_XP + 4 * cq) = v;
        }
    }
    __syncthreads();
    ec_pointwise<EXACT, NW>(a, S, 4, wcur, 5, tid, [&](int row, int nn0, const f32x4 v) {
        f32x4* xp = (f32x4*)(S.X + row * EC_XP + nn0);
        *xp = v + *xp;                                           // attn + res (:134)
    });
    __syncthreads();
    ec_stats(S, tid, a.eps);
    __syncthreads();
    {
        const f32x4 gg = *(const f32x4*)(a.ln_g[5] + 4 * cq), be = *(const f32x4*)(a.ln_b[5] + 4 * cq);
#pragma unroll
        for (int rr = 0; rr < RPG; ++rr) {
            const int row = rg + NG * rr;
            const f32x4 v = *(const f32x4*)(S.X + row * EC_XP + 4 * cq);
            const float2 st = S.stat[row];
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (v[c] - st.x) * st.y * gg[c] + be[c];
            if (g0 + row >= a.rows) o = zero;
            if (!EXACT) {
                bad = bad || ec_bad4(o);
                ec_store_split(o, Ah, Ar, row, cq);
            } else {
                *(f32x4*)((float*)S.A2 + row * EC_XP + 4 * cq) = o;
            }
        }
    }
    __syncthreads();
    // (through the tile, then out by whole rows: stored straight from the accumulator layout every instruction touched 64 different lines)
    ec_pointwise<EXACT, NW>(a, S, 5, wcur, -1, tid, [&](int row, int nn0, const f32x4 v) {
        f32x4* xp = (f32x4*)(S.X + row * EC_XP + nn0);
        *xp = ec_relu(v) + *xp;                                  // relu(fc) + res (:135-136)
    });
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPG; ++i) {
        const int row = rg + NG * i;
        const long g = g0 + row;
        if (g < a.rows) *(f32x4*)(a.out + g * EC_D + 4 * cq) = *(const f32x4*)(S.X + row * EC_XP + 4 * cq);
    }
    if (!EXACT && bad) atomicMax(a.act_bad, a.run_id);
}

}  // namespace

bool grd_enc_chain_supported(int d, int ks) {
    static const bool off = getenv("BIGC_ENC_CHAIN") && atoi(getenv("BIGC_ENC_CHAIN")) == 0;
    return !off && d == EC_D && (ks == 3 || ks == 7) && !bigc_linear_get_exact() && !bigc_linear_get_fast();
}

size_t grd_enc_frag_bytes() { return (size_t)EC_FRAG_ELEMS * sizeof(a16x8) + 256 + sizeof(float) * 4 * 8 * EC_D; }      // + the flag words + the depth-wise weights by tap

namespace {
void enc_args(const Enc& E, long rows, int T, const int2* info, void* frag, EncChainArgs& a) {
    a.rows = rows; a.T = T > 0 ? T : 1; a.info = info; a.frag = (const a16x8*)frag; a.eps = 1e-5f;
    int* flags = (int*)((char*)frag + (size_t)EC_FRAG_ELEMS * sizeof(a16x8));
    a.frag_bad = flags;
    a.act_bad = flags + 1;
    static std::atomic<int> g_run{0};
    a.run_id = g_run.fetch_add(1, std::memory_order_relaxed) + 1;
    a.dwt = (const float*)((char*)flags + 256); a.ks = E.ks;
    const Lin* ln[6] = {&E.normb, &E.norm_seq[0], &E.norm_seq[1], &E.norm_seq[2], &E.norm_seq[3], &E.norme};
    for (int i = 0; i < 6; ++i) { a.ln_g[i] = ln[i]->w; a.ln_b[i] = ln[i]->b; }
    for (int i = 0; i < 4; ++i) {
        a.dw_w[i] = E.conv[i].dw_w; a.dw_b[i] = E.conv[i].dw_b;
        a.pw_w[i] = E.conv[i].pw.w; a.pw_b[i] = E.conv[i].pw.b;
    }
    a.pw_w[4] = E.out_proj.w; a.pw_b[4] = E.out_proj.b;
    a.pw_w[5] = E.fc.w; a.pw_b[5] = E.fc.b;
}
}  // namespace

// `frag`: grd_enc_frag_bytes() of device memory owned by the model for this encoder; `refresh`: rebuild it (the weights changed since)
int grd_enc_chain(hipStream_t s, const Enc& E, const float* x, const float* pos, long rows, int T, const int2* info, void* frag, bool refresh,
                  float* out, float* ln_out) {
    if (rows <= 0) return BIGC_OK;
    EncChainArgs a;
    enc_args(E, rows, T, info, frag, a);
    a.x = x; a.pos = pos; a.res = nullptr; a.out = out; a.ln_out = ln_out;
    if (refresh) {                             // (also the first use of the buffer: both flag words start at zero)
        if (hipMemsetAsync((void*)a.frag_bad, 0, 2 * sizeof(int), s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "grd_enc_chain: memset failed");
        hipLaunchKernelGGL(grd_enc_prep_kernel, dim3(EC_LAYERS * 32 + 4), dim3(64), 0, s, a, (a16x8*)frag, (int*)a.frag_bad);
        BIGC_LAUNCH_CHECK("grd_enc_prep");
    }
    const int tr = EC_CR - 8 * (E.ks / 2);
    const long tiles = (rows + tr - 1) / tr;
    ProfScope ps((void*)s, FAM_GROUNDING, 2.0 * rows * EC_D * 4.0 * EC_D, 4.0 * rows * 3.0 * EC_D);
    static const int waves = getenv("BIGC_ENC_WAVES") ? atoi(getenv("BIGC_ENC_WAVES")) : 8;
    if (E.ks == 7) {
        if (waves == 8) hipLaunchKernelGGL((grd_enc_chain_kernel<7, false, 8>), dim3((unsigned)tiles), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((grd_enc_chain_kernel<7, false, 4>), dim3((unsigned)tiles), dim3(256), 0, s, a);
        hipLaunchKernelGGL((grd_enc_chain_kernel<7, true, 4>), dim3((unsigned)tiles), dim3(256), 0, s, a);     // gated: returns at once when no flag is up
    } else {
        if (waves == 8) hipLaunchKernelGGL((grd_enc_chain_kernel<3, false, 8>), dim3((unsigned)tiles), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((grd_enc_chain_kernel<3, false, 4>), dim3((unsigned)tiles), dim3(256), 0, s, a);
        hipLaunchKernelGGL((grd_enc_chain_kernel<3, true, 4>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    }
    BIGC_LAUNCH_CHECK("grd_enc_chain");
    return BIGC_OK;
}

// (the fragments were written by this layer's grd_enc_chain call of the same forward)
int grd_attn_tail(hipStream_t s, const Enc& E, const float* att, const float* res, long rows, void* frag, float* out) {
    if (rows <= 0) return BIGC_OK;
    EncChainArgs a;
    enc_args(E, rows, 1, nullptr, frag, a);
    a.x = att; a.pos = nullptr; a.res = res; a.out = out; a.ln_out = nullptr;
    const long tiles = (rows + EC_CR - 1) / EC_CR;
    ProfScope ps((void*)s, FAM_GROUNDING, 2.0 * rows * EC_D * 2.0 * EC_D, 4.0 * rows * 3.0 * EC_D);
    static const int waves = getenv("BIGC_ENC_WAVES") ? atoi(getenv("BIGC_ENC_WAVES")) : 8;
    if (waves == 8) hipLaunchKernelGGL((grd_attn_tail_kernel<false, 8>), dim3((unsigned)tiles), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((grd_attn_tail_kernel<false, 4>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    hipLaunchKernelGGL((grd_attn_tail_kernel<true, 4>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    BIGC_LAUNCH_CHECK("grd_attn_tail");
    return BIGC_OK;
}
