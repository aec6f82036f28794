// A QANet encoder layer of the grounding scorer (models/grd_model_v5.py:110-137, eval mode) in TWO kernels around its attention core:
//
//   grd_enc_chain   out = x + pos;  4 x [ out = relu(point_wise(depth_wise_k(LayerNorm_i(out)))) + out ];  ln = LayerNorm_4(out);
//                   qkv = in_proj(ln)   (:124-131, the normalisation in front of the attention and its packed projection, :132-133)
//                                                                                          -> `out` (rows, 128) and `qkv` (rows, 384)
//   grd_attn_tail   r = out_proj(att) + out;  y = relu(fc(LayerNorm_e(r))) + r                (:133-136)        -> y (rows, 128)
//
// Layer by layer (csrc/grounding.hip: qanet) the first part is 1 + 4 x 2 + 1 launches that each read and write the whole (rows, 128)
// activation -- 10 passes of 123 MB each way on the combined encoder of the benchmark's 241 k rows, 0.65 of the leg's 3.3 ms -- and the
// second 3.  Here a workgroup keeps a tile of 64 consecutive rows in LDS through all four convolution layers: the activation is read once
// (1.6 times with the halo; 1.23 times on the 128-row tiles of big launches), the normalised and the convolved tensors never exist in
// memory, `out` and `qkv` are written once.
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
#include <mutex>

typedef _Float16 a16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int EC_D = 128, EC_CR = 64;
constexpr int EC_XP = 132;                  // pitch (floats) of the f32 activation tile
constexpr int EC_AP = 272;                  // pitch (bytes) of a row of a split plane: 128 f16 + 16
constexpr int EC_LAYERS = 9;                // 4 point-wise convolutions, out_proj, fc, and the attention's in_proj as three 128 x 128 blocks (q, k, v)
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
    float* qkv_out;                                       // (rows, 384): in_proj(LayerNorm_4(out)) instead of ln_out (chain kernel)
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
template <int NRB, bool ZERO = true>
__device__ __forceinline__ void ec_mm(const unsigned char* Ah, const unsigned char* Ar, const a16x8 (&w)[16], int rb0, int l31, int hi,
                                      f32x16 (&acc0)[NRB], f32x16 (&acc1)[NRB]) {
    if (ZERO) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[rb][r] = 0.f; acc1[rb][r] = 0.f; }
    }
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

template <int CR = EC_CR>
struct EcSmemT {
    float X[CR * EC_XP];                              // the running activation (`out` of the reference), f32
    unsigned char A2[2 * CR * EC_AP];                 // split planes of the current GEMM operand; EXACT: the same operand in f32
    int2 rinfo[CR];                                   // (position, length of the sequence) of a tile row; length 0 = outside the tensor
    float2 stat[CR];                                  // (mean, 1 / std) of a tile row for the current LayerNorm
};
typedef EcSmemT<EC_CR> EcSmem;

// sum over the four lanes of a quad, in every lane (two DPP quad permutes: no LDS traffic, no bpermute latency)
__device__ __forceinline__ float ec_quad_sum(float t) {
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));     // lanes [1, 0, 3, 2]
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, true));     // lanes [2, 3, 0, 1]
    return t;
}

// LayerNorm statistics of the tile's 64 rows -> S.stat: four threads per row, 32 channels each, two-pass in f32 (mean, then the squared
// deviations from it), biased variance, eps inside the root (torch.nn.LayerNorm)
template <int CR>
__device__ __forceinline__ void ec_stats(EcSmemT<CR>& S, int tid, float eps) {
    if (tid >= 4 * CR) return;                    // (eight / sixteen-wave workgroups: the first half of the waves)
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
template <bool EXACT, int NW, int CR, typename Epi>
__device__ __forceinline__ void ec_pointwise(const EncChainArgs& a, EcSmemT<CR>& S, int layer, a16x8 (&wcur)[16], int next, int tid, Epi epi) {
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    constexpr int NRB = (CR / 32) * 4 / NW;               // 32-row blocks per wave: 2 at four waves, 1 at eight (64 rows) and sixteen (128 rows)
    const int fb = wave & 3, rb0 = (wave >> 2) * NRB;
    if (!EXACT) {
        unsigned char* Ah = S.A2;
        unsigned char* Ar = S.A2 + CR * EC_AP;
        f32x16 acc0[NRB], acc1[NRB];
        if (NW >= 8) {
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

// CR = rows of the tile: 64 (eight waves, two workgroups per CU), or 128 with sixteen waves (one workgroup per CU, the same four waves per SIMD)
// for big launches of the k = 7 encoders, whose 24 halo rows are 37 % of a 64-row tile and 19 % of a 128-row one: 480 -> 427 us on the
// benchmark's combined encoder (eight waves on 128 rows, two row blocks per wave and half the weight fetches: 462).
template <int KS, bool EXACT, int NW, int CR = EC_CR>
__global__ __launch_bounds__(64 * NW, NW == 16 ? 4 : NW / 2) void grd_enc_chain_kernel(const EncChainArgs a) {
    constexpr int NG = 2 * NW, RPG = CR / NG;              // 32-lane groups of the row-wise stages; consecutive rows per group
    if (EXACT && a.act_bad[0] != a.run_id && a.frag_bad[0] == 0) return;
    constexpr int H = KS / 2, HALO = 4 * H, TR = CR - 2 * HALO;
    extern __shared__ __attribute__((aligned(16))) unsigned char ec_raw[];
    EcSmemT<CR>& S = *(EcSmemT<CR>*)ec_raw;
    unsigned char* Ah = S.A2;
    unsigned char* Ar = S.A2 + CR * EC_AP;
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
                if (r < 0 || r >= CR) return zero;            // (only the rim rows see these)
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
        ec_pointwise<EXACT, NW, CR>(a, S, layer, wcur, layer < 3 ? layer + 1 : (a.qkv_out ? 6 : -1), tid, [&](int row, int nn0, const f32x4 v) {
            f32x4* xp = (f32x4*)(S.X + row * EC_XP + nn0);
            *xp = ec_relu(v) + *xp;                              // relu(conv) + res  (:130-131)
        });
        __syncthreads();
    }

    // ---- the tile's exact rows: `out` and LayerNorm_4(out) (the attention's input, :132) -- or, with qkv_out, the attention's in_proj on
    // the normalised tile right here (:132-133, nn.MultiheadAttention's packed projection): the normalised tensor is neither written nor
    // read back, q / k / v leave through the tile as whole rows
    ec_stats(S, tid, a.eps);
    __syncthreads();
    {
        const f32x4 gg = *(const f32x4*)(a.ln_g[4] + 4 * cq), be = *(const f32x4*)(a.ln_b[4] + 4 * cq);
#pragma unroll
        for (int rr = 0; rr < RPG; ++rr) {
            const int row = rg + NG * rr;
            const long g = g0 + row;
            const bool own = !(row < HALO || row >= CR - HALO || g < 0 || g >= a.rows);
            if (!own && !a.qkv_out) continue;
            const f32x4 v = *(const f32x4*)(S.X + row * EC_XP + 4 * cq);
            const float2 st = S.stat[row];
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (v[c] - st.x) * st.y * gg[c] + be[c];
            if (own) *(f32x4*)(a.out + g * EC_D + 4 * cq) = v;
            if (!a.qkv_out) {
                *(f32x4*)(a.ln_out + g * EC_D + 4 * cq) = o;
            } else {
                if (!own) o = f32x4{0.f, 0.f, 0.f, 0.f};          // (rim rows: computed and dropped)
                if (!EXACT) {
                    bad = bad || ec_bad4(o);
                    ec_store_split(o, Ah, Ar, row, cq);
                } else {
                    *(f32x4*)((float*)S.A2 + row * EC_XP + 4 * cq) = o;
                }
            }
        }
    }
    if (a.qkv_out) {
        __syncthreads();
#pragma unroll 1
        for (int part = 0; part < 3; ++part) {
            ec_pointwise<EXACT, NW, CR>(a, S, 6 + part, wcur, part < 2 ? 7 + part : -1, tid, [&](int row, int nn0, const f32x4 v) {
                *(f32x4*)(S.X + row * EC_XP + nn0) = v;
            });
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < RPG; ++rr) {
                const int row = rg + NG * rr;
                const long g = g0 + row;
                if (row < HALO || row >= CR - HALO || g < 0 || g >= a.rows) continue;
                *(f32x4*)(a.qkv_out + g * (3 * EC_D) + part * EC_D + 4 * cq) = *(const f32x4*)(S.X + row * EC_XP + 4 * cq);
            }
            __syncthreads();
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
    ec_pointwise<EXACT, NW, EC_CR>(a, S, 4, wcur, 5, tid, [&](int row, int nn0, const f32x4 v) {
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
    ec_pointwise<EXACT, NW, EC_CR>(a, S, 5, wcur, -1, tid, [&](int row, int nn0, const f32x4 v) {
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

// ---- co-attention + the fusion layer in one kernel (round 6) -----------------------------------------------------------------------------
// models/grd_model_v5.py:346-366: sim = proj2sim(video) @ query^T, its two softmaxes, A = sim_r @ query, B = sim_r @ (sim_c^T @ video), then
// vq_fc([video, A, video * A, video * B]) (4 d -> d).  As two launches (coattention_kernel + a GEMM, csrc/grounding.hip) the (rows, 4 d)
// fusion tensor is written and read back: 470 MB each way on the benchmark's 230 k rows, 0.30 of the leg's 2.2 ms.  Here the workgroup of a
// query keeps the three softmaxed score rows in LDS as coattention_kernel does (the same statements, except the order of the S2 sum) and
// then walks its T clips in blocks of 64.  Of vq_fc's four d-wide K chunks two do not need the (query, clip) rows at all:
//     W0 video[t] + bias                   depends on the clip only      -> `vp` (total_T, d), one small GEMM per call
//     W1 A[t] = sum_l sim_r[l][t] (W1 query_l)  is a rank-3 combination  -> `qu` (3 n, d) = query rows through W1, one small GEMM per call
// The other two (video * A, video * B) are built in registers block by block, split into the two f16 planes of the encoder kernels and
// multiplied on the matrix cores into one pair of accumulators; the output row = that + vp[t] + sum_l sim_r[l][t] qu[l] leaves through the
// tile as whole rows.  The fusion tensor never exists in memory.  Range flag and exact fallback as the encoder kernels.
struct CoattArgs {
    const float* video; const float* proj; const float* query; float* out; int T; const int32_t* q_vid; const int32_t* vid_row_off;
    const int32_t* q_row_off; const a16x8* frag; const int* frag_bad; int* act_bad; int run_id; const float* w;
    const float* vp; const float* qu;                     // (total_T, d) = video @ W0^T + bias; (3 n, d) = query @ W1^T
};

__device__ __forceinline__ float ca_block_reduce(float v, bool is_max, float* red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float u = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, u) : v + u;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// vq_fc.weight (128, 512) as four K chunks in the fragment order of grd_enc_prep_kernel: (chunk, wave, k16 step, plane) x 64 lanes x 16 bytes
__global__ __launch_bounds__(64) void grd_vq_prep_kernel(const float* __restrict__ W, a16x8* frag, int* frag_bad) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    const int s = blockIdx.x & 7, wave = (blockIdx.x >> 3) & 3, chunk = blockIdx.x >> 5;
    const int n = wave * 32 + l31;
    float x8[8];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x8[j] = W[(size_t)n * (4 * EC_D) + chunk * EC_D + 16 * s + 8 * hi + j]; bad = bad || !(fabsf(x8[j]) < EC_LIM); }
    if (bad) atomicOr(frag_bad, 1);
    union { bigc_fp16x2 p[4]; a16x8 v; } hh, rr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float u = x8[2 * j], v = x8[2 * j + 1];
        hh.p[j] = bigc_pk_rtn(u, v);
        rr.p[j] = bigc_pk_rtn((u - (float)hh.p[j][0]) * EC_RS, (v - (float)hh.p[j][1]) * EC_RS);
    }
    a16x8* dst = frag + ((((long)chunk * 4 + wave) * 8 + s) * 2) * 64 + lane;
    dst[0] = hh.v;
    dst[64] = rr.v;
}

template <bool EXACT>
__global__ __launch_bounds__(256, 2) void grd_coatt_fc_kernel(const CoattArgs a) {
    if (EXACT && a.act_bad[0] != a.run_id && a.frag_bad[0] == 0) return;
    constexpr int d = EC_D;
    extern __shared__ __attribute__((aligned(16))) unsigned char ca_lds[];
    unsigned char* Ah = ca_lds;                               // the two split planes of a 64 x 128 operand piece; EXACT: the piece in f32; then the output tile
    unsigned char* Ar = ca_lds + EC_CR * EC_AP;
    float* Xt = (float*)ca_lds;                               // (64, EC_XP) f32: 33.8 KB of the planes' 34.8 KB
    float* q = (float*)(ca_lds + 2 * EC_CR * EC_AP);          // 3 d
    float* s2 = q + 3 * d;                                    // 3 d
    float* qu = s2 + 3 * d;                                   // 3 d: this query's rows of `qu`
    float* red = qu + 3 * d;                                  // 4
    float* sim_r = red + 4;                                   // 3 T
    const float* video = a.video;
    const float* proj = a.proj;
    const float* vp = a.vp;
    int T = a.T;
    size_t out_row = (size_t)blockIdx.x * T;
    if (a.q_vid) {
        const int v = a.q_vid[blockIdx.x], r0 = a.vid_row_off[v];
        T = a.vid_row_off[v + 1] - r0;
        video += (size_t)r0 * d;
        proj += (size_t)r0 * d;
        vp += (size_t)r0 * d;
        out_row = a.q_row_off[blockIdx.x];
    }
    float* sim_c = sim_r + 3 * T;                             // 3 T
    const int i = blockIdx.x, tid = threadIdx.x;
    // ---- scores and their two softmaxes: coattention_kernel's statements (csrc/grounding.hip)
    for (int c = tid; c < 3 * d; c += 256) { q[c] = a.query[(size_t)i * 3 * d + c]; qu[c] = a.qu[(size_t)i * 3 * d + c]; }
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        const f32x4* pr = (const f32x4*)(proj + (size_t)t * d);
        float s[3] = {0.f, 0.f, 0.f};
#pragma unroll 4
        for (int c4 = 0; c4 < d / 4; ++c4) {
            const f32x4 p = pr[c4];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const f32x4 qq = *(const f32x4*)(q + l * d + c4 * 4);
                s[l] += p[0] * qq[0] + p[1] * qq[1] + p[2] * qq[2] + p[3] * qq[3];
            }
        }
        const float m = fmaxf(s[0], fmaxf(s[1], s[2]));
        const float e0 = expf(s[0] - m), e1 = expf(s[1] - m), e2 = expf(s[2] - m);
        const float sum = e0 + e1 + e2;
        sim_r[t] = e0 / sum; sim_r[T + t] = e1 / sum; sim_r[2 * T + t] = e2 / sum;
        sim_c[t] = s[0]; sim_c[T + t] = s[1]; sim_c[2 * T + t] = s[2];
    }
    __syncthreads();
    for (int l = 0; l < 3; ++l) {
        float m = -INFINITY;
        for (int t = tid; t < T; t += 256) m = fmaxf(m, sim_c[l * T + t]);
        m = ca_block_reduce(m, true, red);
        float sum = 0.f;
        for (int t = tid; t < T; t += 256) {
            const float e = expf(sim_c[l * T + t] - m);
            sim_c[l * T + t] = e;
            sum += e;
        }
        sum = ca_block_reduce(sum, false, red);
        for (int t = tid; t < T; t += 256) sim_c[l * T + t] /= sum;
    }
    __syncthreads();
    const int cq = tid & 31, rg = tid >> 5;                   // channels 4 cq .. + 3 of rows rg, rg + 8, ...
    {   // S2 = sim_c^T @ video (3, d): eight partial sums per element (clips rg, rg + 8, ...: a 512-byte row per 32 lanes and load), added in
        // a fixed order.  (coattention_kernel walks the T clips of an element one after the other: a chain of T L2 round trips that took half
        // of its time; the sum's order differs from it, the values agree to f32 rounding.)
        f32x4 part[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 4
        for (int t = rg; t < T; t += 8) {
            const f32x4 v = *(const f32x4*)(video + (size_t)t * d + 4 * cq);
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const float w = sim_c[l * T + t];
#pragma unroll
                for (int c = 0; c < 4; ++c) part[l][c] = fmaf(w, v[c], part[l][c]);
            }
        }
        float* ps = (float*)ca_lds;                           // (8, 3 d) in the operand planes, which are not in use yet
#pragma unroll
        for (int l = 0; l < 3; ++l) *(f32x4*)(ps + (rg * 3 + l) * d + 4 * cq) = part[l];
        __syncthreads();
        for (int e = tid; e < 3 * d; e += 256) {
            float acc = ps[e];
#pragma unroll
            for (int g = 1; g < 8; ++g) acc += ps[g * 3 * d + e];
            s2[e] = acc;
        }
    }
    __syncthreads();

    // ---- the fusion rows, 64 clips at a time, straight into vq_fc
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    bool bad = false;
    float* ob = a.out + out_row * d;
    const f32x4 u0 = *(const f32x4*)(qu + 4 * cq), u1 = *(const f32x4*)(qu + d + 4 * cq), u2 = *(const f32x4*)(qu + 2 * d + 4 * cq);
#pragma unroll 1
    for (int t0 = 0; t0 < T; t0 += EC_CR) {
        f32x16 acc0[2], acc1[2];
        float ex[EXACT ? 8 : 1][4];                           // EXACT: thread (row tid & 63, features 4 (tid >> 6) + 16 j .. + 3)
        if (EXACT) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) ex[j][c] = 0.f;
        }
        if (!EXACT) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[rb][r] = 0.f; acc1[rb][r] = 0.f; }
        }
        f32x4 vrow[8];                                        // this thread's video values of the block: the same for the four pieces
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int t = t0 + rg + 8 * rr;
            vrow[rr] = t < T ? *(const f32x4*)(video + (size_t)t * d + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int piece = 2; piece < 4; ++piece) {             // (K chunks 0 and 1 of vq_fc: see vp / qu)
            a16x8 w[16];
            if (!EXACT) ec_load_w(a.frag, piece, wave, lane, w);
            const float* z = piece == 3 ? s2 : q;             // video * A combines the query rows, video * B the S2 rows
            const f32x4 z0 = *(const f32x4*)(z + 4 * cq), z1 = *(const f32x4*)(z + d + 4 * cq), z2 = *(const f32x4*)(z + 2 * d + 4 * cq);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int row = rg + 8 * rr, t = t0 + row;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (t < T) {
                    const f32x4 v = vrow[rr];
                    const float r0 = sim_r[t], r1 = sim_r[T + t], r2 = sim_r[2 * T + t];
                    // (coattention_kernel's expressions: a = r0 q0 + r1 q1 + r2 q2, b likewise on S2, then a v and b v)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float ab = r0 * z0[c] + r1 * z1[c] + r2 * z2[c];
                        o[c] = ab * v[c];
                    }
                }
                if (!EXACT) {
                    bad = bad || ec_bad4(o);
                    ec_store_split(o, Ah, Ar, row, cq);
                } else {
                    *(f32x4*)((float*)ca_lds + row * EC_XP + 4 * cq) = o;
                }
            }
            __syncthreads();
            if (!EXACT) {
                ec_mm<2, false>(Ah, Ar, w, 0, l31, hi, acc0, acc1);
            } else {
                const float* Dw = (const float*)ca_lds;
                const int row = tid & 63;
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float* wr = a.w + (size_t)(4 * (tid >> 6) + 16 * j + c) * (4 * d) + piece * d;
                        float acc = ex[j][c];
#pragma unroll 4
                        for (int k = 0; k < d; ++k) acc = fmaf(Dw[row * EC_XP + k], wr[k], acc);
                        ex[j][c] = acc;
                    }
            }
            __syncthreads();
        }
        // ---- through the tile, + the clip's and the query's terms, out by whole rows
        if (!EXACT) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int nn0 = wave * 32 + 8 * r4 + 4 * hi;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = acc0[rb][4 * r4 + c] + acc1[rb][4 * r4 + c] * EC_RI;
                    *(f32x4*)(Xt + (rb * 32 + l31) * EC_XP + nn0) = v;
                }
            }
        } else {
            const int row = tid & 63;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int nn0 = 4 * (tid >> 6) + 16 * j;
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = ex[j][c];
                *(f32x4*)(Xt + row * EC_XP + nn0) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = rg + 8 * rr, t = t0 + row;
            if (t < T) {
                const f32x4 x = *(const f32x4*)(Xt + row * EC_XP + 4 * cq), pv = *(const f32x4*)(vp + (size_t)t * d + 4 * cq);
                const float r0 = sim_r[t], r1 = sim_r[T + t], r2 = sim_r[2 * T + t];
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (x[c] + pv[c]) + (r0 * u0[c] + r1 * u1[c] + r2 * u2[c]);
                *(f32x4*)(ob + (size_t)t * d + 4 * cq) = o;
            }
        }
        __syncthreads();
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
    for (int i = 0; i < 3; ++i) { a.pw_w[6 + i] = E.in_proj.w + (size_t)i * EC_D * EC_D; a.pw_b[6 + i] = E.in_proj.b + i * EC_D; }
    a.qkv_out = nullptr;
}
}  // namespace

// `frag`: grd_enc_frag_bytes() of device memory owned by the model for this encoder; `refresh`: rebuild it (the weights changed since)
int grd_enc_chain(hipStream_t s, const Enc& E, const float* x, const float* pos, long rows, int T, const int2* info, void* frag, bool refresh,
                  float* out, float* ln_out, float* qkv_out) {
    if (rows <= 0) return BIGC_OK;
    EncChainArgs a;
    enc_args(E, rows, T, info, frag, a);
    a.x = x; a.pos = pos; a.res = nullptr; a.out = out; a.ln_out = ln_out; a.qkv_out = qkv_out;
    if (refresh) {                             // (also the first use of the buffer: both flag words start at zero)
        if (hipMemsetAsync((void*)a.frag_bad, 0, 2 * sizeof(int), s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "grd_enc_chain: memset failed");
        hipLaunchKernelGGL(grd_enc_prep_kernel, dim3(EC_LAYERS * 32 + 4), dim3(64), 0, s, a, (a16x8*)frag, (int*)a.frag_bad);
        BIGC_LAUNCH_CHECK("grd_enc_prep");
    }
    const int tr = EC_CR - 8 * (E.ks / 2);
    const long tiles = (rows + tr - 1) / tr;                 // 64-row tiles (the gated f32 instantiation always runs on these)
    ProfScope ps((void*)s, FAM_GROUNDING, 2.0 * rows * EC_D * (qkv_out ? 7.0 : 4.0) * EC_D, 4.0 * rows * (qkv_out ? 5.0 : 3.0) * EC_D);
    static const int waves = getenv("BIGC_ENC_WAVES") ? atoi(getenv("BIGC_ENC_WAVES")) : 8;
    // k = 7, launches of several 128-row tiles per CU: sixteen waves on 128 rows (104 of them exact instead of 40 of 64)
    static const int big_env = getenv("BIGC_ENC_TILE128") ? atoi(getenv("BIGC_ENC_TILE128")) : -1;
    const int tr128 = 128 - 8 * (E.ks / 2);
    const long tiles128 = (rows + tr128 - 1) / tr128;
    const bool big = E.ks == 7 && waves == 8 && (big_env >= 0 ? big_env != 0 : tiles128 >= 1024);
    constexpr size_t lds64 = sizeof(EcSmemT<64>), lds128 = sizeof(EcSmemT<128>);
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)grd_enc_chain_kernel<7, false, 16, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        (void)hipFuncSetAttribute((const void*)grd_enc_chain_kernel<7, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
        (void)hipFuncSetAttribute((const void*)grd_enc_chain_kernel<7, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
        (void)hipFuncSetAttribute((const void*)grd_enc_chain_kernel<7, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
        (void)hipFuncSetAttribute((const void*)grd_enc_chain_kernel<3, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
        (void)hipFuncSetAttribute((const void*)grd_enc_chain_kernel<3, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
        (void)hipFuncSetAttribute((const void*)grd_enc_chain_kernel<3, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
    });
    if (E.ks == 7) {
        if (big) hipLaunchKernelGGL((grd_enc_chain_kernel<7, false, 16, 128>), dim3((unsigned)tiles128), dim3(1024), lds128, s, a);
        else if (waves == 8) hipLaunchKernelGGL((grd_enc_chain_kernel<7, false, 8>), dim3((unsigned)tiles), dim3(512), lds64, s, a);
        else hipLaunchKernelGGL((grd_enc_chain_kernel<7, false, 4>), dim3((unsigned)tiles), dim3(256), lds64, s, a);
        hipLaunchKernelGGL((grd_enc_chain_kernel<7, true, 4>), dim3((unsigned)tiles), dim3(256), lds64, s, a);     // gated: returns at once when no flag is up
    } else {
        if (waves == 8) hipLaunchKernelGGL((grd_enc_chain_kernel<3, false, 8>), dim3((unsigned)tiles), dim3(512), lds64, s, a);
        else hipLaunchKernelGGL((grd_enc_chain_kernel<3, false, 4>), dim3((unsigned)tiles), dim3(256), lds64, s, a);
        hipLaunchKernelGGL((grd_enc_chain_kernel<3, true, 4>), dim3((unsigned)tiles), dim3(256), lds64, s, a);
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

bool grd_coatt_fc_supported(int d, int max_T) {
    static const bool off = getenv("BIGC_COATT_FC") && atoi(getenv("BIGC_COATT_FC")) == 0;
    return !off && d == EC_D && max_T <= 2048 && !bigc_linear_get_exact() && !bigc_linear_get_fast();
}

size_t grd_vq_frag_bytes() { return (size_t)4 * 4 * 8 * 2 * 64 * sizeof(a16x8) + 256; }

// out (rows, 128) = vq_fc(fusion rows of the co-attention); one video (q_vid == nullptr: n queries x T clips) or several (ragged descriptors)
int grd_coatt_fc(hipStream_t s, const float* video, const float* proj, const float* query, const Lin& vq_fc, const float* vp, const float* qu,
                 float* out, int n, int T, int max_T, long total_rows, const int32_t* q_vid, const int32_t* vid_row_off, const int32_t* q_row_off,
                 void* frag, bool refresh) {
    if (n <= 0) return BIGC_OK;
    CoattArgs a;
    a.video = video; a.proj = proj; a.query = query; a.out = out; a.T = T; a.q_vid = q_vid; a.vid_row_off = vid_row_off; a.q_row_off = q_row_off;
    a.frag = (const a16x8*)frag; a.w = vq_fc.w; a.vp = vp; a.qu = qu;
    int* flags = (int*)((char*)frag + (size_t)4 * 4 * 8 * 2 * 64 * sizeof(a16x8));
    a.frag_bad = flags; a.act_bad = flags + 1;
    static std::atomic<int> g_run{0};
    a.run_id = g_run.fetch_add(1, std::memory_order_relaxed) + 1;
    if (refresh) {
        if (hipMemsetAsync((void*)flags, 0, 2 * sizeof(int), s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "grd_coatt_fc: memset failed");
        hipLaunchKernelGGL(grd_vq_prep_kernel, dim3(4 * 32), dim3(64), 0, s, vq_fc.w, (a16x8*)frag, flags);
        BIGC_LAUNCH_CHECK("grd_vq_prep");
    }
    const size_t lds = (size_t)2 * EC_CR * EC_AP + sizeof(float) * (9 * (size_t)EC_D + 4 + 6 * (size_t)max_T);
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)grd_coatt_fc_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EC_CR * EC_AP + 4 * (9 * EC_D + 4 + 6 * 2048));
        (void)hipFuncSetAttribute((const void*)grd_coatt_fc_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * EC_CR * EC_AP + 4 * (9 * EC_D + 4 + 6 * 2048));
    });
    ProfScope ps((void*)s, FAM_GROUNDING, 2.0 * total_rows * 2.0 * EC_D * EC_D, 4.0 * total_rows * 2.0 * EC_D);
    hipLaunchKernelGGL(grd_coatt_fc_kernel<false>, dim3(n), dim3(256), lds, s, a);
    hipLaunchKernelGGL(grd_coatt_fc_kernel<true>, dim3(n), dim3(256), lds, s, a);      // gated: returns at once when no flag is up
    BIGC_LAUNCH_CHECK("grd_coatt_fc");
    return BIGC_OK;
}
