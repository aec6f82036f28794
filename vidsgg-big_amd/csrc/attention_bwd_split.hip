// Backward of the attention core (gradients of softmax(q k^T / sqrt(hd)) v, torch autograd of nn.MultiheadAttention's core) for
// head_dim 64 on the f16 matrix cores with exactly split f32 operands -- the scheme of attention_split.hip / linear_split.hip: every
// f32 operand x = h + r 2^-11, three v_mfma_f32_32x32x16_f16 per product in two accumulators.  The f32-MFMA kernels of
// attention_bwd.hip spend 96 v_mfma_f32_32x32x2f32 per (query tile, key tile) pair; these spend 36 MFMAs of half the cycles each.
// They stay as the exact mode, for the other head dims, and as the range fallback: an operand of magnitude >= 32768 (or NaN / Inf)
// raises the launch's flag word and the f32 kernels, launched right behind and gated on it, redo the work.
//
//   mha_bwd_dq_h_kernel    one wave per (sequence, head, 32-query tile), loop over the key tiles:
//        S^T = K Q^T, dP^T = V dO^T   (lane = query column, registers = keys)     p = exp(s - lse),  dS = p (dP - delta)
//        dQ += dS K                   (dS, in A-operand layout as it stands, split on the fly)
//   mha_bwd_dkv_h_kernel   one wave per (sequence, head, 32-key tile), loop over the query tiles:
//        S = Q K^T, dP = dO V^T       (lane = key column, registers = queries)
//        dV += P^T dO,  dK += dS^T Q
// Dropout on the probabilities (training mode) as in the f32 kernels: the mask is regenerated from (seed, site, index).
#include "common.h"
#include "prof.h"
#include <atomic>
#include <mutex>
#include <stdlib.h>

typedef _Float16 a16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr float RS = 2048.0f, RI = 1.0f / 2048.0f;
constexpr int HD = 64, NST = 4, NT = 2;

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

__device__ __forceinline__ void split8(const float (&x)[8], a16x8& h, a16x8& r, float& chk, float& amax) {
    union { bigc_fp16x2 p[4]; a16x8 v; } hh, rr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        hh.p[j] = bigc_pk_rtn(a, b);
        const float ra = (a - (float)hh.p[j][0]) * RS, rb = (b - (float)hh.p[j][1]) * RS;
        rr.p[j] = bigc_pk_rtn(ra, rb);
        chk += ra + rb;
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(amax) : "v"(amax), "v"(a), "v"(b));
    }
    h = hh.v;
    r = rr.v;
}

// The split is exact down to an ABSOLUTE floor (2^-35), and d scores = p (dP - delta) cancels: under a sharp softmax dP of the winning
// key equals delta up to rounding, so an error of eps |dP| in dP is an error of eps / p_other in d scores.  The upstream gradient dO of an
// encoder layer can be 1e-7 and smaller (eps = 2^-35 / |dO| = 3e-4: measured as 19 % wrong conv-weight gradients in the second training
// step of tests/test_hip_train.py).  So dO is brought to a magnitude of [8, 16) by a power of two taken from its largest entry (found by
// mha_delta_kernel, which reads all of dO anyway): everything below is linear in dO, the outputs are scaled back exactly.
__device__ __forceinline__ void grad_scale(const float* amax_slot, float& gs, float& gi) {
    const float am = *amax_slot;
    gs = 1.0f;
    gi = 1.0f;
    if (am > 0.f && am < INFINITY) {
        int e;
        frexpf(am, &e);                                  // am = f 2^e, f in [0.5, 1)
        const int sh = max(-100, min(100, 4 - e));
        gs = ldexpf(1.0f, sh);
        gi = ldexpf(1.0f, -sh);
    }
}

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// this lane's row of a (rows, ld) operand: the four 8-float chunks at head dims 16 st + 8 hi, times `mul`, split
__device__ __forceinline__ void load_row_split(const float* p, float mul, a16x8 (&h)[NST], a16x8 (&r)[NST], float& chk, float& amax) {
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const f32x4 t0 = *(const f32x4*)(p + 16 * st), t1 = *(const f32x4*)(p + 16 * st + 4);
        const float x[8] = {t0[0] * mul, t0[1] * mul, t0[2] * mul, t0[3] * mul, t1[0] * mul, t1[1] * mul, t1[2] * mul, t1[3] * mul};
        split8(x, h[st], r[st], chk, amax);
    }
}

// C^T tile = A_rows B_rows^T over the head dimension: A rows are this tile's rows (operand A of the MFMA, one row per lane), B the
// fixed rows held split in registers
__device__ __forceinline__ void dot_tile(const float* arow, float mul, const a16x8 (&bh)[NST], const a16x8 (&br)[NST], f32x16& out, float& chk, float& amax) {
    f32x16 c0 = zero16(), c1 = zero16();
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const f32x4 t0 = *(const f32x4*)(arow + 16 * st), t1 = *(const f32x4*)(arow + 16 * st + 4);
        const float x[8] = {t0[0] * mul, t0[1] * mul, t0[2] * mul, t0[3] * mul, t1[0] * mul, t1[1] * mul, t1[2] * mul, t1[3] * mul};
        a16x8 ah, ar;
        split8(x, ah, ar, chk, amax);
        c0 = MFMA_H(ah, bh[st], c0);
        c1 = MFMA_H(ah, br[st], c1);
        c1 = MFMA_H(ar, bh[st], c1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = c0[r] + c1[r] * RI;
}

// acc[t] += W X with W this lane's 16 register values (rows mfma_row(r, hi) of the contraction, in A-operand layout as they stand) and
// X (rows of the contraction x 64 head columns) read from memory: column NT c + t of row mfma_row(r, hi)
__device__ __forceinline__ void accumulate(const float (&w)[16], const float* xbase, int ldx, float xmul, int row0, int n_rows, int l31, int hi,
                                           f32x16 (&o0)[NT], f32x16 (&o1)[NT], float& chk, float& amax) {
    struct xrow { float x[NT]; };
    xrow xv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) xv[r] = *(const xrow*)(xbase + (size_t)min(row0 + mfma_row(r, hi), n_rows - 1) * ldx + NT * l31);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float w8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w8[j] = w[8 * u + j];
        a16x8 wh, wr;
        split8(w8, wh, wr, chk, amax);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float x8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x8[j] = xv[8 * u + j].x[t] * xmul;
            a16x8 xh, xr;
            split8(x8, xh, xr, chk, amax);
            o0[t] = MFMA_H(wh, xh, o0[t]);
            o1[t] = MFMA_H(wh, xr, o1[t]);
            o1[t] = MFMA_H(wr, xh, o1[t]);
        }
    }
}

__global__ __launch_bounds__(256) void mha_bwd_dq_h_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                           const float* __restrict__ v, int ldv, const float* __restrict__ dO, int lddo,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dq, int lddq, const int32_t* __restrict__ q_off,
                                                           const int32_t* __restrict__ k_off, int n_seq, int n_qt, int n_head, float scale,
                                                           const DropSpec drop, int* __restrict__ flag, int run_id, const float* __restrict__ amax_slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)n_seq * n_qt * n_head) return;
    float gs, gi;
    grad_scale(amax_slot, gs, gi);
    const int h = item % n_head;
    const int qt = (item / n_head) % n_qt;
    const int s = item / ((long)n_head * n_qt);
    const int q0 = q_off[s], nq = q_off[s + 1] - q0;
    const int k0 = k_off[s], nk = k_off[s + 1] - k0;
    if (qt * 32 >= nq || nk <= 0) return;
    const int qrow = min(qt * 32 + l31, nq - 1);
    const bool qok = qt * 32 + l31 < nq;
    float chk = 0.f, amax = 0.f;
    a16x8 qh[NST], qr[NST], gh[NST], gr[NST];
    load_row_split(q + (size_t)(q0 + qrow) * ldq + h * HD + 8 * hi, scale, qh, qr, chk, amax);
    load_row_split(dO + (size_t)(q0 + qrow) * lddo + h * HD + 8 * hi, gs, gh, gr, chk, amax);
    const float my_lse = lse[(size_t)(q0 + qrow) * n_head + h], my_delta = delta[(size_t)(q0 + qrow) * n_head + h] * gs;
    f32x16 o0[NT], o1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { o0[t] = zero16(); o1[t] = zero16(); }
    const int n_kt = (nk + 31) / 32;
    const float* kbase = k + (size_t)k0 * ldk + h * HD;
    const float* vbase = v + (size_t)k0 * ldv + h * HD;
    for (int kt = 0; kt < n_kt; ++kt) {
        const int krow = min(kt * 32 + l31, nk - 1);
        f32x16 sc, dp;
        dot_tile(kbase + (size_t)krow * ldk + 8 * hi, 1.0f, qh, qr, sc, chk, amax);
        dot_tile(vbase + (size_t)krow * ldv + 8 * hi, 1.0f, gh, gr, dp, chk, amax);
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + mfma_row(r, hi);
            const bool ok = key < nk && qok;
            const float p = ok ? expf(sc[r] - my_lse) : 0.f;
            float dpr = dp[r];
            if (drop.thresh) dpr = drop_keep(drop, drop_att_idx(q0 + qrow, n_head, h, key)) ? dpr * drop.scale : 0.f;
            ds[r] = p * (dpr - my_delta);
        }
        accumulate(ds, kbase, ldk, 1.0f, kt * 32, nk, l31, hi, o0, o1, chk, amax);
    }
    if (!(fabsf(chk) < INFINITY) || amax >= 32768.0f) atomicMax(flag, run_id);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = qt * 32 + mfma_row(r, hi);
        if (row < nq) {
            float* dst = dq + (size_t)(q0 + row) * lddq + h * HD + NT * l31;
#pragma unroll
            for (int t = 0; t < NT; ++t) dst[t] = (o0[t][r] + o1[t][r] * RI) * scale * gi;
        }
    }
}

__global__ __launch_bounds__(256) void mha_bwd_dkv_h_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                            const float* __restrict__ v, int ldv, const float* __restrict__ dO, int lddo,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv,
                                                            const int32_t* __restrict__ q_off, const int32_t* __restrict__ k_off,
                                                            int n_seq, int n_kt, int n_head, float scale, const DropSpec drop,
                                                            int* __restrict__ flag, int run_id, const float* __restrict__ amax_slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)n_seq * n_kt * n_head) return;
    float gs, gi;
    grad_scale(amax_slot, gs, gi);
    const int h = item % n_head;
    const int kt = (item / n_head) % n_kt;
    const int s = item / ((long)n_head * n_kt);
    const int q0 = q_off[s], nq = q_off[s + 1] - q0;
    const int k0 = k_off[s], nk = k_off[s + 1] - k0;
    if (kt * 32 >= nk) return;
    const int krow = min(kt * 32 + l31, nk - 1);
    const bool kok = kt * 32 + l31 < nk;
    float chk = 0.f, amax = 0.f;
    a16x8 kh[NST], kr[NST], vh[NST], vr[NST];
    load_row_split(k + (size_t)(k0 + krow) * ldk + h * HD + 8 * hi, scale, kh, kr, chk, amax);       // (the 1 / sqrt(hd) of the scores rides on K here)
    load_row_split(v + (size_t)(k0 + krow) * ldv + h * HD + 8 * hi, 1.0f, vh, vr, chk, amax);
    f32x16 ak0[NT], ak1[NT], av0[NT], av1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { ak0[t] = zero16(); ak1[t] = zero16(); av0[t] = zero16(); av1[t] = zero16(); }
    const int n_qt = (nq + 31) / 32;
    const float* qbase = q + (size_t)q0 * ldq + h * HD;
    const float* gbase = dO + (size_t)q0 * lddo + h * HD;
    for (int qt = 0; qt < n_qt; ++qt) {
        const int qrow = min(qt * 32 + l31, nq - 1);
        f32x16 sc, dp;
        dot_tile(qbase + (size_t)qrow * ldq + 8 * hi, 1.0f, kh, kr, sc, chk, amax);           // S[query][key]: lane = key column
        dot_tile(gbase + (size_t)qrow * lddo + 8 * hi, gs, vh, vr, dp, chk, amax);
        float pd[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = qt * 32 + mfma_row(r, hi);
            const int qc = min(qi, nq - 1);
            const bool ok = qi < nq && kok;
            const float lr = lse[(size_t)(q0 + qc) * n_head + h], dr = delta[(size_t)(q0 + qc) * n_head + h] * gs;
            const float p = ok ? expf(sc[r] - lr) : 0.f;
            float pdr = p, dpr = dp[r];
            if (drop.thresh) {
                const bool keep = drop_keep(drop, drop_att_idx(q0 + qc, n_head, h, kt * 32 + l31));
                pdr = keep ? p * drop.scale : 0.f;
                dpr = keep ? dpr * drop.scale : 0.f;
            }
            pd[r] = pdr;
            ds[r] = p * (dpr - dr);
        }
        accumulate(pd, gbase, lddo, gs, qt * 32, nq, l31, hi, av0, av1, chk, amax);             // dV += P^T dO
        accumulate(ds, qbase, ldq, 1.0f, qt * 32, nq, l31, hi, ak0, ak1, chk, amax);              // dK += dS^T Q
    }
    if (!(fabsf(chk) < INFINITY) || amax >= 32768.0f) atomicMax(flag, run_id);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = kt * 32 + mfma_row(r, hi);
        if (row < nk) {
            float* kd = dk + (size_t)(k0 + row) * lddk + h * HD + NT * l31;
            float* vd = dv + (size_t)(k0 + row) * lddv + h * HD + NT * l31;
#pragma unroll
            for (int t = 0; t < NT; ++t) { kd[t] = (ak0[t][r] + ak1[t][r] * RI) * scale * gi; vd[t] = (av0[t][r] + av1[t][r] * RI) * gi; }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 6: dQ, dK and dV in ONE launch, the shared operand of a step split ONCE per workgroup into LDS.
//
// The two kernels above give one wave a tile of its own rows and let it walk the other side's tiles alone: every wave loads and splits
// the same K / V / Q / dO rows again (272 split elements per lane and tile pair beside 84 MFMAs), and the walk is a chain of dependent
// global loads (35 + 60 us per decoder layer for 8 videos -- the same for the encoder's 64-row sequences: latency, not work).  Here a
// workgroup is (sequence, head, side, group of <= 4 own tiles): side KV owns key tiles (dK, dV), side Q owns query tiles (dQ); the
// workgroups of both sides run side by side in the same launch.  Per step the whole workgroup stages ONE tile of the other side --
// 32 rows of the two operands it multiplies with -- into LDS in the split form, in the two layouts the MFMAs read:
//   R (row-major: 16-byte chunks of 8 head dims, a row every 272 bytes) -> A fragments of the score / dP products,
//   T (transposed: per head dim the tile's 32 rows in the register order of an accumulator, 8 per 16-byte chunk, a column every 144
//      bytes) -> B fragments of the products that contract over the tile's rows (P^T dO, dS^T Q, dS K),
// both pitches conflict-free for ds_read_b128.  The next tile's global loads are issued before the current tile's MFMAs and split into
// the other LDS buffer behind them: one barrier per step.  Per lane and tile pair 48 elements are split on the fly (P, dS) instead of 272.
// Same arithmetic per product as above (three f16 MFMAs, two accumulators), same dropout masks, same range flag.
constexpr float LOG2E = 1.4426950408889634f;
constexpr int RP = 272, TP = 144;                       // bytes per row of R / per column of T
constexpr int LDS_R = 32 * RP, LDS_T = HD * TP;         // 8704, 9216
constexpr int LDS_BUF = 2 * LDS_R + 2 * LDS_T + 256;    // [R1 | R2 | T1 | T2 | lse (32 f32) | delta (32 f32)] = 36096 bytes; two of them

struct FusedArgs {
    const float *q, *k, *v, *dO, *lse, *delta;
    float *dq, *dk, *dv;
    int ldq, ldk, ldv, lddo, lddq, lddk, lddv;
    const int32_t *q_off, *k_off;
    int n_head, gq_groups, gk_groups;
    float scale;
    DropSpec drop;
    int* flag; int run_id; const float* amax_slot;
};

template <int NW, bool KV>
__device__ __forceinline__ void bwd_side(const FusedArgs& a, unsigned char* lds, const int s, const int h, const int group, const float gs, const float gi) {
    constexpr int NTH = 64 * NW;
    constexpr int N_ITEMS = 256;                          // (operand 1 | operand 2) x 16 row pairs x 8 chunks of 8 head dims
    constexpr int ITERS = (N_ITEMS + NTH - 1) / NTH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int q0 = a.q_off[s], nq = a.q_off[s + 1] - q0;
    const int k0 = a.k_off[s], nk = a.k_off[s + 1] - k0;
    const int n_own = KV ? nk : nq, n_oth = KV ? nq : nk;
    if (n_own <= 0 || n_oth <= 0 || group * NW * 32 >= n_own) return;          // (uniform over the workgroup)
    const int ot = group * NW + wave;
    const bool own_valid = ot * 32 < n_own;                                     // (uniform over the wave)
    const int orow = min(ot * 32 + l31, n_own - 1);
    const bool rok = ot * 32 + l31 < n_own;
    // the operands of the other side: X1 meets own1 in the scores, X2 meets own2 in dP
    const float* x1 = KV ? a.q + (size_t)q0 * a.ldq + h * HD : a.k + (size_t)k0 * a.ldk + h * HD;
    const float* x2 = KV ? a.dO + (size_t)q0 * a.lddo + h * HD : a.v + (size_t)k0 * a.ldv + h * HD;
    const int ld1 = KV ? a.ldq : a.ldk, ld2 = KV ? a.lddo : a.ldv;
    const float mul2 = KV ? gs : 1.0f;
    float chk = 0.f, amax = 0.f;
    a16x8 o1h[NST], o1r[NST], o2h[NST], o2r[NST];
    {
        const float* p1 = KV ? a.k + (size_t)(k0 + orow) * a.ldk + h * HD + 8 * hi : a.q + (size_t)(q0 + orow) * a.ldq + h * HD + 8 * hi;
        const float* p2 = KV ? a.v + (size_t)(k0 + orow) * a.ldv + h * HD + 8 * hi : a.dO + (size_t)(q0 + orow) * a.lddo + h * HD + 8 * hi;
        load_row_split(p1, a.scale, o1h, o1r, chk, amax);                       // (the 1 / sqrt(hd) of the scores rides on the own side)
        load_row_split(p2, KV ? 1.0f : gs, o2h, o2r, chk, amax);
    }
    float my_lse = 0.f, my_delta = 0.f;
    if (!KV) { my_lse = a.lse[(size_t)(q0 + orow) * a.n_head + h] * LOG2E; my_delta = a.delta[(size_t)(q0 + orow) * a.n_head + h] * gs; }
    // accumulators: [0] = P^T dO (dV, KV only), [1] = dS^T Q (dK) / dS K (dQ)
    f32x16 acc0[KV ? 2 : 1][2], acc1[KV ? 2 : 1][2];
#pragma unroll
    for (int o = 0; o < (KV ? 2 : 1); ++o)
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc0[o][t] = zero16(); acc1[o][t] = zero16(); }

    // ---- staging of one tile of the other side: item = (operand, row pair e >> 3, chunk e & 7): 8 head dims of two adjacent rows, split
    // ONCE and written in both layouts -- 16 + 16 bytes per row into R, and per head dim one dword (the two rows' halves) into T, where row
    // 2 p becomes position (chunk g = 2 (row >> 4) + ((row >> 2) & 1), slot j = (row & 3) + 4 ((row >> 3) & 1)) and row 2 p + 1 slot j + 1:
    // the inverse of mfma_row(8 u + j, hi).  T2 (the dP operand transposed) is only read by the KV side (dV = P^T dO).
    float pre[ITERS][16];
    auto fetch = [&](int tile) {
        const int row0 = tile * 32;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int id = tid + it * NTH;
            const int kind = id >> 7, e = id & 127;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const float* src = (kind ? x2 : x1) + (size_t)min(row0 + 2 * (e >> 3) + rr, n_oth - 1) * (kind ? ld2 : ld1) + 8 * (e & 7);
                const f32x4 t0 = *(const f32x4*)src, t1 = *(const f32x4*)(src + 4);
                pre[it][8 * rr + 0] = t0[0]; pre[it][8 * rr + 1] = t0[1]; pre[it][8 * rr + 2] = t0[2]; pre[it][8 * rr + 3] = t0[3];
                pre[it][8 * rr + 4] = t1[0]; pre[it][8 * rr + 5] = t1[1]; pre[it][8 * rr + 6] = t1[2]; pre[it][8 * rr + 7] = t1[3];
            }
        }
    };
    auto stash = [&](unsigned char* buf) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int id = tid + it * NTH;
            const int kind = id >> 7, e = id & 127;
            const float mul = kind ? mul2 : 1.0f;
            union { a16x8 v; _Float16 s[8]; } hh[2], rr[2];
            const int row = 2 * (e >> 3), c = e & 7;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = pre[it][8 * q + j] * mul;
                split8(x, hh[q].v, rr[q].v, chk, amax);
                unsigned char* dst = buf + kind * LDS_R + (row + q) * RP + c * 16;
                *(a16x8*)dst = hh[q].v;
                *(a16x8*)(dst + 128) = rr[q].v;
            }
            if (KV || kind == 0) {
                const int g = 2 * (row >> 4) + ((row >> 2) & 1), j = (row & 3) + 4 * ((row >> 3) & 1);       // (row even: j even, row + 1 -> j + 1)
                unsigned char* tp = buf + 2 * LDS_R + kind * LDS_T + (8 * c) * TP + g * 32 + j * 2;
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    *(h2*)(tp + d * TP) = h2{hh[0].s[d], hh[1].s[d]};
                    *(h2*)(tp + d * TP + 16) = h2{rr[0].s[d], rr[1].s[d]};
                }
            }
        }
    };
    auto stash_stats = [&](unsigned char* buf, int tile) {        // KV: lse / delta of the tile's 32 query rows
        if (KV && tid < 64) {
            const int r = min(tile * 32 + (tid & 31), nq - 1);
            float* dst = (float*)(buf + 2 * LDS_R + 2 * LDS_T);
            dst[tid] = tid < 32 ? a.lse[(size_t)(q0 + r) * a.n_head + h] * LOG2E : a.delta[(size_t)(q0 + r) * a.n_head + h] * gs;
        }
    };
    // C^T tile = (rows of R) x (own rows)^T over the head dimension
    auto dot_lds = [&](const unsigned char* R, const a16x8 (&bh)[NST], const a16x8 (&br)[NST], f32x16& out) {
        f32x16 c0 = zero16(), c1 = zero16();
        const unsigned char* rowp = R + l31 * RP + hi * 16;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const a16x8 ah = *(const a16x8*)(rowp + st * 32), ar = *(const a16x8*)(rowp + st * 32 + 128);
            c0 = MFMA_H(ah, bh[st], c0);
            c1 = MFMA_H(ah, br[st], c1);
            c1 = MFMA_H(ar, bh[st], c1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) out[r] = c0[r] + c1[r] * RI;
    };
    // acc[t] += W^T X: W = this lane's 16 register values (A operand as they stand), X^T from T: column 32 t + l31
    auto accum_lds = [&](const float (&w)[16], const unsigned char* T, f32x16 (&o0)[2], f32x16 (&o1)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float w8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = w[8 * u + j];
            a16x8 wh, wr;
            split8(w8, wh, wr, chk, amax);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned char* cp = T + (32 * t + l31) * TP + (2 * u + hi) * 32;
                const a16x8 xh = *(const a16x8*)cp, xr = *(const a16x8*)(cp + 16);
                o0[t] = MFMA_H(wh, xh, o0[t]);
                o1[t] = MFMA_H(wh, xr, o1[t]);
                o1[t] = MFMA_H(wr, xh, o1[t]);
            }
        }
    };

    const int n_tiles = (n_oth + 31) / 32;
    fetch(0);
    stash(lds);
    stash_stats(lds, 0);
    __syncthreads();
    for (int tile = 0; tile < n_tiles; ++tile) {
        unsigned char* cur = lds + (tile & 1) * LDS_BUF;
        unsigned char* nxt = lds + ((tile + 1) & 1) * LDS_BUF;
        const bool more = tile + 1 < n_tiles;
        if (more) fetch(tile + 1);
        if (own_valid) {
            f32x16 sc, dp;
            dot_lds(cur, o1h, o1r, sc);
            dot_lds(cur + LDS_R, o2h, o2r, dp);
            const float* stats = (const float*)(cur + 2 * LDS_R + 2 * LDS_T);
            float lrs[16], drs[16];
            if (KV) {                                                // registers 4 a .. 4 a + 3 are the tile's rows 8 a + 4 hi + (0 .. 3)
#pragma unroll
                for (int a4 = 0; a4 < 4; ++a4) {
                    const f32x4 l4 = *(const f32x4*)(stats + 8 * a4 + 4 * hi), d4 = *(const f32x4*)(stats + 32 + 8 * a4 + 4 * hi);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { lrs[4 * a4 + c] = l4[c]; drs[4 * a4 + c] = d4[c]; }
                }
            }
            float pd[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orr = mfma_row(r, hi);                    // row of the staged tile this register belongs to
                const int oi = tile * 32 + orr;
                const bool ok = oi < n_oth && rok;
                const float lr = KV ? lrs[r] : my_lse, dr = KV ? drs[r] : my_delta;
                // exp(s - lse) as 2^(s log2 e - lse log2 e): one v_exp_f32 (1 ulp) where expf is a dozen instructions; lse arrives pre-multiplied
                const float p = ok ? __builtin_amdgcn_exp2f(sc[r] * LOG2E - lr) : 0.f;
                float pdr = p, dpr = dp[r];
                if (a.drop.thresh) {
                    const int qrow_g = KV ? q0 + min(oi, nq - 1) : q0 + orow;
                    const int key = KV ? ot * 32 + l31 : oi;
                    const bool keep = drop_keep(a.drop, drop_att_idx(qrow_g, a.n_head, h, key));
                    pdr = keep ? p * a.drop.scale : 0.f;
                    dpr = keep ? dpr * a.drop.scale : 0.f;
                }
                pd[r] = pdr;
                ds[r] = p * (dpr - dr);
            }
            if (KV) {
                accum_lds(pd, cur + 2 * LDS_R + LDS_T, acc0[0], acc1[0]);                    // dV += P^T dO
                accum_lds(ds, cur + 2 * LDS_R, acc0[KV ? 1 : 0], acc1[KV ? 1 : 0]);          // dK += dS^T Q
            } else {
                accum_lds(ds, cur + 2 * LDS_R, acc0[0], acc1[0]);                            // dQ += dS K
            }
        }
        if (more) { stash(nxt); stash_stats(nxt, tile + 1); }
        __syncthreads();
    }
    if (!(fabsf(chk) < INFINITY) || amax >= 32768.0f) atomicMax(a.flag, a.run_id);
    if (!own_valid) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = ot * 32 + mfma_row(r, hi);
        if (row >= n_own) continue;
        if (KV) {
            float* vd = a.dv + (size_t)(k0 + row) * a.lddv + h * HD + l31;
            float* kd = a.dk + (size_t)(k0 + row) * a.lddk + h * HD + l31;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                vd[32 * t] = (acc0[0][t][r] + acc1[0][t][r] * RI) * gi;
                kd[32 * t] = (acc0[KV ? 1 : 0][t][r] + acc1[KV ? 1 : 0][t][r] * RI) * a.scale * gi;
            }
        } else {
            float* qd = a.dq + (size_t)(q0 + row) * a.lddq + h * HD + l31;
#pragma unroll
            for (int t = 0; t < 2; ++t) qd[32 * t] = (acc0[0][t][r] + acc1[0][t][r] * RI) * a.scale * gi;
        }
    }
}

// ---- one OUTPUT per wave (late round 6) --------------------------------------------------------------------------------------------
// The kernel above gives a wave of the KV side both outputs of its key tile (dV and dK: 128 accumulator registers, 442 - 500 registers
// in all) -- one wave per SIMD, which then sits out every LDS / barrier wait alone (rocprofv3 --pmc, profiles/r06h_pmc_mha_bwd_fused_64_videos.log:
// 48 % of the wave cycles waiting, VALU 33 %, matrix pipe 10 %).  Here a workgroup has 2 x NWT waves and every wave ONE output:
//   KV side: waves [0, NWT) own key tile g NWT + w and compute dK (scores, dP, dS, dS^T Q), waves [NWT, 2 NWT) own the same tiles and
//            compute dV (scores, P, P^T dO) -- the scores of a tile pair are computed twice, by two waves side by side;
//   Q side:  the 2 NWT waves own query tiles g 2 NWT + w and compute dQ.
// <= 256 registers per wave: two waves per SIMD.  Staging, arithmetic per product and results are those of the kernel above.
template <int NWT>
__global__ __launch_bounds__(128 * NWT, 2) void mha_bwd_roles_h_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    unsigned char* lds = fused_lds;
    constexpr int NTH = 128 * NWT;
    const int per = a.gk_groups + a.gq_groups;
    const int item = blockIdx.x / per, rr_ = blockIdx.x - item * per;
    const int h = item % a.n_head, s = item / a.n_head;
    const bool KV = rr_ < a.gk_groups;
    const int group = KV ? rr_ : rr_ - a.gk_groups;
    float gs, gi;
    grad_scale(a.amax_slot, gs, gi);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int q0 = a.q_off[s], nq = a.q_off[s + 1] - q0;
    const int k0 = a.k_off[s], nk = a.k_off[s + 1] - k0;
    const int n_own = KV ? nk : nq, n_oth = KV ? nq : nk;
    const int per_group = KV ? NWT : 2 * NWT;
    if (n_own <= 0 || n_oth <= 0 || group * per_group * 32 >= n_own) return;   // (uniform over the workgroup)
    // role of this wave: 0 = dQ, 1 = dK, 2 = dV
    const int role = KV ? (wave < NWT ? 1 : 2) : 0;
    const int ot = KV ? group * NWT + (wave < NWT ? wave : wave - NWT) : group * 2 * NWT + wave;
    const bool own_valid = ot * 32 < n_own;
    const int orow = min(ot * 32 + l31, n_own - 1);
    const bool rok = ot * 32 + l31 < n_own;
    const float* x1 = KV ? a.q + (size_t)q0 * a.ldq + h * HD : a.k + (size_t)k0 * a.ldk + h * HD;
    const float* x2 = KV ? a.dO + (size_t)q0 * a.lddo + h * HD : a.v + (size_t)k0 * a.ldv + h * HD;
    const int ld1 = KV ? a.ldq : a.ldk, ld2 = KV ? a.lddo : a.ldv;
    const float mul2 = KV ? gs : 1.0f;
    float chk = 0.f, amax = 0.f;
    a16x8 o1h[NST], o1r[NST], o2h[NST], o2r[NST];
    {
        const float* p1 = KV ? a.k + (size_t)(k0 + orow) * a.ldk + h * HD + 8 * hi : a.q + (size_t)(q0 + orow) * a.ldq + h * HD + 8 * hi;
        load_row_split(p1, a.scale, o1h, o1r, chk, amax);
        if (role != 2) {
            const float* p2 = KV ? a.v + (size_t)(k0 + orow) * a.ldv + h * HD + 8 * hi : a.dO + (size_t)(q0 + orow) * a.lddo + h * HD + 8 * hi;
            load_row_split(p2, KV ? 1.0f : gs, o2h, o2r, chk, amax);
        }
    }
    float my_lse = 0.f, my_delta = 0.f;
    if (!KV) { my_lse = a.lse[(size_t)(q0 + orow) * a.n_head + h] * LOG2E; my_delta = a.delta[(size_t)(q0 + orow) * a.n_head + h] * gs; }
    f32x16 acc0[2], acc1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { acc0[t] = zero16(); acc1[t] = zero16(); }

    // staging: 256 items (operand, row pair, chunk) by the first 256 threads -- see bwd_side
    const bool stager = tid < 256;
    float pre[16];
    const int skind = (tid >> 7) & 1, se = tid & 127;
    auto fetch = [&](int tile) {
        if (!stager) return;
        const int row0 = tile * 32;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float* src = (skind ? x2 : x1) + (size_t)min(row0 + 2 * (se >> 3) + q, n_oth - 1) * (skind ? ld2 : ld1) + 8 * (se & 7);
            const f32x4 t0 = *(const f32x4*)src, t1 = *(const f32x4*)(src + 4);
            pre[8 * q + 0] = t0[0]; pre[8 * q + 1] = t0[1]; pre[8 * q + 2] = t0[2]; pre[8 * q + 3] = t0[3];
            pre[8 * q + 4] = t1[0]; pre[8 * q + 5] = t1[1]; pre[8 * q + 6] = t1[2]; pre[8 * q + 7] = t1[3];
        }
    };
    auto stash = [&](unsigned char* buf, int tile) {
        if (stager) {
            const float mul = skind ? mul2 : 1.0f;
            union { a16x8 v; _Float16 s[8]; } hh[2], rr[2];
            const int row = 2 * (se >> 3), c = se & 7;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = pre[8 * q + j] * mul;
                split8(x, hh[q].v, rr[q].v, chk, amax);
                unsigned char* dst = buf + skind * LDS_R + (row + q) * RP + c * 16;
                *(a16x8*)dst = hh[q].v;
                *(a16x8*)(dst + 128) = rr[q].v;
            }
            if (KV || skind == 0) {
                const int g = 2 * (row >> 4) + ((row >> 2) & 1), j = (row & 3) + 4 * ((row >> 3) & 1);
                unsigned char* tp = buf + 2 * LDS_R + skind * LDS_T + (8 * c) * TP + g * 32 + j * 2;
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    *(h2*)(tp + d * TP) = h2{hh[0].s[d], hh[1].s[d]};
                    *(h2*)(tp + d * TP + 16) = h2{rr[0].s[d], rr[1].s[d]};
                }
            }
        }
        if (KV && tid >= NTH - 64) {                       // lse / delta of the tile's 32 query rows (the last wave's lanes)
            const int t64 = tid - (NTH - 64);
            const int r = min(tile * 32 + (t64 & 31), nq - 1);
            float* dst = (float*)(buf + 2 * LDS_R + 2 * LDS_T);
            dst[t64] = t64 < 32 ? a.lse[(size_t)(q0 + r) * a.n_head + h] * LOG2E : a.delta[(size_t)(q0 + r) * a.n_head + h] * gs;
        }
    };
    auto dot_lds = [&](const unsigned char* R, const a16x8 (&bh)[NST], const a16x8 (&br)[NST], f32x16& out) {
        f32x16 c0 = zero16(), c1 = zero16();
        const unsigned char* rowp = R + l31 * RP + hi * 16;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const a16x8 ah = *(const a16x8*)(rowp + st * 32), ar = *(const a16x8*)(rowp + st * 32 + 128);
            c0 = MFMA_H(ah, bh[st], c0);
            c1 = MFMA_H(ah, br[st], c1);
            c1 = MFMA_H(ar, bh[st], c1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) out[r] = c0[r] + c1[r] * RI;
    };
    auto accum_lds = [&](const float (&w)[16], const unsigned char* T) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float w8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = w[8 * u + j];
            a16x8 wh, wr;
            split8(w8, wh, wr, chk, amax);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned char* cp = T + (32 * t + l31) * TP + (2 * u + hi) * 32;
                const a16x8 xh = *(const a16x8*)cp, xr = *(const a16x8*)(cp + 16);
                acc0[t] = MFMA_H(wh, xh, acc0[t]);
                acc1[t] = MFMA_H(wh, xr, acc1[t]);
                acc1[t] = MFMA_H(wr, xh, acc1[t]);
            }
        }
    };

    const int n_tiles = (n_oth + 31) / 32;
    fetch(0);
    stash(lds, 0);
    __syncthreads();
    for (int tile = 0; tile < n_tiles; ++tile) {
        unsigned char* cur = lds + (tile & 1) * LDS_BUF;
        unsigned char* nxt = lds + ((tile + 1) & 1) * LDS_BUF;
        const bool more = tile + 1 < n_tiles;
        if (more && !own_valid) fetch(tile + 1);
        if (own_valid) {
            f32x16 sc, dp;
            dot_lds(cur, o1h, o1r, sc);
            __builtin_amdgcn_sched_barrier(0);              // (the two products one after the other: interleaved they hold 128 accumulator registers)
            if (role != 2) dot_lds(cur + LDS_R, o2h, o2r, dp);
            __builtin_amdgcn_sched_barrier(0);
            const float* stats = (const float*)(cur + 2 * LDS_R + 2 * LDS_T);
            float w[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orr = mfma_row(r, hi);
                const int oi = tile * 32 + orr;
                const bool ok = oi < n_oth && rok;
                const float lr = KV ? stats[orr] : my_lse, dr = KV ? stats[32 + orr] : my_delta;
                const float p = ok ? __builtin_amdgcn_exp2f(sc[r] * LOG2E - lr) : 0.f;
                bool keep = true;
                if (a.drop.thresh) {
                    const int qrow_g = KV ? q0 + min(oi, nq - 1) : q0 + orow;
                    const int key = KV ? ot * 32 + l31 : oi;
                    keep = drop_keep(a.drop, drop_att_idx(qrow_g, a.n_head, h, key));
                }
                if (role == 2) {
                    w[r] = (a.drop.thresh && !keep) ? 0.f : p * (a.drop.thresh ? a.drop.scale : 1.0f);       // P (dropped)
                } else {
                    const float dpr = (a.drop.thresh && !keep) ? 0.f : dp[r] * (a.drop.thresh ? a.drop.scale : 1.0f);
                    w[r] = p * (dpr - dr);                                                                       // dS
                }
            }
            // (the next tile's global loads go out here, behind the two score products: their 16 registers are not live beside the
            // products' 64 accumulator registers -- 20 spilled registers otherwise -- and still have the accumulation to hide behind)
            __builtin_amdgcn_sched_barrier(0);
            if (more) fetch(tile + 1);
            accum_lds(w, cur + 2 * LDS_R + (role == 2 ? LDS_T : 0));       // dV += P^T dO (T2)  |  dK += dS^T Q, dQ += dS K (T1)
        }
        if (more) stash(nxt, tile + 1);
        __syncthreads();
    }
    if (!(fabsf(chk) < INFINITY) || amax >= 32768.0f) atomicMax(a.flag, a.run_id);
    if (!own_valid) return;
    float* outp = role == 0 ? a.dq + (size_t)q0 * a.lddq : (role == 1 ? a.dk + (size_t)k0 * a.lddk : a.dv + (size_t)k0 * a.lddv);
    const int ldo = role == 0 ? a.lddq : (role == 1 ? a.lddk : a.lddv);
    const float osc = role == 2 ? gi : a.scale * gi;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = ot * 32 + mfma_row(r, hi);
        if (row >= n_own) continue;
        float* dst = outp + (size_t)row * ldo + h * HD + l31;
#pragma unroll
        for (int t = 0; t < 2; ++t) dst[32 * t] = (acc0[t][r] + acc1[t][r] * RI) * osc;
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void mha_bwd_fused_h_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
    const int per = a.gk_groups + a.gq_groups;
    const int item = blockIdx.x / per, r = blockIdx.x - item * per;
    const int h = item % a.n_head, s = item / a.n_head;
    float gs, gi;
    grad_scale(a.amax_slot, gs, gi);
    if (r < a.gk_groups) bwd_side<NW, true>(a, fused_lds, s, h, r, gs, gi);
    else bwd_side<NW, false>(a, fused_lds, s, h, r - a.gk_groups, gs, gi);
}

}  // namespace

namespace {
std::atomic<int> g_run{0};
constexpr int SLOTS = 1024;
std::mutex g_mu;
int* g_flags[64] = {};
// one flag word per launch (ring of SLOTS), as attention.hip's mha_flag_slot, and behind the ring one word per launch for max |dO|
int* flag_slot(int run_id) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(g_mu);
    if (!g_flags[dev]) {
        if (hipMalloc((void**)&g_flags[dev], 2 * SLOTS * sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(g_flags[dev], 0, 2 * SLOTS * sizeof(int)) != hipSuccess) return nullptr;
    }
    return g_flags[dev] + ((unsigned)run_id % SLOTS);
}
}  // namespace

// head_dim 64, step 1 (before mha_delta_kernel): 1 = the split kernels apply; the launch's flag word, run id and the (zeroed) word
// mha_delta_kernel leaves max |dO| in.  0 = not applicable (strides, BIGC_MHA_BWD_SPLIT=0), < 0 = no flag memory
int bigc_mha_bwd_h_prepare(hipStream_t s, int ldq, int ldk, int ldv, int lddo, int lddq, int lddk, int lddv, int** flag, int* run_id, float** amax_slot) {
    static const bool off = getenv("BIGC_MHA_BWD_SPLIT") && atoi(getenv("BIGC_MHA_BWD_SPLIT")) == 0;
    if (off || ldq % 4 || ldk % 4 || ldv % 4 || lddo % 4 || lddq % 2 || lddk % 2 || lddv % 2) return 0;
    const int id = g_run.fetch_add(1, std::memory_order_relaxed) + 1;
    int* f = flag_slot(id);
    if (!f) return -1;
    float* a = (float*)(f + SLOTS);
    if (hipMemsetAsync(a, 0, sizeof(float), s) != hipSuccess) return -1;
    *flag = f;
    *run_id = id;
    *amax_slot = a;
    return 1;
}

// step 2 (after mha_delta_kernel): dq / dk / dv on the split-operand MFMAs; the caller then launches the f32 kernels gated on *flag == run_id
void bigc_mha_bwd_h_launch(hipStream_t s, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* dO, int lddo,
                           const float* lse, const float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                           const int32_t* q_off, const int32_t* k_off, int n_seq, int n_qt, int n_kt, int n_head, float scale,
                           const DropSpec& drop, int* flag, int run_id, const float* amax_slot) {
    // default (round 6): one launch, the other side's tiles staged split in LDS; BIGC_MHA_BWD_LDS=0 = the two wave-per-tile kernels
    static const bool lds_off = getenv("BIGC_MHA_BWD_LDS") && atoi(getenv("BIGC_MHA_BWD_LDS")) == 0;
    if (!lds_off) {
        FusedArgs a;
        a.q = q; a.k = k; a.v = v; a.dO = dO; a.lse = lse; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv;
        a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
        a.q_off = q_off; a.k_off = k_off; a.n_head = n_head; a.scale = scale; a.drop = drop; a.flag = flag; a.run_id = run_id; a.amax_slot = amax_slot;
        const int nw = (n_qt <= 2 && n_kt <= 2) ? 2 : 4;              // own tiles per workgroup: sequences of at most 64 rows (the encoder) fit two waves
        static const bool roles_off = getenv("BIGC_MHA_BWD_ROLES") && atoi(getenv("BIGC_MHA_BWD_ROLES")) == 0;
        static std::atomic<unsigned long long> attr{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!((attr.load(std::memory_order_relaxed) >> (dev & 63)) & 1ull)) {
            (void)hipFuncSetAttribute((const void*)mha_bwd_fused_h_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF);
            (void)hipFuncSetAttribute((const void*)mha_bwd_fused_h_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF);
            (void)hipFuncSetAttribute((const void*)mha_bwd_roles_h_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF);
            (void)hipFuncSetAttribute((const void*)mha_bwd_roles_h_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF);
            attr.fetch_or(1ull << (dev & 63), std::memory_order_relaxed);
        }
        if (!roles_off) {
            // one output per wave: 2 nw waves per workgroup; KV groups of nw key tiles, Q groups of 2 nw query tiles
            a.gk_groups = (n_kt + nw - 1) / nw; a.gq_groups = (n_qt + 2 * nw - 1) / (2 * nw);
            const int grid = n_seq * n_head * (a.gq_groups + a.gk_groups);
            if (nw == 2) hipLaunchKernelGGL(mha_bwd_roles_h_kernel<2>, dim3(grid), dim3(256), 2 * LDS_BUF, s, a);
            else hipLaunchKernelGGL(mha_bwd_roles_h_kernel<4>, dim3(grid), dim3(512), 2 * LDS_BUF, s, a);
            return;
        }
        a.gq_groups = (n_qt + nw - 1) / nw; a.gk_groups = (n_kt + nw - 1) / nw;
        const int grid = n_seq * n_head * (a.gq_groups + a.gk_groups);
        if (nw == 2) hipLaunchKernelGGL(mha_bwd_fused_h_kernel<2>, dim3(grid), dim3(128), 2 * LDS_BUF, s, a);
        else hipLaunchKernelGGL(mha_bwd_fused_h_kernel<4>, dim3(grid), dim3(256), 2 * LDS_BUF, s, a);
        return;
    }
    const int gq = (int)(((long)n_seq * n_qt * n_head + 3) / 4), gk = (int)(((long)n_seq * n_kt * n_head + 3) / 4);
    hipLaunchKernelGGL(mha_bwd_dq_h_kernel, dim3(gq), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dO, lddo, lse, delta, dq, lddq, q_off, k_off, n_seq, n_qt,
                       n_head, scale, drop, flag, run_id, amax_slot);
    hipLaunchKernelGGL(mha_bwd_dkv_h_kernel, dim3(gk), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dO, lddo, lse, delta, dk, lddk, dv, lddv, q_off, k_off, n_seq,
                       n_kt, n_head, scale, drop, flag, run_id, amax_slot);
}
