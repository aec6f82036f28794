// Attention core for head_dim 64 on the f16 matrix cores with exactly split f32 operands (the scheme of linear_split.hip):
// every f32 operand x = h + r * 2^-11 (h = f16(x) leading bits, r = f16 of the scaled remainder), a product costs three
// v_mfma_f32_32x32x16_f16 (h h, h r, r h) accumulated in f32 in two accumulators -- f32-level error at 1/5 of the MFMA
// cycles of the f32-MFMA kernels in attention.hip.  Same organisation as mha_core_sp_kernel: one wave per (sequence, head,
// 32-query tile), transposed score tiles S^T = K Q^T so that a lane owns one query column, all score tiles of the (short)
// sequence in registers, softmax exactly as torch (exp(s - max) / sum), then O = P V with the probabilities -- already in
// the A-operand layout -- split on the fly.
//
// Operand slots of v_mfma_f32_32x32x16: lane l supplies A[row l&31][slots 8(l>>5) + j] and B[slots 8(l>>5) + j][col l&31],
// j = 0..7; which contraction index a slot stands for is free as long as A and B agree:
//   S^T:  slot (hi, j) of step s  <->  head dimension 16 s + 8 hi + j            (8 consecutive floats of a K / Q row)
//   P V:  slot (hi, j) of step u  <->  key 16 u + 8 (j>>2) + 4 hi + (j&3)        (= accumulator register 8 u + j of S^T)
// Range: an operand with |x| >= 32768 (or NaN/Inf) raises `flag`; the launcher runs the f32-MFMA kernel right behind,
// which only does work when the flag carries this launch's id.
#include "common.h"
#include "prof.h"
#include <cstdlib>

typedef _Float16 a16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 a16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr float NEG_BIG = -1e30f;
constexpr float RS = 2048.0f, RI = 1.0f / 2048.0f;
// e^x through v_exp_f32 (2^x, ~1 ulp): the chunked head_dim 16 kernels evaluate 16 exponentials per lane and key tile and are bound by
// their VALU work, the library expf costs ~10 instructions each
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// 8 floats -> 8 heads + 8 scaled residuals; `amax` / `chk` track the range as in linear_split.hip
__device__ __forceinline__ void split8(const float (&x)[8], a16x8& h, a16x8& r, float& chk, float& amax) {
    union { a16x2 p[4]; a16x8 v; } hh, rr;
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

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// HD = 64: four 16-dimension steps per score tile, two 32-column output tiles; HD = 16 (the grounding encoders, 8 heads of
// d = 128): one step, one output tile whose upper 16 columns are idle.
template <int NKT, int HD = 64>
__global__ __launch_bounds__(256, 2) void mha_core_h_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                         const float* __restrict__ v, int ldv, float* __restrict__ o, int ldo,
                                                         const int32_t* __restrict__ q_off, const int32_t* __restrict__ k_off,
                                                         int n_seq, int n_qt, int n_head, float scale, float* __restrict__ lse,
                                                         const DropSpec drop, int* __restrict__ flag, int run_id) {
    constexpr int NST = HD / 16;                 // 16-dimension steps of a score tile
    constexpr int NT = HD >= 32 ? HD / 32 : 1;   // 32-column output tiles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)n_seq * n_qt * n_head) return;
    const int h = item % n_head;
    const int qt = (item / n_head) % n_qt;
    const int s = item / ((long)n_head * n_qt);
    const int q0 = q_off[s], nq = q_off[s + 1] - q0;
    const int k0 = k_off[s], nk = k_off[s + 1] - k0;
    if (qt * 32 >= nq || nk <= 0) return;
    const int n_kt = (nk + 31) / 32;
    const int qrow = min(qt * 32 + l31, nq - 1);
    float chk = 0.f, amax = 0.f;

    // this lane's query row, its 8-float chunks (head dims 16 s + 8 hi ..), pre-scaled and split
    a16x8 qh[NST], qr[NST];
    {
        const float* qp = q + (size_t)(q0 + qrow) * ldq + h * HD + 8 * hi;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const f32x4 t0 = *(const f32x4*)(qp + 16 * st), t1 = *(const f32x4*)(qp + 16 * st + 4);
            const float x[8] = {t0[0] * scale, t0[1] * scale, t0[2] * scale, t0[3] * scale, t1[0] * scale, t1[1] * scale, t1[2] * scale, t1[3] * scale};
            split8(x, qh[st], qr[st], chk, amax);
        }
    }

    // ---- scores: S^T[key][query], all tiles kept
    f32x16 sc[NKT];
    float m = NEG_BIG;
    const float* kbase = k + (size_t)k0 * ldk + h * HD + 8 * hi;
    f32x4 ka[2 * NST], kb[2 * NST];
    auto fetch_k = [&](int kt, f32x4 (&kv)[2 * NST]) {
        const float* kp = kbase + (size_t)min(min(kt, n_kt - 1) * 32 + l31, nk - 1) * ldk;
#pragma unroll
        for (int st = 0; st < NST; ++st) { kv[2 * st] = *(const f32x4*)(kp + 16 * st); kv[2 * st + 1] = *(const f32x4*)(kp + 16 * st + 4); }
    };
    auto score = [&](int kt, const f32x4 (&kv)[2 * NST]) {
        f32x16 c0 = zero16(), c1 = zero16();
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const float x[8] = {kv[2 * st][0], kv[2 * st][1], kv[2 * st][2], kv[2 * st][3], kv[2 * st + 1][0], kv[2 * st + 1][1], kv[2 * st + 1][2], kv[2 * st + 1][3]};
            a16x8 kh, kr;
            split8(x, kh, kr, chk, amax);
            c0 = MFMA_H(kh, qh[st], c0);
            c1 = MFMA_H(kh, qr[st], c1);
            c1 = MFMA_H(kr, qh[st], c1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[kt][r] = c0[r] + c1[r] * RI;
            if (kt * 32 + mfma_row(r, hi) < nk) m = fmaxf(m, sc[kt][r]);
        }
    };
    fetch_k(0, ka);
#pragma unroll
    for (int kt = 0; kt < NKT; kt += 2) {
        if (kt < n_kt) { fetch_k(kt + 1, kb); score(kt, ka); } else { sc[kt] = zero16(); }
        if (kt + 1 < NKT) {
            if (kt + 1 < n_kt) { fetch_k(kt + 2, ka); score(kt + 1, kb); } else { sc[kt + 1] = zero16(); }
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = kt < n_kt && kt * 32 + mfma_row(r, hi) < nk;
            const float e = ok ? expf(sc[kt][r] - m) : 0.f;
            sc[kt][r] = e;
            l += e;
        }
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    if (lse && hi == 0 && qt * 32 + l31 < nq) lse[(size_t)(q0 + qt * 32 + l31) * n_head + h] = m + logf(l);

    // ---- O = P V: output column c of d-tile t is head column NT c + t (one load per key row, one store per query row); head
    // widths below 32 leave the lanes past the width on column 0, their results unused
    struct vrow { float x[NT]; };
    f32x16 o0[NT], o1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { o0[t] = zero16(); o1[t] = zero16(); }
    const bool col_ok = NT * l31 < HD;
    const float* vbase = v + (size_t)k0 * ldv + h * HD + (col_ok ? NT * l31 : 0);
    vrow va[16], vb[16];
    auto fetch_v = [&](int kt, vrow (&vv)[16]) {
        kt = min(kt, n_kt - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = *(const vrow*)(vbase + (size_t)min(kt * 32 + mfma_row(r, hi), nk - 1) * ldv);
    };
    auto pv_tile = [&](int kt, const vrow (&vv)[16]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float p8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p = sc[kt][8 * u + j] * inv;
                if (drop.thresh) p = drop_keep(drop, drop_att_idx(q0 + qrow, n_head, h, kt * 32 + mfma_row(8 * u + j, hi))) ? p * drop.scale : 0.f;
                p8[j] = p;
            }
            a16x8 ph, pr;
            float c_ = 0.f, a_ = 0.f;                         // probabilities are in [0, 1 / (1 - p_drop)]: no range check needed
            split8(p8, ph, pr, c_, a_);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float v8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v8[j] = vv[8 * u + j].x[t];
                a16x8 vh, vr;
                split8(v8, vh, vr, chk, amax);
                o0[t] = MFMA_H(ph, vh, o0[t]);
                o1[t] = MFMA_H(ph, vr, o1[t]);
                o1[t] = MFMA_H(pr, vh, o1[t]);
            }
        }
    };
    fetch_v(0, va);
#pragma unroll
    for (int kt = 0; kt < NKT; kt += 2) {
        if (kt < n_kt) { fetch_v(kt + 1, vb); pv_tile(kt, va); }
        if (kt + 1 < NKT && kt + 1 < n_kt) { fetch_v(kt + 2, va); pv_tile(kt + 1, vb); }
    }
    if (!(fabsf(chk) < INFINITY) || amax >= 32768.0f) atomicMax(flag, run_id);
    if (!col_ok) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = qt * 32 + mfma_row(r, hi);
        if (row < nq) {
            vrow ov;
#pragma unroll
            for (int t = 0; t < NT; ++t) ov.x[t] = o0[t][r] + o1[t][r] * RI;
            *(vrow*)(o + (size_t)(q0 + row) * ldo + h * HD + NT * l31) = ov;
        }
    }
}

// head_dim 16, sequences LONGER than the 256 keys whose scores mha_core_h_kernel<8, 16> keeps in registers (the grounding scorer's
// clip sequences run to T = 400): the keys are walked in chunks of 8 tiles with a running maximum and sum
// (the running output and sum are rescaled by exp(m_old - m_new) when a chunk raises the maximum), so any length is one pass over
// K and V on the split-f16 MFMAs -- the two-pass f32-MFMA kernel it replaces took 550 us per launch on the grounding leg.
// Same lane roles as mha_core_h_kernel<., 16>; the normalisation happens once at the end (o / l instead of p / l before P V).
__global__ __launch_bounds__(256, 2) void mha_core_h16_long_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                                   const float* __restrict__ v, int ldv, float* __restrict__ o, int ldo,
                                                                   const int32_t* __restrict__ q_off, const int32_t* __restrict__ k_off,
                                                                   int n_seq, int n_qt, int n_head, float scale, int* __restrict__ flag, int run_id) {
    constexpr int HD = 16, CH = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)n_seq * n_qt * n_head) return;
    const int h = item % n_head;
    const int qt = (item / n_head) % n_qt;
    const int s = item / ((long)n_head * n_qt);
    const int q0 = q_off[s], nq = q_off[s + 1] - q0;
    const int k0 = k_off[s], nk = k_off[s + 1] - k0;
    if (qt * 32 >= nq || nk <= 0) return;
    const int n_kt = (nk + 31) / 32;
    const int qrow = min(qt * 32 + l31, nq - 1);
    float chk = 0.f, amax = 0.f;
    a16x8 qh, qr;
    {
        const float* qp = q + (size_t)(q0 + qrow) * ldq + h * HD + 8 * hi;
        const f32x4 t0 = *(const f32x4*)qp, t1 = *(const f32x4*)(qp + 4);
        const float x[8] = {t0[0] * scale, t0[1] * scale, t0[2] * scale, t0[3] * scale, t1[0] * scale, t1[1] * scale, t1[2] * scale, t1[3] * scale};
        split8(x, qh, qr, chk, amax);
    }
    const float* kbase = k + (size_t)k0 * ldk + h * HD + 8 * hi;
    const bool col_ok = l31 < HD;
    const float* vbase = v + (size_t)k0 * ldv + h * HD + (col_ok ? l31 : 0);
    float m = NEG_BIG, l = 0.f;
    f32x16 o0 = zero16(), o1 = zero16();
    for (int c0 = 0; c0 < n_kt; c0 += CH) {
        f32x16 sc[CH];
        float mc = NEG_BIG;
#pragma unroll
        for (int kt = 0; kt < CH; ++kt) {
            if (c0 + kt < n_kt) {
                const float* kp = kbase + (size_t)min((c0 + kt) * 32 + l31, nk - 1) * ldk;
                const f32x4 a = *(const f32x4*)kp, b = *(const f32x4*)(kp + 4);
                const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                a16x8 kh, kr;
                split8(x, kh, kr, chk, amax);
                f32x16 c_0 = zero16(), c_1 = zero16();
                c_0 = MFMA_H(kh, qh, c_0);
                c_1 = MFMA_H(kh, qr, c_1);
                c_1 = MFMA_H(kr, qh, c_1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sc[kt][r] = c_0[r] + c_1[r] * RI;
                    if ((c0 + kt) * 32 + mfma_row(r, hi) < nk) mc = fmaxf(mc, sc[kt][r]);
                }
            } else {
                sc[kt] = zero16();
            }
        }
        mc = fmaxf(mc, __shfl_xor(mc, 32));
        const float mn = fmaxf(m, mc);
        const float alpha = fast_exp(m - mn);               // 0 for the first chunk (m = NEG_BIG)
        m = mn;
        l *= alpha;
        // the running output is held TRANSPOSED to the scores: register r of a lane belongs to query row mfma_row(r, hi), whose
        // rescale factor lives in lane mfma_row(r, hi)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float ar = __shfl(alpha, mfma_row(r, hi)); o0[r] *= ar; o1[r] *= ar; }
#pragma unroll
        for (int kt = 0; kt < CH; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = c0 + kt < n_kt && (c0 + kt) * 32 + mfma_row(r, hi) < nk;
                const float e = ok ? fast_exp(sc[kt][r] - m) : 0.f;
                sc[kt][r] = e;
                l += e;
            }
#pragma unroll
        for (int kt = 0; kt < CH; ++kt) {
            if (c0 + kt < n_kt) {
                float vv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[r] = vbase[(size_t)min((c0 + kt) * 32 + mfma_row(r, hi), nk - 1) * ldv];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float p8[8], v8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { p8[j] = sc[kt][8 * u + j]; v8[j] = vv[8 * u + j]; }
                    a16x8 ph, pr, vh, vr;
                    float c_ = 0.f, a_ = 0.f;
                    split8(p8, ph, pr, c_, a_);
                    split8(v8, vh, vr, chk, amax);
                    o0 = MFMA_H(ph, vh, o0);
                    o1 = MFMA_H(ph, vr, o1);
                    o1 = MFMA_H(pr, vh, o1);
                }
            }
        }
    }
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    if (!(fabsf(chk) < INFINITY) || amax >= 32768.0f) atomicMax(flag, run_id);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = qt * 32 + mfma_row(r, hi);
        const float ir = __shfl(inv, mfma_row(r, hi));           // (all lanes take part in the shuffle: no early return above it)
        if (col_ok && row < nq) o[(size_t)(q0 + row) * ldo + h * HD + l31] = (o0[r] + o1[r] * RI) * ir;
    }
}

// head_dim 16 self-attention with K and V of a (sequence, head) split ONCE into LDS and shared by the sequence's query tiles (the
// register kernels above split every key / value tile again in each of the T / 32 waves that walk it: at T = 200 .. 400 that was most
// of their VALU work).  One workgroup of NW waves per (sequence, head); wave w takes the 32-query tiles w, w + NW, ...; keys are walked
// in chunks of 8 tiles with a running maximum and sum as in mha_core_h16_long_kernel, so every length runs the SAME arithmetic -- a
// sequence's result does not depend on what else is in the launch.  LDS: K as f16 head / residual planes [key][16] (32 bytes a key),
// V transposed [column][keys + 8] with the keys of a 32-key tile permuted so that a lane's eight k-slots of the P V MFMA are one
// 16-byte read.
// CHK: key tiles per chunk of the running maximum (the arithmetic of a sequence is a function of CHK, not of its length or its company);
// MINW: waves per SIMD the register allocation must allow.  Long sequences run <4, 4, 3>: three four-wave workgroups per CU (50 KB of LDS
// each at 376 keys) where the eight-wave form with its 8 x 16 score registers fitted one (256 registers): the kernel is VALU-bound and
// the SIMDs were busy 55 % of the time.
template <int NW, int CHK = 8, int MINW = 1>
__global__ __launch_bounds__(64 * NW, MINW) void mha_core_hd16_lds_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                                    const float* __restrict__ v, int ldv, float* __restrict__ o, int ldo,
                                                                    const int32_t* __restrict__ off, int n_head, float scale, int tp_max,
                                                                    int* __restrict__ flag, int run_id) {
    constexpr int HD = 16, CH = CHK, NTH = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char hd16_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int h = blockIdx.x % n_head, s = blockIdx.x / n_head;
    const int q0 = off[s], nq = off[s + 1] - q0;
    if (nq <= 0) return;
    const int n_kt = (nq + 31) / 32, tp = n_kt * 32, vs = tp_max + 8;
    unsigned char* const Kh = hd16_smem;
    unsigned char* const Kr = Kh + (size_t)tp_max * 32;
    unsigned char* const Vh = Kr + (size_t)tp_max * 32;
    unsigned char* const Vr = Vh + (size_t)HD * vs * 2;
    float chk = 0.f, amax = 0.f;
    // ---- stage K: unit = (key, half of the head dimension)
    for (int unit = tid; unit < tp * 2; unit += NTH) {
        const int key = unit >> 1, c = unit & 1;
        const float* kp = k + (size_t)(q0 + min(key, nq - 1)) * ldk + h * HD + 8 * c;
        const f32x4 a = *(const f32x4*)kp, b = *(const f32x4*)(kp + 4);
        const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        a16x8 kh, kr;
        split8(x, kh, kr, chk, amax);
        *(a16x8*)(Kh + key * 32 + 16 * c) = kh;
        *(a16x8*)(Kr + key * 32 + 16 * c) = kr;
    }
    // ---- stage V^T: unit = (tile, group of 8 permuted keys, column); position 8 grp + j of a tile holds key (j & 3) + 8 (j >> 2) + 16 (grp >> 1) + 4 (grp & 1)
    for (int unit = tid; unit < n_kt * 4 * HD; unit += NTH) {
        const int col = unit & 15, grp = (unit >> 4) & 3, tile = unit >> 6;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = tile * 32 + (j & 3) + 8 * (j >> 2) + 16 * (grp >> 1) + 4 * (grp & 1);
            x[j] = v[(size_t)(q0 + min(key, nq - 1)) * ldv + h * HD + col];
        }
        a16x8 vh, vr;
        split8(x, vh, vr, chk, amax);
        *(a16x8*)(Vh + ((size_t)col * vs + tile * 32 + grp * 8) * 2) = vh;
        *(a16x8*)(Vr + ((size_t)col * vs + tile * 32 + grp * 8) * 2) = vr;
    }
    __syncthreads();
    const int colc = min(l31, HD - 1);
    for (int qt = wave; qt < n_kt; qt += NW) {
        const int qrow = min(qt * 32 + l31, nq - 1);
        a16x8 qh, qr;
        {
            const float* qp = q + (size_t)(q0 + qrow) * ldq + h * HD + 8 * hi;
            const f32x4 t0 = *(const f32x4*)qp, t1 = *(const f32x4*)(qp + 4);
            const float x[8] = {t0[0] * scale, t0[1] * scale, t0[2] * scale, t0[3] * scale, t1[0] * scale, t1[1] * scale, t1[2] * scale, t1[3] * scale};
            split8(x, qh, qr, chk, amax);
        }
        float m = NEG_BIG, l = 0.f;
        f32x16 o0 = zero16(), o1 = zero16();
        for (int c0 = 0; c0 < n_kt; c0 += CH) {
            f32x16 sc[CH];
            float mc = NEG_BIG;
#pragma unroll
            for (int kt = 0; kt < CH; ++kt) {
                if (c0 + kt < n_kt) {
                    const int key = (c0 + kt) * 32 + l31;
                    const a16x8 kh = *(const a16x8*)(Kh + key * 32 + 16 * hi), kr = *(const a16x8*)(Kr + key * 32 + 16 * hi);
                    f32x16 c_0 = zero16(), c_1 = zero16();
                    c_0 = MFMA_H(kh, qh, c_0);
                    c_1 = MFMA_H(kh, qr, c_1);
                    c_1 = MFMA_H(kr, qh, c_1);
                    // keys past the sequence (the last tile only: a uniform branch) get the score NEG_BIG here, once: they then drop out of the
                    // maximum and their exponential is exactly 0 -- the same bits as selecting 0 per element in the loops below, which cost
                    // three VALU instructions per score on a kernel that is VALU-bound (530 -> 430 us on the benchmark's combined encoder)
                    if ((c0 + kt + 1) * 32 > nq) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[kt][r] = (c0 + kt) * 32 + mfma_row(r, hi) < nq ? c_0[r] + c_1[r] * RI : NEG_BIG;
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[kt][r] = c_0[r] + c_1[r] * RI;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) mc = fmaxf(mc, sc[kt][r]);
                } else {
                    sc[kt] = zero16();
                }
            }
            mc = fmaxf(mc, __shfl_xor(mc, 32));
            const float mn = fmaxf(m, mc);
            const float alpha = fast_exp(m - mn);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float ar = __shfl(alpha, mfma_row(r, hi)); o0[r] *= ar; o1[r] *= ar; }
#pragma unroll
            for (int kt = 0; kt < CH; ++kt) {
                if (c0 + kt >= n_kt) continue;              // (uniform; a tile past the sequence would only add zeros: skipping it changes no bit)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = fast_exp(sc[kt][r] - m);
                    sc[kt][r] = e;
                    l += e;
                }
            }
#pragma unroll
            for (int kt = 0; kt < CH; ++kt) {
                if (c0 + kt < n_kt) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float p8[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) p8[j] = sc[kt][8 * u + j];
                        a16x8 ph, pr;
                        float c_ = 0.f, a_ = 0.f;
                        split8(p8, ph, pr, c_, a_);
                        const size_t va = ((size_t)colc * vs + (c0 + kt) * 32 + 8 * (2 * u + hi)) * 2;
                        const a16x8 vh = *(const a16x8*)(Vh + va), vr = *(const a16x8*)(Vr + va);
                        o0 = MFMA_H(ph, vh, o0);
                        o1 = MFMA_H(ph, vr, o1);
                        o1 = MFMA_H(pr, vh, o1);
                    }
                }
            }
        }
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = qt * 32 + mfma_row(r, hi);
            const float ir = __shfl(inv, mfma_row(r, hi));
            if (l31 < HD && row < nq) o[(size_t)(q0 + row) * ldo + h * HD + l31] = (o0[r] + o1[r] * RI) * ir;
        }
    }
    if (!(fabsf(chk) < INFINITY) || amax >= 32768.0f) atomicMax(flag, run_id);
}

// split without the NaN/Inf sum: only the magnitude is tracked (a NaN operand gives NaN outputs on either path)
__device__ __forceinline__ void split8a(const float (&x)[8], a16x8& h, a16x8& r, float& amax) {
    union { a16x2 p[4]; a16x8 v; } hh, rr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        hh.p[j] = bigc_pk_rtn(a, b);
        rr.p[j] = bigc_pk_rtn((a - (float)hh.p[j][0]) * RS, (b - (float)hh.p[j][1]) * RS);
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(amax) : "v"(amax), "v"(a), "v"(b));
    }
    h = hh.v;
    r = rr.v;
}

// ---------------------------------------------------------------------------------------------
// Role attention and its values (models/model_0v10.py:193-211) for videos of at most 64 entities, on split f16 operands:
// the organisation of role_fused_kernel in attention.hip -- one wave per (video, 32-query tile), the transposed scores of
// both roles against both 32-entity tiles in registers, softmax over the entities of a role, then over the two roles,
// values_r = att_r @ enco_output -- with three f16 MFMAs per f32 product.  Score k-slot (hi, j) of step s stands for
// dimension hi * half/2 + 8 s + j of the role's half, so a lane streams a contiguous half row (a 128-byte line per four
// steps); the value k-slots are the entities in accumulator-register order as in mha_core_h_kernel.
__global__ __launch_bounds__(256, 2) void role_fused_h_kernel(const float* __restrict__ p2a, int ldp, const float* __restrict__ e2a, int lde,
                                                              const float* __restrict__ enco, int ldenc, float* __restrict__ att,
                                                              float* __restrict__ values, const int32_t* __restrict__ e_off,
                                                              int n_vid, int nq, int n_qt, int half, int d, float scale,
                                                              int* __restrict__ flag, int run_id) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)n_vid * n_qt) return;
    const int qt = item % n_qt, vid = item / n_qt;
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    if (qt * 32 >= nq || ne <= 0) return;
    const int qrow = min(qt * 32 + l31, nq - 1);
    const int hh = half / 2, n_burst = hh / 32;           // half % 64 == 0 (launcher)
    const float* qp = p2a + (size_t)(vid * nq + qrow) * ldp + hi * hh;
    float amax = 0.f;

    // ---- scores: sc[r][et] = S_r^T tile (entity rows, query columns)
    f32x16 sc[2][2];
    f32x4 qraw[8], eraw[8];
    auto fetch_row = [&](const float* p, f32x4 (&v)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(p + 4 * j);
    };
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int et = 0; et < 2; ++et) {
            const float* qb = qp + r * half;
            const float* eb = e2a + (size_t)(e0 + min(et * 32 + l31, ne - 1)) * lde + hi * hh + r * half;
            f32x16 c0 = zero16(), c1 = zero16();
            fetch_row(qb, qraw);
            fetch_row(eb, eraw);
            for (int kc = 0; kc < n_burst; ++kc) {
                const int nx = min(kc + 1, n_burst - 1) * 32;
                a16x8 qh[4], qr[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float x[8] = {qraw[2 * s4][0], qraw[2 * s4][1], qraw[2 * s4][2], qraw[2 * s4][3],
                                        qraw[2 * s4 + 1][0], qraw[2 * s4 + 1][1], qraw[2 * s4 + 1][2], qraw[2 * s4 + 1][3]};
                    split8a(x, qh[s4], qr[s4], amax);
                }
                fetch_row(qb + nx, qraw);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float x[8] = {eraw[2 * s4][0], eraw[2 * s4][1], eraw[2 * s4][2], eraw[2 * s4][3],
                                        eraw[2 * s4 + 1][0], eraw[2 * s4 + 1][1], eraw[2 * s4 + 1][2], eraw[2 * s4 + 1][3]};
                    a16x8 eh, er;
                    split8a(x, eh, er, amax);
                    c0 = MFMA_H(eh, qh[s4], c0);
                    c1 = MFMA_H(eh, qr[s4], c1);
                    c1 = MFMA_H(er, qh[s4], c1);
                }
                fetch_row(eb + nx, eraw);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) sc[r][et][g] = c0[g] + c1[g] * RI;
        }

    // ---- softmax over the entities of each role (torch.softmax: exp(s - max) / sum), then over the two roles
    float m[2] = {NEG_BIG, NEG_BIG}, l[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                sc[r][et][g] *= scale;
                if (et * 32 + mfma_row(g, hi) < ne) m[r] = fmaxf(m[r], sc[r][et][g]);
            }
        m[r] = fmaxf(m[r], __shfl_xor(m[r], 32));
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int g = 0; g < 16; ++g)
                if (et * 32 + mfma_row(g, hi) < ne) l[r] += expf(sc[r][et][g] - m[r]);
        l[r] += __shfl_xor(l[r], 32);
    }
    const int qq = qt * 32 + l31;
    float* abase = att ? att + (size_t)2 * nq * e0 : nullptr;
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int n = et * 32 + mfma_row(g, hi);
            const float s0 = sc[0][et][g], s1 = sc[1][et][g];
            const float pe0 = expf(s0 - m[0]) / l[0], pe1 = expf(s1 - m[1]) / l[1];
            const float mx = fmaxf(s0, s1);
            const float w0 = expf(s0 - mx), w1 = expf(s1 - mx), ws = w0 + w1;
            const bool ok = n < ne;
            const float a0 = ok ? pe0 * (w0 / ws) : 0.f, a1 = ok ? pe1 * (w1 / ws) : 0.f;
            sc[0][et][g] = a0;
            sc[1][et][g] = a1;
            if (abase && ok && qq < nq) {
                abase[((size_t)0 * nq + qq) * ne + n] = a0;
                abase[((size_t)1 * nq + qq) * ne + n] = a1;
            }
        }

    // ---- values_r = att_r @ enco_output, 32 output columns per pass; the split probabilities are kept for all passes
    a16x8 ph[2][2][2], pr[2][2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float p8[8], a_ = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) p8[j] = sc[r][et][8 * u + j];
                split8a(p8, ph[r][et][u], pr[r][et][u], a_);          // probabilities: no range check needed
            }
    float va[32];
    // entity rows of this lane's k-slots (clamped: their probabilities are zero)
    int voff[32];
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int g = 0; g < 16; ++g) voff[16 * et + g] = (e0 + min(et * 32 + mfma_row(g, hi), ne - 1)) * ldenc + l31;
    auto fetch_v = [&](int dt) {
        const float* vb_ = enco + min(dt * 32, d - 32);
#pragma unroll
        for (int i = 0; i < 32; ++i) va[i] = vb_[voff[i]];
    };
    const int n_dt = d / 32;                               // d % 32 == 0 (launcher)
    fetch_v(0);
#pragma unroll 1
    for (int dt = 0; dt < n_dt; ++dt) {
        a16x8 vh[2][2], vr[2][2];
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v8[j] = va[16 * et + 8 * u + j];
                split8a(v8, vh[et][u], vr[et][u], amax);
            }
        fetch_v(dt + 1);                                   // the raw registers are free again: next pass's rows (clamped at the end)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x16 c0 = zero16(), c1 = zero16();
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    c0 = MFMA_H(ph[r][et][u], vh[et][u], c0);
                    c1 = MFMA_H(ph[r][et][u], vr[et][u], c1);
                    c1 = MFMA_H(pr[r][et][u], vh[et][u], c1);
                }
            float* op = values + ((size_t)r * n_vid * nq + (size_t)vid * nq) * d + dt * 32 + l31;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = qt * 32 + mfma_row(g, hi);
                if (row < nq) op[(size_t)row * d] = c0[g] + c1[g] * RI;
            }
        }
    }
    if (amax >= 32768.0f) atomicMax(flag, run_id);
}

// ---------------------------------------------------------------------------------------------
// Self-attention core with K and V of a (sequence, head) split ONCE per workgroup into LDS (mha_core_h_kernel splits them
// again in every 32-query wave: 6x the conversions and 6x the L2 round trips at 192 queries).  One workgroup of NQT waves
// per (sequence, head); wave w owns the query tile w.  LDS image, f16 head / residual planes:
//   K   [key][64 dims], 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ (r & 7)   (ds_read_b128 fragment reads)
//   V^T [dim][keys of the sequence], rows of 2 * 32 * NQT + 16 bytes; inside a 32-key tile key 16u + 8a + 4hi + b sits at
//       position 16u + 8hi + 4a + b, so that the 8 k-slots (hi, j = 4a + b) of a lane are one 16-byte read
template <int NQT, bool DROP>
__global__ __launch_bounds__(64 * NQT) void mha_core_lds_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                                const float* __restrict__ v, int ldv, float* __restrict__ o, int ldo,
                                                                const int32_t* __restrict__ off, int n_head, float scale,
                                                                float* __restrict__ lse, const DropSpec drop, int* __restrict__ flag, int run_id) {
    constexpr int HD = 64, NK = 32 * NQT, VS = 2 * NK + 16, NTH = 64 * NQT;
    constexpr int KP = NK * 128, VP = HD * VS;                 // bytes of a K plane / a V^T plane
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KP + 2 * VP];
    unsigned char* const Kh = smem;
    unsigned char* const Kr = smem + KP;
    unsigned char* const Vh = smem + 2 * KP;
    unsigned char* const Vr = smem + 2 * KP + VP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int h = blockIdx.x % n_head, s = blockIdx.x / n_head;
    const int q0 = off[s], nq = off[s + 1] - q0;               // self-attention: keys = queries
    if (nq <= 0) return;
    const int n_kt = (nq + 31) / 32;
    float chk = 0.f, amax = 0.f;

    // ---- fill: K rows (8 threads per row, 32 bytes of f32 each), V columns (a wave takes 8 keys x 64 dims)
    {
        f32x4 kraw[4][2];
        float vraw[4][8];
        const float* kb = k + (size_t)q0 * ldk + h * HD;
        const float* vb = v + (size_t)q0 * ldv + h * HD + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int unit = tid + NTH * r, key = unit >> 3, c = unit & 7;
            const float* kp = kb + (size_t)min(key, nq - 1) * ldk + 8 * c;
            kraw[r][0] = *(const f32x4*)kp;
            kraw[r][1] = *(const f32x4*)(kp + 4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k0 = 8 * (wave + NQT * r);
#pragma unroll
            for (int j = 0; j < 8; ++j) vraw[r][j] = vb[(size_t)min(k0 + j, nq - 1) * ldv];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int unit = tid + NTH * r, key = unit >> 3, c = unit & 7;
            const float x[8] = {kraw[r][0][0], kraw[r][0][1], kraw[r][0][2], kraw[r][0][3], kraw[r][1][0], kraw[r][1][1], kraw[r][1][2], kraw[r][1][3]};
            a16x8 hh, rr;
            split8(x, hh, rr, chk, amax);
            const int a = key * 128 + ((c ^ (key & 7)) << 4);
            *(a16x8*)(Kh + a) = hh;
            *(a16x8*)(Kr + a) = rr;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k0 = 8 * (wave + NQT * r);                // keys k0 .. k0+7 = tile kt, u, a: k0 = 32 kt + 16 u + 8 a
            a16x8 hh, rr;
            split8(vraw[r], hh, rr, chk, amax);
            // keys k0 + 4 hi' + b (hi' = 0, 1) go to positions (32 kt + 16 u) + 8 hi' + 4 a + b
            const int base = (k0 & ~15), a4 = ((k0 >> 3) & 1) * 4;
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const int row = lane * VS;
            *(h4*)(Vh + row + 2 * (base + a4)) = h4{hh[0], hh[1], hh[2], hh[3]};
            *(h4*)(Vh + row + 2 * (base + 8 + a4)) = h4{hh[4], hh[5], hh[6], hh[7]};
            *(h4*)(Vr + row + 2 * (base + a4)) = h4{rr[0], rr[1], rr[2], rr[3]};
            *(h4*)(Vr + row + 2 * (base + 8 + a4)) = h4{rr[4], rr[5], rr[6], rr[7]};
        }
    }
    // this wave's queries, pre-scaled and split (overlaps the fill of the other waves)
    const int qt = wave;
    const bool active = qt * 32 < nq;
    const int qrow = min(qt * 32 + l31, nq - 1);
    a16x8 qh[4], qr[4];
    {
        const float* qp = q + (size_t)(q0 + qrow) * ldq + h * HD + 8 * hi;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const f32x4 t0 = *(const f32x4*)(qp + 16 * st), t1 = *(const f32x4*)(qp + 16 * st + 4);
            const float x[8] = {t0[0] * scale, t0[1] * scale, t0[2] * scale, t0[3] * scale, t1[0] * scale, t1[1] * scale, t1[2] * scale, t1[3] * scale};
            split8(x, qh[st], qr[st], chk, amax);
        }
    }
    if (!(fabsf(chk) < INFINITY) || amax >= 32768.0f) atomicMax(flag, run_id);
    __syncthreads();
    if (!active) return;

    // ---- scores S^T[key][query]
    f32x16 sc[NQT];
    float m = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < NQT; ++kt) {
        if (kt < n_kt) {
            f32x16 c0 = zero16(), c1 = zero16();
            const int row = kt * 32 + l31;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int a = row * 128 + (((2 * st + hi) ^ (row & 7)) << 4);
                const a16x8 kh = *(const a16x8*)(Kh + a), kr = *(const a16x8*)(Kr + a);
                c0 = MFMA_H(kh, qh[st], c0);
                c1 = MFMA_H(kh, qr[st], c1);
                c1 = MFMA_H(kr, qh[st], c1);
            }
            if (kt * 32 + 32 <= nq) {                           // a full tile: no per-key test (uniform branch)
#pragma unroll
                for (int r = 0; r < 16; ++r) { sc[kt][r] = c0[r] + c1[r] * RI; m = fmaxf(m, sc[kt][r]); }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sc[kt][r] = (kt * 32 + mfma_row(r, hi) < nq) ? c0[r] + c1[r] * RI : NEG_BIG;    // exp(NEG_BIG - m) = 0
                    m = fmaxf(m, sc[kt][r]);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = NEG_BIG;
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
    // exp through v_exp_f32 (2^x): relative error ~1e-7 * max(1, |s - m|), far below the probabilities' own sensitivity to
    // the f32 round-off of s; masked keys hold NEG_BIG and come out as exactly 0
    const float mlog = m * 1.44269504088896341f;
#pragma unroll
    for (int kt = 0; kt < NQT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], 1.44269504088896341f, -mlog));
            sc[kt][r] = e;
            l += e;
        }
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    if (lse && hi == 0 && qt * 32 + l31 < nq) lse[(size_t)(q0 + qt * 32 + l31) * n_head + h] = m + logf(l);

    // ---- O = P V, output column of d-tile t = 32 t + l31
    f32x16 o0[2], o1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { o0[t] = zero16(); o1[t] = zero16(); }
#pragma unroll
    for (int kt = 0; kt < NQT; ++kt) {
        if (kt < n_kt) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float p8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float p = sc[kt][8 * u + j] * inv;
                    if (DROP) p = drop_keep(drop, drop_att_idx(q0 + qrow, n_head, h, kt * 32 + mfma_row(8 * u + j, hi))) ? p * drop.scale : 0.f;
                    p8[j] = p;
                }
                a16x8 ph, pr;
                float a_ = 0.f;
                split8a(p8, ph, pr, a_);                     // probabilities: no range check needed
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int a = (32 * t + l31) * VS + 2 * (32 * kt + 16 * u + 8 * hi);
                    const a16x8 vh = *(const a16x8*)(Vh + a), vr = *(const a16x8*)(Vr + a);
                    o0[t] = MFMA_H(ph, vh, o0[t]);
                    o1[t] = MFMA_H(ph, vr, o1[t]);
                    o1[t] = MFMA_H(pr, vh, o1[t]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = qt * 32 + mfma_row(r, hi);
            if (row < nq) o[(size_t)(q0 + row) * ldo + h * HD + 32 * t + l31] = o0[t][r] + o1[t][r] * RI;
        }
}

// ---------------------------------------------------------------------------------------------
// The same with 16-query tiles on v_mfma_f32_16x16x32_f16: NW = 4 or 12 waves per (sequence, head) workgroup instead of 2 or
// 6, half the registers per wave (the transposed score tiles of 192 keys are 48 registers), i.e. three waves per SIMD
// where the 32-query kernel has one or two: these kernels are bound by the issue of dependent instructions, not by a pipe.
// Operand slots of 16x16x32: lane l supplies A[row l & 15][slots 8 (l >> 4) + j] and B[slots 8 (l >> 4) + j][col l & 15];
// C[row 4 (l >> 4) + reg][col l & 15].  S^T: slot (g, j) of step s = head dimension 32 s + 8 g + j.  P V: slot (g, j) of
// step u = key 32 u + 16 (j >> 2) + 4 g + (j & 3) (registers j & 3 of the score tiles 2 u and 2 u + 1); V^T keeps that key at
// position 32 u + 8 g + 4 (j >> 2) + (j & 3), one 16-byte read per lane.
#define MFMA_H16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

template <int NW, bool DROP>
__global__ __launch_bounds__(64 * NW) void mha_core_lds16_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                                 const float* __restrict__ v, int ldv, float* __restrict__ o, int ldo,
                                                                 const int32_t* __restrict__ off, int n_head, float scale,
                                                                 float* __restrict__ lse, const DropSpec drop, int* __restrict__ flag, int run_id,
                                                                 unsigned char* __restrict__ os = nullptr, long ldos = 0, int* __restrict__ range_flag = nullptr,
                                                                 const float* __restrict__ qpos = nullptr, const float* __restrict__ kpos = nullptr, int ldpos = 0) {
    // qpos / kpos != nullptr: q and k are read as q + qpos[position in the sequence], k + kpos[position] (the decoder's `q = k = query + pos`,
    // models/model_0v10.py:181-183, as W query + W pos: the second term used to enter the q / k / v GEMM as a per-video periodic residual, 16
    // bytes from each of 32 rows per load in a kernel whose lanes own data rows; here a row is read whole.  One f32 addition either way.)
    // os != nullptr: the output goes out in the split form "hr8" of csrc/gemm_ps.hip (row pitch ldos bytes) INSTEAD of f32 -- its only
    // reader is the out_proj GEMM -- and an operand or result outside the form's range raises *range_flag (the step is then redone on
    // the exact path, there is no f32 fallback launch behind this one)
    constexpr int HD = 64, NK = 16 * NW, NKT = NW, VS = 2 * NK + 16, NTH = 64 * NW;
    constexpr int KP = NK * 128, VP = HD * VS;
    constexpr int KUNITS = NK * 8 / NTH, VGROUPS = NK / 8 / NW;          // = 2, 2
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KP + 2 * VP];
    unsigned char* const Kh = smem;
    unsigned char* const Kr = smem + KP;
    unsigned char* const Vh = smem + 2 * KP;
    unsigned char* const Vr = smem + 2 * KP + VP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x % n_head, s = blockIdx.x / n_head;
    const int q0 = off[s], nq = off[s + 1] - q0;
    if (nq <= 0) return;
    const int n_kt = (nq + 15) / 16;
    float chk = 0.f, amax = 0.f;
    {
        f32x4 kraw[KUNITS][2];
        float vraw[VGROUPS][8];
        const float* kb = k + (size_t)q0 * ldk + h * HD;
        const float* vb = v + (size_t)q0 * ldv + h * HD + lane;
#pragma unroll
        for (int r = 0; r < KUNITS; ++r) {
            const int unit = tid + NTH * r, key = unit >> 3, c = unit & 7;
            const float* kp = kb + (size_t)min(key, nq - 1) * ldk + 8 * c;
            kraw[r][0] = *(const f32x4*)kp;
            kraw[r][1] = *(const f32x4*)(kp + 4);
            if (kpos) {
                const float* pp = kpos + (size_t)min(key, nq - 1) * ldpos + h * HD + 8 * c;
                kraw[r][0] += *(const f32x4*)pp;
                kraw[r][1] += *(const f32x4*)(pp + 4);
            }
        }
#pragma unroll
        for (int r = 0; r < VGROUPS; ++r) {
            const int k0 = 8 * (wave + NW * r);
#pragma unroll
            for (int j = 0; j < 8; ++j) vraw[r][j] = vb[(size_t)min(k0 + j, nq - 1) * ldv];
        }
#pragma unroll
        for (int r = 0; r < KUNITS; ++r) {
            const int unit = tid + NTH * r, key = unit >> 3, c = unit & 7;
            const float x[8] = {kraw[r][0][0], kraw[r][0][1], kraw[r][0][2], kraw[r][0][3], kraw[r][1][0], kraw[r][1][1], kraw[r][1][2], kraw[r][1][3]};
            a16x8 hh, rr;
            split8(x, hh, rr, chk, amax);
            const int a = key * 128 + ((c ^ (key & 7)) << 4);
            *(a16x8*)(Kh + a) = hh;
            *(a16x8*)(Kr + a) = rr;
        }
#pragma unroll
        for (int r = 0; r < VGROUPS; ++r) {
            const int k0 = 8 * (wave + NW * r);                  // keys k0 .. k0+7: 32-group u = k0 / 32, m = (k0 / 8) % 4
            a16x8 hh, rr;
            split8(vraw[r], hh, rr, chk, amax);
            // key 32u + 8m + j = 32u + 16a + 4g' + b with a = m >> 1, g' = 2 (m & 1) + (j >> 2), b = j & 3 -> position 32u + 8g' + 4a + b
            const int mm = (k0 >> 3) & 3;
            const int p0 = (k0 & ~31) + 16 * (mm & 1) + 4 * (mm >> 1);
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const int row = lane * VS;
            *(h4*)(Vh + row + 2 * p0) = h4{hh[0], hh[1], hh[2], hh[3]};
            *(h4*)(Vh + row + 2 * (p0 + 8)) = h4{hh[4], hh[5], hh[6], hh[7]};
            *(h4*)(Vr + row + 2 * p0) = h4{rr[0], rr[1], rr[2], rr[3]};
            *(h4*)(Vr + row + 2 * (p0 + 8)) = h4{rr[4], rr[5], rr[6], rr[7]};
        }
    }
    const int qt = wave;
    const bool active = qt * 16 < nq;
    const int qrow = min(qt * 16 + l15, nq - 1);
    a16x8 qh[2], qr[2];
    {
        const float* qp = q + (size_t)(q0 + qrow) * ldq + h * HD + 8 * g;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x4 t0 = *(const f32x4*)(qp + 32 * st), t1 = *(const f32x4*)(qp + 32 * st + 4);
            if (qpos) {
                const float* pp = qpos + (size_t)qrow * ldpos + h * HD + 8 * g + 32 * st;
                t0 += *(const f32x4*)pp;
                t1 += *(const f32x4*)(pp + 4);
            }
            const float x[8] = {t0[0] * scale, t0[1] * scale, t0[2] * scale, t0[3] * scale, t1[0] * scale, t1[1] * scale, t1[2] * scale, t1[3] * scale};
            split8(x, qh[st], qr[st], chk, amax);
        }
    }
    if (!(fabsf(chk) < INFINITY) || amax >= 32768.0f) { atomicMax(flag, run_id); if (range_flag) atomicOr(range_flag, 1); }
    __syncthreads();
    if (!active) return;

    // ---- scores S^T[key][query]: tile kt = keys 16 kt .. + 15, this lane holds keys 16 kt + 4 g + reg of query l15
    f32x4 sc[NKT];
    float m = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt < n_kt) {
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
            const int row = kt * 16 + l15;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int a = row * 128 + (((4 * st + g) ^ (row & 7)) << 4);
                const a16x8 kh = *(const a16x8*)(Kh + a), kr = *(const a16x8*)(Kr + a);
                c0 = MFMA_H16(kh, qh[st], c0);
                c1 = MFMA_H16(kh, qr[st], c1);
                c1 = MFMA_H16(kr, qh[st], c1);
            }
            if (kt * 16 + 16 <= nq) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[kt][r] = c0[r] + c1[r] * RI; m = fmaxf(m, sc[kt][r]); }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[kt][r] = (kt * 16 + 4 * g + r < nq) ? c0[r] + c1[r] * RI : NEG_BIG;
                    m = fmaxf(m, sc[kt][r]);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[kt][r] = NEG_BIG;
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
    const float mlog = m * 1.44269504088896341f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], 1.44269504088896341f, -mlog));       // masked keys: exactly 0
            sc[kt][r] = e;
            l += e;
        }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.f / l;
    if (lse && g == 0 && qt * 16 + l15 < nq) lse[(size_t)(q0 + qt * 16 + l15) * n_head + h] = m + logf(l);

    // ---- O = P V: C[query 4 g + reg][column 16 t + l15]
    f32x4 o0[4], o1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { o0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; o1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
        if (2 * u < n_kt) {
            float p8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p = sc[2 * u + (j >> 2)][j & 3] * inv;
                if (DROP) p = drop_keep(drop, drop_att_idx(q0 + qrow, n_head, h, 32 * u + 16 * (j >> 2) + 4 * g + (j & 3))) ? p * drop.scale : 0.f;
                p8[j] = p;
            }
            a16x8 ph, pr;
            float a_ = 0.f;
            split8a(p8, ph, pr, a_);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int a = (16 * t + l15) * VS + 2 * (32 * u + 8 * g);
                const a16x8 vh = *(const a16x8*)(Vh + a), vr = *(const a16x8*)(Vr + a);
                o0[t] = MFMA_H16(ph, vh, o0[t]);
                o1[t] = MFMA_H16(ph, vr, o1[t]);
                o1[t] = MFMA_H16(pr, vh, o1[t]);
            }
        }
    }
    if (os) {
        bool oob = false;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = qt * 16 + 4 * g + r, col = h * HD + 16 * t + l15;
                const float X = (o0[t][r] + o1[t][r] * RI) * 8.0f;              // PS_ACT_EXP = 3
                const _Float16 hh = (_Float16)X, rr = (_Float16)(X - (float)hh);
                if (row < nq) {
                    unsigned char* gp = os + (size_t)(q0 + row) * ldos + (size_t)(col >> 3) * 32 + (col & 7) * 2;
                    *(_Float16*)gp = hh;
                    *(_Float16*)(gp + 16) = rr;
                    oob = oob || !(fabsf(X) < 32768.f);
                }
            }
        if (oob && range_flag) atomicOr(range_flag, 1);
        return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = qt * 16 + 4 * g + r;
            if (row < nq) o[(size_t)(q0 + row) * ldo + h * HD + 16 * t + l15] = o0[t][r] + o1[t][r] * RI;
        }
}

// ---------------------------------------------------------------------------------------------
// Role attention + values with the entity-side operands split ONCE per video into LDS (role_fused_h_kernel splits e2a and
// enco_output again in each of the video's 32-query waves).  One workgroup of six waves per video, wave w = query tile w
// (nq <= 192), 64 KB of LDS used in turn for: e2a of role 0, e2a of role 1 (f16 head / residual planes, [entity][half] rows,
// 16-byte chunks XOR-swizzled by entity & 7), then enco_output^T in chunks of 256 columns ([column][64 entity positions],
// chunks swizzled by column & 7, entity positions permuted inside a 32-entity tile as in mha_core_lds_kernel).  The global
// loads of the next LDS image are issued before the MFMA phase of the current one and held in registers.
__global__ __launch_bounds__(384) void role_fused_lds_kernel(const float* __restrict__ p2a, int ldp, const float* __restrict__ e2a, int lde,
                                                             const float* __restrict__ enco, int ldenc, float* __restrict__ att,
                                                             float* __restrict__ values, const int32_t* __restrict__ e_off,
                                                             int n_vid, int nq, int half, int d, float scale,
                                                             int* __restrict__ flag, int run_id,
                                                             long role_stride, const float* __restrict__ vbias0, const float* __restrict__ vbias1, int relu,
                                                             long out_role_stride, int ldval) {
    constexpr int NTH = 384, PLANE = 32768, DC = 256;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PLANE];
    unsigned char* const Ph = smem;
    unsigned char* const Pr = smem + PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int vid = blockIdx.x;
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    if (ne <= 0) return;
    const int qt = wave;
    const bool active = qt * 32 < nq;
    const int qrow = min(qt * 32 + l31, nq - 1);
    const int hh = half / 2, per_row = half / 8, n_eunits = 64 * per_row, row_bytes = 2 * half;
    const float* qp = p2a + (size_t)(vid * nq + qrow) * ldp + hi * hh;
    float amax = 0.f;
    float raw[6][8];

    auto e_load = [&](int r) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int unit = min(tid + NTH * i, n_eunits - 1), ent = unit / per_row, c = unit - ent * per_row;
            const float* p = e2a + (size_t)(e0 + min(ent, ne - 1)) * lde + r * half + 8 * c;
            const f32x4 t0 = *(const f32x4*)p, t1 = *(const f32x4*)(p + 4);
            raw[i][0] = t0[0]; raw[i][1] = t0[1]; raw[i][2] = t0[2]; raw[i][3] = t0[3];
            raw[i][4] = t1[0]; raw[i][5] = t1[1]; raw[i][6] = t1[2]; raw[i][7] = t1[3];
        }
    };
    auto e_store = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int unit = tid + NTH * i;
            if (unit < n_eunits) {
                const int ent = unit / per_row, c = unit - ent * per_row;
                a16x8 h_, r_;
                split8a(raw[i], h_, r_, amax);
                const int a = ent * row_bytes + ((c ^ (ent & 7)) << 4);
                *(a16x8*)(Ph + a) = h_;
                *(a16x8*)(Pr + a) = r_;
            }
        }
    };
    // enco_output^T chunk dc: unit = (entity group g of 8, column dl of the chunk); lanes run along the columns
    auto v_load = [&](int dc, int vr) {
        const int dw = min(DC, d - dc * DC), n_units = 8 * dw;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int unit = min(tid + NTH * i, n_units - 1), g = unit / dw, dl = unit - g * dw;
            const float* p = enco + vr * role_stride + (size_t)e0 * ldenc + dc * DC + dl;
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[i][j] = p[(size_t)min(8 * g + j, ne - 1) * ldenc];
        }
    };
    auto v_store = [&](int dc) {
        const int dw = min(DC, d - dc * DC), n_units = 8 * dw;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int unit = tid + NTH * i;
            if (unit < n_units) {
                const int g = unit / dw, dl = unit - g * dw;
                a16x8 h_, r_;
                split8a(raw[i], h_, r_, amax);
                // entities 8g + 4hi' + b -> tile et = g>>2, u = (g>>1)&1, a = g&1: positions 16u + 8hi' + 4a + b of the tile
                const int c0 = 4 * (g >> 2) + 2 * ((g >> 1) & 1), sw = dl & 7, ab = 8 * (g & 1);
                const int b0 = dl * 128 + ((c0 ^ sw) << 4) + ab, b1 = dl * 128 + (((c0 + 1) ^ sw) << 4) + ab;
                *(h4*)(Ph + b0) = h4{h_[0], h_[1], h_[2], h_[3]};
                *(h4*)(Ph + b1) = h4{h_[4], h_[5], h_[6], h_[7]};
                *(h4*)(Pr + b0) = h4{r_[0], r_[1], r_[2], r_[3]};
                *(h4*)(Pr + b1) = h4{r_[4], r_[5], r_[6], r_[7]};
            }
        }
    };

    // ---- scores of both roles: sc[r][et] = S_r^T tile (entity rows, query columns)
    f32x16 sc[2][2];
    const int n_burst = hh / 32;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        f32x4 qraw[8];
        e_load(r);
#pragma unroll
        for (int j = 0; j < 8; ++j) qraw[j] = *(const f32x4*)(qp + r * half + 4 * j);
        e_store();
        __syncthreads();
        f32x16 c0[2], c1[2];
#pragma unroll
        for (int et = 0; et < 2; ++et) { c0[et] = zero16(); c1[et] = zero16(); }
        if (active) {
            for (int kc = 0; kc < n_burst; ++kc) {
                a16x8 qh[4], qr[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float x[8] = {qraw[2 * s4][0], qraw[2 * s4][1], qraw[2 * s4][2], qraw[2 * s4][3],
                                        qraw[2 * s4 + 1][0], qraw[2 * s4 + 1][1], qraw[2 * s4 + 1][2], qraw[2 * s4 + 1][3]};
                    split8a(x, qh[s4], qr[s4], amax);
                }
                const float* qn = qp + r * half + min(kc + 1, n_burst - 1) * 32;
#pragma unroll
                for (int j = 0; j < 8; ++j) qraw[j] = *(const f32x4*)(qn + 4 * j);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int chunk = hi * (hh / 8) + kc * 4 + s4;
#pragma unroll
                    for (int et = 0; et < 2; ++et) {
                        const int row = 32 * et + l31;
                        const int a = row * row_bytes + ((chunk ^ (row & 7)) << 4);
                        const a16x8 eh = *(const a16x8*)(Ph + a), er = *(const a16x8*)(Pr + a);
                        c0[et] = MFMA_H(eh, qh[s4], c0[et]);
                        c1[et] = MFMA_H(eh, qr[s4], c1[et]);
                        c1[et] = MFMA_H(er, qh[s4], c1[et]);
                    }
                }
            }
        }
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int g = 0; g < 16; ++g) sc[r][et][g] = (c0[et][g] + c1[et][g] * RI) * scale;
        __syncthreads();                                        // every wave is done reading this image
    }

    // ---- softmax over the entities of each role (torch.softmax: exp(s - max) / sum), then over the two roles: the same
    // operation sequence as role_fused_kernel, so that arg-max decisions between nearly equal weights fall the same way
    a16x8 ph[2][2][2], pr[2][2][2];
    if (active) {
        float m[2] = {NEG_BIG, NEG_BIG}, l[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    if (et * 32 + mfma_row(g, hi) < ne) m[r] = fmaxf(m[r], sc[r][et][g]);
            m[r] = fmaxf(m[r], __shfl_xor(m[r], 32));
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    if (et * 32 + mfma_row(g, hi) < ne) l[r] += expf(sc[r][et][g] - m[r]);
            l[r] += __shfl_xor(l[r], 32);
        }
        const int qq = qt * 32 + l31;
        float* abase = att ? att + (size_t)2 * nq * e0 : nullptr;
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n = et * 32 + mfma_row(g, hi);
                const float s0 = sc[0][et][g], s1 = sc[1][et][g];
                const float pe0 = expf(s0 - m[0]) / l[0], pe1 = expf(s1 - m[1]) / l[1];
                // softmax over the two roles: the larger score gives expf(0) = 1 exactly, one exponential for the other
                const float es = expf(-fabsf(s0 - s1));
                const float w0 = s0 >= s1 ? 1.0f : es, w1 = s0 >= s1 ? es : 1.0f, ws = w0 + w1;
                const bool ok = n < ne;
                const float a0 = ok ? pe0 * (w0 / ws) : 0.f, a1 = ok ? pe1 * (w1 / ws) : 0.f;
                sc[0][et][g] = a0;
                sc[1][et][g] = a1;
                if (abase && ok && qq < nq) {
                    abase[((size_t)0 * nq + qq) * ne + n] = a0;
                    abase[((size_t)1 * nq + qq) * ne + n] = a1;
                }
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float p8[8], a_ = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) p8[j] = sc[r][et][8 * u + j];
                    split8a(p8, ph[r][et][u], pr[r][et][u], a_);          // probabilities: no range check needed
                }
    }

    // ---- values_r = att_r @ enco_output, 256 columns of enco_output^T in LDS at a time.  With role_stride != 0 each role has
    // its own right-hand rows (the first role-wise linear already applied to enco_output, see bigc_role_fused_proj_f32):
    // one pass over the chunks per role, bias and ReLU in the store.
    const int n_dc = (d + DC - 1) / DC, n_vr = role_stride ? 2 : 1;
    for (int vr = 0; vr < n_vr; ++vr)
    for (int dc = 0; dc < n_dc; ++dc) {
        v_load(dc, vr);
        v_store(dc);
        __syncthreads();
        if (active) {
            const int n_dt = min(DC, d - dc * DC) / 32;                    // d % 32 == 0 (launcher)
#pragma unroll 1
            for (int dtl = 0; dtl < n_dt; ++dtl) {
                const int dl = 32 * dtl + l31;
                a16x8 vh[2][2], vr_[2][2];
#pragma unroll
                for (int et = 0; et < 2; ++et)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int a = dl * 128 + (((4 * et + 2 * u + hi) ^ (dl & 7)) << 4);
                        vh[et][u] = *(const a16x8*)(Ph + a);
                        vr_[et][u] = *(const a16x8*)(Pr + a);
                    }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (role_stride && r != vr) continue;
                    f32x16 c0 = zero16(), c1 = zero16();
#pragma unroll
                    for (int et = 0; et < 2; ++et)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            c0 = MFMA_H(ph[r][et][u], vh[et][u], c0);
                            c1 = MFMA_H(ph[r][et][u], vr_[et][u], c1);
                            c1 = MFMA_H(pr[r][et][u], vh[et][u], c1);
                        }
                    const int col = dc * DC + dl;
                    const float bv = vbias0 ? (r ? vbias1 : vbias0)[col] : 0.f;
                    float* op = values + r * out_role_stride + (size_t)vid * nq * ldval + col;
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int row = qt * 32 + mfma_row(g, hi);
                        float ov = (c0[g] + c1[g] * RI) + bv;
                        if (relu) ov = fmaxf(ov, 0.f);
                        if (row < nq) op[(size_t)row * ldval] = ov;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (amax >= 32768.0f) atomicMax(flag, run_id);
}

// ---------------------------------------------------------------------------------------------
// role_fused_lds_kernel on 16-query tiles (v_mfma_f32_16x16x32_f16): twelve waves per video instead of six -- three per SIMD
// instead of two on two SIMDs and one on the others -- at half the registers.  Same LDS images; score k-slot (g, j) of step s =
// dimension 32 s + 8 g + j of the role's half; value k-slot (g, j) of step u = entity 32 u + 16 (j >> 2) + 4 g + (j & 3), kept
// in V^T at position 32 u + 8 g + 4 (j >> 2) + (j & 3).  C[row 4 g + reg][col l15].
// NWV = waves per workgroup: 12 (one workgroup per video: every 16-query tile of up to 192 queries) or 4 (three workgroups per video,
// blockIdx.y picks the tiles 4 y .. 4 y + 3, each workgroup staging the video's entity rows itself: small batches -- at 8 videos eight
// twelve-wave workgroups left 248 CUs idle for 76 us).  A tile's arithmetic is the same in both forms: a video's bits do not depend on it.
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void role_fused_lds16_kernel(const float* __restrict__ p2a, int ldp, const float* __restrict__ e2a, int lde,
                                                               const float* __restrict__ enco, int ldenc, float* __restrict__ att,
                                                               float* __restrict__ values, const int32_t* __restrict__ e_off,
                                                               int n_vid, int nq, int half, int d, float scale,
                                                               int* __restrict__ flag, int run_id,
                                                               long role_stride, const float* __restrict__ vbias0, const float* __restrict__ vbias1, int relu,
                                                               long out_role_stride, int ldval,
                                                               unsigned char* __restrict__ vals_s = nullptr, long ldvs = 0, int* __restrict__ range_flag = nullptr) {
    // vals_s != nullptr: the values go out in the split form "hr8" (row pitch ldvs bytes, role r at byte offset 4 r out_role_stride)
    // INSTEAD of f32 -- their only reader is the GEMM of the second role-wise linears -- see mha_core_lds16_kernel
    constexpr int NTH = 64 * NWV, PLANE = 32768, DC = 256, NU = (2048 + NTH - 1) / NTH;      // 2048 staging units of 8 floats over the workgroup's threads
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PLANE + NWV * 1024];      // + 1 KB per wave: the split-form values go out through it
    unsigned char* const Ph = smem;
    unsigned char* const Pr = smem + PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int vid = blockIdx.x;
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    if (ne <= 0 || (int)blockIdx.y * NWV * 16 >= nq) return;
    const int qt = (int)blockIdx.y * NWV + wave;
    const bool active = qt * 16 < nq;
    const int qrow = min(qt * 16 + l15, nq - 1);
    const int per_row = half / 8, n_eunits = 64 * per_row, row_bytes = 2 * half, n_step = half / 32;
    const float* qp = p2a + (size_t)(vid * nq + qrow) * ldp + 8 * g;
    float amax = 0.f;
    bool hr_oob = false;
    float raw[NU][8];

    auto e_load = [&](int r) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int unit = min(tid + NTH * i, n_eunits - 1), ent = unit / per_row, c = unit - ent * per_row;
            const float* p = e2a + (size_t)(e0 + min(ent, ne - 1)) * lde + r * half + 8 * c;
            const f32x4 t0 = *(const f32x4*)p, t1 = *(const f32x4*)(p + 4);
            raw[i][0] = t0[0]; raw[i][1] = t0[1]; raw[i][2] = t0[2]; raw[i][3] = t0[3];
            raw[i][4] = t1[0]; raw[i][5] = t1[1]; raw[i][6] = t1[2]; raw[i][7] = t1[3];
        }
    };
    auto e_store = [&]() {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int unit = tid + NTH * i;
            if (unit < n_eunits) {
                const int ent = unit / per_row, c = unit - ent * per_row;
                a16x8 h_, r_;
                split8a(raw[i], h_, r_, amax);
                const int a = ent * row_bytes + ((c ^ (ent & 7)) << 4);
                *(a16x8*)(Ph + a) = h_;
                *(a16x8*)(Pr + a) = r_;
            }
        }
    };
    auto v_load = [&](int dc, int vr) {
        const int dw = min(DC, d - dc * DC), n_units = 8 * dw;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int unit = min(tid + NTH * i, n_units - 1), m = unit / dw, dl = unit - m * dw;
            const float* p = enco + vr * role_stride + (size_t)e0 * ldenc + dc * DC + dl;
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[i][j] = p[(size_t)min(8 * m + j, ne - 1) * ldenc];
        }
    };
    auto v_store = [&](int dc) {
        const int dw = min(DC, d - dc * DC), n_units = 8 * dw;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int unit = tid + NTH * i;
            if (unit < n_units) {
                const int m = unit / dw, dl = unit - m * dw;
                a16x8 h_, r_;
                split8a(raw[i], h_, r_, amax);
                // entities 8m + j = 32u + 16a + 4g' + b (u = m >> 2, a = (m >> 1) & 1, g' = 2 (m & 1) + (j >> 2), b = j & 3) -> position
                // 32u + 8g' + 4a + b: 16-byte chunk 4u + 2 (m & 1) (+1 for j >= 4), byte 8a inside it
                const int c0 = 4 * (m >> 2) + 2 * (m & 1), sw = dl & 7, ab = 8 * ((m >> 1) & 1);
                const int b0 = dl * 128 + ((c0 ^ sw) << 4) + ab, b1 = dl * 128 + (((c0 + 1) ^ sw) << 4) + ab;
                *(h4*)(Ph + b0) = h4{h_[0], h_[1], h_[2], h_[3]};
                *(h4*)(Ph + b1) = h4{h_[4], h_[5], h_[6], h_[7]};
                *(h4*)(Pr + b0) = h4{r_[0], r_[1], r_[2], r_[3]};
                *(h4*)(Pr + b1) = h4{r_[4], r_[5], r_[6], r_[7]};
            }
        }
    };

    // ---- scores: sc[r][et] = S_r^T tile et (entities 16 et + 4 g + reg, query l15)
    f32x4 sc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        e_load(r);
        f32x4 qraw[4][2];                                          // four k32 steps of this lane's query row
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { qraw[s4][0] = *(const f32x4*)(qp + r * half + 32 * s4); qraw[s4][1] = *(const f32x4*)(qp + r * half + 32 * s4 + 4); }
        e_store();
        __syncthreads();
        f32x4 c0[4], c1[4];
#pragma unroll
        for (int et = 0; et < 4; ++et) { c0[et] = f32x4{0.f, 0.f, 0.f, 0.f}; c1[et] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (active) {
            for (int sb = 0; sb < n_step; sb += 4) {
                a16x8 qh[4], qr[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const float x[8] = {qraw[s4][0][0], qraw[s4][0][1], qraw[s4][0][2], qraw[s4][0][3], qraw[s4][1][0], qraw[s4][1][1], qraw[s4][1][2], qraw[s4][1][3]};
                    split8a(x, qh[s4], qr[s4], amax);
                }
                const float* qn = qp + r * half + 32 * min(sb + 4, n_step - 4);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { qraw[s4][0] = *(const f32x4*)(qn + 32 * s4); qraw[s4][1] = *(const f32x4*)(qn + 32 * s4 + 4); }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int chunk = 4 * (sb + s4) + g;
#pragma unroll
                    for (int et = 0; et < 4; ++et) {
                        const int row = 16 * et + l15;
                        const int a = row * row_bytes + ((chunk ^ (row & 7)) << 4);
                        const a16x8 eh = *(const a16x8*)(Ph + a), er = *(const a16x8*)(Pr + a);
                        c0[et] = MFMA_H16(eh, qh[s4], c0[et]);
                        c1[et] = MFMA_H16(eh, qr[s4], c1[et]);
                        c1[et] = MFMA_H16(er, qh[s4], c1[et]);
                    }
                }
            }
        }
#pragma unroll
        for (int et = 0; et < 4; ++et)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) sc[r][et][rg] = (c0[et][rg] + c1[et][rg] * RI) * scale;
        __syncthreads();
    }

    // ---- softmax over the entities of each role, then over the two roles: the operation sequence of role_fused_kernel
    a16x8 ph[2][2], pr[2][2];
    if (active) {
        float m[2] = {NEG_BIG, NEG_BIG}, l[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int et = 0; et < 4; ++et)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (16 * et + 4 * g + rg < ne) m[r] = fmaxf(m[r], sc[r][et][rg]);
            m[r] = fmaxf(m[r], __shfl_xor(m[r], 16));
            m[r] = fmaxf(m[r], __shfl_xor(m[r], 32));
#pragma unroll
            for (int et = 0; et < 4; ++et)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (16 * et + 4 * g + rg < ne) l[r] += expf(sc[r][et][rg] - m[r]);
            l[r] += __shfl_xor(l[r], 16);
            l[r] += __shfl_xor(l[r], 32);
        }
        const int qq = qt * 16 + l15;
        float* abase = att ? att + (size_t)2 * nq * e0 : nullptr;
#pragma unroll
        for (int et = 0; et < 4; ++et)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = 16 * et + 4 * g + rg;
                const float s0 = sc[0][et][rg], s1 = sc[1][et][rg];
                const float pe0 = expf(s0 - m[0]) / l[0], pe1 = expf(s1 - m[1]) / l[1];
                const float es = expf(-fabsf(s0 - s1));
                const float w0 = s0 >= s1 ? 1.0f : es, w1 = s0 >= s1 ? es : 1.0f, ws = w0 + w1;
                const bool ok = n < ne;
                const float a0 = ok ? pe0 * (w0 / ws) : 0.f, a1 = ok ? pe1 * (w1 / ws) : 0.f;
                sc[0][et][rg] = a0;
                sc[1][et][rg] = a1;
                if (abase && ok && qq < nq) {
                    abase[((size_t)0 * nq + qq) * ne + n] = a0;
                    abase[((size_t)1 * nq + qq) * ne + n] = a1;
                }
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float p8[8], a_ = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) p8[j] = sc[r][2 * u + (j >> 2)][j & 3];
                split8a(p8, ph[r][u], pr[r][u], a_);
            }
    }

    // ---- values_r = act(att_r @ right-hand rows + bias), 256 columns of the right-hand rows^T in LDS at a time
    const int n_dc = (d + DC - 1) / DC, n_vr = role_stride ? 2 : 1;
    for (int vr = 0; vr < n_vr; ++vr)
    for (int dc = 0; dc < n_dc; ++dc) {
        v_load(dc, vr);
        v_store(dc);
        __syncthreads();
        if (active) {
            const int n_dt = min(DC, d - dc * DC) / 16;                    // d % 32 == 0 (launcher)
#pragma unroll 1
            for (int dtl = 0; dtl < n_dt; ++dtl) {
                const int dl = 16 * dtl + l15;
                a16x8 vh[2], vr_[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int a = dl * 128 + (((4 * u + g) ^ (dl & 7)) << 4);
                    vh[u] = *(const a16x8*)(Ph + a);
                    vr_[u] = *(const a16x8*)(Pr + a);
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (role_stride && r != vr) continue;
                    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        c0 = MFMA_H16(ph[r][u], vh[u], c0);
                        c1 = MFMA_H16(ph[r][u], vr_[u], c1);
                        c1 = MFMA_H16(pr[r][u], vh[u], c1);
                    }
                    const int col = dc * DC + dl;
                    const float bv = vbias0 ? (r ? vbias1 : vbias0)[col] : 0.f;
                    if (vals_s && (out_role_stride & 15) == 0 && (ldvs & 15) == 0) {
                        // (a lane holds one column of four rows: straight to memory the split form is two 2-byte stores per value, 128 bytes per
                        // store instruction, 1.6 M of them per launch; through 1 KB of LDS per wave the 16 x 16 patch is ONE 16-byte store per
                        // lane: 0.825 -> 0.745 ms per step for the six layers.  The same change in the attention core above gained nothing.)
                        unsigned char* const sc = smem + 2 * PLANE + wave * 1024;
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            float ov = (c0[rg] + c1[rg] * RI) + bv;
                            if (relu) ov = fmaxf(ov, 0.f);
                            const float X = ov * 8.0f;
                            const _Float16 hh = (_Float16)X, rr = (_Float16)(X - (float)hh);
                            unsigned char* lp = sc + (4 * g + rg) * 64 + (l15 >> 3) * 32 + (l15 & 7) * 2;
                            *(_Float16*)lp = hh;
                            *(_Float16*)(lp + 16) = rr;
                            hr_oob = hr_oob || (qt * 16 + 4 * g + rg < nq && !(fabsf(X) < 32768.f));
                        }
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        {
                            const int rl = lane >> 2, ch = lane & 3, row = qt * 16 + rl;
                            const uint4 x = *(const uint4*)(sc + rl * 64 + ch * 16);
                            const int ocol0 = (int)(r * out_role_stride) + dc * DC + 16 * dtl;      // a multiple of 16: 64 bytes of the row's split form
                            if (row < nq) *(uint4*)(vals_s + ((size_t)vid * nq + row) * ldvs + (size_t)ocol0 * 4 + ch * 16) = x;
                        }
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        continue;
                    }
                    if (vals_s) {
                        const int ocol = (int)(r * out_role_stride) + col;            // (the two roles side by side in one row)
                        unsigned char* gp = vals_s + (size_t)vid * nq * ldvs + (size_t)(ocol >> 3) * 32 + (ocol & 7) * 2;
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int row = qt * 16 + 4 * g + rg;
                            float ov = (c0[rg] + c1[rg] * RI) + bv;
                            if (relu) ov = fmaxf(ov, 0.f);
                            const float X = ov * 8.0f;
                            const _Float16 hh = (_Float16)X, rr = (_Float16)(X - (float)hh);
                            if (row < nq) {
                                *(_Float16*)(gp + (size_t)row * ldvs) = hh;
                                *(_Float16*)(gp + (size_t)row * ldvs + 16) = rr;
                                hr_oob = hr_oob || !(fabsf(X) < 32768.f);
                            }
                        }
                        continue;
                    }
                    float* op = values + r * out_role_stride + (size_t)vid * nq * ldval + col;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int row = qt * 16 + 4 * g + rg;
                        float ov = (c0[rg] + c1[rg] * RI) + bv;
                        if (relu) ov = fmaxf(ov, 0.f);
                        if (row < nq) op[(size_t)row * ldval] = ov;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (amax >= 32768.0f) { atomicMax(flag, run_id); if (range_flag) atomicOr(range_flag, 1); }
    if (hr_oob && range_flag) atomicOr(range_flag, 1);
}

}  // namespace

#include <atomic>
static std::atomic<int> g_hr8_run{0};
// the per-launch flag word the kernels raise for the f32 fallback of the f32-output path: the hr8-output launches have no fallback
// (they raise the step's range flag instead) and share one word that nobody reads
static int* hr8_dummy_flag() {
    static int* p = nullptr;
    if (!p && (hipMalloc((void**)&p, sizeof(int)) != hipSuccess || hipMemset(p, 0, sizeof(int)) != hipSuccess)) p = nullptr;
    return p;
}

int bigc_role_fused_h_launch(hipStream_t s, const float* p2a, int ldp, const float* e2a, int lde, const float* enco, int ldenc, float* att,
                             float* values, const int32_t* e_off, int n_vid, int nq, int n_qt, int half, int d, float scale, int grid,
                             int* flag, int run_id, long role_stride, const float* vbias0, const float* vbias1, int relu, long out_role_stride, int ldval) {
    // at most 192 queries and role halves of at most 256: one workgroup per video with the entity operands shared through LDS
    static const int use_lds = getenv("BIGC_ROLE_LDS") ? atoi(getenv("BIGC_ROLE_LDS")) : 1;
    const bool projected = role_stride != 0 || vbias0 || relu;      // only the LDS kernel (and the f32 fallback) implement it; the caller checked the shape
    static const int tile16 = getenv("BIGC_ROLE_TILE16") ? atoi(getenv("BIGC_ROLE_TILE16")) : 1;
    if ((use_lds || projected) && n_qt <= 6 && half <= 256 && tile16 && half % 128 == 0) {
        // fewer than 128 videos: three four-wave workgroups per video (see the kernel)
        static const int force_nwv = getenv("BIGC_ROLE_NWV") ? atoi(getenv("BIGC_ROLE_NWV")) : 0;
        if (force_nwv == 4 || (force_nwv != 12 && n_vid < 128))
            hipLaunchKernelGGL(role_fused_lds16_kernel<4>, dim3(n_vid, (nq + 63) / 64), dim3(256), 0, s, p2a, ldp, e2a, lde, enco, ldenc, att, values, e_off, n_vid, nq, half, d,
                               scale, flag, run_id, role_stride, vbias0, vbias1, relu, out_role_stride, ldval, (unsigned char*)nullptr, 0L, (int*)nullptr);
        else
            hipLaunchKernelGGL(role_fused_lds16_kernel<12>, dim3(n_vid), dim3(768), 0, s, p2a, ldp, e2a, lde, enco, ldenc, att, values, e_off, n_vid, nq, half, d,
                               scale, flag, run_id, role_stride, vbias0, vbias1, relu, out_role_stride, ldval, (unsigned char*)nullptr, 0L, (int*)nullptr);
        return 1;
    }
    if ((use_lds || projected) && n_qt <= 6 && half <= 256) {
        hipLaunchKernelGGL(role_fused_lds_kernel, dim3(n_vid), dim3(384), 0, s, p2a, ldp, e2a, lde, enco, ldenc, att, values, e_off, n_vid, nq, half, d,
                           scale, flag, run_id, role_stride, vbias0, vbias1, relu, out_role_stride, ldval);
        return 1;
    }
    hipLaunchKernelGGL(role_fused_h_kernel, dim3(grid), dim3(256), 0, s, p2a, ldp, e2a, lde, enco, ldenc, att, values, e_off, n_vid, nq, n_qt,
                       half, d, scale, flag, run_id);
    return 1;
}

// launches the split kernel when it applies (head_dim 64, self-attention of at most 192 keys); returns 1 if launched
int bigc_mha_core_h_launch(hipStream_t s, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                           const int32_t* q_off, const int32_t* k_off, int n_seq, int n_qt, int n_head, float scale, float* lse,
                           const DropSpec& drop, int grid, int* flag, int run_id) {
    // keys = queries of the same sequence (the launcher only comes here for self-attention): K / V shared through LDS
    static const int use_lds = getenv("BIGC_MHA_LDS") ? atoi(getenv("BIGC_MHA_LDS")) : 1;
    if (use_lds && n_qt <= 6) {
        static const int tile16 = getenv("BIGC_MHA_TILE16") ? atoi(getenv("BIGC_MHA_TILE16")) : 1;
        if (tile16) {
#define LAUNCH_L16(NW, DROP) hipLaunchKernelGGL((mha_core_lds16_kernel<NW, DROP>), dim3(n_seq * n_head), dim3(64 * NW), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, n_head, scale, lse, drop, flag, run_id)
            if (n_qt <= 2) { if (drop.thresh) LAUNCH_L16(4, true); else LAUNCH_L16(4, false); }
            else { if (drop.thresh) LAUNCH_L16(12, true); else LAUNCH_L16(12, false); }
#undef LAUNCH_L16
            return 1;
        }
#define LAUNCH_LDS(NQT, DROP) hipLaunchKernelGGL((mha_core_lds_kernel<NQT, DROP>), dim3(n_seq * n_head), dim3(64 * NQT), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, n_head, scale, lse, drop, flag, run_id)
        if (n_qt <= 2) { if (drop.thresh) LAUNCH_LDS(2, true); else LAUNCH_LDS(2, false); }
        else { if (drop.thresh) LAUNCH_LDS(6, true); else LAUNCH_LDS(6, false); }
#undef LAUNCH_LDS
        return 1;
    }
    if (n_qt <= 2) hipLaunchKernelGGL(mha_core_h_kernel<2>, dim3(grid), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, lse, drop, flag, run_id);
    else if (n_qt <= 6) hipLaunchKernelGGL(mha_core_h_kernel<6>, dim3(grid), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, lse, drop, flag, run_id);
    else return 0;
    return 1;
}

// head_dim 16 (self- or cross-attention with at most 8 key tiles): the register kernel above
int bigc_mha_core_h16_launch(hipStream_t s, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                             const int32_t* q_off, const int32_t* k_off, int n_seq, int n_qt, int nkt, int n_head, float scale, float* lse,
                             const DropSpec& drop, int grid, int* flag, int run_id) {
    if (nkt <= 2) hipLaunchKernelGGL((mha_core_h_kernel<2, 16>), dim3(grid), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, lse, drop, flag, run_id);
    else if (nkt <= 4) hipLaunchKernelGGL((mha_core_h_kernel<4, 16>), dim3(grid), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, lse, drop, flag, run_id);
    else if (nkt <= 8) hipLaunchKernelGGL((mha_core_h_kernel<8, 16>), dim3(grid), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, lse, drop, flag, run_id);
    else if (!lse && !drop.thresh) hipLaunchKernelGGL(mha_core_h16_long_kernel, dim3(grid), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, flag, run_id);
    else return 0;
    return 1;
}

// Self-attention core (head_dim 64, at most 192 keys per sequence) with the output in the split form "hr8" for the out_proj GEMM on
// pre-split operands; returns 0 when the shape is outside what the LDS kernel covers (the caller then uses the f32 path)
int bigc_mha_core_hr8_launch(hipStream_t s, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, void* os, long ldos,
                             const int32_t* q_off, int n_seq, int max_q, int n_head, int hd, int* range_flag,
                             const float* qpos, const float* kpos, int ldpos) {
    const int n_qt = (max_q + 31) / 32;
    if (hd != 64 || n_qt > 6 || n_seq <= 0 || !os || ldos % 16 != 0) return 0;
    const float scale = 1.0f / sqrtf((float)hd);
    const int run_id = g_hr8_run.fetch_add(1, std::memory_order_relaxed) + 1;
    int* flag = hr8_dummy_flag();
    if (!flag) return 0;
    DropSpec drop = make_drop(0.f, 0, 0);
    ProfScope ps((void*)s, FAM_MHA, 0.0, 0.0);
#define LAUNCH_L16H(NW) hipLaunchKernelGGL((mha_core_lds16_kernel<NW, false>), dim3(n_seq * n_head), dim3(64 * NW), 0, s, q, ldq, k, ldk, v, ldv, (float*)nullptr, 0, q_off, n_head, scale, (float*)nullptr, drop, flag, run_id, (unsigned char*)os, ldos, range_flag, qpos, kpos, ldpos)
    if (n_qt <= 2) LAUNCH_L16H(4); else LAUNCH_L16H(12);
#undef LAUNCH_L16H
    return 1;
}

// role_fused_lds16_kernel in its projected form (bigc_role_fused_proj_f32) with the values in the split form; returns 0 if the shape
// is outside the kernel's
int bigc_role_fused_proj_hr8_launch(hipStream_t s, const float* p2a, int ldp, const float* e2a, int lde, const float* rhs, int ldrhs, long role_stride,
                                    const float* vbias0, const float* vbias1, int relu, float* att, void* vals_s, long ldvs, long out_role_stride,
                                    const int32_t* e_off, int n_vid, int nq, int max_trk, int half, int d, float scale, int* range_flag) {
    const int n_qt = (nq + 31) / 32;
    if (max_trk > 64 || n_qt > 6 || half > 256 || half % 128 != 0 || d % 32 != 0 || !vals_s || ldvs % 16 != 0 || n_vid <= 0) return 0;
    const int run_id = g_hr8_run.fetch_add(1, std::memory_order_relaxed) + 1;
    int* flag = hr8_dummy_flag();
    if (!flag) return 0;
    ProfScope ps((void*)s, FAM_ROLE_ATT, 0.0, 0.0);
    static const int force_nwv = getenv("BIGC_ROLE_NWV") ? atoi(getenv("BIGC_ROLE_NWV")) : 0;
    if (force_nwv == 4 || (force_nwv != 12 && n_vid < 128))
        hipLaunchKernelGGL(role_fused_lds16_kernel<4>, dim3(n_vid, (nq + 63) / 64), dim3(256), 0, s, p2a, ldp, e2a, lde, rhs, ldrhs, att, (float*)nullptr, e_off, n_vid, nq, half, d,
                           scale, flag, run_id, role_stride, vbias0, vbias1, relu, out_role_stride, 0, (unsigned char*)vals_s, ldvs, range_flag);
    else
        hipLaunchKernelGGL(role_fused_lds16_kernel<12>, dim3(n_vid), dim3(768), 0, s, p2a, ldp, e2a, lde, rhs, ldrhs, att, (float*)nullptr, e_off, n_vid, nq, half, d,
                           scale, flag, run_id, role_stride, vbias0, vbias1, relu, out_role_stride, 0, (unsigned char*)vals_s, ldvs, range_flag);
    return 1;
}

// head_dim 16 self-attention through the LDS kernel above (eval: no log-sum-exp, no dropout): every sequence of up to 1024 rows;
// returns 1 if launched
int bigc_mha_core_hd16_lds_launch(hipStream_t s, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                  const int32_t* off, int n_seq, int max_q, int n_head, float scale, int* flag, int run_id) {
    static const bool off_env = getenv("BIGC_MHA_HD16_LDS") && atoi(getenv("BIGC_MHA_HD16_LDS")) == 0;
    if (off_env || max_q < 1 || max_q > 1024 || n_seq <= 0) return 0;
    const int tp = (max_q + 31) / 32 * 32;
    const size_t lds = (size_t)2 * tp * 32 + (size_t)2 * 16 * (tp + 8) * 2;
    // waves per workgroup by the longest sequence (the arithmetic of a sequence does not depend on it: wave w takes the query tiles
    // w, w + NW, ...): one wave for sequences of at most 32 rows (the query encoder's 3 tokens: 8784 workgroups of four waves, three
    // of them idle, took 203 us), two up to 64, four up to 128, eight above
#define LAUNCH_HD16(NWV, ...) do { static std::atomic<unsigned long long> attr{0}; int dev_ = 0; (void)hipGetDevice(&dev_);      /* one bit per device */ \
        if (!((attr.load(std::memory_order_relaxed) >> (dev_ & 63)) & 1ull)) { \
            (void)hipFuncSetAttribute((const void*)mha_core_hd16_lds_kernel<NWV, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); \
            attr.fetch_or(1ull << (dev_ & 63), std::memory_order_relaxed); } \
        hipLaunchKernelGGL((mha_core_hd16_lds_kernel<NWV, ##__VA_ARGS__>), dim3(n_seq * n_head), dim3(64 * NWV), lds, s, q, ldq, k, ldk, v, ldv, o, ldo, off, n_head, scale, tp, flag, run_id); } while (0)
    // (the chunk length is part of a sequence's arithmetic: it must not depend on the batch.  max_q > 128 is a property of the LAUNCH, so
    // both forms for long launches use the SAME chunk length 4; launches of short sequences only -- the video and the query encoder -- keep 8,
    // where a sequence of at most 128 keys has at most 4 tiles: one chunk either way, the same bits)
    // (which of the two long forms runs is a matter of speed only -- a wave does the same work on a query tile in both: eight waves per
    // sequence when the launch would not fill the chip with four (the video encoder's 8 x 8 workgroups: 20 vs 33 us), four otherwise
    // (the combined encoder: 314 vs 419 us))
    static const int long_form = getenv("BIGC_MHA_HD16_LONG") ? atoi(getenv("BIGC_MHA_HD16_LONG")) : 0;
    if (max_q <= 32) LAUNCH_HD16(1);
    else if (max_q <= 64) LAUNCH_HD16(2);
    else if (max_q <= 128) LAUNCH_HD16(4);
    else if (long_form == 8 || (long_form == 0 && (long)n_seq * n_head < 3 * 256)) LAUNCH_HD16(8, 4, 2);
    else LAUNCH_HD16(4, 4, 3);
#undef LAUNCH_HD16
    return 1;
}
