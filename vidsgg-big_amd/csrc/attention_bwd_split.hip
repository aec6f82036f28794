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
    const int gq = (int)(((long)n_seq * n_qt * n_head + 3) / 4), gk = (int)(((long)n_seq * n_kt * n_head + 3) / 4);
    hipLaunchKernelGGL(mha_bwd_dq_h_kernel, dim3(gq), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dO, lddo, lse, delta, dq, lddq, q_off, k_off, n_seq, n_qt,
                       n_head, scale, drop, flag, run_id, amax_slot);
    hipLaunchKernelGGL(mha_bwd_dkv_h_kernel, dim3(gk), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dO, lddo, lse, delta, dk, lddk, dv, lddv, q_off, k_off, n_seq,
                       n_kt, n_head, scale, drop, flag, run_id, amax_slot);
}
