// Attention kernels of the BIG hot path on f32 MFMA, one wave per 32-query tile, no LDS.
//
//   mha_core        softmax(q k^T / sqrt(hd)) v per (sequence, head)      -- nn.MultiheadAttention core
//                   (reference: models/model_0v10.py:109,183; models/grd_model_v5.py:105)
//   role_attention  softmax over entities x softmax over the two roles    -- models/model_0v10.py:193-207
//   role_values     att_r @ enco_output per (video, role)                 -- models/model_0v10.py:211
//
// All three compute the TRANSPOSED score tile S^T = K Q^T (keys on MFMA rows, queries on MFMA
// columns): a lane then owns one query (column lane&31) and 16 keys in its 16 accumulator
// registers, so the softmax statistics of a query are an in-register reduction plus one
// exchange with lane^32, and the probabilities are already laid out as the A operand of the
// P V product (register r of lanes <32 / >=32 holds keys mfma_row(r,0) / mfma_row(r,1), exactly
// the two k-slices one v_mfma_f32_32x32x2_f32 consumes).  Sequences are short (N <= 180
// tracklets, 192 queries, T clips), so the kernels make two passes over the keys -- pass 1 for
// the row max / sum, pass 2 recomputing the scores, normalising exactly as torch.softmax does
// (exp(s - max) / sum) and accumulating P V -- instead of rescaling accumulators on line.
// Operands come straight from L2 (the whole working set of a video is < 1 MB).
#include "common.h"
#include "prof.h"
#include <atomic>
#include <mutex>

extern "C" int bigc_linear_get_exact(void);

namespace {

constexpr float NEG_BIG = -1e30f;

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void mha_core_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                       const float* __restrict__ v, int ldv, float* __restrict__ o, int ldo,
                                                       const int32_t* __restrict__ q_off, const int32_t* __restrict__ k_off,
                                                       int n_seq, int n_qt, int n_head, float scale, float* __restrict__ lse,
                                                       const DropSpec drop, const int* __restrict__ run_flag = nullptr, int run_id = 0) {
    if (run_flag && *run_flag != run_id) return;        // fallback launch behind mha_core_h16_long_kernel: only when that one flagged
    constexpr int HALF = HD / 2;
    constexpr int NTD = (HD + 31) / 32;
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

    // this lane's query row (clamped), its half of the head dim, pre-scaled
    const int qrow = min(qt * 32 + l31, nq - 1);
    float qf[HALF];
    {
        const float* qp = q + (size_t)(q0 + qrow) * ldq + h * HD + hi * HALF;
#pragma unroll
        for (int i = 0; i < HALF; i += 4) {
            f32x4 t = *(const f32x4*)(qp + i);
            qf[i] = t[0] * scale; qf[i + 1] = t[1] * scale; qf[i + 2] = t[2] * scale; qf[i + 3] = t[3] * scale;
        }
    }
    const int n_kt = (nk + 31) / 32;

    auto score_tile = [&](int kt) -> f32x16 {
        const int krow = min(kt * 32 + l31, nk - 1);
        const float* kp = k + (size_t)(k0 + krow) * ldk + h * HD + hi * HALF;
        float kf[HALF];
#pragma unroll
        for (int i = 0; i < HALF; i += 4) {
            f32x4 t = *(const f32x4*)(kp + i);
            kf[i] = t[0]; kf[i + 1] = t[1]; kf[i + 2] = t[2]; kf[i + 3] = t[3];
        }
        f32x16 acc = zero16();
#pragma unroll
        for (int i = 0; i < HALF; ++i) acc = mfma32(kf[i], qf[i], acc);
        return acc;
    };

    // pass 1: max and sum of exp over this lane's keys, then merge the two half-waves
    float m = NEG_BIG, l = 0.f;
    for (int kt = 0; kt < n_kt; ++kt) {
        f32x16 sc = score_tile(kt);
        float tm = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = kt * 32 + mfma_row(r, hi) < nk;
            tm = ok ? fmaxf(tm, sc[r]) : tm;
        }
        const float mn = fmaxf(m, tm);
        float add = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = kt * 32 + mfma_row(r, hi) < nk;
            add += ok ? expf(sc[r] - mn) : 0.f;
        }
        l = l * expf(m - mn) + add;
        m = mn;
    }
    {
        const float mo = __shfl_xor(m, 32), lo = __shfl_xor(l, 32);
        const float M = fmaxf(m, mo);
        l = l * expf(m - M) + lo * expf(mo - M);
        m = M;
    }
    const float inv = 1.f / l;
    // log-sum-exp of the scaled scores, kept for the backward pass (training only)
    if (lse && hi == 0 && qt * 32 + l31 < nq) lse[(size_t)(q0 + qt * 32 + l31) * n_head + h] = m + logf(l);

    // pass 2: P = exp(s - M) / L, out += P V
    f32x16 out[NTD];
#pragma unroll
    for (int t = 0; t < NTD; ++t) out[t] = zero16();
    for (int kt = 0; kt < n_kt; ++kt) {
        f32x16 sc = score_tile(kt);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + mfma_row(r, hi);
            const bool ok = key < nk;
            float p = ok ? expf(sc[r] - m) * inv : 0.f;
            if (drop.thresh) p = drop_keep(drop, drop_att_idx(q0 + qrow, n_head, h, key)) ? p * drop.scale : 0.f;
            const float* vp = v + (size_t)(k0 + min(key, nk - 1)) * ldv + h * HD;
#pragma unroll
            for (int t = 0; t < NTD; ++t) {
                const int d = t * 32 + l31;
                const float vv = vp[d < HD ? d : HD - 1];       // clamped, never predicated (columns >= HD are not stored)
                out[t] = mfma32(p, vv, out[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NTD; ++t) {
        const int d = t * 32 + l31;
        if (d >= HD) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = qt * 32 + mfma_row(r, hi);
            if (row < nq) o[(size_t)(q0 + row) * ldo + h * HD + d] = out[t][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Single-pass variant for sequences of at most 32*NKT keys: the transposed score tiles stay in registers, so the scores
// are computed once (the two-pass kernel above recomputes them for the P V product).
template <int HD, int NKT>
__global__ __launch_bounds__(256) void mha_core_sp_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                          const float* __restrict__ v, int ldv, float* __restrict__ o, int ldo,
                                                          const int32_t* __restrict__ q_off, const int32_t* __restrict__ k_off,
                                                          int n_seq, int n_qt, int n_head, float scale, float* __restrict__ lse,
                                                          const DropSpec drop, const int* __restrict__ run_flag, int run_id) {
    constexpr int HALF = HD / 2;
    constexpr int NTD = (HD + 31) / 32;
    if (run_flag && *run_flag != run_id) return;        // fallback launch behind attention_split.hip: only when that one flagged
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
    const int n_kt = (nk + 31) / 32;            // <= NKT, guaranteed by the launcher

    const int qrow = min(qt * 32 + l31, nq - 1);
    float qf[HALF];
    {
        const float* qp = q + (size_t)(q0 + qrow) * ldq + h * HD + hi * HALF;
#pragma unroll
        for (int i = 0; i < HALF; i += 4) {
            f32x4 t = *(const f32x4*)(qp + i);
            qf[i] = t[0] * scale; qf[i + 1] = t[1] * scale; qf[i + 2] = t[2] * scale; qf[i + 3] = t[3] * scale;
        }
    }
    f32x16 sc[NKT];
    float m = NEG_BIG;
    // the key rows of tile kt+1 are requested before the MFMAs of tile kt
    f32x4 ka[HALF / 4], kb[HALF / 4];
    const float* kbase = k + (size_t)k0 * ldk + h * HD + hi * HALF;
    auto fetch_k = [&](int kt, f32x4 (&kv)[HALF / 4]) {
        const float* kp = kbase + (size_t)min(min(kt, n_kt - 1) * 32 + l31, nk - 1) * ldk;
#pragma unroll
        for (int i = 0; i < HALF / 4; ++i) kv[i] = *(const f32x4*)(kp + 4 * i);
    };
    auto score = [&](int kt, const f32x4 (&kv)[HALF / 4]) {
        sc[kt] = zero16();
#pragma unroll
        for (int i = 0; i < HALF / 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) sc[kt] = mfma32(kv[i][c], qf[4 * i + c], sc[kt]);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (kt * 32 + mfma_row(r, hi) < nk) m = fmaxf(m, sc[kt][r]);
    };
    fetch_k(0, ka);
#pragma unroll
    for (int kt = 0; kt < NKT; kt += 2) {
        if (kt < n_kt) {
            fetch_k(kt + 1, kb);
            score(kt, ka);
        } else {
            sc[kt] = zero16();
        }
        if (kt + 1 < NKT) {
            if (kt + 1 < n_kt) {
                fetch_k(kt + 2, ka);
                score(kt + 1, kb);
            } else {
                sc[kt + 1] = zero16();
            }
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

    // P V.  Output column c of d-tile t is head column NTD*c + t, so that a lane fetches its NTD values of a key row with
    // ONE load (8 bytes for head_dim 64) and stores its outputs the same way; the V values of key tile kt+1 are requested
    // before the MFMAs of tile kt.
    typedef float vrow_t __attribute__((ext_vector_type(NTD)));
    f32x16 out[NTD];
#pragma unroll
    for (int t = 0; t < NTD; ++t) out[t] = zero16();
    const int dcol = (NTD * l31 < HD) ? NTD * l31 : 0;        // lanes past the head width (head_dim < 32) reuse column 0
    const float* vbase = v + (size_t)k0 * ldv + h * HD + dcol;
    vrow_t va[16], vb[16];
    auto fetch_v = [&](int kt, vrow_t (&vv)[16]) {
        kt = min(kt, n_kt - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = *(const vrow_t*)(vbase + (size_t)min(kt * 32 + mfma_row(r, hi), nk - 1) * ldv);
    };
    auto pv_tile = [&](int kt, const vrow_t (&vv)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = sc[kt][r] * inv;
            if (drop.thresh) p = drop_keep(drop, drop_att_idx(q0 + qrow, n_head, h, kt * 32 + mfma_row(r, hi))) ? p * drop.scale : 0.f;
#pragma unroll
            for (int t = 0; t < NTD; ++t) out[t] = mfma32(p, vv[r][t], out[t]);
        }
    };
    fetch_v(0, va);
#pragma unroll
    for (int kt = 0; kt < NKT; kt += 2) {
        if (kt < n_kt) {
            fetch_v(kt + 1, vb);
            pv_tile(kt, va);
        }
        if (kt + 1 < NKT && kt + 1 < n_kt) {
            fetch_v(kt + 2, va);
            pv_tile(kt + 1, vb);
        }
    }
    if (NTD * l31 < HD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = qt * 32 + mfma_row(r, hi);
            if (row < nq) {
                vrow_t ov;
#pragma unroll
                for (int t = 0; t < NTD; ++t) ov[t] = out[t][r];
                *(vrow_t*)(o + (size_t)(q0 + row) * ldo + h * HD + dcol) = ov;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void role_attention_kernel(const float* __restrict__ p2a, int ldp, const float* __restrict__ e2a, int lde,
                                                             float* __restrict__ att, const int32_t* __restrict__ e_off,
                                                             int n_vid, int nq, int n_qt, int half, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)blockIdx.x * (blockDim.x >> 6) + wave;          // one to four waves per workgroup (the launcher spreads small batches over more CUs)
    if (item >= (long)n_vid * n_qt) return;
    const int qt = item % n_qt, vid = item / n_qt;
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    if (qt * 32 >= nq || ne <= 0) return;
    const int qrow = min(qt * 32 + l31, nq - 1);
    const int hh = half / 2;  // k-range of one half-wave
    const float* qp = p2a + (size_t)(vid * nq + qrow) * ldp + hi * hh;
    const int n_et = (ne + 31) / 32;

    auto score_tile = [&](int et, f32x16& s0, f32x16& s1) {
        const int erow = min(et * 32 + l31, ne - 1);
        const float* ep = e2a + (size_t)(e0 + erow) * lde + hi * hh;
        s0 = zero16();
        s1 = zero16();
        for (int i = 0; i < hh; i += 4) {
            const f32x4 a0 = *(const f32x4*)(ep + i), a1 = *(const f32x4*)(ep + half + i);
            const f32x4 b0 = *(const f32x4*)(qp + i), b1 = *(const f32x4*)(qp + half + i);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s0 = mfma32(a0[c], b0[c], s0);
                s1 = mfma32(a1[c], b1[c], s1);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] *= scale; s1[r] *= scale; }
    };

    float m0 = NEG_BIG, l0 = 0.f, m1 = NEG_BIG, l1 = 0.f;
    for (int et = 0; et < n_et; ++et) {
        f32x16 s0, s1;
        score_tile(et, s0, s1);
        float t0 = NEG_BIG, t1 = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = et * 32 + mfma_row(r, hi) < ne;
            t0 = ok ? fmaxf(t0, s0[r]) : t0;
            t1 = ok ? fmaxf(t1, s1[r]) : t1;
        }
        const float n0 = fmaxf(m0, t0), n1 = fmaxf(m1, t1);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = et * 32 + mfma_row(r, hi) < ne;
            a0 += ok ? expf(s0[r] - n0) : 0.f;
            a1 += ok ? expf(s1[r] - n1) : 0.f;
        }
        l0 = l0 * expf(m0 - n0) + a0; m0 = n0;
        l1 = l1 * expf(m1 - n1) + a1; m1 = n1;
    }
    {
        const float mo = __shfl_xor(m0, 32), lo = __shfl_xor(l0, 32);
        const float M = fmaxf(m0, mo);
        l0 = l0 * expf(m0 - M) + lo * expf(mo - M); m0 = M;
    }
    {
        const float mo = __shfl_xor(m1, 32), lo = __shfl_xor(l1, 32);
        const float M = fmaxf(m1, mo);
        l1 = l1 * expf(m1 - M) + lo * expf(mo - M); m1 = M;
    }
    const int qq = qt * 32 + l31;
    float* base = att + (size_t)2 * nq * e0;
    for (int et = 0; et < n_et; ++et) {
        f32x16 s0, s1;
        score_tile(et, s0, s1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = et * 32 + mfma_row(r, hi);
            if (n < ne && qq < nq) {
                const float pe0 = expf(s0[r] - m0) / l0, pe1 = expf(s1[r] - m1) / l1;
                const float mx = fmaxf(s0[r], s1[r]);
                const float w0 = expf(s0[r] - mx), w1 = expf(s1[r] - mx);
                const float ws = w0 + w1;
                base[((size_t)0 * nq + qq) * ne + n] = pe0 * (w0 / ws);
                base[((size_t)1 * nq + qq) * ne + n] = pe1 * (w1 / ws);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void role_values_kernel(const float* __restrict__ att, const float* __restrict__ enco, int lde,
                                                          float* __restrict__ values, const int32_t* __restrict__ e_off,
                                                          int n_vid, int nq, int n_qt, int d, int n_dc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)n_vid * 2 * n_qt * n_dc) return;
    const int dc = item % n_dc;
    const int qt = (item / n_dc) % n_qt;
    const int role = (item / ((long)n_dc * n_qt)) % 2;
    const int vid = item / ((long)n_dc * n_qt * 2);
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    if (qt * 32 >= nq) return;
    const int qrow = min(qt * 32 + l31, nq - 1);
    const float* ap = att + (size_t)2 * nq * e0 + ((size_t)role * nq + qrow) * ne;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero16();
    const int d0 = dc * 128;
    for (int n2 = 0; n2 < ne; n2 += 2) {
        const int n = n2 + hi;
        const bool ok = n < ne;
        const float a_raw = ap[min(n, ne - 1)];
        const float a = ok ? a_raw : 0.f;
        const float* ep = enco + (size_t)(e0 + min(n, ne - 1)) * lde;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma32(a, ep[min(d0 + t * 32 + l31, d - 1)], acc[t]);   // clamped, never predicated
    }
    float* op = values + ((size_t)role * n_vid * nq + (size_t)vid * nq) * d;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = d0 + t * 32 + l31;
        if (col >= d) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = qt * 32 + mfma_row(r, hi);
            if (row < nq) op[(size_t)row * d + col] = acc[t][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// role attention AND its values in one pass (eval path): the scores of a 32-query tile against all entities of the
// video stay in registers (NET tiles of 32 entities per role), so there is no second scoring pass, and the
// probabilities feed the att_r @ enco_output MFMAs directly as in mha_core (models/model_0v10.py:193-211).
template <int NET>
__global__ __launch_bounds__(256) void role_fused_kernel(const float* __restrict__ p2a, int ldp, const float* __restrict__ e2a, int lde,
                                                         const float* __restrict__ enco, int ldenc, float* __restrict__ att,
                                                         float* __restrict__ values, const int32_t* __restrict__ e_off,
                                                         int n_vid, int nq, int n_qt, int half, int d, float scale,
                                                         const int* __restrict__ run_flag, int run_id,
                                                         long role_stride, const float* __restrict__ vbias0, const float* __restrict__ vbias1, int relu,
                                                         long out_role_stride, int ldval) {
    if (run_flag && *run_flag != run_id) return;        // fallback launch behind role_fused_h_kernel: only when that one flagged
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)n_vid * n_qt) return;
    const int qt = item % n_qt, vid = item / n_qt;
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    if (qt * 32 >= nq || ne <= 0) return;
    const int qrow = min(qt * 32 + l31, nq - 1);
    const int hh = half / 2;
    const float* qp = p2a + (size_t)(vid * nq + qrow) * ldp + hi * hh;
    const float* ep[NET];
#pragma unroll
    for (int et = 0; et < NET; ++et) ep[et] = e2a + (size_t)(e0 + min(et * 32 + l31, ne - 1)) * lde + hi * hh;

    f32x16 sc[2][NET];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int et = 0; et < NET; ++et) sc[r][et] = zero16();
    // A lane streams its own row of each operand along K.  It takes a whole 128-byte line (32 k) per burst of eight
    // dwordx4 loads, so that a line is consumed while it is still in the L1 (16 bytes per visit re-fetched every line from
    // L2: the working set of the four waves of a CU is several times the L1); the next burst is requested before the 32
    // MFMAs of the current one.
    constexpr int BURST = 32;
    f32x4 qa[8], ea[8], qb[8], eb[8];
    const int n_burst = hh / BURST;                 // half % 64 == 0 is checked by the launcher
    auto fetch = [&](int job, f32x4 (&qv)[8], f32x4 (&ev)[8]) {
        job = min(job, 2 * NET * n_burst - 1);
        const int kc = job % n_burst, et = (job / n_burst) % NET, r = job / (n_burst * NET);
        const float* q_ = qp + r * half + kc * BURST;
        const float* e_ = ep[0] + r * half + kc * BURST;
#pragma unroll
        for (int t = 1; t < NET; ++t) if (et == t) e_ = ep[t] + r * half + kc * BURST;
#pragma unroll
        for (int j = 0; j < 8; ++j) { qv[j] = *(const f32x4*)(q_ + 4 * j); ev[j] = *(const f32x4*)(e_ + 4 * j); }
    };
    auto fma_burst = [&](f32x16& acc, const f32x4 (&qv)[8], const f32x4 (&ev)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = mfma32(ev[j][c], qv[j][c], acc);
    };
    fetch(0, qa, ea);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int et = 0; et < NET; ++et) {
            const int job0 = (r * NET + et) * n_burst;
            for (int kc = 0; kc < n_burst; kc += 2) {
                fetch(job0 + kc + 1, qb, eb);
                fma_burst(sc[r][et], qa, ea);
                fetch(job0 + kc + 2, qa, ea);
                if (kc + 1 < n_burst) fma_burst(sc[r][et], qb, eb);
            }
        }
    // softmax over the entities of each role (torch.softmax: exp(s - max) / sum), then over the two roles
    float m[2] = {NEG_BIG, NEG_BIG}, l[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int et = 0; et < NET; ++et)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                sc[r][et][g] *= scale;
                if (et * 32 + mfma_row(g, hi) < ne) m[r] = fmaxf(m[r], sc[r][et][g]);
            }
        m[r] = fmaxf(m[r], __shfl_xor(m[r], 32));
#pragma unroll
        for (int et = 0; et < NET; ++et)
#pragma unroll
            for (int g = 0; g < 16; ++g)
                if (et * 32 + mfma_row(g, hi) < ne) l[r] += expf(sc[r][et][g] - m[r]);
        l[r] += __shfl_xor(l[r], 32);
    }
    const int qq = qt * 32 + l31;
    float* abase = att ? att + (size_t)2 * nq * e0 : nullptr;
#pragma unroll
    for (int et = 0; et < NET; ++et)
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
    // values_r = att_r @ enco_output, 128 output columns at a time.  Output column c of tile t is column d0 + 4c + t, so a
    // lane fetches its four values of an entity row with one dwordx4 load and stores its outputs the same way; loads are
    // unconditional from clamped addresses (a predicated load serialises load -> wait -> MFMA).
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float* op = values + r * out_role_stride + (size_t)vid * nq * ldval;     // default: role planes (2, n_vid * nq, d)
#pragma unroll 1
        for (int d0 = 0; d0 < d; d0 += 128) {
            const bool cok = d0 + 4 * l31 < d;            // d % 4 == 0 (checked by the launcher)
            const float* vb = enco + r * role_stride + (size_t)e0 * ldenc + (cok ? d0 + 4 * l31 : 0);     // role_stride != 0: per-role rows
            f32x16 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = zero16();
#pragma unroll
            for (int et = 0; et < NET; ++et) {
                f32x4 vv[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) vv[g] = *(const f32x4*)(vb + (size_t)min(et * 32 + mfma_row(g, hi), ne - 1) * ldenc);
#pragma unroll
                for (int g = 0; g < 16; ++g)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = mfma32(sc[r][et][g], vv[g][t], acc[t]);
            }
            if (cok) {
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (vbias0) bv = *(const f32x4*)((r ? vbias1 : vbias0) + d0 + 4 * l31);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = qt * 32 + mfma_row(g, hi);
                    f32x4 ov = f32x4{acc[0][g], acc[1][g], acc[2][g], acc[3][g]} + bv;
                    if (relu) { ov[0] = fmaxf(ov[0], 0.f); ov[1] = fmaxf(ov[1], 0.f); ov[2] = fmaxf(ov[2], 0.f); ov[3] = fmaxf(ov[3], 0.f); }
                    if (row < nq) *(f32x4*)(op + (size_t)row * ldval + d0 + 4 * l31) = ov;
                }
            }
        }
    }
}

}  // namespace

extern "C" int bigc_mha_core_drop_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                      float* o, int ldo, const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q,
                                      int n_head, int hd, float* lse, float drop_p, uint32_t drop_seed, uint32_t drop_site);

int bigc_mha_core_h16_launch(hipStream_t s, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                             const int32_t* q_off, const int32_t* k_off, int n_seq, int n_qt, int nkt, int n_head, float scale, float* lse,
                             const DropSpec& drop, int grid, int* flag, int run_id);
int bigc_mha_core_h_launch(hipStream_t s, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                           const int32_t* q_off, const int32_t* k_off, int n_seq, int n_qt, int n_head, float scale, float* lse,
                           const DropSpec& drop, int grid, int* flag, int run_id);

int bigc_role_fused_h_launch(hipStream_t s, const float* p2a, int ldp, const float* e2a, int lde, const float* enco, int ldenc, float* att,
                             float* values, const int32_t* e_off, int n_vid, int nq, int n_qt, int half, int d, float scale, int grid,
                             int* flag, int run_id, long role_stride, const float* vbias0, const float* vbias1, int relu, long out_role_stride, int ldval);

int bigc_mha_core_hd16_lds_launch(hipStream_t s, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                  const int32_t* off, int n_seq, int max_q, int n_head, float scale, int* flag, int run_id);
namespace {
// BIGC_LINEAR_EXACT / bigc_linear_set_exact also select the f32-MFMA attention kernels
std::atomic<int> g_mha_run{0};
constexpr int MHA_SLOTS = 1024;
std::mutex g_mha_mu;
int* g_mha_flags[64] = {};
// one flag word per launch (ring of MHA_SLOTS): the split kernel writes its run id there when an operand is out of range
int* mha_flag_slot(int run_id) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(g_mha_mu);
    if (!g_mha_flags[dev]) {
        if (hipMalloc((void**)&g_mha_flags[dev], MHA_SLOTS * sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(g_mha_flags[dev], 0, MHA_SLOTS * sizeof(int)) != hipSuccess) return nullptr;
    }
    return g_mha_flags[dev] + ((unsigned)run_id % MHA_SLOTS);      // (unsigned: the launch counter may wrap)
}
}  // namespace

extern "C" int bigc_mha_core_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                 float* o, int ldo, const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q,
                                 int n_head, int hd, float* lse) {
    return bigc_mha_core_drop_f32(stream, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, max_q, n_head, hd, lse, 0.f, 0, 0);
}

extern "C" int bigc_mha_core_drop_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                      float* o, int ldo, const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q,
                                      int n_head, int hd, float* lse, float drop_p, uint32_t drop_seed, uint32_t drop_site) {
    const DropSpec drop = make_drop(drop_p, drop_seed, drop_site);
    BIGC_CHECK_ARG(q && k && v && o && q_off && k_off, "mha_core: null pointer");
    BIGC_CHECK_ARG(hd == 8 || hd == 16 || hd == 32 || hd == 64, "mha_core: head_dim %d not in {8,16,32,64}", hd);
    BIGC_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0, "mha_core: ldq/ldk must be multiples of 4");
    if (n_seq <= 0 || max_q <= 0) return BIGC_OK;
    const int n_qt = cdiv(max_q, 32);
    const long items = (long)n_seq * n_qt * n_head;
    const int grid = (int)((items + 3) / 4);
    const float scale = 1.0f / sqrtf((float)hd);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(stream, FAM_MHA, 0.0, 0.0);
#define LAUNCH_MHA(HD) hipLaunchKernelGGL(mha_core_kernel<HD>, dim3(grid), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, lse, drop)
#define LAUNCH_SP(HD, NKT) hipLaunchKernelGGL((mha_core_sp_kernel<HD, NKT>), dim3(grid), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, lse, drop, run_flag, run_id)
    // self-attention (same offsets for queries and keys): the longest key sequence is max_q, known on the host ->
    // single-pass kernel with the scores in registers when it fits
    const int nkt = (q_off == k_off) ? n_qt : 1 << 30;
    // head_dim 64: split-operand f16 MFMA kernel first; the f32-MFMA kernel behind it runs only if an operand was out of range
    int* run_flag = nullptr;
    int run_id = 0;
    if (hd == 64 && nkt <= 6 && ldv % 2 == 0 && ldo % 2 == 0 && !bigc_linear_get_exact()) {
        run_id = g_mha_run.fetch_add(1, std::memory_order_relaxed) + 1;
        run_flag = mha_flag_slot(run_id);
        if (!run_flag) return BIGC_ERR_HIP;
        bigc_mha_core_h_launch(s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, lse, drop, grid, run_flag, run_id);
    }
    bool long16 = false;
    if (hd == 16 && !bigc_linear_get_exact() && q_off == k_off && !lse && !drop.thresh && max_q <= 1024) {
        // eval self-attention of the grounding encoders: K / V split once per (sequence, head) into LDS, the same chunked arithmetic for
        // every length (so a sequence's bits do not depend on its batch); the gated two-pass f32 kernel behind it is the range fallback
        run_id = g_mha_run.fetch_add(1, std::memory_order_relaxed) + 1;
        run_flag = mha_flag_slot(run_id);
        if (!run_flag) return BIGC_ERR_HIP;
        long16 = bigc_mha_core_hd16_lds_launch(s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, n_seq, max_q, n_head, scale, run_flag, run_id) != 0;
        if (!long16) { run_flag = nullptr; run_id = 0; }
    }
    if (long16) {
        hipLaunchKernelGGL(mha_core_kernel<16>, dim3(grid), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, lse, drop, run_flag, run_id);
        BIGC_LAUNCH_CHECK("mha_core");
        return BIGC_OK;
    }
    if (hd == 16 && !bigc_linear_get_exact() && (nkt <= 8 || (q_off == k_off && !lse && !drop.thresh))) {
        // (beyond 8 key tiles: the chunked kernel for eval self-attention -- no log-sum-exp output, no dropout; the two-pass f32
        // kernel behind it then only runs when an operand was out of range)
        run_id = g_mha_run.fetch_add(1, std::memory_order_relaxed) + 1;
        run_flag = mha_flag_slot(run_id);
        if (!run_flag) return BIGC_ERR_HIP;
        long16 = bigc_mha_core_h16_launch(s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, nkt, n_head, scale, lse, drop, grid, run_flag, run_id) != 0 && nkt > 8;
        if (nkt > 8 && !long16) { run_flag = nullptr; run_id = 0; }
    }
    if (hd == 64 && nkt <= 2) LAUNCH_SP(64, 2);
    else if (hd == 64 && nkt <= 6) LAUNCH_SP(64, 6);
    else if (hd == 16 && nkt <= 2) LAUNCH_SP(16, 2);
    else if (hd == 16 && nkt <= 8) LAUNCH_SP(16, 8);
    else if (hd == 8 && nkt <= 2) LAUNCH_SP(8, 2);
    else if (hd == 8 && nkt <= 8) LAUNCH_SP(8, 8);          // (same single-pass arithmetic as <8, 2>: a batch with a longer video must not change a shorter one's bits)
    else if (hd == 32 && nkt <= 6) LAUNCH_SP(32, 6);
    else
    switch (hd) {
        case 8: LAUNCH_MHA(8); break;
        case 16:
            if (long16) hipLaunchKernelGGL(mha_core_kernel<16>, dim3(grid), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, q_off, k_off, n_seq, n_qt, n_head, scale, lse, drop, run_flag, run_id);
            else LAUNCH_MHA(16);
            break;
        case 32: LAUNCH_MHA(32); break;
        default: LAUNCH_MHA(64); break;
    }
#undef LAUNCH_SP
#undef LAUNCH_MHA
    BIGC_LAUNCH_CHECK("mha_core");
    return BIGC_OK;
}

extern "C" int bigc_role_attention_f32(void* stream, const float* p2a, int ldp, const float* e2a, int lde, float* att,
                                       const int32_t* e_off, int n_vid, int nq, int half, float scale) {
    BIGC_CHECK_ARG(p2a && e2a && att && e_off, "role_attention: null pointer");
    BIGC_CHECK_ARG(half % 8 == 0 && ldp % 4 == 0 && lde % 4 == 0, "role_attention: half=%d must be a multiple of 8", half);
    if (n_vid <= 0) return BIGC_OK;
    const int n_qt = cdiv(nq, 32);
    // small batches (the training step at 8 videos: 48 items): one wave per workgroup, so that the items spread over 48 CUs instead of 12
    const int wpb = n_vid * n_qt <= 512 ? 1 : 4;
    const int grid = cdiv(n_vid * n_qt, wpb);
    ProfScope ps(stream, FAM_ROLE_ATT, 0.0, 0.0);
    hipLaunchKernelGGL(role_attention_kernel, dim3(grid), dim3(64 * wpb), 0, (hipStream_t)stream, p2a, ldp, e2a, lde, att, e_off,
                       n_vid, nq, n_qt, half, scale);
    BIGC_LAUNCH_CHECK("role_attention");
    return BIGC_OK;
}

extern "C" int bigc_role_values_f32(void* stream, const float* att, const float* enco, int lde, float* values,
                                    const int32_t* e_off, int n_vid, int nq, int d) {
    BIGC_CHECK_ARG(att && enco && values && e_off, "role_values: null pointer");
    if (n_vid <= 0) return BIGC_OK;
    const int n_qt = cdiv(nq, 32), n_dc = cdiv(d, 128);
    const long items = (long)n_vid * 2 * n_qt * n_dc;
    ProfScope ps(stream, FAM_ROLE_VAL, 0.0, 0.0);
    hipLaunchKernelGGL(role_values_kernel, dim3((int)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, att, enco, lde, values,
                       e_off, n_vid, nq, n_qt, d, n_dc);
    BIGC_LAUNCH_CHECK("role_values");
    return BIGC_OK;
}

static int role_fused_launch(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* enco, int ldenc, long role_stride,
                             const float* vbias0, const float* vbias1, int relu, float* att, float* values, const int32_t* e_off, int n_vid, int nq, int max_ent,
                             int half, int d, float scale, long out_role_stride = 0, int ldval = 0) {
    if (ldval == 0) { ldval = d; out_role_stride = (long)n_vid * nq * d; }          // role planes (2, n_vid * nq, d)
    BIGC_CHECK_ARG(p2a && e2a && enco && values && e_off, "role_fused: null pointer");
    BIGC_CHECK_ARG(half % 64 == 0 && ldp % 4 == 0 && lde % 4 == 0 && d % 4 == 0 && ldenc % 4 == 0, "role_fused: half=%d must be a multiple of 64, d=%d of 4", half, d);
    BIGC_CHECK_ARG(max_ent >= 1 && max_ent <= 64, "role_fused: at most 64 entities per video (got %d): use role_attention + role_values", max_ent);
    if (n_vid <= 0) return BIGC_OK;
    const int n_qt = cdiv(nq, 32);
    const int grid = cdiv(n_vid * n_qt, 4);
    ProfScope ps(stream, FAM_ROLE_ATT, 0.0, 0.0);
    // split-operand f16 MFMA kernel first; the f32-MFMA kernel behind it runs only if an operand was out of the f16 range
    int* run_flag = nullptr;
    int run_id = 0;
    const bool projected = role_stride != 0 || vbias0 || relu;
    if (d % 32 == 0 && !bigc_linear_get_exact() && (!projected || (n_qt <= 6 && half <= 256))) {
        run_id = g_mha_run.fetch_add(1, std::memory_order_relaxed) + 1;
        run_flag = mha_flag_slot(run_id);
        if (!run_flag) return BIGC_ERR_HIP;
        bigc_role_fused_h_launch((hipStream_t)stream, p2a, ldp, e2a, lde, enco, ldenc, att, values, e_off, n_vid, nq, n_qt, half, d, scale, grid,
                                 run_flag, run_id, role_stride, vbias0, vbias1, relu, out_role_stride, ldval);
    }
    if (max_ent <= 32)
        hipLaunchKernelGGL(role_fused_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p2a, ldp, e2a, lde, enco, ldenc, att, values,
                           e_off, n_vid, nq, n_qt, half, d, scale, run_flag, run_id, role_stride, vbias0, vbias1, relu, out_role_stride, ldval);
    else
        hipLaunchKernelGGL(role_fused_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p2a, ldp, e2a, lde, enco, ldenc, att, values,
                           e_off, n_vid, nq, n_qt, half, d, scale, run_flag, run_id, role_stride, vbias0, vbias1, relu, out_role_stride, ldval);
    BIGC_LAUNCH_CHECK("role_fused");
    return BIGC_OK;
}

extern "C" int bigc_role_fused_f32(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* enco, int ldenc,
                                   float* att, float* values, const int32_t* e_off, int n_vid, int nq, int max_ent, int half, int d,
                                   float scale) {
    return role_fused_launch(stream, p2a, ldp, e2a, lde, enco, ldenc, 0, nullptr, nullptr, 0, att, values, e_off, n_vid, nq, max_ent, half, d, scale);
}

extern "C" int bigc_role_fused_proj_f32(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* proj, int ldproj,
                                        long role_stride, const float* bias0, const float* bias1, int relu, float* att, float* values,
                                        long out_role_stride, int ldval, const int32_t* e_off, int n_vid, int nq, int max_ent, int half, int d,
                                        float scale) {
    BIGC_CHECK_ARG(role_stride % 4 == 0 && (bias0 == nullptr) == (bias1 == nullptr), "role_fused_proj: role_stride must be a multiple of 4, both biases or none");
    BIGC_CHECK_ARG(!bias0 || ((((uintptr_t)bias0) | ((uintptr_t)bias1)) & 15) == 0, "role_fused_proj: biases must be 16-byte aligned");
    BIGC_CHECK_ARG(ldval == 0 || (ldval >= d && ldval % 4 == 0 && out_role_stride % 4 == 0), "role_fused_proj: ldval=%d must be 0 or a multiple of 4 >= d", ldval);
    return role_fused_launch(stream, p2a, ldp, e2a, lde, proj, ldproj, role_stride, bias0, bias1, relu, att, values, e_off, n_vid, nq, max_ent, half, d, scale,
                             out_role_stride, ldval);
}
