// Backward of the attention kernels (attention.hip) on f32 MFMA, same one-wave-per-32-row-tile scheme:
//   mha_bwd_dq / mha_bwd_dkv   gradients of softmax(q k^T / sqrt(hd)) v, recomputing the probabilities from
//                              the saved log-sum-exp (torch autograd of nn.MultiheadAttention's core)
//   role_attention_bwd         d scores of att_r = softmax_n(s_r) * softmax_roles(s)_r  (models/model_0v10.py:201-207)
//   small_gemm                 the per-video products around it: d att = d vals @ enco^T, d enco += att^T @ d vals,
//                              d p2a = d s @ e2a, d e2a = d s^T @ p2a                      (:198-201,211)
#include "common.h"
#include "prof.h"

namespace {

constexpr float NEG_BIG = -1e30f;

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// delta[row, h] = sum_d dO[row, h*hd + d] * O[row, h*hd + d].  hd / 4 lanes per (row, head), one 16-byte load each, a butterfly over the
// group (hd in {8, 16, 32, 64}: groups of 2 .. 16 lanes): coalesced lines instead of one lane walking a 256-byte row segment
__global__ __launch_bounds__(256) void mha_delta_kernel(const float* __restrict__ dO, int lddo, const float* __restrict__ O, int ldo,
                                                        float* __restrict__ delta, long rows, int n_head, int hd, float* __restrict__ amax_slot) {
    const int g = hd >> 2;                                   // lanes per (row, head)
    const long items = rows * n_head;
    float am = 0.f;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < items * g; t += (long)gridDim.x * 256) {
        const long i = t / g;
        const int c = (int)(t - i * g);
        const long r = i / n_head;
        const int h = (int)(i - r * n_head);
        const f32x4 a = *(const f32x4*)(dO + r * lddo + h * hd + 4 * c), b = *(const f32x4*)(O + r * ldo + h * hd + 4 * c);
        float s = (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
        for (int o = g >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (c == 0) delta[i] = s;
        am = fmaxf(am, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
        if (a[0] != a[0] || a[1] != a[1] || a[2] != a[2] || a[3] != a[3]) am = INFINITY;
    }
    // max |dO| for the split kernels' gradient scale (attention_bwd_split.hip); bit patterns of non-negative floats order like the floats
    // (one atomic per workgroup: 16 k same-address atomics, one per wave, cost 130 us)
    if (amax_slot) {
        __shared__ float wam[4];
        for (int o = 32; o >= 1; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
        if ((threadIdx.x & 63) == 0) wam[threadIdx.x >> 6] = am;
        __syncthreads();
        if (threadIdx.x == 0) {
            am = fmaxf(fmaxf(wam[0], wam[1]), fmaxf(wam[2], wam[3]));
            if (am > 0.f) atomicMax((unsigned int*)amax_slot, __float_as_uint(am));
        }
    }
}

template <int HD>
__device__ __forceinline__ void mha_bwd_dq_body(const int bid, const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                const float* __restrict__ v, int ldv, const float* __restrict__ dO, int lddo,
                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                float* __restrict__ dq, int lddq, const int32_t* __restrict__ q_off,
                                                const int32_t* __restrict__ k_off, int n_seq, int n_qt, int n_head, float scale,
                                                const DropSpec drop, const int* __restrict__ run_flag, int run_id) {
    if (run_flag && *run_flag != run_id) return;        // fallback launch behind the split-operand kernels: only when they flagged
    constexpr int HALF = HD / 2;
    constexpr int NTD = (HD + 31) / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)bid * 4 + wave;
    if (item >= (long)n_seq * n_qt * n_head) return;
    const int h = item % n_head;
    const int qt = (item / n_head) % n_qt;
    const int s = item / ((long)n_head * n_qt);
    const int q0 = q_off[s], nq = q_off[s + 1] - q0;
    const int k0 = k_off[s], nk = k_off[s + 1] - k0;
    if (qt * 32 >= nq || nk <= 0) return;
    const int qrow = min(qt * 32 + l31, nq - 1);
    const bool qok = qt * 32 + l31 < nq;
    float qf[HALF], dof[HALF];
    {
        const float* qp = q + (size_t)(q0 + qrow) * ldq + h * HD + hi * HALF;
        const float* gp = dO + (size_t)(q0 + qrow) * lddo + h * HD + hi * HALF;
#pragma unroll
        for (int i = 0; i < HALF; ++i) { qf[i] = qp[i] * scale; dof[i] = gp[i]; }
    }
    const float my_lse = lse[(size_t)(q0 + qrow) * n_head + h], my_delta = delta[(size_t)(q0 + qrow) * n_head + h];
    f32x16 acc[NTD];
#pragma unroll
    for (int t = 0; t < NTD; ++t) acc[t] = zero16();
    const int n_kt = (nk + 31) / 32;
    for (int kt = 0; kt < n_kt; ++kt) {
        const int krow = min(kt * 32 + l31, nk - 1);
        const float* kp = k + (size_t)(k0 + krow) * ldk + h * HD + hi * HALF;
        const float* vp = v + (size_t)(k0 + krow) * ldv + h * HD + hi * HALF;
        f32x16 sc = zero16(), dp = zero16();
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            sc = mfma32(kp[i], qf[i], sc);
            dp = mfma32(vp[i], dof[i], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + mfma_row(r, hi);
            const bool ok = key < nk && qok;
            const float p = ok ? expf(sc[r] - my_lse) : 0.f;
            float dpr = dp[r];                                   // d of the (dropped) probability
            if (drop.thresh) dpr = drop_keep(drop, drop_att_idx(q0 + qrow, n_head, h, key)) ? dpr * drop.scale : 0.f;
            const float ds = p * (dpr - my_delta);
            const float* kr = k + (size_t)(k0 + min(key, nk - 1)) * ldk + h * HD;
#pragma unroll
            for (int t = 0; t < NTD; ++t) {
                const int d = t * 32 + l31;
                acc[t] = mfma32(ds, kr[d < HD ? d : HD - 1], acc[t]);
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
            if (row < nq) dq[(size_t)(q0 + row) * lddq + h * HD + d] = acc[t][r] * scale;
        }
    }
}

template <int HD>
__device__ __forceinline__ void mha_bwd_dkv_body(const int bid, const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                 const float* __restrict__ v, int ldv, const float* __restrict__ dO, int lddo,
                                                 const float* __restrict__ lse, const float* __restrict__ delta,
                                                 float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv,
                                                 const int32_t* __restrict__ q_off, const int32_t* __restrict__ k_off,
                                                 int n_seq, int n_kt, int n_head, float scale, const DropSpec drop,
                                                 const int* __restrict__ run_flag, int run_id) {
    if (run_flag && *run_flag != run_id) return;
    constexpr int HALF = HD / 2;
    constexpr int NTD = (HD + 31) / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)bid * 4 + wave;
    if (item >= (long)n_seq * n_kt * n_head) return;
    const int h = item % n_head;
    const int kt = (item / n_head) % n_kt;
    const int s = item / ((long)n_head * n_kt);
    const int q0 = q_off[s], nq = q_off[s + 1] - q0;
    const int k0 = k_off[s], nk = k_off[s + 1] - k0;
    if (kt * 32 >= nk) return;
    const int krow = min(kt * 32 + l31, nk - 1);
    const bool kok = kt * 32 + l31 < nk;
    float kf[HALF], vf[HALF];
    {
        const float* kp = k + (size_t)(k0 + krow) * ldk + h * HD + hi * HALF;
        const float* vp = v + (size_t)(k0 + krow) * ldv + h * HD + hi * HALF;
#pragma unroll
        for (int i = 0; i < HALF; ++i) { kf[i] = kp[i]; vf[i] = vp[i]; }
    }
    f32x16 ak[NTD], av[NTD];
#pragma unroll
    for (int t = 0; t < NTD; ++t) { ak[t] = zero16(); av[t] = zero16(); }
    const int n_qt = (nq + 31) / 32;
    for (int qt = 0; qt < n_qt; ++qt) {
        const int qrow = min(qt * 32 + l31, nq - 1);
        const float* qp = q + (size_t)(q0 + qrow) * ldq + h * HD + hi * HALF;
        const float* gp = dO + (size_t)(q0 + qrow) * lddo + h * HD + hi * HALF;
        f32x16 sc = zero16(), dp = zero16();
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            sc = mfma32(qp[i] * scale, kf[i], sc);     // rows = queries, cols = keys (this lane's key)
            dp = mfma32(gp[i], vf[i], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = qt * 32 + mfma_row(r, hi);
            const int qc = min(qi, nq - 1);
            const bool ok = qi < nq && kok;
            const float lr = lse[(size_t)(q0 + qc) * n_head + h], dr = delta[(size_t)(q0 + qc) * n_head + h];
            const float p = ok ? expf(sc[r] - lr) : 0.f;
            float pd = p, dpr = dp[r];                           // dropped probability (for dV) and d of it (for dS)
            if (drop.thresh) {
                const bool keep = drop_keep(drop, drop_att_idx(q0 + qc, n_head, h, kt * 32 + l31));
                pd = keep ? p * drop.scale : 0.f;
                dpr = keep ? dpr * drop.scale : 0.f;
            }
            const float ds = p * (dpr - dr);
            const float* gor = dO + (size_t)(q0 + qc) * lddo + h * HD;
            const float* qr = q + (size_t)(q0 + qc) * ldq + h * HD;
#pragma unroll
            for (int t = 0; t < NTD; ++t) {
                const int d = t * 32 + l31;
                av[t] = mfma32(pd, gor[d < HD ? d : HD - 1], av[t]);
                ak[t] = mfma32(ds, qr[d < HD ? d : HD - 1], ak[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NTD; ++t) {
        const int d = t * 32 + l31;
        if (d >= HD) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = kt * 32 + mfma_row(r, hi);
            if (row < nk) {
                dk[(size_t)(k0 + row) * lddk + h * HD + d] = ak[t][r] * scale;
                dv[(size_t)(k0 + row) * lddv + h * HD + d] = av[t][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// d scores of the role attention, already multiplied by `scale` (= d of the raw dot products).
// dQ and dK / dV in ONE launch: workgroups [0, gq) take the query tiles, [gq, gq + gk) the key tiles (the gated fallback behind the
// split-operand kernel, and the exact mode / the other head dims: one launch where there were two)
template <int HD>
__global__ __launch_bounds__(256) void mha_bwd_both_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                           const float* __restrict__ v, int ldv, const float* __restrict__ dO, int lddo,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dq, int lddq, float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv,
                                                           const int32_t* __restrict__ q_off, const int32_t* __restrict__ k_off,
                                                           int n_seq, int n_qt, int n_kt, int gq, int n_head, float scale, const DropSpec drop,
                                                           const int* __restrict__ run_flag, int run_id) {
    if ((int)blockIdx.x < gq)
        mha_bwd_dq_body<HD>((int)blockIdx.x, q, ldq, k, ldk, v, ldv, dO, lddo, lse, delta, dq, lddq, q_off, k_off, n_seq, n_qt, n_head, scale, drop, run_flag, run_id);
    else
        mha_bwd_dkv_body<HD>((int)blockIdx.x - gq, q, ldq, k, ldk, v, ldv, dO, lddo, lse, delta, dk, lddk, dv, lddv, q_off, k_off, n_seq, n_kt, n_head, scale, drop,
                             run_flag, run_id);
}

// Videos of at most 64 entities (round 6): one workgroup of four waves per (video, query tile).  Wave (role, entity tile) computes ITS
// 32 x 32 score tile (a chain of half dependent f32 MFMAs -- the four tiles run side by side instead of one after the other on one wave)
// and leaves it in LDS as S[role][query][entity]; then wave w takes the query rows w, w + 4, ... with lane = entity: the softmax
// statistics of a row are wave reductions and d att is read, d scores written, as whole 256-byte rows (with lane = query every load
// instruction touched 64 cache lines).  Which form runs depends on the video's own entity count only.
__global__ __launch_bounds__(256) void role_attention_bwd_tiles_kernel(const float* __restrict__ p2a, int ldp, const float* __restrict__ e2a, int lde,
                                                                       const float* __restrict__ datt, float* __restrict__ ds,
                                                                       const int32_t* __restrict__ e_off, int n_vid, int nq, int n_qt, int half,
                                                                       float scale) {
    constexpr int SP = 68;                                  // floats per query row of S (64 entities + pad)
    __shared__ __attribute__((aligned(16))) float S[2][32][SP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int item = blockIdx.x;
    const int qt = item % n_qt, vid = item / n_qt;
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    if (qt * 32 >= nq || ne <= 0 || ne > 64) return;        // (uniform; ne > 64: role_attention_bwd_kernel, launched beside this one)
    {
        const int role = wave >> 1, et = wave & 1;
        if (et * 32 < ne) {
            const int hh = half / 2;
            const int qrow = min(qt * 32 + l31, nq - 1);
            const int erow = min(et * 32 + l31, ne - 1);
            const float* qp = p2a + (size_t)(vid * nq + qrow) * ldp + role * half + hi * hh;
            const float* ep = e2a + (size_t)(e0 + erow) * lde + role * half + hi * hh;
            f32x16 acc = zero16();
            for (int i = 0; i < hh; i += 16) {              // sixteen k of both operands in flight per round
                f32x4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[u] = *(const f32x4*)(ep + i + 4 * u); b[u] = *(const f32x4*)(qp + i + 4 * u); }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc = mfma32(a[u][c], b[u][c], acc);
            }
            // register r of lane (l31, hi) = S[entity et * 32 + mfma_row(r, hi)][query l31]: four consecutive entities per register quad
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *(f32x4*)&S[role][l31][et * 32 + 8 * r4 + 4 * hi] = f32x4{acc[4 * r4] * scale, acc[4 * r4 + 1] * scale, acc[4 * r4 + 2] * scale, acc[4 * r4 + 3] * scale};
        }
    }
    __syncthreads();
    const float* gbase = datt + (size_t)2 * nq * e0;
    float* dbase = ds + (size_t)2 * nq * e0;
    const bool nok = lane < ne;
    for (int ql = wave; ql < 32; ql += 4) {
        const int qq = qt * 32 + ql;
        if (qq >= nq) break;
        const float s0 = nok ? S[0][ql][lane] : NEG_BIG, s1 = nok ? S[1][ql][lane] : NEG_BIG;
        float m0 = s0, m1 = s1;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); }
        const float x0 = nok ? expf(s0 - m0) : 0.f, x1 = nok ? expf(s1 - m1) : 0.f;
        float l0 = x0, l1 = x1;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { l0 += __shfl_xor(l0, o); l1 += __shfl_xor(l1, o); }
        const float E0 = x0 / l0, E1 = x1 / l1;
        const float mx = fmaxf(s0, s1);
        const float w0 = expf(s0 - mx), w1 = expf(s1 - mx);
        const float R0 = w0 / (w0 + w1), R1 = w1 / (w0 + w1);
        const float ga0 = nok ? gbase[((size_t)0 * nq + qq) * ne + lane] * E0 * R0 : 0.f;
        const float ga1 = nok ? gbase[((size_t)1 * nq + qq) * ne + lane] * E1 * R1 : 0.f;
        float c0 = ga0, c1 = ga1;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { c0 += __shfl_xor(c0, o); c1 += __shfl_xor(c1, o); }
        const float both = ga0 + ga1;
        if (nok) {
            dbase[((size_t)0 * nq + qq) * ne + lane] = (2.f * ga0 - E0 * c0 - R0 * both) * scale;
            dbase[((size_t)1 * nq + qq) * ne + lane] = (2.f * ga1 - E1 * c1 - R1 * both) * scale;
        }
    }
}

// The general form (any entity count): one wave per (video, query tile), lane = query, registers = entities.  `skip_small`: videos of
// at most 64 entities are left to role_attention_bwd_tiles_kernel.
__global__ __launch_bounds__(256) void role_attention_bwd_kernel(const float* __restrict__ p2a, int ldp, const float* __restrict__ e2a, int lde,
                                                                 const float* __restrict__ datt, float* __restrict__ ds,
                                                                 const int32_t* __restrict__ e_off, int n_vid, int nq, int n_qt, int half,
                                                                 float scale, int skip_small) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long item = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (item >= (long)n_vid * n_qt) return;
    const int qt = item % n_qt, vid = item / n_qt;
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    if (qt * 32 >= nq || ne <= 0 || (skip_small && ne <= 64)) return;
    const int qq = qt * 32 + l31;
    const int qrow = min(qq, nq - 1);
    const int hh = half / 2;
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

    // videos of at most 64 entities: the score tiles of both roles are computed ONCE and kept in registers for the three passes below
    const bool cached = n_et <= 2;
    f32x16 cs0[2], cs1[2];
    if (cached) {
        score_tile(0, cs0[0], cs1[0]);
        if (n_et > 1) score_tile(1, cs0[1], cs1[1]); else { cs0[1] = zero16(); cs1[1] = zero16(); }
    }
    auto tile = [&](int et, f32x16& s0, f32x16& s1) {
        if (cached) { s0 = et == 0 ? cs0[0] : cs0[1]; s1 = et == 0 ? cs1[0] : cs1[1]; }
        else score_tile(et, s0, s1);
    };
    // pass 1: softmax-over-entities statistics of both roles
    float m0 = NEG_BIG, l0 = 0.f, m1 = NEG_BIG, l1 = 0.f;
    for (int et = 0; et < n_et; ++et) {
        f32x16 s0, s1;
        tile(et, s0, s1);
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
        float mo = __shfl_xor(m0, 32), lo = __shfl_xor(l0, 32), M = fmaxf(m0, mo);
        l0 = l0 * expf(m0 - M) + lo * expf(mo - M); m0 = M;
        mo = __shfl_xor(m1, 32); lo = __shfl_xor(l1, 32); M = fmaxf(m1, mo);
        l1 = l1 * expf(m1 - M) + lo * expf(mo - M); m1 = M;
    }
    const float* gbase = datt + (size_t)2 * nq * e0;
    float* dbase = ds + (size_t)2 * nq * e0;
    // pass 2: c_r = sum_n g_r * att_r
    float c0 = 0.f, c1 = 0.f;
    for (int et = 0; et < n_et; ++et) {
        f32x16 s0, s1;
        tile(et, s0, s1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = et * 32 + mfma_row(r, hi);
            if (n < ne && qq < nq) {
                const float E0 = expf(s0[r] - m0) / l0, E1 = expf(s1[r] - m1) / l1;
                const float mx = fmaxf(s0[r], s1[r]);
                const float w0 = expf(s0[r] - mx), w1 = expf(s1[r] - mx);
                const float R0 = w0 / (w0 + w1), R1 = w1 / (w0 + w1);
                c0 += gbase[((size_t)0 * nq + qq) * ne + n] * E0 * R0;
                c1 += gbase[((size_t)1 * nq + qq) * ne + n] * E1 * R1;
            }
        }
    }
    c0 += __shfl_xor(c0, 32);
    c1 += __shfl_xor(c1, 32);
    // pass 3: ds_r = 2 g_r att_r - E_r c_r - R_r (g_0 att_0 + g_1 att_1)
    for (int et = 0; et < n_et; ++et) {
        f32x16 s0, s1;
        tile(et, s0, s1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = et * 32 + mfma_row(r, hi);
            if (n < ne && qq < nq) {
                const float E0 = expf(s0[r] - m0) / l0, E1 = expf(s1[r] - m1) / l1;
                const float mx = fmaxf(s0[r], s1[r]);
                const float w0 = expf(s0[r] - mx), w1 = expf(s1[r] - mx);
                const float R0 = w0 / (w0 + w1), R1 = w1 / (w0 + w1);
                const float ga0 = gbase[((size_t)0 * nq + qq) * ne + n] * E0 * R0;
                const float ga1 = gbase[((size_t)1 * nq + qq) * ne + n] * E1 * R1;
                const float both = ga0 + ga1;
                dbase[((size_t)0 * nq + qq) * ne + n] = (2.f * ga0 - E0 * c0 - R0 * both) * scale;
                dbase[((size_t)1 * nq + qq) * ne + n] = (2.f * ga1 - E1 * c1 - R1 * both) * scale;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-video small GEMMs C[i,j] (+)= sum_k A[i,k] B[k,j] with arbitrary strides; one wave per
// (problem, 32-row tile, 128-column chunk).
enum SmallGemmMode { SG_DATT = 0, SG_DENCO = 1, SG_DP2A = 2, SG_DE2A = 3 };

struct SmallGemmArgs {
    int mode; int n_vid; int nq; int d; int half;
    const float* att; const float* dvals; const float* enco; int lde;      // DATT / DENCO
    const float* ds; const float* p2a; int ldp; const float* e2a; int lda;  // DP2A / DE2A
    float* out; int ldo; const int32_t* e_off; int max_rows; int n_cc;
    int narrow;                                                              // every video has at most 64 columns (role_bgemm_kernel)
};

// Two of the four products in one launch (blockIdx.y selects): DATT with DENCO and DP2A with DE2A are independent of each other and each
// fills a fraction of the chip at training batch sizes (8 videos: a few hundred workgroups of dependent loads and MFMAs).
__global__ __launch_bounds__(256) void small_gemm_kernel(const SmallGemmArgs p0, const SmallGemmArgs p1) {
    const SmallGemmArgs& p = blockIdx.y ? p1 : p0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // problem enumeration: (video, role) for DATT/DP2A/DE2A, (video) for DENCO (roles are summed inside)
    const int n_role = p.mode == SG_DENCO ? 1 : 2;
    const int n_rt = (p.max_rows + 31) / 32;
    const int n_cc = p.n_cc;       // 128-column chunks (host: from the largest column count over the videos)
    // one workgroup per tile; its four waves split the reduction range and meet in LDS (these products are a few hundred waves
    // of dependent f32 MFMAs in total: what they need is more waves, not more flops per wave)
    __shared__ float red[3][4][16][64];
    const long item = blockIdx.x;
    if (item >= (long)p.n_vid * n_role * n_rt * n_cc) return;
    const int cc = item % n_cc;
    const int rt = (item / n_cc) % n_rt;
    const int role = (item / ((long)n_cc * n_rt)) % n_role;
    const int vid = item / ((long)n_cc * n_rt * n_role);
    const int e0 = p.e_off[vid], ne = p.e_off[vid + 1] - e0;
    if (ne <= 0) return;
    const int nq = p.nq;
    const size_t abase = (size_t)2 * nq * e0;

    // C is m x n, reduction length kdim; element strides of A (i,k), B (k,j) and C rows
    int m, n, kdim, sa_i, sa_k, sb_k, sb_j, sc_i, n_pass = 1;
    const float *A, *B;
    float* C;
    bool accumulate = false;
    if (p.mode == SG_DATT) {           // datt[r] (nq x ne) = dvals[r][vid] (nq x d) @ enco_v^T (d x ne)
        m = nq; n = ne; kdim = p.d;
        A = p.dvals + ((size_t)role * p.n_vid * nq + (size_t)vid * nq) * p.lde; sa_i = p.lde; sa_k = 1;
        B = p.enco + (size_t)e0 * p.lde; sb_k = 1; sb_j = p.lde;
        C = p.out + abase + (size_t)role * nq * ne; sc_i = ne;
    } else if (p.mode == SG_DENCO) {   // denco_v (ne x d) += sum_r att[r]^T (ne x nq) @ dvals[r][vid] (nq x d)
        m = ne; n = p.d; kdim = nq; n_pass = 2; accumulate = true;
        A = p.att + abase; sa_i = 1; sa_k = ne;
        B = p.dvals + (size_t)vid * nq * p.lde; sb_k = p.lde; sb_j = 1;
        C = p.out + (size_t)e0 * p.ldo; sc_i = p.ldo;
    } else if (p.mode == SG_DP2A) {    // dp2a[vid*nq + q, role*half + c] = sum_n ds[r][q][n] e2a[e0+n, role*half + c]
        m = nq; n = p.half; kdim = ne;
        A = p.ds + abase + (size_t)role * nq * ne; sa_i = ne; sa_k = 1;
        B = p.e2a + (size_t)e0 * p.lda + role * p.half; sb_k = p.lda; sb_j = 1;
        C = p.out + (size_t)vid * nq * p.ldo + role * p.half; sc_i = p.ldo;
    } else {                           // SG_DE2A: de2a[e0+n, role*half + c] = sum_q ds[r][q][n] p2a[vid*nq + q, role*half + c]
        m = ne; n = p.half; kdim = nq;
        A = p.ds + abase + (size_t)role * nq * ne; sa_i = 1; sa_k = ne;
        B = p.p2a + (size_t)vid * nq * p.ldp + role * p.half; sb_k = p.ldp; sb_j = 1;
        C = p.out + (size_t)e0 * p.ldo + role * p.half; sc_i = p.ldo;
    }
    if (rt * 32 >= m || cc * 128 >= n) return;
    const int irow = min(rt * 32 + l31, m - 1);
    const bool iok = rt * 32 + l31 < m;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero16();
    const int j0 = cc * 128;
    size_t joff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) joff[t] = (size_t)min(j0 + t * 32 + l31, n - 1) * sb_j;
    // The loop is a chain of dependent load -> MFMA steps on a handful of waves per video: what it costs is load latency, so
    // each round issues the loads of SG_U k-pairs (5 per pair) before the first MFMA consumes any.
    constexpr int SG_U = 8;
    const int kq = ((kdim + 3) / 4 + 1) / 2 * 2;            // this wave's share of the reduction range (even: k pairs)
    const int kb = wave * kq, ke = min(kdim, kb + kq);
    int nt = 0;                                             // live 32-column tiles of this 128-column chunk
#pragma unroll
    for (int t = 0; t < 4; ++t) nt += (j0 + t * 32 < n) ? 1 : 0;
    // Both operands contiguous along the reduction (DATT: rows of dvals and of enco, 512 floats each): a lane reads 16 bytes = four k of
    // its row per load instead of one float per k -- with one float per load every instruction touched 32 different cache lines for 4
    // useful bytes each, and this product was the slowest of the four (105 us per decoder layer at 64 videos).  The two k-pairs of a
    // quad go to two MFMAs; hi selects the k of its half-wave as before.
    if (p.mode == SG_DATT && (p.lde & 3) == 0 && (kdim & 15) == 0) {
        const float* Ap = A + (size_t)irow * sa_i;
        constexpr int QU = 4;                                   // quads in flight per round
        const int kq4 = kdim / 4;                               // this wave's quarter of the reduction, in whole quads
        const int qb = wave * kq4, qe = qb + kq4;
        for (int k4 = qb; k4 < qe; k4 += 4 * QU) {
            f32x4 a4[QU], b4[QU][4];
#pragma unroll
            for (int u = 0; u < QU; ++u) {
                const int kk = min(k4 + 4 * u, kdim - 4);
                a4[u] = *(const f32x4*)(Ap + kk);
                if (!(iok && k4 + 4 * u < qe)) a4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) if (t < nt) b4[u][t] = *(const f32x4*)(B + joff[t] + kk);
            }
#pragma unroll
            for (int u = 0; u < QU; ++u)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const float av = hi ? a4[u][2 * pr + 1] : a4[u][2 * pr];
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (t < nt) acc[t] = mfma32(av, hi ? b4[u][t][2 * pr + 1] : b4[u][t][2 * pr], acc[t]);
                }
        }
        n_pass = 0;                                             // (skips the generic loop below)
    }
    for (int pass = 0; pass < n_pass; ++pass) {
        const float* Ap = A + (p.mode == SG_DENCO ? (size_t)pass * nq * ne : 0) + (size_t)irow * sa_i;
        const float* Bp = B + (p.mode == SG_DENCO ? (size_t)pass * p.n_vid * nq * p.lde : 0);
        for (int k2 = kb; k2 < ke; k2 += 2 * SG_U) {
            // loads are unconditional from clamped addresses (a predicated load serialises load -> wait -> MFMA): a = 0
            // cancels the k tail, columns j >= n are never stored
            float a[SG_U], b[SG_U][4];
#pragma unroll
            for (int u = 0; u < SG_U; ++u) {
                const int kk = k2 + 2 * u + hi, kc = min(kk, kdim - 1);
                const float a_raw = Ap[(size_t)kc * sa_k];
                a[u] = (iok && kk < ke) ? a_raw : 0.f;
                const float* bp = Bp + (size_t)kc * sb_k;
#pragma unroll
                for (int t = 0; t < 4; ++t) if (t < nt) b[u][t] = bp[joff[t]];
            }
#pragma unroll
            for (int u = 0; u < SG_U; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) if (t < nt) acc[t] = mfma32(a[u], b[u][t], acc[t]);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][t][r][lane] = acc[t][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = j0 + t * 32 + l31;
        if (j >= n) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = rt * 32 + mfma_row(r, hi);
            if (i >= m) continue;
            const float v = ((acc[t][r] + red[0][t][r][lane]) + red[1][t][r][lane]) + red[2][t][r][lane];
            float* cp = C + (size_t)i * sc_i + j;
            *cp = accumulate ? *cp + v : v;
        }
    }
}

// ---- the same four per-video products with the operands staged in LDS (round 6) -------------------------------------------------------------
// small_gemm_kernel above feeds its f32 MFMAs with one 4-byte global load per lane and operand element: 40 load instructions per round of 32
// MFMAs, a chain of load -> wait -> MFMA rounds on waves that sit alone on their SIMDs (rocprofv3 --pmc at 8 videos: 71 % of the wave cycles
// waiting, matrix pipe 14 % busy; more loads in flight or more workgroups did not change that).  Here a workgroup (four waves) owns a 32-row x
// BN-column tile of C and walks K in slabs of KS: the 32 x KS slab of A and the BN x KS slab of B go to LDS through coalesced 16-byte global
// loads -- along k where k is contiguous in memory (dvals / enco rows in DATT, ds rows in DP2A), along the row index where that is (att^T,
// ds^T, dvals, e2a, p2a) -- the next slab's loads in flight (registers) while the current one multiplies.  What bounds these launches at training sizes
// is the chain of global-load round trips of a workgroup that is alone on its CU, so the slabs are deep: 64 k for 128-column tiles (wave w
// multiplies columns 32 w .. 32 w + 31), 128 k for tiles of at most 64 live columns (DATT: the video's entities; wave pairs split each slab's
// k and add their halves through LDS at the end).  DATT's K = 512 is four round trips instead of sixteen 32-k rounds.  Exact f32 arithmetic
// as before (gradients of any magnitude, no range flag); the order of the K sum differs from the wave-split form above.
constexpr int BG_LDS_FLOATS = (32 + 64) * (128 + 4);     // the largest of the forms below (DATT, 64-column tiles)
template <int BN, int KS, bool A_KC, bool B_KC>
__device__ __forceinline__ void role_bgemm_body(const float* A, const float* B, float* C, int m, int n, int kdim, int sa_i, int sa_k, int sb_k,
                                                int sb_j, int sc_i, int n_pass, size_t a_pass, size_t b_pass, bool accumulate, int i0, int j0,
                                                float* lds) {
    // LDS images: an operand whose k is contiguous in memory is kept as rows of KS k (pitch KS + 4 floats: the MFMA loop reads 16 bytes = four k
    // per lane, lane = row), one whose row index is contiguous as KS rows of 32 / BN row values (the loop reads 4 bytes per
    // lane and k); either way a staging item is 16 contiguous bytes of memory and of LDS.
    // (the 4-byte reads take k and k + 1 in the two half-waves: pitches of 32 mod 64 floats keep them on different banks)
    constexpr int PA = A_KC ? KS + 4 : 32, PB = B_KC ? KS + 4 : BN + 32;
    constexpr int A_FLOATS = A_KC ? 32 * PA : KS * PA;
    static_assert(A_FLOATS + (B_KC ? BN : KS) * PB <= BG_LDS_FLOATS, "role_bgemm: LDS");
    constexpr int NA = 32 * KS / 4 / 256, NB = BN * KS / 4 / 256;          // staging items per thread
    constexpr int QA = A_KC ? KS / 4 : 8, QB = B_KC ? KS / 4 : BN / 4;     // 16-byte items per LDS row
    constexpr bool KSPLIT = BN == 64;
    float* As = lds;
    float* Bs = lds + A_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int ncols = min(BN, n - j0);
    const int nb = KSPLIT ? (wave & 1) : wave;                // this wave's 32-column block
    const int kh = KSPLIT ? (wave >> 1) : 0;                  // and (64-column tiles) its half of every slab's k
    const bool active = nb * 32 < ncols;
    const bool a_vec = ((A_KC ? sa_i : sa_k) & 3) == 0 && (((uintptr_t)A) & 15) == 0 && (a_pass & 3) == 0;
    const bool b_vec = ((B_KC ? sb_j : sb_k) & 3) == 0 && (((uintptr_t)B) & 15) == 0 && (b_pass & 3) == 0;
    const int spp = (kdim + KS - 1) / KS, n_slab = n_pass * spp;
    f32x4 pa[NA], pb[NB];
    // 16 bytes of an operand: KC = row `r`, k0 .. k0 + 3; else k = `k`, rows r0 .. r0 + 3 (r_lim rows, stride_o = the other index's stride)
    auto fetch_kc = [&](const float* base, int r, int r_lim, int stride_r, bool vec, int k0) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r >= r_lim || k0 >= kdim) return v;
        const float* src = base + (size_t)r * stride_r + k0;
        if (vec && k0 + 4 <= kdim) {
            v = *(const f32x4*)src;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (k0 + c < kdim) v[c] = src[c];
        }
        return v;
    };
    auto fetch_rc = [&](const float* base, int r0, int r_lim, int stride_k, bool vec, int k) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 >= r_lim || k >= kdim) return v;
        const float* src = base + (size_t)k * stride_k + r0;
        if (vec && r0 + 4 <= r_lim) {
            v = *(const f32x4*)src;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (r0 + c < r_lim) v[c] = src[c];
        }
        return v;
    };
    auto fetch = [&](int sl) {
        const int pass = sl / spp, k0 = (sl - pass * spp) * KS;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int id = tid + 256 * u;
            if (A_KC) pa[u] = fetch_kc(A + pass * a_pass, i0 + id / QA, m, sa_i, a_vec, k0 + 4 * (id % QA));
            else pa[u] = fetch_rc(A + pass * a_pass, i0 + 4 * (id % QA), m, sa_k, a_vec && (i0 & 3) == 0, k0 + id / QA);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int id = tid + 256 * u;
            if (B_KC) pb[u] = fetch_kc(B + pass * b_pass, j0 + id / QB, n, sb_j, b_vec, k0 + 4 * (id % QB));
            else pb[u] = fetch_rc(B + pass * b_pass, j0 + 4 * (id % QB), n, sb_k, b_vec, k0 + id / QB);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int id = tid + 256 * u;
            *(f32x4*)(&As[(id / QA) * PA + 4 * (id % QA)]) = pa[u];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int id = tid + 256 * u;
            *(f32x4*)(&Bs[(id / QB) * PB + 4 * (id % QB)]) = pb[u];
        }
    };
    f32x16 acc = zero16();
    fetch(0);
    for (int sl = 0; sl < n_slab; ++sl) {
        stash();
        __syncthreads();
        if (sl + 1 < n_slab) fetch(sl + 1);
        if (active) {
            constexpr int QW = KSPLIT ? KS / 8 : KS / 4;      // k-quads this wave multiplies per slab
            const int q0 = kh * QW;
            const int bc = nb * 32 + l31;
#pragma unroll
            for (int qq = 0; qq < QW; ++qq) {
                const int k = 4 * (q0 + qq);
                float a0, a1, b0, b1;
                if (A_KC) {
                    const f32x4 a4 = *(const f32x4*)(&As[l31 * PA + k]);
                    a0 = hi ? a4[1] : a4[0]; a1 = hi ? a4[3] : a4[2];
                } else {
                    a0 = As[(k + hi) * PA + l31]; a1 = As[(k + 2 + hi) * PA + l31];
                }
                if (B_KC) {
                    const f32x4 b4 = *(const f32x4*)(&Bs[bc * PB + k]);
                    b0 = hi ? b4[1] : b4[0]; b1 = hi ? b4[3] : b4[2];
                } else {
                    b0 = Bs[(k + hi) * PB + bc]; b1 = Bs[(k + 2 + hi) * PB + bc];
                }
                acc = mfma32(a0, b0, acc);
                acc = mfma32(a1, b1, acc);
            }
        }
        __syncthreads();
    }
    if (KSPLIT) {                                             // the second k-half's tile through LDS (the staging rows are free now)
        float* red = lds;                                     // [2 column blocks][16][64]
        if (kh == 1 && active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(nb * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (kh == 1 || !active) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(nb * 16 + r) * 64 + lane];
    } else if (!active) {
        return;
    }
    const int j = j0 + nb * 32 + l31;
    if (j >= n) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + mfma_row(r, hi);
        if (i >= m) continue;
        float* cp = C + (size_t)i * sc_i + j;
        *cp = accumulate ? *cp + acc[r] : acc[r];
    }
}

__global__ __launch_bounds__(256, 3) void role_bgemm_kernel(const SmallGemmArgs p0, const SmallGemmArgs p1) {
    const SmallGemmArgs& p = blockIdx.y ? p1 : p0;
    __shared__ __attribute__((aligned(16))) float lds[BG_LDS_FLOATS];
    const int n_role = p.mode == SG_DENCO ? 1 : 2;
    const int n_rt = (p.max_rows + 31) / 32;
    const int n_cc = p.n_cc;
    const long item = blockIdx.x;
    if (item >= (long)p.n_vid * n_role * n_rt * n_cc) return;
    const int cc = item % n_cc;
    const int rt = (item / n_cc) % n_rt;
    const int role = (item / ((long)n_cc * n_rt)) % n_role;
    const int vid = item / ((long)n_cc * n_rt * n_role);
    const int e0 = p.e_off[vid], ne = p.e_off[vid + 1] - e0;
    if (ne <= 0) return;
    const int nq = p.nq;
    const size_t abase = (size_t)2 * nq * e0;
    // C is m x n, reduction length kdim; element strides of A (i,k), B (k,j) and C rows; pass strides of DENCO's two roles
    int m, n, kdim, sa_i, sa_k, sb_k, sb_j, sc_i, n_pass = 1;
    size_t a_pass = 0, b_pass = 0;
    const float *A, *B;
    float* C;
    bool accumulate = false;
    if (p.mode == SG_DATT) {
        m = nq; n = ne; kdim = p.d;
        A = p.dvals + ((size_t)role * p.n_vid * nq + (size_t)vid * nq) * p.lde; sa_i = p.lde; sa_k = 1;
        B = p.enco + (size_t)e0 * p.lde; sb_k = 1; sb_j = p.lde;
        C = p.out + abase + (size_t)role * nq * ne; sc_i = ne;
    } else if (p.mode == SG_DENCO) {
        m = ne; n = p.d; kdim = nq; n_pass = 2; accumulate = true;
        A = p.att + abase; sa_i = 1; sa_k = ne; a_pass = (size_t)nq * ne;
        B = p.dvals + (size_t)vid * nq * p.lde; sb_k = p.lde; sb_j = 1; b_pass = (size_t)p.n_vid * nq * p.lde;
        C = p.out + (size_t)e0 * p.ldo; sc_i = p.ldo;
    } else if (p.mode == SG_DP2A) {
        m = nq; n = p.half; kdim = ne;
        A = p.ds + abase + (size_t)role * nq * ne; sa_i = ne; sa_k = 1;
        B = p.e2a + (size_t)e0 * p.lda + role * p.half; sb_k = p.lda; sb_j = 1;
        C = p.out + (size_t)vid * nq * p.ldo + role * p.half; sc_i = p.ldo;
    } else {
        m = ne; n = p.half; kdim = nq;
        A = p.ds + abase + (size_t)role * nq * ne; sa_i = 1; sa_k = ne;
        B = p.p2a + (size_t)vid * nq * p.ldp + role * p.half; sb_k = p.ldp; sb_j = 1;
        C = p.out + (size_t)e0 * p.ldo + role * p.half; sc_i = p.ldo;
    }
    const int i0 = rt * 32, j0 = cc * 128;
    if (i0 >= m || j0 >= n) return;                          // (uniform over the workgroup)
#define BG_ARGS A, B, C, m, n, kdim, sa_i, sa_k, sb_k, sb_j, sc_i, n_pass, a_pass, b_pass, accumulate, i0, j0, lds
    if (p.mode == SG_DATT) {
        if (p.narrow) role_bgemm_body<64, 128, true, true>(BG_ARGS);
        else role_bgemm_body<128, 64, true, true>(BG_ARGS);
    } else if (p.mode == SG_DP2A) {
        role_bgemm_body<128, 64, true, false>(BG_ARGS);
    } else {
        role_bgemm_body<128, 64, false, false>(BG_ARGS);
    }
#undef BG_ARGS
}

int grid_for(long n, int cap = 4096) { long g = (n + 255) / 256; return (int)(g < cap ? (g < 1 ? 1 : g) : cap); }

}  // namespace

extern "C" int bigc_mha_bwd_drop_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                     const float* o, int ldo, const float* dO, int lddo, const float* lse, float* delta_scratch,
                                     float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                                     const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q, int max_k, long total_q_rows,
                                     int n_head, int hd, float drop_p, uint32_t drop_seed, uint32_t drop_site);
// attention_bwd_split.hip
int bigc_mha_bwd_h_prepare(hipStream_t s, int ldq, int ldk, int ldv, int lddo, int lddq, int lddk, int lddv, int** flag, int* run_id, float** amax_slot);
void bigc_mha_bwd_h_launch(hipStream_t s, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* dO, int lddo,
                           const float* lse, const float* delta, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                           const int32_t* q_off, const int32_t* k_off, int n_seq, int n_qt, int n_kt, int n_head, float scale,
                           const DropSpec& drop, int* flag, int run_id, const float* amax_slot);

extern "C" int bigc_mha_bwd_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                const float* o, int ldo, const float* dO, int lddo, const float* lse, float* delta_scratch,
                                float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                                const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q, int max_k, long total_q_rows,
                                int n_head, int hd) {
    return bigc_mha_bwd_drop_f32(stream, q, ldq, k, ldk, v, ldv, o, ldo, dO, lddo, lse, delta_scratch, dq, lddq, dk, lddk, dv, lddv, q_off, k_off,
                                 n_seq, max_q, max_k, total_q_rows, n_head, hd, 0.f, 0, 0);
}

extern "C" int bigc_mha_bwd_drop_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                     const float* o, int ldo, const float* dO, int lddo, const float* lse, float* delta_scratch,
                                     float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                                     const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q, int max_k, long total_q_rows,
                                     int n_head, int hd, float drop_p, uint32_t drop_seed, uint32_t drop_site) {
    const DropSpec drop = make_drop(drop_p, drop_seed, drop_site);
    BIGC_CHECK_ARG(q && k && v && o && dO && lse && delta_scratch && dq && dk && dv && q_off && k_off, "mha_bwd: null pointer");
    BIGC_CHECK_ARG(hd == 8 || hd == 16 || hd == 32 || hd == 64, "mha_bwd: head_dim %d", hd);
    if (n_seq <= 0 || max_q <= 0 || max_k <= 0) return BIGC_OK;
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)hd);
    // algorithmic work of the attention-core gradient: five L_q x L_k x hd products per (sequence, head) -- the scores recomputed, dP = dO V^T,
    // dV = P^T dO, dQ = dS K, dK = dS^T Q -- with sum L_q L_k >= rows^2 / n_seq (the equal-length value); q, k, v, o, dO read and dq, dk, dv written once
    ProfScope ps(stream, FAM_ATTN_BWD, 10.0 * (double)total_q_rows * ((double)total_q_rows / n_seq) * n_head * hd, 32.0 * (double)total_q_rows * n_head * hd);
    // head_dim 64 (the BIG-C encoder / decoder layers): split-operand f16 MFMA kernels first; the f32-MFMA kernels behind them run only if
    // an operand was out of the split's range
    int* run_flag = nullptr;
    int run_id = 0;
    float* amax_slot = nullptr;
    if (hd == 64 && !bigc_linear_get_exact() && bigc_mha_bwd_h_prepare(s, ldq, ldk, ldv, lddo, lddq, lddk, lddv, &run_flag, &run_id, &amax_slot) < 0)
        return bigc_fail(BIGC_ERR_HIP, "mha_bwd: no flag memory");
    hipLaunchKernelGGL(mha_delta_kernel, dim3(grid_for(total_q_rows * n_head * (hd / 4), amax_slot ? 1024 : 4096)), dim3(256), 0, s, dO, lddo, o, ldo, delta_scratch, total_q_rows, n_head, hd,
                       amax_slot);
    BIGC_LAUNCH_CHECK("mha_delta");
    const int n_qt = cdiv(max_q, 32), n_kt = cdiv(max_k, 32);
    const int gq = (int)(((long)n_seq * n_qt * n_head + 3) / 4), gk = (int)(((long)n_seq * n_kt * n_head + 3) / 4);
#define LAUNCH_BWD(HD) \
    hipLaunchKernelGGL(mha_bwd_both_kernel<HD>, dim3(gq + gk), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dO, lddo, lse, delta_scratch, dq, lddq, dk, lddk, dv, lddv, \
                       q_off, k_off, n_seq, n_qt, n_kt, gq, n_head, scale, drop, run_flag, run_id)
    if (run_flag)
        bigc_mha_bwd_h_launch(s, q, ldq, k, ldk, v, ldv, dO, lddo, lse, delta_scratch, dq, lddq, dk, lddk, dv, lddv, q_off, k_off, n_seq, n_qt, n_kt, n_head, scale,
                              drop, run_flag, run_id, amax_slot);
    switch (hd) {
        case 8: LAUNCH_BWD(8); break;
        case 16: LAUNCH_BWD(16); break;
        case 32: LAUNCH_BWD(32); break;
        default: LAUNCH_BWD(64); break;
    }
#undef LAUNCH_BWD
    BIGC_LAUNCH_CHECK("mha_bwd");
    return BIGC_OK;
}

// max_ent: an upper bound of the videos' entity counts (the caller's bigc_batch.max_trk), INT_MAX when unknown
int bigc_role_attention_bwd_launch(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* datt,
                                   float* ds, const int32_t* e_off, int n_vid, int nq, int half, float scale, int max_ent);

extern "C" int bigc_role_attention_bwd_f32(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* datt,
                                           float* ds, const int32_t* e_off, int n_vid, int nq, int half, float scale) {
    return bigc_role_attention_bwd_launch(stream, p2a, ldp, e2a, lde, datt, ds, e_off, n_vid, nq, half, scale, 0x7fffffff);
}

int bigc_role_attention_bwd_launch(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* datt,
                                   float* ds, const int32_t* e_off, int n_vid, int nq, int half, float scale, int max_ent) {
    BIGC_CHECK_ARG(p2a && e2a && datt && ds && e_off, "role_attention_bwd: null pointer");
    if (n_vid <= 0) return BIGC_OK;
    const int n_qt = cdiv(nq, 32);
    ProfScope ps(stream, FAM_ATTN_BWD, 0.0, 0.0);
    // videos of at most 64 entities: the four-wave tile kernel; larger ones: the wave-per-item kernel.  Both are launched (each leaves the
    // other's videos alone: the entity counts live on the device), the second only when the caller's bound says it is needed
    const bool tiles = half % 32 == 0 && (ldp % 4) == 0 && (lde % 4) == 0;
    if (tiles)
        hipLaunchKernelGGL(role_attention_bwd_tiles_kernel, dim3(n_vid * n_qt), dim3(256), 0, (hipStream_t)stream, p2a, ldp, e2a, lde, datt, ds,
                           e_off, n_vid, nq, n_qt, half, scale);
    if (!tiles || max_ent > 64) {
        const int wpb = n_vid * n_qt <= 512 ? 1 : 4;             // small batches: one wave per workgroup (more CUs)
        hipLaunchKernelGGL(role_attention_bwd_kernel, dim3(cdiv(n_vid * n_qt, wpb)), dim3(64 * wpb), 0, (hipStream_t)stream, p2a, ldp, e2a, lde, datt, ds,
                           e_off, n_vid, nq, n_qt, half, scale, tiles ? 1 : 0);
    }
    BIGC_LAUNCH_CHECK("role_attention_bwd");
    return BIGC_OK;
}

namespace {
long small_gemm_args(SmallGemmArgs& p, int mode, const float* att_or_ds, const float* dvals_or_p2a, int ld1, const float* enco_or_e2a, int ld2,
                     float* out, int ldo, const int32_t* e_off, int n_vid, int nq, int d, int half, int max_trk) {
    p.mode = mode; p.n_vid = n_vid; p.nq = nq; p.d = d; p.half = half;
    p.att = att_or_ds; p.ds = att_or_ds; p.dvals = dvals_or_p2a; p.p2a = dvals_or_p2a; p.ldp = ld1;
    p.enco = enco_or_e2a; p.e2a = enco_or_e2a; p.lde = (mode <= 1) ? ld2 : 0; p.lda = ld2;
    p.out = out; p.ldo = ldo; p.e_off = e_off;
    int rows, cols, n_role;
    if (mode == SG_DATT) { rows = nq; cols = max_trk; n_role = 2; }
    else if (mode == SG_DENCO) { rows = max_trk; cols = d; n_role = 1; }
    else if (mode == SG_DP2A) { rows = nq; cols = half; n_role = 2; }
    else { rows = max_trk; cols = half; n_role = 2; }
    p.max_rows = rows;
    p.n_cc = cdiv(cols, 128);
    p.narrow = cols <= 64;
    return (long)n_vid * n_role * cdiv(rows, 32) * p.n_cc;
}
}  // namespace

namespace {
void role_bgemm_launch(hipStream_t s, const SmallGemmArgs& p0, const SmallGemmArgs& p1, long items, int ny) {
    static const bool lds_off = getenv("BIGC_SG_LDS") && atoi(getenv("BIGC_SG_LDS")) == 0;
    if (lds_off) hipLaunchKernelGGL(small_gemm_kernel, dim3((int)items, ny), dim3(256), 0, s, p0, p1);
    else hipLaunchKernelGGL(role_bgemm_kernel, dim3((int)items, ny), dim3(256), 0, s, p0, p1);
}
}  // namespace

// mode 0: datt = dvals @ enco^T; 1: denco += att^T @ dvals; 2: dp2a = ds @ e2a; 3: de2a = ds^T @ p2a
extern "C" int bigc_role_small_gemm_f32(void* stream, int mode, const float* att_or_ds, const float* dvals_or_p2a, int ld1,
                                        const float* enco_or_e2a, int ld2, float* out, int ldo, const int32_t* e_off, int n_vid, int nq,
                                        int d, int half, int max_trk) {
    BIGC_CHECK_ARG(mode >= 0 && mode <= 3 && out && e_off, "role_small_gemm: bad argument");
    if (n_vid <= 0) return BIGC_OK;
    SmallGemmArgs p;
    const long items = small_gemm_args(p, mode, att_or_ds, dvals_or_p2a, ld1, enco_or_e2a, ld2, out, ldo, e_off, n_vid, nq, d, half, max_trk);
    ProfScope ps(stream, FAM_ATTN_BWD, 0.0, 0.0);
    role_bgemm_launch((hipStream_t)stream, p, p, items, 1);
    BIGC_LAUNCH_CHECK("role_small_gemm");
    return BIGC_OK;
}

// Two of them in one launch: pair 0 = (datt = dvals @ enco^T -> out_a (ld unused), denco += att^T @ dvals -> out_b (ldo_b));
// pair 1 = (dp2a = ds @ e2a -> out_a, de2a = ds^T @ p2a -> out_b), both with leading dimension ldo_b.  x = att (pair 0) / ds (pair 1),
// y = dvals (ld_y) / p2a (ld_y), z = enco (ld_z) / e2a (ld_z).
extern "C" int bigc_role_small_gemm_pair_f32(void* stream, int pair, const float* x, const float* y, int ld_y, const float* z, int ld_z,
                                             float* out_a, float* out_b, int ldo_b, const int32_t* e_off, int n_vid, int nq, int d, int half,
                                             int max_trk) {
    BIGC_CHECK_ARG((pair == 0 || pair == 1) && out_a && out_b && e_off, "role_small_gemm_pair: bad argument");
    if (n_vid <= 0) return BIGC_OK;
    SmallGemmArgs p0, p1;
    long i0, i1;
    if (pair == 0) {
        i0 = small_gemm_args(p0, SG_DATT, nullptr, y, ld_y, z, ld_z, out_a, 0, e_off, n_vid, nq, d, half, max_trk);
        i1 = small_gemm_args(p1, SG_DENCO, x, y, ld_y, nullptr, ld_z, out_b, ldo_b, e_off, n_vid, nq, d, half, max_trk);
    } else {
        i0 = small_gemm_args(p0, SG_DP2A, x, nullptr, 0, z, ld_z, out_a, ldo_b, e_off, n_vid, nq, d, half, max_trk);
        i1 = small_gemm_args(p1, SG_DE2A, x, y, ld_y, nullptr, 0, out_b, ldo_b, e_off, n_vid, nq, d, half, max_trk);
    }
    ProfScope ps(stream, FAM_ATTN_BWD, 0.0, 0.0);
    role_bgemm_launch((hipStream_t)stream, p0, p1, i0 > i1 ? i0 : i1, 2);
    BIGC_LAUNCH_CHECK("role_small_gemm_pair");
    return BIGC_OK;
}
