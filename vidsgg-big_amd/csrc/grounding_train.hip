// Training step of the temporal grounding scorer on the HIP path (reference: ``DEBUG._forward_train``
// models/grd_model_v5.py:375-389 with ``map2bins`` :391-442 and ``loss`` :444-527; network :331-373).
//   grd_train_forward   forward_propagation for one video with every activation kept in the workspace
//   grd_loss_grad       the five loss terms for the (query, bin, GT instance) lists the host derived from the GT
//                       (get_gt_labels :224-250 evaluated on the fly) and d regr / d conf / d cls
//   grd_train_backward  the whole backward pass, gradients accumulated into the registered destinations
// Dropout is not applied (parity is defined with the reference modules in eval() mode).
#include "grounding_internal.h"
#include "prof.h"

int bigc_transpose_pad(hipStream_t s, const float* in, float* out, int rows, int cols, int ld_out);
int bigc_scratch_floats(hipStream_t s, size_t need_floats, int slot, float** out);                                         // backward.hip
int bigc_partial_rows_finish(hipStream_t s, const float* part, float* out, float* out2, int blocks, int w, int split);     // backward.hip
int bigc_ordered_rows_add(hipStream_t s, const float* src, int ld_src, const int32_t* key, int key_stride, int rows, int width, float* dst, int ld_dst);   // backward.hip
int grd_query_input(hipStream_t hs, const float* enti, const float* pred, const int32_t* quint, const int32_t* dura, float video_len,
                    float* out, int n, int de, int kq);
extern "C" int bigc_linear_dw_f32(void* stream, const float* dY, int ldy, const float* X, int ldx, const int32_t* x_idx, int nseg,
                                  int seg_len, float* dW, int ldw, float* db, int M, int N, int K);
extern "C" int bigc_relu_bwd_f32(void* stream, float* dY, const float* Y, long n);
extern "C" int bigc_axpy_f32(void* stream, float* y, const float* x, long n);
extern "C" int bigc_layernorm_bwd_f32(void* stream, const float* x, const float* gamma, const float* dy, float* dx, float* dgamma,
                                      float* dbeta, int rows, int d, float eps);
extern "C" int bigc_mha_bwd_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                const float* o, int ldo, const float* dO, int lddo, const float* lse, float* delta_scratch,
                                float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                                const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q, int max_k, long total_q_rows,
                                int n_head, int hd);

extern "C" int bigc_mha_core_drop_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                      float* o, int ldo, const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q,
                                      int n_head, int hd, float* lse, float drop_p, uint32_t drop_seed, uint32_t drop_site);
extern "C" int bigc_mha_bwd_drop_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                     const float* o, int ldo, const float* dO, int lddo, const float* lse, float* delta_scratch,
                                     float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                                     const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q, int max_k, long total_q_rows,
                                     int n_head, int hd, float drop_p, uint32_t drop_seed, uint32_t drop_site);
extern "C" int bigc_dropout_f32(void* stream, float* x, long rows, int n, int ld, float p, uint32_t seed, uint32_t site);

namespace {

// dropout sites of a QANet layer (models/grd_model_v5.py:123-136): base + {0,1: after conv blocks 1 and 3 (p = 0.05, 0.1),
// 2: attention probabilities, 3: after the attention residual, 4: layer output}; bases 0x100 video, 0x200 query, 0x300 combined
struct QDrop { float p; uint32_t seed; uint32_t base; };

int grid_for(long n, int cap = 4096) { long g = (n + 255) / 256; return (int)(g < cap ? (g < 1 ? 1 : g) : cap); }
int round4(int x) { return (x + 3) & ~3; }

// ---- depth-wise conv backward -------------------------------------------------------------------------------
// dX[s,t,c] = sum_j w[c,j] dY[s, t - j + ks/2, c]
__global__ __launch_bounds__(256) void dwconv_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                            long total4, int T, int d4, int ks) {
    const int half = ks / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long row = i / d4;
        const int c4 = (int)(i - row * d4);
        const int t = (int)(row % T);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* wp = w + (size_t)c4 * 4 * ks;
        for (int j = 0; j < ks; ++j) {
            const int tt = t - j + half;
            if (tt < 0 || tt >= T) continue;
            const f32x4 g = ((const f32x4*)dy)[(row + (half - j)) * d4 + c4];
            acc[0] += wp[j] * g[0]; acc[1] += wp[ks + j] * g[1]; acc[2] += wp[2 * ks + j] * g[2]; acc[3] += wp[3 * ks + j] * g[3];
        }
        ((f32x4*)dx)[i] = acc;
    }
}

// dW[c,j] += sum_{s,t} dY[s,t,c] X[s,t+j-ks/2,c] ; db[c] += sum dY[s,t,c]; one thread per channel and row chunk.  Every row chunk
// (blockIdx.y) stores its sums as one partial row [d * ks weight terms | d bias terms] of `part`; bigc_partial_rows_finish adds the chunks
// in order (no float atomics: bit-reproducible parameter gradients)
__global__ __launch_bounds__(256) void dwconv_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                           long rows, int T, int d, int ks, int rows_per_block) {
    const int half = ks / 2;
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int c = blockIdx.x * 256 + threadIdx.x; c < d; c += gridDim.x * 256) {
        float acc[7] = {0, 0, 0, 0, 0, 0, 0};
        float sb = 0.f;
        for (long r = r0; r < r1; ++r) {
            const int t = (int)(r % T);
            const float g = dy[r * d + c];
            sb += g;
            for (int j = 0; j < ks; ++j) {
                const int tt = t + j - half;
                if (tt >= 0 && tt < T) acc[j] += g * x[(r + (j - half)) * d + c];
            }
        }
        float* out = part + (size_t)blockIdx.y * ((size_t)d * ks + d);
        for (int j = 0; j < ks; ++j) out[(size_t)c * ks + j] = acc[j];
        out[(size_t)d * ks + c] = sb;
    }
}

// ---- co-attention backward (forward: coattention_kernel in grounding.hip, models/grd_model_v5.py:346-365) ------------
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
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

__global__ __launch_bounds__(256) void coattention_bwd_kernel(const float* __restrict__ video, const float* __restrict__ proj,
                                                              const float* __restrict__ query, const float* __restrict__ dout,
                                                              float* __restrict__ part,
                                                              float* __restrict__ g_query, int T, int d) {
    // part: (n, 2 T d) -- this query's contribution to [d video | d proj]; bigc_partial_rows_finish adds the queries in order (the two
    // gradients are sums over the queries of the video: float atomics here made the step's bits depend on the order the blocks ran in)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* q = lds;                  // 3d
    float* s2 = q + 3 * d;           // 3d
    float* ds2 = s2 + 3 * d;         // 3d
    float* dq = ds2 + 3 * d;         // 3d
    float* R = dq + 3 * d;           // 3T  softmax over the 3 words
    float* Cc = R + 3 * T;           // 3T  softmax over time
    float* dR = Cc + 3 * T;          // 3T  (then d sim)
    float* dC = dR + 3 * T;          // 3T
    float* red = dC + 3 * T;         // 4
    const int i = blockIdx.x, tid = threadIdx.x;
    const float* gout = dout + (size_t)i * T * 4 * d;
    for (int c = tid; c < 3 * d; c += 256) { q[c] = query[(size_t)i * 3 * d + c]; dq[c] = 0.f; }
    __syncthreads();
    // recompute the forward quantities
    for (int t = tid; t < T; t += 256) {
        const float* pr = proj + (size_t)t * d;
        float s[3] = {0.f, 0.f, 0.f};
        for (int c = 0; c < d; ++c) { const float p = pr[c]; s[0] += p * q[c]; s[1] += p * q[d + c]; s[2] += p * q[2 * d + c]; }
        const float m = fmaxf(s[0], fmaxf(s[1], s[2]));
        const float e0 = expf(s[0] - m), e1 = expf(s[1] - m), e2 = expf(s[2] - m), sum = e0 + e1 + e2;
        R[t] = e0 / sum; R[T + t] = e1 / sum; R[2 * T + t] = e2 / sum;
        Cc[t] = s[0]; Cc[T + t] = s[1]; Cc[2 * T + t] = s[2];
    }
    __syncthreads();
    for (int l = 0; l < 3; ++l) {
        float m = -INFINITY;
        for (int t = tid; t < T; t += 256) m = fmaxf(m, Cc[l * T + t]);
        m = block_reduce(m, true, red);
        float sum = 0.f;
        for (int t = tid; t < T; t += 256) { const float e = expf(Cc[l * T + t] - m); Cc[l * T + t] = e; sum += e; }
        sum = block_reduce(sum, false, red);
        for (int t = tid; t < T; t += 256) Cc[l * T + t] /= sum;
    }
    __syncthreads();
    for (int e = tid; e < 3 * d; e += 256) {
        const int l = e / d, c = e - l * d;
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += Cc[l * T + t] * video[(size_t)t * d + c];
        s2[e] = acc;
    }
    __syncthreads();
    // phase 1 (thread per t): dR[t,l] = sum_c dA q[l,c] + dB s2[l,c]
    for (int t = tid; t < T; t += 256) {
        const float* g = gout + (size_t)t * 4 * d;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int c = 0; c < d; ++c) {
            const float v = video[(size_t)t * d + c];
            const float dA = g[d + c] + g[2 * d + c] * v, dB = g[3 * d + c] * v;
            a0 += dA * q[c] + dB * s2[c];
            a1 += dA * q[d + c] + dB * s2[d + c];
            a2 += dA * q[2 * d + c] + dB * s2[2 * d + c];
        }
        dR[t] = a0; dR[T + t] = a1; dR[2 * T + t] = a2;
    }
    // phase 2 (thread per (l,c)): dq[l,c] = sum_t R[t,l] dA[t,c]; ds2[l,c] = sum_t R[t,l] dB[t,c]
    for (int e = tid; e < 3 * d; e += 256) {
        const int l = e / d, c = e - l * d;
        float aq = 0.f, as = 0.f;
        for (int t = 0; t < T; ++t) {
            const float* g = gout + (size_t)t * 4 * d;
            const float v = video[(size_t)t * d + c], r = R[l * T + t];
            aq += r * (g[d + c] + g[2 * d + c] * v);
            as += r * (g[3 * d + c] * v);
        }
        dq[e] = aq;
        ds2[e] = as;
    }
    __syncthreads();
    // dCc[t,l] = sum_c video[t,c] ds2[l,c]
    for (int t = tid; t < T; t += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int c = 0; c < d; ++c) {
            const float v = video[(size_t)t * d + c];
            a0 += v * ds2[c]; a1 += v * ds2[d + c]; a2 += v * ds2[2 * d + c];
        }
        dC[t] = a0; dC[T + t] = a1; dC[2 * T + t] = a2;
    }
    __syncthreads();
    // d sim = softmax-over-words backward + softmax-over-time backward (stored in dR)
    for (int l = 0; l < 3; ++l) {
        float part = 0.f;
        for (int t = tid; t < T; t += 256) part += dC[l * T + t] * Cc[l * T + t];
        const float col = block_reduce(part, false, red);
        for (int t = tid; t < T; t += 256) dC[l * T + t] = Cc[l * T + t] * (dC[l * T + t] - col);
    }
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        const float dot = dR[t] * R[t] + dR[T + t] * R[T + t] + dR[2 * T + t] * R[2 * T + t];
        for (int l = 0; l < 3; ++l) dR[l * T + t] = R[l * T + t] * (dR[l * T + t] - dot) + dC[l * T + t];
    }
    __syncthreads();
    // dq[l,c] += sum_t dsim[t,l] proj[t,c]
    for (int e = tid; e < 3 * d; e += 256) {
        const int l = e / d, c = e - l * d;
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += dR[l * T + t] * proj[(size_t)t * d + c];
        g_query[(size_t)i * 3 * d + e] = dq[e] + acc;
    }
    // d proj[t,c] += sum_l dsim[t,l] q[l,c];  d video[t,c] += g0 + g2 A + g3 B + sum_l Cc[t,l] ds2[l,c]
    for (int e = tid; e < T * d; e += 256) {
        const int t = e / d, c = e - t * d;
        const float* g = gout + (size_t)t * 4 * d;
        const float r0 = R[t], r1 = R[T + t], r2 = R[2 * T + t];
        const float A = r0 * q[c] + r1 * q[d + c] + r2 * q[2 * d + c];
        const float Bv = r0 * s2[c] + r1 * s2[d + c] + r2 * s2[2 * d + c];
        const float dv = g[c] + g[2 * d + c] * A + g[3 * d + c] * Bv + Cc[t] * ds2[c] + Cc[T + t] * ds2[d + c] + Cc[2 * T + t] * ds2[2 * d + c];
        part[(size_t)i * 2 * T * d + e] = dv;
        part[(size_t)i * 2 * T * d + (size_t)T * d + e] = dR[t] * q[c] + dR[T + t] * q[d + c] + dR[2 * T + t] * q[2 * d + c];
    }
}

// ---- loss (models/grd_model_v5.py:444-527 with the labels of :224-250) ------------------------------------------
__device__ __forceinline__ float bce_logits(float x, float z) { return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

struct GrdLossArgs {
    const float *regrs, *conf, *cls, *clip, *target;     // regrs already through the sigmoid
    const int32_t *pos_i, *pos_b, *pos_j, *neg_i, *neg_b;
    int n_pos, n_neg, T, k, ldc, ldr;                    // ldc / ldr: leading dimensions of the gradient arrays (multiples of 4)
    float inv_pos, inv_neg, inv_mask;                    // 1 / element counts of the means over the whole batch
    float f_cls, f_ct, f_regr;
    float gs[5];                                         // upstream gradients of [pos_cls, neg_cls, pos_ct, neg_ct, regr]
    float *loss, *dregr, *dconf, *dcls;                  // loss[5] accumulated; gradients accumulated (pre-sigmoid for regr)
    float* part;                                         // (gridDim.x, 5) loss terms per workgroup, added in block order by the finish launch
};

// Positive entries are (query i, bin b, GT j); two GT members of a query may fall into the same bin, i.e. write the same gradient
// elements.  No float atomics: the FIRST entry of a (query, bin) pair owns its elements and adds the later entries of the same pair in
// entry order (entries of a query are few and adjacent); negative entries are unique (query, bin) pairs that no positive entry uses.
__global__ __launch_bounds__(256) void grd_loss_grad_kernel(const GrdLossArgs p) {
    const int T = p.T, k = p.k;
    float part[5] = {0, 0, 0, 0, 0};
    const long total = (long)(p.n_pos + p.n_neg) * T;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ent = (int)(e / T), t = (int)(e - (long)ent * T);
        if (ent < p.n_pos) {
            const int i = p.pos_i[ent], b = p.pos_b[ent];
            bool first = true;
            for (int o = 0; o < ent; ++o) first = first && !(p.pos_i[o] == i && p.pos_b[o] == b);
            if (!first) continue;
            const size_t ck = ((size_t)i * T + t) * k + b, rk = ((size_t)i * T + t) * 2 * k + b;
            const size_t gck = ((size_t)i * T + t) * p.ldc + b, grk = ((size_t)i * T + t) * p.ldr + b;
            const float y = p.cls[ck], x = p.conf[ck];
            const float rl = p.regrs[rk], rr = p.regrs[rk + k];
            float g_cls = 0.f, g_conf = 0.f, g_l = 0.f, g_r = 0.f;
            for (int o = ent; o < p.n_pos; ++o) {
                if (p.pos_i[o] != i || p.pos_b[o] != b) continue;
                const int j = p.pos_j[o];
                const float left = p.clip[t] - p.target[2 * j], right = p.target[2 * j + 1] - p.clip[t];
                const bool outside = left <= 0.f || right <= 0.f;
                const float score = outside ? 0.f : 1.f;
                part[0] += bce_logits(y, score) * p.inv_pos * p.f_cls;
                g_cls += (sigm(y) - score) * p.inv_pos * p.f_cls * p.gs[0];
                const float ct = outside ? 0.f : sqrtf(fminf(left, right) / fmaxf(left, right));
                if (ct > 0.f) {
                    part[2] += bce_logits(x, ct) * p.inv_mask * p.f_ct;
                    g_conf += (sigm(x) - ct) * p.inv_mask * p.f_ct * p.gs[2];
                    const float num = fminf(rr, right) + fminf(rl, left), den = fmaxf(rr, right) + fmaxf(rl, left);
                    const float tiou = num / den;
                    part[4] += -logf(tiou + 1e-6f) * p.inv_mask * p.f_regr;
                    const float gt_ = -1.f / (tiou + 1e-6f) * p.inv_mask * p.f_regr * p.gs[4];
                    const float dl = gt_ * ((rl < left ? 1.f : 0.f) / den - (rl > left ? num / (den * den) : 0.f));
                    const float dr = gt_ * ((rr < right ? 1.f : 0.f) / den - (rr > right ? num / (den * den) : 0.f));
                    g_l += dl * rl * (1.f - rl);
                    g_r += dr * rr * (1.f - rr);
                }
            }
            p.dcls[gck] += g_cls;
            p.dconf[gck] += g_conf;
            p.dregr[grk] += g_l;
            p.dregr[grk + k] += g_r;
        } else {
            const int m = ent - p.n_pos;
            const int i = p.neg_i[m], b = p.neg_b[m];
            const size_t ck = ((size_t)i * T + t) * k + b, gck = ((size_t)i * T + t) * p.ldc + b;
            const float x = p.conf[ck], y = p.cls[ck];
            part[1] += bce_logits(y, 0.f) * p.inv_neg * p.f_cls;
            part[3] += bce_logits(x, 0.f) * p.inv_neg * p.f_ct;
            p.dcls[gck] += sigm(y) * p.inv_neg * p.f_cls * p.gs[1];
            p.dconf[gck] += sigm(x) * p.inv_neg * p.f_ct * p.gs[3];
        }
    }
    __shared__ float red[4][5];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        float v = part[a];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][a] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) p.part[(size_t)blockIdx.x * 5 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- workspace ------------------------------------------------------------------------------------------------
struct EncSave { float *a, *ln[5], *dw[4], *r[4], *c[4], *qkv, *lse, *att, *o, *l5, *r5; int n_seq, T; };
struct HeadSave { float *dw[5], *x[5]; };    // x[0] unused (= encoder output)

struct GTW {
    float *qin, *qemb, *vemb, *venc, *qenc, *proj, *comb, *cin, *cenc;
    EncSave ev, eq, ec;
    HeadSave hr, hc, hl;
    int32_t* off;
    // backward scratch
    float *g0, *g1, *g2, *g3, *gqkv, *delta, *gcomb, *gvenc, *gproj, *gqenc, *gqin, *wT, *dump;
};

void enc_layout(Arena& a, EncSave& e, size_t rows, size_t d, int n_seq, int T) {
    e.n_seq = n_seq; e.T = T;
    e.a = a.take<float>(rows * d);
    for (int i = 0; i < 5; ++i) e.ln[i] = a.take<float>(rows * d);
    for (int i = 0; i < 4; ++i) { e.dw[i] = a.take<float>(rows * d); e.r[i] = a.take<float>(rows * d); e.c[i] = a.take<float>(rows * d); }
    e.qkv = a.take<float>(rows * 3 * d); e.lse = a.take<float>(rows * 8); e.att = a.take<float>(rows * d);
    e.o = a.take<float>(rows * d); e.l5 = a.take<float>(rows * d); e.r5 = a.take<float>(rows * d);
}

void head_layout(Arena& a, HeadSave& h, size_t rows, size_t d) {
    for (int i = 0; i < 5; ++i) { h.dw[i] = a.take<float>(rows * d); h.x[i] = (i == 0) ? nullptr : a.take<float>(rows * d); }
}

bool gt_layout(const grd_model* m, int T, int n, Arena& a, GTW& w) {
    const size_t d = m->cfg.dim_hidden, k = m->cfg.num_bins;
    const size_t rc = (size_t)n * T, rq = (size_t)3 * n, rv = T;
    size_t rows = rc > rq ? rc : rq;
    if (rows < rv) rows = rv;
    w.qin = a.take<float>(rq * m->kq); w.qemb = a.take<float>(rq * d); w.vemb = a.take<float>(rv * d);
    w.venc = a.take<float>(rv * d); w.qenc = a.take<float>(rq * d); w.proj = a.take<float>(rv * d);
    w.comb = a.take<float>(rc * 4 * d); w.cin = a.take<float>(rc * d); w.cenc = a.take<float>(rc * d);
    enc_layout(a, w.ev, rv, d, 1, T);
    enc_layout(a, w.eq, rq, d, n, 3);
    enc_layout(a, w.ec, rc, d, n, T);
    head_layout(a, w.hr, rc, d); head_layout(a, w.hc, rc, d); head_layout(a, w.hl, rc, d);
    w.off = a.take<int32_t>(rows + 2);
    w.g0 = a.take<float>(rows * d); w.g1 = a.take<float>(rows * d); w.g2 = a.take<float>(rows * d); w.g3 = a.take<float>(rows * d);
    w.gqkv = a.take<float>(rows * 3 * d); w.delta = a.take<float>(rows * 8);
    w.gcomb = a.take<float>(rc * 4 * d);
    w.gvenc = a.take<float>(rv * d); w.gproj = a.take<float>(rv * d); w.gqenc = a.take<float>(rq * d);
    w.gqin = a.take<float>(rq * m->kq);
    size_t wmax = (size_t)4 * d * d;
    if ((size_t)m->kq * d > wmax) wmax = (size_t)m->kq * d;
    if ((size_t)round4(2 * (int)k) * d > wmax) wmax = (size_t)round4(2 * (int)k) * d;
    if ((size_t)m->cfg.dim_feat * d > wmax) wmax = (size_t)m->cfg.dim_feat * d;
    w.wT = a.take<float>(wmax);
    w.dump = a.take<float>(2 * d);
    return !a.overflow;
}

int open_ws(const grd_model* m, int T, int n, void* workspace, size_t bytes, GTW& w) {
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (!workspace || bytes < (size_t)(base - (char*)workspace)) return bigc_fail(BIGC_ERR_WORKSPACE, "grd_train: no workspace");
    Arena a{base, bytes - (size_t)(base - (char*)workspace)};
    if (!gt_layout(m, T, n, a, w)) return bigc_fail(BIGC_ERR_WORKSPACE, "grd_train: workspace of %zu bytes too small", bytes);
    return BIGC_OK;
}

float* gof(grd_model* m, const std::string& key) {
    auto it = m->grads.find(key);
    return it == m->grads.end() ? nullptr : it->second;
}
float* gof_or(grd_model* m, const std::string& key, float* fallback) {
    float* p = gof(m, key);
    return p ? p : fallback;
}

// QANetEncoderLayer forward keeping what the backward needs (models/grd_model_v5.py:110-137)
int qanet_fwd(void* s, const Enc& E, const float* x, const float* pos, int d, EncSave& S, int32_t* off, float* out, const QDrop& dr) {
    const int rows = S.n_seq * S.T;
    TRY(bigc_add_rows_f32(s, x, pos, S.a, rows, d, S.T));
    TRY(bigc_layernorm_f32(s, S.a, E.normb.w, E.normb.b, S.ln[0], rows, d, 1e-5f));
    const float* res = S.a;
    for (int i = 0; i < 4; ++i) {
        TRY(bigc_dwconv_f32(s, S.ln[i], E.conv[i].dw_w, E.conv[i].dw_b, S.dw[i], S.n_seq, S.T, d, E.ks));
        TRY(linear(s, S.dw[i], d, E.conv[i].pw, S.r[i], d, rows, d, d, BIGC_ACT_RELU));
        hipError_t e = hipMemcpyAsync(S.c[i], S.r[i], sizeof(float) * (size_t)rows * d, hipMemcpyDeviceToDevice, (hipStream_t)s);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "qanet_fwd: memcpy");
        TRY(bigc_axpy_f32(s, S.c[i], res, (long)rows * d));
        if (dr.p > 0.f && (i + 1) % 2 == 0) TRY(bigc_dropout_f32(s, S.c[i], rows, d, d, dr.p * (i + 1) / 4, dr.seed, dr.base + i / 2));
        res = S.c[i];
        TRY(bigc_layernorm_f32(s, S.c[i], E.norm_seq[i].w, E.norm_seq[i].b, S.ln[i + 1], rows, d, 1e-5f));
    }
    TRY(linear(s, S.ln[4], d, E.in_proj, S.qkv, 3 * d, rows, 3 * d, d, BIGC_ACT_NONE));
    TRY(bigc_fill_offsets((hipStream_t)s, off, S.n_seq + 1, S.T));
    TRY(bigc_mha_core_drop_f32(s, S.qkv, 3 * d, S.qkv + d, 3 * d, S.qkv + 2 * d, 3 * d, S.att, d, off, off, S.n_seq, S.T, 8, d / 8, S.lse, dr.p,
                               dr.seed, dr.base + 2));
    TRY(linear(s, S.att, d, E.out_proj, S.o, d, rows, d, d, BIGC_ACT_NONE, res, d));
    if (dr.p > 0.f) TRY(bigc_dropout_f32(s, S.o, rows, d, d, dr.p, dr.seed, dr.base + 3));
    TRY(bigc_layernorm_f32(s, S.o, E.norme.w, E.norme.b, S.l5, rows, d, 1e-5f));
    TRY(linear(s, S.l5, d, E.fc, S.r5, d, rows, d, d, BIGC_ACT_RELU));
    hipError_t e = hipMemcpyAsync(out, S.r5, sizeof(float) * (size_t)rows * d, hipMemcpyDeviceToDevice, (hipStream_t)s);
    if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "qanet_fwd: memcpy");
    TRY(bigc_axpy_f32(s, out, S.o, (long)rows * d));
    if (dr.p > 0.f) TRY(bigc_dropout_f32(s, out, rows, d, d, dr.p, dr.seed, dr.base + 4));
    return BIGC_OK;
}

int head_fwd(void* s, const Head& H, const float* x, int n_seq, int T, int d, HeadSave& S, float* out, int act) {
    const int rows = n_seq * T;
    const float* cur = x;
    for (int i = 0; i < 4; ++i) {
        TRY(bigc_dwconv_f32(s, cur, H.layer[i].dw_w, H.layer[i].dw_b, S.dw[i], n_seq, T, d, 3));
        TRY(linear(s, S.dw[i], d, H.layer[i].pw, S.x[i + 1], d, rows, d, d, BIGC_ACT_RELU));
        cur = S.x[i + 1];
    }
    TRY(bigc_dwconv_f32(s, cur, H.layer[4].dw_w, H.layer[4].dw_b, S.dw[4], n_seq, T, d, 3));
    TRY(linear(s, S.dw[4], d, H.layer[4].pw, out, H.c_out, rows, H.c_out, d, act));
    return BIGC_OK;
}

struct GBwd {
    grd_model* m; void* s; GTW* w;
    // dX[M,K] (=|+=) dY[M,N] W[N,K]
    int dx(const float* W, const float* dY, int ldy, int M, int N, int K, float* dX, int lddx, bool acc) {
        const int n4 = round4(N);
        BIGC_CHECK_ARG(ldy % 4 == 0 && ldy >= n4, "grd_train: dY leading dimension %d must be a multiple of 4 and >= %d", ldy, n4);
        const float* wT = nullptr;             // cached per weights epoch, refreshed by the step's batched preparation
        TRY(bigc_weight_transposed((hipStream_t)s, W, N, K, K, &wT));
        return bigc_linear_f32(s, dY, ldy, nullptr, 1, n4, wT, n4, nullptr, acc ? dX : nullptr, lddx, dX, lddx, M, K, n4, BIGC_ACT_NONE);
    }
    int lin(const std::string& key, const Lin& L, const float* dY, int ldy, const float* X, int ldx, int M, int N, int K, float* dX, int lddx,
            bool acc, bool bias = true) {
        if (float* gw = gof(m, key + ".weight"))
            TRY(bigc_linear_dw_f32(s, dY, ldy, X, ldx, nullptr, 1, K, gw, K, bias ? gof(m, key + ".bias") : nullptr, M, N, K));
        if (dX) TRY(dx(L.w, dY, ldy, M, N, K, dX, lddx, acc));
        return BIGC_OK;
    }
    // depth-wise separable conv: y = pw(dwconv(x)); given dY -> parameter grads and dX (written to dx_out)
    int dws(const std::string& key, const DW& P, const float* dY, int ldy, int c_out, const float* x, const float* dwsave, int n_seq, int T, int d,
            int ks, float* scratch, float* dx_out) {
        const int rows = n_seq * T;
        TRY(lin(key + ".point_wise", P.pw, dY, ldy, dwsave, d, rows, c_out, d, scratch, d, false));
        float* gw = gof(m, key + ".depth_wise.weight");
        if (gw) {
            const int rpb = rows > 4096 ? (rows + 255) / 256 : 16;
            const int chunks = (rows + rpb - 1) / rpb;
            float* part = nullptr;
            TRY(bigc_scratch_floats((hipStream_t)s, (size_t)chunks * ((size_t)d * ks + d), 3, &part));
            hipLaunchKernelGGL(dwconv_bwd_w_kernel, dim3((d + 255) / 256, chunks), dim3(256), 0, (hipStream_t)s, scratch, x, part, (long)rows, T, d, ks, rpb);
            BIGC_LAUNCH_CHECK("dwconv_bwd_w");
            TRY(bigc_partial_rows_finish((hipStream_t)s, part, gw, gof(m, key + ".depth_wise.bias"), chunks, d * ks + d, d * ks));
        }
        if (dx_out) {
            const long total4 = (long)rows * (d / 4);
            hipLaunchKernelGGL(dwconv_bwd_dx_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, scratch, P.dw_w, dx_out, total4, T, d / 4, ks);
            BIGC_LAUNCH_CHECK("dwconv_bwd_dx");
        }
        return BIGC_OK;
    }
};

// backward of qanet_fwd: g_out (rows,d) is consumed (overwritten); result d x is left in w.g0
int qanet_bwd(GBwd& B, const Enc& E, int d, EncSave& S, float* g_out, const QDrop& dr) {
    void* s = B.s; GTW& w = *B.w; grd_model* m = B.m;
    const int rows = S.n_seq * S.T;
    const std::string& p = E.key;
    // out = dropout(r5 + o)
    if (dr.p > 0.f) TRY(bigc_dropout_f32(s, g_out, rows, d, d, dr.p, dr.seed, dr.base + 4));
    hipError_t e = hipMemcpyAsync(w.g1, g_out, sizeof(float) * (size_t)rows * d, hipMemcpyDeviceToDevice, (hipStream_t)s);   // g1 = d r5
    if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "qanet_bwd: memcpy");
    TRY(bigc_relu_bwd_f32(s, w.g1, S.r5, (long)rows * d));
    TRY(B.lin(p + ".fc", E.fc, w.g1, d, S.l5, d, rows, d, d, w.g2, d, false));                      // g2 = d l5
    TRY(bigc_layernorm_bwd_f32(s, S.o, E.norme.w, w.g2, w.g1, gof_or(m, p + ".norme.weight", w.dump), gof_or(m, p + ".norme.bias", w.dump + d), rows, d, 1e-5f));
    TRY(bigc_axpy_f32(s, g_out, w.g1, (long)rows * d));                                             // g_out = d o (after its dropout)
    if (dr.p > 0.f) TRY(bigc_dropout_f32(s, g_out, rows, d, d, dr.p, dr.seed, dr.base + 3));       // -> d (attn + res)
    TRY(B.lin(p + ".mh_attn.out_proj", E.out_proj, g_out, d, S.att, d, rows, d, d, w.g1, d, false));  // g1 = d att ; g_out also = d c3
    TRY(bigc_fill_offsets((hipStream_t)s, w.off, S.n_seq + 1, S.T));
    TRY(bigc_mha_bwd_drop_f32(s, S.qkv, 3 * d, S.qkv + d, 3 * d, S.qkv + 2 * d, 3 * d, S.att, d, w.g1, d, S.lse, w.delta, w.gqkv, 3 * d, w.gqkv + d,
                              3 * d, w.gqkv + 2 * d, 3 * d, w.off, w.off, S.n_seq, S.T, S.T, rows, 8, d / 8, dr.p, dr.seed, dr.base + 2));
    if (float* gw = gof(m, p + ".mh_attn.in_proj_weight"))
        TRY(bigc_linear_dw_f32(s, w.gqkv, 3 * d, S.ln[4], d, nullptr, 1, d, gw, d, gof(m, p + ".mh_attn.in_proj_bias"), rows, 3 * d, d));
    TRY(B.dx(E.in_proj.w, w.gqkv, 3 * d, rows, 3 * d, d, w.g2, d, false));                          // g2 = d ln[4]
    // g_out = d c[3] (residual path); conv blocks in reverse
    for (int i = 3; i >= 0; --i) {
        const std::string pn = p + ".norm_seq." + std::to_string(i);
        TRY(bigc_layernorm_bwd_f32(s, S.c[i], E.norm_seq[i].w, w.g2, w.g1, gof_or(m, pn + ".weight", w.dump), gof_or(m, pn + ".bias", w.dump + d), rows, d, 1e-5f));
        TRY(bigc_axpy_f32(s, g_out, w.g1, (long)rows * d));                                         // g_out = d c[i]
        if (dr.p > 0.f && (i + 1) % 2 == 0) TRY(bigc_dropout_f32(s, g_out, rows, d, d, dr.p * (i + 1) / 4, dr.seed, dr.base + i / 2));
        e = hipMemcpyAsync(w.g1, g_out, sizeof(float) * (size_t)rows * d, hipMemcpyDeviceToDevice, (hipStream_t)s);   // g1 = d r[i]
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "qanet_bwd: memcpy");
        TRY(bigc_relu_bwd_f32(s, w.g1, S.r[i], (long)rows * d));
        TRY(B.dws(p + ".convs." + std::to_string(i), E.conv[i], w.g1, d, d, S.ln[i], S.dw[i], S.n_seq, S.T, d, E.ks, w.g3, w.g2));   // g2 = d ln[i]
        // g_out stays = d res_i (= d c[i-1], or d a for i == 0)
    }
    TRY(bigc_layernorm_bwd_f32(s, S.a, E.normb.w, w.g2, w.g0, gof_or(m, p + ".normb.weight", w.dump), gof_or(m, p + ".normb.bias", w.dump + d), rows, d, 1e-5f));
    TRY(bigc_axpy_f32(s, w.g0, g_out, (long)rows * d));                                              // g0 = d a = d x
    return BIGC_OK;
}

// backward of head_fwd; dY (rows, ldy) -> accumulates d x into g_enc
int head_bwd(GBwd& B, const Head& H, const float* x, const float* dY, int ldy, int n_seq, int T, int d, HeadSave& S, float* g_enc) {
    void* s = B.s; GTW& w = *B.w;
    const int rows = n_seq * T;
    TRY(B.dws(H.key + ".4", H.layer[4], dY, ldy, H.c_out, S.x[4], S.dw[4], n_seq, T, d, 3, w.g3, w.g1));       // g1 = d x[4]
    for (int i = 3; i >= 0; --i) {
        TRY(bigc_relu_bwd_f32(s, w.g1, S.x[i + 1], (long)rows * d));
        const float* xin = (i == 0) ? x : S.x[i];
        TRY(B.dws(H.key + "." + std::to_string(i) + ".0", H.layer[i], w.g1, d, d, xin, S.dw[i], n_seq, T, d, 3, w.g3, w.g2));   // g2 = d x[i]
        float* t = w.g1; w.g1 = w.g2; w.g2 = t;
    }
    TRY(bigc_axpy_f32(s, g_enc, w.g1, (long)rows * d));
    return BIGC_OK;
}

}  // namespace

extern "C" int grd_set_dropout(grd_model* m, float p, uint32_t seed) {
    BIGC_CHECK_ARG(m && p >= 0.f && p < 1.f, "grd_set_dropout: p must be in [0, 1)");
    m->drop_p = p; m->drop_seed = seed;
    return BIGC_OK;
}

extern "C" int grd_set_grad(grd_model* m, const char* key, float* ptr) {
    BIGC_CHECK_ARG(m && key, "grd_set_grad: bad argument");
    if (ptr) m->grads[key] = ptr; else m->grads.erase(key);
    return BIGC_OK;
}

extern "C" size_t grd_train_workspace_bytes(const grd_model* m, int T, int n) {
    if (!m || T <= 0 || n <= 0) return 0;
    Arena a{nullptr, (size_t)-1};
    GTW w;
    gt_layout(m, T, n, a, w);
    return a.off + 256;
}

extern "C" int grd_train_forward(grd_model* m, void* stream, const float* video_feat, int T, const int32_t* quint, const int32_t* dura, int n,
                                 float video_len, const float* pos_video, const float* pos_query, void* workspace, size_t workspace_bytes,
                                 float* regrs, float* conf, float* cls) {
    StaticWeights static_weights(m ? m->epoch : 0);
    ModeScope mode_scope(m ? m->mode_exact : -1, m ? m->mode_fast : -1);
    SplitRecorder recorder(m ? &m->train_jobs : nullptr);
    BIGC_CHECK_ARG(m && video_feat && quint && dura && pos_video && pos_query && regrs && conf && cls, "grd_train_forward: null pointer");
    BIGC_CHECK_ARG(m->finalized && T > 0 && T <= 2048 && n > 0 && video_len > 0, "grd_train_forward: bad argument");
    const grd_config& c = m->cfg;
    const int d = c.dim_hidden;
    GTW w;
    TRY(open_ws(m, T, n, workspace, workspace_bytes, w));
    void* s = stream;
    TRY(bigc_linear_prepare((hipStream_t)stream, m->train_jobs.data(), (int)m->train_jobs.size()));
    TRY(grd_query_input((hipStream_t)stream, m->enti, m->pred, quint, dura, video_len, w.qin, n, c.dim_clsme, m->kq));
    Lin qlin{m->wq, m->bq};
    TRY(linear(s, w.qin, m->kq, qlin, w.qemb, d, 3 * n, d, m->kq, BIGC_ACT_NONE));
    TRY(linear(s, video_feat, c.dim_feat, m->video_fc, w.vemb, d, T, d, c.dim_feat, BIGC_ACT_NONE));
    const QDrop dv{m->drop_p, m->drop_seed, 0x100}, dq{m->drop_p, m->drop_seed, 0x200}, dc{m->drop_p, m->drop_seed, 0x300};
    TRY(qanet_fwd(s, m->video_enc, w.vemb, pos_video, d, w.ev, w.off, w.venc, dv));
    TRY(qanet_fwd(s, m->query_enc, w.qemb, pos_query, d, w.eq, w.off, w.qenc, dq));
    TRY(linear(s, w.venc, d, m->proj2sim, w.proj, d, T, d, d, BIGC_ACT_NONE));
    TRY(grd_coattention_f32(s, w.venc, w.proj, w.qenc, w.comb, n, T, d));
    TRY(linear(s, w.comb, 4 * d, m->vq_fc, w.cin, d, n * T, d, 4 * d, BIGC_ACT_NONE));
    TRY(qanet_fwd(s, m->comb_enc, w.cin, pos_video, d, w.ec, w.off, w.cenc, dc));
    TRY(head_fwd(s, m->regr_head, w.cenc, n, T, d, w.hr, regrs, BIGC_ACT_SIGMOID));
    TRY(head_fwd(s, m->conf_head, w.cenc, n, T, d, w.hc, conf, BIGC_ACT_NONE));
    TRY(head_fwd(s, m->cls_head, w.cenc, n, T, d, w.hl, cls, BIGC_ACT_NONE));
    return BIGC_OK;
}

extern "C" int grd_loss_grad(void* stream, const float* regrs, const float* conf, const float* cls, const float* clip, const float* target,
                             const int32_t* pos_i, const int32_t* pos_b, const int32_t* pos_j, int n_pos, const int32_t* neg_i,
                             const int32_t* neg_b, int n_neg, int T, int k, float count_pos, float count_neg, float count_mask,
                             float f_cls, float f_ct, float f_regr, const float* gs_host, float* loss, float* dregr, float* dconf, float* dcls) {
    BIGC_CHECK_ARG(regrs && conf && cls && clip && target && loss && dregr && dconf && dcls && gs_host, "grd_loss_grad: null pointer");
    if (n_pos + n_neg <= 0) return BIGC_OK;
    GrdLossArgs p;
    p.regrs = regrs; p.conf = conf; p.cls = cls; p.clip = clip; p.target = target;
    p.pos_i = pos_i; p.pos_b = pos_b; p.pos_j = pos_j; p.neg_i = neg_i; p.neg_b = neg_b;
    p.n_pos = n_pos; p.n_neg = n_neg; p.T = T; p.k = k; p.ldc = round4(k); p.ldr = round4(2 * k);
    p.inv_pos = count_pos > 0 ? 1.f / count_pos : 0.f; p.inv_neg = count_neg > 0 ? 1.f / count_neg : 0.f;
    p.inv_mask = count_mask > 0 ? 1.f / count_mask : 0.f;
    p.f_cls = f_cls; p.f_ct = f_ct; p.f_regr = f_regr;
    for (int a = 0; a < 5; ++a) p.gs[a] = gs_host[a];
    p.loss = loss; p.dregr = dregr; p.dconf = dconf; p.dcls = dcls;
    ProfScope ps(stream, FAM_LOSS, 0.0, 0.0);
    const int grid = grid_for((long)(n_pos + n_neg) * T, 1024);
    if (int rc = bigc_scratch_floats((hipStream_t)stream, (size_t)grid * 5, 2, &p.part)) return rc;
    hipLaunchKernelGGL(grd_loss_grad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    BIGC_LAUNCH_CHECK("grd_loss_grad");
    return bigc_partial_rows_finish((hipStream_t)stream, p.part, loss, nullptr, grid, 5, 5);
}

#define GZERO(ptr, count) do { hipError_t e_ = hipMemsetAsync((ptr), 0, sizeof(float) * (size_t)(count), hs); \
    if (e_ != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "grd_train_backward: memset: %s", hipGetErrorString(e_)); } while (0)

extern "C" int grd_train_backward(grd_model* m, void* stream, const float* video_feat, int T, const int32_t* quint, int n,
                                  void* workspace, size_t workspace_bytes, const float* dregr, const float* dconf, const float* dcls) {
    StaticWeights static_weights(m ? m->epoch : 0);
    ModeScope mode_scope(m ? m->mode_exact : -1, m ? m->mode_fast : -1);
    SplitRecorder recorder(m ? &m->train_jobs : nullptr);
    BIGC_CHECK_ARG(m && video_feat && quint && dregr && dconf && dcls, "grd_train_backward: null pointer");
    const grd_config& c = m->cfg;
    const int d = c.dim_hidden, k = c.num_bins;
    BIGC_CHECK_ARG(m->finalized && T > 0 && n > 0 && 12 * (d + T) + 4 <= 16384, "grd_train_backward: T=%d too long for the co-attention backward (12(d+T)+4 floats of LDS)", T);
    GTW w;
    TRY(open_ws(m, T, n, workspace, workspace_bytes, w));
    void* s = stream;
    hipStream_t hs = (hipStream_t)stream;
    GBwd B{m, s, &w};
    const int rc = n * T;
    // heads -> d combined encoder output (accumulated in gcomb[0 : rc*d])
    float* g_enc = w.gcomb;
    GZERO(g_enc, (size_t)rc * d);
    TRY(head_bwd(B, m->regr_head, w.cenc, dregr, round4(2 * k), n, T, d, w.hr, g_enc));
    TRY(head_bwd(B, m->conf_head, w.cenc, dconf, round4(k), n, T, d, w.hc, g_enc));
    TRY(head_bwd(B, m->cls_head, w.cenc, dcls, round4(k), n, T, d, w.hl, g_enc));
    // combined encoder
    const QDrop dv{m->drop_p, m->drop_seed, 0x100}, dq{m->drop_p, m->drop_seed, 0x200}, dc{m->drop_p, m->drop_seed, 0x300};
    TRY(qanet_bwd(B, m->comb_enc, d, w.ec, g_enc, dc));                                   // w.g0 = d cin
    TRY(B.lin("vq_fc", m->vq_fc, w.g0, d, w.comb, 4 * d, rc, d, 4 * d, w.gcomb, 4 * d, false));    // gcomb = d comb (n*T, 4d)
    // co-attention
    GZERO(w.gvenc, (size_t)T * d);
    GZERO(w.gproj, (size_t)T * d);
    {
        const size_t lds = sizeof(float) * (12 * (size_t)d + 12 * (size_t)T + 4);
        ProfScope ps(stream, FAM_GROUNDING, 0.0, 0.0);
        float* part = nullptr;
        TRY(bigc_scratch_floats(hs, (size_t)n * 2 * T * d, 3, &part));
        hipLaunchKernelGGL(coattention_bwd_kernel, dim3(n), dim3(256), lds, hs, w.venc, w.proj, w.qenc, w.gcomb, part, w.gqenc, T, d);
        BIGC_LAUNCH_CHECK("coattention_bwd");
        TRY(bigc_partial_rows_finish(hs, part, w.gvenc, w.gproj, n, 2 * T * d, T * d));
    }
    TRY(B.lin("proj2sim", m->proj2sim, w.gproj, d, w.venc, d, T, d, d, w.gvenc, d, true, false));
    // video branch
    TRY(qanet_bwd(B, m->video_enc, d, w.ev, w.gvenc, dv));                                 // w.g0 = d vemb
    TRY(B.lin("video_fc", m->video_fc, w.g0, d, video_feat, c.dim_feat, T, d, c.dim_feat, nullptr, 0, false));
    // query branch
    TRY(qanet_bwd(B, m->query_enc, d, w.eq, w.gqenc, dq));                                 // w.g0 = d qemb (3n, d)
    {
        // the fused input layer [query_fc | temp_fc]: weight gradients split by column range, biases both = column sums
        float* gq = gof(m, "query_fc.weight");
        float* gt = gof(m, "temp_fc.weight");
        if (gq) TRY(bigc_linear_dw_f32(s, w.g0, d, w.qin, m->kq, nullptr, 1, c.dim_clsme, gq, c.dim_clsme, gof(m, "query_fc.bias"), 3 * n, d, c.dim_clsme));
        if (gt) TRY(bigc_linear_dw_f32(s, w.g0, d, w.qin + c.dim_clsme, m->kq, nullptr, 1, 2, gt, 2, gof(m, "temp_fc.bias"), 3 * n, d, 2));
        float* ge = gof(m, "EntiNameEmb");
        float* gp = gof(m, "PredNameEmb");
        if (ge || gp) {
            TRY(B.dx(m->wq, w.g0, d, 3 * n, d, m->kq, w.gqin, m->kq, false));
            // d of the query input rows (3n, kq): the first dim_clsme columns are embedding rows (word l of query qi: subject = quint column 1,
            // predicate = column 0, object = column 2) -> added to the two tables in query order per table row (no float atomics)
            const int de = c.dim_clsme;
            if (ge) TRY(bigc_ordered_rows_add(hs, w.gqin, 3 * m->kq, quint + 1, 5, n, de, ge, de));
            if (gp) TRY(bigc_ordered_rows_add(hs, w.gqin + m->kq, 3 * m->kq, quint + 0, 5, n, de, gp, de));
            if (ge) TRY(bigc_ordered_rows_add(hs, w.gqin + 2 * m->kq, 3 * m->kq, quint + 2, 5, n, de, ge, de));
        }
    }
    return BIGC_OK;
}
