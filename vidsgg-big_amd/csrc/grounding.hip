// Temporal grounding scorer on the HIP path: eval forward of the reference's ``DEBUG`` model
// (models/grd_model_v5.py:196-221,310-373,530-576,667-737) for one video.
//
// Kernels specific to this stage (the dense layers reuse linear_f32 / mha_core / layernorm / add_rows):
//   dwconv            depth-wise temporal conv (k=7 / k=3) on (sequence, time, channel) rows   -- HBM-bound
//   query_input       embedding gather [subject, predicate, object] + normalised overlap span  -- gather
//   coattention       video<->query similarity, row/column softmax, A and B, 4d feature fusion -- per query in LDS
//   postprocess       sigmoid scores, per-(query,bin) temporal pooling, overlap clipping, 1-D NMS, masks
//                     (replaces the reference's two Python loops, :711-733 and :688-692)
//
// Layout: every activation is (rows, d) row-major with rows = sequence-major time (n*T for the combined
// encoder, 3n for the query encoder, T for the video encoder); the reference's (N, d, T) tensors are
// the transposes, so Conv1d(k=1) is a plain linear and LayerNorm needs no transposes.
#include "grounding_internal.h"
#include "linear_internal.h"
#include "prof.h"

namespace {

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                     float* __restrict__ y, long total4, int T, int d4, int ks) {
    const int half = ks / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long row = i / d4;
        const int c4 = (int)(i - row * d4);
        const int t = (int)(row % T);
        f32x4 acc = ((const f32x4*)b)[c4];
        const float* wp = w + (size_t)c4 * 4 * ks;
        for (int j = 0; j < ks; ++j) {
            const int tt = t + j - half;
            if (tt < 0 || tt >= T) continue;
            const f32x4 xv = ((const f32x4*)x)[(row + (j - half)) * d4 + c4];
            acc[0] += wp[j] * xv[0];
            acc[1] += wp[ks + j] * xv[1];
            acc[2] += wp[2 * ks + j] * xv[2];
            acc[3] += wp[3 * ks + j] * xv[3];
        }
        ((f32x4*)y)[i] = acc;
    }
}

// ---- several videos per call: sequences of different lengths back to back -------------------------------------------------
// row_info[r] = (position of row r inside its sequence, length of that sequence); one workgroup per sequence
__global__ __launch_bounds__(256) void seq_rows_kernel(const int32_t* __restrict__ seq_off, int2* __restrict__ row_info) {
    const int r0 = seq_off[blockIdx.x], T = seq_off[blockIdx.x + 1] - r0;
    for (int t = threadIdx.x; t < T; t += 256) row_info[r0 + t] = make_int2(t, T);
}

__global__ __launch_bounds__(256) void dwconv_ragged_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                            float* __restrict__ y, long total4, const int2* __restrict__ row_info, int d4, int ks) {
    const int half = ks / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long row = i / d4;
        const int c4 = (int)(i - row * d4);
        const int2 tT = row_info[row];
        f32x4 acc = ((const f32x4*)b)[c4];
        const float* wp = w + (size_t)c4 * 4 * ks;
        for (int j = 0; j < ks; ++j) {
            const int tt = tT.x + j - half;
            if (tt < 0 || tt >= tT.y) continue;
            const f32x4 xv = ((const f32x4*)x)[(row + (j - half)) * d4 + c4];
            acc[0] += wp[j] * xv[0];
            acc[1] += wp[ks + j] * xv[1];
            acc[2] += wp[2 * ks + j] * xv[2];
            acc[3] += wp[3 * ks + j] * xv[3];
        }
        ((f32x4*)y)[i] = acc;
    }
}

// The depth-wise convolution as a stream: a thread owns four channels and a run of DW_RUN consecutive rows, keeps the ks input rows
// under the filter in registers (each row is loaded once per run, not once per tap: the per-tap version above issues 5 loads per
// tap and output quad -- 35 for k = 7 -- and runs at 2.2 TB/s) and its 4 x ks filter weights too.  Same operation order per output
// as dwconv_kernel (bias, then the taps inside the sequence in ascending order).  row_info == nullptr: sequences of T rows.
constexpr int DW_RUN = 16;
template <int KS>
__global__ __launch_bounds__(256) void dwconv_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ y, long rows, int T, const int2* __restrict__ row_info, int d4) {
    constexpr int H = KS / 2;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = (int)(g % d4);
    const long r0 = (g / d4) * DW_RUN;
    if (r0 >= rows) return;
    float wt[4][KS];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < KS; ++j) wt[c][j] = w[(size_t)(4 * c4 + c) * KS + j];
    const f32x4 bias = ((const f32x4*)b)[c4];
    const f32x4* xp = (const f32x4*)x + c4;
    f32x4 win[KS];                                            // win[j] = row r + j - H (clamped to the tensor; taps outside the sequence are skipped)
#pragma unroll
    for (int j = 1; j < KS; ++j) win[j] = xp[min(max(r0 + j - 1 - H, 0L), rows - 1) * d4];
    const long r1 = min(r0 + DW_RUN, rows);
    for (long r = r0; r < r1; ++r) {
#pragma unroll
        for (int j = 0; j + 1 < KS; ++j) win[j] = win[j + 1];
        win[KS - 1] = xp[min(r + H, rows - 1) * d4];
        int t, len;
        if (row_info) { const int2 v = row_info[r]; t = v.x; len = v.y; } else { t = (int)(r % T); len = T; }
        f32x4 acc = bias;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int tt = t + j - H;
            if (tt < 0 || tt >= len) continue;
            acc[0] += wt[0][j] * win[j][0];
            acc[1] += wt[1][j] * win[j][1];
            acc[2] += wt[2][j] * win[j][2];
            acc[3] += wt[3][j] * win[j][3];
        }
        ((f32x4*)y)[r * d4 + c4] = acc;
    }
}

// LayerNorm + depth-wise convolution in one pass (the four LN -> DepthWiseSeparableConv pairs of a QANet encoder layer,
// models/grd_model_v5.py:126-131: the normalised tensor feeds only the convolution): d = 128, a run of rows per group of 32 lanes
// (four channels per lane), the statistics of every row entering the filter window by butterfly shuffles inside the group --
// the arithmetic of layernorm_rows_kernel<32, 1, ...> (rowwise.hip) operation for operation -- and the normalised rows never
// touch memory.  Halo rows (ks - 1 per run) are normalised twice.
template <int KS>
__global__ __launch_bounds__(256) void ln_dwconv_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                                             long rows, int T, const int2* __restrict__ row_info) {
    constexpr int H = KS / 2, D4 = 32;
    const int c4 = threadIdx.x & 31;
    const long r0 = ((long)blockIdx.x * 8 + (threadIdx.x >> 5)) * DW_RUN;
    if (r0 >= rows) return;                                   // (whole 32-lane groups leave together)
    float wt[4][KS];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < KS; ++j) wt[c][j] = w[(size_t)(4 * c4 + c) * KS + j];
    const f32x4 bias = ((const f32x4*)b)[c4], gg = ((const f32x4*)gamma)[c4], bb = ((const f32x4*)beta)[c4];
    const f32x4* xp = (const f32x4*)x + c4;
    auto gsum = [](float t) {
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) t += __shfl_xor(t, o);
        return t;
    };
    auto ln_row = [&](long r) {
        const f32x4 v = xp[min(max(r, 0L), rows - 1) * D4];
        const float mean = gsum((v[0] + v[1]) + (v[2] + v[3])) / (float)(4 * D4);
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float t = v[c] - mean; ss += t * t; }
        const float rstd = 1.0f / sqrtf(gsum(ss) / (float)(4 * D4) + eps);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = (v[c] - mean) * rstd * gg[c] + bb[c];
        return o;
    };
    f32x4 win[KS];
#pragma unroll
    for (int j = 1; j < KS; ++j) win[j] = ln_row(r0 + j - 1 - H);
    const long r1 = min(r0 + DW_RUN, rows);
    for (long r = r0; r < r1; ++r) {
#pragma unroll
        for (int j = 0; j + 1 < KS; ++j) win[j] = win[j + 1];
        win[KS - 1] = ln_row(r + H);
        int t, len;
        if (row_info) { const int2 v = row_info[r]; t = v.x; len = v.y; } else { t = (int)(r % T); len = T; }
        f32x4 acc = bias;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int tt = t + j - H;
            if (tt < 0 || tt >= len) continue;
            acc[0] += wt[0][j] * win[j][0];
            acc[1] += wt[1][j] * win[j][1];
            acc[2] += wt[2][j] * win[j][2];
            acc[3] += wt[3][j] * win[j][3];
        }
        ((f32x4*)y)[r * D4 + c4] = acc;
    }
}

// y = x + pos[position of the row in its sequence]
__global__ __launch_bounds__(256) void add_pos_ragged_kernel(const float* __restrict__ x, const float* __restrict__ pos, float* __restrict__ y,
                                                             long n4, int d4, const int2* __restrict__ row_info) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long row = i / d4;
        const int c = (int)(i - row * d4);
        ((f32x4*)y)[i] = ((const f32x4*)x)[i] + ((const f32x4*)pos)[(size_t)row_info[row].x * d4 + c];
    }
}

// rows (3n, kq): [emb(300) | s/len | e/len | 0 0]; word order [subject, predicate, object] (:323)
__global__ __launch_bounds__(256) void query_input_kernel(const float* __restrict__ enti, const float* __restrict__ pred,
                                                          const int32_t* __restrict__ quint, const int32_t* __restrict__ dura,
                                                          float video_len, const int32_t* __restrict__ q_vid,
                                                          const float* __restrict__ vid_len, float* __restrict__ out, int n, int de, int kq) {
    const long total = (long)3 * n * kq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / kq;
        const int c = (int)(i - row * kq);
        const int q = (int)(row / 3), l = (int)(row - 3 * (long)q);
        float v = 0.f;
        if (c < de) {
            const int id = quint[q * 5 + (l == 0 ? 1 : (l == 1 ? 0 : 2))];
            v = (l == 1 ? pred : enti)[(size_t)id * de + c];
        } else if (c < de + 2) {
            v = (float)dura[q * 2 + (c - de)] / (q_vid ? vid_len[q_vid[q]] : video_len);       // uniq_dura_inters.float() / video_len (:324)
        }
        out[i] = v;
    }
}

// Wq (d, kq) = [query_fc.weight | temp_fc.weight | 0 0], bq = query_fc.bias + temp_fc.bias
__global__ void build_query_weight_kernel(const float* __restrict__ wq, const float* __restrict__ bq, const float* __restrict__ wt,
                                          const float* __restrict__ bt, float* __restrict__ w_out, float* __restrict__ b_out,
                                          int d, int de, int kq) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d * kq) {
        const int r = i / kq, c = i - r * kq;
        w_out[i] = c < de ? wq[r * de + c] : (c < de + 2 ? wt[r * 2 + (c - de)] : 0.f);
    }
    if (i < d) b_out[i] = bq[i] + bt[i];
}

// info (n,2) = dura / video_len
__global__ void span_info_kernel(const int32_t* __restrict__ dura, float video_len, const int32_t* __restrict__ q_vid,
                                 const float* __restrict__ vid_len, float* __restrict__ info, int n2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n2) info[i] = (float)dura[i] / (q_vid ? vid_len[q_vid[i >> 1]] : video_len);
}

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

// reference: models/grd_model_v5.py:346-365
__global__ __launch_bounds__(256) void coattention_kernel(const float* __restrict__ video, const float* __restrict__ proj,
                                                          const float* __restrict__ query, float* __restrict__ out, int T, int d,
                                                          const int32_t* __restrict__ q_vid, const int32_t* __restrict__ vid_row_off,
                                                          const int32_t* __restrict__ q_row_off) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    size_t out_row = (size_t)blockIdx.x * T;
    if (q_vid) {                    // several videos: this query's video rows, its own T and its place in the output
        const int v = q_vid[blockIdx.x], r0 = vid_row_off[v];
        T = vid_row_off[v + 1] - r0;
        video += (size_t)r0 * d;
        proj += (size_t)r0 * d;
        out_row = q_row_off[blockIdx.x];
    }
    float* q = lds;                 // 3*d
    float* s2 = q + 3 * d;          // 3*d
    float* sim_r = s2 + 3 * d;      // 3*T
    float* sim_c = sim_r + 3 * T;   // 3*T (raw scores first)
    float* red = sim_c + 3 * T;     // 4
    const int i = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < 3 * d; c += 256) q[c] = query[(size_t)i * 3 * d + c];
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        const f32x4* pr = (const f32x4*)(proj + (size_t)t * d);
        float s[3] = {0.f, 0.f, 0.f};
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
        m = block_reduce(m, true, red);
        float sum = 0.f;
        for (int t = tid; t < T; t += 256) {
            const float e = expf(sim_c[l * T + t] - m);
            sim_c[l * T + t] = e;
            sum += e;
        }
        sum = block_reduce(sum, false, red);
        for (int t = tid; t < T; t += 256) sim_c[l * T + t] /= sum;
    }
    __syncthreads();
    for (int e = tid; e < 3 * d; e += 256) {            // S2 = sim_c^T @ video  (3, d)
        const int l = e / d, c = e - l * d;
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += sim_c[l * T + t] * video[(size_t)t * d + c];
        s2[e] = acc;
    }
    __syncthreads();
    float* ob = out + out_row * 4 * d;
    for (int e = tid; e < T * d; e += 256) {
        const int t = e / d, c = e - t * d;
        const float v = video[e];
        const float r0 = sim_r[t], r1 = sim_r[T + t], r2 = sim_r[2 * T + t];
        const float a = r0 * q[c] + r1 * q[d + c] + r2 * q[2 * d + c];
        const float b = r0 * s2[c] + r1 * s2[d + c] + r2 * s2[2 * d + c];
        float* o = ob + (size_t)t * 4 * d + c;
        o[0] = v; o[d] = a; o[2 * d] = a * v; o[3 * d] = b * v;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// order of torch.max / argmax: a NaN is the maximum (the first one wins) -- with plain comparisons a sequence of NaN scores left the arg-max
// at its initial value and the kernel read clip[0x7fffffff] (found by tests/test_hip_grounding.py: a NaN weight in a head)
__device__ __forceinline__ bool score_gt(float a, float b) { return a > b || (a != a && b == b); }
__device__ __forceinline__ bool score_eq(float a, float b) { return a == b || (a != a && b != b); }

// reference: models/grd_model_v5.py:533-572, 697-737, 667-695; tIoU utils/utils_func.py:375-390
// One workgroup of 256 threads per query.  Round 6: all bins at once -- thread (row slot tr, bin b) of the first TPR * k threads walks the clips
// tr, tr + TPR, ...: consecutive threads read consecutive floats of the (T, k) score arrays (one wave per query walking one bin after the
// other read 4 bytes of every 40 and took 102 us for 1098 queries; the maxima and the pooled spans are order-independent reductions with the
// explicit lowest-clip tie-break, so the results are the same bits).
constexpr int PP_THREADS = 256;
__global__ __launch_bounds__(PP_THREADS) void postprocess_kernel(const float* __restrict__ regrs, const float* __restrict__ conf,
                                                         const float* __restrict__ cls, const float* __restrict__ clip,
                                                         const float* __restrict__ info, int T, int k, float score_th, float tiou_th,
                                                         float bins_th, float nms_th, float* __restrict__ pooled_se,
                                                         float* __restrict__ bins_probs, uint8_t* __restrict__ bins_mask,
                                                         int32_t* __restrict__ status, const int32_t* __restrict__ q_vid,
                                                         const int32_t* __restrict__ vid_row_off, const int32_t* __restrict__ q_row_off) {
    __shared__ float sp_s[17], sp_e[17], prob[17];
    __shared__ int ovl[17];
    __shared__ float r_v[PP_THREADS], r_e[PP_THREADS];
    __shared__ int r_t[PP_THREADS];
    __shared__ float top_s[16], top_e[16], top_v[16];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid;
    const float is = info[2 * i], ie = info[2 * i + 1];
    size_t row0 = (size_t)i * T;
    if (q_vid) {                    // several videos: clip = the concatenated per-video linspace tables
        const int v = q_vid[i], r0 = vid_row_off[v];
        T = vid_row_off[v + 1] - r0;
        clip += r0;
        row0 = q_row_off[i];
    }
    const float* cf = conf + row0 * k;
    const float* cl = cls + row0 * k;
    const float* rg = regrs + row0 * 2 * k;
    const int TPR = PP_THREADS / k;                       // clip rows in flight (k <= 16: at least 16)
    const bool act = tid < TPR * k;
    const int b = act ? tid % k : 0, tr = act ? tid / k : 0;
    // pass 1: per bin the largest score, lowest clip on ties
    {
        float bv = -INFINITY;
        int bt = 0x7fffffff;
        if (act)
            for (int t = tr; t < T; t += TPR) {
                const float sc = sigmoidf_(cf[t * k + b]) * sigmoidf_(cl[t * k + b]);
                if (score_gt(sc, bv) || (score_eq(sc, bv) && t < bt)) { bv = sc; bt = t; }
            }
        r_v[tid] = bv; r_t[tid] = bt;
        __syncthreads();
        if (tid < k) {
            float v = r_v[tid];
            int t = r_t[tid];
            for (int j = 1; j < TPR; ++j) {
                const float ov = r_v[j * k + tid];
                const int ot = r_t[j * k + tid];
                if (score_gt(ov, v) || (score_eq(ov, v) && ot < t)) { v = ov; t = ot; }
            }
            if (t >= T) t = 0;                            // (T >= 1: cannot happen; keeps the reads below inside the sequence whatever the scores are)
            top_v[tid] = v;
            top_s[tid] = clip[t] - rg[t * 2 * k + tid];
            top_e[tid] = clip[t] + rg[t * 2 * k + k + tid];
        }
        __syncthreads();
    }
    // pass 2: per bin the span of the clips that score above score_th * top and overlap the top clip's span
    {
        const float bv = top_v[b], ts = top_s[b], te = top_e[b];
        const float thr = score_th * bv;
        float ps = INFINITY, pe = -INFINITY;
        if (act)
            for (int t = tr; t < T; t += TPR) {
                const float sc = sigmoidf_(cf[t * k + b]) * sigmoidf_(cl[t * k + b]);
                const float st = clip[t] - rg[t * 2 * k + b], en = clip[t] + rg[t * 2 * k + k + b];
                const float g = (fminf(te, en) - fmaxf(ts, st)) / (fmaxf(te, en) - fminf(ts, st));
                if (sc > thr && g > tiou_th) { ps = fminf(ps, st); pe = fmaxf(pe, en); }
            }
        r_v[tid] = ps; r_e[tid] = pe;
        __syncthreads();
        if (tid < k) {
            float ps2 = r_v[tid], pe2 = r_e[tid];
            for (int j = 1; j < TPR; ++j) { ps2 = fminf(ps2, r_v[j * k + tid]); pe2 = fmaxf(pe2, r_e[j * k + tid]); }
            if (ps2 == INFINITY) { status[0] = 1; if (q_vid) status[1 + q_vid[i]] = 1; ps2 = top_s[tid]; pe2 = top_e[tid]; }
            const float s_ = fmaxf(is, ps2), e_ = fminf(ie, pe2);
            const bool ov = s_ <= e_;
            sp_s[tid] = ov ? s_ : is; sp_e[tid] = ov ? e_ : ie; ovl[tid] = ov; prob[tid] = top_v[tid];
        }
        __syncthreads();
    }
    if (lane == 0) {
        sp_s[k] = is; sp_e[k] = ie; ovl[k] = 1; prob[k] = 1.0f;
        const int nb = k + 1;
        int order[17];
        for (int a = 0; a < nb; ++a) {                 // stable ascending insertion sort by probability
            int j = a;
            while (j > 0 && prob[order[j - 1]] > prob[a]) { order[j] = order[j - 1]; --j; }
            order[j] = a;
        }
        bool alive[17], keep[17];
        for (int a = 0; a < nb; ++a) { alive[a] = true; keep[a] = false; }
        for (int pos = nb - 1; pos >= 0; --pos) {
            const int a = order[pos];
            if (!alive[a]) continue;
            keep[a] = true;
            for (int p2 = pos - 1; p2 >= 0; --p2) {
                const int c = order[p2];
                if (!alive[c]) continue;
                const bool ok = (sp_e[a] >= sp_s[c]) && (sp_e[c] >= sp_s[a]);
                float t = (fminf(sp_e[a], sp_e[c]) - fmaxf(sp_s[a], sp_s[c])) / (fmaxf(sp_e[a], sp_e[c]) - fminf(sp_s[a], sp_s[c]));
                if (!ok) t = 0.f;
                alive[c] = t < nms_th;
            }
        }
        bool any = false, m[17];
        for (int a = 0; a < nb; ++a) { m[a] = (prob[a] > bins_th) && ovl[a] && keep[a]; any = any || m[a]; }
        if (!any) {
            int best = 0;
            for (int a = 1; a < nb; ++a) if (prob[a] > prob[best]) best = a;
            m[best] = true;
        }
        float mx = -INFINITY;
        for (int a = 0; a < k; ++a) mx = fmaxf(mx, prob[a]);
        const float last = (mx <= bins_th) ? 0.0f : prob[k];
        for (int a = 0; a < nb; ++a) {
            pooled_se[((size_t)i * nb + a) * 2] = sp_s[a];
            pooled_se[((size_t)i * nb + a) * 2 + 1] = sp_e[a];
            bins_probs[(size_t)i * nb + a] = (a == k) ? last : prob[a];
            bins_mask[(size_t)i * nb + a] = m[a] ? 1 : 0;
        }
    }
}

int grid_for(long n, int cap = 4096) { long g = (n + 255) / 256; return (int)(g < cap ? (g < 1 ? 1 : g) : cap); }

}  // namespace

// shared with grounding_train.hip
int grd_query_input(hipStream_t hs, const float* enti, const float* pred, const int32_t* quint, const int32_t* dura, float video_len,
                    float* out, int n, int de, int kq) {
    hipLaunchKernelGGL(query_input_kernel, dim3(grid_for((long)3 * n * kq)), dim3(256), 0, hs, enti, pred, quint, dura, video_len, nullptr, nullptr,
                       out, n, de, kq);
    BIGC_LAUNCH_CHECK("query_input");
    return BIGC_OK;
}

extern "C" int bigc_dwconv_f32(void* stream, const float* x, const float* w, const float* b, float* y, int n_seq, int T, int d, int ks) {
    BIGC_CHECK_ARG(x && w && b && y, "dwconv: null pointer");
    BIGC_CHECK_ARG(d % 4 == 0 && ks % 2 == 1 && ks <= 7 && T > 0, "dwconv: d=%d ks=%d T=%d", d, ks, T);
    if (n_seq <= 0) return BIGC_OK;
    const long total4 = (long)n_seq * T * (d / 4);
    ProfScope ps(stream, FAM_GROUNDING, 0.0, 8.0 * n_seq * (double)T * d);
    hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, total4, T, d / 4, ks);
    BIGC_LAUNCH_CHECK("dwconv");
    return BIGC_OK;
}

extern "C" int grd_coattention_f32(void* stream, const float* video, const float* proj, const float* query, float* out, int n, int T, int d) {
    BIGC_CHECK_ARG(video && proj && query && out, "coattention: null pointer");
    BIGC_CHECK_ARG(T > 0 && T <= 2048 && d % 4 == 0 && d <= 256, "coattention: T=%d (<=2048) d=%d (<=256)", T, d);
    if (n <= 0) return BIGC_OK;
    const size_t lds = sizeof(float) * (6 * (size_t)d + 6 * (size_t)T + 4);
    ProfScope ps(stream, FAM_GROUNDING, 0.0, 16.0 * n * (double)T * d);
    hipLaunchKernelGGL(coattention_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, video, proj, query, out, T, d, nullptr, nullptr, nullptr);
    BIGC_LAUNCH_CHECK("coattention");
    return BIGC_OK;
}

extern "C" int grd_postprocess(void* stream, const float* regrs, const float* conf, const float* cls, const float* clip, const float* info,
                               int n, int T, int k, float score_th, float tiou_th, float bins_th, float nms_th,
                               float* pooled_se, float* bins_probs, uint8_t* bins_mask, int32_t* status) {
    BIGC_CHECK_ARG(regrs && conf && cls && clip && info && pooled_se && bins_probs && bins_mask && status, "postprocess: null pointer");
    BIGC_CHECK_ARG(k >= 1 && k <= 16 && T > 0, "postprocess: num_bins=%d must be in [1,16]", k);
    hipError_t e = hipMemsetAsync(status, 0, sizeof(int32_t), (hipStream_t)stream);
    if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "postprocess: memset: %s", hipGetErrorString(e));
    if (n <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_GROUNDING, 0.0, 0.0);
    hipLaunchKernelGGL(postprocess_kernel, dim3(n), dim3(PP_THREADS), 0, (hipStream_t)stream, regrs, conf, cls, clip, info, T, k, score_th, tiou_th,
                       bins_th, nms_th, pooled_se, bins_probs, bins_mask, status, nullptr, nullptr, nullptr);
    BIGC_LAUNCH_CHECK("postprocess");
    return BIGC_OK;
}

// ------------------------------------------------------------------------------------------------
// model-level
// ------------------------------------------------------------------------------------------------
namespace {

struct GResolver {
    grd_model* m;
    int err = 0;
    const float* get(const std::string& key, int64_t numel) {
        auto it = m->params.find(key);
        if (it == m->params.end()) {
            if (!err) err = bigc_fail(BIGC_ERR_MISSING, "parameter '%s' was never set", key.c_str());
            return nullptr;
        }
        if (it->second.second != numel) {
            if (!err) err = bigc_fail(BIGC_ERR_ARG, "parameter '%s' has %lld elements, expected %lld", key.c_str(),
                                      (long long)it->second.second, (long long)numel);
            return nullptr;
        }
        return it->second.first;
    }
    Lin lin(const std::string& p, int n_out, int n_in, bool bias = true) {
        Lin l;
        l.w = get(p + ".weight", (int64_t)n_out * n_in);
        if (bias) l.b = get(p + ".bias", n_out);
        return l;
    }
    Lin ln(const std::string& p, int d) {
        Lin l;
        l.w = get(p + ".weight", d);
        l.b = get(p + ".bias", d);
        return l;
    }
    DW dws(const std::string& p, int d, int c_out, int ks) {
        DW x;
        x.dw_w = get(p + ".depth_wise.weight", (int64_t)d * ks);
        x.dw_b = get(p + ".depth_wise.bias", d);
        x.pw = lin(p + ".point_wise", c_out, d);
        return x;
    }
    Enc enc(const std::string& p, int d, int ks) {
        Enc e;
        e.ks = ks;
        e.key = p;
        for (int i = 0; i < 4; ++i) e.conv[i] = dws(p + ".convs." + std::to_string(i), d, d, ks);
        e.in_proj.w = get(p + ".mh_attn.in_proj_weight", (int64_t)3 * d * d);
        e.in_proj.b = get(p + ".mh_attn.in_proj_bias", 3 * d);
        e.out_proj = lin(p + ".mh_attn.out_proj", d, d);
        e.fc = lin(p + ".fc", d, d);
        e.normb = ln(p + ".normb", d);
        for (int i = 0; i < 4; ++i) e.norm_seq[i] = ln(p + ".norm_seq." + std::to_string(i), d);
        e.norme = ln(p + ".norme", d);
        return e;
    }
    Head head(const std::string& p, int d, int c_out) {
        Head h;
        h.c_out = c_out;
        h.key = p;
        for (int i = 0; i < 4; ++i) h.layer[i] = dws(p + "." + std::to_string(i) + ".0", d, d, 3);
        h.layer[4] = dws(p + ".4", d, c_out, 3);
        return h;
    }
};

struct GArena {
    char* base; size_t cap; size_t off = 0; bool overflow = false;
    template <typename T> T* take(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (off + bytes > cap) { overflow = true; return nullptr; }
        T* p = (T*)(base + off);
        off += bytes;
        return p;
    }
};

struct GWork {
    float *b0, *b1, *b2, *b3, *qkv, *att, *venc, *qenc, *proj, *qin, *comb, *regrs, *conf, *cls, *info;
    int32_t* off;
    int2 *vinfo, *cinfo;      // several videos per call: (position, length) of every video row / every combined row
    float *q_b[4], *q_qkv, *q_att; int32_t* q_off;     // the query encoder's own scratch (3n rows): it runs beside the video encoder on a second stream
    float *vp, *qu;           // vq_fc's clip-only and query-only terms (grd_coatt_fc)
};

// total_T = sum of the videos' clip counts, total_rows = sum over queries of their video's clip count (one video: T and n*T)
bool glayout(const grd_model* m, long total_T, int n, long total_rows, bool ragged, GArena& a, GWork& w) {
    const size_t d = m->cfg.dim_hidden, k = m->cfg.num_bins;
    size_t rows = (size_t)total_rows;
    if (rows < (size_t)3 * n) rows = (size_t)3 * n;
    if (rows < (size_t)total_T) rows = total_T;
    w.b0 = a.take<float>(rows * d); w.b1 = a.take<float>(rows * d); w.b2 = a.take<float>(rows * d); w.b3 = a.take<float>(rows * d);
    w.qkv = a.take<float>(rows * 3 * d);
    w.att = a.take<float>(rows * d);
    w.venc = a.take<float>((size_t)total_T * d);
    w.qenc = a.take<float>((size_t)3 * n * d);
    w.proj = a.take<float>((size_t)total_T * d);
    w.qin = a.take<float>((size_t)3 * n * m->kq);
    w.comb = a.take<float>((size_t)total_rows * 4 * d);
    w.regrs = a.take<float>((size_t)total_rows * 2 * k);
    w.conf = a.take<float>((size_t)total_rows * k);
    w.cls = a.take<float>((size_t)total_rows * k);
    w.info = a.take<float>((size_t)2 * n);
    w.off = a.take<int32_t>(rows + 2);
    w.vinfo = ragged ? a.take<int2>((size_t)total_T) : nullptr;
    w.cinfo = ragged ? a.take<int2>((size_t)total_rows) : nullptr;
    for (int i = 0; i < 4; ++i) w.q_b[i] = a.take<float>((size_t)3 * n * d);
    w.q_qkv = a.take<float>((size_t)3 * n * 3 * d);
    w.q_att = a.take<float>((size_t)3 * n * d);
    w.q_off = a.take<int32_t>((size_t)n + 2);
    w.vp = a.take<float>((size_t)total_T * d);
    w.qu = a.take<float>((size_t)3 * n * d);
    return !a.overflow;
}

#define GTRY(expr) do { int rc_ = (expr); if (rc_ != BIGC_OK) return rc_; } while (0)

inline int glinear(void* s, const float* A, int lda, const Lin& l, float* C, int ldc, int M, int N, int K, int act,
                   const float* res = nullptr, int ldr = 0) {
    return bigc_linear_f32(s, A, lda, nullptr, 1, K, l.w, K, l.b, res, ldr, C, ldc, M, N, K, act);
}

// A set of sequences laid back to back as rows: n_seq of the same length T (info == nullptr), or of different lengths given by
// their row offsets `off` (n_seq + 1) and the per-row (position, length) table `info` (several videos per call).
struct Seqs { int n_seq; int T; long rows; int max_T; const int32_t* off; const int2* info; };

int seq_dwconv(void* s, const float* x, const DW& c, float* y, const Seqs& q, int d, int ks) {
    if (ks == 3 || ks == 7) {
        ProfScope ps(s, FAM_GROUNDING, 0.0, 8.0 * q.rows * (double)d);
        const long threads = ((q.rows + DW_RUN - 1) / DW_RUN) * (d / 4);
        const int grid = (int)((threads + 255) / 256);
        if (ks == 3) hipLaunchKernelGGL(dwconv_rows_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)s, x, c.dw_w, c.dw_b, y, q.rows, q.T, q.info, d / 4);
        else hipLaunchKernelGGL(dwconv_rows_kernel<7>, dim3(grid), dim3(256), 0, (hipStream_t)s, x, c.dw_w, c.dw_b, y, q.rows, q.T, q.info, d / 4);
        BIGC_LAUNCH_CHECK("dwconv_rows");
        return BIGC_OK;
    }
    if (!q.info) return bigc_dwconv_f32(s, x, c.dw_w, c.dw_b, y, q.n_seq, q.T, d, ks);
    const long total4 = q.rows * (d / 4);
    ProfScope ps(s, FAM_GROUNDING, 0.0, 8.0 * q.rows * (double)d);
    hipLaunchKernelGGL(dwconv_ragged_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)s, x, c.dw_w, c.dw_b, y, total4, q.info, d / 4, ks);
    BIGC_LAUNCH_CHECK("dwconv_ragged");
    return BIGC_OK;
}

// whether a depth-wise-separable layer runs with its convolution inside the GEMM's operand staging (see dwsep)
bool dwsep_fused(int n_out, int d, int ks) {
    // k = 7 with a residual (the QANet encoders): the streaming depth-wise kernel + the plain GEMM beat the convolution evaluated in
    // the GEMM's operand staging (60 + 80 us against 221 us per layer on 240 k rows: seven taps per staged quad leave that kernel one
    // 4-wave workgroup per CU); k = 3 (the heads) stays fused (105 against 50 + 65 us).  BIGC_DWSEP_FUSED=1 / 0 forces either form.
    static const int fused_env = getenv("BIGC_DWSEP_FUSED") ? atoi(getenv("BIGC_DWSEP_FUSED")) : -1;
    const bool fuse = fused_env >= 0 ? fused_env != 0 : ks != 7;
    return fuse && bigc_linear_dwsep_supported(n_out, d, ks);
}

// out = act(point_wise(depth_wise(x))) (+ res): one pass over x when the split-operand GEMM can evaluate the depth-wise
// convolution while it stages its rows, else the two kernels (scratch: `dw`)
int dwsep(void* s, const float* x, const DW& c, float* dw, float* out, int ldo, int n_out, const Seqs& q, int d, int ks, int act,
          const float* res = nullptr, int ldr = 0) {
    if (dwsep_fused(n_out, d, ks))
        return bigc_linear_dwsep(s, x, d, c.dw_w, c.dw_b, ks, q.T, q.info, c.pw.w, d, c.pw.b, res, ldr, out, ldo, (int)q.rows, n_out, d, act);
    GTRY(seq_dwconv(s, x, c, dw, q, d, ks));
    return glinear(s, dw, d, c.pw, out, ldo, (int)q.rows, n_out, d, act, res, ldr);
}

// QANetEncoderLayer.forward, eval mode (models/grd_model_v5.py:110-137). x (rows, d) -> out (same shape).
// Uses w.b0..b3, w.qkv, w.att, w.off; `out` may be one of the caller's buffers but not b0..b3.
int qanet(void* s, grd_model* m, int which, const Enc& E, const float* x, const float* pos, const Seqs& q, int d, const GWork& w, float* out) {
    const int rows = (int)q.rows;
    float *res = w.b0, *ln = w.b1, *dw = w.b2, *nxt = w.b3;
    if (grd_enc_chain_supported(d, E.ks)) {
        // the convolution stack in one launch and the two GEMMs behind the attention in another (csrc/grounding_encoder.hip)
        if (!m->enc_frag[which] && hipMalloc(&m->enc_frag[which], grd_enc_frag_bytes()) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "grd_forward: hipMalloc failed");
        const bool refresh = m->enc_frag_epoch[which] != m->epoch;
        // (the attention's packed in_proj runs inside the chain kernel on the normalised tile: BIGC_ENC_QKV=0 = as a GEMM of its own)
        static const bool qkv_off = getenv("BIGC_ENC_QKV") && atoi(getenv("BIGC_ENC_QKV")) == 0;
        GTRY(grd_enc_chain((hipStream_t)s, E, x, pos, q.rows, q.T, q.info, m->enc_frag[which], refresh, res, ln, qkv_off ? nullptr : w.qkv));
        m->enc_frag_epoch[which] = m->epoch;
        if (qkv_off) GTRY(glinear(s, ln, d, E.in_proj, w.qkv, 3 * d, rows, 3 * d, d, BIGC_ACT_NONE));
        const int32_t* off = q.off;
        if (!q.info) { GTRY(bigc_fill_offsets((hipStream_t)s, w.off, q.n_seq + 1, q.T)); off = w.off; }
        bigc_prof_hint(4.0 * (double)rows * ((double)rows / q.n_seq) * d, 16.0 * rows * (double)d);
        GTRY(bigc_mha_core_f32(s, w.qkv, 3 * d, w.qkv + d, 3 * d, w.qkv + 2 * d, 3 * d, w.att, d, off, off, q.n_seq, q.max_T, 8, d / 8, nullptr));
        return grd_attn_tail((hipStream_t)s, E, w.att, res, q.rows, m->enc_frag[which], out);
    }
    if (!q.info) {
        GTRY(bigc_add_rows_f32(s, x, pos, res, rows, d, q.T));                 // out = x + pos ; res = out
    } else {
        ProfScope ps(s, FAM_GROUNDING, 0.0, 8.0 * q.rows * (double)d);
        hipLaunchKernelGGL(add_pos_ragged_kernel, dim3(grid_for(q.rows * (d / 4))), dim3(256), 0, (hipStream_t)s, x, pos, res, q.rows * (d / 4), d / 4, q.info);
        BIGC_LAUNCH_CHECK("add_pos_ragged");
    }
    // LN -> depth-wise conv -> point-wise GEMM: where the depth-wise stage is a kernel of its own (k = 7, see dwsep), the LayerNorm in
    // front of it is evaluated inside that kernel and the normalised tensor is never written
    static const bool ln_fuse_off = getenv("BIGC_LN_DWCONV") && atoi(getenv("BIGC_LN_DWCONV")) == 0;
    const bool ln_fused = !ln_fuse_off && d == 128 && (E.ks == 3 || E.ks == 7) && !dwsep_fused(d, d, E.ks);
    const Lin* lnw = &E.normb;
    for (int i = 0; i < 4; ++i) {
        if (ln_fused) {
            ProfScope ps(s, FAM_GROUNDING, 0.0, 8.0 * rows * (double)d);
            const int grid = (int)((((long)rows + DW_RUN - 1) / DW_RUN + 7) / 8);
            if (E.ks == 3) hipLaunchKernelGGL(ln_dwconv_rows_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)s, res, lnw->w, lnw->b, 1e-5f, E.conv[i].dw_w, E.conv[i].dw_b, dw, (long)rows, q.T, q.info);
            else hipLaunchKernelGGL(ln_dwconv_rows_kernel<7>, dim3(grid), dim3(256), 0, (hipStream_t)s, res, lnw->w, lnw->b, 1e-5f, E.conv[i].dw_w, E.conv[i].dw_b, dw, (long)rows, q.T, q.info);
            BIGC_LAUNCH_CHECK("ln_dwconv_rows");
            GTRY(glinear(s, dw, d, E.conv[i].pw, nxt, d, rows, d, d, BIGC_ACT_RELU, res, d));    // relu(pw(conv(LN(res)))) + res
        } else {
            GTRY(bigc_layernorm_f32(s, res, lnw->w, lnw->b, ln, rows, d, 1e-5f));
            GTRY(dwsep(s, ln, E.conv[i], dw, nxt, d, d, q, d, E.ks, BIGC_ACT_RELU, res, d));    // relu(conv) + res
        }
        float* t = res; res = nxt; nxt = t;
        lnw = &E.norm_seq[i];
    }
    GTRY(bigc_layernorm_f32(s, res, lnw->w, lnw->b, ln, rows, d, 1e-5f));
    GTRY(glinear(s, ln, d, E.in_proj, w.qkv, 3 * d, rows, 3 * d, d, BIGC_ACT_NONE));
    const int32_t* off = q.off;
    if (!q.info) { GTRY(bigc_fill_offsets((hipStream_t)s, w.off, q.n_seq + 1, q.T)); off = w.off; }
    // (profiling: 4 L^2 d flop per sequence -- sum L^2 >= rows^2 / n_seq, the equal-length value --, q / k / v read and the output written once)
    bigc_prof_hint(4.0 * (double)rows * ((double)rows / q.n_seq) * d, 16.0 * rows * (double)d);
    GTRY(bigc_mha_core_f32(s, w.qkv, 3 * d, w.qkv + d, 3 * d, w.qkv + 2 * d, 3 * d, w.att, d, off, off, q.n_seq, q.max_T, 8, d / 8, nullptr));
    GTRY(glinear(s, w.att, d, E.out_proj, nxt, d, rows, d, d, BIGC_ACT_NONE, res, d));        // attn + res
    res = nxt;
    GTRY(bigc_layernorm_f32(s, res, E.norme.w, E.norme.b, ln, rows, d, 1e-5f));
    GTRY(glinear(s, ln, d, E.fc, out, d, rows, d, d, BIGC_ACT_RELU, res, d));                 // relu(fc) + res
    return BIGC_OK;
}

// 4 x (DWsep k3 + ReLU) + DWsep -> c_out   (models/grd_model_v5.py:182-193)
int run_head(void* s, const Head& H, const float* x, const Seqs& q, int d, const GWork& w, float* out, int act) {
    const float* cur = x;
    float* pp[2] = {w.b0, w.b1};
    for (int i = 0; i < 4; ++i) {
        GTRY(dwsep(s, cur, H.layer[i], w.b2, pp[i & 1], d, d, q, d, 3, BIGC_ACT_RELU));
        cur = pp[i & 1];
    }
    GTRY(dwsep(s, cur, H.layer[4], w.b2, out, H.c_out, H.c_out, q, d, 3, act));
    return BIGC_OK;
}

// a12-a16 for one video (b == nullptr: T clips, n queries) or for several videos at once (b: ragged descriptors)
int forward_impl(grd_model* m, void* stream, const float* video_feat, int T, const int32_t* quint, const int32_t* dura, int n,
                 float video_len, const grd_batch* b, const float* pos_video, const float* pos_query, const float* clip,
                 float score_th, float tiou_th, float bins_th, float nms_th, void* workspace, size_t workspace_bytes, const grd_outputs* out) {
    const grd_config& c = m->cfg;
    const int d = c.dim_hidden, k = c.num_bins;
    const long total_T = b ? b->total_T : T, total_rows = b ? b->total_rows : (long)n * T;
    const int max_T = b ? b->max_T : T, n_vid = b ? b->n_vid : 1;
    BIGC_CHECK_ARG(total_rows < (1L << 31) / (4 * d) * 4, "grd_forward: %ld combined rows do not fit 32-bit row indexing, split the batch", total_rows);
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (!workspace || workspace_bytes < (size_t)(base - (char*)workspace)) return bigc_fail(BIGC_ERR_WORKSPACE, "grd_forward: no workspace");
    GArena a{base, workspace_bytes - (size_t)(base - (char*)workspace)};
    GWork w;
    if (!glayout(m, total_T, n, total_rows, b != nullptr, a, w))
        return bigc_fail(BIGC_ERR_WORKSPACE, "grd_forward: workspace of %zu bytes too small, need %zu", workspace_bytes, a.off + 256);
    void* s = stream;
    hipStream_t hs = (hipStream_t)stream;
    const int32_t* q_vid = b ? b->q_vid : nullptr;
    const float* vid_len = b ? b->vid_len : nullptr;
    Seqs sv{n_vid, T, total_T, max_T, b ? b->vid_row_off : nullptr, nullptr};       // the videos' clip sequences
    Seqs sq{n, 3, (long)3 * n, 3, nullptr, nullptr};                                  // the queries' three words
    Seqs sc{n, T, total_rows, max_T, b ? b->q_row_off : nullptr, nullptr};            // every query against its video's clips
    if (b) {
        hipLaunchKernelGGL(seq_rows_kernel, dim3(n_vid), dim3(256), 0, hs, b->vid_row_off, w.vinfo);
        hipLaunchKernelGGL(seq_rows_kernel, dim3(n), dim3(256), 0, hs, b->q_row_off, w.cinfo);
        BIGC_LAUNCH_CHECK("seq_rows");
        sv.info = w.vinfo; sc.info = w.cinfo;
    }

    // The query side (a12 + the query encoder: 3n rows, ~10 launches that each fill a fraction of the chip) and the video side (video_fc + the
    // video encoder: total_T rows, the same) do not depend on each other until the co-attention: the query side goes to a second stream of
    // the model's, forked from and joined back into the caller's stream by events (BIGC_GRD_STREAMS=0: one stream), with its own scratch.
    static const bool streams_off = getenv("BIGC_GRD_STREAMS") && atoi(getenv("BIGC_GRD_STREAMS")) == 0;
    bool two = !streams_off;
    if (two && !m->side_stream) {
        hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess)
            return bigc_fail(BIGC_ERR_HIP, "grd_forward: could not create the query encoder's stream");
        m->side_stream = st; m->ev_fork = e0; m->ev_join = e1;
    }
    hipStream_t qhs = two ? (hipStream_t)m->side_stream : hs;
    void* qs = (void*)qhs;
    if (two) {
        if (hipEventRecord((hipEvent_t)m->ev_fork, hs) != hipSuccess || hipStreamWaitEvent(qhs, (hipEvent_t)m->ev_fork, 0) != hipSuccess)
            return bigc_fail(BIGC_ERR_HIP, "grd_forward: stream fork failed");
    }
    // a12: prepare_data (:310-328) fused with query_fc + temp_fc (:337-339)
    hipLaunchKernelGGL(query_input_kernel, dim3(grid_for((long)3 * n * m->kq)), dim3(256), 0, qhs, m->enti, m->pred, quint, dura, video_len,
                       q_vid, vid_len, w.qin, n, c.dim_clsme, m->kq);
    BIGC_LAUNCH_CHECK("query_input");
    hipLaunchKernelGGL(span_info_kernel, dim3(cdiv(2 * n, 256)), dim3(256), 0, hs, dura, video_len, q_vid, vid_len, w.info, 2 * n);
    BIGC_LAUNCH_CHECK("span_info");
    Lin qlin{m->wq, m->bq};
    float* qemb = w.comb;                         // (3n, d) scratch, comb is free until the co-attention
    GTRY(glinear(qs, w.qin, m->kq, qlin, qemb, d, 3 * n, d, m->kq, BIGC_ACT_NONE));
    float* vemb = w.comb + (size_t)3 * n * d;     // (total_T, d)
    GTRY(glinear(s, video_feat, c.dim_feat, m->video_fc, vemb, d, (int)total_T, d, c.dim_feat, BIGC_ACT_NONE));

    // a13: the two encoders, co-attention, fusion, combined encoder, three heads (:341-372)
    float* venc = out->video_enc ? out->video_enc : w.venc;
    float* qenc = out->query_enc ? out->query_enc : w.qenc;
    GWork wq = w;
    wq.b0 = w.q_b[0]; wq.b1 = w.q_b[1]; wq.b2 = w.q_b[2]; wq.b3 = w.q_b[3]; wq.qkv = w.q_qkv; wq.att = w.q_att; wq.off = w.q_off;
    const bool fused_ca = grd_coatt_fc_supported(d, max_T);
    GTRY(qanet(qs, m, 1, m->query_enc, qemb, pos_query, sq, d, wq, qenc));
    // vq_fc's K chunks 0 and 1 do not need the (query, clip) rows (csrc/grounding_encoder.hip: grd_coatt_fc): two small GEMMs on the
    // encoders' outputs, each on its encoder's stream
    if (fused_ca) GTRY(bigc_linear_f32(qs, qenc, d, nullptr, 1, d, m->vq_fc.w + d, 4 * d, nullptr, nullptr, 0, w.qu, d, 3 * n, d, d, BIGC_ACT_NONE));
    GTRY(qanet(s, m, 0, m->video_enc, vemb, pos_video, sv, d, w, venc));
    GTRY(glinear(s, venc, d, m->proj2sim, w.proj, d, (int)total_T, d, d, BIGC_ACT_NONE));
    if (fused_ca) GTRY(bigc_linear_f32(s, venc, d, nullptr, 1, d, m->vq_fc.w, 4 * d, m->vq_fc.b, nullptr, 0, w.vp, d, (int)total_T, d, d, BIGC_ACT_NONE));
    if (two) {
        if (hipEventRecord((hipEvent_t)m->ev_join, qhs) != hipSuccess || hipStreamWaitEvent(hs, (hipEvent_t)m->ev_join, 0) != hipSuccess)
            return bigc_fail(BIGC_ERR_HIP, "grd_forward: stream join failed");
    }
    float* cin = w.att;                           // (rows, d) free between qanet calls
    if (fused_ca) {
        // co-attention + vq_fc in one launch: the (rows, 4 d) fusion tensor is never written
        if (!m->vq_frag && hipMalloc(&m->vq_frag, grd_vq_frag_bytes()) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "grd_forward: hipMalloc failed");
        const bool refresh = m->vq_frag_epoch != m->epoch;
        GTRY(grd_coatt_fc(hs, venc, w.proj, qenc, m->vq_fc, w.vp, w.qu, cin, n, T, max_T, total_rows, q_vid, b ? b->vid_row_off : nullptr,
                          b ? b->q_row_off : nullptr, m->vq_frag, refresh));
        m->vq_frag_epoch = m->epoch;
    } else {
        if (!b) {
            GTRY(grd_coattention_f32(s, venc, w.proj, qenc, w.comb, n, T, d));
        } else {
            BIGC_CHECK_ARG(max_T <= 2048 && d <= 256, "grd_forward_batch: max_T=%d (<=2048) d=%d (<=256)", max_T, d);
            const size_t lds = sizeof(float) * (6 * (size_t)d + 6 * (size_t)max_T + 4);
            ProfScope ps(stream, FAM_GROUNDING, 0.0, 16.0 * total_rows * (double)d);
            hipLaunchKernelGGL(coattention_kernel, dim3(n), dim3(256), lds, hs, venc, w.proj, qenc, w.comb, max_T, d, q_vid, b->vid_row_off, b->q_row_off);
            BIGC_LAUNCH_CHECK("coattention");
        }
        GTRY(glinear(s, w.comb, 4 * d, m->vq_fc, cin, d, (int)total_rows, d, 4 * d, BIGC_ACT_NONE));
    }
    float* cenc = out->combined_enc ? out->combined_enc : w.comb;     // comb (rows*4d) is free again
    // qanet reads its input into b0 first, and w.att is only rewritten by the attention step, after
    // the input has been consumed
    GTRY(qanet(s, m, 2, m->comb_enc, cin, pos_video, sc, d, w, cenc));
    float* regrs = out->regrs ? out->regrs : w.regrs;
    float* conf = out->conf ? out->conf : w.conf;
    float* cls = out->cls ? out->cls : w.cls;
    if (grd_head_chain_supported(d, m->regr_head, m->conf_head, m->cls_head)) {
        // the three heads' five layers each in ONE launch, the intermediate activations in LDS (csrc/grounding_heads.hip)
        const Head* const heads[3] = {&m->regr_head, &m->conf_head, &m->cls_head};
        float* const outs[3] = {regrs, conf, cls};
        const int acts[3] = {BIGC_ACT_SIGMOID, BIGC_ACT_NONE, BIGC_ACT_NONE};
        if (!m->head_frag && hipMalloc(&m->head_frag, grd_head_frag_bytes()) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "grd_forward: hipMalloc failed");
        const bool refresh = m->head_frag_epoch != m->epoch;
        GTRY(grd_head_chain(hs, cenc, sc.rows, sc.T, sc.info, heads, outs, acts, m->head_frag, refresh));
        m->head_frag_epoch = m->epoch;
    } else {
        GTRY(run_head(s, m->regr_head, cenc, sc, d, w, regrs, BIGC_ACT_SIGMOID));
        GTRY(run_head(s, m->conf_head, cenc, sc, d, w, conf, BIGC_ACT_NONE));
        GTRY(run_head(s, m->cls_head, cenc, sc, d, w, cls, BIGC_ACT_NONE));
    }

    // a14-a16: pooling, clipping, NMS, masks (:533-572)
    if (!b) {
        GTRY(grd_postprocess(s, regrs, conf, cls, clip, w.info, n, T, k, score_th, tiou_th, bins_th, nms_th, out->pooled_se, out->bins_probs,
                             out->bins_mask, out->status));
    } else {
        hipError_t e = hipMemsetAsync(out->status, 0, sizeof(int32_t) * (size_t)(1 + n_vid), hs);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "grd_forward_batch: memset: %s", hipGetErrorString(e));
        ProfScope ps(stream, FAM_GROUNDING, 0.0, 0.0);
        hipLaunchKernelGGL(postprocess_kernel, dim3(n), dim3(PP_THREADS), 0, hs, regrs, conf, cls, clip, w.info, max_T, k, score_th, tiou_th, bins_th, nms_th,
                           out->pooled_se, out->bins_probs, out->bins_mask, out->status, q_vid, b->vid_row_off, b->q_row_off);
        BIGC_LAUNCH_CHECK("postprocess");
    }
    return BIGC_OK;
}

}  // namespace

extern "C" int grd_create(const grd_config* cfg, grd_model** out) {
    BIGC_CHECK_ARG(cfg && out, "grd_create: null pointer");
    const int hd = cfg->dim_hidden / 8;
    BIGC_CHECK_ARG(cfg->dim_hidden % 8 == 0 && (hd == 8 || hd == 16 || hd == 32 || hd == 64), "grd_create: dim_hidden=%d (8 heads of 8/16/32/64)", cfg->dim_hidden);
    BIGC_CHECK_ARG(cfg->dim_feat % 4 == 0 && cfg->dim_clsme % 4 == 0 && cfg->num_bins >= 1 && cfg->num_bins <= 16, "grd_create: bad dims");
    grd_model* m = new grd_model();
    m->cfg = *cfg;
    m->kq = cfg->dim_clsme + 4;
    *out = m;
    return BIGC_OK;
}

extern "C" void grd_destroy(grd_model* m) {
    if (!m) return;
    for (const auto& kv : m->params) bigc_weight_cache_forget(kv.second.first);      // cached split forms of this model's weights
    if (m->wq) bigc_weight_cache_forget(m->wq);
    if (m->wq) (void)hipFree(m->wq);
    if (m->bq) (void)hipFree(m->bq);
    if (m->head_frag) (void)hipFree(m->head_frag);
    if (m->vq_frag) (void)hipFree(m->vq_frag);
    if (m->ev_fork) (void)hipEventDestroy((hipEvent_t)m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy((hipEvent_t)m->ev_join);
    if (m->side_stream) (void)hipStreamDestroy((hipStream_t)m->side_stream);
    for (int i = 0; i < 3; ++i) if (m->enc_frag[i]) (void)hipFree(m->enc_frag[i]);
    delete m;
}

extern "C" int grd_set_param(grd_model* m, const char* key, const float* ptr, int64_t numel) {
    BIGC_CHECK_ARG(m && key && ptr && numel > 0, "grd_set_param: bad argument");
    BIGC_CHECK_ARG(((uintptr_t)ptr & 15) == 0, "grd_set_param: '%s' is not 16-byte aligned", key);
    { auto it = m->params.find(key); if (it == m->params.end() || it->second.first != ptr) m->train_jobs.clear(); }
    m->params[key] = std::make_pair(ptr, numel);
    m->finalized = false;
    return BIGC_OK;
}

extern "C" int grd_set_mode(grd_model* m, int exact, int fast) {
    BIGC_CHECK_ARG(m && exact >= -1 && exact <= 1 && fast >= -1 && fast <= 2, "grd_set_mode: exact must be -1 (process default), 0 or 1; fast -1, 0, 1 (f16) or 2 (bf16)");
    m->mode_exact = exact; m->mode_fast = fast;
    return BIGC_OK;
}

extern "C" int grd_finalize(grd_model* m, void* stream) {
    BIGC_CHECK_ARG(m, "grd_finalize: null model");
    const grd_config& c = m->cfg;
    const int d = c.dim_hidden, k = c.num_bins;
    GResolver r{m};
    m->enti = r.get("EntiNameEmb", (int64_t)c.n_enti_emb * c.dim_clsme);
    m->pred = r.get("PredNameEmb", (int64_t)c.n_pred_emb * c.dim_clsme);
    m->video_fc = r.lin("video_fc", d, c.dim_feat);
    Lin qfc = r.lin("query_fc", d, c.dim_clsme);
    Lin tfc = r.lin("temp_fc", d, 2);
    m->query_fc = qfc; m->temp_fc = tfc;
    m->vq_fc = r.lin("vq_fc", d, 4 * d);
    m->proj2sim = r.lin("proj2sim", d, d, false);
    m->video_enc = r.enc("video_encoder", d, 7);
    m->query_enc = r.enc("query_encoder", d, 3);
    m->comb_enc = r.enc("combined_encoder", d, 7);
    m->cls_head = r.head("cls_head", d, k);
    m->conf_head = r.head("conf_head", d, k);
    m->regr_head = r.head("regr_head", d, 2 * k);
    if (r.err) return r.err;
    if (!m->wq) {
        if (hipMalloc((void**)&m->wq, sizeof(float) * (size_t)d * m->kq) != hipSuccess || hipMalloc((void**)&m->bq, sizeof(float) * d) != hipSuccess)
            return bigc_fail(BIGC_ERR_HIP, "grd_finalize: hipMalloc failed");
    }
    hipLaunchKernelGGL(build_query_weight_kernel, dim3(cdiv(d * m->kq, 256)), dim3(256), 0, (hipStream_t)stream, qfc.w, qfc.b, tfc.w, tfc.b,
                       m->wq, m->bq, d, c.dim_clsme, m->kq);
    BIGC_LAUNCH_CHECK("build_query_weight");
    m->epoch = bigc_next_weights_epoch();
    m->finalized = true;
    return BIGC_OK;
}

extern "C" size_t grd_workspace_bytes(const grd_model* m, int T, int n) {
    if (!m || T <= 0 || n <= 0) return 0;
    GArena a{nullptr, (size_t)-1};
    GWork w;
    glayout(m, T, n, (long)n * T, false, a, w);
    return a.off + 256;
}

extern "C" size_t grd_batch_workspace_bytes(const grd_model* m, const grd_batch* b) {
    if (!m || !b || b->n_query <= 0 || b->total_T <= 0) return 0;
    GArena a{nullptr, (size_t)-1};
    GWork w;
    glayout(m, b->total_T, b->n_query, b->total_rows, true, a, w);
    return a.off + 256;
}

extern "C" int grd_forward(grd_model* m, void* stream, const float* video_feat, int T, const int32_t* quint, const int32_t* dura, int n,
                           float video_len, const float* pos_video, const float* pos_query, const float* clip,
                           float score_th, float tiou_th, float bins_th, float nms_th,
                           void* workspace, size_t workspace_bytes, const grd_outputs* out) {
    StaticWeights static_weights(m ? m->epoch : 0);
    ModeScope mode_scope(m ? m->mode_exact : -1, m ? m->mode_fast : -1);
    BIGC_CHECK_ARG(m && video_feat && quint && dura && pos_video && pos_query && clip && out, "grd_forward: null pointer");
    BIGC_CHECK_ARG(m->finalized, "grd_forward: call grd_finalize after setting the parameters");
    BIGC_CHECK_ARG(T > 0 && n > 0 && video_len > 0, "grd_forward: T=%d n=%d video_len=%f", T, n, video_len);
    BIGC_CHECK_ARG(out->pooled_se && out->bins_probs && out->bins_mask && out->status, "grd_forward: required outputs missing");
    return forward_impl(m, stream, video_feat, T, quint, dura, n, video_len, nullptr, pos_video, pos_query, clip, score_th, tiou_th, bins_th, nms_th,
                        workspace, workspace_bytes, out);
}

extern "C" int grd_forward_batch(grd_model* m, void* stream, const float* video_feat, const grd_batch* b, const int32_t* quint, const int32_t* dura,
                                 const float* pos_video, const float* pos_query, const float* clip,
                                 float score_th, float tiou_th, float bins_th, float nms_th,
                                 void* workspace, size_t workspace_bytes, const grd_outputs* out) {
    StaticWeights static_weights(m ? m->epoch : 0);
    ModeScope mode_scope(m ? m->mode_exact : -1, m ? m->mode_fast : -1);
    BIGC_CHECK_ARG(m && video_feat && b && quint && dura && pos_video && pos_query && clip && out, "grd_forward_batch: null pointer");
    BIGC_CHECK_ARG(m->finalized, "grd_forward_batch: call grd_finalize after setting the parameters");
    BIGC_CHECK_ARG(b->n_vid > 0 && b->n_query > 0 && b->total_T > 0 && b->total_rows > 0 && b->max_T > 0, "grd_forward_batch: empty batch");
    BIGC_CHECK_ARG(b->vid_row_off && b->q_vid && b->q_row_off && b->vid_len, "grd_forward_batch: batch tables missing");
    BIGC_CHECK_ARG(out->pooled_se && out->bins_probs && out->bins_mask && out->status, "grd_forward_batch: required outputs missing");
    return forward_impl(m, stream, video_feat, b->max_T, quint, dura, b->n_query, 1.0f, b, pos_video, pos_query, clip, score_th, tiou_th, bins_th,
                        nms_th, workspace, workspace_bytes, out);
}
