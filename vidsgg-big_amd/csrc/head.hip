// Prediction-head gather: subject/object argmax over the role attention, then the concat row
// the predicate classifier consumes and the frequency-bias row it adds
// (reference: models/model_0v10.py:485-503, models/model_0v7.py:485-508).
// One workgroup per (video, query) row; pure gather traffic (HBM/L2-bound).
#include "common.h"
#include "prof.h"

int bigc_scratch_floats(hipStream_t s, size_t need_floats, int slot, float** out);                                                                    // backward.hip
int bigc_ordered_rows_add(hipStream_t s, const float* src, int ld_src, const int32_t* key, int key_stride, int rows, int width, float* dst, int ld_dst);   // backward.hip

namespace {

struct HeadSegs {
    int n;
    const float* src[4];
    int width[4];
    int by_cat[4];
};

// first index of the maximum (torch.argmax on CPU returns the lowest index among equal maxima)
__device__ __forceinline__ int wave_argmax_first(const float* p, int n, int lane) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const float x = p[i];
        if (x > bv || (x == bv && i < bi)) { bv = x; bi = i; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    return bi;
}

__global__ __launch_bounds__(256) void head_gather_kernel(const float* __restrict__ att, const int32_t* __restrict__ e_off,
                                                          const int32_t* __restrict__ trk_cat, int nq,
                                                          const float* __restrict__ query, int dq, HeadSegs segs,
                                                          const float* __restrict__ bias_matrix, int n_ecat, int n_pcat,
                                                          int32_t* __restrict__ soid, float* __restrict__ head_in, int ld_head,
                                                          float* __restrict__ bias_out) {
    __shared__ int sh_so[2];
    const int row = blockIdx.x;
    const int vid = row / nq, qi = row - vid * nq;
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    if (ne <= 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 2) {
        const float* p = att + (size_t)2 * nq * e0 + ((size_t)wave * nq + qi) * ne;
        const int a = wave_argmax_first(p, ne, lane);
        if (lane == 0) {
            sh_so[wave] = a;
            soid[((size_t)vid * 2 + wave) * nq + qi] = a;
        }
    }
    __syncthreads();
    const int ts = e0 + sh_so[0], to = e0 + sh_so[1];
    const int cs = trk_cat[ts], co = trk_cat[to];
    float* hrow = head_in + (size_t)row * ld_head;
    const float* qrow = query + (size_t)row * dq;
    for (int i = threadIdx.x; i < dq; i += 256) hrow[i] = qrow[i];
    int col = dq;
    for (int s = 0; s < segs.n; ++s) {
        const int w = segs.width[s];
        const float* a = segs.src[s] + (size_t)(segs.by_cat[s] ? cs : ts) * w;
        const float* b = segs.src[s] + (size_t)(segs.by_cat[s] ? co : to) * w;
        for (int i = threadIdx.x; i < w; i += 256) {
            hrow[col + i] = a[i];
            hrow[col + w + i] = b[i];
        }
        col += 2 * w;
    }
    const float* bm = bias_matrix + ((size_t)cs * n_ecat + co) * n_pcat;
    for (int i = threadIdx.x; i < n_pcat; i += 256) bias_out[(size_t)row * n_pcat + i] = bm[i];
}

// Factorised head (eval, single-layer classifier): the concat row is never built.  logits = query W_q^T + b + bias_matrix row
// + U[subject] + V[object], where U / V = the per-tracklet products of the subject / object column blocks of the classifier
// weight (computed by the caller with NT rows instead of B*Q).  This kernel: arg-max, then acc = bias row + U[ts] + V[to].
__global__ __launch_bounds__(256) void head_pick_kernel(const float* __restrict__ att, const int32_t* __restrict__ e_off,
                                                        const int32_t* __restrict__ trk_cat, int nq, const float* __restrict__ U,
                                                        const float* __restrict__ V, int lduv, const float* __restrict__ bias_matrix,
                                                        int n_ecat, int n_pcat, int32_t* __restrict__ soid, float* __restrict__ acc) {
    const int sub = threadIdx.x >> 6, lane = threadIdx.x & 63;          // one wave per (video, query) row
    const int row = blockIdx.x * 4 + sub;
    const int vid = row / nq, qi = row - vid * nq;
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    if (ne <= 0) return;
    int so[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        so[r] = wave_argmax_first(att + (size_t)2 * nq * e0 + ((size_t)r * nq + qi) * ne, ne, lane);
        if (lane == 0) soid[((size_t)vid * 2 + r) * nq + qi] = so[r];
    }
    const int ts = e0 + so[0], to = e0 + so[1];
    const float* bm = bias_matrix + ((size_t)trk_cat[ts] * n_ecat + trk_cat[to]) * n_pcat;
    const float* u = U + (size_t)ts * lduv;
    const float* v = V + (size_t)to * lduv;
    for (int i = lane; i < n_pcat; i += 64) acc[(size_t)row * n_pcat + i] = (bm[i] + u[i]) + v[i];
}

}  // namespace

int bigc_head_pick(hipStream_t s, const float* att, const int32_t* e_off, const int32_t* trk_cat, int n_vid, int nq, const float* U,
                   const float* V, int lduv, const float* bias_matrix, int n_ecat, int n_pcat, int32_t* soid, float* acc) {
    ProfScope ps((void*)s, FAM_HEAD_GATHER, 0.0, 12.0 * n_vid * (double)nq * n_pcat);
    hipLaunchKernelGGL(head_pick_kernel, dim3((n_vid * nq + 3) / 4), dim3(256), 0, s, att, e_off, trk_cat, nq, U, V, lduv, bias_matrix, n_ecat,
                       n_pcat, soid, acc);
    BIGC_LAUNCH_CHECK("head_pick");
    return BIGC_OK;
}

extern "C" int bigc_head_gather(void* stream, const float* att, const int32_t* e_off, const int32_t* trk_cat, int n_vid, int nq,
                                const float* query, int dq, int n_seg, const float* const* seg_src_host, const int32_t* seg_width_host,
                                const int32_t* seg_by_cat_host, const float* bias_matrix, int n_ecat, int n_pcat,
                                int32_t* soid, float* head_in, int ld_head, float* bias_out) {
    BIGC_CHECK_ARG(att && e_off && trk_cat && query && bias_matrix && soid && head_in && bias_out, "head_gather: null pointer");
    BIGC_CHECK_ARG(n_seg >= 0 && n_seg <= 4, "head_gather: n_seg=%d not in [0,4]", n_seg);
    if (n_vid <= 0) return BIGC_OK;
    HeadSegs segs;
    segs.n = n_seg;
    int total = dq;
    for (int i = 0; i < 4; ++i) {
        segs.src[i] = i < n_seg ? seg_src_host[i] : nullptr;
        segs.width[i] = i < n_seg ? seg_width_host[i] : 0;
        segs.by_cat[i] = i < n_seg ? seg_by_cat_host[i] : 0;
        total += 2 * segs.width[i];
    }
    BIGC_CHECK_ARG(total <= ld_head, "head_gather: concat width %d exceeds ld_head %d", total, ld_head);
    ProfScope ps(stream, FAM_HEAD_GATHER, 0.0, 8.0 * n_vid * (double)nq * total);
    hipLaunchKernelGGL(head_gather_kernel, dim3(n_vid * nq), dim3(256), 0, (hipStream_t)stream, att, e_off, trk_cat, nq, query, dq,
                       segs, bias_matrix, n_ecat, n_pcat, soid, head_in, ld_head, bias_out);
    BIGC_LAUNCH_CHECK("head_gather");
    return BIGC_OK;
}

// ---- pairwise baseline (models/model_pairwise_baseline.py:105-112,236-273) -------------------------------------------------
namespace {

// pair p of video v: row-major (s, o) over the N_v x N_v grid without its diagonal (trajid2pairid, :105-112)
__device__ __forceinline__ void pair_of(long p, const int32_t* vid_trk_off, const int32_t* vid_pair_off, int n_vid, int& v, int& s, int& o) {
    int lo = 0, hi = n_vid - 1;
    while (lo < hi) {                                   // last video whose first pair is <= p
        const int mid = (lo + hi + 1) >> 1;
        if ((long)vid_pair_off[mid] <= p) lo = mid; else hi = mid - 1;
    }
    v = lo;
    const int n = vid_trk_off[v + 1] - vid_trk_off[v];
    const int q = (int)(p - vid_pair_off[v]);
    s = q / (n - 1);
    const int r = q - s * (n - 1);
    o = r < s ? r : r + 1;
}

__global__ __launch_bounds__(256) void pair_hidden_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                          const int32_t* __restrict__ vid_trk_off, const int32_t* __restrict__ vid_pair_off,
                                                          float* __restrict__ H, int n_vid, long n_pairs, int d4) {
    const long total = n_pairs * d4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / d4;
        const int c = (int)(i - p * d4);
        int v, s, o;
        pair_of(p, vid_trk_off, vid_pair_off, n_vid, v, s, o);
        const int t0 = vid_trk_off[v];
        const f32x4 a = ((const f32x4*)U)[(size_t)(t0 + s) * d4 + c], b = ((const f32x4*)V)[(size_t)(t0 + o) * d4 + c];
        f32x4 h;
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = fmaxf(a[k] + b[k], 0.f);
        ((f32x4*)H)[i] = h;
    }
}

__global__ __launch_bounds__(256) void pair_bias_kernel(const float* __restrict__ bias_matrix, const int32_t* __restrict__ trk_cat,
                                                        const int32_t* __restrict__ vid_trk_off, const int32_t* __restrict__ vid_pair_off,
                                                        float* __restrict__ out, int n_vid, long n_pairs, int n_cat, int n_pred) {
    const long total = n_pairs * n_pred;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / n_pred;
        const int c = (int)(i - p * n_pred);
        int v, s, o;
        pair_of(p, vid_trk_off, vid_pair_off, n_vid, v, s, o);
        const int t0 = vid_trk_off[v];
        out[i] = bias_matrix[((size_t)trk_cat[t0 + s] * n_cat + trk_cat[t0 + o]) * n_pred + c];
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, float* __restrict__ out,
                                                          long total, int d) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / d;
        out[i] = src[(size_t)idx[r] * d + (i - r * d)];
    }
}

// ---- the same with explicit (subject, object) rows per pair: training on the labelled pairs (model_pairwise_baseline.py:276-312)
__global__ __launch_bounds__(256) void pair_hidden_idx_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                              const int32_t* __restrict__ pair_so, float* __restrict__ H, long n_pairs, int d4) {
    const long total = n_pairs * d4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / d4;
        const int c = (int)(i - p * d4);
        const f32x4 a = ((const f32x4*)U)[(size_t)pair_so[2 * p] * d4 + c], b = ((const f32x4*)V)[(size_t)pair_so[2 * p + 1] * d4 + c];
        f32x4 h;
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = fmaxf(a[k] + b[k], 0.f);
        ((f32x4*)H)[i] = h;
    }
}

__global__ __launch_bounds__(256) void pair_bias_idx_kernel(const float* __restrict__ bias_matrix, const int32_t* __restrict__ trk_cat,
                                                            const int32_t* __restrict__ pair_so, float* __restrict__ out, long n_pairs, int n_cat,
                                                            int n_pred) {
    const long total = n_pairs * n_pred;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / n_pred;
        const int c = (int)(i - p * n_pred);
        out[i] = bias_matrix[((size_t)trk_cat[pair_so[2 * p]] * n_cat + trk_cat[pair_so[2 * p + 1]]) * n_pred + c];
    }
}

// destination row of pair p in d bias_matrix: (category of its subject) * n_cat + (category of its object)
__global__ __launch_bounds__(256) void pair_bias_key_kernel(const int32_t* __restrict__ trk_cat, const int32_t* __restrict__ pair_so, long n_pairs, int n_cat,
                                                            int32_t* __restrict__ key) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p < n_pairs) key[p] = trk_cat[pair_so[2 * p]] * n_cat + trk_cat[pair_so[2 * p + 1]];
}

int grid_of(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

}  // namespace

int bigc_pair_hidden(hipStream_t s, const float* U, const float* V, const int32_t* vid_trk_off, const int32_t* vid_pair_off, float* H,
                     int n_vid, long n_pairs, int d) {
    ProfScope ps((void*)s, FAM_HEAD_GATHER, 0.0, 4.0 * n_pairs * (double)d);
    hipLaunchKernelGGL(pair_hidden_kernel, dim3(grid_of(n_pairs * (d / 4))), dim3(256), 0, s, U, V, vid_trk_off, vid_pair_off, H, n_vid, n_pairs, d / 4);
    BIGC_LAUNCH_CHECK("pair_hidden");
    return BIGC_OK;
}

int bigc_pair_bias(hipStream_t s, const float* bias_matrix, const int32_t* trk_cat, const int32_t* vid_trk_off, const int32_t* vid_pair_off,
                   float* out, int n_vid, long n_pairs, int n_cat, int n_pred) {
    ProfScope ps((void*)s, FAM_HEAD_GATHER, 0.0, 8.0 * n_pairs * (double)n_pred);
    hipLaunchKernelGGL(pair_bias_kernel, dim3(grid_of(n_pairs * n_pred)), dim3(256), 0, s, bias_matrix, trk_cat, vid_trk_off, vid_pair_off, out, n_vid,
                       n_pairs, n_cat, n_pred);
    BIGC_LAUNCH_CHECK("pair_bias");
    return BIGC_OK;
}

int bigc_gather_rows(hipStream_t s, const float* src, const int32_t* idx, float* out, int n, int d) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_of((long)n * d)), dim3(256), 0, s, src, idx, out, (long)n * d, d);
    BIGC_LAUNCH_CHECK("gather_rows");
    return BIGC_OK;
}

int bigc_pair_hidden_idx(hipStream_t s, const float* U, const float* V, const int32_t* pair_so, float* H, long n_pairs, int d) {
    hipLaunchKernelGGL(pair_hidden_idx_kernel, dim3(grid_of(n_pairs * (d / 4))), dim3(256), 0, s, U, V, pair_so, H, n_pairs, d / 4);
    BIGC_LAUNCH_CHECK("pair_hidden_idx");
    return BIGC_OK;
}

int bigc_pair_bias_idx(hipStream_t s, const float* bias_matrix, const int32_t* trk_cat, const int32_t* pair_so, float* out, long n_pairs,
                       int n_cat, int n_pred) {
    hipLaunchKernelGGL(pair_bias_idx_kernel, dim3(grid_of(n_pairs * n_pred)), dim3(256), 0, s, bias_matrix, trk_cat, pair_so, out, n_pairs, n_cat, n_pred);
    BIGC_LAUNCH_CHECK("pair_bias_idx");
    return BIGC_OK;
}

// dU[s_p] += dH[p], dV[o_p] += dH[p] and d bias_matrix[cat_s, cat_o] += d logits[p]: scatter-adds with collisions, added in PAIR ORDER per
// destination (bigc_ordered_rows_add, backward.hip) -- no float atomics, the training step's gradients are bit-reproducible
int bigc_pair_scatter_add(hipStream_t s, const float* dH, const int32_t* pair_so, float* dU, float* dV, long n_pairs, int d) {
    if (int rc = bigc_ordered_rows_add(s, dH, d, pair_so, 2, (int)n_pairs, d, dU, d)) return rc;
    return bigc_ordered_rows_add(s, dH, d, pair_so + 1, 2, (int)n_pairs, d, dV, d);
}

int bigc_pair_bias_bwd(hipStream_t s, const float* dlogits, int ldg, const int32_t* trk_cat, const int32_t* pair_so, float* dbias, long n_pairs,
                       int n_cat, int n_pred) {
    if (n_pairs <= 0) return BIGC_OK;
    float* key = nullptr;
    if (int rc = bigc_scratch_floats(s, (size_t)n_pairs, 4, &key)) return rc;
    hipLaunchKernelGGL(pair_bias_key_kernel, dim3((int)((n_pairs + 255) / 256)), dim3(256), 0, s, trk_cat, pair_so, n_pairs, n_cat, (int32_t*)key);
    BIGC_LAUNCH_CHECK("pair_bias_key");
    return bigc_ordered_rows_add(s, dlogits, ldg, (const int32_t*)key, 1, (int)n_pairs, n_pred, dbias, n_pred);
}
