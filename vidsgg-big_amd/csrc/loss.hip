// Matching cost and loss of the BIG-C training step (models/model_0v10.py:606-704).
//   match_cost   cost[v, q, g] = cc * CE(logits[v,q], gt_pred[g]) + ca * mean_{role,n} BCE(att[v,:,q,n], gt_adj[v,:,g,n])
//                (the Hungarian assignment itself runs on the host with scipy, as in the reference :637)
//   loss_grad    cls_pos / cls_neg / adj loss terms and d logits, d att for the matched queries (:648-703)
// One wave per (video, query); latency-bound integer/transcendental work.
#include "common.h"
#include "prof.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// torch.nn.functional.binary_cross_entropy clamps each log at -100
__device__ __forceinline__ float bce(float a, float t) {
    return -(t * fmaxf(logf(a), -100.f) + (1.f - t) * fmaxf(logf(1.f - a), -100.f));
}

__device__ __forceinline__ float row_lse(const float* lp, int P, int lane) {
    float mx = -INFINITY;
    for (int c = lane; c < P; c += 64) mx = fmaxf(mx, lp[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < P; c += 64) s += expf(lp[c] - mx);
    return mx + logf(wave_sum(s));
}

__global__ __launch_bounds__(64) void match_cost_kernel(const float* __restrict__ logits, int P, const float* __restrict__ att,
                                                        const int32_t* __restrict__ e_off, const int32_t* __restrict__ g_off,
                                                        const int32_t* __restrict__ gt_pred, const float* __restrict__ gt_adj,
                                                        const int64_t* __restrict__ adj_off, int nq, int g_max, float cc, float ca,
                                                        float* __restrict__ cost) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const int v = row / nq, q = row - v * nq;
    const int e0 = e_off[v], ne = e_off[v + 1] - e0;
    const int g0 = g_off[v], ng = g_off[v + 1] - g0;
    if (ne <= 0) return;
    const float* lp = logits + (size_t)row * P;
    const float lse = row_lse(lp, P, lane);
    const float* a0 = att + (size_t)2 * nq * e0 + (size_t)q * ne;             // role 0 row
    const float* a1 = a0 + (size_t)nq * ne;                                   // role 1 row
    const float* adj = gt_adj + adj_off[v];
    for (int g = 0; g < ng; ++g) {
        const float* t0 = adj + (size_t)g * ne;
        const float* t1 = adj + ((size_t)ng + g) * ne;
        float s = 0.f;
        for (int n = lane; n < ne; n += 64) s += bce(a0[n], t0[n]) + bce(a1[n], t1[n]);
        s = wave_sum(s) / (float)(2 * ne);
        if (lane == 0) cost[((size_t)row) * g_max + g] = cc * (lse - lp[gt_pred[g0 + g]]) + ca * s;
    }
}

__global__ __launch_bounds__(64) void loss_grad_kernel(const float* __restrict__ logits, int P, const float* __restrict__ att,
                                                       const int32_t* __restrict__ e_off, const int32_t* __restrict__ g_off,
                                                       const float* __restrict__ gt_adj, const int64_t* __restrict__ adj_off,
                                                       const int32_t* __restrict__ target, const int32_t* __restrict__ match_g, int nq,
                                                       float n_pos, float n_neg, float n_adj, float fc, float fa, float neg_weight,
                                                       float gs_pos, float gs_neg, float gs_adj, float* __restrict__ loss_out, float* __restrict__ dlogits, int ldg,
                                                       float* __restrict__ datt) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const int v = row / nq, q = row - v * nq;
    const int e0 = e_off[v], ne = e_off[v + 1] - e0;
    if (ne <= 0) return;
    const int ng = g_off[v + 1] - g_off[v];
    const float* lp = logits + (size_t)row * P;
    const float lse = row_lse(lp, P, lane);
    const int tgt = target[row];
    const bool pos = tgt != 0;
    const float wcls = fc / (pos ? n_pos : n_neg);
    for (int c = lane; c < P; c += 64) dlogits[(size_t)row * ldg + c] = (expf(lp[c] - lse) - (c == tgt ? 1.f : 0.f)) * wcls * (pos ? gs_pos : gs_neg);
    if (lane == 0) atomicAdd(&loss_out[pos ? 0 : 1], (lse - lp[tgt]) * wcls);
    const int mg = match_g[row];
    const size_t abase = (size_t)2 * nq * e0;
    float s = 0.f;
    for (int r = 0; r < 2; ++r) {
        const size_t arow = abase + ((size_t)r * nq + q) * ne;
        const float* t = gt_adj + adj_off[v] + ((size_t)r * ng + (mg >= 0 ? mg : 0)) * ne;
        for (int n = lane; n < ne; n += 64) {
            float g = 0.f;
            if (mg >= 0) {
                const float a = att[arow + n], tt = t[n];
                const float w = tt != 0.f ? 1.f : neg_weight;
                s += w * bce(a, tt);
                g = w * (a - tt) / fmaxf((1.f - a) * a, 1e-12f) * (fa / n_adj) * gs_adj;
            }
            datt[arow + n] = g;
        }
    }
    s = wave_sum(s);
    if (lane == 0 && mg >= 0) atomicAdd(&loss_out[2], s * fa / n_adj);
}

}  // namespace

extern "C" int bigc_match_cost(void* stream, const float* logits, int n_pcat, const float* att, const int32_t* e_off, const int32_t* g_off,
                               const int32_t* gt_pred, const float* gt_adj, const int64_t* adj_off, int n_vid, int nq, int g_max,
                               float cost_cls, float cost_adj, float* cost) {
    BIGC_CHECK_ARG(logits && att && e_off && g_off && gt_pred && gt_adj && adj_off && cost, "match_cost: null pointer");
    if (n_vid <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_LOSS, 0.0, 0.0);
    hipLaunchKernelGGL(match_cost_kernel, dim3(n_vid * nq), dim3(64), 0, (hipStream_t)stream, logits, n_pcat, att, e_off, g_off, gt_pred,
                       gt_adj, adj_off, nq, g_max, cost_cls, cost_adj, cost);
    BIGC_LAUNCH_CHECK("match_cost");
    return BIGC_OK;
}

extern "C" int bigc_loss_grad(void* stream, const float* logits, int n_pcat, const float* att, const int32_t* e_off, const int32_t* g_off,
                              const float* gt_adj, const int64_t* adj_off, const int32_t* target, const int32_t* match_g, int n_vid, int nq,
                              float n_pos, float n_neg, float n_adj, float loss_cls, float loss_adj, float neg_weight, float gs_pos, float gs_neg, float gs_adj,
                              float* loss_out, float* dlogits, int ldg, float* datt) {
    BIGC_CHECK_ARG(logits && att && e_off && g_off && gt_adj && adj_off && target && match_g && loss_out && dlogits && datt,
                   "loss_grad: null pointer");
    hipError_t e = hipMemsetAsync(loss_out, 0, 3 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "loss_grad: memset: %s", hipGetErrorString(e));
    if (n_vid <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_LOSS, 0.0, 0.0);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(n_vid * nq), dim3(64), 0, (hipStream_t)stream, logits, n_pcat, att, e_off, g_off, gt_adj,
                       adj_off, target, match_g, nq, n_pos, n_neg, n_adj, loss_cls, loss_adj, neg_weight, gs_pos, gs_neg, gs_adj, loss_out, dlogits, ldg, datt);
    BIGC_LAUNCH_CHECK("loss_grad");
    return BIGC_OK;
}

// ---------------------------------------------------------------------------------------------------
// vIoU of every (proposal, GT trajectory) pair of a video (reference: ``vIoU_ts`` utils/utils_func.py:437-471
// called from ``enti_viou_align`` models/model_0v10.py:569-581): intersection volume over the frames both
// trajectories cover, divided by the union of the two whole volumes; boxes are inclusive pixel rectangles (+1).
// One wave per pair; 0 where the durations do not overlap.
namespace {
__global__ __launch_bounds__(64) void viou_matrix_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ row_off,
                                                         const int32_t* __restrict__ dura, const int32_t* __restrict__ e_off,
                                                         const float* __restrict__ gboxes, const int32_t* __restrict__ grow_off,
                                                         const int32_t* __restrict__ gdura, const int32_t* __restrict__ gt_off,
                                                         int max_trk, int g_max, float* __restrict__ out) {
    const int v = blockIdx.z, p = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
    const int e0 = e_off[v], ne = e_off[v + 1] - e0;
    const int g0 = gt_off[v], ng = gt_off[v + 1] - g0;
    if (p >= ne || g >= ng) return;
    const int tp = e0 + p, tg = g0 + g;
    const int ps = dura[2 * tp], pe = dura[2 * tp + 1], gs = gdura[2 * tg], ge = gdura[2 * tg + 1];
    const int s = max(ps, gs), e = min(pe, ge);
    float res = 0.f;
    if (s <= e) {
        const f32x4* bp = (const f32x4*)boxes + row_off[tp];
        const f32x4* bg = (const f32x4*)gboxes + grow_off[tg];
        const int lp = row_off[tp + 1] - row_off[tp], lg = grow_off[tg + 1] - grow_off[tg];
        float ap = 0.f, ag = 0.f, inter = 0.f;
        for (int i = lane; i < lp; i += 64) { const f32x4 b = bp[i]; ap += (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f); }
        for (int i = lane; i < lg; i += 64) { const f32x4 b = bg[i]; ag += (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f); }
        for (int f = s + lane; f <= e; f += 64) {
            const f32x4 a = bp[f - ps], b = bg[f - gs];
            const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]) + 1.f, 0.f);
            const float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]) + 1.f, 0.f);
            inter += w * h;
        }
        ap = wave_sum(ap); ag = wave_sum(ag); inter = wave_sum(inter);
        res = inter / (ap + ag - inter);
    }
    if (lane == 0) out[((size_t)v * max_trk + p) * g_max + g] = res;
}
}  // namespace

// vIoU of listed (trajectory slice, trajectory slice) pairs: the evaluator's primitive (VidVRDhelperEvalAPIs/common.py:65-106,
// called per (prediction, ground-truth relation) with equal triplets from visual_relation_detection.py:10-37, where it is two
// nested Python loops over frames).  A slice = [first box row, first frame, number of frames]; one wave per pair.
namespace {
__global__ __launch_bounds__(64) void viou_pairs_kernel(const float* __restrict__ boxes_a, const int32_t* __restrict__ seg_a,
                                                        const float* __restrict__ boxes_b, const int32_t* __restrict__ seg_b,
                                                        const int32_t* __restrict__ pairs, int n_pairs, float* __restrict__ out) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n_pairs) return;
    const int a = pairs[2 * i], b = pairs[2 * i + 1];
    const int ra = seg_a[3 * a], fa = seg_a[3 * a + 1], la = seg_a[3 * a + 2];
    const int rb = seg_b[3 * b], fb = seg_b[3 * b + 1], lb = seg_b[3 * b + 2];
    const f32x4* pa = (const f32x4*)boxes_a + ra;
    const f32x4* pb = (const f32x4*)boxes_b + rb;
    const int s = max(fa, fb), e = min(fa + la, fb + lb);              // [s, e)
    float res = 0.f;
    if (s < e) {
        float va = 0.f, vb = 0.f, inter = 0.f;
        for (int k = lane; k < la; k += 64) { const f32x4 x = pa[k]; va += (x[2] - x[0] + 1.f) * (x[3] - x[1] + 1.f); }
        for (int k = lane; k < lb; k += 64) { const f32x4 x = pb[k]; vb += (x[2] - x[0] + 1.f) * (x[3] - x[1] + 1.f); }
        for (int f = s + lane; f < e; f += 64) {
            const f32x4 x = pa[f - fa], y = pb[f - fb];
            const float w = fmaxf(fminf(x[2], y[2]) - fmaxf(x[0], y[0]) + 1.f, 0.f);
            const float h = fmaxf(fminf(x[3], y[3]) - fmaxf(x[1], y[1]) + 1.f, 0.f);
            inter += w * h;
        }
        va = wave_sum(va); vb = wave_sum(vb); inter = wave_sum(inter);
        res = inter / (va + vb - inter);
    }
    if (lane == 0) out[i] = res;
}
}  // namespace

extern "C" int bigc_viou_pairs(void* stream, const float* boxes_a, const int32_t* seg_a, const float* boxes_b, const int32_t* seg_b,
                               const int32_t* pairs, int n_pairs, float* out) {
    BIGC_CHECK_ARG(boxes_a && seg_a && boxes_b && seg_b && pairs && out, "viou_pairs: null pointer");
    BIGC_CHECK_ARG((((uintptr_t)boxes_a | (uintptr_t)boxes_b) & 15) == 0, "viou_pairs: boxes must be 16-byte aligned");
    if (n_pairs <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_LOSS, 0.0, 0.0);
    hipLaunchKernelGGL(viou_pairs_kernel, dim3(n_pairs), dim3(64), 0, (hipStream_t)stream, boxes_a, seg_a, boxes_b, seg_b, pairs, n_pairs, out);
    BIGC_LAUNCH_CHECK("viou_pairs");
    return BIGC_OK;
}

namespace {
// F.binary_cross_entropy_with_logits(x, y, reduction='mean') over a (rows, cols) block and its gradient
//   l = max(x, 0) - x y + log1p(exp(-|x|)),  dl/dx = sigmoid(x) - y     (Base_C.loss, models/model_pairwise_baseline.py:301-312)
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy, long rows,
                                                         int cols, float inv_n, float gscale, float* __restrict__ loss,
                                                         float* __restrict__ dx, int lddx) {
    const long total = rows * cols;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const float v = x[(size_t)r * ldx + c], t = y[(size_t)r * ldy + c];
        acc += fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
        if (dx) dx[(size_t)r * lddx + c] = (1.0f / (1.0f + expf(-v)) - t) * inv_n * gscale;
    }
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss) atomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) * inv_n);
}
}  // namespace

extern "C" int bigc_bce_logits_f32(void* stream, const float* x, int ldx, const float* y, int ldy, long rows, int cols, float gscale, float* loss,
                                   float* dx, int lddx) {
    BIGC_CHECK_ARG(x && y && (loss || dx), "bce_logits: null pointer");
    BIGC_CHECK_ARG(ldx >= cols && ldy >= cols && (!dx || lddx >= cols), "bce_logits: leading dimension smaller than cols=%d", cols);
    if (rows <= 0 || cols <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_LOSS, 0.0, 0.0);
    const long total = rows * cols;
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(bce_logits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, cols, 1.0f / (float)total, gscale,
                       loss, dx, lddx);
    BIGC_LAUNCH_CHECK("bce_logits");
    return BIGC_OK;
}

extern "C" int bigc_viou_matrix(void* stream, const float* boxes, const int32_t* trk_row_off, const int32_t* trk_dura, const int32_t* e_off,
                                const float* gt_boxes, const int32_t* gt_row_off, const int32_t* gt_dura, const int32_t* gt_traj_off,
                                int n_vid, int max_trk, int g_max, float* out) {
    BIGC_CHECK_ARG(boxes && trk_row_off && trk_dura && e_off && gt_boxes && gt_row_off && gt_dura && gt_traj_off && out, "viou_matrix: null pointer");
    if (n_vid <= 0 || max_trk <= 0 || g_max <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_LOSS, 0.0, 0.0);
    hipLaunchKernelGGL(viou_matrix_kernel, dim3(g_max, max_trk, n_vid), dim3(64), 0, (hipStream_t)stream, boxes, trk_row_off, trk_dura, e_off,
                       gt_boxes, gt_row_off, gt_dura, gt_traj_off, max_trk, g_max, out);
    BIGC_LAUNCH_CHECK("viou_matrix");
    return BIGC_OK;
}
