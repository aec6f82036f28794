// Matching cost and loss of the BIG-C training step (models/model_0v10.py:606-704).
//   match_cost   cost[v, q, g] = cc * CE(logits[v,q], gt_pred[g]) + ca * mean_{role,n} BCE(att[v,:,q,n], gt_adj[v,:,g,n])
//                (the Hungarian assignment itself runs on the host with scipy, as in the reference :637)
//   loss_grad    cls_pos / cls_neg / adj loss terms and d logits, d att for the matched queries (:648-703)
// One wave per (video, query); latency-bound integer/transcendental work.
#include "common.h"
#include "prof.h"
#include <algorithm>
#include <map>
#include <unordered_map>
#include <utility>

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

// One wave per (video, query) row, LG_ROWS rows per workgroup of four waves: the three loss terms are summed in registers and LDS first
// and reach loss_out as ONE atomic per term and workgroup (before: two atomics per row, 24 576 of them on three addresses -- the L2
// serialises those; 168 us per launch at 64 videos for a few MB of traffic).
constexpr int LG_ROWS = 16;
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ logits, int P, const float* __restrict__ att,
                                                        const int32_t* __restrict__ e_off, const int32_t* __restrict__ g_off,
                                                        const float* __restrict__ gt_adj, const int64_t* __restrict__ adj_off,
                                                        const int32_t* __restrict__ target, const int32_t* __restrict__ match_g, int nq, int n_rows,
                                                        float n_pos, float n_neg, float n_adj, float fc, float fa, float neg_weight,
                                                        float gs_pos, float gs_neg, float gs_adj, const float* __restrict__ gs_dev,
                                                        float* __restrict__ loss_partial, float* __restrict__ dlogits, int ldg,
                                                        float* __restrict__ datt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (gs_dev) { gs_pos = gs_dev[0]; gs_neg = gs_dev[1]; gs_adj = gs_dev[2]; }
    float part[3] = {0.f, 0.f, 0.f};                      // this wave's share of {cls_pos, cls_neg, adj} (lane 0)
    for (int rr = wave; rr < LG_ROWS; rr += 4) {
        const int row = blockIdx.x * LG_ROWS + rr;
        if (row >= n_rows) break;
        const int v = row / nq, q = row - v * nq;
        const int e0 = e_off[v], ne = e_off[v + 1] - e0;
        if (ne <= 0) continue;
        const int ng = g_off[v + 1] - g_off[v];
        const float* lp = logits + (size_t)row * P;
        const float lse = row_lse(lp, P, lane);
        const int tgt = target[row];
        const bool pos = tgt != 0;
        const float wcls = fc / (pos ? n_pos : n_neg);
        if (dlogits)
            for (int c = lane; c < P; c += 64) dlogits[(size_t)row * ldg + c] = (expf(lp[c] - lse) - (c == tgt ? 1.f : 0.f)) * wcls * (pos ? gs_pos : gs_neg);
        const float ce = (lse - lp[tgt]) * wcls;
        part[0] += pos ? ce : 0.f;
        part[1] += pos ? 0.f : ce;
        const int mg = match_g[row];
        if (!datt && mg < 0) continue;                        // loss terms only: an unmatched query adds nothing to the adjacency term
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
                if (datt) datt[arow + n] = g;
            }
        }
        s = wave_sum(s);
        if (mg >= 0) part[2] += s * fa / n_adj;
    }
    __shared__ float red[4][3];
    if (lane == 0) { red[wave][0] = part[0]; red[wave][1] = part[1]; red[wave][2] = part[2]; }
    __syncthreads();
    // one partial per term and workgroup; loss_finish_kernel adds them in a fixed order (float atomics here made the low bits
    // of the reported loss terms vary from run to run)
    if (threadIdx.x < 3) loss_partial[(size_t)blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// loss_out[t] = sum over the workgroups' partials of term t: lane l adds the partials l, l + 64, ... in order, then the 64 lane sums are
// folded in a fixed butterfly -- the same order in every run
__global__ __launch_bounds__(64) void loss_finish_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ loss_out) {
    const int lane = threadIdx.x;
    for (int t = 0; t < 3; ++t) {
        float s = 0.f;
        for (int b = lane; b < blocks; b += 64) s += partial[(size_t)b * 3 + t];
        s = wave_sum(s);
        if (lane == 0) loss_out[t] = s;
    }
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

static int loss_grad_launch(void* stream, const float* logits, int n_pcat, const float* att, const int32_t* e_off, const int32_t* g_off,
                            const float* gt_adj, const int64_t* adj_off, const int32_t* target, const int32_t* match_g, int n_vid, int nq,
                            float n_pos, float n_neg, float n_adj, float loss_cls, float loss_adj, float neg_weight, float gs_pos, float gs_neg, float gs_adj,
                            const float* gs_dev, float* loss_out, float* dlogits, int ldg, float* datt) {
    BIGC_CHECK_ARG(logits && att && e_off && g_off && gt_adj && adj_off && target && match_g && loss_out, "loss_grad: null pointer");
    BIGC_CHECK_ARG((dlogits != nullptr) == (datt != nullptr), "loss_grad: dlogits and datt are given together (both NULL: the loss terms only)");
    hipError_t e = hipMemsetAsync(loss_out, 0, 3 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "loss_grad: memset: %s", hipGetErrorString(e));
    if (n_vid <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_LOSS, 0.0, 0.0);
    const int blocks = (n_vid * nq + LG_ROWS - 1) / LG_ROWS;
    // per-workgroup partial terms: a stream-ordered scratch per (calling thread, device, stream), grown on demand
    struct Buf { float* p = nullptr; size_t cap = 0; };
    thread_local std::map<std::pair<int, void*>, Buf> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "loss_grad: hipGetDevice failed");
    Buf& b = bufs[std::make_pair(dev, stream)];
    if ((size_t)blocks * 3 > b.cap) {
        if (b.p) { if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "loss_grad: sync failed"); (void)hipFree(b.p); }
        b.cap = (size_t)blocks * 6;
        if (hipMalloc((void**)&b.p, b.cap * sizeof(float)) != hipSuccess) { b.p = nullptr; b.cap = 0; return bigc_fail(BIGC_ERR_HIP, "loss_grad: hipMalloc of the partial terms failed"); }
    }
    hipLaunchKernelGGL(loss_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, n_pcat, att, e_off, g_off, gt_adj,
                       adj_off, target, match_g, nq, n_vid * nq, n_pos, n_neg, n_adj, loss_cls, loss_adj, neg_weight, gs_pos, gs_neg, gs_adj, gs_dev, b.p, dlogits, ldg, datt);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)b.p, blocks, loss_out);
    BIGC_LAUNCH_CHECK("loss_grad");
    return BIGC_OK;
}

extern "C" int bigc_loss_grad(void* stream, const float* logits, int n_pcat, const float* att, const int32_t* e_off, const int32_t* g_off,
                              const float* gt_adj, const int64_t* adj_off, const int32_t* target, const int32_t* match_g, int n_vid, int nq,
                              float n_pos, float n_neg, float n_adj, float loss_cls, float loss_adj, float neg_weight, float gs_pos, float gs_neg, float gs_adj,
                              float* loss_out, float* dlogits, int ldg, float* datt) {
    return loss_grad_launch(stream, logits, n_pcat, att, e_off, g_off, gt_adj, adj_off, target, match_g, n_vid, nq, n_pos, n_neg, n_adj, loss_cls, loss_adj,
                            neg_weight, gs_pos, gs_neg, gs_adj, nullptr, loss_out, dlogits, ldg, datt);
}

// the same with the three upstream gradients (d loss / d cls_pos, cls_neg, adj) read from device memory: the backward of a step that
// never waits for the GPU (bigc_hip/train_step.py)
extern "C" int bigc_loss_grad_dev(void* stream, const float* logits, int n_pcat, const float* att, const int32_t* e_off, const int32_t* g_off,
                                  const float* gt_adj, const int64_t* adj_off, const int32_t* target, const int32_t* match_g, int n_vid, int nq,
                                  float n_pos, float n_neg, float n_adj, float loss_cls, float loss_adj, float neg_weight, const float* upstream3,
                                  float* loss_out, float* dlogits, int ldg, float* datt) {
    BIGC_CHECK_ARG(upstream3, "loss_grad_dev: null pointer");
    return loss_grad_launch(stream, logits, n_pcat, att, e_off, g_off, gt_adj, adj_off, target, match_g, n_vid, nq, n_pos, n_neg, n_adj, loss_cls, loss_adj,
                            neg_weight, 1.f, 1.f, 1.f, upstream3, loss_out, dlogits, ldg, datt);
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
                                                         int cols, float inv_n, float gscale, float* __restrict__ part,
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
    __shared__ float wsum4[4];
    if ((threadIdx.x & 63) == 0) wsum4[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && part) part[blockIdx.x] = (wsum4[0] + wsum4[1] + wsum4[2] + wsum4[3]) * inv_n;    // one partial per workgroup
}

// loss += part[0] + part[1] + ... in a fixed order (one wave: lane l adds l, l + 64, ...; then the lanes, butterfly): no float atomics
__global__ __launch_bounds__(64) void bce_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) *loss += s;
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
    float* part = nullptr;
    if (loss) {                             // per-workgroup partial sums: stream-ordered scratch per (calling thread, device, stream)
        struct Buf { float* p = nullptr; };
        thread_local std::map<std::pair<int, void*>, Buf> bufs;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "bce_logits: hipGetDevice failed");
        Buf& b = bufs[std::make_pair(dev, stream)];
        if (!b.p && hipMalloc((void**)&b.p, 1024 * sizeof(float)) != hipSuccess) { b.p = nullptr; return bigc_fail(BIGC_ERR_HIP, "bce_logits: hipMalloc of the partial sums failed"); }
        part = b.p;
    }
    hipLaunchKernelGGL(bce_logits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, cols, 1.0f / (float)total, gscale,
                       part, dx, lddx);
    if (loss) hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)part, grid, loss);
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

// ---------------------------------------------------------------------------------------------------
// The host half of the reference's matching, on the device (bigc_hip/train_step.py: a training step without a host round trip).
//
// align_adjacency: the tail of ``enti_viou_align`` (models/model_0v10.py:583-602).  From the vIoU table vi (n proposals, nt GT
// trajectories) of a video: mask = vi > th; a GT trajectory no proposal passes for takes its best proposal; has[p] = any(mask[p, :]);
// pick[p] = argmax_g vi[p, g]; out[r, g, p] = adj[r, g, pick[p]] * has[p].  torch.argmax returns the first maximum.  One workgroup per video.
namespace {
__global__ __launch_bounds__(256) void align_adjacency_kernel(const float* __restrict__ viou, int max_trk, int t_max, const int32_t* __restrict__ e_off,
                                                              const int32_t* __restrict__ gt_traj_off, const float* __restrict__ adj_raw,
                                                              const int64_t* __restrict__ raw_off, const int32_t* __restrict__ g_off,
                                                              const int64_t* __restrict__ adj_off, float th, float* __restrict__ out) {
    extern __shared__ int align_sm[];
    const int v = blockIdx.x, tid = threadIdx.x;
    const int ne = e_off[v + 1] - e_off[v], nt = gt_traj_off[v + 1] - gt_traj_off[v], ng = g_off[v + 1] - g_off[v];
    if (ne <= 0 || nt <= 0) return;
    int* pick = align_sm;
    int* has = align_sm + ne;
    const float* vi = viou + (size_t)v * max_trk * t_max;
    for (int p = tid; p < ne; p += 256) {
        const float* r = vi + (size_t)p * t_max;
        int best = 0, any = 0;
        float bv = r[0];
        for (int g = 0; g < nt; ++g) {
            const float x = r[g];
            any |= x > th;
            if (x > bv) { bv = x; best = g; }
        }
        pick[p] = best;
        has[p] = any;
    }
    __syncthreads();
    for (int g = tid; g < nt; g += 256) {
        int best = 0, cnt = 0;
        float bv = vi[g];
        for (int p = 0; p < ne; ++p) {
            const float x = vi[(size_t)p * t_max + g];
            cnt += x > th;
            if (x > bv) { bv = x; best = p; }
        }
        if (cnt == 0) has[best] = 1;                     // (several trajectories may write the same 1)
    }
    __syncthreads();
    const float* adj = adj_raw + raw_off[v];
    float* o = out + adj_off[v];
    const long total = (long)2 * ng * ne;
    for (long i = tid; i < total; i += 256) {
        const long rg = i / ne;
        const int p = (int)(i - rg * ne);
        o[i] = adj[rg * nt + pick[p]] * (has[p] ? 1.f : 0.f);
    }
}

// hungarian: scipy.optimize.linear_sum_assignment (the reference's call, models/model_0v10.py:637) on cost[v, :, :n_gt[v]] of every
// video.  scipy 1.x solves the rectangular problem with the shortest-augmenting-path algorithm of Crouse, "On implementing 2D
// rectangular assignment algorithms" (IEEE T-AES 2016), in float64, transposing so that rows are the smaller side: here rows = the
// n_gt < nq GT predicates, columns = the queries.  This kernel restates that algorithm step for step, including the order in which
// equal candidates are taken (scipy scans the not-yet-visited columns in the order of its `remaining` list -- initially nc-1 .. 0,
// the last entry moved into a visited column's slot -- and among equal lowest reduced costs keeps the first one met, except that a
// later UNASSIGNED column replaces it), so the assignment is scipy's also where the optimum is not unique (duplicated GT predicates).
// The additions are done in the same order in IEEE float64.  One wave per video; every lane holds the scalars of the scan, the
// column scan and the arg-min reduction are spread over the lanes.
// bad[0] |= 1: a cost is NaN / -inf (scipy: ValueError "matrix contains invalid numeric entries") or the problem is infeasible; the
// video is then left unmatched (target 0, match -1).
struct Cand { double val; int un; int it; };
__device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b) {       // a before b in scipy's scan result
    if (a.val != b.val) return a.val < b.val;
    if (a.un != b.un) return a.un > b.un;
    return a.un ? a.it > b.it : a.it < b.it;
}

__global__ __launch_bounds__(64) void hungarian_kernel(const float* __restrict__ cost, const int32_t* __restrict__ g_off,
                                                       const int32_t* __restrict__ gt_pred, int nq, int g_max, int stage_cost,
                                                       int32_t* __restrict__ target, int32_t* __restrict__ match_g, int* __restrict__ bad) {
    extern __shared__ double hung_sm[];
    const int v = blockIdx.x, lane = threadIdx.x;
    const int g0 = g_off[v], nr = g_off[v + 1] - g0, nc = nq;
    double* u = hung_sm;                        // nq each (nr < nq)
    double* vv = u + nq;
    double* sp = vv + nq;
    int* path = (int*)(sp + nq);
    int* row4col = path + nq;
    int* col4row = row4col + nq;
    int* remaining = col4row + nq;
    int* SR = remaining + nq;
    int* SC = SR + nq;
    float* cl = (float*)(SC + nq);              // nr x nc, when staged
    const float* cv = cost + (size_t)v * nq * g_max;
    int invalid = 0;
    for (int j = lane; j < nc; j += 64) {
        vv[j] = 0.0; row4col[j] = -1;
        for (int i = 0; i < nr; ++i) {
            const float x = cv[(size_t)j * g_max + i];
            invalid |= (x != x) || (x == -INFINITY);
            if (stage_cost) cl[i * nc + j] = x;
        }
    }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    invalid = __any(invalid);
    __syncthreads();
    for (int cur = 0; cur < nr && !invalid; ++cur) {
        double minVal = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        for (int it = lane; it < nc; it += 64) { remaining[it] = nc - it - 1; SC[it] = 0; sp[it] = INFINITY; }
        for (int r = lane; r < nr; r += 64) SR[r] = 0;
        __syncthreads();
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            Cand best{INFINITY, 0, 0x7fffffff};
            bool have = false;
            for (int it = lane; it < num_remaining; it += 64) {
                const int j = remaining[it];
                const double c = (double)(stage_cost ? cl[i * nc + j] : cv[(size_t)j * g_max + i]);
                const double r = minVal + c - ui - vv[j];
                double s = sp[j];
                if (r < s) { path[j] = i; sp[j] = r; s = r; }
                const Cand x{s, row4col[j] == -1, it};
                // scipy's scan starts from lowest = inf, index = -1: a candidate at +inf is only taken when its column is unassigned
                if (!have) { if (s < INFINITY || x.un) { best = x; have = true; } }
                else if (cand_better(x, best)) best = x;
            }
            // reduction over the lanes (lanes without a candidate carry val = inf, un = 0, it = int max: they lose against every real one,
            // and an all-inf assigned scan leaves index -1 = infeasible below)
            if (!have) best = Cand{INFINITY, -1, 0x7fffffff};
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                Cand y;
                y.val = __shfl_xor(best.val, o); y.un = __shfl_xor(best.un, o); y.it = __shfl_xor(best.it, o);
                if (y.un >= 0 && (best.un < 0 || cand_better(y, best))) best = y;
            }
            if (best.un < 0 || !(best.val < INFINITY)) { invalid = 1; break; }
            minVal = best.val;
            const int j = remaining[best.it];
            const int r4 = row4col[j];
            __syncthreads();                                 // every lane has read remaining[] / row4col[] before lane 0 rewrites them
            if (r4 == -1) sink = j; else i = r4;
            --num_remaining;
            if (lane == 0) { SC[j] = 1; remaining[best.it] = remaining[num_remaining]; }
            __syncthreads();
        }
        if (invalid) break;
        // dual update, then the augmentation along the path
        if (lane == 0) u[cur] += minVal;
        for (int r = lane; r < nr; r += 64) if (SR[r] && r != cur) u[r] += minVal - sp[col4row[r]];
        for (int j = lane; j < nc; j += 64) if (SC[j]) vv[j] -= minVal - sp[j];
        __syncthreads();
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int r = path[j];
                row4col[j] = r;
                const int t = col4row[r]; col4row[r] = j; j = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    if (invalid && lane == 0) atomicOr(bad, 1);
    for (int j = lane; j < nc; j += 64) {
        const int r = invalid ? -1 : row4col[j];
        match_g[(size_t)v * nq + j] = r;
        target[(size_t)v * nq + j] = r >= 0 ? gt_pred[g0 + r] : 0;
    }
}
}  // namespace

extern "C" int bigc_align_adjacency(void* stream, const float* viou, int max_trk, int t_max, const int32_t* e_off, const int32_t* gt_traj_off,
                                    const float* adj_raw, const int64_t* raw_off, const int32_t* g_off, const int64_t* adj_off, int n_vid,
                                    float viou_th, float* gt_adj) {
    BIGC_CHECK_ARG(viou && e_off && gt_traj_off && adj_raw && raw_off && g_off && adj_off && gt_adj, "align_adjacency: null pointer");
    BIGC_CHECK_ARG(max_trk > 0 && max_trk <= 8192, "align_adjacency: max_trk=%d outside (0, 8192]", max_trk);
    if (n_vid <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_LOSS, 0.0, 0.0);
    hipLaunchKernelGGL(align_adjacency_kernel, dim3(n_vid), dim3(256), 2 * max_trk * sizeof(int), (hipStream_t)stream, viou, max_trk, t_max, e_off,
                       gt_traj_off, adj_raw, raw_off, g_off, adj_off, viou_th, gt_adj);
    BIGC_LAUNCH_CHECK("align_adjacency");
    return BIGC_OK;
}

extern "C" int bigc_hungarian(void* stream, const float* cost, const int32_t* g_off, const int32_t* gt_pred, int n_vid, int nq, int g_max,
                              int32_t* target, int32_t* match_g, int* bad) {
    BIGC_CHECK_ARG(cost && g_off && gt_pred && target && match_g && bad, "hungarian: null pointer");
    BIGC_CHECK_ARG(nq > 0 && nq <= 1024, "hungarian: nq=%d outside (0, 1024]", nq);
    BIGC_CHECK_ARG(g_max > 0 && g_max < nq, "hungarian: g_max=%d must be in (0, nq=%d): every video needs fewer GT predicates than queries", g_max, nq);
    if (n_vid <= 0) return BIGC_OK;
    const size_t base = (size_t)nq * (3 * sizeof(double) + 6 * sizeof(int));
    const size_t staged = (size_t)nq * g_max * sizeof(float);
    const int stage = base + staged <= 60 * 1024;
    ProfScope ps(stream, FAM_LOSS, 0.0, 0.0);
    hipLaunchKernelGGL(hungarian_kernel, dim3(n_vid), dim3(64), base + (stage ? staged : 0), (hipStream_t)stream, cost, g_off, gt_pred, nq, g_max, stage,
                       target, match_g, bad);
    BIGC_LAUNCH_CHECK("hungarian");
    return BIGC_OK;
}
