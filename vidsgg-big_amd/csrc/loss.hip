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
