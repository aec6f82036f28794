// construct_triplet on the device, one video per workgroup, index-exact with the reference
// (models/model_0v10.py:707-785; ``dura_intersection_ts`` utils/utils_func.py:347-373;
// ``unique_with_idx_nd`` utils/utils_func.py:330-345).
//
//  1. per query: softmax over the predicate logits, top-k (descending, ties -> lowest class id)
//  2. candidate (query, i) links subject/object tracklets soid[0/1][query]; keep it when the two
//     closed duration intervals overlap and subject != object (diagonal cleared, :731)
//  3. unique over [pred, s_cat, o_cat, s_tid, o_tid] in lexicographic order (torch.unique(dim=0)):
//     bitonic sort in LDS of a packed 64-bit key  quintuple(43 bits) | candidate index(11 bits)
//  4. per group the representative is the first candidate (ascending index) with the largest
//     predicate probability (:761 argmax), groups with predicate 0 (background) are dropped (:771)
// Integer/byte work, latency-bound: 256 workgroups of 256 threads, everything in LDS.
#include "common.h"
#include "prof.h"

namespace {

constexpr int SORT_N = 2048;  // >= num_querys * topk
constexpr uint64_t KEY_DEAD = ~0ull;

// NTH threads per video: the per-query softmax / top-k (chains of cross-lane reductions, latency bound) is spread over NTH/64 waves;
// the ordered compaction at the end uses the first 256 threads.
constexpr int NTH = 1024;
__global__ __launch_bounds__(NTH) void construct_triplets_kernel(const float* __restrict__ logits, int P, const int32_t* __restrict__ soid,
                                                                 const int32_t* __restrict__ e_off, const int32_t* __restrict__ trk_dura,
                                                                 const int32_t* __restrict__ trk_cat, const float* __restrict__ trk_score,
                                                                 int nq, int topk, int32_t* __restrict__ quint, float* __restrict__ scores,
                                                                 int32_t* __restrict__ dura, int32_t* __restrict__ qid, int32_t* __restrict__ n_out) {
    __shared__ uint64_t keys[SORT_N];
    __shared__ float cand_p[SORT_N];
    __shared__ int16_t cand_cat[SORT_N];
    __shared__ int rep[SORT_N];      // representative candidate of the group headed at sorted position i, or -1
    __shared__ int scan[256];

    const int vid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e0 = e_off[vid], ne = e_off[vid + 1] - e0;
    const int cap = nq * topk;
    if (ne <= 0) {
        if (tid == 0) n_out[vid] = 0;
        return;
    }

    // ---- 1. softmax + top-k, one wave per query ------------------------------------------------
    // the logits of the wave's next query are requested (unconditionally, from clamped addresses) before the current one is
    // processed: a query is ~400 instructions, far less than a memory round trip
    float nx[4];
    auto fetch = [&](int qi) {
        const float* lp = logits + ((size_t)vid * nq + min(qi, nq - 1)) * P;
#pragma unroll
        for (int j = 0; j < 4; ++j) nx[j] = lp[min(lane + 64 * j, P - 1)];
    };
    fetch(wave);
    for (int qi = wave; qi < nq; qi += NTH / 64) {
        float x[4];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            x[j] = c < P ? nx[j] : -INFINITY;
            mx = fmaxf(mx, x[j]);
        }
        fetch(qi + NTH / 64);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            x[j] = c < P ? expf(x[j] - mx) : 0.f;
            sum += x[j];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            x[j] = c < P ? x[j] / sum : -1.f;   // probabilities are >= 0, -1 marks "absent / already taken"
        }
        for (int i = 0; i < topk; ++i) {
            float bv = -1.f;
            int bi = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = lane + 64 * j;
                if (x[j] > bv || (x[j] == bv && x[j] >= 0.f && c < bi)) { bv = x[j]; bi = c; }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const float ov = __shfl_xor(bv, o);
                const int oi = __shfl_xor(bi, o);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (lane + 64 * j == bi) x[j] = -1.f;
            if (lane == 0) {
                cand_p[qi * topk + i] = bv;
                cand_cat[qi * topk + i] = (int16_t)bi;
            }
        }
    }
    __syncthreads();

    // ---- 2. overlap filter + packed sort keys ------------------------------------------------
    for (int c = tid; c < SORT_N; c += NTH) {
        uint64_t key = KEY_DEAD;
        if (c < cap) {
            const int qi = c / topk;
            const int s = soid[((size_t)vid * 2 + 0) * nq + qi], o = soid[((size_t)vid * 2 + 1) * nq + qi];
            const int ss = trk_dura[2 * (e0 + s)], se = trk_dura[2 * (e0 + s) + 1];
            const int os = trk_dura[2 * (e0 + o)], oe = trk_dura[2 * (e0 + o) + 1];
            if (s != o && max(ss, os) <= min(se, oe)) {
                const uint64_t pc = (uint64_t)cand_cat[c], sc = (uint64_t)trk_cat[e0 + s], oc = (uint64_t)trk_cat[e0 + o];
                key = (pc << 45) | (sc << 37) | (oc << 29) | ((uint64_t)s << 20) | ((uint64_t)o << 11) | (uint64_t)c;
            }
        }
        keys[c] = key;
    }
    __syncthreads();

    // ---- 3. bitonic sort (ascending) ---------------------------------------------------------
    for (int k = 2; k <= SORT_N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < SORT_N; i += NTH) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }

    // ---- 4. group heads, representatives ------------------------------------------------------
    for (int i = tid; i < SORT_N; i += NTH) {
        int r = -1;
        const uint64_t key = keys[i];
        if (key != KEY_DEAD) {
            const uint64_t g = key >> 11;
            const bool head = (i == 0) || ((keys[i - 1] >> 11) != g);
            if (head && (g >> 34) != 0) {   // predicate id 0 = background, dropped
                int best = (int)(key & 2047);
                float bp = cand_p[best];
                for (int m = i + 1; m < SORT_N && (keys[m] >> 11) == g; ++m) {
                    const int c = (int)(keys[m] & 2047);
                    if (cand_p[c] > bp) { bp = cand_p[c]; best = c; }
                }
                r = best;
            }
        }
        rep[i] = r;
    }
    __syncthreads();

    // ---- 5. ordered compaction -------------------------------------------------------------------
    constexpr int PER = SORT_N / 256;
    const bool worker = tid < 256;
    int cnt = 0;
    if (worker) {
#pragma unroll
        for (int u = 0; u < PER; ++u) cnt += rep[tid * PER + u] >= 0;
        scan[tid] = cnt;
    }
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int add = (worker && tid >= o) ? scan[tid - o] : 0;
        __syncthreads();
        if (worker) scan[tid] += add;
        __syncthreads();
    }
    if (!worker) return;
    int pos = scan[tid] - cnt;
    if (tid == 255) n_out[vid] = scan[255];
    const size_t ob = (size_t)vid * cap;
    for (int u = 0; u < PER; ++u) {
        const int i = tid * PER + u;
        const int c = rep[i];
        if (c < 0) continue;
        const uint64_t g = keys[i] >> 11;
        const int o = (int)(g & 511), s = (int)((g >> 9) & 511);
        const int oc = (int)((g >> 18) & 255), sc = (int)((g >> 26) & 255), pc = (int)(g >> 34);
        int32_t* qo = quint + (ob + pos) * 5;
        qo[0] = pc; qo[1] = sc; qo[2] = oc; qo[3] = s; qo[4] = o;
        float* so = scores + (ob + pos) * 3;
        so[0] = cand_p[c]; so[1] = trk_score[e0 + s]; so[2] = trk_score[e0 + o];
        const int ss = trk_dura[2 * (e0 + s)], se = trk_dura[2 * (e0 + s) + 1];
        const int os = trk_dura[2 * (e0 + o)], oe = trk_dura[2 * (e0 + o) + 1];
        dura[(ob + pos) * 2] = max(ss, os);
        dura[(ob + pos) * 2 + 1] = min(se, oe);
        qid[ob + pos] = c / topk;
        ++pos;
    }
}

}  // namespace

extern "C" int bigc_construct_triplets(void* stream, const float* logits, int n_pcat, const int32_t* soid, const int32_t* e_off,
                                       const int32_t* trk_dura, const int32_t* trk_cat, const float* trk_score, int n_vid, int nq,
                                       int topk, int32_t* quint, float* scores, int32_t* dura, int32_t* qid, int32_t* n_out) {
    BIGC_CHECK_ARG(logits && soid && e_off && trk_dura && trk_cat && trk_score && quint && scores && dura && qid && n_out,
                   "construct_triplets: null pointer");
    BIGC_CHECK_ARG(n_pcat > 0 && n_pcat <= 256, "construct_triplets: num_pred_cats=%d must be <= 256", n_pcat);
    BIGC_CHECK_ARG(topk > 0 && topk <= n_pcat && nq * topk <= SORT_N, "construct_triplets: num_querys*topk=%d must be <= %d", nq * topk, SORT_N);
    if (n_vid <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_TRIPLETS, 0.0, 0.0);
    hipLaunchKernelGGL(construct_triplets_kernel, dim3(n_vid), dim3(NTH), 0, (hipStream_t)stream, logits, n_pcat, soid, e_off, trk_dura,
                       trk_cat, trk_score, nq, topk, quint, scores, dura, qid, n_out);
    BIGC_LAUNCH_CHECK("construct_triplets");
    return BIGC_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Padded per-video outputs of construct_triplets -> the reference's return types, back to back (models/model_0v10.py:776-785 returns
// int64 quintuples / durations / query ids and f32 scores per video): video v's n_out[v] rows go to rows [off[v], off[v+1]) of the
// compact arrays, off = exclusive prefix sum of n_out (written to out_off, n_vid + 1 entries).  The host then needs ONE D2H copy
// (the offsets) and one split per array instead of four slices per video.  One workgroup per video; every workgroup sums the
// counts before its own (n_vid is a few hundred).
// ------------------------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void compact_triplets_kernel(const int32_t* __restrict__ quint, const float* __restrict__ scores,
                                                               const int32_t* __restrict__ dura, const int32_t* __restrict__ qid,
                                                               const int32_t* __restrict__ n_out, int n_vid, int cap, int64_t* __restrict__ o_quint,
                                                               float* __restrict__ o_scores, int64_t* __restrict__ o_dura, int64_t* __restrict__ o_qid,
                                                               int32_t* __restrict__ out_off) {
    __shared__ int red[4];
    const int v = blockIdx.x, tid = threadIdx.x;
    int part = 0;
    for (int i = tid; i < v; i += 256) part += n_out[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const int off = red[0] + red[1] + red[2] + red[3];
    const int n = n_out[v];
    if (tid == 0) {
        out_off[v] = off;
        if (v == n_vid - 1) out_off[n_vid] = off + n;
    }
    const size_t ib = (size_t)v * cap;
    for (int i = tid; i < 5 * n; i += 256) o_quint[(size_t)off * 5 + i] = (int64_t)quint[ib * 5 + i];
    for (int i = tid; i < 3 * n; i += 256) o_scores[(size_t)off * 3 + i] = scores[ib * 3 + i];
    for (int i = tid; i < 2 * n; i += 256) o_dura[(size_t)off * 2 + i] = (int64_t)dura[ib * 2 + i];
    for (int i = tid; i < n; i += 256) o_qid[off + i] = (int64_t)qid[ib + i];
}

}  // namespace

extern "C" int bigc_compact_triplets(void* stream, const int32_t* quint, const float* scores, const int32_t* dura, const int32_t* qid,
                                     const int32_t* n_out, int n_vid, int cap, int64_t* out_quint, float* out_scores, int64_t* out_dura,
                                     int64_t* out_qid, int32_t* out_off) {
    BIGC_CHECK_ARG(quint && scores && dura && qid && n_out && out_quint && out_scores && out_dura && out_qid && out_off, "compact_triplets: null pointer");
    BIGC_CHECK_ARG(cap > 0, "compact_triplets: cap=%d", cap);
    if (n_vid <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_TRIPLETS, 0.0, 0.0);
    hipLaunchKernelGGL(compact_triplets_kernel, dim3(n_vid), dim3(256), 0, (hipStream_t)stream, quint, scores, dura, qid, n_out, n_vid, cap, out_quint,
                       out_scores, out_dura, out_qid, out_off);
    BIGC_LAUNCH_CHECK("compact_triplets");
    return BIGC_OK;
}
