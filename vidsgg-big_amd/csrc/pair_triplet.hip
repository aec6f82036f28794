// Base_C.construct_triplet on the device (models/model_pairwise_baseline.py:314-395), all videos of a batch at once.
//
// A video with N tracklets has N (N - 1) ordered pairs and up to N (N - 1) k candidates (pair, predicate) -- 96 660 for a VidOR
// video with 180 tracklets and k = 3 -- far beyond what one workgroup sorts in LDS (triplet.hip: 2048).  But here every
// (pair, predicate) occurs once, so `torch.unique(dim=0)` (:358, utils/utils_func.py:330-345) only ORDERS the rows
// lexicographically by [pred, s_cat, o_cat, s_tid, o_tid], and the position of a pair in the (s_cat, o_cat, s_tid, o_tid) order
// has a closed form.  With the tracklets of a video sorted by (category, id): pos[t] = position of t, before[t] = tracklets of
// smaller category, cnt[t] = tracklets of t's category, the pairs of one (s_cat, o_cat) block are a cnt[s] x cnt[o] grid in
// row-major order and the blocks follow each other in (s_cat, o_cat) order:
//     r(s, o) = before[s] N + cnt[s] before[o] + (pos[s] - before[s]) cnt[o] + (pos[o] - before[o])        in [0, N^2)
// So no sort is needed: one bit per (predicate, r) in a bitmap, a prefix sum of the bit counts, and a candidate's output row is
// rows before its predicate + set bits before its own bit.
//   1. pair_rank_kernel      rank[t] per tracklet                                   (one workgroup per video)
//   2. pair_topk_kernel      softmax + top-k per pair (one wave per pair), temporal-overlap filter (closed intervals,
//                            utils/utils_func.py:347-373), candidate (probability, class) kept, bitmap bit set
//   3. pair_prefix_kernel    per video: rows per predicate (background skipped, :373-376), per-word prefixes, row count
//   4. pair_emit_kernel      quintuple / scores / overlap span of every kept candidate written at its row
//   5. pair_select_kernel    optional rt_triplets_topk (:379-384): the R rows with the largest mean score, descending,
//                            by a 3-pass radix select over the f32 bit patterns + a bitonic sort of the R winners in LDS
// Integer / byte work, HBM- and latency-bound.
#include "common.h"
#include "prof.h"

namespace {

// position of the pair (s, o) of a video with n tracklets in the (s_cat, o_cat, s_tid, o_tid) order (see the file header)
__device__ __forceinline__ int pair_rank(const int32_t* __restrict__ rank, int ts, int to, int n) {
    const int ps = rank[3 * ts], bs = rank[3 * ts + 1], cs = rank[3 * ts + 2];
    const int po = rank[3 * to], bo = rank[3 * to + 1], co = rank[3 * to + 2];
    return bs * n + cs * bo + (ps - bs) * co + (po - bo);
}

__device__ __forceinline__ void pair_of_video(int q, int n, int& s, int& o) {      // row-major (s, o) without the diagonal (:105-112)
    s = q / (n - 1);
    const int r = q - s * (n - 1);
    o = r < s ? r : r + 1;
}

__global__ __launch_bounds__(256) void pair_rank_kernel(const int32_t* __restrict__ vid_trk_off, const int32_t* __restrict__ trk_cat,
                                                        int32_t* __restrict__ rank) {
    const int v = blockIdx.x, t0 = vid_trk_off[v], n = vid_trk_off[v + 1] - t0;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int c = trk_cat[t0 + t];
        int pos = 0, before = 0, cnt = 0;
        for (int u = 0; u < n; ++u) {
            const int cu = trk_cat[t0 + u];
            pos += (cu < c) || (cu == c && u < t);
            before += cu < c;
            cnt += cu == c;
        }
        rank[3 * (t0 + t)] = pos; rank[3 * (t0 + t) + 1] = before; rank[3 * (t0 + t) + 2] = cnt;
    }
}

// one wave per pair: softmax over P <= 256 logits, top-k with ties -> lowest class id (as triplet.hip)
__global__ __launch_bounds__(256) void pair_topk_kernel(const float* __restrict__ logits, int P, int topk, const int32_t* __restrict__ vid_trk_off,
                                                        const int32_t* __restrict__ vid_pair_off, const int32_t* __restrict__ trk_dura,
                                                        const int32_t* __restrict__ rank, int n_vid, long n_pairs, float* __restrict__ cand_p,
                                                        int32_t* __restrict__ cand_c, uint32_t* __restrict__ bitmap, const long* __restrict__ bm_off,
                                                        const int32_t* __restrict__ bm_words) {
    const int lane = threadIdx.x & 63;
    for (long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6); p < n_pairs; p += (long)gridDim.x * 4) {
        int lo = 0, hi = n_vid - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((long)vid_pair_off[mid] <= p) lo = mid; else hi = mid - 1;
        }
        const int v = lo, t0 = vid_trk_off[v], n = vid_trk_off[v + 1] - t0;
        int s, o;
        pair_of_video((int)(p - vid_pair_off[v]), n, s, o);
        const float* lp = logits + (size_t)p * P;
        float x[4];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            x[j] = c < P ? lp[c] : -INFINITY;
            mx = fmaxf(mx, x[j]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float e[4], sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { e[j] = (lane + 64 * j) < P ? expf(x[j] - mx) : 0.f; sum += e[j]; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
        float pr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pr[j] = (lane + 64 * j) < P ? e[j] / sum : -1.f;
        // closed-interval overlap of subject and object
        const int ss = trk_dura[2 * (t0 + s)], se = trk_dura[2 * (t0 + s) + 1], os = trk_dura[2 * (t0 + o)], oe = trk_dura[2 * (t0 + o) + 1];
        const bool keep = min(se, oe) >= max(ss, os);
        const int r = pair_rank(rank, t0 + s, t0 + o, n);
        for (int i = 0; i < topk; ++i) {
            // arg-max over the remaining probabilities, ties -> lowest class id
            float best = pr[0]; int bc = lane;
#pragma unroll
            for (int j = 1; j < 4; ++j) if (pr[j] > best) { best = pr[j]; bc = lane + 64 * j; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ob = __shfl_xor(best, off);
                const int oc = __shfl_xor(bc, off);
                if (ob > best || (ob == best && oc < bc)) { best = ob; bc = oc; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (bc == lane + 64 * j) pr[j] = -1.f;
            if (lane == 0) {
                cand_p[p * topk + i] = best;
                cand_c[p * topk + i] = keep ? bc : -1;
                if (keep) atomicOr(bitmap + bm_off[v] + (size_t)bc * bm_words[v] + (r >> 5), 1u << (r & 31));
            }
        }
    }
}

// per video: exclusive prefix of the set-bit counts, predicate-major (background = class 0 skipped); pref holds, per (class, word),
// the number of output rows before the word's first bit
__global__ __launch_bounds__(1024) void pair_prefix_kernel(const uint32_t* __restrict__ bitmap, const long* __restrict__ bm_off,
                                                           const int32_t* __restrict__ bm_words, int P, int32_t* __restrict__ pref,
                                                           int32_t* __restrict__ n_rows) {
    __shared__ int part[1024];
    __shared__ int carry;
    const int v = blockIdx.x, tid = threadIdx.x, W = bm_words[v];
    const uint32_t* bm = bitmap + bm_off[v];
    int32_t* pf = pref + bm_off[v];
    if (tid == 0) carry = 0;
    __syncthreads();
    const long total = (long)(P - 1) * W;                          // classes 1 .. P-1, in order
    for (long base = 0; base < total; base += 1024) {
        const long i = base + tid;
        const int cnt = i < total ? __popc(bm[W + i]) : 0;         // (class 0 occupies the first W words)
        part[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {                 // inclusive scan
            const int add = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (i < total) pf[W + i] = carry + part[tid] - cnt;
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) n_rows[v] = carry;
}

__global__ __launch_bounds__(256) void pair_emit_kernel(const float* __restrict__ cand_p, const int32_t* __restrict__ cand_c, int topk,
                                                        const int32_t* __restrict__ vid_trk_off, const int32_t* __restrict__ vid_pair_off,
                                                        const int32_t* __restrict__ trk_dura, const int32_t* __restrict__ trk_cat,
                                                        const float* __restrict__ trk_score, const int32_t* __restrict__ rank, int n_vid, long n_pairs,
                                                        const uint32_t* __restrict__ bitmap, const int32_t* __restrict__ pref,
                                                        const long* __restrict__ bm_off, const int32_t* __restrict__ bm_words,
                                                        int32_t* __restrict__ quint, float* __restrict__ scores, int32_t* __restrict__ dura) {
    const long total = n_pairs * topk;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = cand_c[i];
        if (c <= 0) continue;                                      // filtered out, or background
        const long p = i / topk;
        int lo = 0, hi = n_vid - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((long)vid_pair_off[mid] <= p) lo = mid; else hi = mid - 1;
        }
        const int v = lo, t0 = vid_trk_off[v], n = vid_trk_off[v + 1] - t0;
        int s, o;
        pair_of_video((int)(p - vid_pair_off[v]), n, s, o);
        const int r = pair_rank(rank, t0 + s, t0 + o, n);
        const size_t w = bm_off[v] + (size_t)c * bm_words[v] + (r >> 5);
        const long row = (long)vid_pair_off[v] * topk + pref[w] + __popc(bitmap[w] & ((1u << (r & 31)) - 1u));
        int32_t* q = quint + row * 5;
        q[0] = c; q[1] = trk_cat[t0 + s]; q[2] = trk_cat[t0 + o]; q[3] = s; q[4] = o;
        float* sc = scores + row * 3;
        sc[0] = cand_p[i]; sc[1] = trk_score[t0 + s]; sc[2] = trk_score[t0 + o];
        dura[row * 2] = max(trk_dura[2 * (t0 + s)], trk_dura[2 * (t0 + o)]);
        dura[row * 2 + 1] = min(trk_dura[2 * (t0 + s) + 1], trk_dura[2 * (t0 + o) + 1]);
    }
}

// rt_triplets_topk: the R rows with the largest mean score, in descending order (ties: lower row first)
constexpr int SEL_MAX = 2048;
__device__ __forceinline__ uint32_t mean_key(const float* sc) {
    // an unsigned key that orders like the float over ALL floats (tracklet scores may be negative, e.g. log-scores): positive values
    // get their sign bit set, negative ones are complemented; -0.0 is folded into +0.0 first (they compare equal as floats)
    const uint32_t b = __float_as_uint((sc[0] + sc[1] + sc[2]) / 3.0f + 0.0f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ __launch_bounds__(1024) void pair_select_kernel(const int32_t* __restrict__ vid_pair_off, int topk, const int32_t* __restrict__ n_rows,
                                                           int R, const int32_t* __restrict__ quint, const float* __restrict__ scores,
                                                           const int32_t* __restrict__ dura, int32_t* __restrict__ quint_o,
                                                           float* __restrict__ scores_o, int32_t* __restrict__ dura_o, int32_t* __restrict__ n_out) {
    __shared__ int hist[4096];
    __shared__ unsigned long long keys[SEL_MAX];
    __shared__ int sh_bin, sh_above, sh_cnt;
    const int v = blockIdx.x, tid = threadIdx.x, n = n_rows[v];
    const long r0 = (long)vid_pair_off[v] * topk;
    const float* sc = scores + r0 * 3;
    const int take = min(n, R);
    if (tid == 0) n_out[v] = take;
    if (take == 0) return;
    // radix select of the take-th largest key: 12 + 12 + 8 bits
    uint32_t prefix = 0, pmask = 0;
    int above = 0;                                                  // rows already known to be above the threshold
    const int shifts[3] = {20, 8, 0}, bits[3] = {12, 12, 8};
    for (int pass = 0; pass < 3; ++pass) {
        const int nb = 1 << bits[pass];
        for (int i = tid; i < nb; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const uint32_t k = mean_key(sc + (size_t)i * 3);
            if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shifts[pass]) & (nb - 1)], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int acc = above, b = nb - 1;
            for (; b > 0; --b) {
                if (acc + hist[b] >= take) break;
                acc += hist[b];
            }
            sh_bin = b; sh_above = acc;
        }
        __syncthreads();
        prefix |= (uint32_t)sh_bin << shifts[pass];
        pmask |= (uint32_t)(nb - 1) << shifts[pass];
        above = sh_above;
        __syncthreads();
    }
    // winners: every row above the threshold key, and the first (take - above) rows that equal it
    if (tid == 0) sh_cnt = 0;
    for (int i = tid; i < SEL_MAX; i += 1024) keys[i] = 0ull;
    __syncthreads();
    const int need_eq = take - above;
    for (int i = tid; i < n; i += 1024) {
        const uint32_t k = mean_key(sc + (size_t)i * 3);
        if (k > prefix) {
            const int slot = atomicAdd(&sh_cnt, 1);
            keys[slot] = ((unsigned long long)k << 32) | (uint32_t)(0x7fffffff - i);      // descending key, ascending row
        }
    }
    __syncthreads();
    // rows equal to the threshold key are taken in row order: thread t owns a contiguous chunk of rows, an exclusive scan of the
    // per-thread counts gives every such row its rank among them
    {
        const int chunk = (n + 1023) / 1024, b0 = tid * chunk, b1 = min(n, b0 + chunk);
        int cnt = 0;
        for (int i = b0; i < b1; ++i) cnt += mean_key(sc + (size_t)i * 3) == prefix;
        hist[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = tid >= off ? hist[tid - off] : 0;
            __syncthreads();
            hist[tid] += add;
            __syncthreads();
        }
        int ord = hist[tid] - cnt;
        for (int i = b0; i < b1 && ord < need_eq; ++i) {
            if (mean_key(sc + (size_t)i * 3) == prefix) { keys[above + ord] = ((unsigned long long)prefix << 32) | (uint32_t)(0x7fffffff - i); ++ord; }
        }
    }
    __syncthreads();
    // bitonic sort, descending, of SEL_MAX keys (empty slots are 0 = smallest)
    for (int k = 2; k <= SEL_MAX; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < SEL_MAX; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], b = keys[l];
                    const bool up = (i & k) == 0;
                    if (up ? a < b : a > b) { keys[i] = b; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    const long o0 = (long)v * R;
    for (int i = tid; i < take; i += 1024) {
        const int row = 0x7fffffff - (int)(uint32_t)(keys[i] & 0xffffffffu);
        const long src = r0 + row, dst = o0 + i;
#pragma unroll
        for (int c = 0; c < 5; ++c) quint_o[dst * 5 + c] = quint[src * 5 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) scores_o[dst * 3 + c] = scores[src * 3 + c];
        dura_o[dst * 2] = dura[src * 2];
        dura_o[dst * 2 + 1] = dura[src * 2 + 1];
    }
}

}  // namespace

extern "C" size_t bigc_pair_triplets_workspace_bytes(int n_vid, int n_trk, long n_pairs, int topk, long bitmap_words_total) {
    if (n_vid <= 0 || n_pairs <= 0) return 0;
    return 4096 + (size_t)n_pairs * topk * 8 + (size_t)bitmap_words_total * 8 + (size_t)n_trk * 12;
}

// Outputs (all device): quint (n_pairs*topk, 5), scores (n_pairs*topk, 3), dura (n_pairs*topk, 2): video v's rows start at row
// vid_pair_off[v] * topk, its first n_rows[v] rows are valid and sorted.  With rt_topk > 0 additionally quint_top (B, rt_topk, 5),
// scores_top, dura_top, n_top (B).  bm_off (B) / bm_words (B): word offset of video v's bitmap inside the (zeroed here)
// bitmap region and its words per predicate class, ceil(N_v^2 / 32); rank (NT) scratch.
extern "C" int bigc_pair_triplets(void* stream, const float* pair_logits, int n_pred, int topk, const int32_t* vid_trk_off,
                                  const int32_t* vid_pair_off, const int32_t* trk_dura, const int32_t* trk_cat, const float* trk_score,
                                  int n_vid, int n_trk, long n_pairs, const long* bm_off, const int32_t* bm_words, long bitmap_words_total,
                                  void* workspace, size_t workspace_bytes, int32_t* quint, float* scores, int32_t* dura, int32_t* n_rows,
                                  int rt_topk, int32_t* quint_top, float* scores_top, int32_t* dura_top, int32_t* n_top) {
    BIGC_CHECK_ARG(pair_logits && vid_trk_off && vid_pair_off && trk_dura && trk_cat && trk_score && bm_off && bm_words && quint && scores && dura && n_rows,
                   "pair_triplets: null pointer");
    BIGC_CHECK_ARG(n_pred >= 2 && n_pred <= 256 && topk >= 1 && topk <= n_pred, "pair_triplets: n_pred=%d topk=%d", n_pred, topk);
    BIGC_CHECK_ARG(rt_topk <= SEL_MAX && (rt_topk <= 0 || (quint_top && scores_top && dura_top && n_top)), "pair_triplets: rt_topk=%d (<= %d)", rt_topk, SEL_MAX);
    if (n_vid <= 0 || n_pairs <= 0) return BIGC_OK;
    const size_t need = bigc_pair_triplets_workspace_bytes(n_vid, n_trk, n_pairs, topk, bitmap_words_total);
    BIGC_CHECK_ARG(workspace && workspace_bytes >= need, "pair_triplets: workspace of %zu bytes, need %zu", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(stream, FAM_TRIPLETS, 0.0, 4.0 * n_pairs * (double)n_pred);
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    auto take = [&](size_t bytes) { char* p = base; base += (bytes + 255) & ~(size_t)255; return p; };
    float* cand_p = (float*)take((size_t)n_pairs * topk * 4);
    int32_t* cand_c = (int32_t*)take((size_t)n_pairs * topk * 4);
    uint32_t* bitmap = (uint32_t*)take((size_t)bitmap_words_total * 4);
    int32_t* pref = (int32_t*)take((size_t)bitmap_words_total * 4);
    int32_t* rank = (int32_t*)take((size_t)n_trk * 12);
    if (hipMemsetAsync(bitmap, 0, (size_t)bitmap_words_total * 4, s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "pair_triplets: memset failed");
    hipLaunchKernelGGL(pair_rank_kernel, dim3(n_vid), dim3(256), 0, s, vid_trk_off, trk_cat, rank);
    const int g1 = (int)((n_pairs + 3) / 4 < 16384 ? (n_pairs + 3) / 4 : 16384);
    hipLaunchKernelGGL(pair_topk_kernel, dim3(g1), dim3(256), 0, s, pair_logits, n_pred, topk, vid_trk_off, vid_pair_off, trk_dura, rank, n_vid, n_pairs,
                       cand_p, cand_c, bitmap, bm_off, bm_words);
    hipLaunchKernelGGL(pair_prefix_kernel, dim3(n_vid), dim3(1024), 0, s, bitmap, bm_off, bm_words, n_pred, pref, n_rows);
    const long total = n_pairs * topk;
    const int g3 = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pair_emit_kernel, dim3(g3), dim3(256), 0, s, cand_p, cand_c, topk, vid_trk_off, vid_pair_off, trk_dura, trk_cat, trk_score, rank,
                       n_vid, n_pairs, bitmap, pref, bm_off, bm_words, quint, scores, dura);
    if (rt_topk > 0)
        hipLaunchKernelGGL(pair_select_kernel, dim3(n_vid), dim3(1024), 0, s, vid_pair_off, topk, n_rows, rt_topk, quint, scores, dura, quint_top,
                           scores_top, dura_top, n_top);
    BIGC_LAUNCH_CHECK("pair_triplets");
    return BIGC_OK;
}
