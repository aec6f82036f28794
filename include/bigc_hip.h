/*
 * bigc_hip.h -- C ABI of the MI355X-native BIG classification-then-grounding hot path.
 *
 * The reference (Dawn-LX/VidSGG-BIG) is pure Python/PyTorch and has no FFI; the drop-in
 * boundary is therefore the nn.Module forward() of
 *     models/model_0v10.py:359  BIG_C.forward(proposal_list, gt_graph_list=None, topk=10)
 *     models/model_0v7.py:362   BIG_C.forward(proposal_list, gt_graph_list=None, topk=3)
 *     models/grd_model_v5.py:196 DEBUG.forward(video_feature_list, data_list, ...)
 * and this header is what those forward()s bind (through ctypes).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the caller owns all buffers (allocated with torch), the library never frees them;
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered, nothing
 *     synchronises the device;
 *   - every function returns 0 on success or a negative bigc_status; the message of the
 *     last failure on the calling thread is returned by bigc_last_error();
 *   - one handle per device and per model; a handle may be used from one thread at a time.
 *   - all tensors are IEEE fp32 and every accumulation is fp32; the GEMMs, the head_dim-64 attention core and the fused
 *     role attention run on the f16 matrix cores with exactly split f32 operands (f32-level error; operands with
 *     |x| >= 32768 are redone on the f32 MFMA; bigc_linear_set_exact(1) selects the f32 MFMA kernels throughout), the
 *     other attention kernels on the f32 MFMA; indices are int32 on the device side (the Python host converts to the
 *     reference's int64 at the boundary).
 */
#ifndef BIGC_HIP_H
#define BIGC_HIP_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: the functions marked BIGC_API below are its ONLY dynamic symbols
 * (tests/test_host_cpu.py compares `nm -D --defined-only` with this header's list, both ways). */
#define BIGC_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    BIGC_OK = 0,
    BIGC_ERR_ARG = -1,       /* bad argument / shape violation (the reference raises AssertionError) */
    BIGC_ERR_MISSING = -2,   /* a parameter named by the model was never set */
    BIGC_ERR_HIP = -3,       /* a HIP runtime call failed */
    BIGC_ERR_WORKSPACE = -4  /* workspace too small */
} bigc_status;

BIGC_API const char* bigc_last_error(void);
/* "bigc_hip gfx950 ... <build id>" -- lets the host verify it loaded the HIP build */
BIGC_API const char* bigc_build_info(void);

/* Optional per-kernel-family timing with HIP events recorded on the caller's stream around every
 * launch (bench.py's roofline figure). Off by default; bigc_prof_collect() must be called after the
 * stream has been synchronised, it folds the recorded pairs into per-family totals.
 * bigc_prof_get: ms = summed launch durations, flops/bytes = summed ALGORITHMIC work of those launches. */
BIGC_API void bigc_prof_enable(int on);
BIGC_API void bigc_prof_reset(void);
BIGC_API int bigc_prof_collect(void);
BIGC_API int bigc_prof_get(int family, double* ms, long long* launches, double* flops, double* bytes);
BIGC_API int bigc_prof_family_count(void);
BIGC_API const char* bigc_prof_family_name(int family);

/* ------------------------------------------------------------------------------------
 * Kernel-level entry points (one per hand-written kernel family).  The model-level
 * functions below are compositions of these; tests call them directly to compare each
 * kernel with the oracle.
 * ---------------------------------------------------------------------------------- */

#define BIGC_ACT_NONE 0
#define BIGC_ACT_RELU 1
#define BIGC_ACT_SIGMOID 2

/* C[m, n] = act( sum_k A'[m, k] * W[n, k] + bias[n] ) + residual[m, n]
 *   replaces torch.nn.Linear / F.linear at every call site of the path
 *   (models/model_0v10.py:446-447,458,113,190-191,212,220,498,504) and, with the
 *   segmented row gather, nn.Conv1d(k=3,stride=2) as an implicit GEMM (:451).
 *   A' is A when a_idx == NULL.  Otherwise the K axis is cut in `nseg` segments of
 *   `seg_len` (K == nseg*seg_len) and row m, segment s of A' is row a_idx[m*nseg+s] of A
 *   (columns [0, seg_len)), or zeros when that index is negative.
 *   W is row-major (N, K) with leading dimension ldw (the torch weight layout).
 *   bias / residual may be NULL. The residual is added AFTER the activation.
 *   K must be a multiple of 4; lda, ldw multiples of 4; A and W 16-byte aligned. */
BIGC_API int bigc_linear_f32(void* stream, const float* A, int lda, const int32_t* a_idx, int nseg, int seg_len,
                    const float* W, int ldw, const float* bias, const float* residual, int ldr,
                    float* C, int ldc, int M, int N, int K, int act);

/* Arithmetic of bigc_linear_f32, bigc_mha_core(_drop)_f32 at head_dim 64 and bigc_role_fused_f32 (process-wide): 0 = f16
 * matrix cores on exactly split f32 operands (default; f32-level error, see csrc/linear_split.hip, csrc/attention_split.hip),
 * 1 = the exact f32 MFMA kernels.  Initial value from the environment variable
 * BIGC_LINEAR_EXACT.  bigc_linear_set_exact returns the previous mode. */
BIGC_API int bigc_linear_set_exact(int exact);
/* Split-K of small-grid GEMMs (csrc/linear_split.hip): allowed on the calling thread while `on` != 0 -- what the training entry points do
 * for the duration of a step (a launch of at most 128 tiles then splits K over up to eight workgroups per tile and adds the partial tiles
 * in split order); the eval forwards never allow it, a GEMM's result there must not depend on how many rows share the launch.  Exposed so
 * that the split and the unsplit form of bigc_linear_f32 can be compared shape by shape (tests/test_hip_ops.py).  Returns the previous state. */
BIGC_API int bigc_linear_splitk_allow(int on);
BIGC_API int bigc_linear_get_exact(void);
/* Opt-in reduced-precision modes of the GEMMs (never the default; also env BIGC_FAST = 1 | f16 | 2 | bf16): fast = 1: operands rounded
 * to ONE f16 each (11-bit mantissa), one f16 MFMA per product instead of three, f32 accumulation; fast = 2: the literal bf16
 * arithmetic (operands rounded to one bf16, one bf16 MFMA per product; model forwards then do not chain GEMMs through the f16 split
 * form).  These are the "bf16 fast mode" of the survey's build plan (BASELINE configs[1]): they do NOT
 * meet the 1e-4 logit tolerance and are gated on Recall@50 parity instead (tests/test_hip_fast.py).  Ignored while the exact mode is
 * on.  bigc_linear_set_fast returns the previous mode; bigc_linear_get_fast the mode in effect (0, 1 or 2). */
BIGC_API int bigc_linear_set_fast(int fast);
BIGC_API int bigc_linear_get_fast(void);
/* Accumulation policy of the model forwards' GEMMs on pre-split operands (bigc_forward, the nn.Linear / nn.Conv1d products of
 * models/model_0v10.py:446-458,178-225; process-wide, also env BIGC_ACCUM = plain | k1024 | front | nodecoder | blocked):
 * 4 = blocked (default): every such GEMM sums K in blocks of 128 that are folded into a running total, so that no product of the path
 * rounds coarser than a blocked CPU GEMM (the reference's) -- every video of the benchmark set within 1e-4 of a float64 evaluation;
 * 0 = plain: one f32 accumulator per element on the largest tiles (faster: 19.6 against 20.7 ms per 256 videos; up to 2e-4 from
 * float64 on the videos whose sharp attention amplifies round-off); 1 / 2 / 3: blocked for K >= 1024 only / for M >= 100 000 rows only /
 * for all but the decoder's GEMMs.  Returns the previous policy. */
BIGC_API int bigc_set_accum_policy(int policy);
BIGC_API int bigc_get_accum_policy(void);

/* y[r, :] = LayerNorm(x[r, :]) * gamma + beta over d columns, eps inside the sqrt
 *   (torch.nn.LayerNorm, models/model_0v10.py:81-82,171-173). d <= 2048, d % 4 == 0. */
BIGC_API int bigc_layernorm_f32(void* stream, const float* x, const float* gamma, const float* beta, float* y,
                       int rows, int d, float eps);

/* LayerNorm fused with the decoder's broadcast add: y[r, :] = LayerNorm(x[r, :]) + add[r % period, :] and, when y_plain is
 *   not NULL, y_plain[r, :] = LayerNorm(x[r, :]) as well  (norm1 followed by `query + pos_embedding`, models/model_0v10.py:187-189;
 *   norm3 of layer i feeding `q = k = query + pos` of layer i + 1, :181).  add == NULL: plain LayerNorm (y_plain must be NULL). */
BIGC_API int bigc_layernorm_add_f32(void* stream, const float* x, const float* gamma, const float* beta, float* y, int rows, int d,
                           float eps, const float* add, int period, float* y_plain);
/* nn.LayerNorm whose result the next GEMM reads in the split form "hr8" of bigc_gemm_ps: y (f32, may be NULL) and ys (hr8 rows of d * 4
 * bytes at pitch ldys_bytes); d in {64, 128, 256, 512}; an element with |y| >= 4096 sets *range_flag */
BIGC_API int bigc_layernorm_hr8_f32(void* stream, const float* x, const float* gamma, const float* beta, float* y, void* ys, long ldys_bytes,
                           int rows, int d, float eps, int* range_flag);
/* nn.LayerNorm(x + pre): the post-norm residual of a transformer sublayer (`norm1(query + self_attn(..))`, `norm2(query + roles)`,
 *   `norm3(t2 + ffn(t2))`, models/model_0v10.py:187-188,209-212,218-223) added by the LayerNorm, which reads it by whole rows, instead of by
 *   the GEMM in front of it (whose lanes own data rows: 16 bytes from each of 32 rows per load).  One f32 addition per element either way:
 *   the same bits.  y (f32; + add[r % period] when add != NULL, then ys must be NULL) and / or ys (hr8, pitch ldys_bytes); pre may alias y;
 *   d in {64, 128, 256, 512}. */
BIGC_API int bigc_layernorm_res_f32(void* stream, const float* x, const float* pre, const float* gamma, const float* beta, float* y, void* ys,
                           long ldys_bytes, int rows, int d, float eps, const float* add, int period, int* range_flag);

/* y[r, :] = x[r, :] + p[r % period, :]   (query + pos_embedding broadcast over the videos of a batch,
 *   models/model_0v10.py:182,189); x == NULL means zeros (tiles pred_query_init over the batch, :465) */
BIGC_API int bigc_add_rows_f32(void* stream, const float* x, const float* p, float* y, int rows, int d, int period);

/* Multi-head attention core for ragged sequences (torch.nn.MultiheadAttention after the
 *   in-projection and before out_proj: softmax(q k^T / sqrt(hd)) v, models/model_0v10.py:109,183;
 *   models/grd_model_v5.py:105).  Sequence s owns query rows [q_off[s], q_off[s+1]) of q/o and key
 *   rows [k_off[s], k_off[s+1]) of k/v; head h uses columns [h*hd, (h+1)*hd).
 *   hd in {8, 16, 32, 64}. lse (total query rows, n_head), optional: log-sum-exp of the scaled scores
 *   (saved for bigc_mha_bwd_f32). */
BIGC_API int bigc_mha_core_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                      float* o, int ldo, const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q,
                      int n_head, int hd, float* lse);

/* Role attention of the decoder (models/model_0v10.py:193-207):
 *   s_r[q, n] = p2a[q, r*half:(r+1)*half] . e2a[n, r*half:(r+1)*half] * scale   (r = 0 subject, 1 object)
 *   att_r = softmax_n(s_r) * softmax_r(s)[r]
 *   Video v owns query rows [v*nq, (v+1)*nq) of p2a and entity rows [e_off[v], e_off[v+1]) of e2a;
 *   att is stored per video as (2, nq, N_v) at float offset 2*nq*e_off[v]. */
BIGC_API int bigc_role_attention_f32(void* stream, const float* p2a, int ldp, const float* e2a, int lde, float* att,
                            const int32_t* e_off, int n_vid, int nq, int half, float scale);

/* values[r][v*nq + q, :] = att_v[r, q, :] @ enco[e_off[v]:e_off[v+1], :]   (models/model_0v10.py:211)
 *   values is (2, n_vid*nq, d) contiguous. */
BIGC_API int bigc_role_values_f32(void* stream, const float* att, const float* enco, int lde, float* values,
                         const int32_t* e_off, int n_vid, int nq, int d);

/* role attention and its values in one launch for videos of at most 64 entities (max_ent = the largest entity count of
 * the batch): same results as bigc_role_attention_f32 followed by bigc_role_values_f32; att may be NULL when only the
 * values are needed (every decoder layer but the last, models/model_0v10.py:467). */
BIGC_API int bigc_role_fused_f32(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* enco, int ldenc,
                        float* att, float* values, const int32_t* e_off, int n_vid, int nq, int max_ent, int half, int d, float scale);
/* The same with the first role-wise linear folded in (models/model_0v10.py:204-208): values_r = act(att_r @ proj_r + bias_r),
 * where proj_r = enco_output W_r^T was computed by the caller on the n_trk entity rows ((att_r @ enco) W^T = att_r @ (enco W^T));
 * proj_1 = proj + role_stride floats, ldproj >= d, bias_r (d) optional (both or none), relu != 0 applies max(., 0).
 * values row (v, q) of role r is written at values + r * out_role_stride + (v * nq + q) * ldval (ldval = 0: the role planes
 * (2, n_vid * nq, d) of bigc_role_fused_f32; out_role_stride = d, ldval = 2 d: the two roles side by side in one row, the A operand
 * of a K = 2 d GEMM).  nq <= 192 and half <= 256 take the split-operand LDS kernel, anything else (and exact mode) the f32-MFMA kernel. */
BIGC_API int bigc_role_fused_proj_f32(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* proj, int ldproj,
                             long role_stride, const float* bias0, const float* bias1, int relu, float* att, float* values,
                             long out_role_stride, int ldval, const int32_t* e_off, int n_vid, int nq, int max_ent, int half, int d,
                             float scale);

/* Tracklet front-end helpers ------------------------------------------------------- */

/* (R,4) xyxy pixel boxes -> (R,8) [cx,dcx,cy,dcy,w,dw,h,dh], boxes divided by the video's (w,h),
 *   forward differences inside each tracklet with 0 at the tracklet's last frame
 *   (models/model_0v10.py:401-420). trk_row_off (NT+1), trk_vid (NT), vid_wh (B,2). */
BIGC_API int bigc_bbox_feat_f32(void* stream, const float* boxes, const int32_t* trk_row_off, const int32_t* trk_vid,
                       const float* vid_wh, float* out, int n_trk);

/* Row gather table of the stride-2 k=3 temporal conv over the virtually stretched sequence
 *   (stretch: models/model_0v10.py:18-46, conv :302,451): for tracklet t, output position j
 *   (0 <= j < T_o = (Tmax-1)/2+1), tap c in {0,1,2}: stretched frame f = 2j-1+c; entry is -1 when
 *   f < 0 or f >= Tmax, else trk_row_off[t] + src(f) with src the stretch map.
 *   conv_idx is (C,3) with C = trk_conv_off[n_trk]. */
BIGC_API int bigc_conv_index(void* stream, const int32_t* trk_row_off, const int32_t* trk_conv_off, const int32_t* trk_vid,
                    const int32_t* vid_tmax, int32_t* conv_idx, int n_trk);

/* adaptive_max_pool1d(T_o -> P) per tracklet and channel, stored channel-major (index c*P+p)
 *   (models/model_0v10.py:456-457). conv (C, d) row-major; pooled (NT, d*P). */
BIGC_API int bigc_adaptive_maxpool(void* stream, const float* conv, const int32_t* trk_conv_off, float* pooled,
                          int n_trk, int d, int P);

/* Mean over the Tmax stretched frames of columns [col0, col0+w) of the tracklet's feature rows
 *   (models/model_0v10.py:470 I3D mean; models/model_0v7.py:473 classeme mean): a weighted mean
 *   with the stretch repeat counts. out (NT, w). */
BIGC_API int bigc_stretched_mean(void* stream, const float* feats, int ldf, int col0, int w, const int32_t* trk_row_off,
                        const int32_t* trk_vid, const int32_t* vid_tmax, float* out, int n_trk);

/* Prediction-head gather (models/model_0v10.py:485-503, models/model_0v7.py:485-508):
 *   soid[v, r, q] = argmax_n att_v[r, q, n] (first maximum), then per query row the concat
 *   [query | seg_1[s] | seg_1[o] | seg_2[s] | seg_2[o] | ...] is written to head_in, where segment i
 *   is a per-tracklet table (src[i], width[i]) indexed by the tracklet (by_cat[i]==0) or a per-category
 *   table indexed by cat id (by_cat[i]==1); bias_out[row, :] = bias_matrix[cat_s, cat_o, :]. */
BIGC_API int bigc_head_gather(void* stream, const float* att, const int32_t* e_off, const int32_t* trk_cat, int n_vid, int nq,
                     const float* query, int dq, int n_seg, const float* const* seg_src_host, const int32_t* seg_width_host,
                     const int32_t* seg_by_cat_host, const float* bias_matrix, int n_ecat, int n_pcat,
                     int32_t* soid, float* head_in, int ld_head, float* bias_out);

/* construct_triplet (models/model_0v10.py:707-785), one video per workgroup, index-exact:
 *   softmax + top-k per query, subject/object from soid, closed-interval overlap filter with the
 *   diagonal cleared, lexicographic unique of [pred, s_cat, o_cat, s_tid, o_tid] with the first
 *   highest-probability candidate as representative, background predicate (0) dropped.
 *   Outputs are padded to cap = nq*topk rows per video; n_out[v] rows are valid (0 => the reference
 *   returns None).  quint (B,cap,5) i32, scores (B,cap,3) f32, dura (B,cap,2) i32, qid (B,cap) i32. */
BIGC_API int bigc_construct_triplets(void* stream, const float* logits, int n_pcat, const int32_t* soid, const int32_t* e_off,
                            const int32_t* trk_dura, const int32_t* trk_cat, const float* trk_score, int n_vid, int nq,
                            int topk, int32_t* quint, float* scores, int32_t* dura, int32_t* qid, int32_t* n_out);

/* ------------------------------------------------------------------------------------
 * Model-level API: BIG-C eval forward (a1..a9 of SURVEY.md section 8) over one packed batch.
 * ---------------------------------------------------------------------------------- */

typedef struct {
    int32_t variant;        /* 0 = VidVRD model_0v10, 1 = VidOR model_0v7 */
    int32_t num_enti_cats, num_pred_cats;
    int32_t dim_feat, dim_extra; /* dim_extra = I3D width (variant 0) or classeme width carried in the features (variant 1), 0 if none */
    int32_t dim_clsme, dim_enti, dim_pred, dim_att, dim_ffn;
    int32_t enco_pool_len, n_enco_layers, n_deco_layers, n_att_head, num_querys;
    int32_t use_clsme;      /* variant 1 only */
    int32_t has_enti_emb;   /* EntiNameEmb table present (always for variant 0) */
} bigc_config;

typedef struct bigc_model bigc_model;

BIGC_API int bigc_create(const bigc_config* cfg, bigc_model** out);
BIGC_API void bigc_destroy(bigc_model* m);
/* register a parameter by its reference state_dict key; the pointer must stay valid */
BIGC_API int bigc_set_param(bigc_model* m, const char* key, const float* ptr, int64_t numel);
/* resolve all keys, build derived device tables (tap-major conv weight); call again after weights change */
BIGC_API int bigc_finalize(bigc_model* m, void* stream);
/* This handle's arithmetic mode: exact (1 = every GEMM / attention kernel on the f32 MFMA) and fast (1 / 2 = the opt-in f16 / bf16
 * reduced-precision GEMMs), each -1 to follow the process-wide default of bigc_linear_set_exact / bigc_linear_set_fast.  The mode applies to the calls
 * made THROUGH the handle (bigc_forward, bigc_train_*, bigc_pair_*), on the calling thread only: handles are independent of each
 * other (one per GPU and thread, as the reference's DataParallel runs its replicas, utils/DataParallel.py:68). */
BIGC_API int bigc_set_mode(bigc_model* m, int exact, int fast);

/* One packed batch of videos: the device-resident ragged tracklet store. */
typedef struct {
    int32_t n_vid, n_trk, n_rows, n_conv;  /* B, NT = sum N_v, R = sum L_i, C = sum N_v*T_o(v) */
    int32_t max_trk;               /* max_v N_v (<= 512) */
    const float* feats;            /* (R, dim_feat+dim_extra) */
    const float* boxes;            /* (R, 4) xyxy pixels */
    const int32_t* trk_row_off;    /* (NT+1) */
    const int32_t* trk_conv_off;   /* (NT+1) */
    const int32_t* trk_vid;        /* (NT) */
    const int32_t* trk_dura;       /* (NT,2) closed */
    const int32_t* trk_cat;        /* (NT) */
    const float* trk_score;        /* (NT) */
    const int32_t* vid_trk_off;    /* (B+1) */
    const int32_t* vid_tmax;       /* (B) */
    const float* vid_wh;           /* (B,2) */
    /* optional (NULL / 0 = absent): the first dim_feat columns of `feats` once more in the split form "hr8" of bigc_gemm_ps
     * (R rows of dim_feat * 4 bytes at pitch ld_feats_hr8 bytes, written by bigc_split_rows_f32 when the batch is packed), and the
     * int that pass raised when a feature was out of the form's range (|x| >= 4096, NaN, Inf).  With it the eval forward runs
     * fc_feat2enti.0 (models/model_0v10.py:447) on operands that are already split: the K = 2048 GEMM's loop is DMA + MFMA only. */
    const void* feats_hr8; int64_t ld_feats_hr8; const int32_t* feats_hr8_flag;
    /* optional (NULL = absent): the last dim_extra columns of `feats` (I3D / classeme, read by the time-mean of
     * models/model_0v10.py:470) as a tensor of their own, (R, dim_extra) f32 at pitch ld_extra floats.  A batch packed for the eval
     * forward (bigc_pack_tracklets) carries feats_hr8 + extra and NO `feats` (NULL): a feature element then exists once in HBM.
     * Every path that is not the default eval forward (training, exact / bf16 mode, BIGC_CHAINED=0; see bigc_forward_reads_hr8)
     * needs `feats` and fails with BIGC_ERR_ARG without it. */
    const float* extra; int64_t ld_extra;
    /* optional (NULL = absent), used when `feats` and `feats_hr8` are both NULL: address of every feature row of the batch, (R) entries,
     * row r = the (dim_feat + dim_extra) f32 values at feat_row_ptr[r], 16-byte aligned (bigc_row_pointers builds the table from the
     * per-tracklet addresses).  The default eval forward then reads the features IN PLACE from the callers' per-tracklet tensors --
     * fc_feat2enti.0 (models/model_0v10.py:447) splits them on their way into the GEMM, the time-mean of :470 walks the rows -- and
     * no packed copy of the features exists (the reference stacks them, :427-428). */
    const uint64_t* feat_row_ptr;
} bigc_batch;

typedef struct {
    float* pred_logits;   /* (B*Q, P) */
    float* att;           /* per video (2,Q,N_v) at float offset 2*Q*vid_trk_off[v]; total 2*Q*NT */
    float* pred_query;    /* (B*Q, dim_pred) last decoder layer output, may be NULL */
    float* enti2enco;     /* (NT, dim_enti) may be NULL */
    float* enco_output;   /* (NT, dim_enti) may be NULL */
    int32_t* soid;        /* (B,2,Q) */
    /* construct_triplet outputs, see bigc_construct_triplets; all may be NULL to skip a8 */
    int32_t* quint; float* scores; int32_t* dura; int32_t* qid; int32_t* n_out;
    /* may be NULL.  Set to 0 / 1 by bigc_forward: 1 = an operand of the split-operand GEMM path (csrc/gemm_ps.hip) left the range
     * its f16 pair format represents (|x| >= 4096 for an activation, NaN, Inf): the outputs are then INVALID and the caller must run
     * the step again with bigc_linear_set_exact(1).  Without this pointer such a step fails with BIGC_ERR_ARG up front (the
     * forward then uses the in-kernel-fallback GEMM of csrc/linear_split.hip instead). */
    int32_t* range_flag;
} bigc_outputs;

/* 1 if bigc_forward with a range_flag output, on this handle in its current arithmetic mode, takes a batch WITHOUT the packed f32 rows
 * (bigc_batch.feats == NULL): the visual features in the split form (feats_hr8 + extra) or read in place (feat_row_ptr); else 0 */
BIGC_API int bigc_forward_reads_hr8(const bigc_model* m);
BIGC_API size_t bigc_workspace_bytes(const bigc_model* m, const bigc_batch* b);
BIGC_API int bigc_forward(bigc_model* m, void* stream, const bigc_batch* b, void* workspace, size_t workspace_bytes,
                 int topk, const bigc_outputs* out);

/* ------------------------------------------------------------------------------------
 * Training step of BIG-C (models/model_0v10.py:510-704; tools/train_vidvrd.py:151-164): forward with saved
 * activations, Hungarian matching cost, loss + its gradients, and the backward pass.  Replaces torch autograd
 * over the reference's modules.  Dropout is not applied (parity is defined in eval() mode).
 * ---------------------------------------------------------------------------------- */

/* out (cols, rows) = in (rows, cols)^T */
BIGC_API int bigc_transpose_f32(void* stream, const float* in, float* out, int rows, int cols);
/* dW[n,k] += sum_m dY[m,n] X'[m,k] (X' optionally segment-gathered like bigc_linear_f32) and, if db != NULL,
 *   db[n] += sum_m dY[m,n]: weight/bias gradient of every nn.Linear / the conv; fp32 atomic accumulation. */
BIGC_API int bigc_linear_dw_f32(void* stream, const float* dY, int ldy, const float* X, int ldx, const int32_t* x_idx, int nseg, int seg_len,
                       float* dW, int ldw, float* db, int M, int N, int K);
/* dY *= (Y > 0) with Y the post-ReLU activation; n % 4 == 0 */
BIGC_API int bigc_relu_bwd_f32(void* stream, float* dY, const float* Y, long n);
/* y += x; n % 4 == 0 */
BIGC_API int bigc_axpy_f32(void* stream, float* y, const float* x, long n);
/* nn.LayerNorm backward: dx (rows,d); dgamma, dbeta accumulated */
BIGC_API int bigc_layernorm_bwd_f32(void* stream, const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta,
                           int rows, int d, float eps);
/* out[r % period, :] += x[r, :] summed over r (gradients of pos_embedding / pred_query_init, shared by all videos) */
BIGC_API int bigc_reduce_periodic_f32(void* stream, const float* x, float* out, int rows, int d, int period);
/* adaptive_max_pool1d backward (first maximum of each window); dconv (C,d) is fully overwritten */
BIGC_API int bigc_maxpool_bwd_f32(void* stream, const float* conv, const int32_t* trk_conv_off, const float* dpooled, float* dconv, int n_trk, int d, int P);
/* inverse of bigc_conv_index: G[s, k, :] = sum of the dconv rows whose tap k reads packed row s; G (R, 3, d) */
BIGC_API int bigc_conv_gather_bwd_f32(void* stream, const float* dconv, const int32_t* trk_row_off, const int32_t* trk_conv_off,
                             const int32_t* trk_vid, const int32_t* vid_tmax, float* G, int n_trk, int d);
/* backward of bigc_mha_core_f32 from the saved output o and log-sum-exp lse; delta_scratch (total_q_rows, n_head) */
BIGC_API int bigc_mha_bwd_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                     const float* dO, int lddo, const float* lse, float* delta_scratch, float* dq, int lddq, float* dk, int lddk,
                     float* dv, int lddv, const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q, int max_k, long total_q_rows,
                     int n_head, int hd);
/* d(raw role scores) (same layout as att) from d att; already multiplied by `scale` */
BIGC_API int bigc_role_attention_bwd_f32(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* datt, float* ds,
                                const int32_t* e_off, int n_vid, int nq, int half, float scale);
/* per-video products around the role attention. mode 0: datt = dvals @ enco^T; 1: denco += att^T @ dvals (both roles);
 *   2: dp2a = ds @ e2a; 3: de2a = ds^T @ p2a. d = dim_enti, half = dim_att/2. */
BIGC_API int bigc_role_small_gemm_f32(void* stream, int mode, const float* att_or_ds, const float* dvals_or_p2a, int ld1, const float* enco_or_e2a,
                             int ld2, float* out, int ldo, const int32_t* e_off, int n_vid, int nq, int d, int half, int max_trk);
/* two of those products in ONE launch (they are independent and each fills a fraction of the chip at training batch sizes):
 *   pair 0: datt = dvals @ enco^T -> out_a and denco += att^T @ dvals -> out_b (x = att, y = dvals, z = enco; ld_y == ld_z);
 *   pair 1: dp2a = ds @ e2a -> out_a and de2a = ds^T @ p2a -> out_b (x = ds, y = p2a, z = e2a); ldo_b: leading dimension of out_b (and of
 *   out_a in pair 1).  Same arithmetic as the single calls, bit for bit. */
BIGC_API int bigc_role_small_gemm_pair_f32(void* stream, int pair, const float* x, const float* y, int ld_y, const float* z, int ld_z, float* out_a,
                                  float* out_b, int ldo_b, const int32_t* e_off, int n_vid, int nq, int d, int half, int max_trk);
/* vIoU of every (proposal, GT trajectory) pair (vIoU_ts utils/utils_func.py:437-471 as used by enti_viou_align,
 *   models/model_0v10.py:569-581): out (B, max_trk, g_max), 0 where the closed durations do not overlap or the pair
 *   does not exist. GT trajectories are packed like the proposals: gt_boxes (sum frames, 4), gt_row_off (G+1),
 *   gt_dura (G,2) closed, gt_traj_off (B+1). */
BIGC_API int bigc_viou_matrix(void* stream, const float* boxes, const int32_t* trk_row_off, const int32_t* trk_dura, const int32_t* e_off,
                     const float* gt_boxes, const int32_t* gt_row_off, const int32_t* gt_dura, const int32_t* gt_traj_off,
                     int n_vid, int max_trk, int g_max, float* out);
/* vIoU of n_pairs listed pairs (a, b) of trajectory slices: slice i of side A = boxes_a rows [seg_a[3i], seg_a[3i] + seg_a[3i+2]) covering
 *   the frames [seg_a[3i+1], seg_a[3i+1] + seg_a[3i+2]); same for side B.  The evaluator's `viou` (VidVRDhelperEvalAPIs/common.py:65-106),
 *   f32 accumulation.  out (n_pairs). */
BIGC_API int bigc_viou_pairs(void* stream, const float* boxes_a, const int32_t* seg_a, const float* boxes_b, const int32_t* seg_b,
                    const int32_t* pairs, int n_pairs, float* out);
/* Hungarian cost (models/model_0v10.py:617-636): cost[v,q,g] (B, nq, g_max) = cost_cls * CE + cost_adj * mean BCE;
 *   g_off (B+1) prefix of GT predicates per video, gt_pred (sum G), gt_adj per video (2, G_v, N_v) at float offset adj_off[v]. */
BIGC_API int bigc_match_cost(void* stream, const float* logits, int n_pcat, const float* att, const int32_t* e_off, const int32_t* g_off,
                    const int32_t* gt_pred, const float* gt_adj, const int64_t* adj_off, int n_vid, int nq, int g_max,
                    float cost_cls, float cost_adj, float* cost);
/* Loss terms (models/model_0v10.py:648-703) and their gradients. target (B*nq) = matched GT predicate class or 0,
 *   match_g (B*nq) = matched GT index or -1; n_pos/n_neg/n_adj = element counts of the three means.
 *   loss_out[3] = {cls_pos, cls_neg, adj}; dlogits (B*nq, ldg) (pad columns untouched), datt like att; both NULL: the loss terms only. */
BIGC_API int bigc_loss_grad(void* stream, const float* logits, int n_pcat, const float* att, const int32_t* e_off, const int32_t* g_off,
                   const float* gt_adj, const int64_t* adj_off, const int32_t* target, const int32_t* match_g, int n_vid, int nq,
                   float n_pos, float n_neg, float n_adj, float loss_cls, float loss_adj, float neg_weight,
                   float gs_pos, float gs_neg, float gs_adj, /* upstream gradients of the three loss terms */
                   float* loss_out, float* dlogits, int ldg, float* datt);
/* 1 when a training step of this handle (in its current arithmetic mode) can take a batch without packed f32 rows (bigc_batch.feats
 *   NULL) and read the tracklets' feature rows in place through bigc_batch.feat_row_ptr, like the default eval forward: the first
 *   visual-feature GEMM (models/model_0v10.py:447), its weight gradient and the time-mean of the I3D / classeme columns (:470) then
 *   walk the callers' tensors; 0: pack f32 rows (exact-GEMM mode, the pairwise baseline, feature widths that are not multiples of 4) */
BIGC_API int bigc_train_reads_rows(const bigc_model* m);
/* bigc_loss_grad with the three upstream gradients read from device memory (upstream3[0..2]): the backward of a step that does not
 *   wait for the GPU (autograd hands them over as a device tensor) */
BIGC_API int bigc_loss_grad_dev(void* stream, const float* logits, int n_pcat, const float* att, const int32_t* e_off, const int32_t* g_off,
                       const float* gt_adj, const int64_t* adj_off, const int32_t* target, const int32_t* match_g, int n_vid, int nq,
                       float n_pos, float n_neg, float n_adj, float loss_cls, float loss_adj, float neg_weight, const float* upstream3,
                       float* loss_out, float* dlogits, int ldg, float* datt);
/* The host half of the reference's matching, on the device.
 * bigc_align_adjacency: the tail of enti_viou_align (models/model_0v10.py:583-602): viou (B, max_trk, t_max) as written by
 *   bigc_viou_matrix, adj_raw = per video the GT adjacency (2, G_v, T_v) f32 at float offset raw_off[v] (T_v = gt_traj_off[v+1] -
 *   gt_traj_off[v] GT trajectories) -> gt_adj per video (2, G_v, N_v) at float offset adj_off[v], re-indexed by proposals: a proposal
 *   takes the column of its best-overlapping GT trajectory if it passes viou_th for some trajectory or is the best proposal of a
 *   trajectory nobody passes for, else zeros.
 * bigc_hungarian: scipy.optimize.linear_sum_assignment (models/model_0v10.py:637) on cost[v, :, :G_v] (B, nq, g_max) of every video, G_v <
 *   nq: the shortest-augmenting-path algorithm scipy implements (Crouse 2016), in float64 and in scipy's candidate order, so the
 *   assignment is scipy's also where the optimum is not unique.  -> target (B*nq) = class of the GT predicate matched to the query or 0,
 *   match_g (B*nq) = its index or -1.  bad[0] |= 1 when a video's costs hold NaN / -inf or admit no assignment (scipy raises
 *   ValueError there); that video is left unmatched. */
BIGC_API int bigc_align_adjacency(void* stream, const float* viou, int max_trk, int t_max, const int32_t* e_off, const int32_t* gt_traj_off,
                         const float* adj_raw, const int64_t* raw_off, const int32_t* g_off, const int64_t* adj_off, int n_vid,
                         float viou_th, float* gt_adj);
BIGC_API int bigc_hungarian(void* stream, const float* cost, const int32_t* g_off, const int32_t* gt_pred, int n_vid, int nq, int g_max,
                   int32_t* target, int32_t* match_g, int* bad);

/* gradient destination of a trainable parameter (accumulated; NULL unregisters). Frozen parameters are never registered. */
BIGC_API int bigc_set_grad(bigc_model* m, const char* key, float* ptr);
/* moves every registered gradient destination from the buffer [old_base, old_base + numel) to the same offset of new_base: the
 * host side lays all gradients out in one flat buffer, allocates a fresh one per step and keeps the layout (one call per step
 * instead of one bigc_set_grad per parameter) */
BIGC_API int bigc_rebase_grads(bigc_model* m, float* old_base, float* new_base, int64_t numel);
BIGC_API size_t bigc_train_workspace_bytes(const bigc_model* m, const bigc_batch* b);
/* forward keeping every activation the backward needs inside `workspace` (which must stay untouched until
 * bigc_train_backward); pred_logits (B*Q, P), att (2*Q*NT) of the last decoder layer */
BIGC_API int bigc_train_forward(bigc_model* m, void* stream, const bigc_batch* b, void* workspace, size_t workspace_bytes,
                       float* pred_logits, float* att);
/* dlogits (B*Q, ldg) with ldg = num_pred_cats rounded up to 4 and zero pad columns; datt (2*Q*NT) */
BIGC_API int bigc_train_backward(bigc_model* m, void* stream, const bigc_batch* b, void* workspace, size_t workspace_bytes,
                        const float* dlogits, int ldg, const float* datt);

/* ------------------------------------------------------------------------------------
 * Temporal grounding scorer (models/grd_model_v5.py, class DEBUG), eval path, one video per call
 * (the reference asserts batch size 1 at test time, :211).
 * ---------------------------------------------------------------------------------- */

/* depth-wise Conv1d(groups=C, kernel ks, padding ks/2, stride 1) over n_seq sequences of length T stored as
 *   rows (n_seq*T, d): y[s,t,c] = b[c] + sum_j w[c,j] * x[s, t+j-ks/2, c]  (models/grd_model_v5.py:44,54).
 *   w is the torch (d,1,ks) weight; d % 4 == 0; ks odd <= 7. */
BIGC_API int bigc_dwconv_f32(void* stream, const float* x, const float* w, const float* b, float* y, int n_seq, int T, int d, int ks);

/* Video <-> query co-attention and feature fusion (models/grd_model_v5.py:346-365) for n queries of 3 words:
 *   sim[i,t,l] = proj[t,:] . query[i,l,:]; sim_r = softmax_l, sim_c = softmax_t;
 *   A = sim_r @ query_i, B = (sim_r sim_c^T) @ video evaluated as sim_r @ (sim_c^T @ video);
 *   out[i,t,:] = [video[t], A, A*video[t], B*video[t]]  (4d wide).
 *   video, proj (T,d); query (n*3,d); out (n*T, 4d). T <= 2048, d % 4 == 0, d <= 256. */
BIGC_API int grd_coattention_f32(void* stream, const float* video, const float* proj, const float* query, float* out, int n, int T, int d);

/* Multi-bin post-processing of the three heads (models/grd_model_v5.py:533-572 with temporal_pooling :697-737
 *   and temporal_nms :667-695): scores = sigmoid(conf)*sigmoid(cls); per (query, bin) the spans
 *   [clip[t]-left, clip[t]+right] of the clips scoring above score_th*top whose gIoU with the top clip's span
 *   exceeds tiou_th are merged, clipped to the subject/object overlap span `info` (or replaced by it when
 *   disjoint), the overlap span is appended as bin k with probability 1, greedy 1-D NMS (tIoU >= nms_th
 *   suppresses), at least one bin kept per query, and the overlap bin's probability zeroed when no real bin
 *   exceeds bins_th.  regrs (n,T,2k) already passed through the sigmoid; conf, cls (n,T,k) logits;
 *   clip (T) = linspace(0,1,T); info (n,2). Outputs pooled_se (n,k+1,2), bins_probs (n,k+1), bins_mask (n,k+1) u8.
 *   k <= 16. status[0] is set to 1 if some (query,bin) had an empty pooling set (the reference raises there, :726). */
BIGC_API int grd_postprocess(void* stream, const float* regrs, const float* conf, const float* cls, const float* clip, const float* info,
                    int n, int T, int k, float score_th, float tiou_th, float bins_th, float nms_th,
                    float* pooled_se, float* bins_probs, uint8_t* bins_mask, int32_t* status);

typedef struct {
    int32_t dim_feat, dim_clsme, dim_hidden, num_bins;
    int32_t n_enti_emb, n_pred_emb;   /* rows of EntiNameEmb / PredNameEmb (81 / 51 in the reference) */
} grd_config;

typedef struct grd_model grd_model;

BIGC_API int grd_create(const grd_config* cfg, grd_model** out);
BIGC_API void grd_destroy(grd_model* m);
BIGC_API int grd_set_param(grd_model* m, const char* key, const float* ptr, int64_t numel);
BIGC_API int grd_finalize(grd_model* m, void* stream);
/* as bigc_set_mode, for the grounding scorer's handle (grd_forward*, grd_train_*) */
BIGC_API int grd_set_mode(grd_model* m, int exact, int fast);
BIGC_API size_t grd_workspace_bytes(const grd_model* m, int T, int n);

typedef struct {
    float* pooled_se;     /* (n, k+1, 2) normalised [0,1] */
    float* bins_probs;    /* (n, k+1) */
    uint8_t* bins_mask;   /* (n, k+1) */
    int32_t* status;      /* (1) */
    float* regrs;         /* (n*T, 2k) optional */
    float* conf;          /* (n*T, k) optional */
    float* cls;           /* (n*T, k) optional */
    float* video_enc;     /* (T, d) optional */
    float* query_enc;     /* (3n, d) optional */
    float* combined_enc;  /* (n*T, d) optional */
} grd_outputs;

/* quint (n,5) int32 [pred, s_cat, o_cat, s_tid, o_tid] and dura (n,2) int32 as produced by bigc_forward;
 * pos_video (T,d), pos_query (3,d): PosEncoder tables (models/grd_model_v5.py:58-78) evaluated by the host
 * exactly as the reference does; clip (T) = linspace(0,1,T). */
BIGC_API int grd_forward(grd_model* m, void* stream, const float* video_feat, int T, const int32_t* quint, const int32_t* dura, int n,
                float video_len, const float* pos_video, const float* pos_query, const float* clip,
                float score_th, float tiou_th, float bins_th, float nms_th,
                void* workspace, size_t workspace_bytes, const grd_outputs* out);

/* Several videos per call.  The reference scores one video per forward (models/grd_model_v5.py:196-221, the loop of
 * tools/eval_vidor.py); every step of it is row-wise or per (query, clip sequence), so the videos of a batch are laid back to
 * back -- clip rows of all videos (total_T), queries of all videos (n_query), and per query the clips of its video (total_rows =
 * sum over queries of T of its video) -- and the per-sequence steps (depth-wise convolutions, attention, co-attention, temporal
 * pooling / NMS) follow the offset tables below.  Results are identical to the per-video calls.  All tables in device memory. */
typedef struct {
    int32_t n_vid, n_query, max_T;
    int64_t total_T, total_rows;
    const int32_t* vid_row_off;   /* (n_vid + 1) first clip row of every video */
    const int32_t* q_vid;         /* (n_query) video of every query; queries grouped by video, in video order */
    const int32_t* q_row_off;     /* (n_query + 1) first combined row of every query */
    const float* vid_len;         /* (n_vid) video_len in frames */
} grd_batch;
BIGC_API size_t grd_batch_workspace_bytes(const grd_model* m, const grd_batch* b);
/* video_feat (total_T, dim_feat), quint (n_query, 5), dura (n_query, 2); pos_video (>= max_T, d), pos_query (3, d); clip (total_T) =
 * the videos' linspace(0, 1, T_v) tables back to back; outputs as grd_outputs with n = n_query and n*T -> total_rows, T -> total_T;
 * status (1 + n_vid): [any, per video] set to 1 where temporal pooling selected no clip (the reference raises there). */
BIGC_API int grd_forward_batch(grd_model* m, void* stream, const float* video_feat, const grd_batch* b, const int32_t* quint, const int32_t* dura,
                      const float* pos_video, const float* pos_query, const float* clip,
                      float score_th, float tiou_th, float bins_th, float nms_th,
                      void* workspace, size_t workspace_bytes, const grd_outputs* out);

/* ---- grounding scorer, training step (models/grd_model_v5.py:375-389 DEBUG._forward_train) -------------------------
 * One video per call, as the reference loops (:379-385); the loss means run over the whole batch (:444-527), so
 * the host passes the batch-wide element counts to grd_loss_grad.
 *
 * grd_set_grad        gradient destination for a parameter key (same keys as grd_set_param; NULL unregisters);
 *                     gradients are ACCUMULATED (zero them once per step).
 * grd_train_forward   forward_propagation (:331-373) for the n queries (positives then sampled negatives,
 *                     prepare_gt_data :253-306) of one video, activations kept in the workspace.
 *                     regrs (n*T, 2k) post-sigmoid, conf / cls (n*T, k) logits.
 * grd_loss_grad       the five loss terms (:444-527) over the entries the host derived with map2bins (:391-442):
 *                     positive entries (query pos_i, bin pos_b, GT predicate pos_j) and negative entries
 *                     (query neg_i, bin neg_b); labels of get_gt_labels (:224-250) are evaluated on the fly from
 *                     clip (T) = linspace(0,1,T) and target (n_gt,2) normalised spans.  loss[5] accumulates
 *                     [pos_cls, neg_cls, pos_ct, neg_ct, regr] (already multiplied by the loss factors) and
 *                     dregr (n*T, round4(2k)) / dconf, dcls (n*T, round4(k)) accumulate d total / d pre-activation,
 *                     scaled by gs_host[5] (upstream gradient of each term).  count_* are the denominators of the
 *                     batch-wide means: positive elements, negative elements, elements with centerness > 0.
 * grd_train_backward  backward pass from those three gradient arrays through the network. */
BIGC_API int grd_set_grad(grd_model* m, const char* key, float* ptr);
/* training-mode dropout of the three QANet layers (models/grd_model_v5.py:123-136; base p = 0.1, 0.05 after conv block 1),
 * counter-based masks as bigc_set_dropout; the host passes a different seed for every video of a batch. 0 = off. */
BIGC_API int grd_set_dropout(grd_model* m, float p, uint32_t seed);
BIGC_API size_t grd_train_workspace_bytes(const grd_model* m, int T, int n);
BIGC_API int grd_train_forward(grd_model* m, void* stream, const float* video_feat, int T, const int32_t* quint, const int32_t* dura, int n,
                      float video_len, const float* pos_video, const float* pos_query, void* workspace, size_t workspace_bytes,
                      float* regrs, float* conf, float* cls);
BIGC_API int grd_loss_grad(void* stream, const float* regrs, const float* conf, const float* cls, const float* clip, const float* target,
                  const int32_t* pos_i, const int32_t* pos_b, const int32_t* pos_j, int n_pos, const int32_t* neg_i,
                  const int32_t* neg_b, int n_neg, int T, int k, float count_pos, float count_neg, float count_mask,
                  float f_cls, float f_ct, float f_regr, const float* gs_host, float* loss, float* dregr, float* dconf, float* dcls);
BIGC_API int grd_train_backward(grd_model* m, void* stream, const float* video_feat, int T, const int32_t* quint, int n,
                       void* workspace, size_t workspace_bytes, const float* dregr, const float* dconf, const float* dcls);

/* ---- dropout of the training paths --------------------------------------------------------------------------------------
 * The reference trains with nn.Dropout / attention dropout p = 0.1 in its transformer layers (models/model_0v10.py:75-84,
 * 109-114,151,164-169).  Masks here are counter-based: element idx of site `site` is kept iff hash(seed, site, idx) >=
 * p * 2^32 (common.h: drop_keep), kept values scaled by 1/(1-p); forward and backward regenerate the same mask.
 * torch's Philox stream cannot be reproduced, so parity with the reference is defined with dropout off.
 *   bigc_set_dropout      p and per-step seed used by bigc_train_forward / bigc_train_backward (0 = off)
 *   bigc_dropout_f32      in place on the logical (rows, n) matrix x with leading dimension ld, idx = r * n + c
 *   bigc_mha_core_drop_f32 / bigc_mha_bwd_drop_f32   attention core with dropout on the probabilities,
 *                         idx = ((global query row * n_head + head) << 12) + key index inside the sequence */
BIGC_API int bigc_set_dropout(bigc_model* m, float p, uint32_t seed);
BIGC_API int bigc_dropout_f32(void* stream, float* x, long rows, int n, int ld, float p, uint32_t seed, uint32_t site);
BIGC_API int bigc_mha_core_drop_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                           float* o, int ldo, const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q,
                           int n_head, int hd, float* lse, float drop_p, uint32_t drop_seed, uint32_t drop_site);
BIGC_API int bigc_mha_bwd_drop_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                          const float* o, int ldo, const float* dO, int lddo, const float* lse, float* delta_scratch,
                          float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                          const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q, int max_k, long total_q_rows,
                          int n_head, int hd, float drop_p, uint32_t drop_seed, uint32_t drop_site);

/* ---- Base_C, the explicit pairwise baseline (models/model_pairwise_baseline.py:174-273), eval forward ------------------------
 * A model created with variant = 2 holds the front-end parameters plus fc_pred2logits.{0,2} and bias_matrix (same keys as
 * the reference's state_dict).  bigc_pair_forward runs the front-end and scores EVERY ordered tracklet pair (s, o), s != o:
 * pair p of video v (row-major over the N_v x N_v grid without its diagonal, trajid2pairid :105-112) is row
 * vid_pair_off[v] + p of pair_logits (n_pairs, num_pred_cats).  enti2enco (n_trk, dim_enti) is optional. */
BIGC_API size_t bigc_pair_workspace_bytes(const bigc_model* m, const bigc_batch* b, long n_pairs);
BIGC_API int bigc_pair_forward(bigc_model* m, void* stream, const bigc_batch* b, const int32_t* vid_pair_off, long n_pairs, void* workspace,
                      size_t workspace_bytes, float* enti2enco, float* pair_logits);

/* ---- Base_C training step (models/model_pairwise_baseline.py:276-312; loop body tools/train_vidor.py:305-330) ---------------
 * pair_so (n_pairs, 2): GLOBAL tracklet rows (the video's first tracklet row added to the reference's pairid2trajids) of
 * the labelled pairs of all videos of the batch, concatenated in batch order.  bigc_pair_train_forward keeps every
 * activation in `workspace` (untouched until the backward) and writes pair_logits (n_pairs, num_pred_cats);
 * bigc_bce_logits_f32 is F.binary_cross_entropy_with_logits(x, y, reduction='mean') over a (rows, cols) block: *loss +=
 * the mean, dx = gscale * d mean / dx (either may be NULL); bigc_pair_train_backward takes dlogits (n_pairs, ldg) with
 * ldg = num_pred_cats rounded up to 4 (zero pad columns) and accumulates into the bigc_set_grad destinations. */
BIGC_API size_t bigc_pair_train_workspace_bytes(const bigc_model* m, const bigc_batch* b, long n_pairs);
BIGC_API int bigc_pair_train_forward(bigc_model* m, void* stream, const bigc_batch* b, const int32_t* pair_so, long n_pairs, void* workspace,
                            size_t workspace_bytes, float* pair_logits);
BIGC_API int bigc_pair_train_backward(bigc_model* m, void* stream, const bigc_batch* b, const int32_t* pair_so, long n_pairs, void* workspace,
                             size_t workspace_bytes, const float* dlogits, int ldg);
BIGC_API int bigc_bce_logits_f32(void* stream, const float* x, int ldx, const float* y, int ldy, long rows, int cols, float gscale, float* loss,
                        float* dx, int lddx);

/* ---- tracking-result ingest (dataloaders/dataloader_vidvrd.py:585-700, utils/utils_func.py:273-317) ---------------------
 * Assembles the feature rows of the kept tracklets from the raw per-box rows of a video's `.npy` (uploaded once):
 * output row r = columns [col0, col0+width) of raw row src_a[r] followed by columns [col02, col02+width2) of raw2 row
 * src_a[r] (the I3D file, optional) when src_b[r] == -2 (a negative src_a gives a zero row: box without detection);
 * otherwise the row numpy.linspace(row src_a[r], row src_b[r], num = step_div[r] + 1, axis=0)[step_i[r]] evaluated in
 * float64 with numpy's arithmetic (incl. its any-step-zero branch) and rounded to float32, a negative index standing
 * for the zero vector.  Bit-identical to the reference's host interpolation. */
BIGC_API int bigc_assemble_rows_f32(void* stream, const float* raw, int ld_raw, int col0, int width, const float* raw2, int ld_raw2,
                           int col02, int width2, const int32_t* src_a, const int32_t* src_b, const int32_t* step_i,
                           const int32_t* step_div, float* out, int ld_out, int n_rows);

/* ---- ragged pack of a batch (the stacking half of models/model_0v10.py:391-430; the per-tracklet tensors are those of
 * dataloaders/dataloader_vidvrd.py:14-77 `TrajProposal`, moved to the device one by one by its .to(), :66-77) --------------------
 * ONE launch gathers every tracklet's rows into the packed store of bigc_batch.  Tables (device): feat_src[t] / box_src[t] =
 * address of tracklet t's (L_t, width) feature rows / (L_t, 4) boxes, f32, unit column stride; feat_ld[t] = row pitch of the
 * feature rows in floats (NULL: width); trk_row_off as in bigc_batch.  Outputs, each optional: feats (R, width) f32 at pitch
 * ld_feats; feats_hr8: columns [0, dim_feat) in the split form of bigc_gemm_ps at pitch ld_hr8_bytes (range_flag is or-ed with 1
 * when an element is outside its range); extra: columns [dim_feat, width) f32 at pitch ld_extra; boxes (R, 4).  feat_src may
 * be NULL (boxes only).  vectorized = 1 promises 16-byte aligned sources / pitches (16-byte loads), 0 = any alignment. */
BIGC_API int bigc_pack_tracklets(void* stream, const uint64_t* feat_src, const int32_t* feat_ld, const uint64_t* box_src,
                        const int32_t* trk_row_off, int n_trk, long n_rows, int width, int dim_feat, int vectorized, float* feats, long ld_feats,
                        void* feats_hr8, long ld_hr8_bytes, float* extra, long ld_extra, float* boxes, int32_t* range_flag);
/* row_ptr[trk_row_off[t] + i] = feat_src[t] + 4 * i * (feat_ld ? feat_ld[t] : width): the per-row address table of bigc_batch.feat_row_ptr */
BIGC_API int bigc_row_pointers(void* stream, const uint64_t* feat_src, const int32_t* feat_ld, int width, const int32_t* trk_row_off, int n_trk,
                      uint64_t* row_ptr);
/* bigc_stretched_mean over rows given by address: columns [col0, col0 + w) of row r at row_ptr[r] (16-byte aligned; col0, w multiples of 4) */
BIGC_API int bigc_stretched_mean_rows(void* stream, const uint64_t* row_ptr, int col0, int w, const int32_t* trk_row_off, const int32_t* trk_vid,
                             const int32_t* vid_tmax, float* out, int n_trk);
/* per-video metadata of the same containers: cat_ids (N_v) int64, scores (N_v) f32, traj_durations (N_v, 2) int64 at the
 * addresses cat_src[v] / score_src[v] / dura_src[v] -> trk_cat int32, trk_score f32, trk_dura int32 of bigc_batch */
BIGC_API int bigc_pack_video_meta(void* stream, const uint64_t* cat_src, const uint64_t* score_src, const uint64_t* dura_src,
                         const int32_t* vid_trk_off, int n_vid, int32_t* trk_cat, float* trk_score, int32_t* trk_dura);
/* host -> device copy by a kernel on `stream`: src_mapped is page-locked host memory mapped into the device's address space
 * (torch pin_memory() / hipHostMalloc); ordered like any other launch, no use of the runtime's copy queues */
BIGC_API int bigc_copy_mapped(void* stream, void* dst, const void* src_mapped, long nbytes);
/* progress mark: mapped_word[0] = value (8-byte aligned page-locked host memory, mapped) once everything queued on `stream` before
 * the call has run; when or_flag (device int, may be NULL) is non-zero, mapped_word[1] = value as well.  The host polls the word. */
BIGC_API int bigc_stamp(void* stream, void* mapped_word, unsigned long long value, const int* or_flag);
/* padded outputs of bigc_construct_triplets / bigc_forward -> the reference's return types back to back
 * (models/model_0v10.py:776-785: int64 quintuples, f32 scores, int64 duration intersections, int64 query ids per video):
 * video v's n_out[v] rows at rows [out_off[v], out_off[v+1]) of the compact arrays (each sized for n_vid * cap rows) */
BIGC_API int bigc_compact_triplets(void* stream, const int32_t* quint, const float* scores, const int32_t* dura, const int32_t* qid,
                          const int32_t* n_out, int n_vid, int cap, int64_t* out_quint, float* out_scores, int64_t* out_dura,
                          int64_t* out_qid, int32_t* out_off);

/* ---- Base_C.construct_triplet (models/model_pairwise_baseline.py:314-395) for all videos of a batch ----------------------------
 * pair_logits (n_pairs, n_pred): logits of every ordered tracklet pair (bigc_pair_forward; video v's pairs start at vid_pair_off[v],
 * row-major (s, o) without the diagonal).  Per pair: softmax, top-k; candidates whose subject / object durations do not overlap
 * are dropped; rows in the lexicographic order of torch.unique(dim=0) over [pred, s_cat, o_cat, s_tid, o_tid]; background
 * (pred 0) removed.  Outputs: quint (n_pairs*topk, 5) int32, scores (n_pairs*topk, 3) = [pred prob, s score, o score], dura
 * (n_pairs*topk, 2) closed overlap span; video v's rows start at row vid_pair_off[v]*topk, n_rows[v] of them are valid.
 * rt_topk > 0 (the reference's rt_triplets_topk, <= 2048): additionally the rt_topk rows of every video with the largest mean
 * score, descending: quint_top (B, rt_topk, 5), scores_top, dura_top, n_top (B).
 * bm_off (B) int64 / bm_words (B): offset (in 32-bit words) of video v's bitmap in the scratch region and its words per
 * predicate class, ceil(N_v^2 / 32); bitmap_words_total = sum_v n_pred * bm_words[v]. */
BIGC_API size_t bigc_pair_triplets_workspace_bytes(int n_vid, int n_trk, long n_pairs, int topk, long bitmap_words_total);
BIGC_API int bigc_pair_triplets(void* stream, const float* pair_logits, int n_pred, int topk, const int32_t* vid_trk_off,
                       const int32_t* vid_pair_off, const int32_t* trk_dura, const int32_t* trk_cat, const float* trk_score,
                       int n_vid, int n_trk, long n_pairs, const long* bm_off, const int32_t* bm_words, long bitmap_words_total,
                       void* workspace, size_t workspace_bytes, int32_t* quint, float* scores, int32_t* dura, int32_t* n_rows,
                       int rt_topk, int32_t* quint_top, float* scores_top, int32_t* dura_top, int32_t* n_top);

/* ---- GEMM on split operands (csrc/gemm_ps.hip) -----------------------------------------------------------------------------
 * Same contract as bigc_linear_f32 -- C = act(A W^T + bias) + residual, replacing nn.Linear / nn.Conv1d(k3,s2) of
 * models/model_0v10.py:290-309,446-458 -- for activations kept in the library's split form "hr8": every 8 consecutive f32 of
 * a row are stored as 16 bytes of f16 heads h = f16(x) followed by 16 bytes of f16 residuals r = f16((x - h) 2^11), round to
 * nearest (same footprint and pitch as f32).  a_fmt 0: A is f32 (split inside the kernel), 1: A is hr8 (optionally row-gathered:
 * a_idx (M, nseg), -1 = zero row, K = nseg segments).  Outputs: f32 C and / or hr8 Cs (at least one).  K % 32 == 0, N % 8 == 0.
 * range_flag (device int, may be null) is OR-ed with 1 when an element being split is >= 32768 in magnitude, NaN or Inf: the
 * result is then invalid and the caller must redo the computation on the exact path (bigc_linear_set_exact(1)). */
BIGC_API int bigc_gemm_ps(void* stream, const void* A, long lda_bytes, int a_fmt, const int32_t* a_idx, int nseg, const float* W, int ldw,
                 const float* bias, const float* residual, long ldr, float* C, long ldc, void* Cs, long ldcs_bytes,
                 int M, int N, int K, int act, int* range_flag);
/* The same product with the rows of A given by ADDRESS (row_ptr[m] -> K floats, 16-byte aligned): the first visual-feature layer
 * fc_feat2enti.0 (models/model_0v10.py:446, on `features[:, :dim_feat]` of :438) reading the callers' per-tracklet tensors where they
 * are -- the reference stacks them first (:391-430, utils/utils_func.py:49-91).  The raw f32 rows travel HBM -> LDS by DMA and are split
 * in LDS.  accum: 0 = one f32 accumulator per element (tile size chosen for speed); 1 = the correction terms of the split products in a
 * second accumulator (the rounding behaviour of the library's in-loop-split GEMM); 2 = blocked accumulation (partial sums over K = 128
 * folded into a running total; needs K % 128 == 0, else as 0): rounds no coarser than a blocked CPU GEMM.  1 and 2 run on 128-row tiles.
 * C and / or Cs as above. */
BIGC_API int bigc_gemm_ps_rows(void* stream, const uint64_t* row_ptr, const float* W, int ldw, const float* bias, float* C, long ldc, void* Cs,
                      long ldcs_bytes, int M, int N, int K, int act, int accum, int* range_flag);
/* the same product with nn.LayerNorm over the N = 512 output features taken in the epilogue (models/model_0v10.py:188,211,223: the three
 * norms of a decoder layer follow a GEMM + residual): C = LayerNorm(act(A W^T + bias) + residual) * gamma + beta (+ add[m % add_period],
 * e.g. `+ pos_embedding`), Cs = the split form of the normalised rows (without the added table).  A must be in the split form. */
BIGC_API int bigc_gemm_ps_ln(void* stream, const void* A, long lda_bytes, const float* W, int ldw, const float* bias, const float* residual, long ldr,
                    const float* gamma, const float* beta, float eps, const float* add, int add_period, float* C, long ldc, void* Cs,
                    long ldcs_bytes, int M, int N, int K, int act, int* range_flag);
/* f32 rows (rows, K) -> hr8 rows; K % 8 == 0 */
BIGC_API int bigc_split_rows_f32(void* stream, const float* X, long ldx, void* Y, long ldy_bytes, long rows, int K, int* range_flag);
/* act(A W^T + bias) + residual on the default split-operand GEMM (f32 A, as bigc_linear_f32) with the result written in the split
 * form "hr8" (Cs, pitch ldcs_bytes; N % 8 == 0) and / or as f32 (C may be NULL): the producer side of a GEMM -> GEMM chain whose
 * second GEMM is bigc_gemm_ps (models/model_0v10.py:446-448 fc_feat2enti.0 -> .2 -> conv, FFN pairs :113, :220).  K == 8 runs
 * the store-bound FMA kernel.  An element the split form cannot hold sets *range_flag. */
BIGC_API int bigc_linear_hr8out_f32(void* stream, const float* A, int lda, const float* W, int ldw, const float* bias, const float* residual, int ldr,
                           float* C, int ldc, void* Cs, long ldcs_bytes, int M, int N, int K, int act, int* range_flag);

#ifdef __cplusplus
}
#endif
#endif /* BIGC_HIP_H */
