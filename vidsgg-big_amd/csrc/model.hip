// BIG-C eval forward over one packed batch of videos: the composition of the kernels into the
// reference's op sequence (models/model_0v10.py:434-507 encode2decode + prediction_head,
// :707-785 construct_triplet; models/model_0v7.py for the VidOR variant).
//
// What differs from the reference by design (same results, different execution):
//   * ragged execution: the front-end MLPs run on the sum(L_i) packed rows; the stretch to Tmax
//     (models/model_0v10.py:427-428) exists only as the conv gather table and as weights of the
//     I3D/classeme mean;
//   * all videos of the batch go through each kernel together (the reference loops per video,
//     :376), attention kernels take per-video row offsets;
//   * everything is stream-ordered on the caller's HIP stream, no host synchronisation.
#include "model_internal.h"
#include "prof.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct Resolver {
    bigc_model* m;
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
    Lin lin(const std::string& p, int n_out, int n_in) {
        Lin l;
        l.w = get(p + ".weight", (int64_t)n_out * n_in);
        l.b = get(p + ".bias", n_out);
        return l;
    }
    LNp ln(const std::string& p, int d) {
        LNp l;
        l.g = get(p + ".weight", d);
        l.b = get(p + ".bias", d);
        return l;
    }
    Lin mha_in(const std::string& p, int d) {
        Lin l;
        l.w = get(p + ".in_proj_weight", (int64_t)3 * d * d);
        l.b = get(p + ".in_proj_bias", 3 * d);
        return l;
    }
};

struct Sizes { size_t R, NT, C, B, Q, BQ; };

// single source of truth for the workspace layout: run with a null base to measure
struct Work {
    float *bbox8, *hb, *hv, *enti, *conv, *pooled, *eh, *e2e, *x0, *x1, *qkv, *ao, *t1, *fh;
    float *query, *qk_in, *qk, *vv, *ao2, *t2, *t3, *p2a, *e2a, *vals, *rh, *enco;
    float *xmean, *i3dfc, *head_in, *bias_g, *hh;
    int32_t *conv_idx, *q_off;
    // split-operand path (forward_ps): hr8 twins of buffers that are needed in both forms, and the range flag
    float *xs, *x1s, *querys, *qkins, *encos, *ecat;
    int32_t* flag;
};

bool layout(const bigc_model* m, const Sizes& z, Arena& a, Work& w) {
    const bigc_config& c = m->cfg;
    const size_t de = c.dim_enti, dp = c.dim_pred, da = c.dim_att, df = c.dim_ffn;
    w.bbox8 = a.take<float>(z.R * 8);
    w.hb = a.take<float>(z.R * de);
    w.hv = a.take<float>(z.R * de);
    w.enti = a.take<float>(z.R * 2 * de);
    w.conv_idx = a.take<int32_t>(z.C * 3);
    w.conv = a.take<float>(z.C * de);
    w.pooled = a.take<float>(z.NT * de * c.enco_pool_len);
    w.eh = a.take<float>(z.NT * de);
    w.e2e = a.take<float>(z.NT * de);
    w.x0 = a.take<float>(z.NT * de);
    w.x1 = a.take<float>(z.NT * de);
    w.enco = a.take<float>(z.NT * de);
    w.qkv = a.take<float>(z.NT * 3 * de);
    w.ao = a.take<float>(z.NT * de);
    w.t1 = a.take<float>(z.NT * de);
    w.fh = a.take<float>(z.NT * df);
    w.q_off = a.take<int32_t>(z.B + 1);
    w.query = a.take<float>(z.BQ * dp);
    w.qk_in = a.take<float>(z.BQ * dp);
    w.qk = a.take<float>(z.BQ * 3 * dp);       // (B Q, 2 dp) q | k  -- or, in the one-GEMM form of the decoder, (B Q, 3 dp) q | k | v
    w.vv = w.qk + z.BQ * 2 * dp;               // (B Q, dp) v of the two-GEMM form: the tail of the same block
    w.ao2 = a.take<float>(z.BQ * dp);
    w.t2 = a.take<float>(z.BQ * dp);
    w.t3 = a.take<float>(z.BQ * dp);
    w.p2a = a.take<float>(z.BQ * da);
    w.e2a = a.take<float>(z.NT * da);
    w.vals = a.take<float>(2 * z.BQ * de);
    {   // also holds the two (NT, dp) projected entity blocks of the fused role path (forward: `ew`)
        const size_t a1 = z.BQ * (dp > df ? dp : df), a2 = 2 * z.NT * dp;
        w.rh = a.take<float>(a1 > a2 ? a1 : a2);
    }
    w.xmean = a.take<float>(z.NT * (c.dim_extra > 0 ? c.dim_extra : 1));
    w.i3dfc = a.take<float>(z.NT * de);
    {   // also holds U and V of the factorised head: 2 x (NT, ldu)
        const size_t a1 = z.BQ * (size_t)m->head_k, a2 = 2 * z.NT * (size_t)((c.num_pred_cats + 3) & ~3);
        w.head_in = a.take<float>(a1 > a2 ? a1 : a2);
    }
    w.bias_g = a.take<float>(z.BQ * c.num_pred_cats);
    w.hh = a.take<float>(z.BQ * df);
    w.xs = a.take<float>(z.NT * de);
    w.x1s = a.take<float>(z.NT * de);
    w.encos = a.take<float>(z.NT * de);
    w.querys = a.take<float>(z.BQ * dp);
    w.qkins = a.take<float>(z.BQ * dp);
    w.ecat = a.take<float>(z.NT * (da + 2 * dp) * (size_t)(c.n_deco_layers > 0 ? c.n_deco_layers : 1));     // per decoder layer [e2a | role0 projection | role1 projection] of the entity rows, ONE GEMM for all layers (build_enco_cat)
    w.flag = a.take<int32_t>(64);
    return !a.overflow;
}

int head_width(const bigc_config& c) {
    if (c.variant == 2) return 2 * c.dim_enti + (c.use_clsme ? 2 * c.dim_clsme : 0);      // Base_C: [s_clsme, o_clsme, s_feat, o_feat]
    if (c.variant == 0) return c.dim_pred + 2 * c.dim_clsme + (c.dim_extra > 0 ? 4 : 2) * c.dim_enti;
    return c.dim_pred + 2 * c.dim_enti + (c.use_clsme ? 2 * c.dim_clsme : 0);
}

Sizes sizes_of(const bigc_model* m, const bigc_batch* b) {
    Sizes z;
    z.R = b->n_rows; z.NT = b->n_trk; z.C = b->n_conv; z.B = b->n_vid; z.Q = m->cfg.num_querys; z.BQ = z.B * z.Q;
    return z;
}

// a1-a4: per-frame front-end on the packed rows, stride-2 temporal conv over the virtual stretch, pool, fc_enti2enco
// (models/model_0v10.py:401-420,446-458; shared by Base_C, models/model_pairwise_baseline.py:176-197)
int run_frontend(const bigc_model* m, void* s, const bigc_batch* b, const Work& w, float* e2e) {
    const bigc_config& c = m->cfg;
    const int R = b->n_rows, NT = b->n_trk, C = b->n_conv;
    const int de = c.dim_enti, D = c.dim_feat + c.dim_extra;
    TRY(bigc_bbox_feat_f32(s, b->boxes, b->trk_row_off, b->trk_vid, b->vid_wh, w.bbox8, NT));
    if (bigc_linear_pre8_supported(de, de)) {     // the 8 -> de layer evaluated inside the staging of the de -> de GEMM: hb is never written
        TRY(bigc_linear_pre8(s, w.bbox8, m->bbox0.w, m->bbox0.b, m->bbox2.w, de, m->bbox2.b, w.enti, 2 * de, R, de, de, BIGC_ACT_RELU));
    } else {
        TRY(linear(s, w.bbox8, 8, m->bbox0, w.hb, de, R, de, 8, BIGC_ACT_RELU));
        TRY(linear(s, w.hb, de, m->bbox2, w.enti, 2 * de, R, de, de, BIGC_ACT_RELU));
    }
    BIGC_CHECK_ARG(b->feats, "forward: this path reads the f32 features (bigc_batch.feats), which the batch was packed without");
    TRY(linear(s, b->feats, D, m->feat0, w.hv, de, R, de, c.dim_feat, BIGC_ACT_RELU));
    TRY(linear(s, w.hv, de, m->feat2, w.enti + de, 2 * de, R, de, de, BIGC_ACT_RELU));
    TRY(bigc_conv_index(s, b->trk_row_off, b->trk_conv_off, b->trk_vid, b->vid_tmax, w.conv_idx, NT));
    TRY(bigc_linear_f32(s, w.enti, 2 * de, w.conv_idx, 3, 2 * de, m->conv_w_tapmajor, 6 * de, m->conv.b, nullptr, 0, w.conv, de,
                        C, de, 6 * de, BIGC_ACT_NONE));
    TRY(bigc_adaptive_maxpool(s, w.conv, b->trk_conv_off, w.pooled, NT, de, c.enco_pool_len));
    TRY(linear(s, w.pooled, de * c.enco_pool_len, m->enco0, w.eh, de, NT, de, de * c.enco_pool_len, BIGC_ACT_RELU));
    TRY(linear(s, w.eh, de, m->enco2, e2e, de, NT, de, de, BIGC_ACT_RELU));
    return BIGC_OK;
}

}  // namespace

extern "C" int bigc_create(const bigc_config* cfg, bigc_model** out) {
    BIGC_CHECK_ARG(cfg && out, "create: null pointer");
    BIGC_CHECK_ARG(cfg->variant >= 0 && cfg->variant <= 2, "create: variant %d", cfg->variant);
    BIGC_CHECK_ARG(cfg->dim_enti % cfg->n_att_head == 0 && cfg->dim_pred % cfg->n_att_head == 0, "create: dims not divisible by heads");
    const int hd_e = cfg->dim_enti / cfg->n_att_head, hd_p = cfg->dim_pred / cfg->n_att_head;
    BIGC_CHECK_ARG((hd_e == 8 || hd_e == 16 || hd_e == 32 || hd_e == 64) && hd_e == hd_p, "create: head_dim %d/%d unsupported", hd_e, hd_p);
    BIGC_CHECK_ARG(cfg->dim_feat % 4 == 0 && cfg->dim_extra % 4 == 0 && cfg->dim_enti % 16 == 0 && cfg->dim_att % 16 == 0 &&
                   cfg->dim_clsme % 4 == 0 && cfg->dim_ffn % 4 == 0 && cfg->dim_pred % 4 == 0, "create: dims must be multiples of 4 (enti/att of 16)");
    BIGC_CHECK_ARG(cfg->num_pred_cats <= 256 && cfg->num_enti_cats <= 256, "create: at most 256 categories");
    BIGC_CHECK_ARG(cfg->dim_enti == cfg->dim_pred, "create: dim_enti must equal dim_pred (the reference adds role queries to pred queries)");
    bigc_model* m = new bigc_model();
    m->cfg = *cfg;
    m->head_k = head_width(*cfg);
    *out = m;
    return BIGC_OK;
}

extern "C" void bigc_destroy(bigc_model* m) {
    if (!m) return;
    // the split forms of this model's weights are cached per pointer (linear_split.hip / gemm_ps.hip): release them with the model
    for (const auto& kv : m->params) bigc_weight_cache_forget(kv.second.first);
    if (m->conv_w_tapmajor) bigc_weight_cache_forget(m->conv_w_tapmajor);
    if (m->role_cat) bigc_weight_cache_forget(m->role_cat);
    if (m->enco_cat) bigc_weight_cache_forget(m->enco_cat);
    if (m->conv_w_bwd) bigc_weight_cache_forget(m->conv_w_bwd);
    if (m->conv_w_tapmajor) (void)hipFree(m->conv_w_tapmajor);
    if (m->role_cat) (void)hipFree(m->role_cat);
    if (m->enco_cat) (void)hipFree(m->enco_cat);
    if (m->conv_w_bwd) (void)hipFree(m->conv_w_bwd);
    for (int i = 0; i < 3; ++i) if (m->dw_ev[i]) (void)hipEventDestroy((hipEvent_t)m->dw_ev[i]);
    if (m->dw_stream) (void)hipStreamDestroy((hipStream_t)m->dw_stream);
    if (m->qkpos) (void)hipFree(m->qkpos);
    delete m;
}

extern "C" int bigc_set_param(bigc_model* m, const char* key, const float* ptr, int64_t numel) {
    BIGC_CHECK_ARG(m && key && ptr && numel > 0, "set_param: bad argument");
    BIGC_CHECK_ARG(((uintptr_t)ptr & 15) == 0, "set_param: '%s' is not 16-byte aligned", key);
    auto it = m->params.find(key);
    if (it == m->params.end() || it->second.first != ptr) m->train_jobs.clear();    // recorded jobs may name the old buffer
    m->params[key] = std::make_pair(ptr, numel);
    m->finalized = false;
    return BIGC_OK;
}

extern "C" int bigc_set_mode(bigc_model* m, int exact, int fast) {
    BIGC_CHECK_ARG(m && exact >= -1 && exact <= 1 && fast >= -1 && fast <= 2, "set_mode: exact must be -1 (process default), 0 or 1; fast -1, 0, 1 (f16) or 2 (bf16)");
    m->mode_exact = exact; m->mode_fast = fast;
    return BIGC_OK;
}

extern "C" int bigc_finalize(bigc_model* m, void* stream) {
    BIGC_CHECK_ARG(m, "finalize: null model");
    const bigc_config& c = m->cfg;
    Resolver r{m};
    const int de = c.dim_enti, dp = c.dim_pred, da = c.dim_att, df = c.dim_ffn;
    if (c.has_enti_emb) m->emb = r.get("EntiNameEmb", (int64_t)c.num_enti_cats * c.dim_clsme);
    if (c.variant != 2) {
        m->pos = r.get("pos_embedding", (int64_t)c.num_querys * dp);
        m->qinit = r.get("pred_query_init", (int64_t)c.num_querys * dp);
    }
    m->bias_matrix = r.get("bias_matrix", (int64_t)c.num_enti_cats * c.num_enti_cats * c.num_pred_cats);
    m->feat0 = r.lin("fc_feat2enti.0", de, c.dim_feat);
    m->feat2 = r.lin("fc_feat2enti.2", de, de);
    m->bbox0 = r.lin("fc_bbox2enti.0", de, 8);
    m->bbox2 = r.lin("fc_bbox2enti.2", de, de);
    m->conv.w = r.get("conv_feat2enti.weight", (int64_t)de * 2 * de * 3);
    m->conv.b = r.get("conv_feat2enti.bias", de);
    m->enco0 = r.lin("fc_enti2enco.0", de, de * c.enco_pool_len);
    m->enco2 = r.lin("fc_enti2enco.2", de, de);
    if (c.variant == 0 && c.dim_extra > 0) m->i3d = r.lin("fc_i3d.0", de, c.dim_extra);
    m->enc.assign(c.n_enco_layers, EncLayer());
    for (int i = 0; i < c.n_enco_layers; ++i) {
        const std::string p = "encoder_layers." + std::to_string(i);
        EncLayer& L = m->enc[i];
        L.in_proj = r.mha_in(p + ".self_attn", de);
        L.out_proj = r.lin(p + ".self_attn.out_proj", de, de);
        L.l1 = r.lin(p + ".linear1", df, de);
        L.l2 = r.lin(p + ".linear2", de, df);
        L.n1 = r.ln(p + ".norm1", de);
        L.n2 = r.ln(p + ".norm2", de);
    }
    m->dec.assign(c.n_deco_layers, DecLayer());
    for (int i = 0; i < c.n_deco_layers; ++i) {
        const std::string p = "decoder_layers." + std::to_string(i);
        DecLayer& L = m->dec[i];
        L.in_proj = r.mha_in(p + ".self_attn", dp);
        L.out_proj = r.lin(p + ".self_attn.out_proj", dp, dp);
        L.role0[0] = r.lin(p + ".fc_rolewise.0.0", dp, de);
        L.role0[1] = r.lin(p + ".fc_rolewise.0.2", dp, dp);
        L.role1[0] = r.lin(p + ".fc_rolewise.1.0", dp, de);
        L.role1[1] = r.lin(p + ".fc_rolewise.1.2", dp, dp);
        L.e2a = r.lin(p + ".fc_enti2att", da, de);
        L.p2a = r.lin(p + ".fc_pred2att", da, dp);
        L.f0 = r.lin(p + ".fc2.0", df, dp);
        L.f3 = r.lin(p + ".fc2.3", dp, df);
        L.n1 = r.ln(p + ".norm1", dp);
        L.n2 = r.ln(p + ".norm2", dp);
        L.n3 = r.ln(p + ".norm3", dp);
    }
    if (c.variant == 0) {
        m->head0 = r.lin("fc_pred2logits", c.num_pred_cats, m->head_k);
    } else {
        m->head0 = r.lin("fc_pred2logits.0", df, m->head_k);
        m->head2 = r.lin("fc_pred2logits.2", c.num_pred_cats, df);
    }
    if (r.err) return r.err;
    if (!m->conv_w_tapmajor) {
        hipError_t e = hipMalloc((void**)&m->conv_w_tapmajor, sizeof(float) * (size_t)de * 2 * de * 3);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "finalize: hipMalloc: %s", hipGetErrorString(e));
    }
    TRY(bigc_repack_conv_weight((hipStream_t)stream, m->conv.w, m->conv_w_tapmajor, de, 2 * de, 3));
    // the backward's repack of the conv weight is a derived buffer whose split form is cached per epoch like any weight's: it
    // must be current BEFORE the step's batched re-split (bigc_linear_prepare in *_train_forward), i.e. here, not in the backward
    if (m->conv_w_bwd) TRY(bigc_repack_conv_weight_bwd((hipStream_t)stream, m->conv.w, m->conv_w_bwd, de, 2 * de, 3));
    m->epoch = bigc_next_weights_epoch();
    m->finalized = true;
    return BIGC_OK;
}

extern "C" size_t bigc_workspace_bytes(const bigc_model* m, const bigc_batch* b) {
    if (!m || !b) return 0;
    Arena a{nullptr, (size_t)-1};
    Work w;
    layout(m, sizes_of(m, b), a, w);
    return a.off + 256;
}

// The second role-wise linears of a decoder layer are summed (query + role0 + role1, models/model_0v10.py:209-211): one GEMM over
// the concatenated inputs with the weights side by side and the biases added.  Built once per weights epoch by the eval forward.
static int build_role_cat(bigc_model* m, void* stream) {
    const bigc_config& c = m->cfg;
    const int dp = c.dim_pred;
    if (m->role_cat_epoch == m->epoch || c.n_deco_layers == 0) return BIGC_OK;
    const size_t per = (size_t)dp * 2 * dp + dp;
    if (!m->role_cat) {
        hipError_t e = hipMalloc((void**)&m->role_cat, sizeof(float) * per * c.n_deco_layers);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "forward: hipMalloc: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < c.n_deco_layers; ++i) {
        const DecLayer& L = m->dec[i];
        float* wc = m->role_cat + per * i;
        float* bc = wc + (size_t)dp * 2 * dp;
        TRY(bigc_copy_cols((hipStream_t)stream, L.role0[1].w, dp, wc, 2 * dp, dp, dp));
        TRY(bigc_copy_cols((hipStream_t)stream, L.role1[1].w, dp, wc + dp, 2 * dp, dp, dp));
        TRY(bigc_copy_cols((hipStream_t)stream, L.role0[1].b, dp, bc, dp, 1, dp));
        TRY(bigc_axpy_f32(stream, bc, L.role1[1].b, dp));
    }
    m->role_cat_epoch = m->epoch;
    return BIGC_OK;
}

// The three products of a decoder layer that read `enco_output` -- fc_enti2att and the first linear of each role-wise MLP
// (models/model_0v10.py:193-208, the latter applied to the entity rows: see the role kernel) -- as ONE GEMM with N = da + 2 dp over weights stored
// one under the other (built once per weights epoch): one launch instead of three, the operand split once for six column tiles instead of three
// times for two.  Every output element is the same accumulation as before (bit-identical).
static int build_enco_cat(bigc_model* m, void* stream) {
    const bigc_config& c = m->cfg;
    const int dp = c.dim_pred, da = c.dim_att, de = c.dim_enti, W = da + 2 * dp, NL = c.n_deco_layers;
    if (m->enco_cat_epoch == m->epoch || NL == 0) return BIGC_OK;
    // (NL W, de) weights -- layer i's [W_e2a; W_role0.0; W_role1.0] are rows i W .. (i + 1) W - 1 -- then the (NL W) biases [b_e2a | 0 | 0] per layer
    const size_t total = (size_t)NL * W * de + (size_t)NL * W;
    if (!m->enco_cat) {
        hipError_t e = hipMalloc((void**)&m->enco_cat, sizeof(float) * total);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "forward: hipMalloc: %s", hipGetErrorString(e));
    }
    if (hipMemsetAsync(m->enco_cat, 0, sizeof(float) * total, (hipStream_t)stream) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "forward: memset failed");
    float* bias = m->enco_cat + (size_t)NL * W * de;
    for (int i = 0; i < NL; ++i) {
        const DecLayer& L = m->dec[i];
        float* wc = m->enco_cat + (size_t)i * W * de;
        TRY(bigc_copy_cols((hipStream_t)stream, L.e2a.w, de, wc, de, da, de));
        TRY(bigc_copy_cols((hipStream_t)stream, L.role0[0].w, de, wc + (size_t)da * de, de, dp, de));
        TRY(bigc_copy_cols((hipStream_t)stream, L.role1[0].w, de, wc + (size_t)(da + dp) * de, de, dp, de));
        TRY(bigc_copy_cols((hipStream_t)stream, L.e2a.b, da, bias + (size_t)i * W, da, 1, da));
    }
    m->enco_cat_epoch = m->epoch;
    return BIGC_OK;
}

// in_proj of a decoder layer's self-attention reads q = k = query + pos and v = query (models/model_0v10.py:181-183).  W (query + pos) =
// W query + W pos, and W pos is the same (num_querys, 2 dp) table for every video: built once per weights epoch as (num_querys, 3 dp)
// rows [pos Wq^T | pos Wk^T | 0], it enters the ONE q / k / v GEMM of the layer as a residual that repeats per video
// (PsGemm.res_period) -- `query + pos` is never materialised and the GEMM reads its operand once instead of twice.
static int build_qkpos(bigc_model* m, void* stream) {
    const bigc_config& c = m->cfg;
    const int dp = c.dim_pred, Q = c.num_querys;
    if (m->qkpos_epoch == m->epoch || c.n_deco_layers == 0) return BIGC_OK;
    const size_t per = (size_t)Q * 3 * dp;
    if (!m->qkpos) {
        hipError_t e = hipMalloc((void**)&m->qkpos, sizeof(float) * per * c.n_deco_layers);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "forward: hipMalloc: %s", hipGetErrorString(e));
    }
    if (hipMemsetAsync(m->qkpos, 0, sizeof(float) * per * c.n_deco_layers, (hipStream_t)stream) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "forward: memset failed");
    StaticWeights once(0);                       // a one-off product per epoch: nothing to cache
    for (int i = 0; i < c.n_deco_layers; ++i)
        TRY(bigc_linear_f32(stream, m->pos, dp, nullptr, 1, dp, m->dec[i].in_proj.w, dp, nullptr, nullptr, 0, m->qkpos + per * i, 3 * dp, Q, 2 * dp, dp,
                            BIGC_ACT_NONE));
    m->qkpos_epoch = m->epoch;
    return BIGC_OK;
}

// ------------------------------------------------------------------------------------------------
// Eval forward on the split-operand GEMM (csrc/gemm_ps.hip).  Same op sequence as bigc_forward below; what changes is the
// FORM in which a GEMM's output is written: when its only consumer is another GEMM it is stored as f16 head / residual pairs
// ("hr8", same footprint as f32) and read by DMA; tensors that LayerNorm, attention or the head also read stay f32 (and get
// an hr8 twin from the producing GEMM's epilogue when a GEMM reads them too).  f32 tensors produced by non-GEMM kernels
// (LayerNorm, attention, max-pool, the raw features) enter a GEMM through its in-loop split (A mode 2).
// ------------------------------------------------------------------------------------------------
int bigc_mha_core_hr8_launch(hipStream_t s, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, void* os, long ldos,
                             const int32_t* q_off, int n_seq, int max_q, int n_head, int hd, int* range_flag,
                             const float* qpos = nullptr, const float* kpos = nullptr, int ldpos = 0);
int bigc_role_fused_proj_hr8_launch(hipStream_t s, const float* p2a, int ldp, const float* e2a, int lde, const float* rhs, int ldrhs, long role_stride,
                                    const float* vbias0, const float* vbias1, int relu, float* att, void* vals_s, long ldvs, long out_role_stride,
                                    const int32_t* e_off, int n_vid, int nq, int max_trk, int half, int d, float scale, int* range_flag);
int bigc_layernorm_hr8(hipStream_t hs, const float* x, const float* gamma, const float* beta, float* y, void* ys, long ldys, int rows, int d,
                       float eps, int* range_flag);
int bigc_layernorm_res(hipStream_t hs, const float* x, const float* pre, const float* gamma, const float* beta, float* y, void* ys, long ldys,
                       int rows, int d, float eps, const float* add, int period, int* range_flag);

namespace {

struct PsCtx { hipStream_t s; int* flag; bool decoder = false; };

__global__ void or_flag_kernel(int* dst, const int32_t* src) { if (*src) atomicOr(dst, 1); }
int bigc_or_flag(hipStream_t s, int* dst, const int32_t* src) {
    hipLaunchKernelGGL(or_flag_kernel, dim3(1), dim3(1), 0, s, dst, src);
    BIGC_LAUNCH_CHECK("or_flag");
    return BIGC_OK;
}

// Accumulation policy of the eval path's GEMMs on pre-split operands (PsGemm.accum, csrc/gemm_ps.hip).  Default "blocked": every such GEMM
// sums K in blocks of 128 folded into a running total -- over the 256 benchmark videos this is what brings every video's logits within 1e-4
// of a float64 evaluation (max 7.8e-5; one accumulator 1.65e-4: benchmarks/accuracy_sweep.py) -- at the price of 128-row tiles.
// BIGC_ACCUM = plain | k1024 (K >= 1024 only) | front (M >= 100 000 rows only) | nodecoder (all but the decoder's) | blocked.
int ps_accum(long M, int K, bool decoder = false) {
    const int mode = bigc_get_accum_policy();                  // bigc_set_accum_policy / BIGC_ACCUM (csrc/linear.hip)
    if (K % 128 != 0) return 0;
    switch (mode) {
        case 0: return 0;
        case 1: return K >= 1024 ? 2 : 0;
        case 2: return M >= 100000 ? 2 : 0;
        case 3: return decoder ? 0 : 2;
        default: return 2;
    }
}

// C / Cs = act(A W^T + b) + res;  a_hr8: A is in split form.  Pitches in floats (the hr8 form has the same pitch in bytes / 4).
int ps(const PsCtx& c, const void* A, long lda, bool a_hr8, const Lin& l, int ldw, float* C, long ldc, void* Cs, long ldcs, int M, int N, int K, int act,
       const float* res = nullptr, long ldr = 0, const int32_t* a_idx = nullptr, int nseg = 1, int res_period = 0) {
    PsGemm g;
    g.res_period = res_period;
    g.A = A; g.lda = lda * 4; g.a_fmt = a_hr8 ? PS_HR8 : PS_F32; g.a_idx = a_idx; g.nseg = nseg; g.W = l.w; g.ldw = ldw; g.bias = l.b;
    g.res = res; g.ldr = ldr; g.C = C; g.ldc = ldc; g.Cs = Cs; g.ldcs = ldcs * 4; g.M = M; g.N = N; g.K = K; g.act = act; g.range_flag = c.flag;
    g.accum = ps_accum(M, K, c.decoder);
    return bigc_gemm_ps_launch(c.s, g);
}

// C = LayerNorm(A W^T + b + res) g + beta (+ add[m % period]); Cs = the split twin of the normalised rows: the GEMM takes the row's LayerNorm in
// its epilogue (gemm_ps.hip, N = 512).  A is in the split form.
int ps_ln(const PsCtx& c, const void* A, long lda, const Lin& l, int ldw, float* C, long ldc, void* Cs, long ldcs, int M, int N, int K, const float* res, long ldr,
          const LNp& n, float eps, const float* add = nullptr, int add_period = 0) {
    PsGemm g;
    g.A = A; g.lda = lda * 4; g.a_fmt = PS_HR8; g.W = l.w; g.ldw = ldw; g.bias = l.b; g.res = res; g.ldr = ldr; g.C = C; g.ldc = ldc; g.Cs = Cs; g.ldcs = ldcs * 4;
    g.M = M; g.N = N; g.K = K; g.act = BIGC_ACT_NONE; g.range_flag = c.flag;
    g.ln_g = n.g; g.ln_b = n.b; g.ln_eps = eps; g.ln_add = add; g.ln_add_period = add_period;
    return bigc_gemm_ps_launch(c.s, g);
}

bool ps_config_ok(const bigc_config& c) {
    // every GEMM of the path: K a multiple of 32, N a multiple of 8 (the logits GEMM, N = num_pred_cats, stays on linear_split)
    const int dims[] = {c.dim_enti, c.dim_pred, c.dim_att, c.dim_ffn, c.dim_feat};
    for (int d : dims) if (d % 32 != 0) return false;
    return (c.dim_enti * c.enco_pool_len) % 32 == 0 && (c.dim_att / 2) % 8 == 0;
}

// Default eval path, GEMM -> GEMM chains ("chained" mode): where a GEMM's output only feeds another GEMM, the first one
// (linear_split_kernel: f32 operand, split in the loop) writes it in the split form and the second one runs on gemm_ps.hip, whose
// K loop is DMA + MFMA only.  Measured in the model (rocprofv3, 256-video step): conv GEMM 4.65 -> 3.79 ms, fc_feat2enti.2
// 1.09 -> 0.82 ms.  GEMMs whose operand comes from a LayerNorm, an attention kernel or the input features stay on linear_split
// (gemm_ps with an f32 operand is slower than it).  hb / hv / enti / eh / fh / the decoder's FFN hidden are hr8 in this mode.
int run_frontend_chained(const bigc_model* m, void* s, const bigc_batch* b, const Work& w, float* e2e) {
    const bigc_config& c = m->cfg;
    const int R = b->n_rows, NT = b->n_trk, C = b->n_conv;
    const int de = c.dim_enti, D = c.dim_feat + c.dim_extra;
    PsCtx cx{(hipStream_t)s, w.flag};
    TRY(bigc_bbox_feat_f32(s, b->boxes, b->trk_row_off, b->trk_vid, b->vid_wh, w.bbox8, NT));
    TRY(bigc_linear_k8_hr8out(s, w.bbox8, m->bbox0.w, m->bbox0.b, w.hb, de, R, de, BIGC_ACT_RELU, w.flag));
    TRY(ps(cx, w.hb, de, true, m->bbox2, de, nullptr, 0, w.enti, 2 * de, R, de, de, BIGC_ACT_RELU));
    if (b->feats_hr8 && b->ld_feats_hr8 % 16 == 0 && c.dim_feat % 32 == 0) {
        // the features were split when the batch was packed (bigc_batch.feats_hr8): the K = dim_feat GEMM reads them by DMA
        TRY(ps(cx, b->feats_hr8, b->ld_feats_hr8 / 4, true, m->feat0, c.dim_feat, nullptr, 0, w.hv, de, R, de, c.dim_feat, BIGC_ACT_RELU));
        if (b->feats_hr8_flag) TRY(bigc_or_flag((hipStream_t)s, w.flag, b->feats_hr8_flag));
    } else if (!b->feats && b->feat_row_ptr && c.dim_feat % 4 == 0) {
        // the visual features read IN PLACE from the callers' per-tracklet rows (bigc_batch.feat_row_ptr): no packed copy of them exists.
        // Round 5: the raw f32 rows go HBM -> LDS by DMA through the address table and are split in LDS by the wave that fetched them
        // (gemm_ps.hip, A mode 3; two accumulators as the in-loop-split kernel it replaces); BIGC_FEAT0_SPLITLOOP=1: that kernel
        // (linear_split_kernel with a_rowptr: loads to registers, split, ds_write)
        // Which of the two wins depends on the accumulation policy (measured, 256 videos, same box): with plain accumulation the DMA kernel runs
        // 256-row tiles, 3.67 against 3.95 ms; with blocked accumulation it is bound to 128-row tiles, where the in-LDS split costs issue slots
        // beside half as many MFMAs per slab: 4.24 ms against the in-loop-split kernel's 3.95 (two accumulators: 5.0 units of 2^-24 at K = 2048
        // against 2.9 blocked and 8.5 plain; the whole step stays within 1e-4 of float64 on every benchmark video either way: 7.8e-5 / 6.7e-5).
        // Default: the faster one for the policy in effect; BIGC_FEAT0_SPLITLOOP = 1 / 0 forces the in-loop-split / the DMA kernel.
        static const int split_env = getenv("BIGC_FEAT0_SPLITLOOP") ? atoi(getenv("BIGC_FEAT0_SPLITLOOP")) : -1;
        const bool split_loop = split_env >= 0 ? split_env != 0 : ps_accum(R, c.dim_feat) != 0;
        PsGemm g;
        g.a_rowptr = (const unsigned long long*)b->feat_row_ptr; g.a_fmt = PS_F32; g.W = m->feat0.w; g.ldw = c.dim_feat; g.bias = m->feat0.b;
        g.Cs = w.hv; g.ldcs = (long)de * 4; g.M = R; g.N = de; g.K = c.dim_feat; g.act = BIGC_ACT_RELU; g.range_flag = w.flag; g.accum = ps_accum(R, c.dim_feat);
        if (!split_loop && c.dim_feat % 32 == 0 && bigc_gemm_ps_supported(g) && bigc_linear_get_fast() != 2) TRY(bigc_gemm_ps_launch((hipStream_t)s, g));
        else TRY(bigc_linear_hr8out_rows(s, b->feat_row_ptr, m->feat0.w, c.dim_feat, m->feat0.b, w.hv, de, R, de, c.dim_feat, BIGC_ACT_RELU, w.flag));
    } else {
        BIGC_CHECK_ARG(b->feats, "forward: batch without f32 features and without a usable split form of them");
        TRY(bigc_linear_hr8out(s, b->feats, D, m->feat0.w, c.dim_feat, m->feat0.b, nullptr, 0, nullptr, 0, w.hv, de, R, de, c.dim_feat, BIGC_ACT_RELU, w.flag));
    }
    TRY(ps(cx, w.hv, de, true, m->feat2, de, nullptr, 0, w.enti + de, 2 * de, R, de, de, BIGC_ACT_RELU));
    TRY(bigc_conv_index(s, b->trk_row_off, b->trk_conv_off, b->trk_vid, b->vid_tmax, w.conv_idx, NT));
    {
        Lin cw{m->conv_w_tapmajor, m->conv.b};
        TRY(ps(cx, w.enti, 2 * de, true, cw, 6 * de, w.conv, de, nullptr, 0, C, de, 6 * de, BIGC_ACT_NONE, nullptr, 0, w.conv_idx, 3));
    }
    TRY(bigc_adaptive_maxpool(s, w.conv, b->trk_conv_off, w.pooled, NT, de, c.enco_pool_len));
    TRY(bigc_linear_hr8out(s, w.pooled, de * c.enco_pool_len, m->enco0.w, de * c.enco_pool_len, m->enco0.b, nullptr, 0, nullptr, 0, w.eh, de, NT, de,
                           de * c.enco_pool_len, BIGC_ACT_RELU, w.flag));
    TRY(ps(cx, w.eh, de, true, m->enco2, de, e2e, de, nullptr, 0, NT, de, de, BIGC_ACT_RELU));
    return BIGC_OK;
}

int forward_ps(bigc_model* m, void* stream, const bigc_batch* b, const Work& w, int topk, const bigc_outputs* out) {
    const bigc_config& c = m->cfg;
    hipStream_t hs = (hipStream_t)stream;
    void* s = stream;
    const int R = b->n_rows, NT = b->n_trk, B = b->n_vid, Q = c.num_querys, BQ = B * Q, C = b->n_conv;
    const int de = c.dim_enti, dp = c.dim_pred, da = c.dim_att, df = c.dim_ffn;
    const int D = c.dim_feat + c.dim_extra;
    const int H = c.n_att_head, hd = de / H;
    const float eps = 1e-5f;
    PsCtx cx{hs, w.flag};
    if (hipMemsetAsync(w.flag, 0, sizeof(int32_t), hs) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "forward: memset failed");

    // ---- a1-a4: front-end.  hb / hv / enti / eh only feed GEMMs: hr8 only.
    float* e2e = out->enti2enco ? out->enti2enco : w.e2e;
    TRY(bigc_bbox_feat_f32(s, b->boxes, b->trk_row_off, b->trk_vid, b->vid_wh, w.bbox8, NT));
    TRY(linear(s, w.bbox8, 8, m->bbox0, w.hb, de, R, de, 8, BIGC_ACT_RELU));                                         // K = 8: the store-bound FMA kernel, f32
    TRY(ps(cx, w.hb, de, false, m->bbox2, de, nullptr, 0, w.enti, 2 * de, R, de, de, BIGC_ACT_RELU));
    BIGC_CHECK_ARG(b->feats, "forward: BIGC_PS=1 reads the f32 features (bigc_batch.feats), which the batch was packed without");
    TRY(ps(cx, b->feats, D, false, m->feat0, c.dim_feat, nullptr, 0, w.hv, de, R, de, c.dim_feat, BIGC_ACT_RELU));
    TRY(ps(cx, w.hv, de, true, m->feat2, de, nullptr, 0, w.enti + de, 2 * de, R, de, de, BIGC_ACT_RELU));
    TRY(bigc_conv_index(s, b->trk_row_off, b->trk_conv_off, b->trk_vid, b->vid_tmax, w.conv_idx, NT));
    {
        Lin cw{m->conv_w_tapmajor, m->conv.b};
        TRY(ps(cx, w.enti, 2 * de, true, cw, 6 * de, w.conv, de, nullptr, 0, C, de, 6 * de, BIGC_ACT_NONE, nullptr, 0, w.conv_idx, 3));
    }
    TRY(bigc_adaptive_maxpool(s, w.conv, b->trk_conv_off, w.pooled, NT, de, c.enco_pool_len));
    TRY(ps(cx, w.pooled, de * c.enco_pool_len, false, m->enco0, de * c.enco_pool_len, nullptr, 0, w.eh, de, NT, de, de * c.enco_pool_len, BIGC_ACT_RELU));
    TRY(ps(cx, w.eh, de, true, m->enco2, de, e2e, de, w.xs, de, NT, de, de, BIGC_ACT_RELU));

    // ---- a5: encoder (x: f32 for the residual, xs: its hr8 twin for the GEMMs)
    const float* x = e2e;
    const void* xs = w.xs;
    for (int i = 0; i < c.n_enco_layers; ++i) {
        const EncLayer& L = m->enc[i];
        TRY(ps(cx, xs, de, true, L.in_proj, de, w.qkv, 3 * de, nullptr, 0, NT, 3 * de, de, BIGC_ACT_NONE));
        bigc_prof_hint(4.0 * NT * ((double)NT / B) * de, 16.0 * NT * (double)de);      // encoder: B sequences of ~NT / B tracklets
        TRY(bigc_mha_core_f32(s, w.qkv, 3 * de, w.qkv + de, 3 * de, w.qkv + 2 * de, 3 * de, w.ao, de, b->vid_trk_off, b->vid_trk_off,
                              B, b->max_trk, H, hd, nullptr));
        TRY(ps(cx, w.ao, de, false, L.out_proj, de, w.t1, de, nullptr, 0, NT, de, de, BIGC_ACT_NONE, x, de));
        TRY(bigc_layernorm_f32(s, w.t1, L.n1.g, L.n1.b, w.x1, NT, de, eps));
        TRY(ps(cx, w.x1, de, false, L.l1, de, nullptr, 0, w.fh, df, NT, df, de, BIGC_ACT_RELU));
        TRY(ps(cx, w.fh, df, true, L.l2, df, w.t1, de, nullptr, 0, NT, de, df, BIGC_ACT_NONE, w.x1, de));
        float* xo = (i == c.n_enco_layers - 1) ? (out->enco_output ? out->enco_output : w.enco) : w.x0;
        TRY(bigc_layernorm_f32(s, w.t1, L.n2.g, L.n2.b, xo, NT, de, eps));
        x = xo;
        xs = nullptr;                                                  // LayerNorm output: f32 only
        if (i + 1 < c.n_enco_layers) { TRY(bigc_split_rows_hr8(hs, xo, de, w.xs, (long)de * 4, NT, de, w.flag)); xs = w.xs; }
    }
    const float* enco = x;
    // enco_output feeds 3 GEMMs per decoder layer: one conversion to the split form for all of them
    TRY(bigc_split_rows_hr8(hs, enco, de, w.encos, (long)de * 4, NT, de, w.flag));

    // ---- a6: role-aware decoder
    TRY(bigc_fill_offsets(hs, w.q_off, B + 1, Q));
    if (c.n_deco_layers == 0) TRY(bigc_add_rows_f32(s, nullptr, m->qinit, w.query, BQ, dp, Q));
    const float att_scale = 1.0f / sqrtf((float)de);
    cx.decoder = true;
    for (int i = 0; i < c.n_deco_layers; ++i) {
        const DecLayer& L = m->dec[i];
        Lin qkp{L.in_proj.w, L.in_proj.b};
        Lin vp{L.in_proj.w + (size_t)2 * dp * dp, L.in_proj.b + 2 * dp};
        const int rows = (i == 0) ? Q : BQ, nseq = (i == 0) ? 1 : B;
        const float* qin = (i == 0) ? m->qinit : w.query;
        if (i == 0) TRY(bigc_add_rows_f32(s, qin, m->pos, w.qk_in, rows, dp, Q));
        TRY(ps(cx, w.qk_in, dp, false, qkp, dp, w.qk, 2 * dp, nullptr, 0, rows, 2 * dp, dp, BIGC_ACT_NONE));
        TRY(ps(cx, qin, dp, false, vp, dp, w.vv, dp, nullptr, 0, rows, dp, dp, BIGC_ACT_NONE));
        TRY(bigc_mha_core_f32(s, w.qk, 2 * dp, w.qk + dp, 2 * dp, w.vv, dp, w.ao2, dp, w.q_off, w.q_off, nseq, Q, H, dp / H, nullptr));
        TRY(ps(cx, w.ao2, dp, false, L.out_proj, dp, w.t2, dp, nullptr, 0, rows, dp, dp, BIGC_ACT_NONE, qin, dp));
        if (i == 0) {
            TRY(bigc_layernorm_add_f32(s, w.t2, L.n1.g, L.n1.b, w.qk_in, Q, dp, eps, m->pos, Q, nullptr));
            TRY(bigc_add_rows_f32(s, nullptr, w.qk_in, w.query, BQ, dp, Q));
        } else {
            TRY(bigc_layernorm_add_f32(s, w.t2, L.n1.g, L.n1.b, w.query, BQ, dp, eps, m->pos, Q, nullptr));   // query += pos, kept (:189)
        }
        TRY(ps(cx, w.encos, de, true, L.e2a, de, w.e2a, da, nullptr, 0, NT, da, de, BIGC_ACT_NONE));
        TRY(ps(cx, w.query, dp, false, L.p2a, dp, w.p2a, da, nullptr, 0, BQ, da, dp, BIGC_ACT_NONE));
        const bool fused = b->max_trk <= 64 && (da / 2) % 64 == 0;
        const bool last = i + 1 == c.n_deco_layers;
        if (fused && Q <= 192 && da / 2 <= 256 && dp % 32 == 0 && dp <= de) {
            float* ew = w.rh;
            Lin r0{L.role0[0].w, nullptr}, r1{L.role1[0].w, nullptr};
            TRY(ps(cx, w.encos, de, true, r0, de, ew, dp, nullptr, 0, NT, dp, de, BIGC_ACT_NONE));
            TRY(ps(cx, w.encos, de, true, r1, de, ew + (size_t)NT * dp, dp, nullptr, 0, NT, dp, de, BIGC_ACT_NONE));
            TRY(build_role_cat(m, s));
            TRY(bigc_role_fused_proj_f32(s, w.p2a, da, w.e2a, da, ew, dp, (long)NT * dp, L.role0[0].b, L.role1[0].b, 1, last ? out->att : nullptr,
                                         w.vals, dp, 2 * dp, b->vid_trk_off, B, Q, b->max_trk, da / 2, dp, att_scale));
            const float* wc = m->role_cat + ((size_t)dp * 2 * dp + dp) * i;
            Lin rc{wc, wc + (size_t)dp * 2 * dp};
            TRY(ps(cx, w.vals, 2 * dp, false, rc, 2 * dp, w.t3, dp, nullptr, 0, BQ, dp, 2 * dp, BIGC_ACT_NONE, w.query, dp));
        } else {
            if (fused) {
                TRY(bigc_role_fused_f32(s, w.p2a, da, w.e2a, da, enco, de, last ? out->att : nullptr, w.vals, b->vid_trk_off, B, Q,
                                        b->max_trk, da / 2, de, att_scale));
            } else {
                TRY(bigc_role_attention_f32(s, w.p2a, da, w.e2a, da, out->att, b->vid_trk_off, B, Q, da / 2, att_scale));
                TRY(bigc_role_values_f32(s, out->att, enco, de, w.vals, b->vid_trk_off, B, Q, de));
            }
            // rh (hr8) only feeds the second role-wise linear
            TRY(ps(cx, w.vals, de, false, L.role0[0], de, nullptr, 0, w.rh, dp, BQ, dp, de, BIGC_ACT_RELU));
            TRY(ps(cx, w.rh, dp, true, L.role0[1], dp, w.t2, dp, nullptr, 0, BQ, dp, dp, BIGC_ACT_NONE, w.query, dp));
            TRY(ps(cx, w.vals + (size_t)BQ * de, de, false, L.role1[0], de, nullptr, 0, w.rh, dp, BQ, dp, de, BIGC_ACT_RELU));
            TRY(ps(cx, w.rh, dp, true, L.role1[1], dp, w.t3, dp, nullptr, 0, BQ, dp, dp, BIGC_ACT_NONE, w.t2, dp));
        }
        TRY(bigc_layernorm_f32(s, w.t3, L.n2.g, L.n2.b, w.t2, BQ, dp, eps));
        TRY(ps(cx, w.t2, dp, false, L.f0, dp, nullptr, 0, w.rh, df, BQ, df, dp, BIGC_ACT_RELU));
        TRY(ps(cx, w.rh, df, true, L.f3, df, w.t3, dp, nullptr, 0, BQ, dp, df, BIGC_ACT_NONE, w.t2, dp));
        if (i + 1 < c.n_deco_layers) TRY(bigc_layernorm_add_f32(s, w.t3, L.n3.g, L.n3.b, w.qk_in, BQ, dp, eps, m->pos, Q, w.query));
        else TRY(bigc_layernorm_f32(s, w.t3, L.n3.g, L.n3.b, w.query, BQ, dp, eps));
    }
    if (out->pred_query) {
        hipError_t e = hipMemcpyAsync(out->pred_query, w.query, sizeof(float) * (size_t)BQ * dp, hipMemcpyDeviceToDevice, hs);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "forward: memcpy: %s", hipGetErrorString(e));
    }

    // ---- a7: prediction head.  Narrow / odd-K GEMMs (N = num_pred_cats, K = 300 / 832) stay on the linear_split kernel.
    const float* seg_src[4];
    int32_t seg_w[4], seg_cat[4];
    int n_seg = 0;
    if (c.variant == 0) {
        if (c.dim_extra > 0) {
            TRY(bigc_extra_mean(s, b, c, w.xmean, NT));
            TRY(linear(s, w.xmean, c.dim_extra, m->i3d, w.i3dfc, de, NT, de, c.dim_extra, BIGC_ACT_RELU));
            seg_src[n_seg] = w.i3dfc; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
            seg_src[n_seg] = e2e; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
            seg_src[n_seg] = m->emb; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 1;
        } else {
            seg_src[n_seg] = m->emb; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 1;
            seg_src[n_seg] = e2e; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
        }
    } else {
        if (c.use_clsme) {
            if (c.has_enti_emb) {
                seg_src[n_seg] = m->emb; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 1;
            } else {
                BIGC_CHECK_ARG(c.dim_extra == c.dim_clsme, "forward: classeme width %d != dim_clsme %d", c.dim_extra, c.dim_clsme);
                TRY(bigc_extra_mean(s, b, c, w.xmean, NT));
                seg_src[n_seg] = w.xmean; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 0;
            }
        }
        seg_src[n_seg] = e2e; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
    }
    if (c.variant == 0) {
        const int P = c.num_pred_cats, hk = m->head_k, ldu = (P + 3) & ~3;
        float* U = w.head_in;
        float* V = U + (size_t)NT * ldu;
        int col = dp;
        for (int g = 0; g < n_seg; ++g) {
            const float* Wg = m->head0.w + col;
            for (int role = 0; role < 2; ++role) {
                float* dst = role == 0 ? U : V;
                TRY(bigc_linear_f32(s, seg_src[g], seg_w[g], seg_cat[g] ? b->trk_cat : nullptr, 1, seg_w[g], Wg + role * seg_w[g], hk, nullptr,
                                    g ? dst : nullptr, ldu, dst, ldu, NT, P, seg_w[g], BIGC_ACT_NONE));
            }
            col += 2 * seg_w[g];
        }
        TRY(bigc_head_pick(hs, out->att, b->vid_trk_off, b->trk_cat, B, Q, U, V, ldu, m->bias_matrix, c.num_enti_cats, P, out->soid, w.bias_g));
        TRY(bigc_linear_f32(s, w.query, dp, nullptr, 1, dp, m->head0.w, hk, m->head0.b, w.bias_g, P, out->pred_logits, P, BQ, P, dp, BIGC_ACT_NONE));
    } else {
        TRY(bigc_head_gather(s, out->att, b->vid_trk_off, b->trk_cat, B, Q, w.query, dp, n_seg, seg_src, seg_w, seg_cat, m->bias_matrix,
                             c.num_enti_cats, c.num_pred_cats, out->soid, w.head_in, m->head_k, w.bias_g));
        TRY(linear(s, w.head_in, m->head_k, m->head0, w.hh, df, BQ, df, m->head_k, BIGC_ACT_RELU));
        TRY(linear(s, w.hh, df, m->head2, out->pred_logits, c.num_pred_cats, BQ, c.num_pred_cats, df, BIGC_ACT_NONE, w.bias_g,
                   c.num_pred_cats));
    }
    if (out->quint) {
        BIGC_CHECK_ARG(out->scores && out->dura && out->qid && out->n_out, "forward: all construct_triplet outputs or none");
        TRY(bigc_construct_triplets(s, out->pred_logits, c.num_pred_cats, out->soid, b->vid_trk_off, b->trk_dura, b->trk_cat, b->trk_score,
                                    B, Q, topk, out->quint, out->scores, out->dura, out->qid, out->n_out));
    }
    if (hipMemcpyAsync(out->range_flag, w.flag, sizeof(int32_t), hipMemcpyDeviceToDevice, hs) != hipSuccess)
        return bigc_fail(BIGC_ERR_HIP, "forward: memcpy failed");
    return BIGC_OK;
}

}  // namespace

// mirrors the path choice of bigc_forward below (chained mode) and of run_frontend_chained (the split features are usable)
extern "C" int bigc_forward_reads_hr8(const bigc_model* m) {
    if (!m || m->cfg.variant == 2) return 0;
    ModeScope mode_scope(m->mode_exact, m->mode_fast);
    static const bool ps_on = getenv("BIGC_PS") && atoi(getenv("BIGC_PS")) != 0;
    static const bool chained_off = getenv("BIGC_CHAINED") && atoi(getenv("BIGC_CHAINED")) == 0;
    const bigc_config& c = m->cfg;
    if (ps_on || chained_off || bigc_linear_get_exact() || bigc_linear_get_fast() == 2 || !ps_config_ok(c)) return 0;
    return c.dim_feat % 32 == 0 ? 1 : 0;
}

extern "C" int bigc_forward(bigc_model* m, void* stream, const bigc_batch* b, void* workspace, size_t workspace_bytes,
                            int topk, const bigc_outputs* out) {
    StaticWeights static_weights(m ? m->epoch : 0);
    ModeScope mode_scope(m ? m->mode_exact : -1, m ? m->mode_fast : -1);
    BIGC_CHECK_ARG(m && b && out, "forward: null pointer");
    BIGC_CHECK_ARG(m->finalized, "forward: call bigc_finalize after setting the parameters");
    BIGC_CHECK_ARG(m->cfg.variant != 2, "forward: the pairwise baseline runs through bigc_pair_forward");
    BIGC_CHECK_ARG(b->n_vid > 0 && b->n_trk > 0 && b->n_rows > 0 && b->n_conv > 0, "forward: empty batch");
    BIGC_CHECK_ARG(b->max_trk > 0 && b->max_trk <= 512, "forward: max_trk=%d must be in [1,512]", b->max_trk);
    BIGC_CHECK_ARG(out->pred_logits && out->att && out->soid, "forward: pred_logits/att/soid outputs are required");
    const bigc_config& c = m->cfg;
    const Sizes z = sizes_of(m, b);
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (!workspace || workspace_bytes < (size_t)(base - (char*)workspace)) return bigc_fail(BIGC_ERR_WORKSPACE, "forward: no workspace");
    Arena a{base, workspace_bytes - (size_t)(base - (char*)workspace)};
    Work w;
    if (!layout(m, z, a, w)) return bigc_fail(BIGC_ERR_WORKSPACE, "forward: workspace of %zu bytes too small, need %zu", workspace_bytes,
                                              bigc_workspace_bytes(m, b));
    hipStream_t hs = (hipStream_t)stream;
    void* s = stream;
    // opt-in (BIGC_PS=1): the GEMMs on pre-split operands (csrc/gemm_ps.hip), when the caller can be told about an out-of-range
    // operand and the dimensions fit.  Measured on the benchmark step: 23.4 ms against 21.1 ms for the default path below
    // (profiles/HISTORY_rounds1-4.md section 3, "gemm_ps"), so it is not the default.
    static const bool ps_on = getenv("BIGC_PS") && atoi(getenv("BIGC_PS")) != 0;
    if (out->range_flag && ps_on && !bigc_linear_get_exact() && bigc_linear_get_fast() != 2 && ps_config_ok(c)) return forward_ps(m, stream, b, w, topk, out);
    if (out->range_flag && hipMemsetAsync(out->range_flag, 0, sizeof(int32_t), hs) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "forward: memset failed");
    // chained mode (default when the caller reads the range flag; BIGC_CHAINED=0 switches it off): see run_frontend_chained
    static const bool chained_off = getenv("BIGC_CHAINED") && atoi(getenv("BIGC_CHAINED")) == 0;
    // (not in the bf16 mode: the split form carries f16 heads, so a GEMM on pre-split operands could only round to f16)
    const bool chained = out->range_flag && !chained_off && !bigc_linear_get_exact() && bigc_linear_get_fast() != 2 && ps_config_ok(c);
    PsCtx cx{hs, w.flag};
    if (chained && hipMemsetAsync(w.flag, 0, sizeof(int32_t), hs) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "forward: memset failed");
    const int NT = b->n_trk, B = b->n_vid, Q = c.num_querys, BQ = B * Q;
    const int de = c.dim_enti, dp = c.dim_pred, da = c.dim_att, df = c.dim_ffn;
    const int H = c.n_att_head, hd = de / H;
    const float eps = 1e-5f;

    float* e2e = out->enti2enco ? out->enti2enco : w.e2e;
    if (chained) TRY(run_frontend_chained(m, s, b, w, e2e));
    else TRY(run_frontend(m, s, b, w, e2e));

    // ---- a5: encoder, sequence = the tracklets of a video (:459-462, layer :103-117)
    const float* x = e2e;
    for (int i = 0; i < c.n_enco_layers; ++i) {
        const EncLayer& L = m->enc[i];
        TRY(linear(s, x, de, L.in_proj, w.qkv, 3 * de, NT, 3 * de, de, BIGC_ACT_NONE));
        bigc_prof_hint(4.0 * NT * ((double)NT / B) * de, 16.0 * NT * (double)de);      // encoder: B sequences of ~NT / B tracklets
        TRY(bigc_mha_core_f32(s, w.qkv, 3 * de, w.qkv + de, 3 * de, w.qkv + 2 * de, 3 * de, w.ao, de, b->vid_trk_off, b->vid_trk_off,
                              B, b->max_trk, H, hd, nullptr));
        TRY(linear(s, w.ao, de, L.out_proj, w.t1, de, NT, de, de, BIGC_ACT_NONE, x, de));
        TRY(bigc_layernorm_f32(s, w.t1, L.n1.g, L.n1.b, w.x1, NT, de, eps));
        if (chained) {
            TRY(bigc_linear_hr8out(s, w.x1, de, L.l1.w, de, L.l1.b, nullptr, 0, nullptr, 0, w.fh, df, NT, df, de, BIGC_ACT_RELU, w.flag));
            TRY(ps(cx, w.fh, df, true, L.l2, df, w.t1, de, nullptr, 0, NT, de, df, BIGC_ACT_NONE, w.x1, de));
        } else {
            TRY(linear(s, w.x1, de, L.l1, w.fh, df, NT, df, de, BIGC_ACT_RELU));
            TRY(linear(s, w.fh, df, L.l2, w.t1, de, NT, de, df, BIGC_ACT_NONE, w.x1, de));
        }
        float* xo = (i == c.n_enco_layers - 1) ? (out->enco_output ? out->enco_output : w.enco) : w.x0;
        TRY(bigc_layernorm_f32(s, w.t1, L.n2.g, L.n2.b, xo, NT, de, eps));
        x = xo;
    }
    const float* enco = x;

    // ---- a6: role-aware decoder (:464-467, layer :178-225)
    TRY(bigc_fill_offsets(hs, w.q_off, B + 1, Q));
    if (c.n_deco_layers == 0) TRY(bigc_add_rows_f32(s, nullptr, m->qinit, w.query, BQ, dp, Q));
    const float att_scale = 1.0f / sqrtf((float)de);
    cx.decoder = true;
    for (int i = 0; i < c.n_deco_layers; ++i) {
        const DecLayer& L = m->dec[i];
        Lin qkp{L.in_proj.w, L.in_proj.b};
        Lin vp{L.in_proj.w + (size_t)2 * dp * dp, L.in_proj.b + 2 * dp};
        // The self-attention block of the FIRST layer only sees pred_query_init and pos_embedding
        // (:465,181-186), which are the same for every video: it is evaluated once on Q rows and the
        // result (after `query += pos`) is tiled over the batch. Later layers run on all B*Q rows.
        const int rows = (i == 0) ? Q : BQ, nseq = (i == 0) ? 1 : B;
        const float* qin = (i == 0) ? m->qinit : w.query;
        const bool fused = b->max_trk <= 64 && (da / 2) % 64 == 0;
        const bool last = i + 1 == c.n_deco_layers;                 // only the last layer's att is used downstream (:467)
        // "pre-split" form of the layer (large batches; BIGC_PRESPLIT): every producer whose result only a GEMM reads next writes the split
        // form -- the LayerNorms an hr8 twin, the attention core and the role attention ONLY the split form -- so that in_proj (one
        // q / k / v GEMM, build_qkpos), out_proj, the second role-wise linears, fc2.0 and fc2.3 run on csrc/gemm_ps.hip (K loop =
        // DMA + MFMA).  fc_pred2att / fc_enti2att, whose rounding the sharp attention amplifies, stay on the two-accumulator kernel.
        // Round 4: ON for every batch size.  csrc/gemm_ps.hip picks 128- / 192- / 256-row tiles per problem (M = 49152 = the decoder of a
        // 256-video step: 192-row tiles, two exact rounds of the 256 persistent workgroups), the result of a GEMM does not depend on the
        // tiling, and so a video's bits do not depend on how many videos share its batch (before, the form switched on at 512 videos).
        // Measured (profiles/r04_presplit_ab.log): 256 videos 18.87 -> 18.69 ms (GEMM family 0.390 -> 0.401), 64 videos 6.79 -> 6.78,
        // 8 videos 3.60 -> 3.63.  BIGC_PRESPLIT=0 switches it off.
        static const int presplit_env = getenv("BIGC_PRESPLIT") ? atoi(getenv("BIGC_PRESPLIT")) : -1;
        const bool presplit_off = presplit_env == 0;
        const bool presplit = chained && !presplit_off && fused && Q <= 192 && da / 2 <= 256 && (da / 2) % 128 == 0 && dp % 32 == 0 && dp <= de &&
                              dp / H == 64 && (dp == 512 || dp == 256 || dp == 128 || dp == 64) && df % 32 == 0;
        // the three LayerNorms of the layer as epilogues of the GEMMs in front of them (dim_pred = 512: the tile spans the whole row); BIGC_LN_FUSED=0: separate passes
        // OPT-IN (BIGC_LN_FUSED=1): measured on the 256-video step, the LayerNorm passes disappear (0.87 -> 0.07 ms) but the 64-row x 512-feature
        // tiles the epilogue needs run the GEMMs LDS-bound (+0.63 ms over the 17 launches; 96-row tiles spill): 19.06 -> 18.86 ms resident,
        // no gain through the API within the run-to-run noise.  Kept for the traffic it removes (3.4 GB per step) and as the starting point of a
        // kernel that feeds W from registers instead of LDS.
        static const bool ln_res = !(getenv("BIGC_LN_RES") && atoi(getenv("BIGC_LN_RES")) == 0);
        static const bool ln_fused_on = getenv("BIGC_LN_FUSED") && atoi(getenv("BIGC_LN_FUSED")) != 0;
        const bool ln_fused = presplit && ln_fused_on && dp == 512 && df % 32 == 0;
        if (presplit && i > 0) {
            TRY(build_qkpos(m, s));
            // w.querys = the split twin of w.query, written by the previous layer's last LayerNorm
            // (W pos, the position term of q and k, is added where q and k are READ -- the attention core stages whole rows -- instead of by
            // the GEMM's epilogue as a periodic residual; BIGC_QKPOS_MHA=0: by the epilogue, the same bits)
            static const bool qkpos_mha = !(getenv("BIGC_QKPOS_MHA") && atoi(getenv("BIGC_QKPOS_MHA")) == 0);
            const float* qkp_i = m->qkpos + (size_t)Q * 3 * dp * i;
            if (qkpos_mha) TRY(ps(cx, w.querys, dp, true, L.in_proj, dp, w.qk, 3 * dp, nullptr, 0, BQ, 3 * dp, dp, BIGC_ACT_NONE));
            else TRY(ps(cx, w.querys, dp, true, L.in_proj, dp, w.qk, 3 * dp, nullptr, 0, BQ, 3 * dp, dp, BIGC_ACT_NONE, qkp_i, 3 * dp, nullptr, 1, Q));
            bigc_prof_hint(4.0 * BQ * (double)Q * dp, 16.0 * BQ * (double)dp);          // B sequences of Q queries: 4 Q^2 dp flop each; q | k | v read, o written
            if (!bigc_mha_core_hr8_launch(hs, w.qk, 3 * dp, w.qk + dp, 3 * dp, w.qk + 2 * dp, 3 * dp, w.ao2, (long)dp * 4, w.q_off, B, Q, H, dp / H, w.flag,
                                          qkpos_mha ? qkp_i : nullptr, qkpos_mha ? qkp_i + dp : nullptr, 3 * dp))
                return bigc_fail(BIGC_ERR_ARG, "forward: self-attention shape outside the split-output kernel");
            if (ln_fused) {
                // out_proj, + query, norm1, + pos (:187-189) in ONE launch: the row's LayerNorm is the GEMM's epilogue (query is read as the
                // residual and overwritten by the lane that read it)
                TRY(ps_ln(cx, w.ao2, dp, L.out_proj, dp, w.query, dp, nullptr, 0, BQ, dp, dp, w.query, dp, L.n1, eps, m->pos, Q));
            } else {
                // (the residual of the three sublayers is added by the LayerNorm behind them, which reads it by whole rows: see layernorm_rows_kernel;
                // BIGC_LN_RES=0: by the GEMMs' epilogues, the same bits)
                if (ln_res) {
                    TRY(ps(cx, w.ao2, dp, true, L.out_proj, dp, w.t2, dp, nullptr, 0, BQ, dp, dp, BIGC_ACT_NONE));
                    TRY(bigc_layernorm_res(hs, w.t2, w.query, L.n1.g, L.n1.b, w.query, nullptr, 0, BQ, dp, eps, m->pos, Q, nullptr));   // LN(t2 + query) + pos, kept (:187-189)
                } else {
                    TRY(ps(cx, w.ao2, dp, true, L.out_proj, dp, w.t2, dp, nullptr, 0, BQ, dp, dp, BIGC_ACT_NONE, w.query, dp));
                    TRY(bigc_layernorm_add_f32(s, w.t2, L.n1.g, L.n1.b, w.query, BQ, dp, eps, m->pos, Q, nullptr));   // query += pos, kept (:189)
                }
            }
        } else {
        // layers > 0: qk_in = query + pos was written by the previous layer's last LayerNorm
        if (i == 0) TRY(bigc_add_rows_f32(s, qin, m->pos, w.qk_in, rows, dp, Q));
        TRY(linear(s, w.qk_in, dp, qkp, w.qk, 2 * dp, rows, 2 * dp, dp, BIGC_ACT_NONE));
        TRY(linear(s, qin, dp, vp, w.vv, dp, rows, dp, dp, BIGC_ACT_NONE));
        TRY(bigc_mha_core_f32(s, w.qk, 2 * dp, w.qk + dp, 2 * dp, w.vv, dp, w.ao2, dp, w.q_off, w.q_off, nseq, Q, H, dp / H, nullptr));
        TRY(linear(s, w.ao2, dp, L.out_proj, w.t2, dp, rows, dp, dp, BIGC_ACT_NONE, qin, dp));
        if (i == 0) {
            TRY(bigc_layernorm_add_f32(s, w.t2, L.n1.g, L.n1.b, w.qk_in, Q, dp, eps, m->pos, Q, nullptr));
            TRY(bigc_add_rows_f32(s, nullptr, w.qk_in, w.query, BQ, dp, Q));
        } else {
            TRY(bigc_layernorm_add_f32(s, w.t2, L.n1.g, L.n1.b, w.query, BQ, dp, eps, m->pos, Q, nullptr));   // query += pos, kept (:189)
        }
        }
        if (!presplit) TRY(linear(s, enco, de, L.e2a, w.e2a, da, NT, da, de, BIGC_ACT_NONE));
        TRY(linear(s, w.query, dp, L.p2a, w.p2a, da, BQ, da, dp, BIGC_ACT_NONE));
        if (presplit) {
            // fc_enti2att and the two role projections of the entity rows: one GEMM into [e2a | ew0 | ew1] rows (build_enco_cat)
            const int EW = da + 2 * dp, EP = EW * c.n_deco_layers;            // this layer's columns of w.ecat, and the pitch of its rows
            if (i == 0) {
                TRY(build_enco_cat(m, s));
                // (round 5, measured and dropped: this N = 6 x 1536 product on the pre-split-operand kernel -- enco_output converted once, blocked
                // accumulation -- is 0.29 ms SLOWER per step than the in-loop-split kernel here (17.45 vs 17.16 ms of GEMM time, same box))
                TRY(bigc_linear_f32(s, enco, de, nullptr, 1, de, m->enco_cat, de, m->enco_cat + (size_t)EP * de, nullptr, 0, w.ecat, EP, NT, EP, de, BIGC_ACT_NONE));
            }
            const float* ec = w.ecat + (size_t)i * EW;
            TRY(build_role_cat(m, s));
            // (profiling: per video and role Q x N scores over da / 2 and Q x N x dp values; N ~ NT / B; reads p2a + the entity rows, writes 2 dp per query)
            bigc_prof_hint(2.0 * 2.0 * BQ * ((double)NT / B) * (da / 2 + dp), 4.0 * (BQ * (double)da + (double)NT * (da + 2 * dp) + BQ * 2.0 * dp));
            if (!bigc_role_fused_proj_hr8_launch(hs, w.p2a, da, ec, EP, ec + da, EP, (long)dp, L.role0[0].b, L.role1[0].b, 1, last ? out->att : nullptr,
                                                 w.vals, (long)2 * dp * 4, dp, b->vid_trk_off, B, Q, b->max_trk, da / 2, dp, att_scale, w.flag))
                return bigc_fail(BIGC_ERR_ARG, "forward: role-attention shape outside the split-output kernel");
            const float* wc = m->role_cat + ((size_t)dp * 2 * dp + dp) * i;
            Lin rc{wc, wc + (size_t)dp * 2 * dp};
            if (ln_fused) {
                // (:209-211) query + role0.2(.) + role1.2(.), norm2 -> t2 and its split twin; (:218-223) fc2, + t2, norm3 -> query and its twin
                TRY(ps_ln(cx, w.vals, 2 * dp, rc, 2 * dp, w.t2, dp, w.qkins, dp, BQ, dp, 2 * dp, w.query, dp, L.n2, eps));
                TRY(ps(cx, w.qkins, dp, true, L.f0, dp, nullptr, 0, w.rh, df, BQ, df, dp, BIGC_ACT_RELU));
                TRY(ps_ln(cx, w.rh, df, L.f3, df, w.query, dp, last ? nullptr : w.querys, dp, BQ, dp, df, w.t2, dp, L.n3, eps));
                continue;
            }
            if (ln_res) {
                TRY(ps(cx, w.vals, 2 * dp, true, rc, 2 * dp, w.t3, dp, nullptr, 0, BQ, dp, 2 * dp, BIGC_ACT_NONE));
                TRY(bigc_layernorm_res(hs, w.t3, w.query, L.n2.g, L.n2.b, w.t2, w.qkins, (long)dp * 4, BQ, dp, eps, nullptr, 0, w.flag));      // LN(t3 + query) -> t2 and its split twin
                TRY(ps(cx, w.qkins, dp, true, L.f0, dp, nullptr, 0, w.rh, df, BQ, df, dp, BIGC_ACT_RELU));
                TRY(ps(cx, w.rh, df, true, L.f3, df, w.t3, dp, nullptr, 0, BQ, dp, df, BIGC_ACT_NONE));
                TRY(bigc_layernorm_res(hs, w.t3, w.t2, L.n3.g, L.n3.b, w.query, last ? nullptr : w.querys, (long)dp * 4, BQ, dp, eps, nullptr, 0, w.flag));   // LN(t3 + t2) -> query (and its split twin)
                continue;
            }
            TRY(ps(cx, w.vals, 2 * dp, true, rc, 2 * dp, w.t3, dp, nullptr, 0, BQ, dp, 2 * dp, BIGC_ACT_NONE, w.query, dp));
            TRY(bigc_layernorm_hr8(hs, w.t3, L.n2.g, L.n2.b, w.t2, w.qkins, (long)dp * 4, BQ, dp, eps, w.flag));           // t2 and its split twin
            TRY(ps(cx, w.qkins, dp, true, L.f0, dp, nullptr, 0, w.rh, df, BQ, df, dp, BIGC_ACT_RELU));
            TRY(ps(cx, w.rh, df, true, L.f3, df, w.t3, dp, nullptr, 0, BQ, dp, df, BIGC_ACT_NONE, w.t2, dp));
            if (!last) TRY(bigc_layernorm_hr8(hs, w.t3, L.n3.g, L.n3.b, w.query, w.querys, (long)dp * 4, BQ, dp, eps, w.flag));   // query and its split twin
            else TRY(bigc_layernorm_f32(s, w.t3, L.n3.g, L.n3.b, w.query, BQ, dp, eps));
            continue;
        }
        if (fused && Q <= 192 && da / 2 <= 256 && dp % 32 == 0 && dp <= de) {
            // (att_r @ enco_output) W_r^T = att_r @ (enco_output W_r^T): the first role-wise linear (:204-208) is applied to the NT
            // entity rows instead of the B*Q value rows (a third as many); the role kernel adds its bias and the ReLU
            float* ew = w.rh;                                        // 2 x (NT, dp), free until the FFN below
            TRY(bigc_linear_f32(s, enco, de, nullptr, 1, de, L.role0[0].w, de, nullptr, nullptr, 0, ew, dp, NT, dp, de, BIGC_ACT_NONE));
            TRY(bigc_linear_f32(s, enco, de, nullptr, 1, de, L.role1[0].w, de, nullptr, nullptr, 0, ew + (size_t)NT * dp, dp, NT, dp, de, BIGC_ACT_NONE));
            TRY(build_role_cat(m, s));
            // the two roles side by side in one row of w.vals: query + role0.2(.) + role1.2(.) (:209-211) is one K = 2 dp GEMM
            TRY(bigc_role_fused_proj_f32(s, w.p2a, da, w.e2a, da, ew, dp, (long)NT * dp, L.role0[0].b, L.role1[0].b, 1, last ? out->att : nullptr,
                                         w.vals, dp, 2 * dp, b->vid_trk_off, B, Q, b->max_trk, da / 2, dp, att_scale));
            const float* wc = m->role_cat + ((size_t)dp * 2 * dp + dp) * i;
            TRY(bigc_linear_f32(s, w.vals, 2 * dp, nullptr, 1, 2 * dp, wc, 2 * dp, wc + (size_t)dp * 2 * dp, w.query, dp, w.t3, dp, BQ, dp, 2 * dp,
                                BIGC_ACT_NONE));
        } else {
        if (fused) {
            TRY(bigc_role_fused_f32(s, w.p2a, da, w.e2a, da, enco, de, last ? out->att : nullptr, w.vals, b->vid_trk_off, B, Q,
                                    b->max_trk, da / 2, de, att_scale));
        } else {
            TRY(bigc_role_attention_f32(s, w.p2a, da, w.e2a, da, out->att, b->vid_trk_off, B, Q, da / 2, att_scale));
            TRY(bigc_role_values_f32(s, out->att, enco, de, w.vals, b->vid_trk_off, B, Q, de));
        }
        TRY(linear(s, w.vals, de, L.role0[0], w.rh, dp, BQ, dp, de, BIGC_ACT_RELU));
        TRY(linear(s, w.rh, dp, L.role0[1], w.t2, dp, BQ, dp, dp, BIGC_ACT_NONE, w.query, dp));
        TRY(linear(s, w.vals + (size_t)BQ * de, de, L.role1[0], w.rh, dp, BQ, dp, de, BIGC_ACT_RELU));
        TRY(linear(s, w.rh, dp, L.role1[1], w.t3, dp, BQ, dp, dp, BIGC_ACT_NONE, w.t2, dp));
        }
        TRY(bigc_layernorm_f32(s, w.t3, L.n2.g, L.n2.b, w.t2, BQ, dp, eps));
        if (chained) {
            TRY(bigc_linear_hr8out(s, w.t2, dp, L.f0.w, dp, L.f0.b, nullptr, 0, nullptr, 0, w.rh, df, BQ, df, dp, BIGC_ACT_RELU, w.flag));
            TRY(ps(cx, w.rh, df, true, L.f3, df, w.t3, dp, nullptr, 0, BQ, dp, df, BIGC_ACT_NONE, w.t2, dp));
        } else {
            TRY(linear(s, w.t2, dp, L.f0, w.rh, df, BQ, df, dp, BIGC_ACT_RELU));
            TRY(linear(s, w.rh, df, L.f3, w.t3, dp, BQ, dp, df, BIGC_ACT_NONE, w.t2, dp));
        }
        if (i + 1 < c.n_deco_layers) TRY(bigc_layernorm_add_f32(s, w.t3, L.n3.g, L.n3.b, w.qk_in, BQ, dp, eps, m->pos, Q, w.query));
        else TRY(bigc_layernorm_f32(s, w.t3, L.n3.g, L.n3.b, w.query, BQ, dp, eps));
    }
    if (out->pred_query) {
        hipError_t e = hipMemcpyAsync(out->pred_query, w.query, sizeof(float) * (size_t)BQ * dp, hipMemcpyDeviceToDevice, hs);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "forward: memcpy: %s", hipGetErrorString(e));
    }

    // ---- a7: prediction head (:478-507; VidOR models/model_0v7.py:483-513)
    const float* seg_src[4];
    int32_t seg_w[4], seg_cat[4];
    int n_seg = 0;
    if (c.variant == 0) {
        if (c.dim_extra > 0) {
            TRY(bigc_extra_mean(s, b, c, w.xmean, NT));
            TRY(linear(s, w.xmean, c.dim_extra, m->i3d, w.i3dfc, de, NT, de, c.dim_extra, BIGC_ACT_RELU));
            seg_src[n_seg] = w.i3dfc; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
            seg_src[n_seg] = e2e; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
            seg_src[n_seg] = m->emb; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 1;
        } else {
            seg_src[n_seg] = m->emb; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 1;
            seg_src[n_seg] = e2e; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
        }
    } else {
        if (c.use_clsme) {
            if (c.has_enti_emb) {
                seg_src[n_seg] = m->emb; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 1;
            } else {
                BIGC_CHECK_ARG(c.dim_extra == c.dim_clsme, "forward: classeme width %d != dim_clsme %d", c.dim_extra, c.dim_clsme);
                TRY(bigc_extra_mean(s, b, c, w.xmean, NT));
                seg_src[n_seg] = w.xmean; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 0;
            }
        }
        seg_src[n_seg] = e2e; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
    }
    if (c.variant == 0) {
        // Factorised classifier (:498-503): the concat row [query | s_seg0 | o_seg0 | s_seg1 | ...] is never built.  The
        // subject / object column blocks of the weight are applied per TRACKLET (NT rows instead of B*Q): U = sum over the
        // segments of seg W[:, s-block]^T, V likewise; then logits = query W[:, :dp]^T + b + (bias row + U[s] + V[o]).
        const int P = c.num_pred_cats, hk = m->head_k, ldu = (P + 3) & ~3;
        float* U = w.head_in;                                  // the arena of the concat rows (B*Q x head_k) is far larger
        float* V = U + (size_t)NT * ldu;
        int col = dp;
        for (int g = 0; g < n_seg; ++g) {
            const float* Wg = m->head0.w + col;
            for (int role = 0; role < 2; ++role) {
                float* dst = role == 0 ? U : V;
                TRY(bigc_linear_f32(s, seg_src[g], seg_w[g], seg_cat[g] ? b->trk_cat : nullptr, 1, seg_w[g], Wg + role * seg_w[g], hk, nullptr,
                                    g ? dst : nullptr, ldu, dst, ldu, NT, P, seg_w[g], BIGC_ACT_NONE));
            }
            col += 2 * seg_w[g];
        }
        TRY(bigc_head_pick(hs, out->att, b->vid_trk_off, b->trk_cat, B, Q, U, V, ldu, m->bias_matrix, c.num_enti_cats, P, out->soid, w.bias_g));
        TRY(bigc_linear_f32(s, w.query, dp, nullptr, 1, dp, m->head0.w, hk, m->head0.b, w.bias_g, P, out->pred_logits, P, BQ, P, dp, BIGC_ACT_NONE));
    } else {
        TRY(bigc_head_gather(s, out->att, b->vid_trk_off, b->trk_cat, B, Q, w.query, dp, n_seg, seg_src, seg_w, seg_cat, m->bias_matrix,
                             c.num_enti_cats, c.num_pred_cats, out->soid, w.head_in, m->head_k, w.bias_g));
        TRY(linear(s, w.head_in, m->head_k, m->head0, w.hh, df, BQ, df, m->head_k, BIGC_ACT_RELU));
        TRY(linear(s, w.hh, df, m->head2, out->pred_logits, c.num_pred_cats, BQ, c.num_pred_cats, df, BIGC_ACT_NONE, w.bias_g,
                   c.num_pred_cats));
    }

    // ---- a8: construct_triplet (:707-785)
    if (out->quint) {
        BIGC_CHECK_ARG(out->scores && out->dura && out->qid && out->n_out, "forward: all construct_triplet outputs or none");
        TRY(bigc_construct_triplets(s, out->pred_logits, c.num_pred_cats, out->soid, b->vid_trk_off, b->trk_dura, b->trk_cat, b->trk_score,
                                    B, Q, topk, out->quint, out->scores, out->dura, out->qid, out->n_out));
    }
    if (chained && hipMemcpyAsync(out->range_flag, w.flag, sizeof(int32_t), hipMemcpyDeviceToDevice, hs) != hipSuccess)
        return bigc_fail(BIGC_ERR_HIP, "forward: memcpy failed");
    return BIGC_OK;
}

// ------------------------------------------------------------------------------------------------
// Base_C, the explicit pairwise baseline (models/model_pairwise_baseline.py:174-197,236-273): front-end as above, then for
// EVERY ordered pair (s, o), s != o, of a video: logits = W2 relu(W1 [s_clsme, o_clsme, s_feat, o_feat] + b1) + b2 + bias[s_cat, o_cat].
// The first layer is linear in the concatenation, so it is evaluated per TRACKLET and per role
//     U[t] = W1[:, s_clsme cols] clsme_t + W1[:, s_feat cols] feat_t + b1,   V[t] = W1[:, o_clsme cols] clsme_t + W1[:, o_feat cols] feat_t
// (4 GEMMs over N rows instead of one over N(N-1) rows of width 1624) and fused per pair as relu(U[s] + V[o]).
// ------------------------------------------------------------------------------------------------
int bigc_pair_hidden(hipStream_t s, const float* U, const float* V, const int32_t* vid_trk_off, const int32_t* vid_pair_off, float* H,
                     int n_vid, long n_pairs, int d);
int bigc_pair_bias(hipStream_t s, const float* bias_matrix, const int32_t* trk_cat, const int32_t* vid_trk_off, const int32_t* vid_pair_off,
                   float* out, int n_vid, long n_pairs, int n_cat, int n_pred);
int bigc_gather_rows(hipStream_t s, const float* src, const int32_t* idx, float* out, int n, int d);

extern "C" size_t bigc_pair_workspace_bytes(const bigc_model* m, const bigc_batch* b, long n_pairs) {
    if (!m || !b) return 0;
    const bigc_config& c = m->cfg;
    const size_t NT = b->n_trk, df = c.dim_ffn;
    return bigc_workspace_bytes(m, b) + 4 * (256 + sizeof(float) * (NT * df * 2 + NT * (size_t)c.dim_clsme + (size_t)n_pairs * (df + c.num_pred_cats)));
}

extern "C" int bigc_pair_forward(bigc_model* m, void* stream, const bigc_batch* b, const int32_t* vid_pair_off, long n_pairs, void* workspace,
                                 size_t workspace_bytes, float* enti2enco, float* pair_logits) {
    StaticWeights static_weights(m ? m->epoch : 0);
    ModeScope mode_scope(m ? m->mode_exact : -1, m ? m->mode_fast : -1);
    BIGC_CHECK_ARG(m && b && vid_pair_off && pair_logits, "pair_forward: null pointer");
    BIGC_CHECK_ARG(m->finalized && m->cfg.variant == 2, "pair_forward: needs a finalized pairwise-baseline model");
    BIGC_CHECK_ARG(b->n_vid > 0 && b->n_trk > 0 && b->n_rows > 0 && b->n_conv > 0 && n_pairs > 0, "pair_forward: empty batch");
    const bigc_config& c = m->cfg;
    const Sizes z = sizes_of(m, b);
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (!workspace || workspace_bytes < (size_t)(base - (char*)workspace)) return bigc_fail(BIGC_ERR_WORKSPACE, "pair_forward: no workspace");
    Arena a{base, workspace_bytes - (size_t)(base - (char*)workspace)};
    Work w;
    layout(m, z, a, w);
    const int NT = b->n_trk, de = c.dim_enti, df = c.dim_ffn, dc = c.use_clsme ? c.dim_clsme : 0, hk = m->head_k, P = c.num_pred_cats;
    float* U = a.take<float>((size_t)NT * df);
    float* V = a.take<float>((size_t)NT * df);
    float* clsme = a.take<float>((size_t)NT * (dc > 0 ? dc : 1));
    float* H = a.take<float>((size_t)n_pairs * df);
    float* bias = a.take<float>((size_t)n_pairs * P);
    if (a.overflow) return bigc_fail(BIGC_ERR_WORKSPACE, "pair_forward: workspace of %zu bytes too small, need %zu", workspace_bytes,
                                     bigc_pair_workspace_bytes(m, b, n_pairs));
    hipStream_t hs = (hipStream_t)stream;
    void* s = stream;
    float* e2e = enti2enco ? enti2enco : w.e2e;
    TRY(run_frontend(m, s, b, w, e2e));
    const float* W1 = m->head0.w;
    const float* clsme_rows = nullptr;
    if (dc > 0) {
        if (c.has_enti_emb) {               // category embedding as classeme (:251-253)
            TRY(bigc_gather_rows(hs, m->emb, b->trk_cat, clsme, NT, dc));
        } else {                            // time-mean of the tracklet's classeme columns over the stretched frames (:191)
            TRY(bigc_extra_mean(s, b, c, clsme, NT));
        }
        clsme_rows = clsme;
    }
    // column blocks of W1 (df, hk): [s_clsme | o_clsme | s_feat | o_feat] (:257), or [s_feat | o_feat] without classemes (:259)
    if (dc > 0) {
        TRY(bigc_linear_f32(s, clsme_rows, dc, nullptr, 1, dc, W1, hk, m->head0.b, nullptr, 0, U, df, NT, df, dc, BIGC_ACT_NONE));
        TRY(bigc_linear_f32(s, e2e, de, nullptr, 1, de, W1 + 2 * dc, hk, nullptr, U, df, U, df, NT, df, de, BIGC_ACT_NONE));
        TRY(bigc_linear_f32(s, clsme_rows, dc, nullptr, 1, dc, W1 + dc, hk, nullptr, nullptr, 0, V, df, NT, df, dc, BIGC_ACT_NONE));
        TRY(bigc_linear_f32(s, e2e, de, nullptr, 1, de, W1 + 2 * dc + de, hk, nullptr, V, df, V, df, NT, df, de, BIGC_ACT_NONE));
    } else {
        TRY(bigc_linear_f32(s, e2e, de, nullptr, 1, de, W1, hk, m->head0.b, nullptr, 0, U, df, NT, df, de, BIGC_ACT_NONE));
        TRY(bigc_linear_f32(s, e2e, de, nullptr, 1, de, W1 + de, hk, nullptr, nullptr, 0, V, df, NT, df, de, BIGC_ACT_NONE));
    }
    TRY(bigc_pair_hidden(hs, U, V, b->vid_trk_off, vid_pair_off, H, b->n_vid, n_pairs, df));
    TRY(bigc_pair_bias(hs, m->bias_matrix, b->trk_cat, b->vid_trk_off, vid_pair_off, bias, b->n_vid, n_pairs, c.num_enti_cats, P));
    TRY(bigc_linear_f32(s, H, df, nullptr, 1, df, m->head2.w, df, m->head2.b, bias, P, pair_logits, P, (int)n_pairs, P, df, BIGC_ACT_NONE));
    return BIGC_OK;
}
