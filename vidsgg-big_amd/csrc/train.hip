// BIG-C training step on the HIP path: forward with saved activations and the full backward pass
// (the autograd graph of the reference's ``_forward_train``: models/model_0v10.py:510-556 through
// encode2decode :434-475 and prediction_head :478-507; VidOR variant models/model_0v7.py).
//
// Dropout is not applied (parity is defined with the reference modules in eval() mode).  All gradients
// are ACCUMULATED into the destinations registered with bigc_set_grad (one flat buffer on the host side,
// which is also the message of the per-step RCCL all-reduce).
#include "model_internal.h"
#include "prof.h"

// internal launchers defined in backward.hip
int bigc_repack_conv_weight_bwd(hipStream_t s, const float* w, float* out, int n_out, int n_in, int taps);
int bigc_head_gather_bwd(hipStream_t s, const float* dhead, int ld, const int32_t* soid, const int32_t* e_off, const int32_t* trk_vid,
                         int n_trk, int nq, int n_seg, float* const* dst, const int* width, const int* col);
int bigc_bias_matrix_bwd(hipStream_t s, const float* dlogits, int ldg, const int32_t* soid, const int32_t* e_off, const int32_t* trk_cat,
                         int n_vid, int nq, int n_ecat, int n_pcat, float* dbias);
int bigc_copy_cols(hipStream_t s, const float* src, int lds_, float* dst, int ldd, long rows, int cols);
int bigc_conv_grad_unpack(hipStream_t s, const float* g_tapmajor, float* g_torch, int n_out, int n_in, int taps);
int bigc_transpose_pad(hipStream_t s, const float* in, float* out, int rows, int cols, int ld_out);

extern "C" int bigc_transpose_f32(void* stream, const float* in, float* out, int rows, int cols);
extern "C" int bigc_linear_dw_f32(void* stream, const float* dY, int ldy, const float* X, int ldx, const int32_t* x_idx, int nseg,
                                  int seg_len, float* dW, int ldw, float* db, int M, int N, int K);
extern "C" int bigc_relu_bwd_f32(void* stream, float* dY, const float* Y, long n);
extern "C" int bigc_axpy_f32(void* stream, float* y, const float* x, long n);
extern "C" int bigc_layernorm_bwd_f32(void* stream, const float* x, const float* gamma, const float* dy, float* dx, float* dgamma,
                                      float* dbeta, int rows, int d, float eps);
extern "C" int bigc_reduce_periodic_f32(void* stream, const float* x, float* out, int rows, int d, int period);
extern "C" int bigc_maxpool_bwd_f32(void* stream, const float* conv, const int32_t* trk_conv_off, const float* dpooled, float* dconv,
                                    int n_trk, int d, int P);
extern "C" int bigc_conv_gather_bwd_f32(void* stream, const float* dconv, const int32_t* trk_row_off, const int32_t* trk_conv_off,
                                        const int32_t* trk_vid, const int32_t* vid_tmax, float* G, int n_trk, int d);
extern "C" int bigc_mha_bwd_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                const float* o, int ldo, const float* dO, int lddo, const float* lse, float* delta_scratch,
                                float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                                const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q, int max_k, long total_q_rows,
                                int n_head, int hd);
void bigc_finish_begin(void);                 // backward.hip: deferred partial-row finishes
int bigc_finish_flush(hipStream_t s);
struct FinishScope {
    hipStream_t s;
    explicit FinishScope(hipStream_t s_) : s(s_) { bigc_finish_begin(); }
    ~FinishScope() { (void)bigc_finish_flush(s); }
};
int bigc_role_attention_bwd_launch(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* datt,
                                   float* ds, const int32_t* e_off, int n_vid, int nq, int half, float scale, int max_ent);        // attention_bwd.hip
extern "C" int bigc_role_attention_bwd_f32(void* stream, const float* p2a, int ldp, const float* e2a, int lde, const float* datt,
                                           float* ds, const int32_t* e_off, int n_vid, int nq, int half, float scale);
extern "C" int bigc_role_small_gemm_f32(void* stream, int mode, const float* att_or_ds, const float* dvals_or_p2a, int ld1,
                                        const float* enco_or_e2a, int ld2, float* out, int ldo, const int32_t* e_off, int n_vid, int nq,
                                        int d, int half, int max_trk);
extern "C" int bigc_role_small_gemm_pair_f32(void* stream, int pair, const float* x, const float* y, int ld_y, const float* z, int ld_z, float* out_a,
                                             float* out_b, int ldo_b, const int32_t* e_off, int n_vid, int nq, int d, int half, int max_trk);

extern "C" int bigc_mha_core_drop_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                      float* o, int ldo, const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q,
                                      int n_head, int hd, float* lse, float drop_p, uint32_t drop_seed, uint32_t drop_site);
extern "C" int bigc_mha_bwd_drop_f32(void* stream, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                     const float* o, int ldo, const float* dO, int lddo, const float* lse, float* delta_scratch,
                                     float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                                     const int32_t* q_off, const int32_t* k_off, int n_seq, int max_q, int max_k, long total_q_rows,
                                     int n_head, int hd, float drop_p, uint32_t drop_seed, uint32_t drop_site);
extern "C" int bigc_dropout_f32(void* stream, float* x, long rows, int n, int ld, float p, uint32_t seed, uint32_t site);
int bigc_scale_f32(hipStream_t s, float* x, long n, float a);

// dropout sites of the reference's transformer layers (models/model_0v10.py:75-84,109-114 encoder; :151,164-169 decoder)
static inline uint32_t enc_site(int layer, int which) { return 16u * layer + which; }          // 0 attention, 1 dropout1, 2 ffn, 3 dropout2
static inline uint32_t dec_site(int layer, int which) { return 4096u + 16u * layer + which; }  // 0 attention, 1 fc2 dropout

namespace {

struct EncS { float *qkv, *lse, *ao, *t1, *x1, *fh, *t1b, *out; };
struct DecS { float *qin, *qk_in, *qk, *vv, *lse, *ao2, *t2, *q2, *e2a, *p2a, *att, *vals, *rh0, *rh1, *t3b, *q3, *f, *t3c, *qout; };

struct TrainWork {
    // saved by the forward
    float *bbox8, *hb, *hv, *enti, *conv, *pooled, *eh, *e2e, *xmean, *i3dfc, *head_in, *bias_g, *hh, *t3tmp, *rtmp;
    int32_t *conv_idx, *q_off, *soid;
    std::vector<EncS> enc;
    std::vector<DecS> dec;
    // backward scratch
    float *gq, *ga, *gb, *gc, *gqk, *gvals, *gp2a, *ghead, *ghh, *gatt, *ds, *genco, *ge2e, *gx, *gy, *gz, *gqkv, *ge2a, *gi3d,
          *gpooled, *delta, *gconv, *G, *genti, *gh, *wT, *gconvw;
    float* gd[8];      // decoder layer: one buffer per gradient that a deferred weight-gradient product reads (see the decoder loop)
    float* alt[10];    // a second set of the ten buffers such products read (ga gb gc gd1 gd2 gd5 gd6 gqk gp2a ge2a): odd layers, when the products run on their own stream
};

struct TSizes { size_t R, NT, C, B, Q, BQ; };

int round4(int x) { return (x + 3) & ~3; }

bool train_layout(const bigc_model* m, const TSizes& z, Arena& a, TrainWork& w) {
    const bigc_config& c = m->cfg;
    const size_t de = c.dim_enti, dp = c.dim_pred, da = c.dim_att, df = c.dim_ffn, H = c.n_att_head;
    const size_t mx = dp > df ? dp : df;
    w.bbox8 = a.take<float>(z.R * 8);
    w.hb = a.take<float>(z.R * de);
    w.hv = a.take<float>(z.R * de);
    w.enti = a.take<float>(z.R * 2 * de);
    w.conv_idx = a.take<int32_t>(z.C * 3);
    w.conv = a.take<float>(z.C * de);
    w.pooled = a.take<float>(z.NT * de * c.enco_pool_len);
    w.eh = a.take<float>(z.NT * de);
    w.e2e = a.take<float>(z.NT * de);
    w.enc.assign(c.n_enco_layers, EncS());
    for (auto& e : w.enc) {
        e.qkv = a.take<float>(z.NT * 3 * de); e.lse = a.take<float>(z.NT * H); e.ao = a.take<float>(z.NT * de);
        e.t1 = a.take<float>(z.NT * de); e.x1 = a.take<float>(z.NT * de); e.fh = a.take<float>(z.NT * df);
        e.t1b = a.take<float>(z.NT * de); e.out = a.take<float>(z.NT * de);
    }
    w.q_off = a.take<int32_t>(z.B + 1);
    w.dec.assign(c.n_deco_layers, DecS());
    for (auto& d : w.dec) {
        d.qin = a.take<float>(z.BQ * dp); d.qk_in = a.take<float>(z.BQ * dp); d.qk = a.take<float>(z.BQ * 2 * dp);
        d.vv = a.take<float>(z.BQ * dp); d.lse = a.take<float>(z.BQ * H); d.ao2 = a.take<float>(z.BQ * dp);
        d.t2 = a.take<float>(z.BQ * dp); d.q2 = a.take<float>(z.BQ * dp); d.e2a = a.take<float>(z.NT * da);
        d.p2a = a.take<float>(z.BQ * da); d.att = a.take<float>(2 * z.Q * z.NT); d.vals = a.take<float>(2 * z.BQ * de);
        d.rh0 = a.take<float>(z.BQ * dp); d.rh1 = a.take<float>(z.BQ * dp); d.t3b = a.take<float>(z.BQ * dp);
        d.q3 = a.take<float>(z.BQ * dp); d.f = a.take<float>(z.BQ * df); d.t3c = a.take<float>(z.BQ * dp);
        d.qout = a.take<float>(z.BQ * dp);
    }
    w.t3tmp = a.take<float>(z.BQ * dp);
    w.rtmp = a.take<float>(z.BQ * dp);
    w.xmean = a.take<float>(z.NT * (c.dim_extra > 0 ? c.dim_extra : 1));
    w.i3dfc = a.take<float>(z.NT * de);
    w.head_in = a.take<float>(z.BQ * (size_t)m->head_k);
    w.bias_g = a.take<float>(z.BQ * c.num_pred_cats);
    w.hh = a.take<float>(z.BQ * df);
    w.soid = a.take<int32_t>(z.B * 2 * z.Q);
    // backward scratch
    w.gq = a.take<float>(z.BQ * mx); w.ga = a.take<float>(z.BQ * mx); w.gb = a.take<float>(z.BQ * mx); w.gc = a.take<float>(z.BQ * mx);
    for (int i = 0; i < 8; ++i) w.gd[i] = a.take<float>(z.BQ * mx);
    for (int i = 0; i < 7; ++i) w.alt[i] = a.take<float>(z.BQ * mx);
    w.alt[7] = a.take<float>(z.BQ * 2 * dp); w.alt[8] = a.take<float>(z.BQ * da); w.alt[9] = a.take<float>(z.NT * da);
    w.gqk = a.take<float>(z.BQ * 2 * dp);
    w.gvals = a.take<float>(2 * z.BQ * de);
    w.gp2a = a.take<float>(z.BQ * da);
    w.ghead = a.take<float>(z.BQ * (size_t)m->head_k);
    w.ghh = a.take<float>(z.BQ * df);
    w.gatt = a.take<float>(2 * z.Q * z.NT);
    w.ds = a.take<float>(2 * z.Q * z.NT);
    w.genco = a.take<float>(z.NT * de); w.ge2e = a.take<float>(z.NT * de);
    w.gx = a.take<float>(z.NT * mx); w.gy = a.take<float>(z.NT * mx); w.gz = a.take<float>(z.NT * mx);
    w.gqkv = a.take<float>(z.NT * 3 * de);
    w.ge2a = a.take<float>(z.NT * da);
    w.gi3d = a.take<float>(z.NT * de);
    w.gpooled = a.take<float>(z.NT * de * c.enco_pool_len);
    w.delta = a.take<float>((z.BQ > z.NT ? z.BQ : z.NT) * H);
    w.gconv = a.take<float>(z.C * de);
    w.G = a.take<float>(z.R * 3 * de);
    w.genti = a.take<float>(z.R * 2 * de);
    w.gh = a.take<float>(z.R * de);
    size_t wmax = (size_t)m->head_k * round4(c.num_pred_cats);
    const size_t cand[] = {(size_t)de * c.enco_pool_len * de, (size_t)3 * de * de, (size_t)3 * dp * dp, (size_t)df * round4(c.num_pred_cats),
                           (size_t)df * m->head_k, (size_t)da * de, (size_t)df * dp};
    for (size_t v : cand) if (v > wmax) wmax = v;
    w.wT = a.take<float>(wmax);
    w.gconvw = a.take<float>((size_t)de * 6 * de);
    return !a.overflow;
}

TSizes tsizes(const bigc_model* m, const bigc_batch* b) {
    TSizes z;
    z.R = b->n_rows; z.NT = b->n_trk; z.C = b->n_conv; z.B = b->n_vid; z.Q = m->cfg.num_querys; z.BQ = z.B * z.Q;
    return z;
}

float* grad_of(bigc_model* m, const std::string& key) {
    auto it = m->grads.find(key);
    return it == m->grads.end() ? nullptr : it->second;
}

// Backward helper of y = act(x W^T + b) (+ residual): given dY (already masked for ReLU by the caller)
//   dW += dY^T X, db += colsum(dY), and (optionally) dX (=|+=) dY W.
struct Bwd {
    bigc_model* m; void* s; TrainWork* w;
    // Deferred weight gradients: while `defer` is on, lin() / dw() only record their product; flush() runs the recorded ones as ONE
    // grouped launch (backward.hip: bigc_linear_dw_group).  The caller guarantees that nothing writes to a recorded dY / X before
    // the flush (the decoder layer below hands every gradient a buffer of its own).
    bool defer = false;
    std::vector<BigcDwJob> pending;
    // The grouped launches on a stream of their own (decoder): forked from the caller's stream when the layer's chain has been issued, so
    // that they run beside the NEXT layer's chain (at the reference's batch of 8 videos neither fills the chip); that layer works on the
    // other buffer set, and the one after it waits for this launch before it reuses the set (wait_set); join() at the end of the pass.
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_done[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int flush_side(int set) {
        if (pending.empty()) return BIGC_OK;
        if (hipEventRecord(ev_fork, (hipStream_t)s) != hipSuccess || hipStreamWaitEvent(side, ev_fork, 0) != hipSuccess)
            return bigc_fail(BIGC_ERR_HIP, "train_backward: stream fork failed");
        int rc = bigc_linear_dw_group((void*)side, pending.data(), (int)pending.size());
        pending.clear();
        if (rc != BIGC_OK) return rc;
        if (hipEventRecord(ev_done[set], side) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "train_backward: event record failed");
        busy[set] = true;
        return BIGC_OK;
    }
    int wait_set(int set) {
        if (!busy[set]) return BIGC_OK;
        busy[set] = false;
        if (hipStreamWaitEvent((hipStream_t)s, ev_done[set], 0) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "train_backward: stream join failed");
        return BIGC_OK;
    }
    int join() { int rc = wait_set(0); if (rc == BIGC_OK) rc = wait_set(1); return rc; }
    int dw(const float* dY, int ldy, const float* X, int ldx, float* gw, int ldw, float* gb, int M, int N, int K) {
        if (!gw) return BIGC_OK;
        if (defer) { pending.push_back(BigcDwJob{dY, ldy, X, ldx, gw, ldw, gb, M, N, K}); return BIGC_OK; }
        return bigc_linear_dw_f32(s, dY, ldy, X, ldx, nullptr, 1, K, gw, ldw, gb, M, N, K);
    }
    int flush() {
        int rc = pending.empty() ? BIGC_OK : bigc_linear_dw_group(s, pending.data(), (int)pending.size());
        pending.clear();
        return rc;
    }
    int lin(const std::string& key, const Lin& L, const float* dY, int ldy, const float* X, int ldx, int M, int N, int K,
            float* dX, int lddx, bool accumulate_dx) {
        TRY(dw(dY, ldy, X, ldx, grad_of(m, key + ".weight"), K, grad_of(m, key + ".bias"), M, N, K));
        if (dX) TRY(dx(L.w, dY, ldy, M, N, K, dX, lddx, accumulate_dx));
        return BIGC_OK;
    }
    // as lin() with dX = (relu_out > 0) ? dY W : 0 -- the layer's input X is the output of a ReLU (X = relu_out), whose backward is
    // folded into the dX GEMM's epilogue instead of a pass of its own over dX (26 such passes per step, 0.9 ms at 64 videos)
    int lin_relu(const std::string& key, const Lin& L, const float* dY, int ldy, const float* X, int ldx, int M, int N, int K,
                 float* dX, int lddx) {
        TRY(dw(dY, ldy, X, ldx, grad_of(m, key + ".weight"), K, grad_of(m, key + ".bias"), M, N, K));
        const int n4 = round4(N);
        BIGC_CHECK_ARG(ldy % 4 == 0 && ldy >= n4, "train: dY leading dimension %d must be a multiple of 4 and >= %d", ldy, n4);
        const float* wT = nullptr;
        TRY(bigc_weight_transposed((hipStream_t)s, L.w, N, K, K, &wT));
        return bigc_linear_relu_masked(s, dY, ldy, wT, n4, X, ldx, dX, lddx, M, K, n4);
    }
    // as lin(), with dX = res + dY W written to a buffer of its own (res is left untouched)
    int lin_res(const std::string& key, const Lin& L, const float* dY, int ldy, const float* X, int ldx, int M, int N, int K,
                const float* res, float* dX, int lddx) {
        TRY(dw(dY, ldy, X, ldx, grad_of(m, key + ".weight"), K, grad_of(m, key + ".bias"), M, N, K));
        return dx_res(L.w, dY, ldy, M, N, K, res, dX, lddx);
    }
    int dx_res(const float* W, const float* dY, int ldy, int M, int N, int K, const float* res, float* dX, int lddx) {
        const int n4 = round4(N);
        BIGC_CHECK_ARG(ldy % 4 == 0 && ldy >= n4, "train: dY leading dimension %d must be a multiple of 4 and >= %d", ldy, n4);
        const float* wT = nullptr;
        TRY(bigc_weight_transposed((hipStream_t)s, W, N, K, K, &wT));
        return bigc_linear_f32(s, dY, ldy, nullptr, 1, n4, wT, n4, nullptr, res, lddx, dX, lddx, M, K, n4, BIGC_ACT_NONE);
    }
    // dX[M,K] (=|+=) dY[M,N] W[N,K] through a transposed, 4-padded copy of W
    // (`parameter`: W is a model parameter, constant for the weights epoch: its transposed copy and split form are cached and
    // refreshed in the step's batched preparation; other buffers are transposed per call)
    int dx(const float* W, const float* dY, int ldy, int M, int N, int K, float* dX, int lddx, bool accumulate, bool parameter = true) {
        const int n4 = round4(N);
        BIGC_CHECK_ARG(ldy % 4 == 0 && ldy >= n4, "train: dY leading dimension %d must be a multiple of 4 and >= %d", ldy, n4);
        if (parameter) {
            const float* wT = nullptr;
            TRY(bigc_weight_transposed((hipStream_t)s, W, N, K, K, &wT));
            return bigc_linear_f32(s, dY, ldy, nullptr, 1, n4, wT, n4, nullptr, accumulate ? dX : nullptr, lddx, dX, lddx, M, K, n4, BIGC_ACT_NONE);
        }
        TRY(bigc_transpose_pad((hipStream_t)s, W, w->wT, N, K, n4));
        StaticWeights transposed_copy(0);      // w->wT is rewritten for every layer
        return bigc_linear_f32(s, dY, ldy, nullptr, 1, n4, w->wT, n4, nullptr, accumulate ? dX : nullptr, lddx, dX, lddx, M, K, n4,
                               BIGC_ACT_NONE);
    }
};

}  // namespace


#define ZERO(ptr, count) do { hipError_t e_ = hipMemsetAsync((ptr), 0, sizeof(float) * (size_t)(count), hs); \
    if (e_ != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "train_backward: memset: %s", hipGetErrorString(e_)); } while (0)

// may a training step of this handle, in its current arithmetic mode, take a batch WITHOUT packed f32 feature rows and read the callers'
// rows in place (bigc_batch.feat_row_ptr)?  The split-operand kernels do (forward GEMM, weight gradient, the time-mean of the extra
// columns); the exact f32 kernels read a dense matrix.
extern "C" int bigc_train_reads_rows(const bigc_model* m) {
    if (!m || m->cfg.variant == 2) return 0;
    ModeScope mode_scope(m->mode_exact, m->mode_fast);
    if (bigc_linear_get_exact()) return 0;
    return (m->cfg.dim_feat % 4 == 0 && m->cfg.dim_extra % 4 == 0) ? 1 : 0;
}

// a1-a4 with every activation kept (models/model_0v10.py:391-458; the same front-end in model_0v7.py and
// model_pairwise_baseline.py:184-199): w.e2e = enti2enco
static int frontend_train_forward(bigc_model* m, void* s, const bigc_batch* b, TrainWork& w) {
    const bigc_config& c = m->cfg;
    const int R = b->n_rows, NT = b->n_trk, C = b->n_conv;
    const int de = c.dim_enti, D = c.dim_feat + c.dim_extra;
    TRY(bigc_bbox_feat_f32(s, b->boxes, b->trk_row_off, b->trk_vid, b->vid_wh, w.bbox8, NT));
    TRY(linear(s, w.bbox8, 8, m->bbox0, w.hb, de, R, de, 8, BIGC_ACT_RELU));
    TRY(linear(s, w.hb, de, m->bbox2, w.enti, 2 * de, R, de, de, BIGC_ACT_RELU));
    if (b->feats) {
        TRY(linear(s, b->feats, D, m->feat0, w.hv, de, R, de, c.dim_feat, BIGC_ACT_RELU));
    } else {                        // in place: the rows of the callers' per-tracklet tensors through the address table (as the eval forward)
        BIGC_CHECK_ARG(b->feat_row_ptr && bigc_train_reads_rows(m), "train: a batch without f32 features (bigc_batch.feats) needs bigc_batch.feat_row_ptr "
                       "and a mode / configuration that reads rows in place (bigc_train_reads_rows)");
        TRY(bigc_linear_rows_f32(s, b->feat_row_ptr, m->feat0.w, c.dim_feat, m->feat0.b, w.hv, de, R, de, c.dim_feat, BIGC_ACT_RELU));
    }
    TRY(linear(s, w.hv, de, m->feat2, w.enti + de, 2 * de, R, de, de, BIGC_ACT_RELU));
    TRY(bigc_conv_index(s, b->trk_row_off, b->trk_conv_off, b->trk_vid, b->vid_tmax, w.conv_idx, NT));
    TRY(bigc_linear_f32(s, w.enti, 2 * de, w.conv_idx, 3, 2 * de, m->conv_w_tapmajor, 6 * de, m->conv.b, nullptr, 0, w.conv, de, C, de,
                        6 * de, BIGC_ACT_NONE));
    TRY(bigc_adaptive_maxpool(s, w.conv, b->trk_conv_off, w.pooled, NT, de, c.enco_pool_len));
    TRY(linear(s, w.pooled, de * c.enco_pool_len, m->enco0, w.eh, de, NT, de, de * c.enco_pool_len, BIGC_ACT_RELU));
    TRY(linear(s, w.eh, de, m->enco2, w.e2e, de, NT, de, de, BIGC_ACT_RELU));
    return BIGC_OK;
}

// adjoint of frontend_train_forward: gx = d enti2enco (overwritten), parameter gradients accumulated
static int frontend_train_backward(bigc_model* m, void* s, const bigc_batch* b, TrainWork& w, Bwd& bw, float* gx) {
    const bigc_config& c = m->cfg;
    hipStream_t hs = (hipStream_t)s;
    const int R = b->n_rows, NT = b->n_trk, C = b->n_conv;
    const int de = c.dim_enti, D = c.dim_feat + c.dim_extra;
    // ---- fc_enti2enco, pool, conv, per-frame MLPs (models/model_0v10.py:446-458) ----------------------
    TRY(bigc_relu_bwd_f32(s, gx, w.e2e, (long)NT * de));
    float* geh = w.gx;
    TRY(bw.lin("fc_enti2enco.2", m->enco2, gx, de, w.eh, de, NT, de, de, geh, de, false));
    TRY(bigc_relu_bwd_f32(s, geh, w.eh, (long)NT * de));
    TRY(bw.lin("fc_enti2enco.0", m->enco0, geh, de, w.pooled, de * c.enco_pool_len, NT, de, de * c.enco_pool_len, w.gpooled,
               de * c.enco_pool_len, false));
    TRY(bigc_maxpool_bwd_f32(s, w.conv, b->trk_conv_off, w.gpooled, w.gconv, NT, de, c.enco_pool_len));
    TRY(bigc_conv_gather_bwd_f32(s, w.gconv, b->trk_row_off, b->trk_conv_off, b->trk_vid, b->vid_tmax, w.G, NT, de));
    static const bool conv_dw_gathered = getenv("BIGC_CONV_DW_GATHERED") && atoi(getenv("BIGC_CONV_DW_GATHERED")) != 0;
    if (float* gcw = grad_of(m, "conv_feat2enti.weight")) {
        ZERO(w.gconvw, (size_t)de * 6 * de);
        if (conv_dw_gathered || bigc_linear_get_exact()) {
            TRY(bigc_linear_dw_f32(s, w.gconv, de, w.enti, 2 * de, w.conv_idx, 3, 2 * de, w.gconvw, 6 * de, grad_of(m, "conv_feat2enti.bias"),
                                   C, de, 6 * de));
            TRY(bigc_conv_grad_unpack(hs, w.gconvw, gcw, de, 2 * de, 3));
        } else {
            // dW[o, c, k] = sum_j dconv[j, o] x[src(j, k), c] regrouped by SOURCE row: = sum_r G[r, k, o] x[r, c] with G the scatter of dconv
            // that the dX product below needs anyway -- a plain (3 de) x (2 de) weight-gradient product over the R packed rows instead of
            // a gathered one over the conv rows (the gathered kernel pays per-lane index arithmetic and selects: 80 TFLOP/s against 250)
            TRY(bigc_linear_dw_f32(s, w.G, 3 * de, w.enti, 2 * de, nullptr, 1, 2 * de, w.gconvw, 2 * de, nullptr, R, 3 * de, 2 * de));
            TRY(bigc_conv_grad_unpack_tapout(hs, w.gconvw, gcw, de, 2 * de, 3));
            if (float* gcb = grad_of(m, "conv_feat2enti.bias")) TRY(bigc_col_sum(hs, w.gconv, gcb, C, de));
        }
    }
    if (!m->conv_w_bwd) {       // first backward of this model: later steps find it repacked by bigc_finalize (before the re-split)
        hipError_t e = hipMalloc((void**)&m->conv_w_bwd, sizeof(float) * (size_t)2 * de * 3 * de);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "train_backward: hipMalloc: %s", hipGetErrorString(e));
        TRY(bigc_repack_conv_weight_bwd(hs, m->conv.w, m->conv_w_bwd, de, 2 * de, 3));
    }
    TRY(bigc_linear_relu_masked(s, w.G, 3 * de, m->conv_w_bwd, 3 * de, w.enti, 2 * de, w.genti, 2 * de, R, 2 * de, 3 * de));   // (+ ReLU of enti)
    TRY(bw.lin_relu("fc_bbox2enti.2", m->bbox2, w.genti, 2 * de, w.hb, de, R, de, de, w.gh, de));
    TRY(bw.lin("fc_bbox2enti.0", m->bbox0, w.gh, de, w.bbox8, 8, R, de, 8, nullptr, 0, false));
    TRY(bw.lin_relu("fc_feat2enti.2", m->feat2, w.genti + de, 2 * de, w.hv, de, R, de, de, w.gh, de));
    if (b->feats) {
        TRY(bw.lin("fc_feat2enti.0", m->feat0, w.gh, de, b->feats, D, R, de, c.dim_feat, nullptr, 0, false));
    } else if (float* gw = grad_of(m, "fc_feat2enti.0.weight")) {
        TRY(bw.flush());
        TRY(bigc_linear_dw_rows(s, w.gh, de, b->feat_row_ptr, gw, c.dim_feat, grad_of(m, "fc_feat2enti.0.bias"), R, de, c.dim_feat));
    }
    return BIGC_OK;
}

extern "C" int bigc_set_dropout(bigc_model* m, float p, uint32_t seed) {
    BIGC_CHECK_ARG(m && p >= 0.f && p < 1.f, "set_dropout: p must be in [0, 1)");
    m->drop_p = p; m->drop_seed = seed;
    return BIGC_OK;
}

extern "C" int bigc_set_grad(bigc_model* m, const char* key, float* ptr) {
    BIGC_CHECK_ARG(m && key, "set_grad: bad argument");
    if (ptr) m->grads[key] = ptr; else m->grads.erase(key);
    return BIGC_OK;
}

extern "C" int bigc_rebase_grads(bigc_model* m, float* old_base, float* new_base, int64_t numel) {
    BIGC_CHECK_ARG(m && old_base && new_base && numel > 0, "rebase_grads: bad argument");
    for (auto& kv : m->grads) {
        BIGC_CHECK_ARG(kv.second >= old_base && kv.second < old_base + numel, "rebase_grads: the destination of '%s' is not inside the old buffer", kv.first.c_str());
        kv.second = new_base + (kv.second - old_base);
    }
    return BIGC_OK;
}

extern "C" size_t bigc_train_workspace_bytes(const bigc_model* m, const bigc_batch* b) {
    if (!m || !b) return 0;
    Arena a{nullptr, (size_t)-1};
    TrainWork w;
    train_layout(m, tsizes(m, b), a, w);
    return a.off + 256;
}

static int open_arena(const bigc_model* m, const bigc_batch* b, void* workspace, size_t workspace_bytes, TrainWork& w) {
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (!workspace || workspace_bytes < (size_t)(base - (char*)workspace)) return bigc_fail(BIGC_ERR_WORKSPACE, "train: no workspace");
    Arena a{base, workspace_bytes - (size_t)(base - (char*)workspace)};
    if (!train_layout(m, tsizes(m, b), a, w))
        return bigc_fail(BIGC_ERR_WORKSPACE, "train: workspace of %zu bytes too small, need %zu", workspace_bytes, bigc_train_workspace_bytes(m, b));
    return BIGC_OK;
}

extern "C" int bigc_train_forward(bigc_model* m, void* stream, const bigc_batch* b, void* workspace, size_t workspace_bytes,
                                  float* pred_logits, float* att_out) {
    SplitKScope splitk;          // small-grid GEMMs of the step may split K (linear_internal.h)
    StaticWeights static_weights(m ? m->epoch : 0);
    ModeScope mode_scope(m ? m->mode_exact : -1, m ? m->mode_fast : -1);
    SplitRecorder recorder(m ? &m->train_jobs : nullptr);
    BIGC_CHECK_ARG(m && b && pred_logits && att_out, "train_forward: null pointer");
    BIGC_CHECK_ARG(m->finalized, "train_forward: call bigc_finalize first");
    BIGC_CHECK_ARG(b->n_vid > 0 && b->n_trk > 0 && b->n_rows > 0 && b->n_conv > 0 && b->max_trk > 0 && b->max_trk <= 512, "train_forward: bad batch");
    const bigc_config& c = m->cfg;
    TrainWork w;
    TRY(open_arena(m, b, workspace, workspace_bytes, w));
    void* s = stream;
    hipStream_t hs = (hipStream_t)stream;
    TRY(bigc_linear_prepare(hs, m->train_jobs.data(), (int)m->train_jobs.size()));     // every weight changed since the last step: one batched re-split
    const int NT = b->n_trk, B = b->n_vid, Q = c.num_querys, BQ = B * Q;
    const int de = c.dim_enti, dp = c.dim_pred, da = c.dim_att, df = c.dim_ffn, H = c.n_att_head, hd = de / H;
    const float eps = 1e-5f;
    const float dp_ = m->drop_p; const uint32_t ds_ = m->drop_seed;      // training-mode dropout (0 = off)

    TRY(frontend_train_forward(m, s, b, w));

    const float* x = w.e2e;
    for (int i = 0; i < c.n_enco_layers; ++i) {
        const EncLayer& L = m->enc[i];
        EncS& S = w.enc[i];
        TRY(linear(s, x, de, L.in_proj, S.qkv, 3 * de, NT, 3 * de, de, BIGC_ACT_NONE));
        TRY(bigc_mha_core_drop_f32(s, S.qkv, 3 * de, S.qkv + de, 3 * de, S.qkv + 2 * de, 3 * de, S.ao, de, b->vid_trk_off, b->vid_trk_off, B,
                                   b->max_trk, H, hd, S.lse, dp_, ds_, enc_site(i, 0)));
        if (dp_ > 0.f) {            // src + dropout1(out_proj(attn)) (:111)
            TRY(linear(s, S.ao, de, L.out_proj, S.t1, de, NT, de, de, BIGC_ACT_NONE));
            TRY(bigc_dropout_f32(s, S.t1, NT, de, de, dp_, ds_, enc_site(i, 1)));
            TRY(bigc_axpy_f32(s, S.t1, x, (long)NT * de));
        } else {
            TRY(linear(s, S.ao, de, L.out_proj, S.t1, de, NT, de, de, BIGC_ACT_NONE, x, de));
        }
        TRY(bigc_layernorm_f32(s, S.t1, L.n1.g, L.n1.b, S.x1, NT, de, eps));
        TRY(linear(s, S.x1, de, L.l1, S.fh, df, NT, df, de, BIGC_ACT_RELU));
        if (dp_ > 0.f) {            // linear2(dropout(relu(linear1))) then src + dropout2(.) (:113-114)
            TRY(bigc_dropout_f32(s, S.fh, NT, df, df, dp_, ds_, enc_site(i, 2)));
            TRY(linear(s, S.fh, df, L.l2, S.t1b, de, NT, de, df, BIGC_ACT_NONE));
            TRY(bigc_dropout_f32(s, S.t1b, NT, de, de, dp_, ds_, enc_site(i, 3)));
            TRY(bigc_axpy_f32(s, S.t1b, S.x1, (long)NT * de));
        } else {
            TRY(linear(s, S.fh, df, L.l2, S.t1b, de, NT, de, df, BIGC_ACT_NONE, S.x1, de));
        }
        TRY(bigc_layernorm_f32(s, S.t1b, L.n2.g, L.n2.b, S.out, NT, de, eps));
        x = S.out;
    }
    const float* enco = x;

    TRY(bigc_fill_offsets(hs, w.q_off, B + 1, Q));
    const float att_scale = 1.0f / sqrtf((float)de);
    const float* qprev = nullptr;
    for (int i = 0; i < c.n_deco_layers; ++i) {
        const DecLayer& L = m->dec[i];
        DecS& S = w.dec[i];
        Lin qkp{L.in_proj.w, L.in_proj.b};
        Lin vp{L.in_proj.w + (size_t)2 * dp * dp, L.in_proj.b + 2 * dp};
        const int rows = (i == 0) ? Q : BQ, nseq = (i == 0) ? 1 : B;
        const float* qin = (i == 0) ? m->qinit : qprev;
        TRY(bigc_add_rows_f32(s, qin, m->pos, S.qk_in, rows, dp, Q));
        TRY(linear(s, S.qk_in, dp, qkp, S.qk, 2 * dp, rows, 2 * dp, dp, BIGC_ACT_NONE));
        TRY(linear(s, qin, dp, vp, S.vv, dp, rows, dp, dp, BIGC_ACT_NONE));
        TRY(bigc_mha_core_drop_f32(s, S.qk, 2 * dp, S.qk + dp, 2 * dp, S.vv, dp, S.ao2, dp, w.q_off, w.q_off, nseq, Q, H, dp / H, S.lse, dp_, ds_,
                                   dec_site(i, 0)));
        TRY(linear(s, S.ao2, dp, L.out_proj, S.t2, dp, rows, dp, dp, BIGC_ACT_NONE, qin, dp));
        TRY(bigc_layernorm_f32(s, S.t2, L.n1.g, L.n1.b, w.t3tmp, rows, dp, eps));
        if (i == 0) {
            TRY(bigc_add_rows_f32(s, w.t3tmp, m->pos, w.rtmp, Q, dp, Q));
            TRY(bigc_add_rows_f32(s, nullptr, w.rtmp, S.q2, BQ, dp, Q));
        } else {
            TRY(bigc_add_rows_f32(s, w.t3tmp, m->pos, S.q2, BQ, dp, Q));
        }
        TRY(linear(s, enco, de, L.e2a, S.e2a, da, NT, da, de, BIGC_ACT_NONE));
        TRY(linear(s, S.q2, dp, L.p2a, S.p2a, da, BQ, da, dp, BIGC_ACT_NONE));
        // role attention and its values: videos of at most 64 tracklets take the eval path's one-launch kernel (csrc/attention_split.hip,
        // split-operand f16 MFMAs, att written for the backward: 15 us where the two f32-MFMA launches took 66 + 18 for 8 videos)
        if (b->max_trk <= 64 && (da / 2) % 64 == 0 && !bigc_linear_get_exact()) {
            TRY(bigc_role_fused_f32(s, S.p2a, da, S.e2a, da, enco, de, S.att, S.vals, b->vid_trk_off, B, Q, b->max_trk, da / 2, de, att_scale));
        } else {
            TRY(bigc_role_attention_f32(s, S.p2a, da, S.e2a, da, S.att, b->vid_trk_off, B, Q, da / 2, att_scale));
            TRY(bigc_role_values_f32(s, S.att, enco, de, S.vals, b->vid_trk_off, B, Q, de));
        }
        TRY(linear(s, S.vals, de, L.role0[0], S.rh0, dp, BQ, dp, de, BIGC_ACT_RELU));
        TRY(linear(s, S.rh0, dp, L.role0[1], w.rtmp, dp, BQ, dp, dp, BIGC_ACT_NONE, S.q2, dp));
        TRY(linear(s, S.vals + (size_t)BQ * de, de, L.role1[0], S.rh1, dp, BQ, dp, de, BIGC_ACT_RELU));
        TRY(linear(s, S.rh1, dp, L.role1[1], S.t3b, dp, BQ, dp, dp, BIGC_ACT_NONE, w.rtmp, dp));
        TRY(bigc_layernorm_f32(s, S.t3b, L.n2.g, L.n2.b, S.q3, BQ, dp, eps));
        TRY(linear(s, S.q3, dp, L.f0, S.f, df, BQ, df, dp, BIGC_ACT_RELU));
        if (dp_ > 0.f) TRY(bigc_dropout_f32(s, S.f, BQ, df, df, dp_, ds_, dec_site(i, 1)));            // fc2's Dropout (:167)
        TRY(linear(s, S.f, df, L.f3, S.t3c, dp, BQ, dp, df, BIGC_ACT_NONE, S.q3, dp));
        TRY(bigc_layernorm_f32(s, S.t3c, L.n3.g, L.n3.b, S.qout, BQ, dp, eps));
        qprev = S.qout;
    }
    const DecS& last = w.dec[c.n_deco_layers - 1];

    const float* seg_src[4];
    int32_t seg_w[4], seg_cat[4];
    int n_seg = 0;
    if (c.variant == 0) {
        if (c.dim_extra > 0) {
            TRY(bigc_extra_mean(s, b, c, w.xmean, NT));
            TRY(linear(s, w.xmean, c.dim_extra, m->i3d, w.i3dfc, de, NT, de, c.dim_extra, BIGC_ACT_RELU));
            seg_src[n_seg] = w.i3dfc; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
            seg_src[n_seg] = w.e2e; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
            seg_src[n_seg] = m->emb; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 1;
        } else {
            seg_src[n_seg] = m->emb; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 1;
            seg_src[n_seg] = w.e2e; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
        }
    } else {
        if (c.use_clsme) {
            if (c.has_enti_emb) {
                seg_src[n_seg] = m->emb; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 1;
            } else {
                TRY(bigc_extra_mean(s, b, c, w.xmean, NT));
                seg_src[n_seg] = w.xmean; seg_w[n_seg] = c.dim_clsme; seg_cat[n_seg++] = 0;
            }
        }
        seg_src[n_seg] = w.e2e; seg_w[n_seg] = de; seg_cat[n_seg++] = 0;
    }
    TRY(bigc_head_gather(s, last.att, b->vid_trk_off, b->trk_cat, B, Q, last.qout, dp, n_seg, seg_src, seg_w, seg_cat, m->bias_matrix,
                         c.num_enti_cats, c.num_pred_cats, w.soid, w.head_in, m->head_k, w.bias_g));
    if (c.variant == 0) {
        TRY(linear(s, w.head_in, m->head_k, m->head0, pred_logits, c.num_pred_cats, BQ, c.num_pred_cats, m->head_k, BIGC_ACT_NONE, w.bias_g,
                   c.num_pred_cats));
    } else {
        TRY(linear(s, w.head_in, m->head_k, m->head0, w.hh, df, BQ, df, m->head_k, BIGC_ACT_RELU));
        TRY(linear(s, w.hh, df, m->head2, pred_logits, c.num_pred_cats, BQ, c.num_pred_cats, df, BIGC_ACT_NONE, w.bias_g, c.num_pred_cats));
    }
    hipError_t e = hipMemcpyAsync(att_out, last.att, sizeof(float) * (size_t)2 * Q * NT, hipMemcpyDeviceToDevice, hs);
    if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "train_forward: memcpy: %s", hipGetErrorString(e));
    return BIGC_OK;
}

extern "C" int bigc_train_backward(bigc_model* m, void* stream, const bigc_batch* b, void* workspace, size_t workspace_bytes,
                                   const float* dlogits, int ldg, const float* datt) {
    SplitKScope splitk;          // small-grid GEMMs of the step may split K (linear_internal.h)
    FinishScope finishes((hipStream_t)stream);   // the LayerNorm parameter gradients' partial rows are added by ONE launch when the pass is over (backward.hip)
    StaticWeights static_weights(m ? m->epoch : 0);
    ModeScope mode_scope(m ? m->mode_exact : -1, m ? m->mode_fast : -1);
    SplitRecorder recorder(m ? &m->train_jobs : nullptr);
    BIGC_CHECK_ARG(m && b && dlogits && datt, "train_backward: null pointer");
    const bigc_config& c = m->cfg;
    BIGC_CHECK_ARG(ldg == round4(c.num_pred_cats), "train_backward: dlogits must have leading dimension %d (zero padded)", round4(c.num_pred_cats));
    TrainWork w;
    TRY(open_arena(m, b, workspace, workspace_bytes, w));
    void* s = stream;
    hipStream_t hs = (hipStream_t)stream;
    const int NT = b->n_trk, B = b->n_vid, Q = c.num_querys, BQ = B * Q;
    const int de = c.dim_enti, dp = c.dim_pred, da = c.dim_att, df = c.dim_ffn, H = c.n_att_head, hd = de / H;
    const int P = c.num_pred_cats, HK = m->head_k;
    const float eps = 1e-5f;
    const float dp_ = m->drop_p; const uint32_t ds_ = m->drop_seed;      // training-mode dropout (0 = off)
    Bwd bw{m, s, &w};
    const float att_scale = 1.0f / sqrtf((float)de);

    // ---- head (models/model_0v10.py:478-507 / model_0v7.py:483-513) ---------------------------------
    if (float* gbm = grad_of(m, "bias_matrix"))
        TRY(bigc_bias_matrix_bwd(hs, dlogits, ldg, w.soid, b->vid_trk_off, b->trk_cat, B, Q, c.num_enti_cats, P, gbm));
    if (c.variant == 0) {
        TRY(bw.lin("fc_pred2logits", m->head0, dlogits, ldg, w.head_in, HK, BQ, P, HK, w.ghead, HK, false));
    } else {
        TRY(bw.lin("fc_pred2logits.2", m->head2, dlogits, ldg, w.hh, df, BQ, P, df, w.ghh, df, false));
        TRY(bigc_relu_bwd_f32(s, w.ghh, w.hh, (long)BQ * df));
        TRY(bw.lin("fc_pred2logits.0", m->head0, w.ghh, df, w.head_in, HK, BQ, df, HK, w.ghead, HK, false));
    }
    TRY(bigc_copy_cols(hs, w.ghead, HK, w.gq, dp, BQ, dp));                   // d query of the last decoder layer
    ZERO(w.ge2e, (size_t)NT * de);
    {
        float* dst[4]; int width[4], col[4]; int ns = 0; int colp = dp;
        const bool i3d = c.variant == 0 && c.dim_extra > 0;
        if (i3d) { ZERO(w.gi3d, (size_t)NT * de); dst[ns] = w.gi3d; width[ns] = de; col[ns++] = colp; colp += 2 * de; }
        if (c.variant == 0) {
            if (i3d) { dst[ns] = w.ge2e; width[ns] = de; col[ns++] = colp; }
            else { colp += 2 * c.dim_clsme; dst[ns] = w.ge2e; width[ns] = de; col[ns++] = colp; }
        } else {
            if (c.use_clsme) colp += 2 * c.dim_clsme;
            dst[ns] = w.ge2e; width[ns] = de; col[ns++] = colp;
        }
        TRY(bigc_head_gather_bwd(hs, w.ghead, HK, w.soid, b->vid_trk_off, b->trk_vid, NT, Q, ns, dst, width, col));
        if (i3d) {
            TRY(bigc_relu_bwd_f32(s, w.gi3d, w.i3dfc, (long)NT * de));
            TRY(bw.lin("fc_i3d.0", m->i3d, w.gi3d, de, w.xmean, c.dim_extra, NT, de, c.dim_extra, nullptr, 0, false));
        }
    }

    // ---- decoder, last layer first (models/model_0v10.py:178-225) --------------------------------------
    ZERO(w.genco, (size_t)NT * de);
    float* gpos = grad_of(m, "pos_embedding");
    // Every gradient that a weight-gradient product of the layer reads has a buffer of its own (nothing is accumulated in place into
    // a buffer such a product still needs), so the layer's eleven products run as ONE grouped launch at its end (bw.flush():
    // 224 tiles with the whole B*Q contraction split in two, instead of eleven launches of 16 tiles x 32 splits + eleven reductions).
    static const bool group_off = getenv("BIGC_DW_GROUP") && atoi(getenv("BIGC_DW_GROUP")) == 0;
    bw.defer = !group_off && !bigc_linear_get_exact();
    // (measured: 64 videos 23.9 -> 23.5 ms per step; at 8 videos the forks and joins cost more than the overlap returns, 8.4 -> 8.7 ms: big
    // batches only; BIGC_DW_STREAM=0 / 1 forces it off / on)
    static const int dw_stream_env = getenv("BIGC_DW_STREAM") ? atoi(getenv("BIGC_DW_STREAM")) : -1;
    const bool dw_side = bw.defer && (dw_stream_env >= 0 ? dw_stream_env != 0 : BQ >= 32 * 192);
    if (dw_side) {
        if (!m->dw_stream) {
            hipStream_t st = nullptr; hipEvent_t e[3] = {nullptr, nullptr, nullptr};
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "train_backward: could not create a stream");
            for (int j = 0; j < 3; ++j)
                if (hipEventCreateWithFlags(&e[j], hipEventDisableTiming) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "train_backward: could not create an event");
            m->dw_stream = st; for (int j = 0; j < 3; ++j) m->dw_ev[j] = e[j];
        }
        bw.side = (hipStream_t)m->dw_stream; bw.ev_fork = (hipEvent_t)m->dw_ev[0]; bw.ev_done[0] = (hipEvent_t)m->dw_ev[1]; bw.ev_done[1] = (hipEvent_t)m->dw_ev[2];
    }
    for (int i = c.n_deco_layers - 1; i >= 0; --i) {
        const DecLayer& L = m->dec[i];
        DecS& S = w.dec[i];
        const std::string p = "decoder_layers." + std::to_string(i);
        const int set = dw_side ? ((c.n_deco_layers - 1 - i) & 1) : 0;
        if (dw_side) TRY(bw.wait_set(set));                  // the products of the layer before last have read this set
        float *A1 = w.ga, *B1 = w.gb, *A2 = w.gd[0], *B2 = w.gd[1], *C1 = w.gc, *C0 = w.gd[2], *B3 = w.gd[3], *C2 = w.gd[4], *A3 = w.gd[5],
              *V1 = w.gd[6], *A4 = w.gd[7];
        float *gqk = w.gqk, *gp2a = w.gp2a, *ge2a = w.ge2a;
        if (set) { A1 = w.alt[0]; B1 = w.alt[1]; C1 = w.alt[2]; B2 = w.alt[3]; C0 = w.alt[4]; A3 = w.alt[5]; V1 = w.alt[6]; gqk = w.alt[7]; gp2a = w.alt[8]; ge2a = w.alt[9]; }
        // w.gq = d qout
        TRY(bigc_layernorm_bwd_f32(s, S.t3c, L.n3.g, w.gq, A1, grad_of(m, p + ".norm3.weight"), grad_of(m, p + ".norm3.bias"), BQ, dp, eps));
        // A1 = d t3c ; t3c = f W3^T + b + q3
        TRY(bw.lin_relu(p + ".fc2.3", L.f3, A1, dp, S.f, df, BQ, dp, df, B1, df));             // B1 = d f (S.f is post-dropout: zero where dropped)
        if (dp_ > 0.f) TRY(bigc_scale_f32(hs, B1, (long)BQ * df, 1.0f / (1.0f - dp_)));
        TRY(bw.lin_res(p + ".fc2.0", L.f0, B1, df, S.q3, dp, BQ, df, dp, A1, A2, dp));         // A2 = d q3 (residual + ffn path)
        TRY(bigc_layernorm_bwd_f32(s, S.t3b, L.n2.g, A2, B2, grad_of(m, p + ".norm2.weight"), grad_of(m, p + ".norm2.bias"), BQ, dp, eps));
        // B2 = d t3b = d(rh1 W^T + b1 + rh0 W^T + b0 + q2)
        TRY(bw.lin_relu(p + ".fc_rolewise.1.2", L.role1[1], B2, dp, S.rh1, dp, BQ, dp, dp, C1, dp));          // C1 = d rh1
        TRY(bw.lin(p + ".fc_rolewise.1.0", L.role1[0], C1, dp, S.vals + (size_t)BQ * de, de, BQ, dp, de, w.gvals + (size_t)BQ * de, de, false));
        TRY(bw.lin_relu(p + ".fc_rolewise.0.2", L.role0[1], B2, dp, S.rh0, dp, BQ, dp, dp, C0, dp));          // C0 = d rh0
        TRY(bw.lin(p + ".fc_rolewise.0.0", L.role0[0], C0, dp, S.vals, de, BQ, dp, de, w.gvals, de, false));
        // B2 stays = d q2 (residual path); values = att @ enco
        const float* enco = w.enc[c.n_enco_layers - 1].out;
        // d att = d vals @ enco^T and d enco += att^T @ d vals in one launch, then d p2a = ds @ e2a and d e2a = ds^T @ p2a in another
        TRY(bigc_role_small_gemm_pair_f32(s, 0, S.att, w.gvals, de, enco, de, w.gatt, w.genco, de, b->vid_trk_off, B, Q, de, da / 2, b->max_trk));
        if (i == c.n_deco_layers - 1) TRY(bigc_axpy_f32(s, w.gatt, datt, (long)2 * Q * NT));
        TRY(bigc_role_attention_bwd_launch(s, S.p2a, da, S.e2a, da, w.gatt, w.ds, b->vid_trk_off, B, Q, da / 2, att_scale, b->max_trk));
        TRY(bigc_role_small_gemm_pair_f32(s, 1, w.ds, S.p2a, da, S.e2a, da, gp2a, ge2a, da, b->vid_trk_off, B, Q, de, da / 2, b->max_trk));
        TRY(bw.lin_res(p + ".fc_pred2att", L.p2a, gp2a, da, S.q2, dp, BQ, da, dp, B2, B3, dp));             // B3 = d q2 total
        TRY(bw.lin(p + ".fc_enti2att", L.e2a, ge2a, da, enco, de, NT, da, de, w.genco, de, true));
        // q2 = LN1(t2) + pos
        if (gpos) TRY(bigc_reduce_periodic_f32(s, B3, gpos, BQ, dp, Q));
        const int rows = (i == 0) ? Q : BQ, nseq = (i == 0) ? 1 : B;
        const float* gt3 = B3;
        if (i == 0) {       // the self-attention block was evaluated once on Q rows and tiled: sum the gradient over the videos
            ZERO(C2, (size_t)Q * dp);
            TRY(bigc_reduce_periodic_f32(s, B3, C2, BQ, dp, Q));
            gt3 = C2;
        }
        TRY(bigc_layernorm_bwd_f32(s, S.t2, L.n1.g, gt3, A3, grad_of(m, p + ".norm1.weight"), grad_of(m, p + ".norm1.bias"), rows, dp, eps));
        // A3 = d t2 ; t2 = ao2 Wo^T + b + qin
        const float* qin = (i == 0) ? m->qinit : w.dec[i - 1].qout;
        TRY(bw.lin(p + ".self_attn.out_proj", L.out_proj, A3, dp, S.ao2, dp, rows, dp, dp, w.gq, dp, false));   // gq = d ao2 (d qout is consumed)
        TRY(bigc_mha_bwd_drop_f32(s, S.qk, 2 * dp, S.qk + dp, 2 * dp, S.vv, dp, S.ao2, dp, w.gq, dp, S.lse, w.delta, gqk, 2 * dp, gqk + dp,
                                  2 * dp, V1, dp, w.q_off, w.q_off, nseq, Q, Q, rows, H, dp / H, dp_, ds_, dec_site(i, 0)));
        // in_proj: [q;k] rows from qk_in, v rows from qin
        {
            float* gw = grad_of(m, p + ".self_attn.in_proj_weight");
            float* gbias = grad_of(m, p + ".self_attn.in_proj_bias");
            if (gw) {
                TRY(bw.dw(gqk, 2 * dp, S.qk_in, dp, gw, dp, gbias, rows, 2 * dp, dp));
                TRY(bw.dw(V1, dp, qin, dp, gw + (size_t)2 * dp * dp, dp, gbias ? gbias + 2 * dp : nullptr, rows, dp, dp));
            }
        }
        // d qin = d t2 (residual) + d vv Wv + d qk_in ; d qk_in = d qk Wqk ; d pos += d qk_in
        TRY(bw.dx_res(L.in_proj.w + (size_t)2 * dp * dp, V1, dp, rows, dp, dp, A3, A4, dp));
        TRY(bw.dx(L.in_proj.w, gqk, 2 * dp, rows, 2 * dp, dp, w.gq, dp, false));                         // gq = d qk_in (d ao2 is consumed)
        if (gpos) TRY(bigc_reduce_periodic_f32(s, w.gq, gpos, rows, dp, Q));
        TRY(bigc_axpy_f32(s, A4, w.gq, (long)rows * dp));                                                    // A4 = d qin
        if (dw_side) TRY(bw.flush_side(set)); else TRY(bw.flush());                                           // the layer's weight gradients
        if (i == 0) {
            if (float* gqi = grad_of(m, "pred_query_init")) TRY(bigc_reduce_periodic_f32(s, A4, gqi, Q, dp, Q));
        } else {
            float* t = w.gq; w.gq = w.gd[7]; w.gd[7] = t;                                                    // gq = d qout of layer i-1
        }
    }
    bw.defer = false;

    // ---- encoder (models/model_0v10.py:103-117) ---------------------------------------------------------
    float* gx = w.genco;        // d output of the current encoder layer (NT, de); scratch: w.gx, w.gy, w.gz
    for (int i = c.n_enco_layers - 1; i >= 0; --i) {
        const EncLayer& L = m->enc[i];
        EncS& S = w.enc[i];
        const std::string p = "encoder_layers." + std::to_string(i);
        const float* xin = (i == 0) ? w.e2e : w.enc[i - 1].out;
        TRY(bigc_layernorm_bwd_f32(s, S.t1b, L.n2.g, gx, w.gx, grad_of(m, p + ".norm2.weight"), grad_of(m, p + ".norm2.bias"), NT, de, eps));
        const float* g_l2 = w.gx;                                    // d of linear2's (pre-dropout) output
        if (dp_ > 0.f) {
            hipError_t e2 = hipMemcpyAsync(w.gz, w.gx, sizeof(float) * (size_t)NT * de, hipMemcpyDeviceToDevice, hs);
            if (e2 != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "train_backward: memcpy: %s", hipGetErrorString(e2));
            TRY(bigc_dropout_f32(s, w.gz, NT, de, de, dp_, ds_, enc_site(i, 3)));
            g_l2 = w.gz;
        }
        TRY(bw.lin_relu(p + ".linear2", L.l2, g_l2, de, S.fh, df, NT, de, df, w.gy, df));            // gy = d fh
        if (dp_ > 0.f) TRY(bigc_scale_f32(hs, w.gy, (long)NT * df, 1.0f / (1.0f - dp_)));
        TRY(bw.lin(p + ".linear1", L.l1, w.gy, df, S.x1, de, NT, df, de, w.gx, de, true));           // gx = d x1
        TRY(bigc_layernorm_bwd_f32(s, S.t1, L.n1.g, w.gx, w.gy, grad_of(m, p + ".norm1.weight"), grad_of(m, p + ".norm1.bias"), NT, de, eps));
        // gy = d t1 = d(ao Wo^T + b + xin)
        const float* g_op = w.gy;                                    // d of out_proj's (pre-dropout) output
        if (dp_ > 0.f) {
            hipError_t e2 = hipMemcpyAsync(w.gx, w.gy, sizeof(float) * (size_t)NT * de, hipMemcpyDeviceToDevice, hs);
            if (e2 != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "train_backward: memcpy: %s", hipGetErrorString(e2));
            TRY(bigc_dropout_f32(s, w.gx, NT, de, de, dp_, ds_, enc_site(i, 1)));
            g_op = w.gx;
        }
        TRY(bw.lin(p + ".self_attn.out_proj", L.out_proj, g_op, de, S.ao, de, NT, de, de, w.gz, de, false));   // gz = d ao
        TRY(bigc_mha_bwd_drop_f32(s, S.qkv, 3 * de, S.qkv + de, 3 * de, S.qkv + 2 * de, 3 * de, S.ao, de, w.gz, de, S.lse, w.delta, w.gqkv, 3 * de,
                                  w.gqkv + de, 3 * de, w.gqkv + 2 * de, 3 * de, b->vid_trk_off, b->vid_trk_off, B, b->max_trk, b->max_trk, NT, H, hd,
                                  dp_, ds_, enc_site(i, 0)));
        {
            float* gw = grad_of(m, p + ".self_attn.in_proj_weight");
            float* gbias = grad_of(m, p + ".self_attn.in_proj_bias");
            if (gw) TRY(bigc_linear_dw_f32(s, w.gqkv, 3 * de, xin, de, nullptr, 1, de, gw, de, gbias, NT, 3 * de, de));
        }
        TRY(bw.dx(L.in_proj.w, w.gqkv, 3 * de, NT, 3 * de, de, w.gy, de, true));                      // gy = d xin
        hipError_t e = hipMemcpyAsync(w.genco, w.gy, sizeof(float) * (size_t)NT * de, hipMemcpyDeviceToDevice, hs);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "train_backward: memcpy: %s", hipGetErrorString(e));
    }
    TRY(bigc_axpy_f32(s, gx, w.ge2e, (long)NT * de));                                                  // + head gather path

    TRY(frontend_train_backward(m, s, b, w, bw, gx));
    TRY(bw.join());              // (the decoder's weight-gradient launches: before the LayerNorm finishes of FinishScope and the caller's optimizer)
    return BIGC_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// Training step of the pairwise baseline Base_C (models/model_pairwise_baseline.py:276-312): the front-end above, the
// prediction head (:236-273) on the GIVEN (subject, object) pairs of every video -- first layer factorised per tracklet and
// role as in bigc_pair_forward -- and its adjoint.  `pair_so` holds GLOBAL tracklet rows (video offset added by the host).
// ------------------------------------------------------------------------------------------------------------------
int bigc_pair_hidden_idx(hipStream_t s, const float* U, const float* V, const int32_t* pair_so, float* H, long n_pairs, int d);
int bigc_pair_bias_idx(hipStream_t s, const float* bias_matrix, const int32_t* trk_cat, const int32_t* pair_so, float* out, long n_pairs,
                       int n_cat, int n_pred);
int bigc_pair_scatter_add(hipStream_t s, const float* dH, const int32_t* pair_so, float* dU, float* dV, long n_pairs, int d);
int bigc_pair_bias_bwd(hipStream_t s, const float* dlogits, int ldg, const int32_t* trk_cat, const int32_t* pair_so, float* dbias, long n_pairs,
                       int n_cat, int n_pred);
int bigc_gather_rows(hipStream_t s, const float* src, const int32_t* idx, float* out, int n, int d);

namespace {
struct PairWork { float *U, *V, *clsme, *H, *bias, *dH, *dU, *dV, *wblk; };

bool pair_layout(const bigc_model* m, const bigc_batch* b, long n_pairs, Arena& a, TrainWork& w, PairWork& pw) {
    const bigc_config& c = m->cfg;
    const size_t NT = b->n_trk, df = c.dim_ffn, dc = c.use_clsme ? c.dim_clsme : 0, de = c.dim_enti;
    train_layout(m, tsizes(m, b), a, w);
    pw.U = a.take<float>(NT * df); pw.V = a.take<float>(NT * df);
    pw.clsme = a.take<float>(NT * (dc > 0 ? dc : 1));
    pw.H = a.take<float>((size_t)n_pairs * df);
    pw.bias = a.take<float>((size_t)n_pairs * c.num_pred_cats);
    pw.dH = a.take<float>((size_t)n_pairs * df);
    pw.dU = a.take<float>(NT * df); pw.dV = a.take<float>(NT * df);
    pw.wblk = a.take<float>(df * (de > dc ? de : dc));
    return !a.overflow;
}

int open_pair_arena(const bigc_model* m, const bigc_batch* b, long n_pairs, void* workspace, size_t workspace_bytes, TrainWork& w, PairWork& pw);
}  // namespace

extern "C" size_t bigc_pair_train_workspace_bytes(const bigc_model* m, const bigc_batch* b, long n_pairs) {
    if (!m || !b || n_pairs <= 0) return 0;
    Arena a{nullptr, (size_t)-1};
    TrainWork w;
    PairWork pw;
    pair_layout(m, b, n_pairs, a, w, pw);
    return a.off + 256;
}

namespace {
int open_pair_arena(const bigc_model* m, const bigc_batch* b, long n_pairs, void* workspace, size_t workspace_bytes, TrainWork& w, PairWork& pw) {
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    if (!workspace || workspace_bytes < (size_t)(base - (char*)workspace)) return bigc_fail(BIGC_ERR_WORKSPACE, "pair_train: no workspace");
    Arena a{base, workspace_bytes - (size_t)(base - (char*)workspace)};
    if (!pair_layout(m, b, n_pairs, a, w, pw))
        return bigc_fail(BIGC_ERR_WORKSPACE, "pair_train: workspace of %zu bytes too small, need %zu", workspace_bytes,
                         bigc_pair_train_workspace_bytes(m, b, n_pairs));
    return BIGC_OK;
}
}  // namespace

extern "C" int bigc_pair_train_forward(bigc_model* m, void* stream, const bigc_batch* b, const int32_t* pair_so, long n_pairs, void* workspace,
                                       size_t workspace_bytes, float* pair_logits) {
    SplitKScope splitk;          // small-grid GEMMs of the step may split K (linear_internal.h)
    StaticWeights static_weights(m ? m->epoch : 0);
    ModeScope mode_scope(m ? m->mode_exact : -1, m ? m->mode_fast : -1);
    SplitRecorder recorder(m ? &m->train_jobs : nullptr);
    BIGC_CHECK_ARG(m && b && pair_so && pair_logits, "pair_train_forward: null pointer");
    BIGC_CHECK_ARG(m->finalized && m->cfg.variant == 2, "pair_train_forward: needs a finalized pairwise-baseline model");
    BIGC_CHECK_ARG(b->n_vid > 0 && b->n_trk > 0 && b->n_rows > 0 && b->n_conv > 0 && n_pairs > 0, "pair_train_forward: empty batch");
    const bigc_config& c = m->cfg;
    TrainWork w;
    PairWork pw;
    TRY(open_pair_arena(m, b, n_pairs, workspace, workspace_bytes, w, pw));
    void* s = stream;
    hipStream_t hs = (hipStream_t)stream;
    TRY(bigc_linear_prepare(hs, m->train_jobs.data(), (int)m->train_jobs.size()));     // every weight changed since the last step: one batched re-split
    const int NT = b->n_trk, de = c.dim_enti, df = c.dim_ffn, dc = c.use_clsme ? c.dim_clsme : 0, hk = m->head_k, P = c.num_pred_cats;
    TRY(frontend_train_forward(m, s, b, w));
    const float* W1 = m->head0.w;
    if (dc > 0) {
        if (c.has_enti_emb) TRY(bigc_gather_rows(hs, m->emb, b->trk_cat, pw.clsme, NT, dc));                      // (:251-253)
        else TRY(bigc_extra_mean(s, b, c, pw.clsme, NT));                                                               // (:201)
        // column blocks of W1 (df, hk): [s_clsme | o_clsme | s_feat | o_feat] (:257)
        TRY(bigc_linear_f32(s, pw.clsme, dc, nullptr, 1, dc, W1, hk, m->head0.b, nullptr, 0, pw.U, df, NT, df, dc, BIGC_ACT_NONE));
        TRY(bigc_linear_f32(s, w.e2e, de, nullptr, 1, de, W1 + 2 * dc, hk, nullptr, pw.U, df, pw.U, df, NT, df, de, BIGC_ACT_NONE));
        TRY(bigc_linear_f32(s, pw.clsme, dc, nullptr, 1, dc, W1 + dc, hk, nullptr, nullptr, 0, pw.V, df, NT, df, dc, BIGC_ACT_NONE));
        TRY(bigc_linear_f32(s, w.e2e, de, nullptr, 1, de, W1 + 2 * dc + de, hk, nullptr, pw.V, df, pw.V, df, NT, df, de, BIGC_ACT_NONE));
    } else {
        TRY(bigc_linear_f32(s, w.e2e, de, nullptr, 1, de, W1, hk, m->head0.b, nullptr, 0, pw.U, df, NT, df, de, BIGC_ACT_NONE));
        TRY(bigc_linear_f32(s, w.e2e, de, nullptr, 1, de, W1 + de, hk, nullptr, nullptr, 0, pw.V, df, NT, df, de, BIGC_ACT_NONE));
    }
    TRY(bigc_pair_hidden_idx(hs, pw.U, pw.V, pair_so, pw.H, n_pairs, df));
    TRY(bigc_pair_bias_idx(hs, m->bias_matrix, b->trk_cat, pair_so, pw.bias, n_pairs, c.num_enti_cats, P));
    TRY(bigc_linear_f32(s, pw.H, df, nullptr, 1, df, m->head2.w, df, m->head2.b, pw.bias, P, pair_logits, P, (int)n_pairs, P, df, BIGC_ACT_NONE));
    return BIGC_OK;
}

extern "C" int bigc_pair_train_backward(bigc_model* m, void* stream, const bigc_batch* b, const int32_t* pair_so, long n_pairs, void* workspace,
                                        size_t workspace_bytes, const float* dlogits, int ldg) {
    SplitKScope splitk;          // small-grid GEMMs of the step may split K (linear_internal.h)
    StaticWeights static_weights(m ? m->epoch : 0);
    ModeScope mode_scope(m ? m->mode_exact : -1, m ? m->mode_fast : -1);
    SplitRecorder recorder(m ? &m->train_jobs : nullptr);
    BIGC_CHECK_ARG(m && b && pair_so && dlogits, "pair_train_backward: null pointer");
    BIGC_CHECK_ARG(m->finalized && m->cfg.variant == 2 && n_pairs > 0, "pair_train_backward: needs a finalized pairwise-baseline model");
    const bigc_config& c = m->cfg;
    BIGC_CHECK_ARG(ldg == round4(c.num_pred_cats), "pair_train_backward: dlogits must have leading dimension %d (zero padded)", round4(c.num_pred_cats));
    TrainWork w;
    PairWork pw;
    TRY(open_pair_arena(m, b, n_pairs, workspace, workspace_bytes, w, pw));
    void* s = stream;
    hipStream_t hs = (hipStream_t)stream;
    const int NT = b->n_trk, de = c.dim_enti, df = c.dim_ffn, dc = c.use_clsme ? c.dim_clsme : 0, hk = m->head_k, P = c.num_pred_cats;
    Bwd bw{m, s, &w};
    if (float* gbm = grad_of(m, "bias_matrix"))
        TRY(bigc_pair_bias_bwd(hs, dlogits, ldg, b->trk_cat, pair_so, gbm, n_pairs, c.num_enti_cats, P));
    // logits = H W2^T + b2 + bias: dW2, db2, dH; ReLU mask; scatter to the per-tracklet pre-activations U (subject) / V (object)
    TRY(bw.lin("fc_pred2logits.2", m->head2, dlogits, ldg, pw.H, df, (int)n_pairs, P, df, pw.dH, df, false));
    TRY(bigc_relu_bwd_f32(s, pw.dH, pw.H, n_pairs * (long)df));
    ZERO(pw.dU, (size_t)NT * df);
    ZERO(pw.dV, (size_t)NT * df);
    TRY(bigc_pair_scatter_add(hs, pw.dH, pair_so, pw.dU, pw.dV, n_pairs, df));
    // U = [clsme | e2e] W1[:, (0 | 2dc)]^T + b1,  V = [clsme | e2e] W1[:, (dc | 2dc+de)]^T: column blocks of dW1, db1 = colsum(dU)
    float* gw1 = grad_of(m, "fc_pred2logits.0.weight");
    float* gb1 = grad_of(m, "fc_pred2logits.0.bias");
    const int off_se = 2 * dc, off_oe = 2 * dc + de;
    if (gw1) {
        if (dc > 0) {
            TRY(bigc_linear_dw_f32(s, pw.dU, df, pw.clsme, dc, nullptr, 1, dc, gw1, hk, gb1, NT, df, dc));
            TRY(bigc_linear_dw_f32(s, pw.dV, df, pw.clsme, dc, nullptr, 1, dc, gw1 + dc, hk, nullptr, NT, df, dc));
            TRY(bigc_linear_dw_f32(s, pw.dU, df, w.e2e, de, nullptr, 1, de, gw1 + off_se, hk, nullptr, NT, df, de));
        } else {
            TRY(bigc_linear_dw_f32(s, pw.dU, df, w.e2e, de, nullptr, 1, de, gw1 + off_se, hk, gb1, NT, df, de));
        }
        TRY(bigc_linear_dw_f32(s, pw.dV, df, w.e2e, de, nullptr, 1, de, gw1 + off_oe, hk, nullptr, NT, df, de));
    }
    // d enti2enco = dU W1[:, s_feat] + dV W1[:, o_feat]   (the classemes are inputs / a frozen embedding: no gradient)
    float* gx = w.genco;            // not w.gx: frontend_train_backward uses that one as scratch
    TRY(bigc_copy_cols(hs, m->head0.w + off_se, hk, pw.wblk, de, df, de));
    TRY(bw.dx(pw.wblk, pw.dU, df, NT, df, de, gx, de, false, false));
    TRY(bigc_copy_cols(hs, m->head0.w + off_oe, hk, pw.wblk, de, df, de));
    TRY(bw.dx(pw.wblk, pw.dV, df, NT, df, de, gx, de, true, false));
    TRY(frontend_train_backward(m, s, b, w, bw, gx));
    return BIGC_OK;
}
