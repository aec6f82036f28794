// Internal definitions shared by model.hip (eval forward) and train.hip (training forward/backward).
#pragma once
#include "common.h"
#include "linear_internal.h"
#include <string>
#include <unordered_map>
#include <vector>

struct Lin { const float* w = nullptr; const float* b = nullptr; };
struct LNp { const float* g = nullptr; const float* b = nullptr; };
struct EncLayer { Lin in_proj, out_proj, l1, l2; LNp n1, n2; };
struct DecLayer { Lin in_proj, out_proj, role0[2], role1[2], e2a, p2a, f0, f3; LNp n1, n2, n3; };

struct bigc_model {
    bigc_config cfg;
    std::unordered_map<std::string, std::pair<const float*, int64_t>> params;
    std::unordered_map<std::string, float*> grads;      // gradient destination of each trainable parameter (training)
    bool finalized = false;
    float drop_p = 0.f; unsigned int drop_seed = 0;   // training-mode dropout (bigc_set_dropout); 0 = off
    unsigned long long epoch = 0;          // drawn at finalize: scope of the cached split weights (linear_internal.h)
    int mode_exact = -1, mode_fast = -1;   // this handle's arithmetic mode (bigc_set_mode); -1 = the process default
    // resolved
    const float *emb = nullptr, *pos = nullptr, *qinit = nullptr, *bias_matrix = nullptr;
    Lin feat0, feat2, bbox0, bbox2, conv, enco0, enco2, i3d, head0, head2;
    std::vector<EncLayer> enc;
    std::vector<DecLayer> dec;
    float* conv_w_tapmajor = nullptr;  // (de, 3*2de), owned
    unsigned long long role_cat_epoch = 0;   // weights epoch role_cat was built for (built lazily by the eval forward)
    float* role_cat = nullptr;         // per decoder layer: [W_role0.2 | W_role1.2] (dp, 2 dp) then b_role0.2 + b_role1.2 (dp), owned
    float* conv_w_bwd = nullptr;       // (2de, 3*de), owned (training)
    void* dw_stream = nullptr; void* dw_ev[3] = {nullptr, nullptr, nullptr};   // the decoder's grouped weight-gradient launches run beside the backward chain (train.hip), created at first use
    unsigned long long enco_cat_epoch = 0;
    float* enco_cat = nullptr;         // all decoder layers: [W_e2a; W_role0.0; W_role1.0] (n_layers (da + 2 dp), de), then the biases [b_e2a | 0] per layer, owned (model.hip: build_enco_cat)
    unsigned long long qkpos_epoch = 0;
    float* qkpos = nullptr;            // per decoder layer: (num_querys, 3 dp) rows [pos Wq^T | pos Wk^T | 0], owned (model.hip: build_qkpos)
    int head_k = 0;
    std::vector<SplitJob> train_jobs;  // (weight, shape) pairs the training step's GEMMs asked for: refreshed in one batch per step (linear_internal.h)
};

struct Arena {
    char* base; size_t cap; size_t off = 0; bool overflow = false;
    template <typename T> T* take(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (off + bytes > cap) { overflow = true; return nullptr; }
        T* p = (T*)((uintptr_t)base + off);       // (measuring runs have a null base: no pointer arithmetic on it)
        off += bytes;
        return p;
    }
};

#define TRY(expr) do { int rc_ = (expr); if (rc_ != BIGC_OK) return rc_; } while (0)

inline int linear(void* s, const float* A, int lda, const Lin& l, float* C, int ldc, int M, int N, int K, int act,
                  const float* res = nullptr, int ldr = 0) {
    return bigc_linear_f32(s, A, lda, nullptr, 1, K, l.w, K, l.b, res, ldr, C, ldc, M, N, K, act);
}

// time-mean of the dim_extra columns (I3D / classeme; models/model_0v10.py:470): they are a tensor of their own in a batch packed
// for the eval forward (bigc_batch.extra), else the tail of the f32 feature rows
inline int bigc_extra_mean(void* s, const bigc_batch* b, const bigc_config& c, float* out, int n_trk) {
    if (b->extra) return bigc_stretched_mean(s, b->extra, (int)b->ld_extra, 0, c.dim_extra, b->trk_row_off, b->trk_vid, b->vid_tmax, out, n_trk);
    if (!b->feats && b->feat_row_ptr && c.dim_feat % 4 == 0 && c.dim_extra % 4 == 0)         // in place, from the callers' rows
        return bigc_stretched_mean_rows(s, b->feat_row_ptr, c.dim_feat, c.dim_extra, b->trk_row_off, b->trk_vid, b->vid_tmax, out, n_trk);
    BIGC_CHECK_ARG(b->feats, "batch without f32 features (bigc_batch.feats) and without bigc_batch.extra");
    return bigc_stretched_mean(s, b->feats, c.dim_feat + c.dim_extra, c.dim_feat, c.dim_extra, b->trk_row_off, b->trk_vid, b->vid_tmax, out, n_trk);
}

int bigc_fill_offsets(hipStream_t s, int32_t* p, int n, int step);
int bigc_copy_cols(hipStream_t s, const float* src, int lds_, float* dst, int ldd, long rows, int cols);
extern "C" int bigc_axpy_f32(void* stream, float* y, const float* x, long n);
int bigc_head_pick(hipStream_t s, const float* att, const int32_t* e_off, const int32_t* trk_cat, int n_vid, int nq, const float* U,
                   const float* V, int lduv, const float* bias_matrix, int n_ecat, int n_pcat, int32_t* soid, float* acc);
int bigc_repack_conv_weight(hipStream_t s, const float* w, float* out, int n_out, int n_in, int taps);
int bigc_repack_conv_weight_bwd(hipStream_t s, const float* w, float* out, int n_out, int n_in, int taps);     // backward.hip
int bigc_conv_grad_unpack_tapout(hipStream_t s, const float* g_tapout, float* g_torch, int n_out, int n_in, int taps);   // backward.hip
int bigc_col_sum(hipStream_t s, const float* x, float* out, long rows, int d);                                          // backward.hip
