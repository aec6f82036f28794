// Definitions shared by grounding.hip (eval forward) and grounding_train.hip (training forward / backward).
#pragma once
#include "model_internal.h"

struct DW { const float *dw_w = nullptr, *dw_b = nullptr; Lin pw; };
struct Enc { DW conv[4]; Lin in_proj, out_proj, fc; Lin normb, norm_seq[4], norme; int ks = 7; std::string key; };
struct Head { DW layer[5]; int c_out = 0; std::string key; };

struct grd_model {
    grd_config cfg;
    std::unordered_map<std::string, std::pair<const float*, int64_t>> params;
    std::unordered_map<std::string, float*> grads;     // gradient destinations (training)
    bool finalized = false;
    float drop_p = 0.f; unsigned int drop_seed = 0;   // training-mode dropout of the QANet layers (grd_set_dropout); 0 = off
    unsigned long long epoch = 0;          // drawn at finalize: scope of the cached split weights
    int mode_exact = -1, mode_fast = -1;   // this handle's arithmetic mode (grd_set_mode); -1 = the process default
    const float *enti = nullptr, *pred = nullptr;
    Lin video_fc, vq_fc, proj2sim, query_fc, temp_fc;
    Enc video_enc, query_enc, comb_enc;
    Head cls_head, conf_head, regr_head;
    float* wq = nullptr;   // (d, kq) owned: [query_fc.weight | temp_fc.weight | 0 0]
    float* bq = nullptr;   // (d) owned: query_fc.bias + temp_fc.bias
    int kq = 0;
    void* side_stream = nullptr; void* ev_fork = nullptr; void* ev_join = nullptr;   // the query encoder's own stream beside the video encoder's (grd_forward), created at first use
    void* vq_frag = nullptr; unsigned long long vq_frag_epoch = 0;       // vq_fc's weight in fragment order (grd_coatt_fc), owned
    void* head_frag = nullptr; unsigned long long head_frag_epoch = 0;   // the heads' point-wise weights in MFMA fragment order (grounding_heads.hip), owned
    void* enc_frag[3] = {nullptr, nullptr, nullptr}; unsigned long long enc_frag_epoch[3] = {0, 0, 0};   // per encoder (video, query, combined): its nine 128 x 128 weights in fragment order (grounding_encoder.hip), owned
    std::vector<SplitJob> train_jobs;  // as bigc_model::train_jobs
};

// csrc/grounding_heads.hip: the three 5-layer prediction heads in one launch, intermediate layers in LDS
bool grd_head_chain_supported(int d, const Head& a, const Head& b, const Head& c);
size_t grd_head_frag_bytes();
int grd_head_chain(hipStream_t s, const float* x, long rows, int T, const int2* info, const Head* const heads[3], float* const outs[3], const int acts[3],
                   void* frag, bool refresh);

// csrc/grounding_encoder.hip: a QANet encoder layer around its attention core in two launches, the convolution stack's intermediates in LDS
bool grd_enc_chain_supported(int d, int ks);
size_t grd_enc_frag_bytes();
int grd_enc_chain(hipStream_t s, const Enc& E, const float* x, const float* pos, long rows, int T, const int2* info, void* frag, bool refresh,
                  float* out, float* ln_out, float* qkv_out = nullptr);   // qkv_out: (rows, 384) = in_proj(ln) written instead of ln_out
int grd_attn_tail(hipStream_t s, const Enc& E, const float* att, const float* res, long rows, void* frag, float* out);
// the co-attention and the fusion layer vq_fc in one launch: the (rows, 4 d) fusion tensor stays on the chip
bool grd_coatt_fc_supported(int d, int max_T);
size_t grd_vq_frag_bytes();
// (vp (total_T, d) = video @ vq_fc.weight[:, :d]^T + bias and qu (3 n, d) = query @ vq_fc.weight[:, d:2d]^T are the caller's two small GEMMs)
int grd_coatt_fc(hipStream_t s, const float* video, const float* proj, const float* query, const Lin& vq_fc, const float* vp, const float* qu,
                 float* out, int n, int T, int max_T, long total_rows, const int32_t* q_vid, const int32_t* vid_row_off, const int32_t* q_row_off,
                 void* frag, bool refresh);
