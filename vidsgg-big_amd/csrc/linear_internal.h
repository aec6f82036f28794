// Shared by linear.hip (exact f32 MFMA kernel) and linear_split.hip (f16x3 split-operand MFMA kernel).
#pragma once
#include "common.h"
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct LinearArgs {
    const float* A; int lda; const int32_t* a_idx; int nseg; int seg_len;
    const float* W; int ldw; const float* bias; const float* res; int ldr;
    float* C; int ldc; int M, N, K; int act; int tiles_m, tiles_n;
    const u32x4* Ws; int ldws;            // split kernel: W as [N][ldws] groups of 4 k: {4 x f16 head, 4 x f16 scaled residual}
    const float* zero_row;                // split kernel, gather mode: seg_len zeros standing for the row index -1
    const int* wbad; int wbad_id;         // split kernel: *wbad == wbad_id when the split W holds an out-of-range element
    // split kernel, depth-wise-separable mode: the A operand is dwconv(A) evaluated while the rows are staged -- a k=dw_ks
    // (3 or 7) depth-wise convolution along the rows of each sequence, zero padded, weights (K, dw_ks), bias (K).  Sequences are
    // dw_T rows each, or of the lengths given per row by dw_info (position, length).  dw_w == nullptr: off.
    const float* dw_w = nullptr; const float* dw_b = nullptr; int dw_ks = 0; int dw_T = 0; const int2* dw_info = nullptr;
    // split kernel, preceding-layer mode: A = relu(pre_x pre_w^T + pre_b) with pre_x (M, 8), pre_w (K, 8), pre_b (K), evaluated while
    // the rows are staged (fc_bbox2enti.0 -> .2, models/model_0v10.py:446: the 8 -> 512 layer's output never exists in HBM)
    const float* pre_x = nullptr; const float* pre_w = nullptr; const float* pre_b = nullptr;
    // split kernel: also / instead of C, the result in the split form "hr8" of gemm_ps.hip (pitch in BYTES; N % 8 == 0); an
    // element that form cannot hold sets *range_flag
    unsigned char* Cs = nullptr; long ldcs = 0; int* range_flag = nullptr;
    // split kernel, row-pointer mode: row m of A starts at address a_rowptr[m] (16-byte aligned) instead of A + m * lda -- the operand is
    // read in place from rows that live in separate allocations (the per-tracklet feature tensors of a batch, bigc_batch.feat_row_ptr)
    const uint64_t* a_rowptr = nullptr;
    // split kernel, split-K (set by the launcher for small grids inside a SplitKScope): workgroup (tile, blockIdx.y) sums the K slabs
    // [blockIdx.y * nkp, ...) and stores the raw partial tile to part[blockIdx.y] (M x N, dense); splitk_reduce_kernel adds the partials
    // in split order and applies bias / activation / residual
    int splits = 1; float* part = nullptr;
};

// Split-K is allowed for the launches of the calling thread while a scope is open (the TRAINING forward / backward: at the reference's
// batch of 8 videos a decoder GEMM is 24 - 48 tiles on 256 CUs and its time is the latency of 16 K slabs).  The eval forwards never open
// one: there a GEMM's result must not depend on how many rows share the launch (a video's bits do not depend on its company).
extern thread_local int bigc_splitk_allowed;
struct SplitKScope {
    int prev;
    SplitKScope() : prev(bigc_splitk_allowed) { bigc_splitk_allowed = 1; }
    ~SplitKScope() { bigc_splitk_allowed = prev; }
};

// Weights that stay constant between launches are split once and cached per (pointer, shape) for an "epoch": the model
// entry points open a StaticWeights scope with the epoch they drew at finalize time; outside such a scope (epoch 0, e.g.
// the raw bigc_linear_f32 entry point or the transposed weight copies of the backward pass) W is split on every launch.
extern "C" int bigc_get_accum_policy(void);    // linear.hip: accumulation policy of the model forwards' pre-split GEMMs (0 plain .. 4 blocked)
extern "C" int bigc_linear_get_fast(void);     // linear.hip: 1 = reduced-precision mode (one f16 MFMA per product)
// per-call override of the process-wide arithmetic mode (linear.hip): -1 = follow the process default
extern thread_local int bigc_mode_exact_override, bigc_mode_fast_override;
struct ModeScope {
    int pe, pf;
    ModeScope(int exact, int fast) : pe(bigc_mode_exact_override), pf(bigc_mode_fast_override) {
        if (exact >= 0) bigc_mode_exact_override = exact;
        if (fast >= 0) bigc_mode_fast_override = fast;
    }
    ~ModeScope() { bigc_mode_exact_override = pe; bigc_mode_fast_override = pf; }
};
extern thread_local unsigned long long bigc_static_weights_epoch;
unsigned long long bigc_next_weights_epoch();
struct StaticWeights {
    unsigned long long prev;
    explicit StaticWeights(unsigned long long e) : prev(bigc_static_weights_epoch) { bigc_static_weights_epoch = e; }
    ~StaticWeights() { bigc_static_weights_epoch = prev; }
};

// drops the cached split forms of the weight at `w` (both kernels' caches) and frees their device memory; called for every
// parameter of a model that is destroyed
void bigc_weight_cache_forget(const float* w);
void bigc_gemm_ps_cache_forget(const float* w);
// Batched weight preparation (linear_split.hip).  A training step changes every weight, so every GEMM of the step would launch
// its own split (and the backward a transpose as well): instead the entry points of the training step record the (weight, shape)
// pairs their GEMMs asked for in a per-model list and refresh all of them with a handful of launches at the next step's start.
struct SplitJob { const float* W; int N, K, ldw; int transposed; };
extern thread_local std::vector<SplitJob>* bigc_split_recorder;      // where stale lookups under a weights epoch are recorded
struct SplitRecorder {
    std::vector<SplitJob>* prev;
    explicit SplitRecorder(std::vector<SplitJob>* r) : prev(bigc_split_recorder) { bigc_split_recorder = r; }
    ~SplitRecorder() { bigc_split_recorder = prev; }
};
// refreshes the stale entries among `jobs` for the current weights epoch (no-op outside a StaticWeights scope)
int bigc_linear_prepare(hipStream_t s, const SplitJob* jobs, int n);
// -> *wT = f32 copy of W^T (K, round4(N)), valid for the current weights epoch, with its split form cached under the copy's own
// key: pass it to bigc_linear_f32 as an (N'=K, K'=round4(N), ldw=round4(N)) weight.  Outside an epoch scope it is rebuilt per call.
int bigc_weight_transposed(hipStream_t s, const float* W, int N, int K, int ldw, const float** wT);
// y = act(dwconv(x) W^T + bias) (+ res) in one pass over x (grounding: every depth-wise-separable convolution of the QANet
// encoders and of the heads, models/grd_model_v5.py:81-107): false when the current mode / shape needs the two-kernel form
// C (f32, may be null) and / or Cs (split form "hr8", pitch ldcs floats = 4 ldcs bytes) = act(A W^T + bias) + res on linear_split_kernel:
// the producer side of a GEMM -> GEMM chain whose second GEMM runs on gemm_ps.hip
int bigc_linear_hr8out(void* stream, const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldr,
                       float* C, int ldc, void* Cs, long ldcs, int M, int N, int K, int act, int* range_flag);
// the same with the rows of A given by address (row m at row_ptr[m], 16-byte aligned, K floats each): csrc/model.hip reads the visual features
// of a batch in place from the callers' per-tracklet tensors
int bigc_linear_hr8out_rows(void* stream, const uint64_t* row_ptr, const float* W, int ldw, const float* bias, void* Cs, long ldcs, int M, int N, int K,
                            int act, int* range_flag);
// f32 output, rows of A by address (training forward, in place); backward.hip: its weight gradient with the rows of X by address
int bigc_linear_rows_f32(void* stream, const uint64_t* row_ptr, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K, int act);
int bigc_linear_dw_rows(void* stream, const float* dY, int ldy, const uint64_t* x_rowptr, float* dW, int ldw, float* db, int M, int N, int K);
// the K = 8 GEMM (linear_k8_kernel) with its output in the split form
int bigc_linear_k8_hr8out(void* stream, const float* A, const float* W, const float* bias, void* Cs, long ldcs, int M, int N, int act, int* range_flag);
// y = act(relu(x8 W0^T + b0) W^T + bias): the K = 8 layer folded into the staging of the next GEMM; false -> two launches
bool bigc_linear_pre8_supported(int N, int K);
int bigc_linear_pre8(void* stream, const float* x8, const float* W0, const float* b0, const float* W, int ldw, const float* bias,
                     float* C, int ldc, int M, int N, int K, int act);
bool bigc_linear_dwsep_supported(int N, int K, int ks);
int bigc_linear_dwsep(void* stream, const float* X, int ldx, const float* dw_w, const float* dw_b, int ks, int T, const int2* row_info,
                      const float* W, int ldw, const float* bias, const float* res, int ldr, float* C, int ldc, int M, int N, int K, int act);
// launches the split-operand kernel (and its W pre-split pass); returns BIGC_OK or an error code
int bigc_linear_split_launch(hipStream_t s, LinearArgs& p, double flops, double bytes);
// (internal) epilogue mode of linear_split_kernel: C = (res > 0) ? A W^T : 0 -- the backward of a ReLU whose output is `res`, folded
// into the dX GEMM that produces its input gradient (train.hip); no residual is added in this mode
#define BIGC_ACT_RELU_MASK 3
// dX (M, N) = mask(dY (M, K) W^T) with W (N, K) and mask = relu_out > 0 (relu_out (M, N), pitch ldm); linear.hip
int bigc_linear_relu_masked(void* stream, const float* A, int lda, const float* W, int ldw, const float* relu_out, int ldm, float* C, int ldc,
                            int M, int N, int K);

// ---- gemm_ps.hip: GEMM on operands stored as split f16 pairs ("hr8": per 8 consecutive k, 16 bytes of heads then 16 bytes of
// scaled residuals; same footprint and row pitch as f32).  A is hr8 (optionally row-gathered) or f32 (split in the loop), W is
// an f32 weight whose hr8 form is cached per weights epoch; outputs: f32 C and / or hr8 Cs (for the next GEMM).
enum { PS_F32 = 0, PS_HR8 = 1 };
#define PS_ACT_EXP 3      /* scale exponent of activations in the hr8 format (gemm_ps.hip): stored value = x * 2^3 */
struct PsGemm {
    const void* A = nullptr; long lda = 0; int a_fmt = PS_F32;      // pitch in BYTES
    const int32_t* a_idx = nullptr; int nseg = 1;                    // hr8 only: source row per (output row, K segment), -1 = zeros
    const unsigned long long* a_rowptr = nullptr;                    // f32 only: address of row m (16-byte aligned, K floats) instead of A + m * lda
    // accumulation (gemm_ps.hip): 0 = one accumulator per element, tile size chosen for speed; 1 = the correction terms in a second accumulator;
    // 2 = blocked (partial sums of K = 128 folded into a running total: rounds like a well-blocked CPU GEMM).  1 and 2 run on 128-row tiles.
    int accum = 0;
    const float* W = nullptr; int ldw = 0;                           // (N, K) f32, pitch in floats
    const float* bias = nullptr; const float* res = nullptr; long ldr = 0;
    int res_period = 0;                                              // > 0: residual row of output row m = m % res_period (e.g. W pos_embedding per query)
    float* C = nullptr; long ldc = 0;                                // pitch in floats
    void* Cs = nullptr; long ldcs = 0;                               // pitch in BYTES
    int M = 0, N = 0, K = 0, act = 0;
    int* range_flag = nullptr;                                       // set to 1 when a split operand left the f16 range
    // LayerNorm epilogue (N = 512, hr8 operand): y = LN(act(A W^T + b) + res) * ln_g + ln_b; C = y (+ ln_add[m % ln_add_period]), Cs = split(y)
    const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 1e-5f; const float* ln_add = nullptr; int ln_add_period = 0;
};
bool bigc_gemm_ps_supported(const PsGemm& g);
int bigc_gemm_ps_launch(hipStream_t s, const PsGemm& g);
int bigc_split_rows_hr8(hipStream_t s, const float* X, long ldx, void* Y, long ldy_bytes, long rows, int K, int* range_flag);
