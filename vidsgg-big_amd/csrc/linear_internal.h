// Shared by linear.hip (exact f32 MFMA kernel) and linear_split.hip (f16x3 split-operand MFMA kernel).
#pragma once
#include "common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct LinearArgs {
    const float* A; int lda; const int32_t* a_idx; int nseg; int seg_len;
    const float* W; int ldw; const float* bias; const float* res; int ldr;
    float* C; int ldc; int M, N, K; int act; int tiles_m, tiles_n;
    const u32x4* Ws; int ldws;            // split kernel: W as [N][ldws] groups of 4 k: {4 x f16 head, 4 x f16 scaled residual}
    const float* zero_row;                // split kernel, gather mode: seg_len zeros standing for the row index -1
    const int* wbad; int wbad_id;         // split kernel: *wbad == wbad_id when the split W holds an out-of-range element
};

// Weights that stay constant between launches are split once and cached per (pointer, shape) for an "epoch": the model
// entry points open a StaticWeights scope with the epoch they drew at finalize time; outside such a scope (epoch 0, e.g.
// the raw bigc_linear_f32 entry point or the transposed weight copies of the backward pass) W is split on every launch.
extern thread_local unsigned long long bigc_static_weights_epoch;
unsigned long long bigc_next_weights_epoch();
struct StaticWeights {
    unsigned long long prev;
    explicit StaticWeights(unsigned long long e) : prev(bigc_static_weights_epoch) { bigc_static_weights_epoch = e; }
    ~StaticWeights() { bigc_static_weights_epoch = prev; }
};

// launches the split-operand kernel (and its W pre-split pass); returns BIGC_OK or an error code
int bigc_linear_split_launch(hipStream_t s, LinearArgs& p, double flops, double bytes);
