// Optional in-library kernel timing with HIP events on the caller's stream (used by bench.py for the
// roofline figure: algorithmic flops/bytes per launch divided by the measured launch duration).
#pragma once
#include <hip/hip_runtime.h>

enum BigcFamily {
    FAM_LINEAR = 0, FAM_MHA, FAM_ROLE_ATT, FAM_ROLE_VAL, FAM_LAYERNORM, FAM_ADD_ROWS, FAM_FRONTEND,
    FAM_HEAD_GATHER, FAM_TRIPLETS, FAM_GROUNDING, FAM_LINEAR_DW, FAM_BACKWARD, FAM_ATTN_BWD, FAM_LOSS, FAM_LINEAR_AUX, FAM_PACK, FAM_COUNT
};

// Algorithmic flops / bytes for the NEXT scope this thread opens with 0 / 0 (launchers that only see device-side offset tables -- attention over
// ragged sequences -- cannot compute them; their callers, who know the row counts, leave them here)
void bigc_prof_hint(double flops, double bytes);

struct ProfScope {
    hipStream_t s; int slot;
    ProfScope(void* stream, int family, double flops, double bytes);
    ~ProfScope();
};
