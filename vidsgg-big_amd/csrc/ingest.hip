// Feature-row assembly of the tracking-result ingest (reference:
// dataloaders/dataloader_vidvrd.py:585-700 with utils/utils_func.py:273-317): one output row per frame of every kept
// tracklet, built from the raw per-box rows that were uploaded once:
//   copy of a raw row, a zero row (box without detection, :510-513), or -- inside a gap of the tracklet -- the row
//   numpy.linspace(left, right, num, axis=0)[i] evaluated in float64 exactly as numpy does it (step = delta / div;
//   y = i * step, or (i / div) * delta when ANY step of the vector is zero; y += left) and rounded to float32
//   (torch.FloatTensor of the float64 array), so the result is bit-identical to the reference's host code.
// The vector is the concatenation [RoI feature | I3D feature] (:654), which matters for the any-step-zero rule.
// HBM-bound: reads <= 2 source rows, writes 1 row; one 256-thread workgroup per output row.
#include "common.h"
#include "prof.h"

namespace {

__global__ __launch_bounds__(256) void assemble_rows_kernel(const float* __restrict__ raw, int ld1, int col1, int w1,
                                                            const float* __restrict__ raw2, int ld2, int col2, int w2,
                                                            const int32_t* __restrict__ src_a, const int32_t* __restrict__ src_b,
                                                            const int32_t* __restrict__ step_i, const int32_t* __restrict__ step_div,
                                                            float* __restrict__ out, int ld_out) {
    const int r = blockIdx.x;
    const int a = src_a[r], b = src_b[r];
    float* o = out + (size_t)r * ld_out;
    const int w = w1 + w2;
    auto fetch = [&](int row, int c) -> float {
        if (row < 0) return 0.f;
        return c < w1 ? raw[(size_t)row * ld1 + col1 + c] : raw2[(size_t)row * ld2 + col2 + (c - w1)];
    };
    if (b == -2) {                                            // present frame: copy (or zeros)
        for (int c = threadIdx.x; c < w; c += 256) o[c] = fetch(a, c);
        return;
    }
    const double div = (double)step_div[r], i = (double)step_i[r];
    int zero = 0;
    for (int c = threadIdx.x; c < w; c += 256) {
        const double delta = (double)fetch(b, c) - (double)fetch(a, c);
        zero |= (delta / div == 0.0) ? 1 : 0;
    }
    const bool any_zero = __syncthreads_or(zero) != 0;
    const double frac = i / div;
    for (int c = threadIdx.x; c < w; c += 256) {
        const double left = (double)fetch(a, c), delta = (double)fetch(b, c) - left;
        const double y = any_zero ? frac * delta : i * (delta / div);
        o[c] = (float)(y + left);
    }
}

}  // namespace

extern "C" int bigc_assemble_rows_f32(void* stream, const float* raw, int ld_raw, int col0, int width, const float* raw2, int ld_raw2,
                                      int col02, int width2, const int32_t* src_a, const int32_t* src_b, const int32_t* step_i,
                                      const int32_t* step_div, float* out, int ld_out, int n_rows) {
    BIGC_CHECK_ARG(raw && src_a && src_b && step_i && step_div && out, "assemble_rows: null pointer");
    BIGC_CHECK_ARG(width > 0 && width2 >= 0 && (width2 == 0 || raw2) && ld_out >= width + width2, "assemble_rows: bad widths");
    if (n_rows <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_FRONTEND, 0.0, 12.0 * n_rows * (double)(width + width2));
    hipLaunchKernelGGL(assemble_rows_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, raw, ld_raw, col0, width, raw2, ld_raw2, col02,
                       width2, src_a, src_b, step_i, step_div, out, ld_out);
    BIGC_LAUNCH_CHECK("assemble_rows");
    return BIGC_OK;
}
