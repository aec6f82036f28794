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
#include "linear_internal.h"
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

// ------------------------------------------------------------------------------------------------------------------------------
// Ragged pack of a batch of videos (reference: the per-tracklet tensors of `TrajProposal`
// -- dataloaders/dataloader_vidvrd.py:66-77 moves them one by one -- are stacked inside forward, models/model_0v10.py:391-430).
// ONE launch copies every tracklet's rows from wherever its tensors live into the packed store of `bigc_batch`:
//   boxes (R, 4) f32;  and the feature row in any of three forms, each optional:
//     feats (R, width) f32                 -- training, exact mode
//     hr8   (R, dim_feat) split form       -- the visual (RoI) columns as the f16 head / residual pairs that csrc/gemm_ps.hip reads by
//                                             DMA (fc_feat2enti.0's operand): written here, there is no separate conversion pass
//     extra (R, width - dim_feat) f32      -- the remaining columns (I3D / classeme) that the time-mean reads
// HBM-bound: reads a row once, writes it once (4 bytes in, 4 bytes out per element in the eval form).
// One 256-thread workgroup per tracklet, wave w takes rows w, w + 4, ...; a row's loads are issued before its stores.
// VEC: every source pointer and pitch is 16-byte aligned (the normal case: torch allocations, widths that are multiples of 4);
// otherwise the element-wise variant runs.
// ------------------------------------------------------------------------------------------------------------------------------
namespace {

typedef _Float16 pk_h2 __attribute__((ext_vector_type(2)));
typedef float pk_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void pk_split2(float x0, float x1, unsigned& h, unsigned& r) {
    const pk_f2 v = {x0, x1};
    const pk_h2 hh = __builtin_convertvector(v, pk_h2);
    const pk_f2 d = v - __builtin_convertvector(hh, pk_f2);
    const pk_h2 rr = __builtin_convertvector(d, pk_h2);
    h = *(const unsigned*)&hh;
    r = *(const unsigned*)&rr;
}

struct PackArgs {
    const uint64_t* feat_src; const int32_t* feat_ld; const uint64_t* box_src; const int32_t* trk_row_off;
    int width, dim_feat;
    float* feats; long ld_feats; unsigned char* hr8; long ld_hr8; float* extra; long ld_extra; float* boxes; int* flag;
    float act_scale;
};

template <bool VEC>
__global__ __launch_bounds__(256) void pack_tracklets_kernel(PackArgs a) {
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = a.trk_row_off[t], n = a.trk_row_off[t + 1] - r0;
    const float* src = a.feat_src ? (const float*)a.feat_src[t] : nullptr;          // (boxes only: no feature table)
    const long lds = a.feat_ld ? (long)a.feat_ld[t] : (long)a.width;
    if (a.boxes && a.box_src) {
        const float* bs = (const float*)a.box_src[t];
        for (int i = threadIdx.x; i < 4 * n; i += 256) a.boxes[(size_t)r0 * 4 + i] = bs[i];
    }
    if (!src) return;
    const int K8 = a.hr8 ? a.dim_feat / 8 : 0;
    const int xw = a.width - a.dim_feat;
    bool oob = false;
    for (int i = wave; i < n; i += 4) {
        const float* s = src + (size_t)i * lds;
        const size_t r = (size_t)r0 + i;
        if (VEC) {
            if (a.hr8) {
                uint4* yp = (uint4*)(a.hr8 + r * a.ld_hr8);
                float* fo = a.feats ? a.feats + r * a.ld_feats : nullptr;
                for (int g0 = 0; g0 < K8; g0 += 256) {                     // four groups of 8 per lane in flight
                    f32x4 u[4], v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int g = g0 + 64 * q + lane;
                        if (g < K8) { u[q] = *(const f32x4*)(s + 8 * g); v[q] = *(const f32x4*)(s + 8 * g + 4); }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int g = g0 + 64 * q + lane;
                        if (g >= K8) continue;
                        if (fo) { *(f32x4*)(fo + 8 * g) = u[q]; *(f32x4*)(fo + 8 * g + 4) = v[q]; }
                        const f32x4 x = u[q] * a.act_scale, y = v[q] * a.act_scale;
                        uint4 h, rr;
                        pk_split2(x[0], x[1], h.x, rr.x); pk_split2(x[2], x[3], h.y, rr.y); pk_split2(y[0], y[1], h.z, rr.z); pk_split2(y[2], y[3], h.w, rr.w);
#pragma unroll
                        for (int c = 0; c < 4; ++c) oob = oob || !(fabsf(x[c]) < 32768.0f) || !(fabsf(y[c]) < 32768.0f);
                        yp[2 * g] = h;
                        yp[2 * g + 1] = rr;
                    }
                }
            } else if (a.feats) {
                float* fo = a.feats + r * a.ld_feats;
                for (int c = 4 * lane; c < a.dim_feat; c += 256) *(f32x4*)(fo + c) = *(const f32x4*)(s + c);
            }
            if (xw > 0 && (a.extra || a.feats)) {
                float* xo = a.extra ? a.extra + r * a.ld_extra : nullptr;
                float* fo = a.feats ? a.feats + r * a.ld_feats + a.dim_feat : nullptr;
                for (int c0 = 0; c0 < xw; c0 += 1024) {
                    f32x4 u[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int c = c0 + 256 * q + 4 * lane; if (c < xw) u[q] = *(const f32x4*)(s + a.dim_feat + c); }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = c0 + 256 * q + 4 * lane;
                        if (c >= xw) continue;
                        if (xo) *(f32x4*)(xo + c) = u[q];
                        if (fo) *(f32x4*)(fo + c) = u[q];
                    }
                }
            }
        } else {
            for (int c = lane; c < a.width; c += 64) {
                const float x = s[c];
                if (a.feats) a.feats[r * a.ld_feats + c] = x;
                if (c >= a.dim_feat && a.extra) a.extra[r * a.ld_extra + (c - a.dim_feat)] = x;
            }
            if (a.hr8) {
                for (int g = lane; g < K8; g += 64) {
                    float x[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) { x[c] = s[8 * g + c] * a.act_scale; oob = oob || !(fabsf(x[c]) < 32768.0f); }
                    uint4 h, rr;
                    pk_split2(x[0], x[1], h.x, rr.x); pk_split2(x[2], x[3], h.y, rr.y); pk_split2(x[4], x[5], h.z, rr.z); pk_split2(x[6], x[7], h.w, rr.w);
                    uint4* yp = (uint4*)(a.hr8 + r * a.ld_hr8);
                    yp[2 * g] = h;
                    yp[2 * g + 1] = rr;
                }
            }
        }
    }
    if (oob && a.flag) atomicOr(a.flag, 1);
}

// per-video metadata: cat_ids (N) i64, scores (N) f32, traj_durations (N, 2) i64 of every video -> trk_cat i32, trk_score f32, trk_dura i32
__global__ __launch_bounds__(256) void pack_video_meta_kernel(const uint64_t* __restrict__ cat_src, const uint64_t* __restrict__ score_src,
                                                              const uint64_t* __restrict__ dura_src, const int32_t* __restrict__ vid_trk_off,
                                                              int32_t* __restrict__ trk_cat, float* __restrict__ trk_score, int32_t* __restrict__ trk_dura) {
    const int v = blockIdx.x, e0 = vid_trk_off[v], n = vid_trk_off[v + 1] - e0;
    const int64_t* cs = (const int64_t*)cat_src[v];
    const float* ss = (const float*)score_src[v];
    const int64_t* ds = (const int64_t*)dura_src[v];
    for (int i = threadIdx.x; i < n; i += 256) {
        trk_cat[e0 + i] = (int32_t)cs[i];
        trk_score[e0 + i] = ss[i];
        trk_dura[2 * (e0 + i)] = (int32_t)ds[2 * i];
        trk_dura[2 * (e0 + i) + 1] = (int32_t)ds[2 * i + 1];
    }
}

}  // namespace

extern "C" int bigc_pack_tracklets(void* stream, const uint64_t* feat_src, const int32_t* feat_ld, const uint64_t* box_src,
                                   const int32_t* trk_row_off, int n_trk, long n_rows, int width, int dim_feat, int vectorized, float* feats, long ld_feats,
                                   void* feats_hr8, long ld_hr8_bytes, float* extra, long ld_extra, float* boxes, int32_t* range_flag) {
    BIGC_CHECK_ARG(trk_row_off && (feat_src || box_src), "pack_tracklets: null pointer");
    BIGC_CHECK_ARG(width > 0 && dim_feat >= 0 && dim_feat <= width, "pack_tracklets: dim_feat=%d outside [0, width=%d]", dim_feat, width);
    BIGC_CHECK_ARG(!feats || ld_feats >= width, "pack_tracklets: ld_feats < width");
    BIGC_CHECK_ARG(!feats_hr8 || (dim_feat % 8 == 0 && ld_hr8_bytes >= 4L * dim_feat && ld_hr8_bytes % 16 == 0 && range_flag),
                   "pack_tracklets: the split form needs dim_feat %% 8 == 0, a 16-byte pitch and a range flag");
    BIGC_CHECK_ARG(!extra || ld_extra >= width - dim_feat, "pack_tracklets: ld_extra < width - dim_feat");
    BIGC_CHECK_ARG(!(boxes && !box_src), "pack_tracklets: boxes without box_src");
    if (vectorized)
        BIGC_CHECK_ARG(width % 4 == 0 && dim_feat % 4 == 0 && (!feats || (ld_feats % 4 == 0 && (uintptr_t)feats % 16 == 0)) &&
                       (!extra || (ld_extra % 4 == 0 && (uintptr_t)extra % 16 == 0)) && (!feats_hr8 || (uintptr_t)feats_hr8 % 16 == 0),
                       "pack_tracklets: vectorized copy needs widths / pitches that are multiples of 4 floats and 16-byte aligned outputs");
    if (n_trk <= 0 || n_rows <= 0) return BIGC_OK;
    PackArgs a{feat_src, feat_ld, box_src, trk_row_off, width, dim_feat, feats, ld_feats, (unsigned char*)feats_hr8, ld_hr8_bytes, extra, ld_extra, boxes,
               range_flag, (float)(1 << PS_ACT_EXP)};
    const double per_row = (feat_src ? 4.0 * width * (1 + (feats ? 1 : 0)) + (feats_hr8 ? 4.0 * dim_feat : 0) + (extra ? 4.0 * (width - dim_feat) : 0) : 0) + (boxes ? 32.0 : 0);
    ProfScope ps(stream, FAM_PACK, 0.0, per_row * (double)n_rows);
    if (vectorized) hipLaunchKernelGGL(pack_tracklets_kernel<true>, dim3(n_trk), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(pack_tracklets_kernel<false>, dim3(n_trk), dim3(256), 0, (hipStream_t)stream, a);
    BIGC_LAUNCH_CHECK("pack_tracklets");
    return BIGC_OK;
}

extern "C" int bigc_pack_video_meta(void* stream, const uint64_t* cat_src, const uint64_t* score_src, const uint64_t* dura_src,
                                    const int32_t* vid_trk_off, int n_vid, int32_t* trk_cat, float* trk_score, int32_t* trk_dura) {
    BIGC_CHECK_ARG(cat_src && score_src && dura_src && vid_trk_off && trk_cat && trk_score && trk_dura, "pack_video_meta: null pointer");
    if (n_vid <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_PACK, 0.0, 0.0);
    hipLaunchKernelGGL(pack_video_meta_kernel, dim3(n_vid), dim3(256), 0, (hipStream_t)stream, cat_src, score_src, dura_src, vid_trk_off, trk_cat,
                       trk_score, trk_dura);
    BIGC_LAUNCH_CHECK("pack_video_meta");
    return BIGC_OK;
}

// ---------------------------------------------------------------------------------------------------
// Host -> device copy done by a kernel: `src` is page-locked host memory mapped into the device's address space (what
// torch's pin_memory() / hipHostMalloc return), read by the GPU over the host link: 300 KB in 10 us, ordered on the compute stream like
// every other launch, no copy queue or copy engine involved (bigc_hip/_lib.py: upload).
namespace {
__global__ __launch_bounds__(256) void copy_mapped_kernel(unsigned char* __restrict__ dst, const unsigned char* __restrict__ src, long n16, long nbytes) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) ((u32x4*)dst)[i] = __builtin_nontemporal_load((const u32x4*)src + i);
    for (long i = n16 * 16 + (long)blockIdx.x * 256 + threadIdx.x; i < nbytes; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" int bigc_copy_mapped(void* stream, void* dst, const void* src_mapped, long nbytes) {
    BIGC_CHECK_ARG(dst && src_mapped, "copy_mapped: null pointer");
    if (nbytes <= 0) return BIGC_OK;
    const bool al = ((((uintptr_t)dst) | ((uintptr_t)src_mapped)) & 15) == 0;
    const long n16 = al ? nbytes / 16 : 0;
    const long units = al ? n16 + 15 : nbytes;
    const int grid = (int)((units + 255) / 256 < 2048 ? (units + 255) / 256 : 2048);
    ProfScope ps(stream, FAM_PACK, 0.0, 2.0 * (double)nbytes);
    hipLaunchKernelGGL(copy_mapped_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst, (const unsigned char*)src_mapped, n16, nbytes);
    BIGC_LAUNCH_CHECK("copy_mapped");
    return BIGC_OK;
}

// A progress mark the host can read without asking the runtime: *mapped_word = value once everything queued on `stream` before this
// call has run (mapped_word: page-locked host memory mapped into the device's address space): the host polls a word instead of
// creating, recording and waiting on an event per step.
// or_flag (device, may be null): when *or_flag != 0, mapped_word[1] = value too -- the error flag of a step travels with its mark.
namespace {
__global__ void stamp_kernel(volatile unsigned long long* mapped_word, unsigned long long value, const int* __restrict__ or_flag) {
    if (or_flag && *or_flag) mapped_word[1] = value;
    __threadfence_system();
    mapped_word[0] = value;
    __threadfence_system();
}
}  // namespace

extern "C" int bigc_stamp(void* stream, void* mapped_word, unsigned long long value, const int* or_flag) {
    BIGC_CHECK_ARG(mapped_word && (((uintptr_t)mapped_word) & 7) == 0, "stamp: null or misaligned pointer");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (volatile unsigned long long*)mapped_word, value, or_flag);
    BIGC_LAUNCH_CHECK("stamp");
    return BIGC_OK;
}
