// Ragged tracklet front-end helpers (all HBM/latency-bound index and reduction work):
//   bbox_feat        per-frame box features with in-tracklet forward differences
//   conv_index       row-gather table of the k3/s2 temporal conv over the VIRTUALLY stretched sequence
//   adaptive_maxpool adaptive_max_pool1d(T_o -> P), channel-major
//   stretched_mean   mean over the stretched frames = repeat-count weighted mean of the source rows
//
// The reference materialises every tracklet stretched to the video's longest one
// (``stack_with_repeat_2d`` models/model_0v10.py:18-46: an (N,Tmax,D) copy); here the stretch is only
// an index map over the packed (sum L_i)-row store: with q,r = divmod(Tmax, L) the stretched frame f
// reads source frame f/(q+1) when f < r(q+1) and r + (f - r(q+1))/q otherwise.
#include "common.h"
#include "prof.h"

namespace {

__device__ __forceinline__ int stretch_src(int f, int q, int r) {
    const int cut = r * (q + 1);
    return f < cut ? f / (q + 1) : r + (f - cut) / q;
}

// reference: models/model_0v10.py:401-420
__global__ __launch_bounds__(256) void bbox_feat_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ row_off,
                                                        const int32_t* __restrict__ trk_vid, const float* __restrict__ vid_wh,
                                                        float* __restrict__ out) {
    const int t = blockIdx.x;
    const int r0 = row_off[t], L = row_off[t + 1] - r0;
    const int v = trk_vid[t];
    const float w = vid_wh[2 * v], h = vid_wh[2 * v + 1];
    for (int f = threadIdx.x; f < L; f += 256) {
        const f32x4 b = ((const f32x4*)boxes)[r0 + f];
        const float x1 = b[0] / w, y1 = b[1] / h, x2 = b[2] / w, y2 = b[3] / h;
        const float cx = (x2 + x1) / 2, cy = (y2 + y1) / 2, bw = x2 - x1, bh = y2 - y1;
        float dcx = 0.f, dcy = 0.f, dw = 0.f, dh = 0.f;
        if (f + 1 < L) {
            const f32x4 n = ((const f32x4*)boxes)[r0 + f + 1];
            const float nx1 = n[0] / w, ny1 = n[1] / h, nx2 = n[2] / w, ny2 = n[3] / h;
            dcx = (nx2 + nx1) / 2 - cx;
            dcy = (ny2 + ny1) / 2 - cy;
            dw = (nx2 - nx1) - bw;
            dh = (ny2 - ny1) - bh;
        }
        f32x4* o = (f32x4*)(out + (size_t)(r0 + f) * 8);
        o[0] = f32x4{cx, dcx, cy, dcy};
        o[1] = f32x4{bw, dw, bh, dh};
    }
}

// reference: stretch models/model_0v10.py:18-46, conv geometry :302 (k=3, padding=1, stride=2)
__global__ __launch_bounds__(256) void conv_index_kernel(const int32_t* __restrict__ row_off, const int32_t* __restrict__ conv_off,
                                                         const int32_t* __restrict__ trk_vid, const int32_t* __restrict__ vid_tmax,
                                                         int32_t* __restrict__ idx) {
    const int t = blockIdx.x;
    const int r0 = row_off[t], L = row_off[t + 1] - r0;
    const int c0 = conv_off[t], To = conv_off[t + 1] - c0;
    const int M = vid_tmax[trk_vid[t]];
    const int q = M / L, r = M - q * L;
    for (int e = threadIdx.x; e < To * 3; e += 256) {
        const int j = e / 3, c = e - 3 * j;
        const int f = 2 * j - 1 + c;
        idx[(size_t)c0 * 3 + e] = (f < 0 || f >= M) ? -1 : r0 + stretch_src(f, q, r);
    }
}

// reference: F.adaptive_max_pool1d + channel-major reshape, models/model_0v10.py:456-457
__global__ __launch_bounds__(256) void adaptive_maxpool_kernel(const float* __restrict__ conv, const int32_t* __restrict__ conv_off,
                                                               float* __restrict__ pooled, int d, int P) {
    const int t = blockIdx.x;
    const int c0 = conv_off[t], To = conv_off[t + 1] - c0;
    if ((d & 3) == 0) {
        // one (dwordx4 column group, window) item per thread -- all 256 threads busy at d = 512, P = 4 -- and the rows of a
        // window read four at a time (independent loads); the windows of adaptive pooling overlap by at most one row
        const int ng = d >> 2;
        for (int it = threadIdx.x; it < ng * P; it += 256) {
            const int p = it / ng, c = (it - p * ng) * 4;
            const int s = (p * To) / P, e = ((p + 1) * To + P - 1) / P;
            const float* src = conv + (size_t)c0 * d + c;
            // eight rows in flight per round, the last one repeated past the window's end (a repeated row changes no maximum): a window
            // of the usual eight rows is ONE round of loads instead of three dependent ones (1 + 4 + 3 x 1)
            f32x4 m;
            for (int j = s; j < e; j += 8) {
                f32x4 x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = *(const f32x4*)(src + (size_t)min(j + u, e - 1) * d);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = fmaxf(fmaxf(fmaxf(x[0][k], x[1][k]), fmaxf(x[2][k], x[3][k])), fmaxf(fmaxf(x[4][k], x[5][k]), fmaxf(x[6][k], x[7][k])));
                    m[k] = j == s ? v : fmaxf(m[k], v);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) pooled[(size_t)t * d * P + (size_t)(c + k) * P + p] = m[k];
        }
        return;
    }
    for (int c = threadIdx.x; c < d; c += 256) {
        for (int p = 0; p < P; ++p) {
            const int s = (p * To) / P, e = ((p + 1) * To + P - 1) / P;
            float m = conv[(size_t)(c0 + s) * d + c];
            for (int j = s + 1; j < e; ++j) m = fmaxf(m, conv[(size_t)(c0 + j) * d + c]);
            pooled[(size_t)t * d * P + (size_t)c * P + p] = m;
        }
    }
}

// reference: traj_i3d.mean(dim=1) over the stretched tensor, models/model_0v10.py:470 (model_0v7.py:473)
__global__ __launch_bounds__(256) void stretched_mean_kernel(const float* __restrict__ feats, int ldf, int col0, int w,
                                                             const int32_t* __restrict__ row_off, const int32_t* __restrict__ trk_vid,
                                                             const int32_t* __restrict__ vid_tmax, float* __restrict__ out) {
    const int t = blockIdx.x;
    const int r0 = row_off[t], L = row_off[t + 1] - r0;
    const int M = vid_tmax[trk_vid[t]];
    const int q = M / L, r = M - q * L;
    const float inv = 1.0f / (float)M;
    if ((w & 3) == 0 && (ldf & 3) == 0 && (col0 & 3) == 0) {
        // one dwordx4 column group per thread, up to eight rows in flight (HBM-bound: the tracklet's rows are read once)
        for (int c = threadIdx.x * 4; c < w; c += 1024) {
            const float* p = feats + (size_t)r0 * ldf + col0 + c;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            int i = 0;
            for (; i + 8 <= L; i += 8) {            // eight rows in flight, summed in row order
                f32x4 x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = *(const f32x4*)(p + (size_t)(i + u) * ldf);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float wt = (i + u < r) ? (float)(q + 1) : (float)q;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] += wt * x[u][k];
                }
            }
            for (; i + 4 <= L; i += 4) {
                f32x4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = *(const f32x4*)(p + (size_t)(i + u) * ldf);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float wt = (i + u < r) ? (float)(q + 1) : (float)q;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] += wt * x[u][k];
                }
            }
            for (; i < L; ++i) {
                const f32x4 x = *(const f32x4*)(p + (size_t)i * ldf);
                const float wt = (i < r) ? (float)(q + 1) : (float)q;
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += wt * x[k];
            }
            *(f32x4*)(out + (size_t)t * w + c) = f32x4{s[0] * inv, s[1] * inv, s[2] * inv, s[3] * inv};
        }
        return;
    }
    for (int c = threadIdx.x; c < w; c += 256) {
        const float* p = feats + (size_t)r0 * ldf + col0 + c;
        float s = 0.f;
        for (int i = 0; i < L; ++i) {
            const float x = p[(size_t)i * ldf];
            s += (i < r ? (float)(q + 1) : (float)q) * x;
        }
        out[(size_t)t * w + c] = s * inv;
    }
}

// the same over rows given by address (row r of the batch at row_ptr[r], 16-byte aligned): the tracklets' rows are read in place from the
// callers' tensors (bigc_batch.feat_row_ptr).  The row addresses are workgroup-uniform (scalar loads).
__global__ __launch_bounds__(256) void stretched_mean_rows_kernel(const uint64_t* __restrict__ row_ptr, int col0, int w,
                                                                  const int32_t* __restrict__ row_off, const int32_t* __restrict__ trk_vid,
                                                                  const int32_t* __restrict__ vid_tmax, float* __restrict__ out) {
    const int t = blockIdx.x;
    const int r0 = row_off[t], L = row_off[t + 1] - r0;
    const int M = vid_tmax[trk_vid[t]];
    const int q = M / L, r = M - q * L;
    const float inv = 1.0f / (float)M;
    for (int c = threadIdx.x * 4; c < w; c += 1024) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int i = 0;
        for (; i + 8 <= L; i += 8) {            // eight rows in flight, summed in row order (the order of stretched_mean_kernel)
            f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *(const __attribute__((address_space(1))) f32x4*)((const __attribute__((address_space(1))) float*)row_ptr[r0 + i + u] + col0 + c);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float wt = (i + u < r) ? (float)(q + 1) : (float)q;
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += wt * x[u][k];
            }
        }
        for (; i + 4 <= L; i += 4) {
            f32x4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = *(const __attribute__((address_space(1))) f32x4*)((const __attribute__((address_space(1))) float*)row_ptr[r0 + i + u] + col0 + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float wt = (i + u < r) ? (float)(q + 1) : (float)q;
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += wt * x[u][k];
            }
        }
        for (; i < L; ++i) {
            const f32x4 x = *(const __attribute__((address_space(1))) f32x4*)((const __attribute__((address_space(1))) float*)row_ptr[r0 + i] + col0 + c);
            const float wt = (i < r) ? (float)(q + 1) : (float)q;
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += wt * x[k];
        }
        *(f32x4*)(out + (size_t)t * w + c) = f32x4{s[0] * inv, s[1] * inv, s[2] * inv, s[3] * inv};
    }
}

// row r0 + i of tracklet t starts at feat_src[t] + i * pitch: the per-row address table the in-place readers take
__global__ __launch_bounds__(256) void row_pointers_kernel(const uint64_t* __restrict__ feat_src, const int32_t* __restrict__ feat_ld, int width,
                                                           const int32_t* __restrict__ row_off, uint64_t* __restrict__ row_ptr) {
    const int t = blockIdx.x;
    const int r0 = row_off[t], L = row_off[t + 1] - r0;
    const uint64_t base = feat_src[t], pitch = 4ull * (uint64_t)(feat_ld ? feat_ld[t] : width);
    for (int i = threadIdx.x; i < L; i += 256) row_ptr[r0 + i] = base + (uint64_t)i * pitch;
}

// (n_out, n_in, taps) torch Conv1d weight -> (n_out, taps*n_in) tap-major rows for the implicit GEMM
__global__ void repack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ out, long total, int n_in, int taps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long per_out = (long)n_in * taps;
    const long o = i / per_out;
    const int rem = (int)(i - o * per_out);
    const int k = rem / n_in, c = rem - k * n_in;
    out[i] = w[o * per_out + (long)c * taps + k];
}

}  // namespace

int bigc_repack_conv_weight(hipStream_t s, const float* w, float* out, int n_out, int n_in, int taps) {
    const long total = (long)n_out * n_in * taps;
    hipLaunchKernelGGL(repack_conv_weight_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, w, out, total, n_in, taps);
    BIGC_LAUNCH_CHECK("repack_conv_weight");
    return BIGC_OK;
}

extern "C" int bigc_bbox_feat_f32(void* stream, const float* boxes, const int32_t* trk_row_off, const int32_t* trk_vid,
                                  const float* vid_wh, float* out, int n_trk) {
    BIGC_CHECK_ARG(boxes && trk_row_off && trk_vid && vid_wh && out, "bbox_feat: null pointer");
    if (n_trk <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_FRONTEND, 0.0, 0.0);
    hipLaunchKernelGGL(bbox_feat_kernel, dim3(n_trk), dim3(256), 0, (hipStream_t)stream, boxes, trk_row_off, trk_vid, vid_wh, out);
    BIGC_LAUNCH_CHECK("bbox_feat");
    return BIGC_OK;
}

extern "C" int bigc_conv_index(void* stream, const int32_t* trk_row_off, const int32_t* trk_conv_off, const int32_t* trk_vid,
                               const int32_t* vid_tmax, int32_t* conv_idx, int n_trk) {
    BIGC_CHECK_ARG(trk_row_off && trk_conv_off && trk_vid && vid_tmax && conv_idx, "conv_index: null pointer");
    if (n_trk <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_FRONTEND, 0.0, 0.0);
    hipLaunchKernelGGL(conv_index_kernel, dim3(n_trk), dim3(256), 0, (hipStream_t)stream, trk_row_off, trk_conv_off, trk_vid, vid_tmax, conv_idx);
    BIGC_LAUNCH_CHECK("conv_index");
    return BIGC_OK;
}

extern "C" int bigc_adaptive_maxpool(void* stream, const float* conv, const int32_t* trk_conv_off, float* pooled,
                                     int n_trk, int d, int P) {
    BIGC_CHECK_ARG(conv && trk_conv_off && pooled && d > 0 && P > 0, "adaptive_maxpool: bad argument");
    if (n_trk <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_FRONTEND, 0.0, 0.0);
    hipLaunchKernelGGL(adaptive_maxpool_kernel, dim3(n_trk), dim3(256), 0, (hipStream_t)stream, conv, trk_conv_off, pooled, d, P);
    BIGC_LAUNCH_CHECK("adaptive_maxpool");
    return BIGC_OK;
}

extern "C" int bigc_stretched_mean(void* stream, const float* feats, int ldf, int col0, int w, const int32_t* trk_row_off,
                                   const int32_t* trk_vid, const int32_t* vid_tmax, float* out, int n_trk) {
    BIGC_CHECK_ARG(feats && trk_row_off && trk_vid && vid_tmax && out && w > 0, "stretched_mean: bad argument");
    if (n_trk <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_FRONTEND, 0.0, 0.0);
    hipLaunchKernelGGL(stretched_mean_kernel, dim3(n_trk), dim3(256), 0, (hipStream_t)stream, feats, ldf, col0, w, trk_row_off, trk_vid, vid_tmax, out);
    BIGC_LAUNCH_CHECK("stretched_mean");
    return BIGC_OK;
}

extern "C" int bigc_stretched_mean_rows(void* stream, const uint64_t* row_ptr, int col0, int w, const int32_t* trk_row_off, const int32_t* trk_vid,
                                        const int32_t* vid_tmax, float* out, int n_trk) {
    BIGC_CHECK_ARG(row_ptr && trk_row_off && trk_vid && vid_tmax && out && w > 0 && w % 4 == 0 && col0 % 4 == 0, "stretched_mean_rows: bad argument");
    if (n_trk <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_FRONTEND, 0.0, 0.0);
    hipLaunchKernelGGL(stretched_mean_rows_kernel, dim3(n_trk), dim3(256), 0, (hipStream_t)stream, row_ptr, col0, w, trk_row_off, trk_vid, vid_tmax, out);
    BIGC_LAUNCH_CHECK("stretched_mean_rows");
    return BIGC_OK;
}

extern "C" int bigc_row_pointers(void* stream, const uint64_t* feat_src, const int32_t* feat_ld, int width, const int32_t* trk_row_off, int n_trk,
                                 uint64_t* row_ptr) {
    BIGC_CHECK_ARG(feat_src && trk_row_off && row_ptr && width > 0, "row_pointers: bad argument");
    if (n_trk <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_PACK, 0.0, 0.0);
    hipLaunchKernelGGL(row_pointers_kernel, dim3(n_trk), dim3(256), 0, (hipStream_t)stream, feat_src, feat_ld, width, trk_row_off, row_ptr);
    BIGC_LAUNCH_CHECK("row_pointers");
    return BIGC_OK;
}
