// Row-wise HBM-bound kernels: LayerNorm and the broadcast row add (query + pos_embedding).
// One wave per row, 16-byte loads, statistics by in-register two-pass + wave reduction.
// Roofline: HBM (read x once, write y once): 8*d bytes per row.
#include "common.h"
#include "prof.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// torch.nn.LayerNorm: biased variance, eps inside the sqrt (reference: models/model_0v10.py:81-82,171-173)
// Optional fused broadcast add (the decoder's `query + pos_embedding`, models/model_0v10.py:181,189): with `add`,
// y = LN(x) + add[row % period] and, when y_plain is given, y_plain = LN(x) as well.
template <bool ADD>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                        float* __restrict__ y, int rows, int d, float eps,
                                                        const float* __restrict__ add, int period, float* __restrict__ y_plain) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = d >> 2;  // float4 per row
    const f32x4* xp = (const f32x4*)(x + (size_t)row * d);
    f32x4 v[8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
            v[j] = xp[i];
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float t = v[j][c] - mean;
                ss += t * t;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)d + eps);
    f32x4* yp = (f32x4*)(y + (size_t)row * d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
            const f32x4 gg = ((const f32x4*)g)[i], bb = ((const f32x4*)b)[i];
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (v[j][c] - mean) * rstd * gg[c] + bb[c];
            if (ADD) {
                if (y_plain) ((f32x4*)(y_plain + (size_t)row * d))[i] = o;
                o += ((const f32x4*)(add + (size_t)(row % period) * d))[i];
            }
            yp[i] = o;
        }
    }
}

// Specialised form for d = 4 * G * NVL with G (lanes per row) a power of two <= 64: a wave holds 64 / G rows side by side
// and UNR such groups in flight (all loads of a wave-iteration are issued before the first reduction), statistics by
// butterfly shuffles inside the G-lane group.  Same arithmetic order per element as the generic kernel.
// HR8: the result also (or only, y == nullptr) in the split form "hr8" of csrc/gemm_ps.hip -- per 8 columns 16 bytes of f16 heads then
// 16 bytes of f16 residuals of 8 x -- for the GEMM that reads it next (a lane owns 4 consecutive columns: half a group of each).
template <int G, int NVL, int UNR, bool ADD, bool HR8 = false>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                             float* y, int rows, float eps,
                                                             const float* __restrict__ add, int period, float* __restrict__ y_plain,
                                                             unsigned char* __restrict__ ys = nullptr, long ldys = 0, int* __restrict__ range_flag = nullptr,
                                                             const float* pre = nullptr) {
    // `y` and `pre` carry no __restrict__: the decoder calls this in place (pre == y, include/bigc_hip.h bigc_layernorm_res_f32) -- a wave
    // reads its rows of `pre` completely before it stores them, and no other wave touches those rows
    // pre != nullptr: the rows are x + pre (a GEMM's output and its residual, models/model_0v10.py:188,211,223: LayerNorm(query + sublayer)).
    // The sum used to be the GEMM's epilogue; there a lane owns a data ROW, so the residual came in as 16 bytes from each of 32 rows per
    // load instruction -- 34 of the 129 us of a decoder 512 x 512 GEMM -- while here a row is read by whole lines.  Same f32 addition, same bits.
    constexpr int RPW = 64 / G;                    // rows per wave pass
    constexpr int D4 = G * NVL;                    // float4 per row
    bool oob = false;
    const int lane = threadIdx.x & 63, sub = lane / G, gl = lane % G;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int row0 = wave * RPW * UNR + sub;
    f32x4 v[UNR][NVL];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int row = min(row0 + u * RPW, rows - 1);
        const f32x4* xp = (const f32x4*)x + (size_t)row * D4;
#pragma unroll
        for (int j = 0; j < NVL; ++j) v[u][j] = xp[gl + G * j];
    }
    if (pre) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const f32x4* pp = (const f32x4*)pre + (size_t)min(row0 + u * RPW, rows - 1) * D4;
#pragma unroll
            for (int j = 0; j < NVL; ++j) v[u][j] += pp[gl + G * j];
        }
    }
    f32x4 gg[NVL], bb[NVL];
#pragma unroll
    for (int j = 0; j < NVL; ++j) { gg[j] = ((const f32x4*)g)[gl + G * j]; bb[j] = ((const f32x4*)b)[gl + G * j]; }
    f32x4 av[ADD ? UNR : 1][NVL];
    if (ADD) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const f32x4* ap = (const f32x4*)add + (size_t)(min(row0 + u * RPW, rows - 1) % period) * D4;
#pragma unroll
            for (int j = 0; j < NVL; ++j) av[u][j] = ap[gl + G * j];
        }
    }
    auto gsum = [](float t) {
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) t += __shfl_xor(t, o);
        return t;
    };
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NVL; ++j) s += (v[u][j][0] + v[u][j][1]) + (v[u][j][2] + v[u][j][3]);
        const float mean = gsum(s) / (float)(4 * D4);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NVL; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float t = v[u][j][c] - mean; ss += t * t; }
        const float rstd = 1.0f / sqrtf(gsum(ss) / (float)(4 * D4) + eps);
        const int row = row0 + u * RPW;
        if (row < rows) {
            f32x4* yp = (f32x4*)y + (size_t)row * D4;
#pragma unroll
            for (int j = 0; j < NVL; ++j) {
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (v[u][j][c] - mean) * rstd * gg[j][c] + bb[j][c];
                if (ADD) {
                    if (y_plain) ((f32x4*)y_plain + (size_t)row * D4)[gl + G * j] = o;
                    o += av[u][j];
                }
                if (!HR8 || y) yp[gl + G * j] = o;
                if (HR8) {
                    const int q = gl + G * j;                       // columns 4 q .. 4 q + 3: half (q & 1) of group q >> 1
                    const f32x4 X = o * 8.0f;                       // PS_ACT_EXP = 3 (linear_internal.h)
                    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                    const h16x4 hh = __builtin_convertvector(X, h16x4);
                    const h16x4 rr = __builtin_convertvector(X - __builtin_convertvector(hh, f32x4), h16x4);
                    unsigned char* gp = ys + (size_t)row * ldys + (size_t)(q >> 1) * 32 + (q & 1) * 8;
                    *(h16x4*)gp = hh;
                    *(h16x4*)(gp + 16) = rr;
                    oob = oob || !(fabsf(X[0]) < 32768.f) || !(fabsf(X[1]) < 32768.f) || !(fabsf(X[2]) < 32768.f) || !(fabsf(X[3]) < 32768.f);
                }
            }
        }
    }
    if (HR8 && oob && range_flag) atomicOr(range_flag, 1);
}

__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ x, const float* __restrict__ p, float* __restrict__ y,
                                                       long n4, int d4, int period) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long row = i / d4;
        const int c = (int)(i - row * d4);
        const f32x4 b = ((const f32x4*)p)[(row % period) * d4 + c];
        ((f32x4*)y)[i] = x ? ((const f32x4*)x)[i] + b : b;
    }
}

__global__ void fill_offsets_kernel(int32_t* p, int n, int step) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = i * step;
}

}  // namespace

// internal: p[i] = i*step (uniform sequence offsets of the 192-query decoder self-attention)
int bigc_fill_offsets(hipStream_t s, int32_t* p, int n, int step) {
    hipLaunchKernelGGL(fill_offsets_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, n, step);
    BIGC_LAUNCH_CHECK("fill_offsets");
    return BIGC_OK;
}

extern "C" int bigc_layernorm_add_f32(void* stream, const float* x, const float* gamma, const float* beta, float* y, int rows, int d,
                                      float eps, const float* add, int period, float* y_plain) {
    BIGC_CHECK_ARG(x && gamma && beta && y, "layernorm: null pointer");
    BIGC_CHECK_ARG(d > 0 && d % 4 == 0 && d <= 2048, "layernorm: d=%d must be a multiple of 4 and <= 2048", d);
    BIGC_CHECK_ARG(add ? period > 0 : !y_plain, "layernorm: add needs a period, y_plain needs add");
    if (rows <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_LAYERNORM, 0.0, (y_plain ? 12.0 : 8.0) * rows * (double)d);
    hipStream_t hs = (hipStream_t)stream;
#define LN_ROWS_(G, NVL, UNR, ADD) hipLaunchKernelGGL((layernorm_rows_kernel<G, NVL, UNR, ADD>), dim3(cdiv(rows, 4 * (64 / G) * UNR)), dim3(256), 0, hs, x, gamma, beta, y, rows, eps, add, period, y_plain)
#define LN_ROWS(G, NVL, UNR) do { if (add) LN_ROWS_(G, NVL, UNR, true); else LN_ROWS_(G, NVL, UNR, false); } while (0)
    if (d == 512) LN_ROWS(64, 2, 4);
    else if (d == 256) LN_ROWS(64, 1, 4);
    else if (d == 128) LN_ROWS(32, 1, 4);
    else if (d == 64) LN_ROWS(16, 1, 4);
    else if (add) hipLaunchKernelGGL(layernorm_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, hs, x, gamma, beta, y, rows, d, eps, add, period, y_plain);
    else hipLaunchKernelGGL(layernorm_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, hs, x, gamma, beta, y, rows, d, eps, add, period, y_plain);
#undef LN_ROWS
#undef LN_ROWS_
    BIGC_LAUNCH_CHECK("layernorm");
    return BIGC_OK;
}

// LayerNorm whose result a GEMM on pre-split operands reads next: y (f32, may be null) and / or ys (split form, pitch ldys bytes)
int bigc_layernorm_hr8(hipStream_t hs, const float* x, const float* gamma, const float* beta, float* y, void* ys, long ldys, int rows, int d,
                       float eps, int* range_flag) {
    BIGC_CHECK_ARG(x && gamma && beta && ys && ldys % 16 == 0 && ldys >= 4L * d, "layernorm_hr8: bad argument");
    BIGC_CHECK_ARG(d == 512 || d == 256 || d == 128 || d == 64, "layernorm_hr8: d=%d not in {64, 128, 256, 512}", d);
    if (rows <= 0) return BIGC_OK;
    ProfScope ps((void*)hs, FAM_LAYERNORM, 0.0, (y ? 12.0 : 8.0) * rows * (double)d);
#define LN_HR8(G, NVL, UNR) hipLaunchKernelGGL((layernorm_rows_kernel<G, NVL, UNR, false, true>), dim3(cdiv(rows, 4 * (64 / G) * UNR)), dim3(256), 0, hs, x, gamma, beta, y, rows, eps, nullptr, 0, nullptr, (unsigned char*)ys, ldys, range_flag)
    if (d == 512) LN_HR8(64, 2, 4);
    else if (d == 256) LN_HR8(64, 1, 4);
    else if (d == 128) LN_HR8(32, 1, 4);
    else LN_HR8(16, 1, 4);
#undef LN_HR8
    BIGC_LAUNCH_CHECK("layernorm_hr8");
    return BIGC_OK;
}

// LayerNorm(x + pre) -> y (f32, + add[row % period] when `add` is given) and / or ys (split form); `pre` may alias y (a row is read before it
// is written, by the lanes that write it).  d in {64, 128, 256, 512}.
int bigc_layernorm_res(hipStream_t hs, const float* x, const float* pre, const float* gamma, const float* beta, float* y, void* ys, long ldys,
                       int rows, int d, float eps, const float* add, int period, int* range_flag) {
    BIGC_CHECK_ARG(x && pre && gamma && beta && (y || ys) && !(add && ys) && (!add || period > 0), "layernorm_res: bad argument");
    BIGC_CHECK_ARG(d == 512 || d == 256 || d == 128 || d == 64, "layernorm_res: d=%d not in {64, 128, 256, 512}", d);
    BIGC_CHECK_ARG(!ys || (ldys % 16 == 0 && ldys >= 4L * d), "layernorm_res: bad split-form pitch");
    if (rows <= 0) return BIGC_OK;
    ProfScope ps((void*)hs, FAM_LAYERNORM, 0.0, ((y && ys) ? 16.0 : 12.0) * rows * (double)d);
#define LN_RES(G, NVL, UNR) do { \
        const dim3 grid(cdiv(rows, 4 * (64 / G) * UNR)); \
        if (ys) hipLaunchKernelGGL((layernorm_rows_kernel<G, NVL, UNR, false, true>), grid, dim3(256), 0, hs, x, gamma, beta, y, rows, eps, nullptr, 0, nullptr, (unsigned char*)ys, ldys, range_flag, pre); \
        else if (add) hipLaunchKernelGGL((layernorm_rows_kernel<G, NVL, UNR, true, false>), grid, dim3(256), 0, hs, x, gamma, beta, y, rows, eps, add, period, nullptr, nullptr, 0L, nullptr, pre); \
        else hipLaunchKernelGGL((layernorm_rows_kernel<G, NVL, UNR, false, false>), grid, dim3(256), 0, hs, x, gamma, beta, y, rows, eps, nullptr, 0, nullptr, nullptr, 0L, nullptr, pre); } while (0)
    if (d == 512) LN_RES(64, 2, 4);
    else if (d == 256) LN_RES(64, 1, 4);
    else if (d == 128) LN_RES(32, 1, 4);
    else LN_RES(16, 1, 4);
#undef LN_RES
    BIGC_LAUNCH_CHECK("layernorm_res");
    return BIGC_OK;
}

extern "C" int bigc_layernorm_res_f32(void* stream, const float* x, const float* pre, const float* gamma, const float* beta, float* y, void* ys,
                                      long ldys_bytes, int rows, int d, float eps, const float* add, int period, int* range_flag) {
    return bigc_layernorm_res((hipStream_t)stream, x, pre, gamma, beta, y, ys, ldys_bytes, rows, d, eps, add, period, range_flag);
}

extern "C" int bigc_layernorm_hr8_f32(void* stream, const float* x, const float* gamma, const float* beta, float* y, void* ys, long ldys_bytes,
                                      int rows, int d, float eps, int* range_flag) {
    return bigc_layernorm_hr8((hipStream_t)stream, x, gamma, beta, y, ys, ldys_bytes, rows, d, eps, range_flag);
}

extern "C" int bigc_layernorm_f32(void* stream, const float* x, const float* gamma, const float* beta, float* y,
                                  int rows, int d, float eps) {
    return bigc_layernorm_add_f32(stream, x, gamma, beta, y, rows, d, eps, nullptr, 0, nullptr);
}

extern "C" int bigc_add_rows_f32(void* stream, const float* x, const float* p, float* y, int rows, int d, int period) {
    BIGC_CHECK_ARG(p && y, "add_rows: null pointer");
    BIGC_CHECK_ARG(d > 0 && d % 4 == 0 && period > 0, "add_rows: d=%d must be a multiple of 4", d);
    if (rows <= 0) return BIGC_OK;
    const long n4 = (long)rows * (d / 4);
    ProfScope ps(stream, FAM_ADD_ROWS, 0.0, (x ? 8.0 : 4.0) * rows * (double)d);
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, p, y, n4, d / 4, period);
    BIGC_LAUNCH_CHECK("add_rows");
    return BIGC_OK;
}
