// Backward kernels of the BIG-C training step (reference: the autograd graph of
// models/model_0v10.py:434-507 + the loss :642-704).  Dense gradients reuse the f32-MFMA machinery:
//   dX = dY W          -> linear_f32 on a transposed copy of W (transpose kernel below)
//   dW += dY^T X       -> linear_dw kernel (same 128x128 MFMA tile, reduction over the rows M, split over
//                         M across workgroups, fp32 atomic accumulation into the gradient buffer)
// plus the row-wise / gather backward kernels (ReLU mask, LayerNorm, column sums, max-pool routing,
// stretched-conv gather-sum, head scatter, periodic reduction for the shared query/pos embeddings).
#include "common.h"
#include "prof.h"
#include <map>
#include <unordered_map>
#include <utility>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int PITCH = 36;

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        if (r < rows && c < cols) tile[j][tx] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = bx + j, c = by + tx;                      // out is (cols, rows)
        if (r < cols && c < rows) out[(size_t)r * rows + c] = tile[tx][j];
    }
}

// ---------------------------------------------------------------------------------------------
// dW[n,k] += sum_m dY[m,n] * X'[m,k]; X' optionally row-gathered in segments like linear_f32.
struct DwArgs {
    const float* dY; int ldy; const float* X; int ldx; const int32_t* x_idx; int nseg; int seg_len;
    float* dW; int ldw; float* db; int M, N, K; int tiles_n, tiles_k, m_chunk;
    float* partial;        // when the M range is split: (splits, tiles_n*128, tiles_k*128) partial products
    int* arrived;          // per tile: how many of its splits have stored their partial product (nullptr: a second kernel sums them)
    int splits;
    float* db_part = nullptr;   // when the M range is split and db is given: (splits, tiles_n*128) column sums of dY, added in split order by the reduction
    // row m of X starts at x_rowptr[m] (16-byte aligned, K floats) instead of X + m * ldx: the operand is read in place from rows that live
    // in separate allocations (bigc_batch.feat_row_ptr; split kernel, no gather)
    const uint64_t* x_rowptr = nullptr;
};
typedef const __attribute__((address_space(1))) float* dw_gptr;
template <bool ROWS>
__device__ __forceinline__ float dw_x_at(const DwArgs& p, int m, int k) {
    return ROWS ? ((dw_gptr)p.x_rowptr[m])[k] : p.X[(size_t)m * p.ldx + k];
}

// Split-M tiles without a second launch: every workgroup stores its partial tile, and the one that arrives LAST at a tile adds
// all of them to dW in split order -- which workgroup does the summing varies from run to run, the order of the sum does not, so
// the gradients stay bit-reproducible (no float atomics).  The counter is left at zero for the next launch.
__device__ __forceinline__ void dw_finish_tile(const DwArgs& p, int tile, int n0, int k0) {
    __shared__ int s_last;
    __threadfence();                        // partial tile visible device-wide before the arrival is counted
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&p.arrived[tile], 1) == p.splits - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();                        // acquire: the other workgroups' partial tiles
    const size_t k_pad = (size_t)p.tiles_k * 128, plane = (size_t)p.tiles_n * 128 * k_pad;
    for (int i = threadIdx.x; i < 128 * 32; i += 256) {
        const int row = n0 + i / 32, col = k0 + (i % 32) * 4;
        if (row >= p.N || col >= p.K) continue;
        const float* src = p.partial + (size_t)row * k_pad + col;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < p.splits; ++sp) {
            const f32x4 v = *(const f32x4*)(src + (size_t)sp * plane);
            acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
        }
        float* dst = p.dW + (size_t)row * p.ldw + col;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (col + c < p.K) dst[c] += acc[c];
    }
    if (p.db_part && k0 == 0 && threadIdx.x < 128 && n0 + (int)threadIdx.x < p.N) {
        float sb = 0.f;
        for (int sp = 0; sp < p.splits; ++sp) sb += p.db_part[(size_t)sp * p.tiles_n * 128 + n0 + threadIdx.x];
        p.db[n0 + threadIdx.x] += sb;
    }
    if (threadIdx.x == 0) p.arrived[tile] = 0;
}

// The bias gradient (column sum of dY over this workgroup's M range; called by the tk == 0 workgroups, uniformly): the two half-slab
// partial sums of a column meet in LDS (`lds`: 128 floats of the staging buffer, free after the K loop) and are added in a fixed order;
// a single split adds its sum to db itself (one writer per column), several splits store theirs to db_part for the reduction stage to
// add in split order.  No float atomics: the gradient's bits do not depend on the order the workgroups ran in.
__device__ __forceinline__ void dw_bias_out(const DwArgs& p, float* lds, float bsum, int c, int mh, int split, int n0, bool nval) {
    __syncthreads();                        // the last slab's fragment reads are done
    if (mh == 1) lds[c] = bsum;
    __syncthreads();
    if (mh == 0 && nval) {
        const float tot = bsum + lds[c];
        if (p.db_part) p.db_part[(size_t)split * p.tiles_n * 128 + n0 + c] = tot;
        else p.db[n0 + c] += tot;
    }
}

__global__ __launch_bounds__(256, 2) void linear_dw_kernel(const DwArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2][128 * PITCH];   // [dY^T tile | X^T tile], each [128][32(m)+pad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.x % (p.tiles_n * p.tiles_k), split = blockIdx.x / (p.tiles_n * p.tiles_k);
    const int tn = tile / p.tiles_k, tk = tile % p.tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int m_begin = split * p.m_chunk, m_end = min(p.M, m_begin + p.m_chunk);
    if (m_begin >= m_end) return;
    // staging: thread owns column c = tid&127 of the tile and 16 consecutive m (half mh of the 32-row slab)
    const int c = tid & 127, mh = tid >> 7;
    const int ncol = n0 + c, kcol = k0 + c;
    const bool nval = ncol < p.N, kval = kcol < p.K;
    int seg = 0, kin = kcol;
    if (p.x_idx) { seg = min(kcol, p.K - 1) / p.seg_len; kin = kcol - seg * p.seg_len; }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float bsum = 0.f;                       // column sum of dY (bias gradient), taken by the tk == 0 workgroups
    for (int m0 = m_begin; m0 < m_end; m0 += 32) {
        float ry[16], rx[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = m0 + mh * 16 + j;
            const bool mv = m < m_end;
            ry[j] = (mv && nval) ? p.dY[(size_t)m * p.ldy + ncol] : 0.f;
            bsum += ry[j];
            float xv = 0.f;
            if (mv && kval) {
                if (p.x_idx) {
                    const int src = p.x_idx[(size_t)m * p.nseg + seg];
                    xv = src >= 0 ? p.X[(size_t)src * p.ldx + kin] : 0.f;
                } else {
                    xv = p.X[(size_t)m * p.ldx + kcol];
                }
            }
            rx[j] = xv;
        }
        __syncthreads();   // previous slab fully consumed
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            *(f32x4*)(&smem[0][c * PITCH + mh * 16 + j]) = f32x4{ry[j], ry[j + 1], ry[j + 2], ry[j + 3]};
            *(f32x4*)(&smem[1][c * PITCH + mh * 16 + j]) = f32x4{rx[j], rx[j + 1], rx[j + 2], rx[j + 3]};
        }
        __syncthreads();
        const float* as = &smem[0][(wm * 64 + l31) * PITCH + 4 * hi];
        const float* bs = &smem[1][(wn * 64 + l31) * PITCH + 4 * hi];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const f32x4 a0 = *(const f32x4*)(as + 8 * cc), a1 = *(const f32x4*)(as + 32 * PITCH + 8 * cc);
            const f32x4 b0 = *(const f32x4*)(bs + 8 * cc), b1 = *(const f32x4*)(bs + 32 * PITCH + 8 * cc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0][0] = mfma32(a0[q], b0[q], acc[0][0]);
                acc[0][1] = mfma32(a0[q], b1[q], acc[0][1]);
                acc[1][0] = mfma32(a1[q], b0[q], acc[1][0]);
                acc[1][1] = mfma32(a1[q], b1[q], acc[1][1]);
            }
        }
    }
    if (p.db && tk == 0) dw_bias_out(p, &smem[0][0], bsum, c, mh, split, n0, nval);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = k0 + wn * 64 + b * 32 + l31;
            if (col >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * 64 + a * 32 + mfma_row(r, hi);
                if (row >= p.N) continue;
                if (p.partial) p.partial[((size_t)split * p.tiles_n * 128 + row) * ((size_t)p.tiles_k * 128) + col] = acc[a][b][r];
                else p.dW[(size_t)row * p.ldw + col] += acc[a][b][r];      // one workgroup per tile: a single writer per element
            }
        }
    if (p.partial && p.arrived) dw_finish_tile(p, tile, n0, k0);
}

// The same product on the f16 matrix cores with exactly split f32 operands (see linear_split.hip): both operands are
// activations here, so both are split by the staging code (columns of dY and X, 16 consecutive m per thread) into the four f16
// planes [n or k][32 m] of a slab -- the layout and the MFMA loop of the forward kernel with m as the contraction index.
// Slab s+1 is fetched into registers while slab s is multiplied.  Out-of-range / non-finite operands: the workgroup redoes its
// (tile, M range) on the exact f32 MFMA.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 hp16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dw_swz(int chunk, int row) { return chunk ^ ((row >> 2) & 3); }

// 16 consecutive-m values of one column -> 32 bytes of heads + 32 bytes of scaled residuals, written into the column's LDS row
// (scalar f32 arithmetic on purpose: the same split written on 2-vectors compiles to v_pk_add_f32 / v_pk_mul_f32, 19 % fewer VALU
// instructions and 7 % MORE time -- packed f32 instructions stall the issue beside MFMAs)
__device__ __forceinline__ void dw_split16(const float (&x)[16], unsigned char* head_row, unsigned char* resid_row, int row, int mh,
                                           float& chk, float& amax) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {               // two 16-byte chunks of 8 m each
        union { hp16x2 h[4]; uint4 u; } hh, rr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = x[8 * g + 2 * j], b = x[8 * g + 2 * j + 1];
            hh.h[j] = bigc_pk_rtn(a, b);
            // (a - h) 2048 as ONE mixed-precision FMA on the f16 head: a 2048 - h 2048, every step exact (a - h has at most 13 significant
            // bits), so the same bits as convert, subtract, scale -- three VALU instructions per element on a loop that is VALU-bound
            float ra, rb;
            const float a2 = a * 2048.0f, b2 = b * 2048.0f;
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hh.h[j]), "v"(-2048.0f), "v"(a2));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hh.h[j]), "v"(-2048.0f), "v"(b2));
            rr.h[j] = bigc_pk_rtn(ra, rb);
            // range: |x| >= 32768 or Inf shows in amax; a NaN operand (which v_max3 drops) makes its products NaN on either path
            asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(amax) : "v"(amax), "v"(a), "v"(b));
        }
        const int off = row * 64 + dw_swz(2 * mh + g, row) * 16;
        *(uint4*)(head_row + off) = hh.u;
        *(uint4*)(resid_row + off) = rr.u;
    }
}

template <bool GATHER, bool ROWS = false>
__device__ __forceinline__ void linear_dw_split_body(const DwArgs& p, const int bid) {
    // [buffer][dY^T head | dY^T resid | X^T head | X^T resid][128 rows][64 bytes = 32 m]
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][4][128 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, l31 = lane & 31;
    const int tile = bid % (p.tiles_n * p.tiles_k), split = bid / (p.tiles_n * p.tiles_k);
    const int tn = tile / p.tiles_k, tk = tile % p.tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int m_begin = split * p.m_chunk, m_end = min(p.M, m_begin + p.m_chunk);
    if (m_begin >= m_end) return;
    const int c = tid & 127;
    const int mh = __builtin_amdgcn_readfirstlane(tid >> 7);                   // wave-uniform: the rows a wave loads are scalar addresses
    const int ncol = min(n0 + c, p.N - 1), kcol = min(k0 + c, p.K - 1);       // clamped: surplus rows of dW are never stored
    const bool nval = n0 + c < p.N;
    int seg = 0, kin = kcol;
    if (GATHER) { seg = kcol / p.seg_len; kin = kcol - seg * p.seg_len; }

    f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[a][b][r] = 0.f; acc1[a][b][r] = 0.f; }
    int aoff[2][2], boff[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ra_ = wm * 64 + mi * 32 + l31, rb_ = wn * 64 + mi * 32 + l31;
            aoff[mi][s] = ra_ * 64 + dw_swz(2 * s + hi, ra_) * 16;
            boff[mi][s] = rb_ * 64 + dw_swz(2 * s + hi, rb_) * 16;
        }

    float ry[16], rx[16];
    // (offsets inside a split must fit the 32-bit scalar offset of a buffer load)
    const bool buf_ok = (size_t)(m_end - m_begin + 32) * (size_t)max(p.ldy, (GATHER || ROWS) ? 0 : p.ldx) * 4 < ((size_t)1 << 31);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dY + (size_t)m_begin * p.ldy), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((GATHER || ROWS) ? p.dY : p.X + (size_t)m_begin * p.ldx), 0, 0x7fffffff, 0x00020000);
    int ridx[16];                                   // gather mode: source rows of the NEXT slab to load (fetched one slab ahead,
                                                    // so the data loads never wait on their own index loads)
    float bsum = 0.f, chk = 0.f, amax = 0.f;
    const bool do_bias = p.db && tk == 0;           // (uniform) the bias gradient is the column sum of dY: only the first k tile adds it up
    auto load_idx = [&](int m0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) ridx[j] = p.x_idx[(size_t)min(m0 + mh * 16 + j, m_end - 1) * p.nseg + seg];
    };
    // The loop is VALU-bound (rocprofv3 --pmc, benchmarks/pmc_dw.sh: 15 VALU instructions per MFMA, the matrix pipe 27 % busy and
    // hardly ever at the same time as the VALU): a full slab -- all but the last of a split -- is 32 loads off wave-uniform row
    // bases without clamps or selects
    auto load_slab = [&](int m0) {
        const int mb = m0 + mh * 16;
        if (mb + 16 <= m_end) {
            if (buf_ok) {
                // buffer loads: descriptor of this split's rows in scalar registers, the lane's column as the vector offset, the row as the
                // scalar offset -- no vector address arithmetic (plain pointers cost a 64-bit vector add per load: a fifth of the VALU
                // instructions of a loop that is VALU-bound)
                const int ro = (mb - m_begin) * p.ldy * 4;
#pragma unroll
                for (int j = 0; j < 16; ++j) ry[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(y_rsrc, ncol * 4, ro + j * p.ldy * 4, 0));
            } else {
                const float* yp = p.dY + (size_t)mb * p.ldy + ncol;
#pragma unroll
                for (int j = 0; j < 16; ++j) ry[j] = yp[(size_t)j * p.ldy];
            }
            if (GATHER) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float xv = p.X[(size_t)max(ridx[j], 0) * p.ldx + kin];
                    rx[j] = ridx[j] < 0 ? 0.f : xv;
                }
            } else if (ROWS) {                       // sixteen scalar address loads per wave and slab
#pragma unroll
                for (int j = 0; j < 16; ++j) rx[j] = ((dw_gptr)p.x_rowptr[mb + j])[kcol];
            } else {
                if (buf_ok) {
                    const int ro = (mb - m_begin) * p.ldx * 4;
#pragma unroll
                    for (int j = 0; j < 16; ++j) rx[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, kcol * 4, ro + j * p.ldx * 4, 0));
                } else {
                    const float* xp = p.X + (size_t)mb * p.ldx + kcol;
#pragma unroll
                    for (int j = 0; j < 16; ++j) rx[j] = xp[(size_t)j * p.ldx];
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = mb + j;
            const bool mv = m < m_end;
            const int mc = min(m, m_end - 1);
            const float y = p.dY[(size_t)mc * p.ldy + ncol];
            float xv;
            if (GATHER) {
                xv = p.X[(size_t)max(ridx[j], 0) * p.ldx + kin];
                if (ridx[j] < 0) xv = 0.f;
            } else {
                xv = dw_x_at<ROWS>(p, mc, kcol);
            }
            ry[j] = mv ? y : 0.f;                    // rows past the M range contribute nothing
            rx[j] = mv ? xv : 0.f;
        }
    };
    auto store_slab = [&](int buf) {
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 16; ++j) bsum += ry[j];
        }
        dw_split16(ry, smem[buf][0], smem[buf][1], c, mh, chk, amax);
        dw_split16(rx, smem[buf][2], smem[buf][3], c, mh, chk, amax);
    };
    if (GATHER) load_idx(m_begin);
    load_slab(m_begin);
    if (GATHER) load_idx(m_begin + 32);
    store_slab(0);
    __syncthreads();
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += 32, buf ^= 1) {
        const bool more = m0 + 32 < m_end;
        if (more) {
            load_slab(m0 + 32);
            if (GATHER) load_idx(m0 + 64);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h16x8 a1[2], a2[2], b1[2], b2[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                a1[mi] = *(const h16x8*)(&smem[buf][0][aoff[mi][s]]);
                a2[mi] = *(const h16x8*)(&smem[buf][1][aoff[mi][s]]);
                b1[mi] = *(const h16x8*)(&smem[buf][2][boff[mi][s]]);
                b2[mi] = *(const h16x8*)(&smem[buf][3][boff[mi][s]]);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc0[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[mi], b1[ni], acc0[mi][ni], 0, 0, 0);
                    acc1[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[mi], b2[ni], acc1[mi][ni], 0, 0, 0);
                    acc1[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[mi], b1[ni], acc1[mi][ni], 0, 0, 0);
                }
        }
        if (more) store_slab(buf ^ 1);
        __syncthreads();
    }
    if (do_bias) dw_bias_out(p, (float*)&smem[0][0][0], bsum, c, mh, split, n0, nval);
    if (__syncthreads_or((!(fabsf(chk) < INFINITY) || amax >= 32768.0f) ? 1 : 0)) {
        // exact f32 MFMA straight from memory: lane (l31, hi) of block a / b owns row n / column k, m = m2 + hi
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[a][b][r] = 0.f; acc1[a][b][r] = 0.f; }
        for (int m2 = m_begin; m2 < m_end; m2 += 2) {
            const int m = m2 + hi;
            const bool mv = m < m_end;
            const int mc = min(m, m_end - 1);
            float av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int n = min(n0 + wm * 64 + a * 32 + l31, p.N - 1), k = min(k0 + wn * 64 + a * 32 + l31, p.K - 1);
                av[a] = mv ? p.dY[(size_t)mc * p.ldy + n] : 0.f;
                if (GATHER) {
                    const int sg = k / p.seg_len;
                    const int src = p.x_idx[(size_t)mc * p.nseg + sg];
                    bv[a] = (mv && src >= 0) ? p.X[(size_t)max(src, 0) * p.ldx + (k - sg * p.seg_len)] : 0.f;
                } else {
                    bv[a] = mv ? dw_x_at<ROWS>(p, mc, k) : 0.f;
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc0[a][b] = mfma32(av[a], bv[b], acc0[a][b]);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = k0 + wn * 64 + b * 32 + l31;
            if (col >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * 64 + a * 32 + mfma_row(r, hi);
                if (row >= p.N) continue;
                const float v = acc0[a][b][r] + acc1[a][b][r] * (1.0f / 2048.0f);
                if (p.partial) p.partial[((size_t)split * p.tiles_n * 128 + row) * ((size_t)p.tiles_k * 128) + col] = v;
                else p.dW[(size_t)row * p.ldw + col] += v;
            }
        }
    if (p.partial && p.arrived) dw_finish_tile(p, tile, n0, k0);
}

template <bool GATHER, bool ROWS = false>
__global__ __launch_bounds__(256, 2) void linear_dw_split_kernel(const DwArgs p) { linear_dw_split_body<GATHER, ROWS>(p, blockIdx.x); }

// Several weight-gradient products in ONE launch (the eleven of a decoder layer: 512 x 512 .. 1024 x 512 outputs whose
// contraction over the B*Q rows has to be split to fill the chip -- one by one they were 16 tiles x 32 splits of twelve slabs
// each plus a reduction launch per product; together they are 224 tiles, two splits, one reduction launch).  The workgroups of
// job j are start[j] .. start[j + 1] - 1.
constexpr int DW_GROUP_MAX = 12;
struct DwGroup { int n; int start[DW_GROUP_MAX + 1]; DwArgs job[DW_GROUP_MAX]; };

__global__ __launch_bounds__(256, 2) void linear_dw_group_kernel(const DwGroup g) {
    int j = 0;
    while (j + 1 < g.n && (int)blockIdx.x >= g.start[j + 1]) ++j;
    linear_dw_split_body<false>(g.job[j], (int)blockIdx.x - g.start[j]);
}

// the reductions of a group's partial products in one launch: workgroups start[j] .. start[j + 1] - 1 walk job j's N x K elements
__global__ __launch_bounds__(256) void dw_reduce_group_kernel(const DwGroup g) {
    int j = 0;
    while (j + 1 < g.n && (int)blockIdx.x >= g.start[j + 1]) ++j;
    const DwArgs& p = g.job[j];
    if (!p.partial) return;
    const int nb = g.start[j + 1] - g.start[j], b = (int)blockIdx.x - g.start[j];
    const size_t n_pad = (size_t)p.tiles_n * 128, k_pad = (size_t)p.tiles_k * 128;
    if (p.db_part)                          // the bias gradient's per-split column sums, added in split order
        for (int n = b * 256 + threadIdx.x; n < p.N; n += nb * 256) {
            float sb = 0.f;
            for (int sp = 0; sp < p.splits; ++sp) sb += p.db_part[(size_t)sp * n_pad + n];
            p.db[n] += sb;
        }
    if (((p.K | p.ldw) & 3) == 0 && ((uintptr_t)p.dW & 15) == 0) {      // (gradient views of the flat buffer need not be 16-byte aligned)
        // four consecutive k per thread (16-byte accesses), one or two passes per thread
        const int K4 = p.K >> 2;
        const long total = (long)p.N * K4;
        for (long i = (long)b * 256 + threadIdx.x; i < total; i += (long)nb * 256) {
            const int n = (int)(i / K4), k = (int)(i - (long)n * K4) * 4;
            const float* src = p.partial + (size_t)n * k_pad + k;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int sp = 0; sp < p.splits; ++sp) s += *(const f32x4*)(src + (size_t)sp * n_pad * k_pad);
            f32x4* dst = (f32x4*)(p.dW + (size_t)n * p.ldw + k);
            *dst = *dst + s;
        }
        return;
    }
    const long total = (long)p.N * p.K;
    for (long i = (long)b * 256 + threadIdx.x; i < total; i += (long)nb * 256) {
        const int n = (int)(i / p.K), k = (int)(i - (long)n * p.K);
        const float* src = p.partial + (size_t)n * k_pad + k;
        float s = 0.f;
        for (int sp = 0; sp < p.splits; ++sp) s += src[(size_t)sp * n_pad * k_pad];
        p.dW[(size_t)n * p.ldw + k] += s;
    }
}

// dW[n, k] += sum over the M splits of their partial products (no atomics: summation order is fixed -> deterministic gradients)
__global__ __launch_bounds__(256) void dw_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int ldw, int N, int K,
                                                        int n_pad, int k_pad, int splits, const float* __restrict__ db_part, float* __restrict__ db) {
    if (db_part)                            // the bias gradient's per-split column sums, added in split order
        for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
            float sb = 0.f;
            for (int sp = 0; sp < splits; ++sp) sb += db_part[(size_t)sp * n_pad + n];
            db[n] += sb;
        }
    const long total = (long)N * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (long)n * K);
        const float* p = partial + (size_t)n * k_pad + k;
        const size_t stride = (size_t)n_pad * k_pad;
        float s = 0.f;
        int sp = 0;
        for (; sp + 8 <= splits; sp += 8) {                      // eight partial products in flight, added in split order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(sp + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sp < splits; ++sp) s += p[(size_t)sp * stride];
        dW[(size_t)n * ldw + k] += s;
    }
}

// dY *= (Y > 0)   (Y = post-ReLU activation)
__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ dY, const float* __restrict__ Y, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 g = ((f32x4*)dY)[i];
        const f32x4 y = ((const f32x4*)Y)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) g[c] = y[c] > 0.f ? g[c] : 0.f;
        ((f32x4*)dY)[i] = g;
    }
}

// y += x  (gradient accumulation of two branches)
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        ((f32x4*)y)[i] = ((f32x4*)y)[i] + ((const f32x4*)x)[i];
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// LayerNorm backward. One wave per row (grid-stride), d <= 2048. dgamma/dbeta accumulated per wave in registers; the four waves of
// a workgroup meet in LDS and the workgroup stores ONE partial row pair to `part` (gridDim.x, 2, d); partial_rows_finish_kernel adds
// the workgroups' rows in block order (no float atomics: the parameter gradients are bit-reproducible from run to run).
// NVL = float4 per lane (d <= 256 * NVL): the per-lane arrays are exactly as long as the row needs (a fixed length of 8
// spilled 688 bytes per lane at d = 512)
template <int NVL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ dy, float* __restrict__ dx,
                                                            float* __restrict__ part, int rows, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int nv = d >> 2;
    f32x4 ag[NVL], ab[NVL];
#pragma unroll
    for (int j = 0; j < NVL; ++j) { ag[j] = f32x4{0, 0, 0, 0}; ab[j] = f32x4{0, 0, 0, 0}; }
    // the next row of the wave is fetched while the current one is reduced (a wave walks >= 8 rows one after the other: without this
    // every row waited out its own loads, 31 us per launch for 75 MB)
    f32x4 nx[NVL], ny[NVL];
    auto fetch = [&](int row) {
        const f32x4* xp = (const f32x4*)(x + (size_t)row * d);
        const f32x4* yp = (const f32x4*)(dy + (size_t)row * d);
#pragma unroll
        for (int j = 0; j < NVL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) { nx[j] = xp[i]; ny[j] = yp[i]; }
        }
    };
    if (wid < rows) fetch(wid);
    for (int row = wid; row < rows; row += nw) {
        f32x4 v[NVL], gy[NVL];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NVL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) { v[j] = nx[j]; gy[j] = ny[j]; s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]); }
        }
        if (row + nw < rows) fetch(row + nw);
        const float mean = wsum(s) / (float)d;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NVL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv)
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float t = v[j][c] - mean; ss += t * t; }
        }
        const float rstd = 1.0f / sqrtf(wsum(ss) / (float)d + eps);
        float s1 = 0.f, s2 = 0.f;   // sum(dxhat), sum(dxhat * xhat)
#pragma unroll
        for (int j = 0; j < NVL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                const f32x4 gg = ((const f32x4*)g)[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xh = (v[j][c] - mean) * rstd;
                    const float dxh = gy[j][c] * gg[c];
                    s1 += dxh; s2 += dxh * xh;
                    ag[j][c] += gy[j][c] * xh;
                    ab[j][c] += gy[j][c];
                    v[j][c] = xh; gy[j][c] = dxh;
                }
            }
        }
        s1 = wsum(s1) / (float)d; s2 = wsum(s2) / (float)d;
        f32x4* op = (f32x4*)(dx + (size_t)row * d);
#pragma unroll
        for (int j = 0; j < NVL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = rstd * (gy[j][c] - s1 - v[j][c] * s2);
                op[i] = o;
            }
        }
    }
    // the four waves of the workgroup are summed through LDS, in wave order: one partial row pair per workgroup
    __shared__ float red[2][4][256 * NVL];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NVL; ++j) {
        const int i = lane + 64 * j;
        if (i < nv)
#pragma unroll
            for (int c = 0; c < 4; ++c) { red[0][wv][i * 4 + c] = ag[j][c]; red[1][wv][i * 4 + c] = ab[j][c]; }
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * 2 * d;
    for (int c = threadIdx.x; c < d; c += 256) {
        out[c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        out[d + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

// out[c] += part[0][c] + part[1][c] + ... over `blocks` partial rows of `w` columns, in a FIXED order: thread (g, c) of a 16 x 64 workgroup
// adds the rows g, g + 16, g + 32, ... (eight loads in flight), the sixteen row groups of a column are then added in group order.
// Columns [0, split) are added to out, columns [split, w) to out2 (LayerNorm's dgamma | dbeta, a depth-wise conv's weight | bias gradient).
__global__ __launch_bounds__(1024) void partial_rows_finish_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ out2,
                                                                    int blocks, int w, int split) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < w) {
        int b = g;
        for (; b + 7 * 16 < blocks; b += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 16 * u) * w + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < blocks; b += 16) s += part[(size_t)b * w + c];
    }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < w) {
        float t = red[0][cl];
#pragma unroll
        for (int u = 1; u < 16; ++u) t += red[u][cl];
        if (c >= split) { if (out2) out2[c - split] += t; }
        else out[c] += t;
    }
}

// Several partial-row sets in ONE launch (the LayerNorm parameter gradients of a whole backward pass: 22 finishes of 5 us each otherwise).
// blockIdx.y = job; same arithmetic per job as partial_rows_finish_kernel.
constexpr int FINISH_JOBS_MAX = 32;
struct FinishJob { const float* part; float* out; float* out2; int blocks, w, split; };
struct FinishJobs { int n; FinishJob job[FINISH_JOBS_MAX]; };
__global__ __launch_bounds__(1024) void partial_rows_finish_multi_kernel(const FinishJobs js) {
    const FinishJob& j = js.job[blockIdx.y];
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    if (blockIdx.x * 64 >= j.w) return;                 // (uniform: jobs narrower than the widest)
    float s = 0.f;
    if (c < j.w) {
        int b = g;
        for (; b + 7 * 16 < j.blocks; b += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = j.part[(size_t)(b + 16 * u) * j.w + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < j.blocks; b += 16) s += j.part[(size_t)b * j.w + c];
    }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < j.w) {
        float t = red[0][cl];
#pragma unroll
        for (int u = 1; u < 16; ++u) t += red[u][cl];
        if (c >= j.split) { if (j.out2) j.out2[c - j.split] += t; }
        else j.out[c] += t;
    }
}

// out[r % period, :] += sum over the rows with that residue (gradients of pos_embedding / pred_query_init)
__global__ __launch_bounds__(256) void reduce_periodic_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int d, int period) {
    const int total = period * d;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int pr = e / d, c = e - pr * d;
        float s = 0.f;
        int r = pr;
        for (; r + 7 * period < rows; r += 8 * period) {          // eight rows in flight, added in row order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(size_t)(r + u * period) * d + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < rows; r += period) s += x[(size_t)r * d + c];
        out[e] += s;
    }
}

// adaptive max-pool backward: route d pooled[t, c*P+p] to the FIRST maximum of its window (torch stores the
// index of the first maximum); windows of one (tracklet, channel) are handled by one thread (they overlap when T_o < P)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ conv, const int32_t* __restrict__ conv_off,
                                                          const float* __restrict__ dpooled, float* __restrict__ dconv, int d, int P) {
    const int t = blockIdx.x;
    const int c0 = conv_off[t], To = conv_off[t + 1] - c0;
    if (To <= 32) {
        // the channel's column in registers: To independent loads in one burst instead of a chain of dependent compares on loads (the
        // kernel was latency-bound: 176 us per 64-video step for 300 MB), the same first-maximum rule, one store per row
        for (int c = threadIdx.x; c < d; c += 256) {
            float v[32], g[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) { v[j] = j < To ? conv[(size_t)(c0 + j) * d + c] : 0.f; g[j] = 0.f; }
            for (int p = 0; p < P; ++p) {
                const int s = (p * To) / P, e = ((p + 1) * To + P - 1) / P;
                float m = 0.f;
                int am = s;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    if (j == s) m = v[j];
                    if (j > s && j < e && v[j] > m) { m = v[j]; am = j; }
                }
                const float up = dpooled[(size_t)t * d * P + (size_t)c * P + p];
#pragma unroll
                for (int j = 0; j < 32; ++j) g[j] += (j == am) ? up : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) if (j < To) dconv[(size_t)(c0 + j) * d + c] = g[j];
        }
        return;
    }
    for (int c = threadIdx.x; c < d; c += 256) {
        for (int j = 0; j < To; ++j) dconv[(size_t)(c0 + j) * d + c] = 0.f;
        for (int p = 0; p < P; ++p) {
            const int s = (p * To) / P, e = ((p + 1) * To + P - 1) / P;
            float m = conv[(size_t)(c0 + s) * d + c];
            int am = s;
            for (int j = s + 1; j < e; ++j) {
                const float v = conv[(size_t)(c0 + j) * d + c];
                if (v > m) { m = v; am = j; }
            }
            dconv[(size_t)(c0 + am) * d + c] += dpooled[(size_t)t * d * P + (size_t)c * P + p];
        }
    }
}

// G[s, k, :] = sum of dconv rows j whose tap k reads packed source row s (inverse of the conv gather table):
// stretched frame f (src(f) = i) feeds output j with tap c iff 2j - 1 + c == f.
__global__ __launch_bounds__(256) void conv_gather_bwd_kernel(const float* __restrict__ dconv, const int32_t* __restrict__ row_off,
                                                              const int32_t* __restrict__ conv_off, const int32_t* __restrict__ trk_vid,
                                                              const int32_t* __restrict__ vid_tmax, float* __restrict__ G, int d) {
    const int t = blockIdx.x;
    const int r0 = row_off[t], L = row_off[t + 1] - r0;
    const int c0 = conv_off[t], To = conv_off[t + 1] - c0;
    const int M = vid_tmax[trk_vid[t]];
    const int q = M / L, r = M - q * L;
    if ((d & 3) == 0) {
        // one (source frame, tap) pair per wave and pass, four channels per lane: the frame arithmetic is wave-uniform, the rows move as
        // 16-byte accesses (before: one float and two integer divisions per thread and element -- 546 us per 64-video step for the
        // 890 MB it writes); the sums run over the same frames in the same order
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d4 = d >> 2;
        for (int pair = wave; pair < L * 3; pair += 4) {
            const int i = pair / 3, k = pair - 3 * i;
            const int fa = i < r ? i * (q + 1) : r * (q + 1) + (i - r) * q;
            const int fb = fa + (i < r ? q + 1 : q);
            f32x4* out = (f32x4*)(G + ((size_t)(r0 + i) * 3 + k) * d);
            for (int c = lane; c < d4; c += 64) {
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
                for (int f = fa; f < fb; ++f) {
                    const int num = f + 1 - k;
                    if (num < 0 || (num & 1)) continue;
                    const int j = num >> 1;
                    if (j < To) s4 += ((const f32x4*)(dconv + (size_t)(c0 + j) * d))[c];
                }
                out[c] = s4;
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < L * 3 * d; e += 256) {
        const int ch = e % d, k = (e / d) % 3, i = e / (3 * d);
        // stretched frames of source frame i: [fa, fb)
        const int fa = i < r ? i * (q + 1) : r * (q + 1) + (i - r) * q;
        const int fb = fa + (i < r ? q + 1 : q);
        float s = 0.f;
        for (int f = fa; f < fb; ++f) {
            const int num = f + 1 - k;
            if (num < 0 || (num & 1)) continue;
            const int j = num >> 1;
            if (j < To) s += dconv[(size_t)(c0 + j) * d + ch];
        }
        G[((size_t)(r0 + i) * 3 + k) * d + ch] = s;
    }
}

// (n_out, n_in, taps) conv weight -> (n_in, taps*n_out): W'[c, k*n_out + o] = w[o, c, k]  (dX of the implicit GEMM)
__global__ void repack_conv_weight_bwd_kernel(const float* __restrict__ w, float* __restrict__ out, long total, int n_out, int n_in, int taps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long per_c = (long)taps * n_out;
    const long c = i / per_c;
    const int rem = (int)(i - c * per_c);
    const int k = rem / n_out, o = rem - k * n_out;
    out[i] = w[((long)o * n_in + c) * taps + k];
}

// head gather backward (models/model_0v10.py:485-503): d head_in (rows, ld) ->
//   dquery[row, :dq]; per tracklet sums of the subject/object slices of every segment table that is indexed by
//   tracklet; d bias_matrix[cs, co, :] += dlogits[row, :]
struct HeadBwdSegs { int n; float* dst[4]; int width[4]; int col[4]; };

__global__ __launch_bounds__(256) void head_gather_bwd_kernel(const float* __restrict__ dhead, int ld, const int32_t* __restrict__ soid,
                                                              const int32_t* __restrict__ e_off, const int32_t* __restrict__ trk_vid,
                                                              int nq, HeadBwdSegs segs) {
    const int t = blockIdx.x;                    // global tracklet
    const int v = trk_vid[t];
    const int lt = t - e_off[v];
    const int32_t* so = soid + (size_t)v * 2 * nq;
    // The (query, role) pairs that picked this tracklet, in the order the sums were always taken in (query ascending, subject before
    // object): about six of the 2 nq candidates.  One wave compacts them (ballot + prefix count keeps the order); before, every thread
    // walked all nq queries for every segment and column chunk -- 800 dependent index loads per workgroup, 540 us per 64-video step.
    __shared__ int hits[2048];
    __shared__ int n_hits;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int n = 0;
        for (int base = 0; base < 2 * nq; base += 64) {
            const int e = base + lane;                              // e = 2 query + role
            const bool hit = e < 2 * nq && so[(e & 1) * nq + (e >> 1)] == lt;
            const unsigned long long m = __ballot(hit);
            if (hit) hits[n + __popcll(m & ((1ull << lane) - 1ull))] = e;
            n += __popcll(m);
        }
        if (lane == 0) n_hits = n;
    }
    __syncthreads();
    const int nh = n_hits;
    for (int s = 0; s < segs.n; ++s) {
        const int w = segs.width[s];
        for (int c = threadIdx.x; c < w; c += 256) {
            float acc = 0.f;
            for (int i0 = 0; i0 < nh; i0 += 8) {              // eight rows in flight, added in list order
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = hits[min(i0 + u, nh - 1)];
                    x[u] = dhead[(size_t)(v * nq + (e >> 1)) * ld + segs.col[s] + (e & 1) * w + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += (i0 + u < nh) ? x[u] : 0.f;
            }
            segs.dst[s][(size_t)t * w + c] += acc;
        }
    }
}

// tap-major conv weight gradient (n_out, taps*n_in) -> torch layout (n_out, n_in, taps), accumulated
__global__ void conv_grad_unpack_kernel(const float* __restrict__ g, float* __restrict__ out, long total, int n_in, int taps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long per_out = (long)n_in * taps;
    const long o = i / per_out;
    const int rem = (int)(i - o * per_out);
    const int c = rem / taps, k = rem - c * taps;
    out[i] += g[o * per_out + (long)k * n_in + c];
}

// tap-by-output conv weight gradient (taps*n_out, n_in) -- row k*n_out + o, column c -- -> torch layout (n_out, n_in, taps), accumulated
__global__ void conv_grad_unpack_tapout_kernel(const float* __restrict__ g, float* __restrict__ out, long total, int n_out, int n_in, int taps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long per_out = (long)n_in * taps;
    const long o = i / per_out;
    const int rem = (int)(i - o * per_out);
    const int c = rem / taps, k = rem - c * taps;
    out[i] += g[((long)k * n_out + o) * n_in + c];
}

// out[c] += sum over the rows of x (rows, d): chunks of 256 rows per workgroup -> one partial row per workgroup, summed IN BLOCK ORDER by
// col_sum_finish_kernel (one float atomic per column and workgroup made the low bits of conv_feat2enti.bias.grad vary from run
// to run; every other gradient of this file has a fixed summation order)
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, float* __restrict__ partial, long rows, int d) {
    const long r0 = (long)blockIdx.x * 256, r1 = r0 + 256 < rows ? r0 + 256 : rows;
    for (int c = threadIdx.x; c < d; c += 256) {
        float s = 0.f;
        long r = r0;
        for (; r + 8 <= r1; r += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(size_t)(r + u) * d + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < r1; ++r) s += x[(size_t)r * d + c];
        partial[(size_t)blockIdx.x * d + c] = s;
    }
}

// out[c] += partial[0][c] + partial[1][c] + ... (one thread per column, eight rows in flight, the additions in block order)
__global__ __launch_bounds__(64) void col_sum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int blocks, int d) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= d) return;
    float s = 0.f;
    int b = 0;
    for (; b + 8 <= blocks; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b + u) * d + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < blocks; ++b) s += partial[(size_t)b * d + c];
    out[c] += s;
}

// out (cols, ld_out) = in (rows, cols)^T with the columns [rows, ld_out) zero-filled
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int ld_out) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;      // bx over cols of `in`, by over rows of `in` (padded to ld_out)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        tile[j][tx] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = bx + j, c = by + tx;
        if (r < cols && c < ld_out) out[(size_t)r * ld_out + c] = tile[tx][j];
    }
}

// d bias_matrix[cat_s, cat_o, :] += d logits[row, :] for every query row (models/model_0v10.py:489-497: the bias rows are gathered by the
// categories of the arg-max subject / object): a scatter-add with collisions.  No float atomics -- bias_matrix_key_kernel writes every
// row's destination, ordered_rows_add_kernel adds the rows that share one in ROW ORDER.
__global__ __launch_bounds__(256) void bias_matrix_key_kernel(const int32_t* __restrict__ soid, const int32_t* __restrict__ e_off,
                                                              const int32_t* __restrict__ trk_cat, int rows, int nq, int n_ecat, int32_t* __restrict__ key) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int v = row / nq, qi = row - v * nq;
    const int e0 = e_off[v];
    int k = -1;
    if (e_off[v + 1] - e0 > 0)
        k = trk_cat[e0 + soid[((size_t)v * 2) * nq + qi]] * n_ecat + trk_cat[e0 + soid[((size_t)v * 2 + 1) * nq + qi]];
    key[row] = k;
}

// dst[key[r] * ld_dst + c] += sum over the rows r' with key[r'] == key[r] of src[r' * ld_src + c], added in ascending r' (rows with a
// negative key are skipped).  One workgroup per row r: it leaves when an earlier row has the same key, else it is the key's first row and
// walks the rows from r on, 256 at a time, compacting the matching ones in order.  The ordered form of a scatter-add with collisions.
__global__ __launch_bounds__(256) void ordered_rows_add_kernel(const float* __restrict__ src, int ld_src, const int32_t* __restrict__ key, int key_stride,
                                                               int rows, int width, float* __restrict__ dst, int ld_dst) {
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int my = key[(size_t)r * key_stride];
    if (my < 0) return;
    int dup = 0;
    for (int j = tid; j < r; j += 256) dup |= key[(size_t)j * key_stride] == my;
    if (__syncthreads_or(dup)) return;
    __shared__ int cnt[4];
    __shared__ int list[256];
    float* out = dst + (size_t)my * ld_dst;
    for (int i0 = 0; i0 < width; i0 += 256) {
        const int col = i0 + tid;
        float acc = 0.f;
        for (int base = r; base < rows; base += 256) {
            const int j = base + tid;
            const bool m = j < rows && key[(size_t)j * key_stride] == my;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(m);
            if (lane == 0) cnt[wv] = __builtin_popcountll(mask);
            __syncthreads();
            int off = 0, total = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { if (u < wv) off += cnt[u]; total += cnt[u]; }
            if (m) list[off + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = j;
            __syncthreads();
            if (col < width)
                for (int t = 0; t < total; ++t) acc += src[(size_t)list[t] * ld_src + col];
            __syncthreads();
        }
        if (col < width) out[col] += acc;
    }
}

// copy the first dq columns of each row (d head_in -> d query)
__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, long rows, int cols4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * cols4; i += (long)gridDim.x * 256) {
        const long r = i / cols4;
        const int c = (int)(i - r * cols4);
        ((f32x4*)(dst + r * ldd))[c] = ((const f32x4*)(src + r * lds_))[c];
    }
}

// in place: x[r, c] = keep(r * n + c) ? x[r, c] / (1 - p) : 0  over the logical (rows, n) matrix with leading dimension ld
__global__ __launch_bounds__(256) void dropout_kernel(float* __restrict__ x, long rows, int n, int ld, const DropSpec drop) {
    const long total = rows * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / n;
        const int c = (int)(i - r * n);
        float* p = x + r * ld + c;
        *p = drop_keep(drop, (unsigned long long)i) ? *p * drop.scale : 0.f;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n4, float a) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = ((f32x4*)x)[i];
        v[0] *= a; v[1] *= a; v[2] *= a; v[3] *= a;
        ((f32x4*)x)[i] = v;
    }
}

int grid_for(long n, int cap = 4096) { long g = (n + 255) / 256; return (int)(g < cap ? (g < 1 ? 1 : g) : cap); }

}  // namespace

// ---- internal launchers (used by train.hip) + exported entry points for the parity tests --------------

extern "C" int bigc_transpose_f32(void* stream, const float* in, float* out, int rows, int cols) {
    BIGC_CHECK_ARG(in && out && rows > 0 && cols > 0, "transpose: bad argument");
    ProfScope ps(stream, FAM_BACKWARD, 0.0, 8.0 * rows * (double)cols);
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, in, out, rows, cols);
    BIGC_LAUNCH_CHECK("transpose");
    return BIGC_OK;
}

// Split-K scratch of the weight-gradient kernels: one buffer per (calling thread, device, slot), grown on demand after a sync of the
// calling stream.  (the buffers used to be function-static -- two threads driving two devices, which the per-handle
// arithmetic modes explicitly allow, would have shared one pointer and raced on the regrow.)
// slots: 0 / 1 weight-gradient partial tiles (single / grouped), 2 column sums, 3 LayerNorm and depth-wise conv parameter partials,
// 4 the keys of the bias_matrix gradient, 5 the deferred LayerNorm partials of a backward pass.  One buffer per (calling thread, device, slot, STREAM): two streams driven from one thread never
// share partials between a producer and its reduction.
static int dw_scratch(void* stream, size_t need_floats, int slot, float** out) {
    struct Buf { float* p = nullptr; size_t cap = 0; };
    thread_local std::map<std::pair<int, void*>, Buf> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "linear_dw: hipGetDevice failed");
    Buf& b = bufs[std::make_pair(dev * 8 + slot, stream)];
    if (need_floats > b.cap) {
        if (b.p) { if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "linear_dw: sync failed"); (void)hipFree(b.p); }
        b.cap = need_floats + need_floats / 2;
        if (hipMalloc((void**)&b.p, b.cap * sizeof(float)) != hipSuccess) { b.p = nullptr; b.cap = 0; return bigc_fail(BIGC_ERR_HIP, "linear_dw: hipMalloc of the split scratch failed"); }
    }
    *out = b.p;
    return BIGC_OK;
}

// for the other training files (grounding_train.hip): the same scratch, and the fixed-order sum of per-workgroup partial rows
int bigc_scratch_floats(hipStream_t s, size_t need_floats, int slot, float** out) { return dw_scratch((void*)s, need_floats, slot, out); }
int bigc_ordered_rows_add(hipStream_t s, const float* src, int ld_src, const int32_t* key, int key_stride, int rows, int width, float* dst, int ld_dst) {
    if (rows <= 0 || width <= 0) return BIGC_OK;
    hipLaunchKernelGGL(ordered_rows_add_kernel, dim3(rows), dim3(256), 0, s, src, ld_src, key, key_stride, rows, width, dst, ld_dst);
    BIGC_LAUNCH_CHECK("ordered_rows_add");
    return BIGC_OK;
}
// Deferred finishes: inside a FinishScope (the BIG-C backward) bigc_layernorm_bwd_f32 does not launch its finish but queues it -- each with its
// own slice of one scratch -- and bigc_finish_flush adds all queued partial-row sets in one launch (same sums, same order per set).
namespace {
struct FinishQueue { bool on = false; FinishJobs jobs; size_t used = 0; float* base = nullptr; size_t cap = 0; };
thread_local FinishQueue g_fq;
}
void bigc_finish_begin(void) { g_fq.on = true; g_fq.jobs.n = 0; g_fq.used = 0; }
int bigc_finish_flush(hipStream_t s) {
    FinishQueue& q = g_fq;
    q.on = false;
    if (q.jobs.n == 0) return BIGC_OK;
    int wmax = 0;
    for (int i = 0; i < q.jobs.n; ++i) wmax = q.jobs.job[i].w > wmax ? q.jobs.job[i].w : wmax;
    hipLaunchKernelGGL(partial_rows_finish_multi_kernel, dim3(cdiv(wmax, 64), q.jobs.n), dim3(1024), 0, s, q.jobs);
    q.jobs.n = 0;
    BIGC_LAUNCH_CHECK("partial_rows_finish_multi");
    return BIGC_OK;
}
// scratch slice for a deferred job (nullptr: not deferring, or the queue / the scratch is full -> the caller finishes at once)
static float* finish_slice(hipStream_t s, size_t floats) {
    FinishQueue& q = g_fq;
    if (!q.on || q.jobs.n >= FINISH_JOBS_MAX) return nullptr;
    if (!q.base || q.used + floats > q.cap) {
        if (q.jobs.n > 0 || q.used > 0) return nullptr;          // (never move a scratch that queued jobs point into)
        float* b = nullptr;
        const size_t want = floats * 40 > ((size_t)8 << 20) ? floats * 40 : ((size_t)8 << 20);
        if (dw_scratch((void*)s, want, 5, &b) != BIGC_OK) return nullptr;
        q.base = b; q.cap = want;
    }
    float* p = q.base + q.used;
    q.used += (floats + 63) & ~(size_t)63;
    return p;
}

int bigc_partial_rows_finish(hipStream_t s, const float* part, float* out, float* out2, int blocks, int w, int split) {
    hipLaunchKernelGGL(partial_rows_finish_kernel, dim3(cdiv(w, 64)), dim3(1024), 0, s, part, out, out2, blocks, w, split);
    BIGC_LAUNCH_CHECK("partial_rows_finish");
    return BIGC_OK;
}

static int dw_launch(void* stream, const float* dY, int ldy, const float* X, int ldx, const int32_t* x_idx, int nseg, int seg_len, float* dW, int ldw,
                     float* db, int M, int N, int K, const uint64_t* x_rowptr) {
    BIGC_CHECK_ARG(dY && (X || x_rowptr) && dW && M >= 0 && N > 0 && K > 0, "linear_dw: bad argument");
    if (M == 0) return BIGC_OK;
    if (x_idx) BIGC_CHECK_ARG(nseg >= 1 && nseg * seg_len == K, "linear_dw: K=%d != nseg*seg_len", K);
    if (x_rowptr) BIGC_CHECK_ARG(!x_idx && !bigc_linear_get_exact(), "linear_dw: the row-pointer operand is read by the split-operand kernel only, without a gather");
    DwArgs p;
    p.x_rowptr = x_rowptr;
    p.dY = dY; p.ldy = ldy; p.X = X; p.ldx = ldx; p.x_idx = x_idx; p.nseg = x_idx ? nseg : 1; p.seg_len = x_idx ? seg_len : K;
    p.dW = dW; p.ldw = ldw; p.db = db; p.M = M; p.N = N; p.K = K;
    p.tiles_n = cdiv(N, 128); p.tiles_k = cdiv(K, 128);
    const int tiles = p.tiles_n * p.tiles_k;
    // about two workgroups per CU; every split reduces at least 128 rows
    int splits = cdiv(512, tiles);
    const int max_splits = cdiv(M, 128);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.m_chunk = cdiv(cdiv(M, splits), 32) * 32;
    splits = cdiv(M, p.m_chunk);
    p.partial = nullptr; p.arrived = nullptr;
    const size_t n_pad = (size_t)p.tiles_n * 128, k_pad = (size_t)p.tiles_k * 128;
    if (splits > 1) {
        // stream-ordered scratch, grown on demand, one per (thread, device): see dw_scratch
        const size_t need = (size_t)splits * n_pad * k_pad + (db ? (size_t)splits * n_pad : 0);
        float* scratch = nullptr;
        if (int rc = dw_scratch(stream, need, 0, &scratch)) return rc;
        p.partial = scratch;
        if (db) p.db_part = scratch + (size_t)splits * n_pad * k_pad;
        // arrival counters of the in-kernel reduction (zero between launches); one set per (calling thread, device) like the scratch
        // (a function-static pointer was shared by threads driving different devices)
        thread_local std::unordered_map<int, int*> arrived_of;
        int dev_now = 0;
        if (hipGetDevice(&dev_now) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "linear_dw: hipGetDevice failed");
        int*& arrived = arrived_of[dev_now];
        // (measured, 8-video training step: the in-kernel form costs 99 us per launch against 19 + 6 us for the two launches --
        // every workgroup pays a device-scope release, an L2 write-back on this chip -- so it is the opt-in, BIGC_DW_REDUCE=inline)
        static const bool two_launches = !(getenv("BIGC_DW_REDUCE") && !strcmp(getenv("BIGC_DW_REDUCE"), "inline"));
        constexpr int ARRIVED_CAP = 16384;
        if (!arrived && !two_launches) {
            if (hipMalloc((void**)&arrived, ARRIVED_CAP * sizeof(int)) != hipSuccess || hipMemsetAsync(arrived, 0, ARRIVED_CAP * sizeof(int), (hipStream_t)stream) != hipSuccess) {
                arrived = nullptr;
                return bigc_fail(BIGC_ERR_HIP, "linear_dw: hipMalloc of the arrival counters failed");
            }
        }
        p.arrived = (!two_launches && tiles <= ARRIVED_CAP) ? arrived : nullptr;
    }
    p.splits = splits;
    {
        ProfScope ps(stream, FAM_LINEAR_DW, 2.0 * M * (double)N * K, 4.0 * ((double)M * N + (double)M * K + (double)N * K));
        if (bigc_linear_get_exact()) hipLaunchKernelGGL(linear_dw_kernel, dim3(tiles * splits), dim3(256), 0, (hipStream_t)stream, p);
        else if (x_idx) hipLaunchKernelGGL(linear_dw_split_kernel<true>, dim3(tiles * splits), dim3(256), 0, (hipStream_t)stream, p);
        else if (x_rowptr) hipLaunchKernelGGL((linear_dw_split_kernel<false, true>), dim3(tiles * splits), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(linear_dw_split_kernel<false>, dim3(tiles * splits), dim3(256), 0, (hipStream_t)stream, p);
        BIGC_LAUNCH_CHECK("linear_dw");
        if (splits > 1 && !p.arrived) {
            hipLaunchKernelGGL(dw_reduce_kernel, dim3(grid_for((long)N * K)), dim3(256), 0, (hipStream_t)stream, p.partial, dW, ldw, N, K, (int)n_pad,
                               (int)k_pad, splits, (const float*)p.db_part, db);
            BIGC_LAUNCH_CHECK("dw_reduce");
        }
    }
    return BIGC_OK;
}

extern "C" int bigc_linear_dw_f32(void* stream, const float* dY, int ldy, const float* X, int ldx, const int32_t* x_idx, int nseg,
                                  int seg_len, float* dW, int ldw, float* db, int M, int N, int K) {
    return dw_launch(stream, dY, ldy, X, ldx, x_idx, nseg, seg_len, dW, ldw, db, M, N, K, nullptr);
}

// the same with the rows of X given by address (row m at x_rowptr[m], 16-byte aligned, K floats): the weight gradient of the first
// visual-feature layer read in place from the callers' per-tracklet tensors (train.hip)
int bigc_linear_dw_rows(void* stream, const float* dY, int ldy, const uint64_t* x_rowptr, float* dW, int ldw, float* db, int M, int N, int K) {
    return dw_launch(stream, dY, ldy, nullptr, 0, nullptr, 1, K, dW, ldw, db, M, N, K, x_rowptr);
}

// Grouped form of bigc_linear_dw_f32 (no gather): jobs[i] = {dY, ldy, X, ldx, dW, ldw, db, M, N, K}.  The operands must stay
// untouched until the launch has run (the caller -- train.hip's decoder layer -- hands over buffers nothing writes to before its
// flush).  Exact mode and single products go through bigc_linear_dw_f32.
int bigc_linear_dw_group(void* stream, const BigcDwJob* jobs, int n) {
    if (n <= 0) return BIGC_OK;
    if (n == 1 || bigc_linear_get_exact()) {
        for (int i = 0; i < n; ++i) {
            const BigcDwJob& q = jobs[i];
            int rc = bigc_linear_dw_f32(stream, q.dY, q.ldy, q.X, q.ldx, nullptr, 1, q.K, q.dW, q.ldw, q.db, q.M, q.N, q.K);
            if (rc != BIGC_OK) return rc;
        }
        return BIGC_OK;
    }
    for (int base = 0; base < n; base += DW_GROUP_MAX) {
        const int cnt = n - base < DW_GROUP_MAX ? n - base : DW_GROUP_MAX;
        DwGroup g, r;
        g.n = r.n = cnt;
        long tiles_total = 0;
        for (int i = 0; i < cnt; ++i) {
            const BigcDwJob& q = jobs[base + i];
            BIGC_CHECK_ARG(q.dY && q.X && q.dW && q.M > 0 && q.N > 0 && q.K > 0, "linear_dw_group: bad job %d", base + i);
            tiles_total += (long)cdiv(q.N, 128) * cdiv(q.K, 128);
        }
        // about two workgroups per CU over the whole group; every split reduces at least 128 rows
        int want = (int)cdiv(512, (int)tiles_total);
        if (want < 1) want = 1;
        size_t need = 0;
        int blocks = 0, rblocks = 0;
        double flops = 0, bytes = 0;
        for (int i = 0; i < cnt; ++i) {
            const BigcDwJob& q = jobs[base + i];
            DwArgs& p = g.job[i];
            p.dY = q.dY; p.ldy = q.ldy; p.X = q.X; p.ldx = q.ldx; p.x_idx = nullptr; p.nseg = 1; p.seg_len = q.K;
            p.dW = q.dW; p.ldw = q.ldw; p.db = q.db; p.M = q.M; p.N = q.N; p.K = q.K;
            p.tiles_n = cdiv(q.N, 128); p.tiles_k = cdiv(q.K, 128);
            int splits = want;
            const int max_splits = cdiv(q.M, 128);
            if (splits > max_splits) splits = max_splits;
            p.m_chunk = cdiv(cdiv(q.M, splits), 32) * 32;
            splits = cdiv(q.M, p.m_chunk);
            p.splits = splits; p.arrived = nullptr; p.partial = nullptr;
            g.start[i] = blocks;
            blocks += p.tiles_n * p.tiles_k * splits;
            if (splits > 1) need += (size_t)splits * p.tiles_n * 128 * p.tiles_k * 128 + (q.db ? (size_t)splits * p.tiles_n * 128 : 0);
            flops += 2.0 * q.M * (double)q.N * q.K;
            bytes += 4.0 * ((double)q.M * q.N + (double)q.M * q.K + (double)q.N * q.K);
        }
        g.start[cnt] = blocks;
        if (need) {
            float* scratch = nullptr;             // stream-ordered, grown on demand, one per (thread, device)
            if (int rc = dw_scratch(stream, need, 1, &scratch)) return rc;
            size_t off = 0;
            for (int i = 0; i < cnt; ++i) {
                DwArgs& p = g.job[i];
                if (p.splits > 1) {
                    p.partial = scratch + off; off += (size_t)p.splits * p.tiles_n * 128 * p.tiles_k * 128;
                    if (p.db) { p.db_part = scratch + off; off += (size_t)p.splits * p.tiles_n * 128; }
                }
            }
        }
        r = g;
        for (int i = 0; i < cnt; ++i) {          // reduction launch: one workgroup per 2 K output elements of a job (two 16-byte quads per thread)
            r.start[i] = rblocks;
            if (g.job[i].partial) { const long e = (long)g.job[i].N * g.job[i].K; rblocks += (int)((e + 2047) / 2048); }
        }
        r.start[cnt] = rblocks;
        ProfScope ps(stream, FAM_LINEAR_DW, flops, bytes);
        hipLaunchKernelGGL(linear_dw_group_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
        BIGC_LAUNCH_CHECK("linear_dw_group");
        if (rblocks > 0) {
            hipLaunchKernelGGL(dw_reduce_group_kernel, dim3(rblocks), dim3(256), 0, (hipStream_t)stream, r);
            BIGC_LAUNCH_CHECK("dw_reduce_group");
        }
    }
    return BIGC_OK;
}

extern "C" int bigc_dropout_f32(void* stream, float* x, long rows, int n, int ld, float p, uint32_t seed, uint32_t site) {
    BIGC_CHECK_ARG(x && n > 0 && ld >= n && p >= 0.f && p < 1.f, "dropout: bad argument");
    if (rows <= 0 || !(p > 0.f)) return BIGC_OK;
    ProfScope ps(stream, FAM_BACKWARD, 0.0, 8.0 * rows * (double)n);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(rows * n)), dim3(256), 0, (hipStream_t)stream, x, rows, n, ld, make_drop(p, seed, site));
    BIGC_LAUNCH_CHECK("dropout");
    return BIGC_OK;
}

int bigc_scale_f32(hipStream_t s, float* x, long n, float a) {
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, x, n / 4, a);
    BIGC_LAUNCH_CHECK("scale");
    return BIGC_OK;
}

extern "C" int bigc_relu_bwd_f32(void* stream, float* dY, const float* Y, long n) {
    BIGC_CHECK_ARG(dY && Y && n % 4 == 0, "relu_bwd: bad argument");
    if (n <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_BACKWARD, 0.0, 12.0 * n);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, dY, Y, n / 4);
    BIGC_LAUNCH_CHECK("relu_bwd");
    return BIGC_OK;
}

extern "C" int bigc_axpy_f32(void* stream, float* y, const float* x, long n) {
    BIGC_CHECK_ARG(y && x && n % 4 == 0, "axpy: bad argument");
    if (n <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_BACKWARD, 0.0, 12.0 * n);
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, y, x, n / 4);
    BIGC_LAUNCH_CHECK("axpy");
    return BIGC_OK;
}

extern "C" int bigc_layernorm_bwd_f32(void* stream, const float* x, const float* gamma, const float* dy, float* dx, float* dgamma,
                                      float* dbeta, int rows, int d, float eps) {
    BIGC_CHECK_ARG(x && gamma && dy && dx && dgamma && dbeta, "layernorm_bwd: null pointer");
    BIGC_CHECK_ARG(d > 0 && d % 4 == 0 && d <= 2048, "layernorm_bwd: d=%d", d);
    if (rows <= 0) return BIGC_OK;
    // every wave takes >= 8 rows (its parameter-gradient partials live in registers): few workgroups, few atomics
    const int grid = cdiv(rows, 32) < 512 ? cdiv(rows, 32) : 512;
    float* part = finish_slice((hipStream_t)stream, (size_t)grid * 2 * d);      // deferred finish (FinishScope): own slice, one launch for all at the flush
    const bool deferred = part != nullptr;
    if (!deferred)                          // (grid, 2, d) partial rows: stream-ordered scratch, see dw_scratch
        if (int rc = dw_scratch(stream, (size_t)grid * 2 * d, 3, &part)) return rc;
    ProfScope ps(stream, FAM_BACKWARD, 0.0, 12.0 * rows * (double)d);
#define LN_BWD(NVL) hipLaunchKernelGGL(layernorm_bwd_kernel<NVL>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, gamma, dy, dx, part, rows, d, eps)
    if (d <= 256) LN_BWD(1);
    else if (d <= 512) LN_BWD(2);
    else if (d <= 1024) LN_BWD(4);
    else LN_BWD(8);
#undef LN_BWD
    if (deferred) {
        FinishJob& j = g_fq.jobs.job[g_fq.jobs.n++];
        j.part = part; j.out = dgamma; j.out2 = dbeta; j.blocks = grid; j.w = 2 * d; j.split = d;
    } else {
        hipLaunchKernelGGL(partial_rows_finish_kernel, dim3(cdiv(2 * d, 64)), dim3(1024), 0, (hipStream_t)stream, (const float*)part, dgamma, dbeta, grid, 2 * d, d);
    }
    BIGC_LAUNCH_CHECK("layernorm_bwd");
    return BIGC_OK;
}

extern "C" int bigc_reduce_periodic_f32(void* stream, const float* x, float* out, int rows, int d, int period) {
    BIGC_CHECK_ARG(x && out && period > 0, "reduce_periodic: bad argument");
    if (rows <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_BACKWARD, 0.0, 4.0 * rows * (double)d);
    hipLaunchKernelGGL(reduce_periodic_kernel, dim3(grid_for((long)period * d)), dim3(256), 0, (hipStream_t)stream, x, out, rows, d, period);
    BIGC_LAUNCH_CHECK("reduce_periodic");
    return BIGC_OK;
}

extern "C" int bigc_maxpool_bwd_f32(void* stream, const float* conv, const int32_t* trk_conv_off, const float* dpooled, float* dconv,
                                    int n_trk, int d, int P) {
    BIGC_CHECK_ARG(conv && trk_conv_off && dpooled && dconv, "maxpool_bwd: null pointer");
    if (n_trk <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_BACKWARD, 0.0, 0.0);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(n_trk), dim3(256), 0, (hipStream_t)stream, conv, trk_conv_off, dpooled, dconv, d, P);
    BIGC_LAUNCH_CHECK("maxpool_bwd");
    return BIGC_OK;
}

extern "C" int bigc_conv_gather_bwd_f32(void* stream, const float* dconv, const int32_t* trk_row_off, const int32_t* trk_conv_off,
                                        const int32_t* trk_vid, const int32_t* vid_tmax, float* G, int n_trk, int d) {
    BIGC_CHECK_ARG(dconv && trk_row_off && trk_conv_off && trk_vid && vid_tmax && G, "conv_gather_bwd: null pointer");
    if (n_trk <= 0) return BIGC_OK;
    ProfScope ps(stream, FAM_BACKWARD, 0.0, 0.0);
    hipLaunchKernelGGL(conv_gather_bwd_kernel, dim3(n_trk), dim3(256), 0, (hipStream_t)stream, dconv, trk_row_off, trk_conv_off, trk_vid, vid_tmax, G, d);
    BIGC_LAUNCH_CHECK("conv_gather_bwd");
    return BIGC_OK;
}

int bigc_repack_conv_weight_bwd(hipStream_t s, const float* w, float* out, int n_out, int n_in, int taps) {
    const long total = (long)n_out * n_in * taps;
    hipLaunchKernelGGL(repack_conv_weight_bwd_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, w, out, total, n_out, n_in, taps);
    BIGC_LAUNCH_CHECK("repack_conv_weight_bwd");
    return BIGC_OK;
}

int bigc_head_gather_bwd(hipStream_t s, const float* dhead, int ld, const int32_t* soid, const int32_t* e_off, const int32_t* trk_vid,
                         int n_trk, int nq, int n_seg, float* const* dst, const int* width, const int* col) {
    HeadBwdSegs segs;
    segs.n = n_seg;
    for (int i = 0; i < 4; ++i) { segs.dst[i] = i < n_seg ? dst[i] : nullptr; segs.width[i] = i < n_seg ? width[i] : 0; segs.col[i] = i < n_seg ? col[i] : 0; }
    if (n_trk <= 0 || n_seg <= 0) return BIGC_OK;
    BIGC_CHECK_ARG(nq <= 1024, "head_gather_bwd: nq=%d > 1024", nq);
    hipLaunchKernelGGL(head_gather_bwd_kernel, dim3(n_trk), dim3(256), 0, s, dhead, ld, soid, e_off, trk_vid, nq, segs);
    BIGC_LAUNCH_CHECK("head_gather_bwd");
    return BIGC_OK;
}

int bigc_bias_matrix_bwd(hipStream_t s, const float* dlogits, int ldg, const int32_t* soid, const int32_t* e_off, const int32_t* trk_cat,
                         int n_vid, int nq, int n_ecat, int n_pcat, float* dbias) {
    if (n_vid <= 0) return BIGC_OK;
    const int rows = n_vid * nq;
    float* key = nullptr;                   // one int32 per row (stream-ordered scratch)
    if (int rc = dw_scratch((void*)s, (size_t)rows, 4, &key)) return rc;
    hipLaunchKernelGGL(bias_matrix_key_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, soid, e_off, trk_cat, rows, nq, n_ecat, (int32_t*)key);
    hipLaunchKernelGGL(ordered_rows_add_kernel, dim3(rows), dim3(256), 0, s, dlogits, ldg, (const int32_t*)key, 1, rows, n_pcat, dbias, n_pcat);
    BIGC_LAUNCH_CHECK("bias_matrix_bwd");
    return BIGC_OK;
}

int bigc_conv_grad_unpack(hipStream_t s, const float* g_tapmajor, float* g_torch, int n_out, int n_in, int taps) {
    const long total = (long)n_out * n_in * taps;
    hipLaunchKernelGGL(conv_grad_unpack_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, g_tapmajor, g_torch, total, n_in, taps);
    BIGC_LAUNCH_CHECK("conv_grad_unpack");
    return BIGC_OK;
}

int bigc_conv_grad_unpack_tapout(hipStream_t s, const float* g_tapout, float* g_torch, int n_out, int n_in, int taps) {
    const long total = (long)n_out * n_in * taps;
    hipLaunchKernelGGL(conv_grad_unpack_tapout_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, g_tapout, g_torch, total, n_out, n_in, taps);
    BIGC_LAUNCH_CHECK("conv_grad_unpack_tapout");
    return BIGC_OK;
}

int bigc_col_sum(hipStream_t s, const float* x, float* out, long rows, int d) {
    if (rows <= 0) return BIGC_OK;
    const int blocks = (int)((rows + 255) / 256);
    float* partial = nullptr;
    if (int rc = dw_scratch((void*)s, (size_t)blocks * d, 2, &partial)) return rc;
    hipLaunchKernelGGL(col_sum_kernel, dim3(blocks), dim3(256), 0, s, x, partial, rows, d);
    hipLaunchKernelGGL(col_sum_finish_kernel, dim3(cdiv(d, 64)), dim3(64), 0, s, partial, out, blocks, d);
    BIGC_LAUNCH_CHECK("col_sum");
    return BIGC_OK;
}

int bigc_transpose_pad(hipStream_t s, const float* in, float* out, int rows, int cols, int ld_out) {
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(cdiv(cols, 32), cdiv(ld_out, 32)), dim3(256), 0, s, in, out, rows, cols, ld_out);
    BIGC_LAUNCH_CHECK("transpose_pad");
    return BIGC_OK;
}

int bigc_copy_cols(hipStream_t s, const float* src, int lds_, float* dst, int ldd, long rows, int cols) {
    if (rows <= 0) return BIGC_OK;
    hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for(rows * (cols / 4))), dim3(256), 0, s, src, lds_, dst, ldd, rows, cols / 4);
    BIGC_LAUNCH_CHECK("copy_cols");
    return BIGC_OK;
}
