// Split-operand "linear" kernel: the same C = act(A' W^T + bias) + residual as linear.hip, with f32 accuracy, on the
// f16 matrix cores (v_mfma_f32_32x32x16_f16, 16x the rate of the f32 MFMA).
//
// Every f32 operand x is split into two f16 numbers, x = h + r * 2^-11 with h = f16(x) (11 bits, round to nearest) and
// r = f16((x - float(h)) * 2^11) (the next 11 bits, signed; x - float(h) is exact in f32): |x - h - r 2^-11| <= 2^-24 |x|.  Then
//     a * w = h_a h_w + 2^-11 (h_a r_w + r_a h_w) + 2^-22 r_a r_w
// and the last term (<= 2^-22 relative, f32 round-off level) is dropped: three f16 MFMAs per product, accumulated in
// f32 in TWO accumulators (leading term / correction terms, combined as acc0 + 2^-11 acc1 in the epilogue, so the small
// terms are not absorbed by the large partial sum).  Products of f16 pairs are exact in f32.
// Range: the staging code tracks max |x| and a NaN-propagating sum of what it splits; when an operand of a tile is
// >= 32768 in magnitude, NaN or Inf (or the cached split of W says so), the workgroup recomputes its tile at the end on the
// exact f32 MFMA straight from memory.  Tiny operands (|x| < 2^-35) contribute 0, an absolute error below 3e-11 per element.
//
// Tiling: 128x256 output tile per 512-thread workgroup (8 waves as 2x4; 128x128 with 4 waves for narrow outputs), wave patch
// 64x64 = 2x2 MFMA blocks x 2 accumulators (128 accumulator registers), K slabs of 32, two waves per SIMD.  LDS holds four f16 planes per slab
// (A head, A residual, W head, W residual), rows of 64 bytes with the 16-byte chunks XOR-swizzled by (row >> 2) & 3:
// both the ds_read_b128 fragment reads (lane = row, 8 consecutive k) and the ds_write_b64 stores are conflict-free.
// W is split once per launch by a small pre-pass into [N][K/4] 16-byte groups {4 heads, 4 residuals} (same footprint
// as f32), A is split by the staging code on its way from HBM to LDS.
// Tried and rejected: persistent workgroups walking the tile list (275.4 vs 273.8 TFLOP/s on the step's shapes, +1 spill);
// the same with the slab pipeline running across tile boundaries (next tile's first slabs fetched during the epilogue):
// correct, but the prefetch registers stay live across the epilogue and the kernel spills 200-300 VGPRs (10x slower).
// Roofline: f16 MFMA dense peak / 3 MFMAs per product.
#include "linear_internal.h"
#include "prof.h"
#include <atomic>
#include <stdlib.h>
#include <mutex>
#include <algorithm>
#include <map>
#include <unordered_map>
#include <utility>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BK = 32;
constexpr float RSCALE = 2048.0f, RINV = 1.0f / 2048.0f;
constexpr float F16_LIMIT = 32768.0f;      // operands at or beyond this magnitude take the exact f32 kernel

union Pack4 { fp16x2 h[2]; uint2 u; };

// four consecutive f32 -> 4 heads + 4 scaled residuals (8 bytes each).  `chk` accumulates the residuals (NaN when an
// element is NaN/Inf), `amax` the largest magnitude (the range is checked on the input).  Both halves are rounded to nearest
// (common.h: bigc_pk_rtn), so the residual is signed.
__device__ __forceinline__ void split4(const f32x4 x, uint2& head, uint2& resid, float& chk, float& amax) {
    Pack4 h, r;
    h.h[0] = bigc_pk_rtn(x[0], x[1]);
    h.h[1] = bigc_pk_rtn(x[2], x[3]);
    const float r0 = (x[0] - (float)h.h[0][0]) * RSCALE, r1 = (x[1] - (float)h.h[0][1]) * RSCALE;
    const float r2 = (x[2] - (float)h.h[1][0]) * RSCALE, r3 = (x[3] - (float)h.h[1][1]) * RSCALE;
    r.h[0] = bigc_pk_rtn(r0, r1);
    r.h[1] = bigc_pk_rtn(r2, r3);
    chk += (r0 + r1) + (r2 + r3);
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(amax) : "v"(amax), "v"(x[0]), "v"(x[1]));
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(amax) : "v"(amax), "v"(x[2]), "v"(x[3]));
    head = h.u;
    resid = r.u;
}

// W (N, K) f32 -> Ws (N, K/4) groups of 16 bytes
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ W, int ldw, u32x4* __restrict__ Ws, int N, int K4,
                                                           int* bad, int bad_id) {
    float chk = 0.f, amax = 0.f;
    const long total = (long)N * K4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / K4), k4 = (int)(i - (long)n * K4);
        const f32x4 x = *(const f32x4*)(W + (size_t)n * ldw + 4 * k4);
        uint2 h, r;
        split4(x, h, r, chk, amax);
        Ws[i] = u32x4{h.x, h.y, r.x, r.y};
    }
    if (!(fabsf(chk) < INFINITY) || amax >= F16_LIMIT) *bad = bad_id;       // ids are unique per split: no clearing needed
}

// Batched weight preparation (training: every weight changes every step, and the backward needs W^T as well): one launch
// splits up to PREP_MAX weights.  A direct job does what split_weight_kernel does; a transposed job reads W (N, K), and writes
// W^T padded to n4 = round4(N) columns both as f32 (K, n4) -- the operand of the exact kernel and of the range fallback -- and
// split (K, n4/4 groups).
constexpr int PREP_MAX = 48;
struct PrepJob { const float* W; u32x4* Ws; float* wT; int* bad; int N, K, ldw, bad_id, blk0, nblk, transposed, pad; };
struct PrepArgs { int n; int pad; PrepJob j[PREP_MAX]; };

__global__ __launch_bounds__(256) void weight_prep_kernel(const PrepArgs a) {
    __shared__ float tile[64][65];
    int ji = 0;
    while (ji + 1 < a.n && (int)blockIdx.x >= a.j[ji + 1].blk0) ++ji;
    const PrepJob& J = a.j[ji];
    const int lb = (int)blockIdx.x - J.blk0, tid = threadIdx.x;
    float chk = 0.f, amax = 0.f;
    if (!J.transposed) {
        const int K4 = (J.K + 3) / 4;
        const long total = (long)J.N * K4;
        for (long i = (long)lb * 256 + tid; i < total; i += (long)J.nblk * 256) {
            const int n = (int)(i / K4), k4 = (int)(i - (long)n * K4);
            const f32x4 x = *(const f32x4*)(J.W + (size_t)n * J.ldw + 4 * k4);
            uint2 h, r;
            split4(x, h, r, chk, amax);
            J.Ws[i] = u32x4{h.x, h.y, r.x, r.y};
        }
    } else {
        // output rows n' = 0..K-1 (columns of W), output columns k' = 0..n4-1 (rows of W, zero beyond N)
        const int n4 = (J.N + 3) / 4 * 4, K4o = n4 / 4;
        const int tr = (J.K + 63) / 64, tc = (n4 + 63) / 64;
        for (int t = lb; t < tr * tc; t += J.nblk) {
            const int n0 = (t / tc) * 64, k0 = (t % tc) * 64;
            __syncthreads();
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int idx = tid + 256 * i, kk = idx >> 6, nn = idx & 63;
                const int row = k0 + kk, col = n0 + nn;
                tile[kk][nn] = (row < J.N && col < J.K) ? J.W[(size_t)row * J.ldw + col] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int item = tid + 256 * i, k4l = item & 15, nn = item >> 4;
                const int orow = n0 + nn, ok4 = k0 / 4 + k4l;
                if (orow < J.K && ok4 < K4o) {
                    const f32x4 x = {tile[4 * k4l][nn], tile[4 * k4l + 1][nn], tile[4 * k4l + 2][nn], tile[4 * k4l + 3][nn]};
                    uint2 h, r;
                    split4(x, h, r, chk, amax);
                    J.Ws[(size_t)orow * K4o + ok4] = u32x4{h.x, h.y, r.x, r.y};
                    *(f32x4*)(J.wT + (size_t)orow * n4 + 4 * ok4) = x;
                }
            }
        }
    }
    if (!(fabsf(chk) < INFINITY) || amax >= F16_LIMIT) *J.bad = J.bad_id;
}

// two f32 (already scaled) -> packed (head, head) and (residual, residual) of the "hr8" form, round to nearest
__device__ __forceinline__ void split_rtn2(float x0, float x1, unsigned& h, unsigned& r) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {x0, x1};
    const h2_t hh = __builtin_convertvector(v, h2_t);
    const f2_t d = v - __builtin_convertvector(hh, f2_t);
    const h2_t rr = __builtin_convertvector(d, h2_t);
    h = *(const unsigned*)&hh;
    r = *(const unsigned*)&rr;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {          // v_cvt_pk_bf16_f32: round to nearest even
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    return *(const unsigned*)&h;
}

__device__ __forceinline__ int swz(int chunk, int row) { return chunk ^ ((row >> 2) & 3); }

// WN = wave columns: 2 -> 128x128 tile, 256 threads, two workgroups per CU; 4 -> 128x256 tile, 512 threads (one workgroup
// per CU, same two waves per SIMD): the A slab is split once for eight waves instead of four.
// DWK > 0: depth-wise-separable mode (LinearArgs::dw_*), DWK = kernel size of the depth-wise convolution.
// PRE8: preceding-layer mode (LinearArgs::pre_*).
// FAST: the opt-in reduced-precision modes (bigc_linear_set_fast / BIGC_FAST, linear.hip): 1 = operands rounded to ONE f16 each,
// 2 = to one bf16 each (the literal "bf16" run of BASELINE configs[1]: 8-bit mantissa, v_mfma_f32_32x32x16_bf16; the weight's bf16
// is rounded from its cached split form h + r 2^-11, which holds 22 bits of it); one MFMA per product -- the residual planes are
// neither written nor read and the correction accumulator stays zero.  The range checks are those of the split (|x| < 32768).
template <bool GATHER, bool KTAIL, int WN, int DWK = 0, bool PRE8 = false, int FAST = 0>
__global__ __launch_bounds__(128 * WN, (WN == 2 && DWK == 0) ? 2 : 1) void linear_split_kernel(const LinearArgs p) {
    constexpr int NTH = 128 * WN, BNT = 64 * WN;     // threads, tile width
    constexpr int RPP = NTH / 8;                     // rows staged per pass (8 threads x 16 bytes per 32-k row)
    constexpr int NA = BM / RPP, NB = BNT / RPP;     // passes over the A / W rows
    constexpr int PA = BM * 64, PW = BNT * 64;       // bytes of an A / W plane
    // [buffer][A head | A resid | W head | W resid], rows of 64 bytes
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2 * PA + 2 * PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, hi = lane >> 5, l31 = lane & 31;

    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int tm = (j / p.tiles_n) * 8 + xcd, tn = j % p.tiles_n;
    if (tm >= p.tiles_m) return;
    const int m0 = tm * BM, n0 = tn * BNT;

    const int lr = tid >> 3, q = tid & 7, lc = q * 4;
    // Loads are unconditional (one basic block per K step, so that the scheduler can interleave the split arithmetic
    // with the MFMAs): rows past M / N are clamped to the last valid row -- their products land in accumulator rows /
    // columns that are never stored -- and only a ragged last slab (KTAIL) or a gather index of -1 (a zero row) select 0.
    // (global-address-space pointers: a row address read from the row-pointer table would otherwise be a FLAT pointer, and flat loads
    // count on the LDS counter too -- measured 4.42 instead of 3.94 ms on the K = 2048 GEMM)
    typedef const __attribute__((address_space(1))) float* gfp;
    typedef const __attribute__((address_space(1))) f32x4* gf4p;
    gfp arow[NA];
    const u32x4* wrow[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = min(m0 + lr + RPP * i, p.M - 1);
        arow[i] = p.a_rowptr ? (gfp)p.a_rowptr[row] : (gfp)(p.A + (size_t)row * p.lda);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) wrow[i] = p.Ws + (size_t)min(n0 + lr + RPP * i, p.N - 1) * p.ldws;
    // depth-wise-separable mode: per staged row, the element offset of every tap's row relative to the row itself, and which
    // taps fall inside the row's sequence (the others read the row itself and are dropped)
    constexpr int DWN = DWK > 0 ? DWK : 1;
    int dwoff[NA][DWN];
    unsigned dwmask[NA];
    if (DWK > 0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = min(m0 + lr + RPP * i, p.M - 1);
            int t, T;
            if (p.dw_info) { const int2 v = p.dw_info[row]; t = v.x; T = v.y; } else { t = row % p.dw_T; T = p.dw_T; }
            dwmask[i] = 0;
#pragma unroll
            for (int jj = 0; jj < DWN; ++jj) {
                const int tt = t + jj - DWK / 2;
                const bool ok = tt >= 0 && tt < T;
                dwmask[i] |= ok ? (1u << jj) : 0u;
                dwoff[i][jj] = ok ? (jj - DWK / 2) * p.lda : 0;
            }
        }
    }
    // A is prefetched two slabs ahead in two register sets (its rows stream from HBM), W one slab ahead (L2-resident)
    f32x4 ra0[NA], ra1[NA];
    u32x4 rb[NB];
    int cur_seg = -1;
    float chk = 0.f, amax = 0.f;
    // split-K: this workgroup owns the slabs [kofs, kofs + nk) of the nk_tot slabs (p.splits == 1: all of them)
    const int nk_tot = (p.K + BK - 1) / BK;
    const int nkp = (nk_tot + p.splits - 1) / p.splits;
    const int kofs = (int)blockIdx.y * nkp;
    const int nk = min(nk_tot, kofs + nkp) - kofs;

    auto slab_cols = [&](int kt, int& kc, int& acol, bool& kval) {
        const int k0 = kt * BK;
        kc = k0 + lc;
        kval = true;
        if (KTAIL) { kval = kc < p.K; kc = min(kc, p.K - 4); }
        acol = kc;
        if (GATHER) {
            const int seg = k0 / p.seg_len;
            acol = kc - seg * p.seg_len;
            if (seg != cur_seg) {
                cur_seg = seg;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int src = p.a_idx[(size_t)min(m0 + lr + RPP * i, p.M - 1) * p.nseg + seg];
                    arow[i] = (gfp)(src >= 0 ? p.A + (size_t)src * p.lda : p.zero_row);      // index -1: a row of zeros, no select needed
                }
            }
        }
    };
    auto load_a = [&](int kt, f32x4 (&ra)[NA]) {
        int kc, acol; bool kval;
        slab_cols(kt, kc, acol, kval);
        if (PRE8) {
            // a[row][k] = relu(b0[k] + sum_j x8[row][j] W0[k][j]), the operation order of linear_k8_kernel (linear.hip)
            f32x4 w0[4], w1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* wp = p.pre_w + (size_t)(kc + c) * 8;
                w0[c] = *(const f32x4*)wp;
                w1[c] = *(const f32x4*)(wp + 4);
            }
            const f32x4 bq = *(const f32x4*)(p.pre_b + kc);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const float* xp = p.pre_x + (size_t)min(m0 + lr + RPP * i, p.M - 1) * 8;
                const f32x4 a0 = *(const f32x4*)xp, a1 = *(const f32x4*)(xp + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = a0[0] * w0[c][0];
                    v = fmaf(a0[1], w0[c][1], v); v = fmaf(a0[2], w0[c][2], v); v = fmaf(a0[3], w0[c][3], v);
                    v = fmaf(a1[0], w1[c][0], v); v = fmaf(a1[1], w1[c][1], v); v = fmaf(a1[2], w1[c][2], v); v = fmaf(a1[3], w1[c][3], v);
                    v += bq[c];
                    ra[i][c] = fmaxf(v, 0.f);
                }
            }
            return;
        }
        if (DWK > 0) {
            // a[row][k] = b[k] + sum_j w[k][j] x[row + j - DWK/2][k]: all tap loads first, then the sums in the tap order of
            // dwconv_kernel (grounding.hip)
            float wf[4 * DWN];
#pragma unroll
            for (int u = 0; u < DWN; ++u) {
                const f32x4 t4 = ((const f32x4*)(p.dw_w + (size_t)kc * DWK))[u];
                wf[4 * u] = t4[0]; wf[4 * u + 1] = t4[1]; wf[4 * u + 2] = t4[2]; wf[4 * u + 3] = t4[3];
            }
            const f32x4 bq = *(const f32x4*)(p.dw_b + kc);
            constexpr int RG = DWK > 3 ? 2 : NA;          // rows whose taps are in flight together (registers: RG * DWK quads)
#pragma unroll
            for (int i0 = 0; i0 < NA; i0 += RG) {
                f32x4 xv[RG][DWN];
#pragma unroll
                for (int i = 0; i < RG; ++i)
#pragma unroll
                    for (int jj = 0; jj < DWN; ++jj) xv[i][jj] = *(gf4p)(arow[i0 + i] + dwoff[i0 + i][jj] + acol);
#pragma unroll
                for (int i = 0; i < RG; ++i) {
                    f32x4 acc = bq;
#pragma unroll
                    for (int jj = 0; jj < DWN; ++jj) {
                        const bool ok = (dwmask[i0 + i] >> jj) & 1u;
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = ok ? acc[c] + wf[c * DWK + jj] * xv[i][jj][c] : acc[c];
                    }
                    ra[i0 + i] = acc;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ra[i] = *(gf4p)(arow[i] + acol);
            if (KTAIL && !kval) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto load_w = [&](int kt) {
        int kc = kt * BK + lc;
        bool kval = true;
        if (KTAIL) { kval = kc < p.K; kc = min(kc, p.K - 4); }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rb[i] = wrow[i][kc >> 2];
            if (KTAIL && !kval) rb[i] = u32x4{0u, 0u, 0u, 0u};
        }
    };
    // this thread's 8 bytes inside a row: chunk q>>1 (swizzled), half q&1
    int soff[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = lr + RPP * i;
        soff[i] = row * 64 + swz(q >> 1, row) * 16 + (q & 1) * 8;
    }
    auto store_tiles = [&](int buf, const f32x4 (&ra)[NA]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            uint2 h, r;
            split4(ra[i], h, r, chk, amax);
            if (FAST == 2) h = make_uint2(pk_bf16(ra[i][0], ra[i][1]), pk_bf16(ra[i][2], ra[i][3]));
            *(uint2*)(&smem[buf][soff[i]]) = h;
            if (!FAST) *(uint2*)(&smem[buf][PA + soff[i]]) = r;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (FAST == 2) {
                float w[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned hu = rb[i][j], ru = rb[i][2 + j];
                    const fp16x2 hh = *(const fp16x2*)&hu, rr = *(const fp16x2*)&ru;
                    w[2 * j] = (float)hh[0] + (float)rr[0] * (1.0f / RSCALE);
                    w[2 * j + 1] = (float)hh[1] + (float)rr[1] * (1.0f / RSCALE);
                }
                *(uint2*)(&smem[buf][2 * PA + soff[i]]) = make_uint2(pk_bf16(w[0], w[1]), pk_bf16(w[2], w[3]));
            } else {
                *(uint2*)(&smem[buf][2 * PA + soff[i]]) = make_uint2(rb[i][0], rb[i][1]);
            }
            if (!FAST) *(uint2*)(&smem[buf][2 * PA + PW + soff[i]]) = make_uint2(rb[i][2], rb[i][3]);
        }
    };

    f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[mi][ni][r] = 0.f; acc1[mi][ni][r] = 0.f; }

    // fragment addresses: lane = row l31 of a 32-row block, k = 16 s + 8 hi .. + 7 -> chunk 2 s + hi
    int aoff[2][2], boff[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ra_ = wm * 64 + mi * 32 + l31, rb_ = wn * 64 + mi * 32 + l31;
            aoff[mi][s] = ra_ * 64 + swz(2 * s + hi, ra_) * 16;
            boff[mi][s] = rb_ * 64 + swz(2 * s + hi, rb_) * 16;
        }

    // one K step: slab kt is in LDS buffer kt&1, `sp` holds the raw A rows of slab kt+1, `ld` receives slab kt+2
    // (slabs past the end repeat the last one: loaded and stored into the idle buffer, never read)
    auto step = [&](int kt, f32x4 (&ld)[NA], const f32x4 (&sp)[NA]) {
        const int buf = kt & 1;
        load_w(kofs + min(kt + 1, nk - 1));
        load_a(kofs + min(kt + 2, nk - 1), ld);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 a1[2], a2[2], b1[2], b2[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                a1[mi] = *(const f16x8*)(&smem[buf][aoff[mi][s]]);
                b1[mi] = *(const f16x8*)(&smem[buf][2 * PA + boff[mi][s]]);
                if (!FAST) {
                    a2[mi] = *(const f16x8*)(&smem[buf][PA + aoff[mi][s]]);
                    b2[mi] = *(const f16x8*)(&smem[buf][2 * PA + PW + boff[mi][s]]);
                }
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if (FAST == 2) acc0[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&a1[mi], *(const bf16x8*)&b1[ni], acc0[mi][ni], 0, 0, 0);
                    else acc0[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[mi], b1[ni], acc0[mi][ni], 0, 0, 0);
                    if (!FAST) {
                        acc1[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[mi], b2[ni], acc1[mi][ni], 0, 0, 0);
                        acc1[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[mi], b1[ni], acc1[mi][ni], 0, 0, 0);
                    }
                }
        }
        store_tiles(buf ^ 1, sp);
        __syncthreads();
    };

    load_a(kofs, ra0);
    load_w(kofs);
    load_a(kofs + min(1, nk - 1), ra1);
    store_tiles(0, ra0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, ra0, ra1);
        if (kt + 1 < nk) step(kt + 1, ra1, ra0);
    }
    // Range fallback: when any operand of this tile left the f16 range (or is NaN/Inf), the whole workgroup redoes its
    // tile on the exact f32 MFMA with operands read straight from memory (slow, rare: |x| >= 32768).  There is no lower
    // bound: an operand below the smallest normal f16 (6.1e-5) keeps an ABSOLUTE error of ~1.5e-11, which is what matters
    // next to operands of ordinary size in the same reduction; tensors that are small as a whole (gradients) are scaled by
    // a power of two by their producer (train_step.py: grad_scale).  A tile-level "all tiny" fallback was measured and
    // dropped: it doubled the training step (most gradient tiles took the slow path) for no change in the parity tests.
    if (__syncthreads_or((!(fabsf(chk) < INFINITY) || amax >= F16_LIMIT || *p.wbad == p.wbad_id) ? 1 : 0)) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[mi][ni][r] = 0.f; acc1[mi][ni][r] = 0.f; }
        // lane (l31, hi) of block mi / ni owns row m / n and the k quads 8c + 4hi .. + 3
        for (int k8 = kofs * BK; k8 < min(p.K, (kofs + nk) * BK); k8 += 8) {
            const int kq = k8 + 4 * hi;
            const bool kok = kq < p.K;
            const int kcl = min(kq, p.K - 4);
            f32x4 av[2], wv[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int mrow = min(m0 + wm * 64 + mi * 32 + l31, p.M - 1);
                const float* src = (p.a_rowptr ? (const float*)(const __attribute__((address_space(1))) float*)p.a_rowptr[mrow] : p.A + (size_t)mrow * p.lda) + kcl;
                bool zero = !kok;
                if (GATHER) {
                    const int seg = kcl / p.seg_len;
                    const int idx = p.a_idx[(size_t)mrow * p.nseg + seg];
                    zero = zero || idx < 0;
                    src = p.A + (size_t)max(idx, 0) * p.lda + (kcl - seg * p.seg_len);
                }
                av[mi] = *(const f32x4*)src;
                if (zero) av[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (PRE8) {
                    const float* xp = p.pre_x + (size_t)mrow * 8;
                    const f32x4 a0 = *(const f32x4*)xp, a1 = *(const f32x4*)(xp + 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float* wp = p.pre_w + (size_t)(kcl + c) * 8;
                        float v = a0[0] * wp[0];
                        v = fmaf(a0[1], wp[1], v); v = fmaf(a0[2], wp[2], v); v = fmaf(a0[3], wp[3], v);
                        v = fmaf(a1[0], wp[4], v); v = fmaf(a1[1], wp[5], v); v = fmaf(a1[2], wp[6], v); v = fmaf(a1[3], wp[7], v);
                        v += p.pre_b[kcl + c];
                        av[mi][c] = kok ? fmaxf(v, 0.f) : 0.f;
                    }
                }
                if (DWK > 0) {
                    int t, T;
                    if (p.dw_info) { const int2 v = p.dw_info[mrow]; t = v.x; T = v.y; } else { t = mrow % p.dw_T; T = p.dw_T; }
                    f32x4 acc = *(const f32x4*)(p.dw_b + kcl);
                    for (int jj = 0; jj < DWK; ++jj) {
                        const int tt = t + jj - DWK / 2;
                        if (tt < 0 || tt >= T) continue;
                        const f32x4 xq = *(const f32x4*)(p.A + (size_t)(mrow + jj - DWK / 2) * p.lda + kcl);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] += p.dw_w[(size_t)(kcl + c) * DWK + jj] * xq[c];
                    }
                    av[mi] = kok ? acc : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const int nrow = min(n0 + wn * 64 + mi * 32 + l31, p.N - 1);
                wv[mi] = *(const f32x4*)(p.W + (size_t)nrow * p.ldw + kcl);
                if (!kok) wv[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc0[mi][ni] = mfma32(av[mi][c], wv[ni][c], acc0[mi][ni]);
        }
    }

    if (p.splits > 1) {
        // split-K: the raw partial tile (no bias, activation or residual: splitk_reduce_kernel applies them to the sum)
        float* part = p.part + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int col = n0 + wn * 64 + ni * 32 + l31;
                if (col >= p.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + mi * 32 + mfma_row(r, hi);
                    if (row < p.M) part[(size_t)row * p.N + col] = acc0[mi][ni][r] + acc1[mi][ni][r] * RINV;
                }
            }
        return;
    }
    int colv[2];
    float bv[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        colv[ni] = n0 + wn * 64 + ni * 32 + l31;
        bv[ni] = p.bias ? p.bias[min(colv[ni], p.N - 1)] : 0.f;
    }
    const bool relu = p.act == BIGC_ACT_RELU, sigm = p.act == BIGC_ACT_SIGMOID, rmask = p.act == BIGC_ACT_RELU_MASK;
    constexpr int HP = 68;                      // pitch (floats) of the transposition corner of the split-form output
    constexpr float PS_SCALE = (float)(1 << PS_ACT_EXP);
    bool hr_oob = false;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int rbase = m0 + wm * 64 + mi * 32;
        f32x16 rv[2];
        if (p.res) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float* rp = p.res + min(colv[ni], p.N - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[ni][r] = rp[(size_t)min(rbase + mfma_row(r, hi), p.M - 1) * p.ldr];
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[ni][r] = 0.f;
        }
        if (!p.Cs) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if (colv[ni] >= p.N) continue;
                float* cp = p.C + colv[ni];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + mfma_row(r, hi);
                    float v = (acc0[mi][ni][r] + acc1[mi][ni][r] * RINV) + bv[ni];
                    if (relu) v = fmaxf(v, 0.f);
                    if (sigm) v = 1.0f / (1.0f + expf(-v));
                    if (rmask) v = rv[ni][r] > 0.f ? v : 0.f; else v += rv[ni][r];
                    if (row < p.M) cp[(size_t)row * p.ldc] = v;
                }
            }
            continue;
        }
        // Output in the split form "hr8" of gemm_ps.hip (the next GEMM's operand: per 8 columns, 16 bytes of f16 heads then 16 of
        // residuals of x * 2^PS_ACT_EXP).  A lane owns one column and 16 rows of a block; a group needs 8 columns of one row: the
        // wave's 32 x 64 values go through its own corner of the (now idle) staging LDS, natural row order, pitch 68 floats.
        float* tr = (float*)&smem[0][0] + wave * (32 * HP);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = (acc0[mi][ni][r] + acc1[mi][ni][r] * RINV) + bv[ni];
                if (relu) v = fmaxf(v, 0.f);
                if (sigm) v = 1.0f / (1.0f + expf(-v));
                if (rmask) v = rv[ni][r] > 0.f ? v : 0.f; else v += rv[ni][r];
                if (p.C && colv[ni] < p.N && rbase + mfma_row(r, hi) < p.M) p.C[(size_t)(rbase + mfma_row(r, hi)) * p.ldc + colv[ni]] = v;
                tr[mfma_row(r, hi) * HP + ni * 32 + l31] = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int lrow = pp * 8 + (lane >> 3), g = lane & 7;
            const f32x4 x0 = *(const f32x4*)(tr + lrow * HP + 8 * g) * PS_SCALE, x1 = *(const f32x4*)(tr + lrow * HP + 8 * g + 4) * PS_SCALE;
            const int row = rbase + lrow, col0 = n0 + wn * 64 + 8 * g;
            const bool ok = row < p.M && col0 < p.N;
            uint4 hh, rr;
            split_rtn2(x0[0], x0[1], hh.x, rr.x); split_rtn2(x0[2], x0[3], hh.y, rr.y);
            split_rtn2(x1[0], x1[1], hh.z, rr.z); split_rtn2(x1[2], x1[3], hh.w, rr.w);
#pragma unroll
            for (int c = 0; c < 4; ++c) hr_oob = hr_oob || (ok && (!(fabsf(x0[c]) < F16_LIMIT) || !(fabsf(x1[c]) < F16_LIMIT)));
            if (ok) {
                uint4* yp = (uint4*)(p.Cs + (size_t)row * p.ldcs + (size_t)(col0 >> 3) * 32);
                yp[0] = hh;
                yp[1] = rr;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (p.Cs && hr_oob && p.range_flag) atomicOr(p.range_flag, 1);
}

// C[m, n] = epilogue(sum over the splits of part[s][m][n] + bias[n]), the partials added in split order (the same order in every run);
// four columns per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, long M, int N, const float* __restrict__ bias,
                                                            int act, const float* __restrict__ res, int ldr, float* __restrict__ C, int ldc) {
    const int n4 = N / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * n4) return;
    const long m = i / n4;
    const int c = (int)(i - m * n4) * 4;
    const size_t plane = (size_t)M * N;
    const float* pp = part + (size_t)m * N + c;
    f32x4 v = *(const f32x4*)pp;
    for (int s = 1; s < splits; ++s) v += *(const f32x4*)(pp + (size_t)s * plane);
    if (bias) v += *(const f32x4*)(bias + c);
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
    if (res) rv = *(const f32x4*)(res + (size_t)m * ldr + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = v[j];
        if (act == BIGC_ACT_RELU) t = fmaxf(t, 0.f);
        if (act == BIGC_ACT_SIGMOID) t = 1.0f / (1.0f + expf(-t));
        if (act == BIGC_ACT_RELU_MASK) t = rv[j] > 0.f ? t : 0.f; else t += rv[j];
        v[j] = t;
    }
    *(f32x4*)(C + (size_t)m * ldc + c) = v;
}

// split weights: one allocation per (pointer, shape) = [int bad | pad to 256 B | N x K/4 groups]; re-split in place when the
// epoch differs (or on every launch at epoch 0, where a single scratch entry is reused in stream order)
struct WKey {
    const float* w; int n, k, ldw;
    bool operator==(const WKey& o) const { return w == o.w && n == o.n && k == o.k && ldw == o.ldw; }
};
struct WKeyHash { size_t operator()(const WKey& k) const { return std::hash<const void*>()(k.w) ^ ((size_t)k.n * 1315423911u) ^ ((size_t)k.k << 20) ^ (size_t)k.ldw; } };
struct WEntry { char* mem = nullptr; size_t cap = 0; unsigned long long epoch = 0; int bad_id = 0; };
std::unordered_map<WKey, WEntry, WKeyHash> g_cache;
WEntry g_scratch;
std::mutex g_cache_mu;
std::atomic<unsigned long long> g_epoch_seq{0};

int ensure_split(hipStream_t s, const LinearArgs& p, int K4, WEntry& e, bool fresh) {
    const size_t need = 256 + (size_t)p.N * K4 * sizeof(u32x4);
    if (need > e.cap) {
        if (e.mem) {
            if (hipStreamSynchronize(s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "linear_split: sync failed");
            (void)hipFree(e.mem);
        }
        e.cap = fresh ? need + need / 2 : need;
        if (hipMalloc((void**)&e.mem, e.cap) != hipSuccess || hipMemsetAsync(e.mem, 0, 256, s) != hipSuccess) {
            e.mem = nullptr; e.cap = 0;
            return bigc_fail(BIGC_ERR_HIP, "linear_split: hipMalloc of the split weights failed");
        }
    }
    ProfScope ps((void*)s, FAM_LINEAR_AUX, 0.0, 8.0 * p.N * (double)p.K);
    static std::atomic<int> bad_seq{0};
    e.bad_id = ++bad_seq;
    long total = (long)p.N * K4;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(split_weight_kernel, dim3(grid), dim3(256), 0, s, p.W, p.ldw, (u32x4*)(e.mem + 256), p.N, K4, (int*)e.mem, e.bad_id);
    BIGC_LAUNCH_CHECK("split_weight");
    return BIGC_OK;
}


// transposed f32 copies of weights (K, round4(N)), one per (pointer, shape); their split form lives in g_cache under the key of
// the copy itself, so the ordinary launch path finds it
struct TEntry { float* wT = nullptr; size_t cap = 0; unsigned long long epoch = 0; };
std::unordered_map<WKey, TEntry, WKeyHash> g_tcache;
std::atomic<int> g_bad_seq{1 << 20};

int reserve_split(hipStream_t s, WEntry& e, size_t need) {
    if (need <= e.cap) return BIGC_OK;
    if (e.mem) {
        if (hipStreamSynchronize(s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "linear_split: sync failed");
        (void)hipFree(e.mem);
    }
    e.cap = need;
    if (hipMalloc((void**)&e.mem, e.cap) != hipSuccess || hipMemsetAsync(e.mem, 0, 256, s) != hipSuccess) {
        e.mem = nullptr; e.cap = 0;
        return bigc_fail(BIGC_ERR_HIP, "linear_split: hipMalloc of the split weights failed");
    }
    return BIGC_OK;
}

// (caller holds g_cache_mu) refreshes every stale job of `jobs` for `epoch` with as few launches as possible
int prepare_locked(hipStream_t s, const SplitJob* jobs, int n, unsigned long long epoch) {
    PrepArgs a;
    a.n = 0; a.pad = 0;
    int blocks = 0;
    double bytes = 0.0;
    auto flush = [&]() -> int {
        if (a.n == 0) return BIGC_OK;
        ProfScope ps((void*)s, FAM_LINEAR_AUX, 0.0, bytes);
        hipLaunchKernelGGL(weight_prep_kernel, dim3(blocks), dim3(256), 0, s, a);
        BIGC_LAUNCH_CHECK("weight_prep");
        a.n = 0; blocks = 0; bytes = 0.0;
        return BIGC_OK;
    };
    for (int i = 0; i < n; ++i) {
        const SplitJob& q = jobs[i];
        PrepJob J;
        J.W = q.W; J.N = q.N; J.K = q.K; J.ldw = q.ldw; J.transposed = q.transposed ? 1 : 0; J.pad = 0; J.wT = nullptr;
        WEntry* e;
        long work;
        if (!q.transposed) {
            e = &g_cache[WKey{q.W, q.N, q.K, q.ldw}];
            if (epoch != 0 && e->epoch == epoch) continue;
            const int K4 = (q.K + 3) / 4;
            int rc = reserve_split(s, *e, 256 + (size_t)q.N * K4 * sizeof(u32x4));
            if (rc != BIGC_OK) return rc;
            work = ((long)q.N * K4 + 1023) / 1024;
        } else {
            TEntry& t = g_tcache[WKey{q.W, q.N, q.K, q.ldw}];
            const int n4 = (q.N + 3) / 4 * 4;
            if (epoch != 0 && t.epoch == epoch && t.wT) continue;
            const size_t need = (size_t)q.K * n4;
            if (need > t.cap) {
                if (t.wT) {
                    if (hipStreamSynchronize(s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "linear_split: sync failed");
                    auto sp = g_cache.find(WKey{t.wT, q.K, n4, n4});
                    if (sp != g_cache.end()) { if (sp->second.mem) (void)hipFree(sp->second.mem); g_cache.erase(sp); }
                    (void)hipFree(t.wT);
                }
                if (hipMalloc((void**)&t.wT, need * sizeof(float)) != hipSuccess) { t.wT = nullptr; t.cap = 0; return bigc_fail(BIGC_ERR_HIP, "linear_split: hipMalloc of a transposed weight failed"); }
                t.cap = need;
            }
            t.epoch = epoch;
            e = &g_cache[WKey{t.wT, q.K, n4, n4}];
            int rc = reserve_split(s, *e, 256 + (size_t)q.K * (n4 / 4) * sizeof(u32x4));
            if (rc != BIGC_OK) return rc;
            J.wT = t.wT;
            work = (long)((q.K + 63) / 64) * ((n4 + 63) / 64);
        }
        e->epoch = epoch;
        e->bad_id = ++g_bad_seq;
        J.Ws = (u32x4*)(e->mem + 256); J.bad = (int*)e->mem; J.bad_id = e->bad_id;
        J.nblk = (int)(work < 1 ? 1 : (work > 256 ? 256 : work));
        J.blk0 = blocks;
        blocks += J.nblk;
        bytes += 8.0 * q.N * (double)q.K * (q.transposed ? 1.5 : 1.0);
        a.j[a.n++] = J;
        if (a.n == PREP_MAX) { int rc = flush(); if (rc != BIGC_OK) return rc; }
    }
    return flush();
}

}  // namespace

void bigc_weight_cache_forget(const float* w) {
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        for (auto it = g_cache.begin(); it != g_cache.end();) {
            if (it->first.w == w) {
                if (it->second.mem) { (void)hipDeviceSynchronize(); (void)hipFree(it->second.mem); }
                it = g_cache.erase(it);
            } else {
                ++it;
            }
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        for (auto it = g_tcache.begin(); it != g_tcache.end();) {
            if (it->first.w == w) {
                if (it->second.wT) {
                    (void)hipDeviceSynchronize();
                    const int n4 = (it->first.n + 3) / 4 * 4;
                    auto sp = g_cache.find(WKey{it->second.wT, it->first.k, n4, n4});
                    if (sp != g_cache.end()) { if (sp->second.mem) (void)hipFree(sp->second.mem); g_cache.erase(sp); }
                    (void)hipFree(it->second.wT);
                }
                it = g_tcache.erase(it);
            } else {
                ++it;
            }
        }
    }
    bigc_gemm_ps_cache_forget(w);
}

thread_local unsigned long long bigc_static_weights_epoch = 0;
thread_local int bigc_splitk_allowed = 0;
extern "C" int bigc_linear_splitk_allow(int on) { const int prev = bigc_splitk_allowed; bigc_splitk_allowed = on ? 1 : 0; return prev; }
unsigned long long bigc_next_weights_epoch() { return ++g_epoch_seq; }
thread_local std::vector<SplitJob>* bigc_split_recorder = nullptr;

static void record_job(const SplitJob& q) {
    if (!bigc_split_recorder) return;
    for (const SplitJob& o : *bigc_split_recorder)
        if (o.W == q.W && o.N == q.N && o.K == q.K && o.ldw == q.ldw && o.transposed == q.transposed) return;
    bigc_split_recorder->push_back(q);
}

int bigc_linear_prepare(hipStream_t s, const SplitJob* jobs, int n) {
    if (n <= 0 || bigc_static_weights_epoch == 0) return BIGC_OK;
    std::lock_guard<std::mutex> lock(g_cache_mu);
    return prepare_locked(s, jobs, n, bigc_static_weights_epoch);
}

int bigc_weight_transposed(hipStream_t s, const float* W, int N, int K, int ldw, const float** wT) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    const SplitJob q{W, N, K, ldw, 1};
    const unsigned long long epoch = bigc_static_weights_epoch;
    auto it = g_tcache.find(WKey{W, N, K, ldw});
    if (epoch == 0 || it == g_tcache.end() || it->second.epoch != epoch || !it->second.wT) {
        int rc = prepare_locked(s, &q, 1, epoch);
        if (rc != BIGC_OK) return rc;
        if (epoch != 0) record_job(q);
        it = g_tcache.find(WKey{W, N, K, ldw});
    }
    *wT = it->second.wT;
    return BIGC_OK;
}


int bigc_linear_split_launch(hipStream_t s, LinearArgs& p, double flops, double bytes) {
    const int K4 = (p.K + 3) / 4;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        const unsigned long long epoch = bigc_static_weights_epoch;
        WEntry* e = &g_scratch;
        bool stale = true;
        if (epoch != 0) {
            e = &g_cache[WKey{p.W, p.N, p.K, p.ldw}];
            stale = e->epoch != epoch;
        }
        if (stale) {
            int rc = ensure_split(s, p, K4, *e, epoch == 0);
            if (rc != BIGC_OK) return rc;
            e->epoch = epoch;
            if (epoch != 0) record_job(SplitJob{p.W, p.N, p.K, p.ldw, 0});
        }
        p.Ws = (const u32x4*)(e->mem + 256); p.ldws = K4; p.wbad = (const int*)e->mem; p.wbad_id = e->bad_id;
    }
    if (p.a_idx) {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        static float* zero_row = nullptr;
        static int zero_cap = 0;
        if (p.seg_len > zero_cap) {
            if (zero_row) { if (hipStreamSynchronize(s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "linear_split: sync failed"); (void)hipFree(zero_row); }
            zero_cap = p.seg_len < 4096 ? 4096 : p.seg_len;
            if (hipMalloc((void**)&zero_row, sizeof(float) * zero_cap) != hipSuccess || hipMemsetAsync(zero_row, 0, sizeof(float) * zero_cap, s) != hipSuccess) {
                zero_row = nullptr; zero_cap = 0;
                return bigc_fail(BIGC_ERR_HIP, "linear_split: cannot allocate the zero row");
            }
        }
        p.zero_row = zero_row;
    }
    ProfScope ps((void*)s, FAM_LINEAR, flops, bytes);
    const bool ktail = p.K % BK != 0;
    static const int force_wn = getenv("BIGC_SPLIT_WN") ? atoi(getenv("BIGC_SPLIT_WN")) : 0;
    // 128x256 tiles (eight waves share one split of the A slab) unless the last column tile would be mostly empty:
    // 277 vs 260 TFLOP/s over the benchmark step's GEMMs (benchmarks/linear_shapes.py)
    const int rem = p.N % 256;
    // ... and unless that leaves half the chip idle (training-step sizes: 1536 x 512 is 24 wide tiles; 23 vs 28 us per launch)
    const bool few = (long)p.tiles_m * cdiv(p.N, 256) <= 128;
    const int wn = p.dw_w ? 2 : (force_wn ? (force_wn == 4 ? 4 : 2) : ((p.N > 192 && (rem == 0 || rem > 128) && !few) ? 4 : 2));
    p.tiles_n = cdiv(p.N, 64 * wn);
    const int grid2 = cdiv(p.tiles_m, 8) * 8 * p.tiles_n;
    // (The partial tiles are summed by a second launch.  Summing them in the kernel -- last workgroup to arrive at a tile, the scheme of
    // dw_finish_tile -- was built and measured: 14.4 instead of 9.4 ms per training step, the device-scope release every workgroup then
    // needs writes back its L2.)
    // split-K for small grids (inside a SplitKScope only: the training step; see linear_internal.h): enough splits to give every CU a
    // workgroup, each at least four slabs long, at most eight; partial tiles go to a stream-ordered scratch (one per thread, device AND stream: two streams driven from one thread
    // must not share the partials between a GEMM and its reduce)
    p.splits = 1; p.part = nullptr;
    static const bool sk_off = getenv("BIGC_SPLITK") && atoi(getenv("BIGC_SPLITK")) == 0;
    {
        const long tiles = (long)p.tiles_m * p.tiles_n;
        const int nk_tot = cdiv(p.K, BK);
        const bool aligned = p.C && !p.Cs && p.N % 4 == 0 && p.ldc % 4 == 0 && ((uintptr_t)p.C & 15) == 0 && (!p.bias || ((uintptr_t)p.bias & 15) == 0) &&
                             (!p.res || (p.ldr % 4 == 0 && ((uintptr_t)p.res & 15) == 0));
        if (bigc_splitk_allowed && !sk_off && aligned && !p.pre_x && !p.dw_w && tiles * 2 <= 256 && nk_tot >= 8) {
            static const int min_slabs = std::max(1, getenv("BIGC_SPLITK_MINSLABS") ? atoi(getenv("BIGC_SPLITK_MINSLABS")) : 4);
            int want = (int)(256 / tiles);
            if (want > nk_tot / min_slabs) want = nk_tot / min_slabs;
            if (want > 8) want = 8;
            if (want >= 2) {
                const int nkp = cdiv(nk_tot, want);
                p.splits = cdiv(nk_tot, nkp);
                struct Buf { float* p = nullptr; size_t cap = 0; };
                thread_local std::map<std::pair<int, void*>, Buf> bufs;
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "linear_split: hipGetDevice failed");
                Buf& b = bufs[std::make_pair(dev, (void*)s)];
                const size_t need = (size_t)p.splits * p.M * p.N;
                if (need > b.cap) {
                    if (b.p) { if (hipStreamSynchronize(s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "linear_split: sync failed"); (void)hipFree(b.p); }
                    b.cap = need + need / 2;
                    if (hipMalloc((void**)&b.p, b.cap * sizeof(float)) != hipSuccess) { b.p = nullptr; b.cap = 0; return bigc_fail(BIGC_ERR_HIP, "linear_split: hipMalloc of the split-K scratch failed"); }
                }
                p.part = b.p;
            }
        }
    }
    const dim3 grid3(grid2, p.splits);
    if (p.pre_x) {
        if (p.a_idx || ktail || p.dw_w) return bigc_fail(BIGC_ERR_ARG, "linear_split: preceding-layer mode needs K %% 32 == 0, no gather");
        if (wn == 4) hipLaunchKernelGGL((linear_split_kernel<false, false, 4, 0, true>), dim3(grid2), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((linear_split_kernel<false, false, 2, 0, true>), dim3(grid2), dim3(256), 0, s, p);
        BIGC_LAUNCH_CHECK("linear_split (pre8)");
        return BIGC_OK;
    }
    if (p.dw_w) {
        if (p.a_idx || ktail || (p.dw_ks != 3 && p.dw_ks != 7)) return bigc_fail(BIGC_ERR_ARG, "linear_split: depth-wise mode needs K %% 32 == 0, no gather, ks 3 or 7");
        if (p.dw_ks == 3) hipLaunchKernelGGL((linear_split_kernel<false, false, 2, 3>), dim3(grid2), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((linear_split_kernel<false, false, 2, 7>), dim3(grid2), dim3(256), 0, s, p);
        BIGC_LAUNCH_CHECK("linear_split (dwsep)");
        return BIGC_OK;
    }
#define LAUNCH_SPLIT(G, T, W) do { if (fast == 2) hipLaunchKernelGGL((linear_split_kernel<G, T, W, 0, false, 2>), grid3, dim3(128 * W), 0, s, p); \
                                   else if (fast) hipLaunchKernelGGL((linear_split_kernel<G, T, W, 0, false, 1>), grid3, dim3(128 * W), 0, s, p); \
                                   else hipLaunchKernelGGL((linear_split_kernel<G, T, W>), grid3, dim3(128 * W), 0, s, p); } while (0)
    const int fast = bigc_linear_get_fast();
    if (wn == 4) {
        if (p.a_idx) { if (ktail) LAUNCH_SPLIT(true, true, 4); else LAUNCH_SPLIT(true, false, 4); }
        else { if (ktail) LAUNCH_SPLIT(false, true, 4); else LAUNCH_SPLIT(false, false, 4); }
    } else {
        if (p.a_idx) { if (ktail) LAUNCH_SPLIT(true, true, 2); else LAUNCH_SPLIT(true, false, 2); }
        else { if (ktail) LAUNCH_SPLIT(false, true, 2); else LAUNCH_SPLIT(false, false, 2); }
    }
#undef LAUNCH_SPLIT
    BIGC_LAUNCH_CHECK("linear_split");
    if (p.splits > 1) {
        const long quads = (long)p.M * (p.N / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)((quads + 255) / 256)), dim3(256), 0, s, (const float*)p.part, p.splits, (long)p.M, p.N, p.bias, p.act,
                           p.res, p.ldr, p.C, p.ldc);
        BIGC_LAUNCH_CHECK("splitk_reduce");
    }
    return BIGC_OK;
}
