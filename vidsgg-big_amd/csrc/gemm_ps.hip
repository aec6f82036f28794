// GEMM on split operands: C = act(A W^T + bias) + residual with f32-level accuracy on the f16 matrix cores, for operands that are
// ALREADY stored as pairs of f16 numbers -- the form in which the producers of this library write their outputs.
//
// Split format "hr8" (same footprint and row pitch as f32).  An f32 x of a tensor with scale exponent E is stored as
//     X = x * 2^E,   h = f16(X),   r = f16(X - h)          (both round-to-nearest; X - h is exact in f32)
// and a row of K elements is K/8 groups of 32 bytes = [8 heads | 8 residuals].  The residual is kept UNSCALED (it may be an
// f16 subnormal: the f16 MFMA takes subnormal inputs exactly, benchmarks/probes/mfma_f16_denorm.hip), so that
//     a w 2^(Ea+Ew) = h_a h_w + h_a r_w + r_a h_w + O(2^-23)
// is ONE f32 accumulator and three v_mfma_f32_32x32x16_f16 per product.  Representation error of an element: 2^-23 |X| while
// |X| >= 0.25, 2^-25 absolute below (the residual's subnormal quantum): relative to the magnitudes that dominate a dot product
// this is the f32 round-off level as long as the tensor's typical |X| is well above 0.25.  Activations use the fixed exponent
// PS_ACT_EXP = 3 (|x| < 4096 representable, full relative precision for |x| >= 0.03, 3.7e-9 absolute below); every weight
// matrix gets its own exponent from its largest magnitude when it is split (max |W| in [2^13, 2^14)).
// (The first version of this kernel kept the residual scaled by 2^11 and needed a second accumulator for the correction terms:
// 128 accumulator registers for a 64 x 64 wave patch capped the tile at 128 x 256, i.e. 256 bytes of LDS fill per MFMA through a
// vector-memory path that takes 64 bytes per clock, and one barrier per 24 MFMAs per wave: 300 TFLOP/s.)
//
// Both operands are LDS-ready in memory, so they go HBM / L2 -> LDS by DMA (global_load_lds_dwordx4: no registers, no ds_write, no
// VALU).  A 128-byte LDS row holds the 32 k of a slab as 8 chunks [h0 r0 h1 r1 h2 r2 h3 r3]; the chunk index is XOR-swizzled
// by (row >> 1) & 7 -- applied on the SOURCE address of the DMA (its LDS image is lane-linear) and on the ds_read_b128 fragment
// reads, which are bank-conflict free (SQ_LDS_BANK_CONFLICT = 0).
//
// Tile: 256 output features x TM data rows per 512-thread workgroup (8 waves as 4 x 2, two per SIMD), TM = 256 (wave patch
// 64 x 128, two LDS stages of 64 KB) for the big-M GEMMs and TM = 128 (patch 64 x 64, three stages of 48 KB) when 256-row tiles
// would leave CUs idle.  The weights are the MFMA's A operand and the activations its B operand, so a lane of the result owns
// ONE data row and runs of four features: the epilogue stores 16 bytes per lane -- f32, and / or the split form for the next
// GEMM (one v_permlane32_swap per dword pairs the half-waves into full 16-byte head / residual chunks).
//
// A modes: 0 = hr8 rows by DMA; 1 = hr8 rows gathered through an index table (the stride-2 temporal conv over the virtual
// stretch: 3 segments of 2*dim_enti; index -1 = a row of zeros); 2 = f32 rows (raw input features, attention outputs):
// loaded to registers one slab ahead, split in the loop and written to LDS with ds_write_b64, W still by DMA;
// 3 = f32 rows by DMA, split IN PLACE in LDS (round 5): a K = 32 slab of an f32 row is 128 bytes, exactly the size of its split form, so
// the raw row goes HBM -> LDS by the same DMA pieces as an hr8 row (same swizzled chunk addresses: raw chunk c = floats 4 c .. 4 c + 3),
// and the wave that issued a piece later turns its 16 bytes per lane into 8 bytes of heads + 8 bytes of residuals and writes them over
// the piece (chunk pair (2 g, 2 g + 1) = [8 heads | 8 residuals] of floats 8 g .. 8 g + 7: a lane holding floats 8 g + 4 p .. + 3 writes
// its heads into half p of chunk 2 g and its residuals into half p of chunk 2 g + 1 -- its own 16-byte slot and its lane neighbour's).
// One ds_read_b128, ~14 VALU and two ds_write_b64 per lane and piece, no registers across the loop, no barrier of its own (a wave converts
// only what its own DMA brought, after its own vmcnt).  Row bases come from a table of addresses (bigc_batch.feat_row_ptr: the callers'
// per-tracklet rows read where they are, models/model_0v10.py:436-448) or A + m lda, kept per lane in LDS behind the stages.
//
// Range: an element with |X| >= 32768, NaN or Inf cannot be split.  Whoever splits (epilogue, A mode 2, the weight pre-pass)
// raises *range_flag; results are then invalid and the caller re-runs the step on the exact f32 path (bigc_hip/engine.py does,
// after the one D2H copy an eval step makes anyway).
//
// Roofline: dense f16 MFMA peak / 3 MFMAs per product.
#include "linear_internal.h"
#include "prof.h"
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

constexpr int TN = 256, TK = 32;
constexpr int ROWB = 128;                                    // bytes of an LDS row: 32 k x (head, residual)
constexpr float F16_LIMIT = 32768.0f;
constexpr float ACT_SCALE = (float)(1 << PS_ACT_EXP), ACT_INV = 1.0f / ACT_SCALE;

// two scaled f32 -> packed (head, head) and (residual, residual), round to nearest
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& r) {
    const f32x2 v = {x0, x1};
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const f32x2 d = v - __builtin_convertvector(hh, f32x2);
    const f16x2 rr = __builtin_convertvector(d, f16x2);
    h = *(const unsigned*)&hh;
    r = *(const unsigned*)&rr;
}

// header of a cached split weight: [0] int bad id, [1] max |W| bits, [2] float 2^-Ew, [3] float 2^Ew
__global__ __launch_bounds__(256) void weight_absmax_kernel(const float* __restrict__ W, int ldw, int N, int K4, unsigned* hdr) {
    float amax = 0.f;
    bool bad = false;
    const long total = (long)N * K4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / K4), k4 = (int)(i - (long)n * K4);
        const f32x4 a = *(const f32x4*)(W + (size_t)n * ldw + 4 * k4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { amax = fmaxf(amax, fabsf(a[c])); bad = bad || !(fabsf(a[c]) < INFINITY); }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((threadIdx.x & 63) == 0) atomicMax(hdr + 1, __float_as_uint(amax));          // non-negative floats order like their bit patterns
    if (bad) atomicMax(hdr + 1, 0x7f800000u);
}

// W (N, K) f32 -> hr8 (N, K/8) groups of 32 bytes, exponent chosen so that max |W| 2^Ew lies in [2^13, 2^14)
__global__ __launch_bounds__(256) void split_weight_hr8_kernel(const float* __restrict__ W, int ldw, uint4* __restrict__ Ws, int N, int K8,
                                                               unsigned* hdr, int bad_id) {
    const float amax = __uint_as_float(hdr[1]);
    int e = 0;
    const bool ok = amax > 0.f && amax < INFINITY;
    if (ok) { int ex; frexpf(amax, &ex); e = 14 - ex; }                       // amax = m 2^ex, m in [0.5, 1)
    e = max(-60, min(60, e));
    const float sc = ldexpf(1.0f, e);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ((float*)hdr)[2] = ldexpf(1.0f, -e);
        ((float*)hdr)[3] = sc;
        if (!(amax < INFINITY)) ((int*)hdr)[0] = bad_id;
    }
    const long total = (long)N * K8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / K8), k8 = (int)(i - (long)n * K8);
        const f32x4 a = *(const f32x4*)(W + (size_t)n * ldw + 8 * k8) * sc, b = *(const f32x4*)(W + (size_t)n * ldw + 8 * k8 + 4) * sc;
        uint4 h, r;
        split2(a[0], a[1], h.x, r.x); split2(a[2], a[3], h.y, r.y); split2(b[0], b[1], h.z, r.z); split2(b[2], b[3], h.w, r.w);
        Ws[2 * i] = h;
        Ws[2 * i + 1] = r;
    }
}

// f32 rows -> hr8 rows at the activation exponent (stand-alone conversion: tests, and producers that have no fused epilogue)
__global__ __launch_bounds__(256) void split_rows_hr8_kernel(const float* __restrict__ X, long ldx, unsigned char* __restrict__ Y, long ldy_bytes,
                                                             long rows, int K8, int* range_flag) {
    bool oob = false;
    const long total = rows * K8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / K8;
        const int k8 = (int)(i - n * K8);
        const f32x4 a = *(const f32x4*)(X + n * ldx + 8 * k8) * ACT_SCALE, b = *(const f32x4*)(X + n * ldx + 8 * k8 + 4) * ACT_SCALE;
        uint4 h, r;
        split2(a[0], a[1], h.x, r.x); split2(a[2], a[3], h.y, r.y); split2(b[0], b[1], h.z, r.z); split2(b[2], b[3], h.w, r.w);
#pragma unroll
        for (int c = 0; c < 4; ++c) oob = oob || !(fabsf(a[c]) < F16_LIMIT) || !(fabsf(b[c]) < F16_LIMIT);
        uint4* yp = (uint4*)(Y + n * ldy_bytes + 32 * (long)k8);
        yp[0] = h;
        yp[1] = r;
    }
    if (oob && range_flag) atomicOr(range_flag, 1);
}

struct PsArgs {
    const unsigned char* A; long lda;                 // A rows: hr8 (modes 0, 1) or f32 (modes 2, 3); pitch in BYTES
    const unsigned long long* a_rowptr;               // mode 3: address of row m (16-byte aligned, K floats) instead of A + m * lda, or null
    const int32_t* a_idx; int nseg; int seg_bytes;    // mode 1: source row of (output row, segment), -1 = zeros; bytes of a segment
    const unsigned char* zero_row;
    const unsigned char* Ws; long ldw;                // hr8 weights (N, K), pitch in bytes
    const unsigned* whdr; int wbad_id;                // header of the split weight (bad id, max bits, 2^-Ew, 2^Ew)
    const float* bias; const float* res; long ldr;    // residual pitch in floats
    int res_period;                                   // > 0: the residual of output row m is row m % res_period of `res` (a table that repeats per sequence)
    // LNM kernels: y = LayerNorm(act(A W^T + b) + res) * ln_g + ln_b (+ ln_add[m % ln_add_period]) over the N = 512 features of a row;
    // C receives y (+ the added table), Cs the split form of y without the table
    const float* ln_g; const float* ln_b; const float* ln_add; int ln_add_period; float ln_eps;
    float* C; long ldc;                               // f32 output (or null), pitch in floats
    unsigned char* Cs; long ldcs;                     // hr8 output (or null), pitch in bytes
    int M, N, K, act, tiles_m, tiles_n;
    int* range_flag;
    int blk;                     // BLOCKED kernels only: K slabs per accumulation block (a power of two), see `tot` in the kernel
    int ablate;                  // benchmarking only (BIGC_PS_ABLATE): 1 = no DMA in the K loop
    unsigned long long* trace;   // benchmarking only (BIGC_PS_TRACE): per wave, cycles spent in the segments of an iteration
};

// Persistent workgroups: the launch has (at most) one workgroup per CU; workgroup b walks the tiles b, b + G, b + 2 G, ... of an
// XCD-aware order and treats their K slabs as ONE stream: the DMA engine runs ahead of the MFMAs across tile boundaries, so a
// tile's first slabs arrive while the previous tile's last MFMAs and its epilogue run (with 128 - 144 KB of LDS there is one
// workgroup per CU: nothing else would hide a tile's pipeline fill).
//   RB = 32-row blocks per wave: 2 -> TM = 128 rows, 3 LDS stages;  3 -> TM = 192 rows, 2 stages (M = 49 152, the decoder of a 256-video
//   step: 256 x 2 tiles = two exact rounds of the 256 persistent workgroups);  4 -> TM = 256 rows, 2 LDS stages.
//   FAST: reduced-precision mode (bigc_linear_set_fast): heads only, one MFMA per product; the residual chunks still travel
//   with their rows (the format is the producers') but are not read from LDS.
//   LNM: the tile spans ALL 512 output features (eight waves side by side, 64 features each, RB row blocks: TM = 32 RB rows), so that the
//   epilogue can take a row's LayerNorm (models/model_0v10.py:188,211,223: norm1 / norm2 / norm3 of a decoder layer) before it stores:
//   the f32 pre-norm tensor and the LayerNorm pass that read it disappear (one write + one read of (B Q, 512) f32 per LayerNorm).
//   ACC2 (128-row tiles): the two correction terms of a product (head x residual, residual x head: 2^-11 of the leading term) go to a SECOND
//   accumulator, so that the leading accumulator is rounded once per k16 step instead of three times -- the scheme of linear_split.hip; the
//   residuals are unscaled in this format, so the two are simply added in the epilogue.  rms error 10.3 -> ~6 units of 2^-24 at K = 3072.
//   NWV (round 5): waves per workgroup.  8 = two waves per SIMD, 256 registers each (everything above).  4 = ONE wave per SIMD with the
//   whole 512-register file: tile 256 features x 128 rows, four waves side by side over the features, each a 64 x 128 patch (RB = 4) -- the
//   fragment-read economy of the 256-row tile (0.25 ds_read_b128 pairs per MFMA instead of 0.33) with room for the second register set of the
//   blocked accumulation (128 + 128 accumulator registers) and for addresses the 256-register kernels recompute per slab.
template <int AMODE, int RB, bool TRACE = false, bool FAST = false, bool BLOCKED = false, bool LNM = false, bool ACC2 = false, int NWV = 8>
__global__ __launch_bounds__(64 * NWV, 1) void gemm_ps_kernel(const PsArgs p) {
    static_assert(NWV == 8 || (NWV == 4 && RB == 4 && BLOCKED && !LNM && !TRACE && !ACC2 && AMODE <= 1), "four-wave form: blocked accumulation, hr8 operands, 64 x 128 wave patches");
    constexpr int NTHR = 64 * NWV;
    static_assert(!ACC2 || (RB == 2 && !FAST && !BLOCKED && !LNM), "second accumulator: 128-row tiles only (registers)");
    static_assert(AMODE != 3 || (RB != 3 && !LNM), "in-LDS split of DMA'd f32 rows: 128- or 256-row tiles");
    static_assert(!LNM || (AMODE == 0 && (RB == 2 || RB == 3) && !BLOCKED && !TRACE), "LayerNorm epilogue: plain hr8 operand, 64- or 96-row tiles");
    constexpr int TNK = LNM ? 512 : TN;                       // output features per tile
    constexpr int NWF = TNK / 64, NWR = NWV / NWF;            // waves side by side over the features / stacked over the rows
    constexpr int TM = 32 * RB * NWR;
    constexpr int WB = TNK * ROWB;
    constexpr int A_BYTES = TM * ROWB, STAGE_BYTES = A_BYTES + WB;
    constexpr int NST = ((RB == 2 && !LNM) || NWV == 4) ? 3 : 2, D = NST - 1;   // LDS stages (48 KB ones: three); slabs in flight beyond the one being read
    constexpr int NL = RB * NWR / 2;                          // mode 2: f32x4 loads per thread and slab
    constexpr int MAXSEG = 4;                                 // mode 1: K segments (gathered row tables) supported
    constexpr int IDX_OFF = NST * STAGE_BYTES;                // mode 1: behind the stages, a lane-private table of source rows (below)
    constexpr int LN_OFF = NST * STAGE_BYTES;                 // LNM: behind the stages, per row and wave the partial sums of the two LayerNorm passes
    constexpr int PTR_OFF = NST * STAGE_BYTES;                // mode 3: behind the stages, a lane-private table of row base addresses (u64)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE_BYTES + (AMODE == 1 ? ((TM / 8 + NWV - 1) / NWV) * MAXSEG * NTHR * 4 : 0) +
                                                                 (LNM ? 2 * TM * 8 * 4 : 0) + (AMODE == 3 ? ((TM / 8 + 7) / 8) * 4096 : 0)];
    AS3 unsigned char* lds = (AS3 unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = wave % NWF, wr = wave / NWF, hi = lane >> 5, l31 = lane & 31;
    const int nk = p.K / TK;

    // ---- tile order.  Workgroup b runs on XCD b % 8 (observed placement; a speed matter only): the workgroups of an XCD take
    // consecutive tiles (column tile fastest) per round, so the column tiles of an A row tile run side by side on one L2.
    const int G = gridDim.x, S = G >> 3, b = blockIdx.x;
    const int n_tiles = p.tiles_m * p.tiles_n;
    auto tile_of = [&](int j) { return (j * 8 + (b & 7)) * S + (b >> 3); };     // linear tile index of this workgroup's j-th tile
    int my_tiles = 0;
    while (tile_of(my_tiles) < n_tiles) ++my_tiles;
    if (my_tiles == 0) return;
    const int n_flat = my_tiles * nk;

    // ---- DMA sources.  One wave instruction moves 8 rows x 128 bytes; lane l lands at LDS position (row l >> 3, chunk l & 7)
    // of its block and therefore FETCHES the chunk (l & 7) ^ swizzle(row) of that row.
    // Wave w owns the row blocks w, w + 8, ... of both operands.
    // (Measured and dropped: waves 4..7 as dedicated loaders issuing all pieces, their SIMD partners 0..3 pure MFMA + ds_read:
    // 16 issues of ~100+ clocks per slab and loader made the loaders the critical path, conv GEMM 373 -> 214 TFLOP/s.)
    // Address = scalar block base + 32-bit per-lane offset (row clamped to the matrix, chunk swizzled), computed per piece.
    constexpr bool loader = true;
    const int sw_src = (((wave & 1) << 2) + (lane >> 4)) & 7;             // ((8 blk + (l >> 3)) >> 1) & 7, same for blk and blk + 8 i
    const int coff = ((lane & 7) ^ sw_src) << 4;
    const int l8 = lane >> 3;
    constexpr int NBLK_A = TM / 8;                                        // 8-row blocks of an A slab
    constexpr int NA = (NBLK_A + NWV - 1) / NWV;                          // A pieces per wave and slab (row blocks w + NWV i; LNM with 96 rows: 12 blocks, waves 4..7 skip their second)
    constexpr int NW = TNK / 8 / NWV;                                     // W pieces per wave and slab
    constexpr int NPCL = (AMODE == 2 ? 0 : NA) + NW;                      // DMA pieces per wave and slab
    // issue-side state (runs ahead of the compute side)
    int ij = 0, ikt = 0;
    const unsigned char* ia_base = nullptr;                               // A + m0 * lda of the tile being issued
    // The DMA pieces of W (every mode) and of a plain hr8 A (mode 0) are BUFFER loads: descriptor in scalar registers (W: the whole
    // matrix; A: from the tile's first row on), the lane's (row in block, swizzled chunk) as a 32-bit vector offset that never changes,
    // and tile / slab / block as the scalar offset -- no vector address arithmetic per piece (with global_load_lds every piece paid a
    // clamp, a multiply and a 64-bit add: 3.5 VALU instructions per MFMA on the conv GEMM, whose matrix pipe was 60 % busy).  Rows past
    // the matrix fail the descriptor's range check and read as zeros (their products are never stored).
    constexpr int RSRC_FLAGS = 0x00020000;
    const size_t w_bytes = (size_t)p.N * p.ldw;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ws, 0, (int)(w_bytes < 0xffffffffull ? w_bytes : 0xffffffffull), RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t a_rsrc = w_rsrc;                               // mode 0: set per tile
    int iw_off = 0;                                                       // n0 * ldw
    // mode 1: source row of this lane's row of piece i in K segment sg (-1 = zeros).  The slabs of a gathered GEMM are walked
    // SEGMENT-INTERLEAVED -- slab ikt = (chunk ikt / nseg, segment ikt % nseg) -- so that the taps of the temporal conv, which
    // read the same source rows (output position t reads the stretched frames 2t-1, 2t, 2t+1: neighbouring positions share a
    // frame, and the stretch repeats rows), fetch a 128-byte line from all segments within a few slabs of each other: the
    // repeats hit in L2 instead of going back to HBM a third of a K loop later (tap-major order: 6.9 GB fetched for 2.4 GB of
    // distinct rows per 256-video step).  Every lane then needs NA * nseg row numbers at hand; the kernel is at its register
    // limit (twelve more live registers spilled in the K loop: 4.5 -> 6.1 ms), so they live in LDS, slot (piece, segment) of
    // lane tid at IDX_OFF + ((piece * MAXSEG + segment) * 512 + tid) * 4: written once per tile, one ds_read_b32 per piece.
    int isg = 0, ichunk = 0;                                              // mode 1: (segment, chunk) of slab ikt
    int wof[NL];
    f32x4 ra[NL];
    if (AMODE == 2) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int row = 64 * i + (tid >> 3), q = tid & 7;
            wof[i] = row * ROWB + (((2 * (q >> 1)) ^ ((row >> 1) & 7)) << 4) + ((q & 1) << 3);      // head chunk; the residual chunk is ^ 16
        }
    }
    auto issue_tile_setup = [&]() {
        const int L = tile_of(ij), tm = L / p.tiles_n, tn = L - tm * p.tiles_n;
        const int m0 = tm * TM, n0 = tn * TNK;
        ia_base = p.A + (size_t)m0 * p.lda;
        iw_off = n0 * (int)p.ldw;
        if (AMODE == 0) {
            const size_t a_bytes = (size_t)(p.M - m0) * p.lda;
            a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ia_base, 0, (int)(a_bytes < 0xffffffffull ? a_bytes : 0xffffffffull), RSRC_FLAGS);
        }
        if (AMODE == 3) {
            // row base addresses of this lane's rows of the tile's NA pieces (one batch of table reads per tile)
            unsigned long long v[NA];
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int row = min(m0 + (wave + NWV * q) * 8 + l8, p.M - 1);
                v[q] = p.a_rowptr ? p.a_rowptr[row] : (unsigned long long)(p.A + (size_t)row * p.lda);
            }
#pragma unroll
            for (int q = 0; q < NA; ++q) *(AS3 unsigned long long*)(lds + PTR_OFF + (q * NTHR + tid) * 8) = v[q];
        }
        if (AMODE == 1) {
            // one batch of table reads per tile (the wait in front of the ds_writes also waits for the slabs in flight)
            int v[NA * MAXSEG];
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int32_t* ip = p.a_idx + (size_t)min(m0 + (wave + NWV * q) * 8 + l8, p.M - 1) * p.nseg;
#pragma unroll
                for (int sg = 0; sg < MAXSEG; ++sg) v[q * MAXSEG + sg] = ip[min(sg, p.nseg - 1)];
            }
#pragma unroll
            for (int q = 0; q < NA * MAXSEG; ++q) *(AS3 int*)(lds + IDX_OFF + (q * NTHR + tid) * 4) = v[q];
        }
    };
    // (`so` = byte offset of an LDS stage, a wave-uniform runtime value)
    const int w_vo = (wave * 8 + l8) * (int)p.ldw + coff, a_vo = (wave * 8 + l8) * (int)p.lda + coff;       // the lane's part of every W / A piece address
    auto issue_w_piece = [&](int i, int so) {
        const int blk = wave + NWV * i;
        const int kb = AMODE == 1 ? isg * p.seg_bytes + ichunk * ROWB : ikt * ROWB;      // K offset of the slab in a W row
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (AS3 void*)(lds + so + A_BYTES + blk * 1024), 16, w_vo, iw_off + kb + i * (NWV * 8) * (int)p.ldw, 0, 0);
    };
    auto issue_a_piece = [&](int i, int so) {
        const int blk = wave + NWV * i;
        if (NBLK_A % NWV != 0 && blk >= NBLK_A) return;
        if (AMODE == 3) {
            const unsigned long long rb = *(const AS3 unsigned long long*)(lds + PTR_OFF + (i * NTHR + tid) * 8);
            __builtin_amdgcn_global_load_lds((const AS1 void*)((const unsigned char*)rb + (size_t)ikt * ROWB + coff), (AS3 void*)(lds + so + blk * 1024), 16, 0, 0);
        } else if (AMODE == 1) {
            const int gi = *(const AS3 int*)(lds + IDX_OFF + ((i * MAXSEG + isg) * NTHR + tid) * 4);
            const unsigned char* src = (gi >= 0 ? p.A + (size_t)gi * p.lda : p.zero_row) + coff + ichunk * ROWB;    // index -1: a row of zeros
            __builtin_amdgcn_global_load_lds((const AS1 void*)src, (AS3 void*)(lds + so + blk * 1024), 16, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (AS3 void*)(lds + so + blk * 1024), 16, a_vo, ikt * ROWB + i * (NWV * 8) * (int)p.lda, 0, 0);
        }
    };
    // piece k of the slab being issued: modes 0 / 1: A pieces then W pieces; mode 2: W pieces only
    auto issue_piece = [&](int k, int so) {
        if (AMODE == 2) issue_w_piece(k, so);
        else if (k < NA) issue_a_piece(k, so);
        else issue_w_piece(k - NA, so);
    };
    auto issue_slab = [&](int so) {
        if (loader) {
#pragma unroll
            for (int k = 0; k < NPCL; ++k) issue_piece(k, so);
        }
    };
    auto issue_advance = [&]() {
        if (AMODE == 1 && ++isg == p.nseg) { isg = 0; ++ichunk; }
        if (++ikt == nk) { ikt = 0; ++ij; isg = 0; ichunk = 0; }
    };
    // mode 2: plain register loads.  They are consumed (split_store) right before a slab's wait + barrier.  (Issued by inline
    // asm to hide them from the compiler's wait bookkeeping they are NOT safe: under register pressure the compiler copies the
    // destination registers before the data has landed.)
    auto load_f32_at = [&](int j, int kt) {
        const int L = tile_of(j), tm = L / p.tiles_n;
        const int m0 = tm * TM;
        const unsigned char* base = p.A + (size_t)m0 * p.lda + (size_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < NL; ++i)
            ra[i] = *(const f32x4*)(base + (size_t)min(64 * i + (tid >> 3), p.M - 1 - m0) * p.lda + 16 * (tid & 7));
    };
    float amax = 0.f;            // mode 2: largest |x| split so far; NaN-aware through the final comparison !(amax < limit) of a max3 chain
    float nanacc = 0.f;          //         (v_max3 drops NaNs: a running sum of x * 0 carries NaN / Inf)
    auto split_store = [&](int so) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            unsigned h0, h1, r0, r1;
            split2(ra[i][0] * ACT_SCALE, ra[i][1] * ACT_SCALE, h0, r0);
            split2(ra[i][2] * ACT_SCALE, ra[i][3] * ACT_SCALE, h1, r1);
            // asm volatile: stays HERE, ahead of the loads that refill ra (the compiler would otherwise sink these uses below
            // them and keep two register sets alive, with a copy and a full vmcnt(0) on the loop's back edge)
            asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(ra[i][0]), "v"(ra[i][1]));
            asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(ra[i][2]), "v"(ra[i][3]));
            asm volatile("v_fmac_f32 %0, 0, %1" : "+v"(nanacc) : "v"((ra[i][0] + ra[i][1]) + (ra[i][2] + ra[i][3])));
            *(AS3 u32x2*)(lds + so + wof[i]) = u32x2{h0, h1};
            *(AS3 u32x2*)(lds + so + (wof[i] ^ 16)) = u32x2{r0, r1};
        }
    };
    // mode 3: piece i of the slab in stage `so` (this wave's own DMA, landed: the caller waited on vmcnt) from raw f32 to the split form, in
    // place.  conv_read fetches the lane's 16 bytes, conv_write splits them and stores heads / residuals (see the header comment).
    const int pbit = ((lane & 7) ^ sw_src) & 1;                            // this lane's chunk holds floats 8 g + 4 pbit .. + 3 of its group
    // (ONE register set: piece i + 1 is fetched after piece i has been converted; four pieces' worth of registers spilled in the 256-row
    // kernel, and a spill reload in the loop is a scratch load, i.e. an s_waitcnt vmcnt(0) that drains the DMAs)
    // Register sets: CVS = 2 at 128-row tiles (both pieces are fetched at the start of the unit and converted ten MFMAs later: an LDS read
    // under the load of eight waves' fragment reads and the DMA writes takes several hundred cycles, and a wave that waits for it issues no
    // MFMAs -- fetched one MFMA pair ahead of its use the conversion cost 0.95 ms on the K = 2048 GEMM); CVS = 1 at 256-row tiles (no
    // registers left: piece i + 1 is fetched after piece i has been converted).
    constexpr int CVS = RB == 2 ? 2 : 1;
    f32x4 cv[CVS];
    const int conv_base = lane * 16;
    auto conv_read = [&](int i, int so) {
        const int blk = wave + NWV * i;
        cv[i % CVS] = *(const AS3 f32x4*)(lds + so + blk * 1024 + conv_base);
    };
    auto conv_write = [&](int i, int so) {
        const int blk = wave + NWV * i;
        const f32x4 x = cv[i % CVS];
        unsigned h0, h1, r0, r1;
        split2(x[0] * ACT_SCALE, x[1] * ACT_SCALE, h0, r0);
        split2(x[2] * ACT_SCALE, x[3] * ACT_SCALE, h1, r1);
        asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(x[0]), "v"(x[1]));
        asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(x[2]), "v"(x[3]));
        asm volatile("v_fmac_f32 %0, 0, %1" : "+v"(nanacc) : "v"((x[0] + x[1]) + (x[2] + x[3])));     // NaN / Inf x 0 = NaN (v_max3 drops NaNs)
        const int a = so + blk * 1024 + conv_base;
        // The stores are issued from inline asm: as plain stores the compiler puts an s_waitcnt vmcnt(0) in front of each of them (a store to
        // LDS while LDS-DMAs are in flight might overlap one of them for all it knows), which drains the W pieces of the slab under issue
        // once per piece: 4.31 instead of 3.40 ms on the K = 2048 GEMM.  They target this wave's own, landed pieces; nothing reads them
        // before the s_waitcnt lgkmcnt(0) + barrier that close the slab (cdna_hip_programming.md 5.7: a store has no destination
        // register, hidden LDS operations only make the compiler's own lgkmcnt waits more conservative).
        const unsigned ah = (unsigned)(unsigned long)(lds + (a ^ (pbit << 4)) + 8 * pbit), ar = (unsigned)(unsigned long)(lds + (a ^ ((pbit ^ 1) << 4)) + 8 * pbit);
        const u32x2 hv = {h0, h1}, rv = {r0, r1};
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %2, %3" :: "v"(ah), "v"(hv), "v"(ar), "v"(rv) : "memory");
    };
    bool oob = false;

    f32x16 acc[2][RB];
    // Blocked accumulation (template parameter BLOCKED, RB = 2 only; opt-in through BIGC_PS_BLK, not used by the model).  Every MFMA
    // rounds its accumulator once, by ~2^-24 of the accumulator's CURRENT magnitude, and this kernel issues three MFMAs per k16 step
    // into ONE accumulator: over K = 3072 that is 576 roundings at the magnitude of the growing sum -- measured 10.3 x 2^-24 rms of
    // the result against 6.0 for the two-accumulator kernel of linear_split.hip and 3.2 for torch's CPU GEMM (benchmarks/probes/
    // gemm_error_stats.py).  With BLOCKED, `acc` only ever holds the partial sum of p.blk slabs (small, so its roundings are small)
    // and is folded into the running total `tot` once per block: measured 2.97 (K = 3072), 2.87 (K = 2048), 2.28 (K = 512) -- below
    // the CPU GEMM's.  Round 5: blocks of BLKC = 4 slabs (K = 128) with the fold behind an inner loop of fixed length (below): 217 - 241
    // registers at 128-row tiles, no spills (as a conditional in the slab loop: 256 + 100 spilled, 60 TFLOP/s).  The 192- and 256-row
    // tiles have no room for the second register set (70 / 163 spilled).  Over the 256 benchmark videos it is what brings EVERY video
    // within 1e-4 of a float64 evaluation (max 7.8e-5; one accumulator: 1.65e-4, second accumulator for the correction terms: 1.78e-4;
    // benchmarks/accuracy_sweep.py, profiles/r05a_accuracy_sweep_policies.jsonl).
    static_assert(!BLOCKED || ((RB == 2 || NWV == 4) && !FAST && !LNM && !ACC2), "blocked accumulation: 128-row tiles only");
    f32x16 tot[2][BLOCKED ? RB : 1];
    f32x16 acc2[2][ACC2 ? RB : 1];
    auto zero_acc = [&]() {
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int ri = 0; ri < RB; ++ri)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[fi][ri][r] = 0.f; if (BLOCKED) tot[fi][ri][r] = 0.f; if (ACC2) acc2[fi][ri][r] = 0.f; }
    };
    zero_acc();
    auto fold_block = [&]() {
        if (BLOCKED) {
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int ri = 0; ri < RB; ++ri) tot[fi][ri] += acc[fi][ri];
        }
    };

    // fragment reads: lane = row l31 of a 32-row block, chunk 2 (2 s + hi) + plane, swizzled by (row >> 1) & 7 = (l31 >> 1) & 7
    const int sw_rd = (l31 >> 1) & 7;
    int cs[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) cs[s][pl] = ((2 * (2 * s + hi) + pl) ^ sw_rd) << 4;
    const int rbw = A_BYTES + (wf * 64 + l31) * ROWB, rba = (wr * 32 * RB + l31) * ROWB;

    // Fragments are read and consumed in UNITS of 12 MFMAs: (k16 step s, row-block pair u) = the wave's 2 feature blocks x 2 of
    // its RB row blocks.  W fragments (2 blocks x head / residual) live per step, A fragments (2 blocks x head / residual) per
    // unit, each double-buffered: 64 registers whatever RB is.
    // RB = 3: one unit per step of 2 x 3 blocks (18 MFMAs).
    constexpr int UR = RB == 3 ? 3 : 2;                        // row blocks per unit
    constexpr int U = RB / UR, NU = 2 * U;                     // units per k16 step / per slab
    constexpr int MM = 6 * UR;                                 // MFMAs per unit
    struct WFrag { f16x8 h[2], l[2]; };
    struct AFrag { f16x8 h[UR], l[UR]; };
    WFrag WF[2];
    AFrag AF[2];
    auto rd_w = [&](int so, int s, WFrag& f) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.h[i] = *(const AS3 f16x8*)(lds + so + rbw + i * 32 * ROWB + cs[s][0]);
            if (!FAST) f.l[i] = *(const AS3 f16x8*)(lds + so + rbw + i * 32 * ROWB + cs[s][1]);
        }
    };
    auto rd_a = [&](int so, int s, int u, AFrag& f) {
#pragma unroll
        for (int i = 0; i < UR; ++i) {
            f.h[i] = *(const AS3 f16x8*)(lds + so + rba + (UR * u + i) * 32 * ROWB + cs[s][0]);
            if (!FAST) f.l[i] = *(const AS3 f16x8*)(lds + so + rba + (UR * u + i) * 32 * ROWB + cs[s][1]);
        }
    };
    // MFMAs [m0, m1) of the MM of unit u: term-major (head x head over the 2 x UR blocks, then head x residual, residual x head), so
    // that consecutive MFMAs never accumulate into the same block
    // `fresh`: the head x head MFMA of each block starts a NEW sum (C = 0, an inline constant) instead of continuing acc -- the first k16 step of
    // an accumulation block's first slab (BLOCKED): the fold then only adds acc to tot and nothing has to be zeroed
    auto mm = [&](const WFrag& w, const AFrag& a, int u, int m0, int m1, bool fresh = false) {
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            if (m < m0 || m >= m1) continue;
            const int t = m / (2 * UR), fi = (m % (2 * UR)) / UR, ri = m % UR;
            if (FAST && t > 0) continue;
            if (ACC2 && t > 0)
                acc2[fi][UR * u + ri] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t == 2 ? w.l[fi] : w.h[fi], t == 1 ? a.l[ri] : a.h[ri], acc2[fi][UR * u + ri], 0, 0, 0);
            else if (BLOCKED && fresh && t == 0) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[fi][UR * u + ri] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.h[fi], a.h[ri], zero, 0, 0, 0);
            } else
                acc[fi][UR * u + ri] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t == 2 ? w.l[fi] : w.h[fi], t == 1 ? a.l[ri] : a.h[ri], acc[fi][UR * u + ri], 0, 0, 0);
        }
    };

    // ---- epilogue of one tile: lane (l31, hi) of block (fi, ri) owns data row m and the features nb + 8 g + 4 hi + {0..3}
    // (registers 4 g ..).  Loads are unconditional from clamped addresses and issued in batches (a null bias points at zeros).
    const bool sigm = p.act == BIGC_ACT_SIGMOID;
    const float lo = p.act == BIGC_ACT_RELU ? 0.f : -INFINITY;
    const float osc = ((const float*)p.whdr)[2] * ACT_INV;                  // 2^-(Ew + Ea)
    auto epilogue = [&](int j) __attribute__((always_inline)) {
        const int L = tile_of(j), tm = L / p.tiles_n, tn = L - tm * p.tiles_n;
        const int m0 = tm * TM, n0 = tn * TNK;
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
            f32x4 bv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[g] = *(const f32x4*)(p.bias + min(n0 + wf * 64 + fi * 32 + 8 * g + 4 * hi, p.N - 4));
#pragma unroll
            for (int ri = 0; ri < RB; ++ri) {
                const int m = m0 + wr * 32 * RB + ri * 32 + l31;
                const bool mok = m < p.M;
                const int mc = min(m, p.M - 1);
                const int nb = n0 + wf * 64 + fi * 32 + 4 * hi;
                f32x4 rv[4];
                if (p.res) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) rv[g] = *(const f32x4*)(p.res + (size_t)(p.res_period > 0 ? mc % p.res_period : mc) * p.ldr + min(nb + 8 * g, p.N - 4));
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) rv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                f32x4 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float t = (BLOCKED ? tot[fi][ri][4 * g + c] : (ACC2 ? acc[fi][ri][4 * g + c] + acc2[fi][ri][4 * g + c] : acc[fi][ri][4 * g + c])) * osc + bv[g][c];      // (blocked: the last block was folded in)
                        t = fmaxf(t, lo);
                        if (sigm) t = 1.0f / (1.0f + expf(-t));
                        v[g][c] = t + rv[g][c];
                    }
                }
                if (p.C) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (mok && nb + 8 * g < p.N) *(f32x4*)(p.C + (size_t)m * p.ldc + nb + 8 * g) = v[g];
                }
                if (p.Cs) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const bool ok = mok && nb + 8 * g < p.N;
                        const f32x4 x = v[g] * ACT_SCALE;
                        unsigned h0, h1, r0, r1;
                        split2(x[0], x[1], h0, r0);
                        split2(x[2], x[3], h1, r1);
#pragma unroll
                        for (int c = 0; c < 4; ++c) oob |= ok && !(fabsf(x[c]) < F16_LIMIT);
                        // half-wave exchange: lanes hi = 0 end up with [own heads | partner's heads], lanes hi = 1 with
                        // [partner's residuals | own residuals] -- the 16-byte head / residual chunks of feature group n / 8
                        const auto sx = __builtin_amdgcn_permlane32_swap(h0, r0, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(h1, r1, false, false);
                        if (ok) *(u32x4*)(p.Cs + (size_t)m * p.ldcs + (size_t)((nb + 8 * g) >> 3) * 32 + hi * 16) = u32x4{sx[0], sy[0], sx[1], sy[1]};
                    }
                }
            }
        }
    };

    // ---- LNM epilogue: the tile holds whole rows (512 features over the eight waves).  Pass 0 turns the accumulators into the pre-norm values
    // (bias, activation, residual) IN PLACE and sums each row; pass 1 sums the squared deviations (two passes, as torch's LayerNorm: no
    // E[x^2] - mean^2 cancellation); pass 2 normalises and stores.  Row sums: registers -> the two half-waves (one shuffle) -> the eight
    // waves (LDS, 8 floats per row, one barrier per pass).  Every wave of the workgroup reaches both barriers (rows past M compute garbage
    // that is never stored).
    auto epilogue_ln = [&](int j) __attribute__((always_inline)) {
        const int m0 = tile_of(j) * TM;
        AS3 float* red = (AS3 float*)(lds + LN_OFF);
        float s1[RB], mean[RB], rstd[RB];
#pragma unroll
        for (int ri = 0; ri < RB; ++ri) s1[ri] = 0.f;
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // (the column does not depend on the tile in this mode: the compiler hoists the loads of bias / gamma / beta -- 96 registers -- out of the
                // persistent tile loop.  At 64-row tiles they fit and the epilogue is faster for it (measured: 17.2 vs 17.8 ms family time with the
                // loads kept inside); at 96-row tiles they are what spills.)
                const int nb = wf * 64 + fi * 32 + 4 * hi + 8 * g;
                const f32x4 bv = *(const f32x4*)(p.bias + nb);
                f32x4 rv[RB];
#pragma unroll
                for (int ri = 0; ri < RB; ++ri) {
                    const int mc = min(m0 + ri * 32 + l31, p.M - 1);
                    rv[ri] = p.res ? *(const f32x4*)(p.res + (size_t)(p.res_period > 0 ? mc % p.res_period : mc) * p.ldr + nb) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int ri = 0; ri < RB; ++ri)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float t = acc[fi][ri][4 * g + c] * osc + bv[c];
                        t = fmaxf(t, lo);
                        if (sigm) t = 1.0f / (1.0f + expf(-t));
                        t += rv[ri][c];
                        acc[fi][ri][4 * g + c] = t;
                        s1[ri] += t;
                    }
            }
        }
#pragma unroll
        for (int ri = 0; ri < RB; ++ri) {
            s1[ri] += __shfl_xor(s1[ri], 32);
            if (hi == 0) red[(ri * 32 + l31) * 8 + wf] = s1[ri];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ri = 0; ri < RB; ++ri) {
            const f32x4 a = *(const AS3 f32x4*)(red + (ri * 32 + l31) * 8), b = *(const AS3 f32x4*)(red + (ri * 32 + l31) * 8 + 4);
            mean[ri] = (((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]))) * (1.0f / (float)TNK);
            float q = 0.f;
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = acc[fi][ri][r] - mean[ri]; q += d * d; }
            q += __shfl_xor(q, 32);
            if (hi == 0) red[TM * 8 + (ri * 32 + l31) * 8 + wf] = q;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ri = 0; ri < RB; ++ri) {
            const f32x4 a = *(const AS3 f32x4*)(red + TM * 8 + (ri * 32 + l31) * 8), b = *(const AS3 f32x4*)(red + TM * 8 + (ri * 32 + l31) * 8 + 4);
            const float var = (((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]))) * (1.0f / (float)TNK);
            rstd[ri] = 1.0f / sqrtf(var + p.ln_eps);
        }
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = wf * 64 + fi * 32 + 4 * hi + 8 * g;
                const f32x4 gv = *(const f32x4*)(p.ln_g + nb), ev = *(const f32x4*)(p.ln_b + nb);
#pragma unroll
                for (int ri = 0; ri < RB; ++ri) {
                    const int m = m0 + ri * 32 + l31;
                    const bool mok = m < p.M;
                    f32x4 y;
#pragma unroll
                    for (int c = 0; c < 4; ++c) y[c] = (acc[fi][ri][4 * g + c] - mean[ri]) * rstd[ri] * gv[c] + ev[c];
                    if (p.C) {
                        f32x4 out = y;
                        if (p.ln_add) out += *(const f32x4*)(p.ln_add + (size_t)(min(m, p.M - 1) % p.ln_add_period) * TNK + nb);
                        if (mok) *(f32x4*)(p.C + (size_t)m * p.ldc + nb) = out;
                    }
                    if (p.Cs) {
                        const f32x4 x = y * ACT_SCALE;
                        unsigned h0, h1, r0, r1;
                        split2(x[0], x[1], h0, r0);
                        split2(x[2], x[3], h1, r1);
#pragma unroll
                        for (int c = 0; c < 4; ++c) oob |= mok && !(fabsf(x[c]) < F16_LIMIT);
                        const auto sx = __builtin_amdgcn_permlane32_swap(h0, r0, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(h1, r1, false, false);
                        if (mok) *(u32x4*)(p.Cs + (size_t)m * p.ldcs + (size_t)(nb >> 3) * 32 + hi * 16) = u32x4{sx[0], sy[0], sx[1], sy[1]};
                    }
                }
            }
        }
    };

    // ---- pipeline.  Flat slab g (tile g / nk, K slab g % nk) lives in LDS stage g % NST and is consumed as NU units; the
    // fragments of a unit are read while the previous unit's MFMAs run, across slab and tile boundaries.  One barrier per slab,
    // at the start of its LAST unit:
    //   ... unit NU-2 of slab g | 2 MFMAs of unit NU-1 | wait slab g + 1, barrier_g | reads of unit 0 of slab g + 1 | 10 MFMAs ...
    // barrier_g tells a wave that (i) everybody's DMAs of slab g + 1 have landed and (ii) everybody's reads of slab g have
    // completed (lgkmcnt(0) in front of it: every unit of slab g is in registers by then), so its stage can be refilled: the DMA
    // pieces of slab g + NST go out between the MFMAs of the units that follow (three per unit).
    // Mode 2 runs the A operand through registers instead: slab g + 1 is split into its stage before barrier_g (its f32 rows
    // were loaded before barrier_{g-1}).
    // stage offsets as three scalars (an array rotated inside several loops ended up in scratch memory: a scratch load in the K loop is a
    // vector-memory load, and its wait drains the DMAs): so0 = stage of slab g, so1 of g + 1, so2 of g + 2 (three-stage kernels)
    int so0 = 0, so1 = STAGE_BYTES, so2 = (NST - 1) * STAGE_BYTES;
    auto so_last = [&]() { return NST == 3 ? so2 : so1; };
    auto so_at = [&](int i) { return i == 0 ? so0 : (i == 1 ? so1 : so2); };
    issue_tile_setup();
    int ikt_a = 0, ij_a = 0;                                                   // mode 2: (tile, slab) of the f32 rows in the registers
    if (AMODE == 2) {
        load_f32_at(0, 0);
        issue_slab(so0); issue_advance();
        split_store(so0);
        if (n_flat > 1) { ikt_a = nk > 1 ? 1 : 0; ij_a = nk > 1 ? 0 : 1; load_f32_at(ij_a, ikt_a); }
        if (n_flat > 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
        issue_slab(so0); issue_advance();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (AMODE == 3) {
#pragma unroll
            for (int i = 0; i < NA; ++i) { conv_read(i, so0); conv_write(i, so0); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int d = 1; d <= D; ++d) {
        if (d < n_flat) { if (ikt == 0) issue_tile_setup(); issue_slab(so_at(d)); issue_advance(); }
    }
    rd_w(so0, 0, WF[0]);
    rd_a(so0, 0, 0, AF[0]);
    int ckt = 0, cj = 0;
    unsigned long long tr[5] = {0, 0, 0, 0, 0}, tp = 0;
    auto stamp = [&](int seg) {
        if (TRACE) {
            unsigned long long t;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            if (seg >= 0) tr[seg] += t - tp;
            tp = t;
        }
    };
    stamp(-1);
    // Piece schedule: the slab opened by barrier_g (slab g + NST, into the stage of slab g) gets its first FIRST pieces between
    // the MFMAs of the unit that holds the barrier (one after every second MFMA), the rest in the unit that follows.
    constexpr int FIRST = NU == 2 ? 6 : 4;
    bool opened = false;                           // the slab under issue has its first pieces out
    constexpr int BLKC = 4;                        // BLOCKED: K slabs per accumulation block (K = 128); the launcher guarantees nk % BLKC == 0
    // STEADY: the slab is neither the stream's first nor among its last NST, and no ablation switch is set: everything the generic body decides at
    // run time (is there a next slab, does the barrier open a new one, is one under issue) is TRUE at compile time.  The generic body guards every
    // DMA piece with a branch -- 10 to 14 s_cbranch per slab between the MFMAs; the steady body has none but the once-per-tile table setup.
    auto slab = [&](int g, auto fresh_c, auto steady_c) __attribute__((always_inline)) {
        constexpr bool FRESH = decltype(fresh_c)::value;           // BLOCKED: this slab opens an accumulation block
        constexpr bool STEADY = decltype(steady_c)::value;
        const bool more = STEADY || g + 1 < n_flat;
        const bool two_more = STEADY || g + 2 < n_flat;
        const int abl = STEADY ? 0 : p.ablate;
        bool op = STEADY ? true : opened;                          // a slab is under issue (its first pieces went out in the previous slab's last unit)
        stamp(4);
#pragma unroll
        for (int n = 0; n < NU; ++n) {
            const int s = n / U, u = n % U;
            const bool last = n == NU - 1;
            if (n == NU / 2 && (abl & 8) && wave >= 4) __builtin_amdgcn_s_setprio(1);
            if (n == 0 && (abl & 8) && wave >= 4) __builtin_amdgcn_s_setprio(0);
            mm(WF[s & 1], AF[n & 1], u, 0, 2, FRESH && s == 0);
            bool open_now = false;
            // mode 3: slab g + 1's A pieces (this wave's own DMA) are converted during unit CONV_N of slab g, one unit or more ahead of
            // barrier_g.  What may still be in flight behind them: 128-row tiles (three stages): their slab's W pieces and all of slab g + 2;
            // 256-row tiles (two stages): their slab's W pieces (issued during unit 0).
            constexpr int CONV_N = NU == 2 ? 0 : 2;
            const bool conv_now = AMODE == 3 && n == CONV_N && more && !(abl & 16);      // (ablate 16: timing without the in-LDS split, results wrong)
            if (!last) {
                // fragments of the next unit of this slab
                if (u == U - 1) rd_w(so0, s + 1, WF[(s + 1) & 1]);
                rd_a(so0, (n + 1) / U, (n + 1) % U, AF[(n + 1) & 1]);
                if (conv_now && !(abl & 32)) {                                  // (ablate 32: timing without this wait, results wrong)
                    if (D > 1 && two_more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPCL + NW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NW) : "memory");
                }
                if (conv_now && !(abl & 64) && CVS > 1) {
#pragma unroll
                    for (int i = 0; i < NA; ++i) conv_read(i, so1);
                }
            } else {
                if (more) {
                    if (AMODE == 2) {
                        split_store(so1);                      // slab g + 1 (the compiler waits for its loads here)
                        __builtin_amdgcn_sched_barrier(0);       // the split's last use of the registers stays ahead of the loads that refill them
                        if (two_more) {
                            if (++ikt_a == nk) { ikt_a = 0; ++ij_a; }
                            load_f32_at(ij_a, ikt_a);            // slab g + 2
                        }
                        // W of slab g + 1 landed: behind it at most the W pieces of slab g + 2 (D = 2) and the loads just issued
                        if (two_more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((D - 1) * NPCL + NL) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    } else {
                        stamp(0);
                        if (D > 1 && two_more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((D - 1) * NPCL) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        stamp(1);
                        __builtin_amdgcn_s_barrier();
                        stamp(2);
                    }
                    open_now = STEADY || (g + NST < n_flat && !(abl & 1));
                    if (open_now && ikt == 0) issue_tile_setup();
                    rd_w(so1, 0, WF[0]);
                    rd_a(so1, 0, 0, AF[0]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the unit's other MM - 2 MFMAs, with up to NS DMA pieces between them (one after every second MFMA)
            constexpr int NS = MM / 2;
            const int pb = last ? 0 : FIRST + NS * n, pe = last ? min(FIRST, NPCL) : min(NPCL, FIRST + NS * (n + 1));
            const bool issuing = loader && (last ? open_now : op);
#pragma unroll
            for (int k = 0; k < NS - 1; ++k) {
                mm(WF[s & 1], AF[n & 1], u, 2 + 2 * k, 4 + 2 * k, FRESH && s == 0);
                __builtin_amdgcn_sched_barrier(0);
                if (pb + k < pe && issuing) issue_piece(pb + k, last ? so0 : so_last());
                // one piece per MFMA pair: piece k - 1 is converted, then piece k fetched (one pair = 64+ matrix-pipe cycles ahead of its use)
                if (AMODE == 3 && conv_now && !(abl & 64)) {                         // (ablate 64: the wait without the conversion)
                    if (CVS > 1) {
                        if (k >= NS - 1 - NA) conv_write(k - (NS - 1 - NA), so1);           // the unit's last NA pairs
                    } else {
                        if (k >= 1 && k - 1 < NA) conv_write(k - 1, so1);
                        if (k < NA) conv_read(k, so1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (pb + NS - 1 < pe && issuing) issue_piece(pb + NS - 1, last ? so0 : so_last());
            if (!last && op && FIRST + NS * (n + 1) >= NPCL) { issue_advance(); op = false; }
            if (last) op = open_now;
        }
        stamp(3);
        const int t = so0;
        so0 = so1;
        if (NST == 3) { so1 = so2; so2 = t; } else { so1 = t; }
        opened = op;
    };
    const bool steady_ok = p.ablate == 0 && !TRACE && !(AMODE == 3 && BLOCKED);      // (f32 rows + blocked: the steady bodies on top do not fit the registers)
    if constexpr (BLOCKED) {
        // blocks of BLKC slabs: the fold sits behind an inner loop of fixed length, so that it is unconditional code (as a conditional inside
        // the slab loop it made the compiler merge two versions of all 128 accumulator registers: 100 spilled)
        for (int g0 = 0; g0 < n_flat; g0 += BLKC) {
            if (steady_ok && g0 >= BLKC && g0 + BLKC - 1 + NST < n_flat) {
                slab(g0, std::true_type{}, std::true_type{});     // its first MFMAs start the block's sums from zero
#pragma unroll 1
                for (int gi = 1; gi < BLKC; ++gi) slab(g0 + gi, std::false_type{}, std::true_type{});
            } else {
                // the stream's first and last blocks run the generic body; their sums start from explicitly zeroed accumulators (one body fewer
                // to compile into this kernel: with four the gathered and the f32-row variants spilled)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                    for (int ri = 0; ri < RB; ++ri)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[fi][ri][r] = 0.f;
#pragma unroll 1
                for (int gi = 0; gi < BLKC; ++gi) slab(g0 + gi, std::false_type{}, std::false_type{});
            }
            fold_block();
            ckt += BLKC;
            if (ckt == nk) {
                epilogue(cj);
                zero_acc();                                // (tot; acc restarts by itself)
                ckt = 0; ++cj;
            }
        }
    } else {
        // three loops -- the stream's first slab, its steady middle, its last NST slabs -- each with its own copy of the tile-end code (a choice
        // between the two bodies INSIDE one loop makes the compiler merge two versions of the accumulators: hundreds of spilled registers)
#define PS_TILE_END(te) \
        if (te) { if constexpr (LNM) epilogue_ln(cj); else epilogue(cj); zero_acc(); ckt = 0; ++cj; } else { ++ckt; }
        int g = 0;
        if (n_flat > 0) { const bool te = ckt + 1 == nk; slab(g, std::false_type{}, std::false_type{}); PS_TILE_END(te) ++g; }
        if (steady_ok) {
#pragma unroll 1
            for (; g + NST < n_flat; ++g) { const bool te = ckt + 1 == nk; slab(g, std::false_type{}, std::true_type{}); PS_TILE_END(te) }
        }
#pragma unroll 1
        for (; g < n_flat; ++g) { const bool te = ckt + 1 == nk; slab(g, std::false_type{}, std::false_type{}); PS_TILE_END(te) }
#undef PS_TILE_END
    }
    if (TRACE && lane == 0) {
        for (int i = 0; i < 5; ++i) p.trace[((size_t)blockIdx.x * 8 + wave) * 5 + i] = tr[i];
    }
    if ((int)p.whdr[0] == p.wbad_id) oob = true;
    if (AMODE == 2 || AMODE == 3) oob |= !(amax * ACT_SCALE < F16_LIMIT) || !(nanacc == 0.f);
    if (__builtin_amdgcn_ballot_w64(oob) != 0 && lane == 0 && p.range_flag) atomicOr(p.range_flag, 1);
}

// ---- hr8 weight cache: one allocation per (pointer, shape) = [256-byte header | N x K/8 groups of 32 bytes]; re-split when
// the weights epoch differs (linear_internal.h); epoch 0 = split on every launch into a scratch entry
struct WKey {
    const float* w; int n, k, ldw;
    bool operator==(const WKey& o) const { return w == o.w && n == o.n && k == o.k && ldw == o.ldw; }
};
struct WKeyHash { size_t operator()(const WKey& k) const { return std::hash<const void*>()(k.w) ^ ((size_t)k.n * 1315423911u) ^ ((size_t)k.k << 20) ^ (size_t)k.ldw; } };
struct WEntry { char* mem = nullptr; size_t cap = 0; unsigned long long epoch = 0; int bad_id = 0; };
std::unordered_map<WKey, WEntry, WKeyHash> g_ps_cache;
WEntry g_ps_scratch;
std::mutex g_ps_mu;
unsigned char* g_zero_row = nullptr;
size_t g_zero_cap = 0;

int ensure_hr8(hipStream_t s, const float* W, int ldw, int N, int K, WEntry& e, bool fresh) {
    const size_t need = 256 + (size_t)N * K * 4;
    if (need > e.cap) {
        if (e.mem) {
            if (hipStreamSynchronize(s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "gemm_ps: sync failed");
            (void)hipFree(e.mem);
        }
        e.cap = fresh ? need + need / 2 : need;
        if (hipMalloc((void**)&e.mem, e.cap) != hipSuccess) {
            e.mem = nullptr; e.cap = 0;
            return bigc_fail(BIGC_ERR_HIP, "gemm_ps: hipMalloc of the split weights failed");
        }
    }
    if (hipMemsetAsync(e.mem, 0, 256, s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "gemm_ps: memset failed");
    ProfScope ps((void*)s, FAM_LINEAR_AUX, 0.0, 12.0 * N * (double)K);
    static std::atomic<int> bad_seq{0};
    e.bad_id = ++bad_seq;
    const long total = (long)N * (K / 8);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(weight_absmax_kernel, dim3(grid), dim3(256), 0, s, W, ldw, N, K / 4, (unsigned*)e.mem);
    hipLaunchKernelGGL(split_weight_hr8_kernel, dim3(grid), dim3(256), 0, s, W, ldw, (uint4*)(e.mem + 256), N, K / 8, (unsigned*)e.mem, e.bad_id);
    BIGC_LAUNCH_CHECK("split_weight_hr8");
    return BIGC_OK;
}

}  // namespace

void bigc_gemm_ps_cache_forget(const float* w) {
    std::lock_guard<std::mutex> lock(g_ps_mu);
    for (auto it = g_ps_cache.begin(); it != g_ps_cache.end();) {
        if (it->first.w == w) {
            if (it->second.mem) { (void)hipDeviceSynchronize(); (void)hipFree(it->second.mem); }
            it = g_ps_cache.erase(it);
        } else {
            ++it;
        }
    }
}

bool bigc_gemm_ps_supported(const PsGemm& g) {
    if (g.M <= 0 || g.N < 8 || g.K < TK || g.K % TK != 0 || g.N % 8 != 0) return false;
    if (g.a_rowptr) { if (g.a_fmt != PS_F32 || g.a_idx) return false; }
    else if (((uintptr_t)g.A & 15) || g.lda % 16 != 0) return false;
    if (((uintptr_t)g.W & 15) || g.ldw % 4 != 0) return false;
    if (g.bias && ((uintptr_t)g.bias & 15)) return false;
    if (g.res && (((uintptr_t)g.res & 15) || g.ldr % 4 != 0)) return false;
    if (g.C && (((uintptr_t)g.C & 15) || g.ldc % 4 != 0)) return false;
    if (g.Cs && (((uintptr_t)g.Cs & 15) || g.ldcs % 16 != 0)) return false;
    if (g.a_idx && (g.a_fmt != PS_HR8 || g.nseg > 4 || g.nseg < 1 || (g.K / g.nseg) % TK != 0)) return false;
    if ((!g.a_rowptr && (double)g.lda * 256.0 >= 4294967296.0) || (double)g.K * 4.0 * 256.0 >= 4294967296.0) return false;     // 32-bit in-tile offsets
    return g.C || g.Cs;
}

int bigc_gemm_ps_launch(hipStream_t s, const PsGemm& g) {
    if (!bigc_gemm_ps_supported(g)) return bigc_fail(BIGC_ERR_ARG, "gemm_ps: unsupported shape / alignment (M=%d N=%d K=%d)", g.M, g.N, g.K);
    // (the split form holds f16 heads, so this kernel has no bf16 arithmetic; in the literal bf16 mode the model keeps its GEMMs on
    // bigc_linear_f32 -- the GEMM -> GEMM chains are off there, csrc/model.hip -- and a direct call is refused instead of silently running f16)
    if (bigc_linear_get_fast() == 2) return bigc_fail(BIGC_ERR_ARG, "gemm_ps: no bf16 arithmetic on pre-split operands (bigc_linear_set_fast(2)): use bigc_linear_f32");
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "gemm_ps: no device");
        n_cu = prop.multiProcessorCount >= 8 ? (prop.multiProcessorCount / 8) * 8 : 8;
    }
    PsArgs p;
    p.A = (const unsigned char*)g.A; p.lda = g.lda; p.a_idx = g.a_idx; p.nseg = g.a_idx ? g.nseg : 1; p.a_rowptr = g.a_rowptr;
    p.seg_bytes = (g.K / p.nseg) * 4;
    p.bias = g.bias; p.res = g.res; p.ldr = g.ldr; p.res_period = g.res ? g.res_period : 0; p.C = g.C; p.ldc = g.ldc; p.Cs = (unsigned char*)g.Cs; p.ldcs = g.ldcs;
    p.M = g.M; p.N = g.N; p.K = g.K; p.act = g.act;
    const bool lnm = g.ln_g != nullptr;
    if (lnm) {
        BIGC_CHECK_ARG(g.N == 512 && g.a_fmt == PS_HR8 && !g.a_idx && g.ln_b && ((uintptr_t)g.ln_g & 15) == 0 && ((uintptr_t)g.ln_b & 15) == 0 &&
                       (!g.ln_add || (g.ln_add_period > 0 && ((uintptr_t)g.ln_add & 15) == 0)), "gemm_ps: the LayerNorm epilogue needs N = 512, a plain hr8 operand, aligned gamma / beta");
    }
    p.ln_g = g.ln_g; p.ln_b = g.ln_b; p.ln_add = g.ln_add; p.ln_add_period = g.ln_add_period; p.ln_eps = g.ln_eps;
    p.tiles_n = lnm ? 1 : cdiv(g.N, TN);
    // 256-row tiles when they still give every CU at least three tiles (else the 128-row tiles fill the chip better)
    static const int force_rb = getenv("BIGC_PS_RB") ? atoi(getenv("BIGC_PS_RB")) : 0;
    // blocked accumulation: asked for per GEMM (PsGemm.accum == 2: the model's policy, csrc/model.hip), or for every GEMM by BIGC_PS_BLK=1
    // (experiments; BIGC_PS_BLK=0 turns the model's requests off)
    static const int blk_env = getenv("BIGC_PS_BLK") ? atoi(getenv("BIGC_PS_BLK")) : -1;
    const bool want_blk = blk_env > 0 || (blk_env != 0 && g.accum == 2);
    // Rows per tile: the persistent workgroups take ceil(tiles / CUs) rounds, a round costs one tile's time (~ its rows, over an efficiency
    // that falls with the tile: a 128-row tile reads 1.33 x the LDS bytes per MFMA of a 256-row one) -> the cheapest of 256 / 192 / 128 rows.
    // M = 49 152 (decoder, 256 videos) x N = 512: 384 tiles of 256 rows = 2 rounds x 4, 512 of 192 = 2 x 3 (exact), 768 of 128 = 3 x 2.
    // The result does not depend on the choice (one accumulator per element, the same K order in every tiling).
    // BIGC_PS_ACC2 = conv: the gathered GEMM (the temporal conv, K = 3 x 2 dim_enti) on 128-row tiles with a second accumulator; = all: also every
    // plain hr8-operand GEMM with K >= 1024
    // The caller asks for it per GEMM (PsGemm.two_acc: the model does for its K >= 1024 GEMMs -- conv, fc_feat2enti.0 -- so that no GEMM of the
    // path rounds coarser than the two-accumulator kernel of linear_split.hip); BIGC_PS_ACC2 = 0 turns that off, = all extends it to every K >= 1024.
    static const char* acc2_env = getenv("BIGC_PS_ACC2");
    const bool acc2_off = acc2_env && !strcmp(acc2_env, "0");
    const bool acc2_want = g.accum == 1 || (acc2_env && ((g.a_idx && (!strcmp(acc2_env, "conv") || !strcmp(acc2_env, "all"))) || (!g.a_idx && !strcmp(acc2_env, "all") && g.K >= 1024)));
    // f32 rows: by DMA with the split in LDS (mode 3) unless BIGC_PS_F32_DMA=0 (mode 2: through registers); a row-address table needs mode 3
    static const bool f32_dma = !(getenv("BIGC_PS_F32_DMA") && atoi(getenv("BIGC_PS_F32_DMA")) == 0);
    const bool mode3 = g.a_fmt == PS_F32 && (f32_dma || g.a_rowptr) && !lnm;
    const bool acc2 = acc2_want && !acc2_off && !bigc_linear_get_fast() && (g.a_fmt == PS_HR8 || mode3) && !lnm && !want_blk;
    static const int blk_mink_early = getenv("BIGC_PS_BLK_MINK") ? atoi(getenv("BIGC_PS_BLK_MINK")) : 0;
    static const long blk_minm_early = getenv("BIGC_PS_BLK_MINM") ? atol(getenv("BIGC_PS_BLK_MINM")) : 0;
    static const long blk_maxm_early = getenv("BIGC_PS_BLK_MAXM") ? atol(getenv("BIGC_PS_BLK_MAXM")) : 0;
    int rb = 2;
    static const int ln_rb = getenv("BIGC_LN_RB") ? atoi(getenv("BIGC_LN_RB")) : 2;
    if (acc2) rb = 2;
    else if (lnm) rb = ln_rb == 3 ? 3 : 2;            // rows per tile = 32 rb, spanning all 512 features
    else if (want_blk && (g.a_fmt == PS_HR8 || mode3) && (g.K / TK) % 4 == 0 && g.K >= blk_mink_early && g.M >= blk_minm_early && (blk_maxm_early == 0 || g.M <= blk_maxm_early)) rb = 2;
    else if (force_rb) rb = force_rb;
    else {
        static const bool rb3_off = getenv("BIGC_PS_RB3") && atoi(getenv("BIGC_PS_RB3")) == 0;
        double best = 0.0;
        const int cand[3] = {4, 3, 2};
        const double eff[3] = {1.0, 0.94, 0.82};
        for (int i = 0; i < 3; ++i) {
            if (cand[i] == 3 && (rb3_off || g.a_fmt == PS_F32)) continue;            // (f32 rows: 128- or 256-row tiles)
            const long tiles = (long)cdiv(g.M, 64 * cand[i]) * p.tiles_n;
            const double cost = (double)((tiles + n_cu - 1) / n_cu) * cand[i] / eff[i];
            if (best == 0.0 || cost < best * 0.999) { best = cost; rb = cand[i]; }
        }
    }
    p.tiles_m = lnm ? cdiv(g.M, 32 * rb) : cdiv(g.M, 64 * rb);
    p.range_flag = g.range_flag;
    // opt-in blocked accumulation (BIGC_PS_BLK = K slabs per block, a power of two): 128-row tiles, plain / gathered hr8 operand
    const int force_blk = want_blk ? 1 : 0;
    // (experiments: BIGC_PS_BLK_MINK / _MINM / _MAXM restrict the blocked kernel to GEMMs with K >= / M >= / M <= the given value)
    static const int blk_mink = getenv("BIGC_PS_BLK_MINK") ? atoi(getenv("BIGC_PS_BLK_MINK")) : 0;
    static const long blk_minm = getenv("BIGC_PS_BLK_MINM") ? atol(getenv("BIGC_PS_BLK_MINM")) : 0;
    static const long blk_maxm = getenv("BIGC_PS_BLK_MAXM") ? atol(getenv("BIGC_PS_BLK_MAXM")) : 0;
    const bool blk_here = g.K >= blk_mink && g.M >= blk_minm && (blk_maxm == 0 || g.M <= blk_maxm);
    p.blk = (force_blk > 0 && (g.a_fmt == PS_HR8 || mode3) && !bigc_linear_get_fast() && !lnm && blk_here && (g.K / TK) % 4 == 0) ? force_blk : 0;
    static const int ablate = getenv("BIGC_PS_ABLATE") ? atoi(getenv("BIGC_PS_ABLATE")) : 0;
    p.ablate = ablate;
    static const int trace = getenv("BIGC_PS_TRACE") ? atoi(getenv("BIGC_PS_TRACE")) : 0;
    p.trace = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_ps_mu);
        const unsigned long long epoch = bigc_static_weights_epoch;
        WEntry* e = &g_ps_scratch;
        bool stale = true;
        if (epoch != 0) {
            e = &g_ps_cache[WKey{g.W, g.N, g.K, g.ldw}];
            stale = e->epoch != epoch;
        }
        if (stale) {
            int rc = ensure_hr8(s, g.W, g.ldw, g.N, g.K, *e, epoch == 0);
            if (rc != BIGC_OK) return rc;
            e->epoch = epoch;
        }
        p.Ws = (const unsigned char*)(e->mem + 256); p.ldw = (long)g.K * 4; p.whdr = (const unsigned*)e->mem; p.wbad_id = e->bad_id;
        p.zero_row = nullptr;
        {
            const size_t need = (size_t)p.seg_bytes > (size_t)g.N * 4 ? (size_t)p.seg_bytes : (size_t)g.N * 4;
            if (need > g_zero_cap) {
                if (g_zero_row) { if (hipStreamSynchronize(s) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "gemm_ps: sync failed"); (void)hipFree(g_zero_row); }
                g_zero_cap = need < 16384 ? 16384 : need;
                if (hipMalloc((void**)&g_zero_row, g_zero_cap) != hipSuccess || hipMemsetAsync(g_zero_row, 0, g_zero_cap, s) != hipSuccess) {
                    g_zero_row = nullptr; g_zero_cap = 0;
                    return bigc_fail(BIGC_ERR_HIP, "gemm_ps: cannot allocate the zero row");
                }
            }
            p.zero_row = g_zero_row;
        }
        if (!p.bias) p.bias = (const float*)g_zero_row;
    }
    const double flops = 2.0 * g.M * (double)g.N * g.K;
    const double bytes = 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * ((g.res && g.res_period <= 0 ? 1 : 0) + (g.C ? 1 : 0) + (g.Cs ? 1 : 0)));
    ProfScope ps((void*)s, FAM_LINEAR, flops, bytes);
    // persistent workgroups: one per CU (a multiple of 8 for the XCD-aware tile order), fewer when there are fewer tiles
    const int n_tiles = p.tiles_m * p.tiles_n;
    const int grid = n_tiles >= n_cu ? n_cu : cdiv(n_tiles, 8) * 8;
#define PS_LAUNCH(MODE, RBV) hipLaunchKernelGGL((gemm_ps_kernel<MODE, RBV>), dim3(grid), dim3(512), 0, s, p)
    if (acc2) {
        if (mode3) hipLaunchKernelGGL((gemm_ps_kernel<3, 2, false, false, false, false, true>), dim3(grid), dim3(512), 0, s, p);
        else if (g.a_idx) hipLaunchKernelGGL((gemm_ps_kernel<1, 2, false, false, false, false, true>), dim3(grid), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_ps_kernel<0, 2, false, false, false, false, true>), dim3(grid), dim3(512), 0, s, p);
    } else if (lnm) {
        if (rb == 3) {
            if (bigc_linear_get_fast()) hipLaunchKernelGGL((gemm_ps_kernel<0, 3, false, true, false, true>), dim3(grid), dim3(512), 0, s, p);
            else hipLaunchKernelGGL((gemm_ps_kernel<0, 3, false, false, false, true>), dim3(grid), dim3(512), 0, s, p);
        } else {
            if (bigc_linear_get_fast()) hipLaunchKernelGGL((gemm_ps_kernel<0, 2, false, true, false, true>), dim3(grid), dim3(512), 0, s, p);
            else hipLaunchKernelGGL((gemm_ps_kernel<0, 2, false, false, false, true>), dim3(grid), dim3(512), 0, s, p);
        }
    } else if (trace && g.a_fmt == PS_HR8 && !g.a_idx) {
        static unsigned long long* buf = nullptr;
        if (!buf && hipMalloc((void**)&buf, 8 * 5 * 8 * 4096) != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "gemm_ps: trace buffer");
        p.trace = buf;
        if (rb == 4) hipLaunchKernelGGL((gemm_ps_kernel<0, 4, true>), dim3(grid), dim3(512), 0, s, p);
        else if (rb == 3) hipLaunchKernelGGL((gemm_ps_kernel<0, 3, true>), dim3(grid), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_ps_kernel<0, 2, true>), dim3(grid), dim3(512), 0, s, p);
        std::vector<unsigned long long> h((size_t)grid * 40);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
        double seg[5] = {0, 0, 0, 0, 0};
        for (size_t i = 0; i < h.size(); ++i) seg[i % 5] += (double)h[i];
        const double slabs = (double)p.tiles_m * p.tiles_n * (p.K / TK) * 8;     // wave-slabs
        fprintf(stderr, "gemm_ps trace M=%d N=%d K=%d rb=%d (s_memtime ticks per wave per slab): step-0 MFMAs + reads %.0f | wait vmcnt/lgkm %.0f | "
                "barrier %.0f | step-1 MFMAs + reads + DMA pieces %.0f | loop end + epilogue %.0f\n", p.M, p.N, p.K, rb, seg[0] / slabs,
                seg[1] / slabs, seg[2] / slabs, seg[3] / slabs, seg[4] / slabs);
    } else if (bigc_linear_get_fast()) {
#define PS_LAUNCH_FAST(MODE, RBV) hipLaunchKernelGGL((gemm_ps_kernel<MODE, RBV, false, true>), dim3(grid), dim3(512), 0, s, p)
        if (mode3) { if (rb == 4) PS_LAUNCH_FAST(3, 4); else PS_LAUNCH_FAST(3, 2); }
        else if (g.a_fmt == PS_F32) { if (rb == 4) PS_LAUNCH_FAST(2, 4); else PS_LAUNCH_FAST(2, 2); }
        else if (g.a_idx) { if (rb == 4) PS_LAUNCH_FAST(1, 4); else if (rb == 3) PS_LAUNCH_FAST(1, 3); else PS_LAUNCH_FAST(1, 2); }
        else { if (rb == 4) PS_LAUNCH_FAST(0, 4); else if (rb == 3) PS_LAUNCH_FAST(0, 3); else PS_LAUNCH_FAST(0, 2); }
#undef PS_LAUNCH_FAST
    } else if (p.blk > 0) {
        // (the four-wave form -- gemm_ps_kernel<MODE, 4, false, false, true, false, false, 4>, 256 threads, same tile, same results -- was built and
        // measured: correct, and SLOWER than two waves per SIMD on every shape: conv GEMM 5.15 vs 4.68 ms, K = 512 at M = 579 295 1.25 vs 1.22 ms,
        // decoder 512 x 512 110 vs 101 us.  With one wave per SIMD nothing covers the wave's own waits; not instantiated.)
        if (mode3) hipLaunchKernelGGL((gemm_ps_kernel<3, 2, false, false, true>), dim3(grid), dim3(512), 0, s, p);
        else if (g.a_idx) hipLaunchKernelGGL((gemm_ps_kernel<1, 2, false, false, true>), dim3(grid), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_ps_kernel<0, 2, false, false, true>), dim3(grid), dim3(512), 0, s, p);
    } else if (mode3) {
        if (rb == 4) PS_LAUNCH(3, 4); else PS_LAUNCH(3, 2);        // (f32 operand: force_rb 3 falls to 128 rows)
    } else if (g.a_fmt == PS_F32) {
        if (rb == 4) PS_LAUNCH(2, 4); else PS_LAUNCH(2, 2);
    } else if (g.a_idx) {
        if (rb == 4) PS_LAUNCH(1, 4); else if (rb == 3) PS_LAUNCH(1, 3); else PS_LAUNCH(1, 2);
    } else {
        if (rb == 4) PS_LAUNCH(0, 4); else if (rb == 3) PS_LAUNCH(0, 3); else PS_LAUNCH(0, 2);
    }
#undef PS_LAUNCH
    BIGC_LAUNCH_CHECK("gemm_ps");
    return BIGC_OK;
}

int bigc_split_rows_hr8(hipStream_t s, const float* X, long ldx, void* Y, long ldy_bytes, long rows, int K, int* range_flag) {
    BIGC_CHECK_ARG(X && Y && K % 8 == 0 && ldx % 4 == 0 && ldy_bytes % 16 == 0, "split_rows: bad argument");
    if (rows <= 0) return BIGC_OK;
    ProfScope ps((void*)s, FAM_LINEAR_AUX, 0.0, 8.0 * rows * (double)K);
    const long total = rows * (K / 8);
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(split_rows_hr8_kernel, dim3(grid), dim3(256), 0, s, X, ldx, (unsigned char*)Y, ldy_bytes, rows, K / 8, range_flag);
    BIGC_LAUNCH_CHECK("split_rows_hr8");
    return BIGC_OK;
}

// Raw entry point (tests, benchmarks): A is f32 (a_fmt 0) or hr8 (a_fmt 1) with pitch lda_bytes; outputs f32 C and / or hr8 Cs
extern "C" int bigc_gemm_ps(void* stream, const void* A, long lda_bytes, int a_fmt, const int32_t* a_idx, int nseg, const float* W, int ldw,
                            const float* bias, const float* residual, long ldr, float* C, long ldc, void* Cs, long ldcs_bytes,
                            int M, int N, int K, int act, int* range_flag) {
    BIGC_CHECK_ARG(A && W && (C || Cs), "gemm_ps: null pointer");
    if (M == 0) return BIGC_OK;
    PsGemm g;
    g.A = A; g.lda = lda_bytes; g.a_fmt = a_fmt ? PS_HR8 : PS_F32; g.a_idx = a_idx; g.nseg = a_idx ? nseg : 1; g.W = W; g.ldw = ldw;
    g.bias = bias; g.res = residual; g.ldr = ldr; g.C = C; g.ldc = ldc; g.Cs = Cs; g.ldcs = ldcs_bytes; g.M = M; g.N = N; g.K = K; g.act = act;
    g.range_flag = range_flag;
    return bigc_gemm_ps_launch((hipStream_t)stream, g);
}

// A given by row addresses (f32 rows read in place: DMA + split in LDS)
extern "C" int bigc_gemm_ps_rows(void* stream, const uint64_t* row_ptr, const float* W, int ldw, const float* bias, float* C, long ldc, void* Cs,
                                 long ldcs_bytes, int M, int N, int K, int act, int two_acc, int* range_flag) {
    BIGC_CHECK_ARG(row_ptr && W && (C || Cs), "gemm_ps_rows: null pointer");
    if (M == 0) return BIGC_OK;
    PsGemm g;
    g.a_rowptr = (const unsigned long long*)row_ptr; g.a_fmt = PS_F32; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.Cs = Cs; g.ldcs = ldcs_bytes;
    g.M = M; g.N = N; g.K = K; g.act = act; g.range_flag = range_flag; g.accum = two_acc;            // 0 / 1 / 2 (PsGemm.accum)
    return bigc_gemm_ps_launch((hipStream_t)stream, g);
}

// LayerNorm(act(A W^T + bias) + residual) * gamma + beta (+ add[m % add_period]) in one launch (N = 512; A in the split form)
extern "C" int bigc_gemm_ps_ln(void* stream, const void* A, long lda_bytes, const float* W, int ldw, const float* bias, const float* residual, long ldr,
                               const float* gamma, const float* beta, float eps, const float* add, int add_period, float* C, long ldc, void* Cs,
                               long ldcs_bytes, int M, int N, int K, int act, int* range_flag) {
    BIGC_CHECK_ARG(A && W && gamma && beta && (C || Cs), "gemm_ps_ln: null pointer");
    if (M == 0) return BIGC_OK;
    PsGemm g;
    g.A = A; g.lda = lda_bytes; g.a_fmt = PS_HR8; g.W = W; g.ldw = ldw; g.bias = bias; g.res = residual; g.ldr = ldr; g.C = C; g.ldc = ldc; g.Cs = Cs;
    g.ldcs = ldcs_bytes; g.M = M; g.N = N; g.K = K; g.act = act; g.range_flag = range_flag;
    g.ln_g = gamma; g.ln_b = beta; g.ln_eps = eps; g.ln_add = add; g.ln_add_period = add_period;
    return bigc_gemm_ps_launch((hipStream_t)stream, g);
}

extern "C" int bigc_split_rows_f32(void* stream, const float* X, long ldx, void* Y, long ldy_bytes, long rows, int K, int* range_flag) {
    return bigc_split_rows_hr8((hipStream_t)stream, X, ldx, Y, ldy_bytes, rows, K, range_flag);
}
