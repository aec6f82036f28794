// What the f16 matrix pipe of THIS chip sustains with nothing else in the way: a register-only loop of independent
// v_mfma_f32_32x32x16_f16 (no LDS, no memory), two waves per SIMD on every CU, for ~2 ms per measurement.
//   hipcc --offload-arch=gfx950 -O3 benchmarks/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Three operand patterns: zeros, random f16 in [-1, 1) (what a GEMM sees), and the split-operand pattern of linear_split /
// gemm_ps (heads of O(1), residuals of O(2^-11)).  Output: TFLOP/s of f16 MFMA work, the effective shader clock (MFMA issue
// slots / wall time: 32 cycles per instruction and SIMD) and the fraction of the 2.5 PFLOP/s dense peak.  The chip clocks to its
// power budget (MI355X_MICROARCH.md, "DVFS give-back"): this number, not 2.5 PFLOP/s, is what a kernel that keeps the matrix
// pipe 100 % busy would reach, and a third of it is the ceiling of the three-MFMA f32-accurate GEMM.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void mfma_loop(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    f16x8 a0 = in[(size_t)t * 4 + 0], a1 = in[(size_t)t * 4 + 1], b0 = in[(size_t)t * 4 + 2], b1 = in[(size_t)t * 4 + 3];
    f32x16 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, c[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += c[i][r];
    out[t] = s;
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int n_cu = prop.multiProcessorCount, n_thr = n_cu * 512;
    std::vector<_Float16> h((size_t)n_thr * 32);
    f16x8* d_in; float* d_out;
    (void)hipMalloc(&d_in, h.size() * 2); (void)hipMalloc(&d_out, (size_t)n_thr * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[3] = {"zeros", "random f16 in [-1, 1)", "split-operand pattern (head O(1), residual O(2^-11))"};
    printf("%s, %d CUs, nominal %d MHz\n", prop.name, n_cu, prop.clockRate / 1000);
    for (int pat = 0; pat < 3; ++pat) {
        srand(1);
        for (size_t i = 0; i < h.size(); ++i) {
            const float u = 2.0f * rand() / (float)RAND_MAX - 1.0f;
            const bool resid = ((i / 8) & 1) != 0;           // a1 / b1 play the residuals in pattern 2
            h[i] = (_Float16)(pat == 0 ? 0.f : (pat == 2 && resid ? u / 2048.0f : u));
        }
        (void)hipMemcpy(d_in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        const int iters = 20000;
        hipLaunchKernelGGL(mfma_loop, dim3(n_cu), dim3(512), 0, 0, d_in, d_out, 2000);        // warm-up
        (void)hipDeviceSynchronize();
        double best = 0.0, sum = 0.0;
        const int reps = 5;
        for (int r = 0; r < reps; ++r) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_loop, dim3(n_cu), dim3(512), 0, 0, d_in, d_out, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)n_cu * 8 /*waves*/ * iters * 8 /*mfma*/ * 2.0 * 32 * 32 * 16;
            const double tf = flops / (ms * 1e-3) / 1e12;
            sum += tf; if (tf > best) best = tf;
        }
        const double tf = sum / reps;
        // per SIMD: 2 waves x iters x 8 MFMAs x 32 cycles back to back
        const double ghz = tf * 1e12 / (2.0 * 32 * 32 * 16) * 32.0 / ((double)n_cu * 4) / 1e9;
        printf("%-56s %8.1f TFLOP/s (best %8.1f)  = %.3f of 2500  effective matrix-pipe clock %.2f GHz  -> three-MFMA GEMM ceiling %.0f TFLOP/s\n",
               names[pat], tf, best, tf / 2500.0, ghz, tf / 3.0);
    }
    return 0;
}
