// How v_mfma_f32_32x32x16_f16 rounds its f32 accumulation (gfx950): D = A B + C for C = 2^23 (ulp 1) and products that sum to a
// fraction of an ulp.  Round-to-nearest-even gives 2^23 + 1 for a sum of 0.75, 2^23 for 0.25 and the even neighbour for 0.5;
// truncation gives 2^23 throughout.  Several products per instruction show whether the 16 products are summed exactly before
// the one rounding, or rounded on the way.  Also a chain: 2^23 + 4096 x 0.25 through 4096 dependent instructions.
//   hipcc --offload-arch=gfx950 -O2 benchmarks/probes/mfma_round.hip -o /tmp/mfma_round && /tmp/mfma_round
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A (32 x 16): lane l holds row l % 32, k = 8 (l / 32) .. + 7;  B (16 x 32): lane l holds column l % 32, same k
__global__ void probe(const float* xs, int nx, float c0, float* out, int chain) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    if (lane % 32 == 0) for (int j = 0; j < 8; ++j) { const int k = 8 * (lane / 32) + j; a[j] = (_Float16)(k < nx ? xs[k] : 0.f); b[j] = (_Float16)1.f; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = c0;
    for (int i = 0; i < chain; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
}

int main() {
    float *dx, *dout;
    (void)hipMalloc(&dx, 64); (void)hipMalloc(&dout, 4);
    struct { const char* name; int n; float x[16]; float c0; int chain; } t[] = {
        {"2^23 + 0.75 (one product)", 1, {0.75f}, 8388608.f, 1},
        {"2^23 + 0.25", 1, {0.25f}, 8388608.f, 1},
        {"2^23 + 0.5 (tie, even below)", 1, {0.5f}, 8388608.f, 1},
        {"2^23+1 + 0.5 (tie, even above)", 1, {0.5f}, 8388609.f, 1},
        {"2^23 + 3 x 0.25 in one instruction", 3, {0.25f, 0.25f, 0.25f}, 8388608.f, 1},
        {"2^23 + 16 x 0.0625 (= 1.0) in one instruction", 16, {.0625f,.0625f,.0625f,.0625f,.0625f,.0625f,.0625f,.0625f,.0625f,.0625f,.0625f,.0625f,.0625f,.0625f,.0625f,.0625f}, 8388608.f, 1},
        {"2^23 + 0.75 - 0.5 + 0.5 in one instruction", 3, {0.75f, -0.5f, 0.5f}, 8388608.f, 1},
        {"-(2^23) - 0.75", 1, {-0.75f}, -8388608.f, 1},
        {"chain: 2^23 + 4096 x 0.75 (RNE: +4096; truncation: +0)", 1, {0.75f}, 8388608.f, 4096},
        {"chain: 2^23 + 4096 x 0.25 (RNE: +0)", 1, {0.25f}, 8388608.f, 4096},
        {"2^24 + 1.5 (ulp 2)", 1, {1.5f}, 16777216.f, 1},
        {"2^24 + 0.99", 1, {0.99f}, 16777216.f, 1},
    };
    for (auto& q : t) {
        (void)hipMemcpy(dx, q.x, 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dx, q.n, q.c0, dout, q.chain);
        float r; (void)hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
        printf("%-55s -> c0 %+g\n", q.name, (double)r - (double)q.c0);
    }
    return 0;
}
