#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float av, float bv) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    float vals[][2] = {{1.0f, 1.0f}, {9.5367431640625e-07f /*2^-20*/, 1.0f}, {5.9604644775390625e-08f /*2^-24*/, 1.0f}, {9.5367431640625e-07f, 9.5367431640625e-07f}, {6.103515625e-05f, 0.5f}};
    for (auto& v : vals) {
        k<<<1, 64>>>(d, v[0], v[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g -> mfma sum over k=16: %g (expected %g)\n", v[0], v[1], h, 16.0 * (double)(_Float16)v[0] * (double)(_Float16)v[1]);
    }
    return 0;
}
