#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a_val, float b_val) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    a[0] = (_Float16)a_val; b[0] = (_Float16)b_val;      // k index 0 (h=0) / 8 (h=1) of every row/col
    f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    float tests[][2] = {{1e-6f, 1024.f}, {3e-5f, 1024.f}, {6.2e-5f, 1024.f}, {1024.f, 1e-6f}, {1e-6f, 1e-6f}, {5.96e-8f, 65504.f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  mfma c[0]=%.9g  expect(2 terms)=%.9g\n", t[0], t[1], h, 2.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1]);
    }
    return 0;
}
