// Does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs on gfx950, or flush them to zero?  (The two-term fp16 split keeps its
// low term out of the subnormals by scaling it, but the HIGH term of an operand below 6.1e-5 is a subnormal fp16.)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/diag/mfma_f16_denorm.hip -o /tmp/mfma_denorm && /tmp/mfma_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
  h16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
  float* d; hipMalloc(&d, 8);
  const float cases[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1024.0f}, {5.96e-8f, 1024.0f}, {6.2e-5f, 1024.0f}, {1024.0f, 3.0e-5f}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.4g (as fp16: %.6g)  b = %.4g : mfma sum over k=16 = %.6g  (expected %.6g if subnormals are honoured)\n", c[0], h[1], c[1], h[0],
           16.0 * (double)(float)(_Float16)c[0] * (double)(float)(_Float16)c[1]);
  }
  return 0;
}
