// Probe (round 6): fp16 conversion overflow behaviour and MFMA subnormal handling on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_fp16 tools/probe_fp16.hip && /tmp/probe_fp16
// (1) v_cvt_pk_f16_f32 of values beyond 65504 with MODE.FP16_OVFL (bit 23) clear / set: inf vs clamp to +-65504?
// (2) v_mfma_f32_32x32x16_f16 with subnormal fp16 inputs: kept or flushed?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void cvt_kernel(const float* in, f16* out_plain, f16* out_ovfl, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  out_plain[i] = (f16)in[i];
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  float v = in[i];
  asm volatile("" : "+v"(v));
  f16x2 p;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(p) : "v"(v));
  out_ovfl[i] = p[0];
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
}
__global__ void mfma_kernel(float a_val, float b_val, float* out) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (f16)a_val; b[e] = (f16)b_val; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  const int n = 8;
  float h[n] = {1.f, 65504.f, 65520.f, 70000.f, 1e6f, -1e6f, 3e-8f, 6.0e-5f};
  float* d; f16 *o1, *o2;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o1, n * 2); hipMalloc(&o2, n * 2);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, d, o1, o2, n);
  f16 r1[n], r2[n];
  hipMemcpy(r1, o1, n * 2, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, n * 2, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt %12g -> plain %12g   FP16_OVFL=1 %12g\n", h[i], (float)r1[i], (float)r2[i]);
  float* out; hipMalloc(&out, 4);
  const float sub = 3.0e-6f;      // subnormal in fp16 (below 6.1e-5)
  hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, sub, 1024.f, out);
  float r; hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
  printf("mfma f16: 16 x (%g as fp16 = %g) x 1024 = %g   (kept: %g, flushed: 0)\n", sub, (float)(f16)sub, r, 16.f * (float)(f16)sub * 1024.f);
  return 0;
}
