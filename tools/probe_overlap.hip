// Does vector work hide under MFMAs on gfx950?  One workgroup per CU, W waves per SIMD, a loop of
//   (a) 16 independent 32x32x16 bf16 MFMAs, (b) NV vector ops (fma + exp mix), (c) both in one basic block,
// timed per iteration with s_memtime-free wall clock (hipEvent) over many iterations.
//   hipcc --offload-arch=gfx950 -O3 -o probe_overlap tools/probe_overlap.hip && ./probe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int NV>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  float v[32];
  for (int e = 0; e < 32; ++e) v[e] = threadIdx.x * 0.01f + e;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    if (MODE & 2) {
#pragma unroll
      for (int j = 0; j < NV / 32; ++j)
#pragma unroll
        for (int e = 0; e < 32; ++e) v[e] = (j & 1) ? __builtin_amdgcn_exp2f(v[e] * 0.001f) : __builtin_fmaf(v[e], 1.0001f, 0.5f);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int e = 0; e < 32; ++e) s += v[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV> float run(int waves_per_simd, float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256), block(256 * waves_per_simd);
  hipLaunchKernelGGL((probe<MODE, NV>), grid, block, 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, NV>), grid, block, 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters;   // ns per iteration
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  for (int w = 1; w <= 2; ++w) {
    printf("waves/SIMD %d: 16 MFMA %.0f ns | 64 VALU %.0f ns | both %.0f ns || 128 VALU %.0f ns | 16 MFMA + 128 VALU %.0f ns || 256 VALU %.0f | 16 MFMA + 256 VALU %.0f ns\n", w,
           run<1, 64>(w, out, iters), run<2, 64>(w, out, iters), run<3, 64>(w, out, iters), run<2, 128>(w, out, iters), run<3, 128>(w, out, iters),
           run<2, 256>(w, out, iters), run<3, 256>(w, out, iters));
  }
  return 0;
}
