// Fused global-norm clip + Adam over one flat parameter buffer (gfx950).
//
// Replaces torch.nn.utils.clip_grad_norm_(model.parameters(), 1.) + torch.optim.Adam.step() +
// optimizer.zero_grad() of the reference training loop (train.py:55,95-97): ~30 small torch
// kernels per parameter tensor become two launches, the clip coefficient never visits the host.
// HBM-bound: 4 arrays read + 3 written per element (28 B / parameter).
#include <algorithm>
#include "pfn_device.h"
#include "pfn_kernels.h"

namespace pfn {

constexpr int NORM_BLOCKS = 1024;
constexpr int SKIPPED_AT = 1 + NORM_BLOCKS;      // scratch[SKIPPED_AT]: number of skipped (non-finite gradient) steps since the caller zeroed the scratch

__global__ __launch_bounds__(256) void grad_sqsum_kernel(const float* g, long n4, float gscale, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float x = v[e] * gscale; s += x * x; }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void clip_adam_kernel(AdamArgs a, long n4, int nparts, int zero_grad) {
  __shared__ float s_coef;
  __shared__ int s_skip;
  {
    // every block re-reduces the (<= 1024) partials in the same order: deterministic, no atomics
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += a.scratch[1 + i];
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
      // clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
      float coef = a.max_norm > 0.f ? a.max_norm / (norm + 1e-6f) : 1.f;
      s_coef = fminf(coef, 1.f) * a.grad_scale;
      // A gradient with an inf / NaN in it (an overflow the fp16 backward's saturation did not catch, a bad batch) would turn every parameter into NaN through
      // the clip coefficient: the step is SKIPPED instead -- parameters and moments untouched, the gradient cleared as asked -- and counted in
      // scratch[SKIPPED_AT] (what torch.cuda.amp.GradScaler.step does with its found_inf flag; the reference itself has no such guard).
      s_skip = !(norm <= 3.0e38f);
      if (blockIdx.x == 0) {
        a.scratch[0] = norm;
        if (s_skip) a.scratch[SKIPPED_AT] += 1.f;
      }
    }
    __syncthreads();
  }
  const float coef = s_coef;
  if (s_skip) {
    if (zero_grad)
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) *reinterpret_cast<f32x4*>(a.g + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  const float bc1 = 1.f - powf(a.beta1, (float)a.step);
  const float bc2 = 1.f - powf(a.beta2, (float)a.step);
  const float step_size = a.lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 p = *reinterpret_cast<f32x4*>(a.p + 4 * i);
    f32x4 g = *reinterpret_cast<f32x4*>(a.g + 4 * i);
    f32x4 m = *reinterpret_cast<f32x4*>(a.m + 4 * i);
    f32x4 v = *reinterpret_cast<f32x4*>(a.v + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = g[e] * coef;
      m[e] = a.beta1 * m[e] + (1.f - a.beta1) * gg;
      v[e] = a.beta2 * v[e] + (1.f - a.beta2) * gg * gg;
      const float denom = sqrtf(v[e]) * inv_sqrt_bc2 + a.eps;
      p[e] -= step_size * m[e] / denom;
    }
    *reinterpret_cast<f32x4*>(a.p + 4 * i) = p;
    *reinterpret_cast<f32x4*>(a.m + 4 * i) = m;
    *reinterpret_cast<f32x4*>(a.v + 4 * i) = v;
    if (zero_grad) *reinterpret_cast<f32x4*>(a.g + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

int launch_clip_adam(const AdamArgs& a, hipStream_t s) {
  if (a.n % 4) return PFN_ERR_ALIGNMENT;
  const long n4 = a.n / 4;
  if (n4 == 0) return PFN_OK;
  const int nparts = (int)std::max<long>(1, std::min<long>((n4 + 255) / 256, NORM_BLOCKS));
  hipLaunchKernelGGL(grad_sqsum_kernel, dim3(nparts), dim3(256), 0, s, a.g, n4, a.grad_scale, a.scratch + 1);
  const int grid = (int)std::max<long>(1, std::min<long>((n4 + 255) / 256, 2048));
  hipLaunchKernelGGL(clip_adam_kernel, dim3(grid), dim3(256), 0, s, a, n4, nparts, a.zero_grad);
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

}  // namespace pfn
