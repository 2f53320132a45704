// Fused bar-distribution ("Riemann distribution") loss kernels (gfx950).
//
// Replaces BarDistribution.forward / FullSupportBarDistribution.forward and .mean
// (reference bar_distribution.py:19-38, 83-117): bucket search, log-softmax, width scaling,
// gather, half-normal tail correction -- one pass over each logits row, one wave per row.
// HBM-bound: algorithmic traffic is one read of the logits row (+ one write in the backward).
#include <algorithm>
#include "pfn_device.h"
#include "pfn_kernels.h"

namespace pfn {

constexpr float HALFNORMAL_ICDF_HALF = 0.6744897501960817f;  // HalfNormal(1).icdf(0.5), bar_distribution.py:85-87
constexpr float LOG_2 = 0.6931471805599453f;
constexpr float HALF_LOG_2PI = 0.9189385332046727f;
constexpr float SQRT_2_OVER_PI = 0.7978845608028654f;

// torch.searchsorted(borders, y) - 1 with the two edge fixes of map_to_bucket_idx
// (bar_distribution.py:19-23): count of borders strictly below y, minus one.
PFN_DEV int bucket_of(const float* borders, int nbars, float y) {
  int lo = 0, hi = nbars + 1;  // first index with borders[idx] >= y
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (borders[mid] < y) lo = mid + 1; else hi = mid;
  }
  int t = lo - 1;
  if (y == borders[0]) t = 0;
  if (y == borders[nbars]) t = nbars - 1;
  return t;
}

PFN_DEV float halfnormal_logprob(float scale, float v) {
  const float z = v / scale;
  return LOG_2 - __logf(scale) - HALF_LOG_2PI - 0.5f * z * z;
}

PFN_DEV void row_max_sumexp(const float* row, int n, int lane, float& mx, float& se) {
  mx = -INFINITY;
  const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
  const int n4 = vec ? (n / 4) : 0;
  for (int i = lane; i < n4; i += 64) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * i);
    mx = fmaxf(mx, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
  }
  for (int i = n4 * 4 + lane; i < n; i += 64) mx = fmaxf(mx, row[i]);
  mx = wave_max(mx);
  se = 0.f;
  for (int i = lane; i < n4; i += 64) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * i);
    se += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx);
  }
  for (int i = n4 * 4 + lane; i < n; i += 64) se += __expf(row[i] - mx);
  se = wave_sum(se);
}

__global__ __launch_bounds__(256) void bar_nll_fwd_kernel(BarArgs a) {
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < a.R; r += (long)gridDim.x * 4) {
    const float* row = a.logits + r * a.ld;
    float mx, se;
    row_max_sumexp(row, a.nbars, lane, mx, se);
    if (lane == 0) {
      const float lse = mx + __logf(se);
      const float y = a.y[r];
      int t = bucket_of(a.borders, a.nbars, y);
      const bool in_support = (t >= 0 && t < a.nbars);
      t = max(0, min(a.nbars - 1, t));
      const float w = a.borders[t + 1] - a.borders[t];
      float lp = row[t] - lse - __logf(w);
      if (a.full_support) {
        if (t == 0) {
          const float w0 = a.borders[1] - a.borders[0];
          lp += halfnormal_logprob(w0 / HALFNORMAL_ICDF_HALF, fmaxf(a.borders[1] - y, 1e-8f)) + __logf(w0);
        }
        if (t == a.nbars - 1) {
          const float w1 = a.borders[a.nbars] - a.borders[a.nbars - 1];
          lp += halfnormal_logprob(w1 / HALFNORMAL_ICDF_HALF, y - a.borders[a.nbars - 1]) + __logf(w1);
        }
      } else if (!in_support) {
        lp = __builtin_nanf("");  // the reference asserts here (bar_distribution.py:27)
      }
      a.nll[r] = -lp;
      a.lse[r] = lse;
      a.bucket[r] = t;
    }
  }
}

__global__ __launch_bounds__(256) void bar_nll_bwd_kernel(BarArgs a) {
  const int lane = threadIdx.x & 63;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < a.R; r += (long)gridDim.x * 4) {
    const float* row = a.logits + r * a.ld;
    float* drow = a.dlogits + r * a.ld;
    const float lse = a.lse[r], g = a.gout[r];
    const int t = a.bucket[r];
    const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15) == 0) && ((reinterpret_cast<uintptr_t>(drow) & 15) == 0);
    const int n4 = vec ? (a.nbars / 4) : 0;
    for (int i = lane; i < n4; i += 64) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * i);
      f32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = g * (__expf(v[e] - lse) - ((4 * i + e) == t ? 1.f : 0.f));
      *reinterpret_cast<f32x4*>(drow + 4 * i) = d;
    }
    for (int i = n4 * 4 + lane; i < a.nbars; i += 64) drow[i] = g * (__expf(row[i] - lse) - (i == t ? 1.f : 0.f));
  }
}

// mean of the bar distribution: softmax(logits) . bucket_means (bar_distribution.py:35-38,110-117)
__global__ __launch_bounds__(256) void bar_mean_kernel(BarArgs a) {
  const int lane = threadIdx.x & 63;
  const int nb = a.nbars;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < a.R; r += (long)gridDim.x * 4) {
    const float* row = a.logits + r * a.ld;
    float mx, se;
    row_max_sumexp(row, nb, lane, mx, se);
    float acc = 0.f;
    for (int i = lane; i < nb; i += 64) {
      float bm = a.borders[i] + 0.5f * (a.borders[i + 1] - a.borders[i]);
      if (a.full_support) {
        if (i == 0) bm = a.borders[1] - (a.borders[1] - a.borders[0]) / HALFNORMAL_ICDF_HALF * SQRT_2_OVER_PI;
        if (i == nb - 1) bm = a.borders[nb - 1] + (a.borders[nb] - a.borders[nb - 1]) / HALFNORMAL_ICDF_HALF * SQRT_2_OVER_PI;
      }
      acc += __expf(row[i] - mx) * bm;
    }
    acc = wave_sum(acc);
    if (lane == 0) a.mean_out[r] = acc / se;
  }
}

static int rows_grid(long R) { return (int)std::max<long>(1, std::min<long>((R + 3) / 4, 8192)); }

int launch_bar_nll_fwd(const BarArgs& a, hipStream_t s) {
  if (a.R == 0) return PFN_OK;
  if (a.nbars < 1) return PFN_ERR_ARGUMENT;
  hipLaunchKernelGGL(bar_nll_fwd_kernel, dim3(rows_grid(a.R)), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}
int launch_bar_nll_bwd(const BarArgs& a, hipStream_t s) {
  if (a.R == 0) return PFN_OK;
  hipLaunchKernelGGL(bar_nll_bwd_kernel, dim3(rows_grid(a.R)), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}
int launch_bar_mean(const BarArgs& a, hipStream_t s) {
  if (a.R == 0) return PFN_OK;
  hipLaunchKernelGGL(bar_mean_kernel, dim3(rows_grid(a.R)), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

}  // namespace pfn
