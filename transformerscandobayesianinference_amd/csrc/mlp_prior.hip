// Batched forward of the BNN ("MLP") prior: every synthetic dataset is the output of a small random
// network on Gaussian inputs,
//     h_0 = causes W_0^T + b_0;   h_l = act(h_{l-1}) W_l^T + b_l + noise_std * eps_l   (l = 1 .. L-1);   y = h_{L-1}[:, 0]
// (reference priors/mlp.py:116-124 network, :150-157 forward, GaussianNoise :32-38; non-causal branch: x = causes,
// y = last layer).  The reference builds one torch module per dataset group and runs B forwards of ~10 tiny
// kernels from a Python loop (:195-197); here one launch covers the whole batch.
//
// One workgroup = one dataset x 32 sequence rows.  The layer's weights (<= 152 x 152 f32, transposed: [in][out])
// and two ping-pong activation tiles ([feature][row]) live in LDS; a thread owns a 4-row x 4-column register
// tile, so a k step is two 16-byte LDS reads for 16 FMAs.  Noise and (optionally) the causes come from the
// counter-based generator; both can be injected for the parity tests.  f32 throughout, like the reference.
#include "pfn_device.h"
#include "pfn_kernels.h"

namespace pfn {

constexpr int MLP_TR = 32;   // rows per workgroup

PFN_DEV float mlp_act(float v, int kind) {
  switch (kind) {
    case 1: return fmaxf(v, 0.f);
    case 2: return tanhf(v);
    case 3: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

__global__ __launch_bounds__(256) void mlp_prior_kernel(MlpPriorArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int HP = a.HP;
  float* Wt = sm;                            // [HP][HP]   current layer, [in][out]
  float* act0 = sm + HP * HP;                // [HP][MLP_TR]
  float* act1 = act0 + HP * MLP_TR;
  const int b = blockIdx.y, t0 = blockIdx.x * MLP_TR;
  const int model = a.model_of[b];
  const int nc = a.dims[model * 3 + 0], L = a.dims[model * 3 + 2];
  const float nstd = a.noise_std[model];
  const float* Wm = a.weights + (long)model * a.Lmax * HP * HP;
  const float* bm = a.biases + (long)model * a.Lmax * HP;

  // layer-0 input: the causes of rows t0 .. t0+31, transposed into act0[k][r]
  for (int i = threadIdx.x; i < MLP_TR * (HP / 4); i += 256) {
    const int r = i / (HP / 4), k4 = (i % (HP / 4)) * 4;
    const int t = t0 + r;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < a.T) {
      float* cp = a.causes + ((long)b * a.T + t) * HP + k4;
      if (a.gen_causes) {
        float n[4];
        normal4(philox4x32_10(((unsigned long long)b * a.T + t) * (HP / 4) + k4 / 4, a.offset * 4, a.seed), n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k4 + e < nc) ? n[e] : 0.f;
        *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
        const f32x4 c = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) act0[(k4 + e) * MLP_TR + r] = v[e];
  }
  float* ain = act0;
  float* aout = act1;
  for (int l = 0; l < L; ++l) {
    __syncthreads();   // previous layer's outputs written; Wt free
    for (int i = threadIdx.x; i < HP * HP / 4; i += 256) *reinterpret_cast<f32x4*>(Wt + i * 4) = *reinterpret_cast<const f32x4*>(Wm + (long)l * HP * HP + i * 4);
    __syncthreads();
    const int ntile = (MLP_TR / 4) * (HP / 4);
    for (int tile = threadIdx.x; tile < ntile; tile += 256) {
      const int r4 = (tile % (MLP_TR / 4)) * 4, j4 = (tile / (MLP_TR / 4)) * 4;
      float acc[4][4];   // [col j][row r]
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.f;
      for (int k = 0; k < HP; ++k) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(Wt + k * HP + j4);
        f32x4 x = *reinterpret_cast<const f32x4*>(ain + k * MLP_TR + r4);
        if (l > 0) {
#pragma unroll
          for (int v = 0; v < 4; ++v) x[v] = mlp_act(x[v], a.activation);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[u][v] += w[u] * x[v];
      }
      const f32x4 bias = *reinterpret_cast<const f32x4*>(bm + (long)l * HP + j4);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int t = t0 + r4 + v;
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (l > 0 && t < a.T) {
          if (a.noise) {
            const f32x4 e4 = *reinterpret_cast<const f32x4*>(a.noise + (((long)b * (a.Lmax - 1) + (l - 1)) * a.T + t) * HP + j4);
#pragma unroll
            for (int u = 0; u < 4; ++u) nz[u] = e4[u];
          } else {
            normal4(philox4x32_10((((unsigned long long)b * a.Lmax + l) * a.T + t) * (HP / 4) + j4 / 4, a.offset * 4 + 1, a.seed), nz);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u][v] += bias[u] + nstd * nz[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(aout + (j4 + u) * MLP_TR + r4) = f32x4{acc[u][0], acc[u][1], acc[u][2], acc[u][3]};
      if (a.hidden && l > 0) {      // causal variant (priors/mlp.py:158-166): every node after the first layer is a candidate feature / target
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int t = t0 + r4 + v;
          if (t < a.T) *reinterpret_cast<f32x4*>(a.hidden + (((long)b * (a.Lmax - 1) + (l - 1)) * a.T + t) * HP + j4) = f32x4{acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
        }
      }
    }
    float* tmp = ain; ain = aout; aout = tmp;
  }
  __syncthreads();
  if (threadIdx.x < MLP_TR && t0 + threadIdx.x < a.T) a.y[(long)b * a.T + t0 + threadIdx.x] = ain[threadIdx.x];   // column 0 of the last layer
}

int launch_mlp_prior(const MlpPriorArgs& a, hipStream_t s) {
  if (a.B < 1 || a.T < 1 || a.HP < 4 || a.HP % 4 || a.Lmax < 1) return PFN_ERR_ARGUMENT;
  const size_t lds = ((size_t)a.HP * a.HP + 2 * (size_t)a.HP * MLP_TR) * sizeof(float);
  if (lds > 160 * 1024) return PFN_ERR_UNSUPPORTED;
  static LdsAllowance allowance;
  allowance.ensure(mlp_prior_kernel, lds);
  hipLaunchKernelGGL(mlp_prior_kernel, dim3((a.T + MLP_TR - 1) / MLP_TR, a.B), dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

}  // namespace pfn
