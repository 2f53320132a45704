// What would the dQ of the attention backward cost if the key-block pass ACCUMULATED it with f32 atomics instead of spilling dS^T (0.86 GB written + read back by a
// query-block pass of 200-210 us at the north-star micro-batch: 32 datasets x 4 heads, bptt 2000, eval position 1604, head dim 128)?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/probe_dq_atomics tools/probe_dq_atomics.hip && tools/probe_dq_atomics
// The probe replays ONLY the memory side of that variant with the key-block pass's own geometry and XCD placement: 7 key blocks x 128 (dataset, head) pairs =
// 896 workgroups of 8 waves; a workgroup walks 63 query tiles and adds one [32 queries x 128 head-dim] f32 tile (16 KiB, already reduced over its 256 keys) per query
// tile into the pair's [2000 x 128] f32 dQ buffer (1 MB per pair; the 7 blocks of a pair hit the same addresses).  Nothing else runs: no MFMAs, no LDS
// reduction, no Q / dO tile traffic -- a LOWER bound on what the atomics would add to the pass.  Variants: global_atomic_add_f32 (returnless), the same bytes as
// plain stores (what the dS^T spill costs per byte, two bytes per pair instead of 4 x 128 / 256), and atomics with the 7 blocks of a pair spread over XCDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int B = 32, H = 4, S = 2000, D = 128, SEP = 1604, KBLK = 256, QT = 32;
constexpr int NKB = (SEP + KBLK - 1) / KBLK, NPAIR = B * H, NTILE = (S + QT - 1) / QT;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {      // the attention kernels' workgroup -> (block, pair) order: whole pairs per XCD
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int MODE>      // 0: atomics, pairs pinned to an XCD; 1: plain stores (same bytes); 2: atomics, no XCD pinning (blockIdx order)
__global__ __launch_bounds__(512) void probe(float* dq, int tiles) {
  const int id = MODE == 2 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int pair = id / NKB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* base = dq + (long)pair * S * D;
  const float v = 1e-3f * (float)(lane + 1);
  for (int t = 0; t < tiles; ++t) {
    // the tile's 32 x 128 floats = 16 KiB: wave w adds rows 4 w .. 4 w + 3 (2 KiB = 8 wave instructions of 64 lanes x 4 B: one 256-byte half row each)
    float* row = base + ((long)t * QT + wave * 4) * D;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float* p = row + (i >> 1) * D + (i & 1) * 64 + lane;
      if ((t * QT + wave * 4 + (i >> 1)) < S) {
        if (MODE == 1) __builtin_nontemporal_store(v, p);
        else unsafeAtomicAdd(p, v);
      }
    }
  }
}

template <int MODE> void run(const char* name, float* dq) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<NKB * NPAIR, 512>>>(dq, NTILE);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    probe<MODE><<<NKB * NPAIR, 512>>>(dq, NTILE);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double bytes = (double)NKB * NPAIR * NTILE * QT * D * 4;
  printf("%-64s %8.1f us   %6.2f GB per launch   %5.2f TB/s\n", name, best * 1e3, bytes / 1e9, bytes / (best * 1e-3) / 1e12);
}

int main() {
  float* dq;
  hipMalloc(&dq, sizeof(float) * (size_t)NPAIR * S * D);
  hipMemset(dq, 0, sizeof(float) * (size_t)NPAIR * S * D);
  printf("dQ by atomics, memory side only: %d key blocks x %d pairs = %d workgroups, %d query tiles of 16 KiB each, dQ buffer %.0f MB\n", NKB, NPAIR, NKB * NPAIR, NTILE,
         (double)NPAIR * S * D * 4 / 1e6);
  run<0>("global_atomic_add_f32, a pair's 7 blocks on one XCD (L2-local)", dq);
  run<2>("global_atomic_add_f32, blocks in launch order (pairs spread over XCDs)", dq);
  run<1>("plain non-temporal stores of the same bytes", dq);
  return 0;
}
