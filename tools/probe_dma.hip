// How fast does a CU take in operand bytes on gfx950, by LDS-DMA (global_load_lds_dwordx4) and by plain 16-byte loads into registers?
// One 512-thread workgroup per CU re-reads a region of `span` bytes (its own, or one shared by all) in 64 KiB rounds:
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_dma tools/probe_dma.hip && tools/probe_dma
// `sync_every` rounds are in flight between barriers (the barrier carries vmcnt(0)).  Prints bytes per clock per CU (at the clock measured by s_memrealtime-free wall time and an assumed 2.1 GHz) and TB/s chip-wide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// MODE 0: LDS-DMA, MODE 1: registers (+ ds_write_b128 when WRITE_LDS)
template <int MODE, bool WRITE_LDS>
__global__ __launch_bounds__(512) void probe(const char* src, long per_wg_stride, long span, int rounds, int sync_every, int seg, int pitch, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  lds_char* smem = (lds_char*)smem_raw;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const char* base = src + (long)blockIdx.x * per_wg_stride;
  u32x4 acc = {0, 0, 0, 0};
  long off = 0;
  // seg > 0: the round's 64 KiB are 65536 / seg row segments of `seg` bytes, `pitch` bytes apart (an operand tile cut out of a
  // row-major matrix); lane l of piece p reads byte (p * 1024 + 16 l) of the tile = row (that / seg), column (that % seg)
  long loff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int byte = (wave + 8 * i) * 1024 + lane * 16;
    loff[i] = seg > 0 ? (long)(byte / seg) * pitch + byte % seg : byte;
  }
  const long round_bytes = seg > 0 ? (long)(65536 / seg) * pitch : 65536;
  for (int r = 0; r < rounds; ++r) {
    // 64 KiB per round: wave w moves pieces w, w + 8, ..., w + 56 (1 KiB each)
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(base + off + loff[i]), (lvoid_t*)(smem + (r & 1) * 65536 + (wave + 8 * i) * 1024), 16, 0, 0);
    } else {
      u32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const u32x4*>(base + off + loff[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (WRITE_LDS) *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(smem + (r & 1) * 65536 + (wave + 8 * i) * 1024 + lane * 16) = v[i];
        else acc ^= v[i];
      }
    }
    off += round_bytes;
    if (off + round_bytes > span) off = 0;
    if ((r + 1) % sync_every == 0) __syncthreads();   // (carries vmcnt(0): at most sync_every rounds in flight)
  }
  __syncthreads();
  if (MODE == 0 || WRITE_LDS) acc = *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(smem + threadIdx.x * 16);
  out[blockIdx.x * 512 + threadIdx.x] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}

template <int MODE, bool WL> void run(const char* name, const char* src, long stride, long span, int sync_every, float* out, int seg = 0, int pitch = 0) {
  const int rounds = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, WL>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  probe<MODE, WL><<<256, 512, 131072>>>(src, stride, span, 50, sync_every, seg, pitch, out);
  hipEventRecord(e0);
  probe<MODE, WL><<<256, 512, 131072>>>(src, stride, span, rounds, sync_every, seg, pitch, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * rounds * 65536;
  printf("%-46s %8.1f us per 64 KiB round  %6.2f TB/s chip-wide  %5.1f B/clk/CU at 2.1 GHz\n", name, ms * 1e3 / rounds, bytes / (ms * 1e-3) / 1e12,
         bytes / 256 / (ms * 1e-3) / 2.1e9);
}

int main() {
  const long total = 1L << 30;
  char* src; float* out;
  hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&out, 256 * 512 * 4);
  struct { const char* what; long stride, span; } cases[] = {
    {"own 64 KiB per workgroup (L2 hits)", 65536, 65536},
    {"own 1 MiB per workgroup (256 MiB: memory-side cache)", 1 << 20, 1 << 20},
    {"own 4 MiB per workgroup (1 GiB: HBM)", 4 << 20, 4 << 20},
    {"one 1 MiB region shared by all workgroups", 0, 1 << 20},
  };
  for (int sync_every : {1, 2, 4})
    for (auto& c : cases) {
      printf("-- %s; a barrier (vmcnt(0)) every %d round(s) of 64 KiB\n", c.what, sync_every);
      run<0, false>("  LDS-DMA (global_load_lds_dwordx4)", src, c.stride, c.span, sync_every, out);
      run<1, false>("  16-byte loads into registers", src, c.stride, c.span, sync_every, out);
      if (sync_every == 1) run<1, true>("  16-byte loads into registers + ds_write_b128", src, c.stride, c.span, sync_every, out);
    }
  // operand tiles cut out of row-major matrices: row segments of `seg` bytes, `pitch` bytes apart, from HBM (own 4 MiB per workgroup)
  struct { int seg, pitch; const char* what; } cuts[] = {
    {128, 1024, "128-byte segments, 1 KiB pitch (A tile of an NT GEMM, K = 512 bf16)"},
    {128, 3072, "128-byte segments, 3 KiB pitch (K = 1536)"},
    {512, 1024, "512-byte segments, 1 KiB pitch (weight-gradient operand of a 512-wide matrix)"},
    {512, 3072, "512-byte segments, 3 KiB pitch (of a 1536-wide matrix)"},
    {2048, 4096, "2-KiB segments, 4 KiB pitch"},
  };
  for (auto& c : cuts) {
    printf("-- HBM, %s; one round in flight\n", c.what);
    run<0, false>("  LDS-DMA (global_load_lds_dwordx4)", src, 4 << 20, 4 << 20, 1, out, c.seg, c.pitch);
  }
  return 0;
}
