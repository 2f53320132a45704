// Hardware-fact probe for gfx950: MFMA fragment layouts and ds_read_b64_tr_b16 lane mapping.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_layouts.hip -o tools/probe_layouts
// Prints PASS/FAIL per hypothesis and the raw tr-read permutation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static inline float bf2f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }

// Hypothesis 32x32x16: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31], D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
__global__ void k_mfma32(const unsigned short* A, const unsigned short* B, float* D) {
  int l = threadIdx.x;
  short8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (short)A[(l & 31) * 16 + 8 * (l >> 5) + e]; b[e] = (short)B[(8 * (l >> 5) + e) * 32 + (l & 31)]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// Hypothesis 16x16x32: A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15], D[row=4*(l>>4)+r][col=l&15]
__global__ void k_mfma16(const unsigned short* A, const unsigned short* B, float* D) {
  int l = threadIdx.x;
  short8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (short)A[(l & 15) * 32 + 8 * (l >> 4) + e]; b[e] = (short)B[(8 * (l >> 4) + e) * 16 + (l & 15)]; }
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
// Hypothesis 32x32x2 f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D as 32x32 above
__global__ void k_mfma32f(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// tr read: lds[e] = e (u16); lane i reads 8 bytes at byte address 8*i (elements 4i..4i+3)
__global__ void k_tr(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  int l = threadIdx.x;
  for (int e = l; e < 1024; e += 64) lds[e] = (unsigned short)e;
  __syncthreads();
  unsigned addr = (unsigned)(size_t)(&lds[0]) + 8u * l;
  short4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
// tr read, second pattern: row-strided. lds viewed as [rows][64] u16 (128 B rows);
// 16-lane group g, lane i in group: row = 4*g + (i>>2), col = 4*(i&3)
__global__ void k_tr2(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  int l = threadIdx.x;
  for (int e = l; e < 4096; e += 64) lds[e] = (unsigned short)e;
  __syncthreads();
  int g = l >> 4, i = l & 15;
  unsigned addr = (unsigned)(size_t)(&lds[0]) + 2u * ((4 * g + (i >> 2)) * 64 + 4 * (i & 3));
  short4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2);} } while (0)

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s arch %s CUs %d clock %d kHz LDS/blk %zu\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  srand(1);
  {
    std::vector<unsigned short> A(32 * 16), B(16 * 32); std::vector<float> D(1024), R(1024, 0.f);
    for (auto& v : A) v = f2bf((rand() % 17 - 8) / 4.f); for (auto& v : B) v = f2bf((rand() % 13 - 6) / 2.f);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += bf2f(A[i * 16 + k]) * bf2f(B[k * 32 + j]);
    unsigned short *dA, *dB; float* dD; CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, 4096));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize()); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("mfma_f32_32x32x16_bf16 layout hypothesis: %s (max err %g)\n", err < 1e-3 ? "PASS" : "FAIL", err);
  }
  {
    std::vector<unsigned short> A(16 * 32), B(32 * 16); std::vector<float> D(256), R(256, 0.f);
    for (auto& v : A) v = f2bf((rand() % 17 - 8) / 4.f); for (auto& v : B) v = f2bf((rand() % 13 - 6) / 2.f);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += bf2f(A[i * 32 + k]) * bf2f(B[k * 16 + j]);
    unsigned short *dA, *dB; float* dD; CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize()); CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("mfma_f32_16x16x32_bf16 layout hypothesis: %s (max err %g)\n", err < 1e-3 ? "PASS" : "FAIL", err);
  }
  {
    std::vector<float> A(64), B(64), D(1024), R(1024, 0.f);
    for (auto& v : A) v = (rand() % 17 - 8) / 4.f; for (auto& v : B) v = (rand() % 13 - 6) / 2.f;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 2; ++k) R[i * 32 + j] += A[i * 2 + k] * B[k * 32 + j];
    float *dA, *dB, *dD; CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dD, 4096));
    CK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
    k_mfma32f<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize()); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("mfma_f32_32x32x2f32 layout hypothesis: %s (max err %g)\n", err < 1e-5 ? "PASS" : "FAIL", err);
  }
  {
    unsigned short* dO; CK(hipMalloc(&dO, 512)); std::vector<unsigned short> O(256);
    k_tr<<<1, 64>>>(dO); CK(hipDeviceSynchronize()); CK(hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost));
    // hypothesis: result[l][j] = element index 64*(l>>4) + 16*j + (l&15)
    bool ok = true; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) ok &= (O[l * 4 + j] == 64 * (l >> 4) + 16 * j + (l & 15));
    printf("ds_read_b64_tr_b16 contiguous hypothesis: %s\n", ok ? "PASS" : "FAIL");
    printf("tr raw (lane: 4 source element indices; lane i read elements 4i..4i+3):\n");
    for (int l = 0; l < 64; ++l) { printf("  l%02d: %4d %4d %4d %4d\n", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3]); }
    k_tr2<<<1, 64>>>(dO); CK(hipDeviceSynchronize()); CK(hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost));
    // hypothesis: lane (g,i) elem j = lds[(4g + j)*64 + i]
    ok = true; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) ok &= (O[l * 4 + j] == (4 * (l >> 4) + j) * 64 + (l & 15));
    printf("ds_read_b64_tr_b16 row-strided hypothesis: %s\n", ok ? "PASS" : "FAIL");
    if (!ok) for (int l = 0; l < 64; ++l) { printf("  l%02d: %4d %4d %4d %4d\n", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3]); }
  }
  return 0;
}
