// Device-side building blocks shared by every gfx950 kernel in this library.
//
// Hardware facts used here were verified on an MI355X by tools/probe_layouts.hip
// (profiles/r01_probe_layouts.txt):
//   * v_mfma_f32_32x32x16_bf16 : A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31],
//                                D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31], r in [0,16)
//   * v_mfma_f32_32x32x2_f32   : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], same D map
//   * ds_read_b64_tr_b16       : inside each 16-lane group, lane i supplies the address of
//                                4 consecutive b16 (R[i][0..3]); lane i receives
//                                R[4j + (i>>2)][i&3] for j = 0..3, i.e. column i of the
//                                4x16 block whose row j is held by lanes 4j..4j+3.
//
// A "fragment" is what one lane feeds one 32x32 MFMA step that contracts 16 indices:
// 8 contraction "slots" (h = lane>>5, e = 0..7).  Slot (h,e) of the A operand always pairs
// with slot (h,e) of the B operand, so any mapping slot -> contraction index is legal as long
// as both operands use the same one.  Two mappings are used:
//   M1 (memory order)      : k = 8h + e
//   M2 (accumulator order) : k = 8*(e>>2) + 4h + (e&3)   -- what a lane already holds when the
//                            operand is a just-computed 32x32 accumulator tile (P, dS)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfn {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// fp16 operands (round 6): 11-bit significand at the same MFMA rate and bytes as bf16 (v_mfma_f32_32x32x16_f16) -- the operand format that brings the
// TIMED forward under the north star's 1e-3 (profiles/r06_operand_format_simulation.json); gradients travel under a power-of-two loss scale (LossScale below)
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// vector types of an operand element type (the 16-bit ones; float only so that discarded `if constexpr` branches still name a type)
template <typename T> struct OpVec;
template <> struct OpVec<bf16> { typedef bf16x8 x8; typedef bf16x4 x4; typedef bf16x2 x2; };
template <> struct OpVec<f16> { typedef f16x8 x8; typedef f16x4 x4; typedef f16x2 x2; };
template <> struct OpVec<float> { typedef f32x4 x8; typedef f32x4 x4; typedef f32x2 x2; };
template <typename T> using X8 = typename OpVec<T>::x8;
template <typename T> using X4 = typename OpVec<T>::x4;
template <typename T> using X2 = typename OpVec<T>::x2;

#define PFN_DEV __device__ __forceinline__
// Scheduling fence for LDS and MFMA instructions only (vector / scalar ALU and global memory instructions may still
// cross it): the machine scheduler otherwise sinks every LDS fragment load to just before the MFMA that consumes it
// -- minimal register pressure, but each MFMA then waits out a full LDS round trip.
#define PFN_PIN_LDS_MFMA() __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x20 | 0x40)

PFN_DEV int lane_id() { return threadIdx.x & 63; }

// ---------------------------------------------------------------------------------------------
// Fragments and the MFMA step
// ---------------------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16> {
  bf16x8 v;
  PFN_DEV void set(int e, float x) { v[e] = (bf16)x; }
};
template <> struct Frag<f16> {
  f16x8 v;
  PFN_DEV void set(int e, float x) { v[e] = (f16)x; }
};
template <> struct Frag<float> {
  float v[8];
  PFN_DEV void set(int e, float x) { v[e] = x; }
};

PFN_DEV f32x16 mma32(const Frag<bf16>& a, const Frag<bf16>& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
}
PFN_DEV f32x16 mma32(const Frag<f16>& a, const Frag<f16>& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, c, 0, 0, 0);
}
// exact-f32 path: eight 32x32x2 steps; step e contracts slots (0,e) and (1,e).
PFN_DEV f32x16 mma32(const Frag<float>& a, const Frag<float>& b, f32x16 c) {
#pragma unroll
  for (int e = 0; e < 8; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[e], b.v[e], c, 0, 0, 0);
  return c;
}

// The same step with the accumulator PINNED to the AGPR half of the register file (inline assembly: the only way to say so).
// For kernels whose long-lived accumulators alone fill 256 registers (attention backward at head dim 256: dK and dV): left to
// itself hipcc spreads every value over both halves and pays v_accvgpr copies plus scratch spills inside the loop; with the
// accumulators fixed in AGPRs everything the vector ALU touches fits the VGPR half.  The leading s_nop covers an operand
// the vector ALU wrote just before (hipcc pads nothing inside an asm statement); accumulate chains need no padding, and the
// epilogue that reads the accumulators first waits out the MFMA -> reader hazard (mma32_acc_drain).
PFN_DEV void mma32_acc(f32x16& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a.v), "v"(b.v));
}
PFN_DEV void mma32_acc(f32x16& acc, const Frag<f16>& a, const Frag<f16>& b) {
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a.v), "v"(b.v));
}
PFN_DEV void mma32_acc(f32x16& acc, const Frag<float>& a, const Frag<float>& b) { acc = mma32(a, b, acc); }
PFN_DEV void mma32_acc_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// row index inside a 32x32 accumulator tile held by (lane, r)
PFN_DEV int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// An accumulator tile re-used as an operand: slots follow mapping M2, chunk c = 0/1 selects
// tile rows 16c..16c+15.  Lane keeps its column (l&31).
template <typename T> PFN_DEV Frag<T> acc_to_frag(const f32x16& p, int c) {
  Frag<T> f;
#pragma unroll
  for (int e = 0; e < 8; ++e) f.set(e, p[8 * c + e]);
  return f;
}

// ---------------------------------------------------------------------------------------------
// LDS tiles.  Everything is addressed in bytes relative to a tile base; rows are RB bytes.
// ---------------------------------------------------------------------------------------------
// 16-byte-chunk XOR swizzle that makes "32 lanes read the same chunk column of 32 different
// rows" (ds_read_b128) conflict free for power-of-two row sizes.
template <int RB> PFN_DEV int swz16(int row, int chunk) {
  constexpr int NCH = RB / 16;
  if constexpr (NCH >= 16) return chunk ^ (row & 15);
  else if constexpr (NCH == 8) return chunk ^ ((row >> 1) & 7);
  else if constexpr (NCH == 4) return chunk ^ ((row >> 2) & 3);
  else if constexpr (NCH == 2) return chunk ^ ((row >> 3) & 1);
  else return chunk;
}
template <int RB> PFN_DEV int lds_off16(int row, int chunk) { return row * RB + swz16<RB>(row, chunk) * 16; }

// 64-byte-unit XOR swizzle for tiles consumed by ds_read_b64_tr_b16 (4 consecutive rows must
// land in 4 different 16-bank quarters).
template <int RB> PFN_DEV int swz64(int row, int unit) {
  constexpr int U = RB / 64;
  if constexpr (U >= 4) return unit ^ (row & 3);
  else if constexpr (U == 2) return unit ^ ((row >> 1) & 1);
  else return unit;
}
template <int RB> PFN_DEV int lds_off64(int row, int byte_in_row) {
  return row * RB + swz64<RB>(row, byte_in_row >> 6) * 64 + (byte_in_row & 63);
}

// LDS pointers are kept in address space 3 so every access lowers to a ds_* instruction.
typedef __attribute__((address_space(3))) char lds_char;
typedef lds_char* LdsPtr;
PFN_DEV LdsPtr lds_cast(void* generic_shared) { return (LdsPtr)generic_shared; }

PFN_DEV u32x4 lds_read16(const lds_char* p) { return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(p); }
PFN_DEV void lds_write16(LdsPtr p, u32x4 v) { *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(p) = v; }
PFN_DEV float lds_read_f32(const lds_char* p) { return *reinterpret_cast<const __attribute__((address_space(3))) float*>(p); }
PFN_DEV void lds_write_f32(LdsPtr p, float v) { *reinterpret_cast<__attribute__((address_space(3))) float*>(p) = v; }

// Row fragment (mapping M1): lane wants row `row`, contraction elements k0 + 8h + e from a
// swz16 tile whose rows hold the contraction index contiguously.
template <typename T, int RB> PFN_DEV Frag<T> load_frag_row(const lds_char* tile, int row, int k0) {
  const int h = lane_id() >> 5;
  Frag<T> f;
  if constexpr (sizeof(T) == 2) {
    u32x4 raw = lds_read16(tile + lds_off16<RB>(row, (k0 >> 3) + h));
    f.v = __builtin_bit_cast(typename OpVec<T>::x8, raw);
  } else {
    const int c = (k0 >> 2) + 2 * h;
    u32x4 r0 = lds_read16(tile + lds_off16<RB>(row, c));
    u32x4 r1 = lds_read16(tile + lds_off16<RB>(row, c + 1));
    f32x4 a = __builtin_bit_cast(f32x4, r0), b = __builtin_bit_cast(f32x4, r1);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f.v[e] = a[e]; f.v[4 + e] = b[e]; }
  }
  return f;
}

typedef __attribute__((ext_vector_type(4))) short i16x4;
template <typename T> PFN_DEV typename OpVec<T>::x4 ds_read_tr16_b64(const lds_char* p) {      // the same instruction for every 2-byte element type
  return __builtin_bit_cast(typename OpVec<T>::x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p)));
}

// Transposed fragment from a swz64 tile whose ROWS are the contraction index and whose columns
// are the operand's i/j index.  Lane wants column col0 + (l&31).  MAP 1: rows k0+8h+e,
// MAP 2: rows k0 + 8*(e>>2) + 4h + (e&3).  bf16 uses the hardware transpose read; f32 uses
// eight b32 reads.
template <typename T, int RB, int MAP> PFN_DEV Frag<T> load_frag_tr(const lds_char* tile, int k0, int col0) {
  const int l = lane_id(), h = l >> 5;
  Frag<T> f;
  if constexpr (sizeof(T) == 2) {
    const int i = l & 15, g = (l >> 4) & 1;
    const int colb = (col0 + 16 * g + 4 * (i & 3)) * 2;
    const int ra = (MAP == 1) ? (k0 + 8 * h) : (k0 + 4 * h);
    const int rb = (MAP == 1) ? (k0 + 8 * h + 4) : (k0 + 8 + 4 * h);
    const auto lo = ds_read_tr16_b64<T>(tile + lds_off64<RB>(ra + (i >> 2), colb));
    const auto hi = ds_read_tr16_b64<T>(tile + lds_off64<RB>(rb + (i >> 2), colb));
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
    const int colb = (col0 + (l & 31)) * 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = (MAP == 1) ? (k0 + 8 * h + e) : (k0 + 8 * (e >> 2) + 4 * h + (e & 3));
      f.v[e] = lds_read_f32(tile + lds_off64<RB>(row, colb));
    }
  }
  return f;
}

// ---------------------------------------------------------------------------------------------
// Padded (linear) LDS images, used by the attention kernels.  The XOR swizzles above make every
// fragment address a different non-linear function of the lane, i.e. one address VGPR (and one
// integer op per tile) per read; with a padded row stride the address is lane_base + constant, so
// a whole tile is read off ONE base register with immediate offsets:
//   row images (ds_read_b128 along the contraction): stride RB + 16 B  -> 16 consecutive rows hit 16 distinct
//                                                   4-bank groups (stride/4 = 4 mod 64 ... an odd multiple of 4)
//   col images (ds_read_b64_tr_b16 across rows)    : stride = 64 or 192 mod 256 -> the 4 rows of one transpose
//                                                   read sit in 4 different 16-bank quarters
// ---------------------------------------------------------------------------------------------
template <int RB> struct PadStride {
  static constexpr int ROW = RB + 16;
  static constexpr int COL = (RB % 256 == 64 || RB % 256 == 192) ? RB : RB + 64;
};

template <typename T, int STRIDE> PFN_DEV Frag<T> load_frag_row_p(const lds_char* tile, int row, int k0) {
  const int h = lane_id() >> 5;
  Frag<T> f;
  if constexpr (sizeof(T) == 2) {
    u32x4 raw = lds_read16(tile + row * STRIDE + ((k0 >> 3) + h) * 16);
    f.v = __builtin_bit_cast(typename OpVec<T>::x8, raw);
  } else {
    const int c = (k0 >> 2) + 2 * h;
    u32x4 r0 = lds_read16(tile + row * STRIDE + c * 16);
    u32x4 r1 = lds_read16(tile + row * STRIDE + c * 16 + 16);
    f32x4 a = __builtin_bit_cast(f32x4, r0), b = __builtin_bit_cast(f32x4, r1);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f.v[e] = a[e]; f.v[4 + e] = b[e]; }
  }
  return f;
}

template <typename T, int STRIDE, int MAP> PFN_DEV Frag<T> load_frag_tr_p(const lds_char* tile, int k0, int col0) {
  const int l = lane_id(), h = l >> 5;
  Frag<T> f;
  if constexpr (sizeof(T) == 2) {
    const int i = l & 15, g = (l >> 4) & 1;
    const int colb = (col0 + 16 * g + 4 * (i & 3)) * 2;
    const int ra = (MAP == 1) ? (k0 + 8 * h) : (k0 + 4 * h);
    const int rb = (MAP == 1) ? (k0 + 8 * h + 4) : (k0 + 8 + 4 * h);
    const auto lo = ds_read_tr16_b64<T>(tile + (ra + (i >> 2)) * STRIDE + colb);
    const auto hi = ds_read_tr16_b64<T>(tile + (rb + (i >> 2)) * STRIDE + colb);
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
    const int colb = (col0 + (l & 31)) * 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = (MAP == 1) ? (k0 + 8 * h + e) : (k0 + 8 * (e >> 2) + 4 * h + (e & 3));
      f.v[e] = lds_read_f32(tile + row * STRIDE + colb);
    }
  }
  return f;
}

// One 32-column block of a row-per-lane accumulator tile (lane = row, half-wave h holds columns 8g + 4h .. +3 of group g)
// to global memory.  bf16: neighbouring groups are exchanged between the half-waves (v_permlane32_swap) so every lane
// owns 8 contiguous columns and the block leaves as two 16-byte stores per lane instead of four 8-byte ones (the store
// tail of these kernels is instruction-issue bound).  All lanes must call it (the exchange is wave-wide); `valid`
// guards the stores.  v[4 g + e] = column 8 g + 4 h + e.
template <typename T> PFN_DEV void store_row_block(T* row_block, const float (&v)[16], int h, bool valid) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      typedef typename OpVec<T>::x2 x2;
      x2 a0 = {(T)v[8 * p + 0], (T)v[8 * p + 1]}, a1 = {(T)v[8 * p + 2], (T)v[8 * p + 3]};
      x2 b0 = {(T)v[8 * p + 4], (T)v[8 * p + 5]}, b1 = {(T)v[8 * p + 6], (T)v[8 * p + 7]};
      const auto r0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, b0), false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, b1), false, false);
      const u32x4 w = {r0[0], r1[0], r0[1], r1[1]};
      if (valid) *reinterpret_cast<u32x4*>(row_block + 16 * p + 8 * h) = w;
    }
  } else {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      if (valid) *reinterpret_cast<f32x4*>(row_block + 8 * rg + 4 * h) = f32x4{v[4 * rg], v[4 * rg + 1], v[4 * rg + 2], v[4 * rg + 3]};
  }
}

// A 32 x 32 tile held as two operand fragments in accumulator order (mapping M2: lane = row j, f0 holds columns 8g + 4h + e of
// groups g = 0, 1, f1 of groups 2, 3 -- what acc_to_frag produced) to a BLOCKED global layout in which every store
// instruction of the wave covers one contiguous KiB (whole 128-byte lines; row-major stores of such a tile put 32 bytes on
// each of 32 rows, and partial-line writes retire several times slower).  Block = 32 rows x 32 columns of T:
//   bf16: chunk (p, j, h) at ((p * 32 + j) * 2 + h) * 16 bytes holds row j, columns 16 p + 8 h .. + 7
//   f32 : chunk (p, s, j, h) at (((2 p + s) * 32 + j) * 2 + h) * 16 bytes holds row j, columns 16 p + 8 s + 4 h .. + 3
// blocked_chunk_col / blocked_chunk_row invert the map for the reader (chunk index w in address order).
template <typename T> PFN_DEV void store_frag_pair_blocked(T* block, const Frag<T>& f0, const Frag<T>& f1, int j, int h, bool valid) {
  char* base = reinterpret_cast<char*>(block);
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const u32x4 d = __builtin_bit_cast(u32x4, p == 0 ? f0.v : f1.v);   // {group 2p: 2 dwords, group 2p+1: 2 dwords}
      const auto r0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
      const u32x4 w = {r0[0], r1[0], r0[1], r1[1]};                      // columns 16 p + 8 h .. + 7 of row j
      // non-temporal: written once, read by a later kernel -- keep it from evicting the operand tiles this kernel re-reads from L2
      if (valid) __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(base + ((p * 32 + j) * 2 + h) * 16));
    }
  } else {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const Frag<T>& f = p == 0 ? f0 : f1;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
        if (valid) __builtin_nontemporal_store(f32x4{f.v[4 * sub], f.v[4 * sub + 1], f.v[4 * sub + 2], f.v[4 * sub + 3]},
                                               reinterpret_cast<f32x4*>(base + (((2 * p + sub) * 32 + j) * 2 + h) * 16));
    }
  }
}
template <typename T> PFN_DEV int blocked_chunk_row(int w) { return (w >> 1) & 31; }
template <typename T> PFN_DEV int blocked_chunk_col(int w) {
  if constexpr (sizeof(T) == 2) return 16 * (w >> 6) + 8 * (w & 1);
  else return 8 * (w >> 6) + 4 * (w & 1);          // w >> 6 = 2 p + s
}

// ---------------------------------------------------------------------------------------------
// Buffer-descriptor loads: 32-bit per-lane byte offsets against a wave-uniform descriptor, and the hardware's range check
// (an offset at or beyond num_records reads as zero) instead of exec-masked branches around every load.
// ---------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t BufRsrc;
PFN_DEV BufRsrc make_rsrc(const void* base, long bytes) {
  const int n = bytes > 0x7ffffff0L ? 0x7ffffff0 : (bytes < 0 ? 0 : (int)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
PFN_DEV u32x4 buf_load16(BufRsrc r, int byte_offset) { return __builtin_amdgcn_raw_buffer_load_b128(r, byte_offset, 0, 0); }
constexpr int BUF_OOB = 0x7ffffff0;

// LDS-DMA of one 1-KiB piece (64 lanes x 16 bytes, lane-linear at the wave-uniform LDS address) through a buffer descriptor,
// issued from inline assembly ON PURPOSE: hipcc tracks the builtin form as a pending LDS write and puts `s_waitcnt vmcnt(0)`
// in front of the next LDS read it cannot prove disjoint -- every ds_read_b64_tr_b16 (the intrinsic carries no address
// information), i.e. in the middle of the tile the DMA was meant to overlap.  The assembly form is invisible to that pass;
// the kernel waits itself (`dma_wait_all`) where the data is needed.  Out-of-range lanes (offset >= the descriptor's
// num_records) write zeros.  M0 carries the LDS address and is restored (the compiler owns it).
struct DmaRsrc { u32x4 w; };
PFN_DEV DmaRsrc make_dma_rsrc(const void* base, long bytes) {
  const unsigned long long p = (unsigned long long)base;
  const int n = bytes > 0x7ffffff0L ? 0x7ffffff0 : (bytes < 0 ? 0 : (int)bytes);
  DmaRsrc r;
  r.w[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
  r.w[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);   // stride 0
  r.w[2] = __builtin_amdgcn_readfirstlane((unsigned)n);
  r.w[3] = 0x00020000u;
  return r;
}
PFN_DEV void dma16(const DmaRsrc& r, LdsPtr lds_dst_uniform, int byte_offset) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds_dst_uniform);
  // readfirstlane again at the point of use: a descriptor that reaches here through a select or a loop-carried value is
  // uniform in fact but not provably, and the "s" constraint would be handed a VGPR (a no-op copy when it already is scalar)
  const u32x4 w = {(unsigned)__builtin_amdgcn_readfirstlane(r.w[0]), (unsigned)__builtin_amdgcn_readfirstlane(r.w[1]),
                   (unsigned)__builtin_amdgcn_readfirstlane(r.w[2]), (unsigned)__builtin_amdgcn_readfirstlane(r.w[3])};
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(dst), "v"(byte_offset), "s"(w) : "memory");
}
// one dword per lane: LDS address = lds_dst_uniform + 4 * lane
PFN_DEV void dma4(const DmaRsrc& r, LdsPtr lds_dst_uniform, int byte_offset) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds_dst_uniform);
  const u32x4 w = {(unsigned)__builtin_amdgcn_readfirstlane(r.w[0]), (unsigned)__builtin_amdgcn_readfirstlane(r.w[1]),
                   (unsigned)__builtin_amdgcn_readfirstlane(r.w[2]), (unsigned)__builtin_amdgcn_readfirstlane(r.w[3])};
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(dst), "v"(byte_offset), "s"(w) : "memory");
}
// Two instructions on purpose.  The builtin is the one hipcc's wait-count pass sees: after it the pass knows nothing is pending
// (without it, loads issued before a loop -- K fragments held in registers -- stay "possibly pending" in its model, and it
// re-waits for them at every use inside the loop with vmcnt(7), (6), ... (0): against the real queue, which holds the DMA
// it cannot see, that drains everything in the middle of the tile).  The pass may drop a builtin wait it considers redundant,
// so the assembly form follows: free when the first one ran, the only one otherwise.
// the same through a per-lane 64-bit global address (no range check: the caller clamps)
PFN_DEV void dma16_global(const void* src, LdsPtr lds_dst_uniform) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds_dst_uniform);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(dst), "v"(src) : "memory");
}
// Workgroup barrier behind an explicit wait: everything but the N most recent vector-memory operations of this wave has landed
// (loads, LDS-DMA and stores retire in order on gfx9), and its LDS traffic is complete.  __syncthreads() instead carries a fence
// that waits for vmcnt(0) -- i.e. for every store still in flight.
template <int N> PFN_DEV void wait_vm_barrier() {
  __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}
PFN_DEV void dma_wait_all() {
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Register-staged global -> LDS tile copy through a buffer descriptor (see TileStage below for the pointer form).  The lane's
// chunk offsets inside a tile are computed once (`init`); a tile is selected by a byte offset added to them; rows beyond
// the descriptor's range read as zero.  ROWS x RB bytes, NT threads.
template <typename T, int ROWS, int RB, int NT> struct TileStageBuf {
  static constexpr int NCH = RB / 16;
  static constexpr int TOTAL = ROWS * NCH;
  static constexpr int PER = (TOTAL + NT - 1) / NT;
  u32x4 regs[PER];
  int voff[PER];
  PFN_DEV void init(int ld_bytes) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = threadIdx.x + i * NT;
      voff[i] = (TOTAL % NT == 0 || id < TOTAL) ? (id / NCH) * ld_bytes + (id % NCH) * 16 : BUF_OOB;
    }
  }
  PFN_DEV void issue(BufRsrc r, int tile_byte_offset) {
#pragma unroll
    for (int i = 0; i < PER; ++i) regs[i] = buf_load16(r, voff[i] + tile_byte_offset);
  }
  template <int STRIDE> PFN_DEV void commit_p(LdsPtr tile) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = threadIdx.x + i * NT;
      if (TOTAL % NT != 0 && id >= TOTAL) break;
      lds_write16(tile + (id / NCH) * STRIDE + (id % NCH) * 16, regs[i]);
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Cooperative global -> LDS tile copy in 16-byte chunks (register staged so the issue and the
// LDS write can be split around compute).  ROWS x RB bytes, NT threads.
// Source: row r of the tile starts at src + r*ld (elements of T); columns beyond `cols_valid`
// and rows beyond `rows_valid` read as zero.  Requires 16-byte aligned rows (ld*sizeof(T)%16==0).
// ---------------------------------------------------------------------------------------------
template <typename T, int ROWS, int RB, int NT> struct TileStage {
  static constexpr int NCH = RB / 16;
  static constexpr int TOTAL = ROWS * NCH;
  static constexpr int PER = (TOTAL + NT - 1) / NT;
  static constexpr int EPC = 16 / sizeof(T);  // elements per chunk
  u32x4 regs[PER];

  PFN_DEV void issue(const T* src, long ld, int rows_valid, int cols_valid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = threadIdx.x + i * NT;
      const int row = id / NCH, c = id % NCH;
      u32x4 v = {0u, 0u, 0u, 0u};
      if ((TOTAL % NT == 0 || id < TOTAL) && row < rows_valid && (c + 1) * EPC <= cols_valid)
        v = *reinterpret_cast<const u32x4*>(src + (long)row * ld + c * EPC);
      else if ((TOTAL % NT == 0 || id < TOTAL) && row < rows_valid && c * EPC < cols_valid) {
        // ragged tail inside a chunk: element-wise
        T tmp[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) tmp[e] = (c * EPC + e < cols_valid) ? src[(long)row * ld + c * EPC + e] : (T)0.f;
        v = *reinterpret_cast<u32x4*>(tmp);
      }
      regs[i] = v;
    }
  }
  // padded linear image (PadStride): chunk c of row r at r * STRIDE + 16 c
  template <int STRIDE> PFN_DEV void commit_p(LdsPtr tile) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = threadIdx.x + i * NT;
      if (TOTAL % NT != 0 && id >= TOTAL) break;
      lds_write16(tile + (id / NCH) * STRIDE + (id % NCH) * 16, regs[i]);
    }
  }
  template <bool SWZ64> PFN_DEV void commit(LdsPtr tile) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int id = threadIdx.x + i * NT;
      if (TOTAL % NT != 0 && id >= TOTAL) break;
      const int row = id / NCH, c = id % NCH;
      const int off = SWZ64 ? lds_off64<RB>(row, c * 16) : lds_off16<RB>(row, c);
      lds_write16(tile + off, regs[i]);
    }
  }
};

// ---------------------------------------------------------------------------------------------
// math helpers
// ---------------------------------------------------------------------------------------------
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. f32 round-off of a value in [-1, 1]):
// one reciprocal, one exp and a 5-term Horner chain instead of the library erff's branchy ~40 instructions --
// the GELU epilogues evaluate it for every element of the FFN activations (65 M per layer pass at the north star).
PFN_DEV float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float r = 1.f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
PFN_DEV float gelu_f(float x) { return 0.5f * x * (1.f + erf_fast(x * 0.70710678118654752440f)); }
PFN_DEV float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.f + erf_fast(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// GELU and its derivative together (one erf evaluation: its exp(-x^2/2) is the Gaussian density the derivative needs):
// the forward epilogue stores gelu'(pre-activation) instead of the pre-activation, the backward one only multiplies.
PFN_DEV void gelu_and_grad(float x, float& y, float& dy) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float ex = __expf(-ax * ax);                       // exp(-x^2 / 2)
  const float cdf = 0.5f * (1.f + copysignf(1.f - p * t * ex, x));
  y = x * cdf;
  dy = cdf + x * 0.39894228040143267794f * ex;
}
PFN_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
PFN_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
template <typename T> PFN_DEV float to_f(T x) { return (float)x; }

// First statement of every kernel that converts f32 to the operand type: with fp16 operands the wave's MODE.FP16_OVFL bit (23) is set, under which an fp16 VALU
// result beyond the format's range is +-65504 instead of +-inf (probed on gfx950: tools/probe_fp16.hip); true infinities and NaNs still pass through.
template <typename T> PFN_DEV void operand_store_mode() {
  if constexpr (sizeof(T) == 2 && !__is_same(T, bf16)) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
}

// ---------------------------------------------------------------------------------------------
// Loss scale of the fp16 backward (PFN_PREC_FP16).  fp16 spans 2^-14 .. 2^16 at full precision; the gradient of a mean loss over ~10^4 rows starts at
// 1e-4 .. 1e-8 per element, so the backward chain runs on dlogits * 2^k and every kernel that WRITES a parameter gradient (or d(src)) multiplies by 2^-k
// on the way out: the flat gradient buffer holds unscaled f32 values, exactly as in the other modes.  k is chosen per backward call ON THE DEVICE -- no host
// round trip -- from amax = max|dlogits| (absmax_kernel, rowwise.hip) so that amax * 2^k lands in [2^t, 2^(t+1)), t = the loss-scale target
// (PFN_TUNE_LOSS_SCALE_TARGET, default rowwise.hip g_loss_scale_target): 16 - t binades of headroom below 65504 for what the chain adds -- |W|, |q|, |k| of a
// trained model, and the SUMS over up to bptt query rows in dK / dV when few keys take all the attention (bptt x the row gradients when they agree) -- and full
// precision down to 2^-(14+t) of the largest element.  The target enters through the stored value (absmax_kernel writes amax * 2^(6 - t)), so the readers below
// keep one constant.  Both factors are powers of two: the scaling itself is exact.  A null pointer (the other precisions) means 1.
// What still overflows SATURATES: every kernel that stores fp16 runs with MODE.FP16_OVFL set (operand_store_mode below), so a conversion beyond 65504 gives
// +-65504 -- an element-wise clip of an outlier gradient ahead of the global-norm clip -- instead of an inf that the next product turns into NaN weights
// (round 6: the GP-fitting recipe trained to a bar NLL of -2.42 by epoch 67 and was all-NaN at epoch 68 without it).
// ---------------------------------------------------------------------------------------------
constexpr int LOSS_SCALE_TARGET_LOG2 = 6;
PFN_DEV float loss_scale_exp(const float* amax, int sign) {      // 2^(sign * k), k = 6 - floor(log2 stored value), clamped to the range where both factors are normal f32
  if (!amax) return 1.f;
  const unsigned bits = __builtin_bit_cast(unsigned, *amax);
  const int e = (int)((bits >> 23) & 255u) - 127;               // floor(log2 amax) for a normal value
  int k = LOSS_SCALE_TARGET_LOG2 - e;
  if (e == -127 || e == 128) k = 0;                             // zero / subnormal / inf / nan gradient: leave it alone
  k = k < -60 ? -60 : (k > 60 ? 60 : k);
  return __builtin_bit_cast(float, (unsigned)(127 + sign * k) << 23);
}
PFN_DEV float loss_scale_up(const float* amax) { return loss_scale_exp(amax, 1); }
PFN_DEV float loss_scale_down(const float* amax) { return loss_scale_exp(amax, -1); }

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter = (index, stream), key = seed
// ---------------------------------------------------------------------------------------------
struct U4 { unsigned x, y, z, w; };
PFN_DEV U4 philox4x32_10(unsigned long long idx, unsigned long long stream, unsigned long long seed) {
  unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = (unsigned)stream, c3 = (unsigned)(stream >> 32);
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
PFN_DEV float u01(unsigned r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }             // [0,1)
PFN_DEV float u01_open(unsigned r) { return ((float)(r >> 8) + 1.f) * 5.9604644775390625e-8f; }  // (0,1]

// four standard normals from one Philox block (Box-Muller)
PFN_DEV void normal4(const U4& r, float (&n)[4]) {
  const float r0 = sqrtf(-2.f * __logf(u01_open(r.x))), r1 = sqrtf(-2.f * __logf(u01_open(r.z)));
  float s0, c0, s1, c1;
  __sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
  __sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
  n[0] = r0 * c0; n[1] = r0 * s0; n[2] = r1 * c1; n[3] = r1 * s1;
}

}  // namespace pfn
