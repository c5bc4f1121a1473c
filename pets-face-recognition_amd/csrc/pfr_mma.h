// pfr_mma.h — the MFMA tile engine shared by the implicit-GEMM conv kernels (fwd / dgrad / wgrad / plain GEMM).
//
// Geometry (identical in BYTES for both dtypes):
//   * an LDS operand tile is ROWS x 64 B of k-values (32 bf16 or 16 f32) per k-step, row stride 80 B
//     (64 + 16 pad → ds_read_b128 by the 16-lane groups of gfx950 is bank-conflict free);
//   * a k-step holds two "k-groups" of 32 B per row; a k-group feeds ONE v_mfma_f32_32x32x16_bf16
//     (lane = row, lane>>5 selects the 16-byte half) or FOUR v_mfma_f32_32x32x2_f32 (element j of both
//     operands' 16-byte chunk is k-slot {j, 4+j}: any k-permutation is legal as long as A and B agree);
//   * 256 threads = 4 waves in a 2x2 grid; wave (wp, wq) owns P-rows [wp*BP/2, +BP/2) x Q-rows [wq*BQ/2, +BQ/2)
//     as (BP/64) x (BQ/64) accumulators of 32x32.  The P operand is the MFMA "A" (accumulator ROW index),
//     the Q operand the MFMA "B" (accumulator COLUMN index = lane & 31).
#pragma once
#include "pfr_common.h"

#define PFR_ROWB 80   // LDS row stride in bytes for operand tiles
#define PFR_KSTEP_BYTES 64

template <typename T> struct KStep;  // elements per k-step
template <> struct KStep<float> { static constexpr int BK = 16; };
template <> struct KStep<bf16_t> { static constexpr int BK = 32; };

template <typename T, int TP, int TQ>
__device__ __forceinline__ void mma_kstep(const char* ldsP, const char* ldsQ, int lane, f32x16 (&acc)[TP][TQ]) {
  // ldsP / ldsQ point at row 0 of THIS WAVE's slice of the operand tiles.
  const int row = lane & 31, half = lane >> 5;
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) {
    u32x4 fp[TP], fq[TQ];
#pragma unroll
    for (int i = 0; i < TP; ++i)
      fp[i] = *reinterpret_cast<const u32x4*>(ldsP + (i * 32 + row) * PFR_ROWB + kg * 32 + half * 16);
#pragma unroll
    for (int j = 0; j < TQ; ++j)
      fq[j] = *reinterpret_cast<const u32x4*>(ldsQ + (j * 32 + row) * PFR_ROWB + kg * 32 + half * 16);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp[i]),
                                                               __builtin_bit_cast(bf16x8, fq[j]), acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fp[i][e]), __uint_as_float(fq[j][e]),
                                                              acc[i][j], 0, 0, 0);
    }
  }
}

// accumulator element (reg r of a 32x32 tile) -> row within the tile; column is lane & 31.
// 16-byte buffer store whose data registers stay untouched until the hardware has READ them.  Measured on gfx950: a
// buffer_store_dwordx4 reads its data VGPRs from the register file late when the wave's vector-memory queue is deep (tens of LDS-DMA /
// store instructions in flight); hipcc assumes store data is read at issue and re-uses the registers in the very next instruction
// (an LDS read of the next pass, or just an address add that the scheduler moved up — a separate `s_waitcnt expcnt(0)` statement
// after the store does NOT stop that), so the last quad lanes of the first data dword went out overwritten.  EXP_CNT tracks the
// read-out: the store and the wait are ONE asm statement here, nothing can be scheduled between them.  (s_nop: soff may come
// straight from the scalar ALU and nothing inside an asm statement is hazard-padded by the compiler.)
__device__ __forceinline__ void buffer_store_b128_sync(u32x4 v, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
#ifdef PFR_STORE_SPLIT_AB   // A/B builds only: the earlier (fragile) form, builtin store + separate wait
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)voff, (int)soff, 0);
  asm volatile("s_waitcnt expcnt(0)" ::: "memory");
  return;
#endif
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_waitcnt expcnt(0)" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// one byte per lane, same discipline (the block tail's ReLU bit masks)
__device__ __forceinline__ void buffer_store_byte_sync(uint32_t v, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  asm volatile("s_nop 4\n\tbuffer_store_byte %0, %1, %2, %3 offen\n\ts_waitcnt expcnt(0)" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ------------------------------------------------------------------------------------------------
// v2 operand tiles: UNPADDED rows of KCH 16-byte chunks of k-values per k-step (KCH = 8: 128 B = 64 bf16 / 32 f32;
// KCH = 4: 64 B), chunk position XOR-swizzled with row bits:
//     KCH = 8: phys_chunk = chunk ^ ((row >> 1) & 7)        KCH = 4: phys_chunk = chunk ^ ((row >> 2) & 3)
// Unpadded rows are what the LDS-DMA path (buffer_load … lds: wave-uniform base + lane*16) needs; the swizzle makes the
// 16-lane groups of ds_read_b128 (rows {0-3,12-15,20-27}+k) hit 16 distinct 4-bank slots (conflict free), and it is
// applied on the SOURCE address of the DMA (each lane group still reads one whole contiguous row segment).
template <int KCH> __device__ __forceinline__ int row_swizzle(int row) {
  return KCH == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

struct NoMid { __device__ __forceinline__ void operator()() const {} };
// a Mid with `static constexpr bool every_kgroup = true` is invoked after EVERY k-group's MFMAs with the k-group index (spread DMA issue)
template <typename M, typename = void> struct MidEvery { static constexpr bool value = false; };
template <typename M> struct MidEvery<M, decltype((void)M::every_kgroup)> { static constexpr bool value = M::every_kgroup; };
template <typename F> struct EveryKg {
  F f;
  static constexpr bool every_kgroup = true;
  __device__ __forceinline__ void operator()(int kg) const { f(kg); }
};
template <typename F> __device__ __forceinline__ EveryKg<F> every_kg(F f) { return EveryKg<F>{f}; }

// `mid` is invoked after the MFMAs of the first k-group have been ISSUED: work placed there (the next tile's DMA issue and
// its address arithmetic) executes while the matrix pipe drains those MFMAs instead of in front of an idle pipe.
template <typename T, int KCH, int TP, int TQ, typename Mid = NoMid>
__device__ __forceinline__ void mma_kstep_sw(const char* ldsP, const char* ldsQ, int lane, f32x16 (&acc)[TP][TQ], Mid mid = Mid(), int midkg = 0) {
  // ldsP / ldsQ: row 0 of this wave's slice (slice bases are multiples of 16 rows, so the swizzle depends on lane only).
  // Fragments of k-group kg+1 are read while the MFMAs of k-group kg run (explicit register double buffering).
  constexpr int ROWB = KCH * 16;
  constexpr int NKG = KCH / 2;
  const int row = lane & 31, half = lane >> 5;
  const int sw = row_swizzle<KCH>(row);
  u32x4 fp[2][TP], fq[2][TQ];
  auto rd = [&](int kg, int b) {
    const int off = (((kg * 2 + half) ^ sw) << 4);
#pragma unroll
    for (int i = 0; i < TP; ++i) fp[b][i] = *reinterpret_cast<const u32x4*>(ldsP + (i * 32 + row) * ROWB + off);
#pragma unroll
    for (int j = 0; j < TQ; ++j) fq[b][j] = *reinterpret_cast<const u32x4*>(ldsQ + (j * 32 + row) * ROWB + off);
  };
  rd(0, 0);
#pragma unroll
  for (int kg = 0; kg < NKG; ++kg) {
    const int b = kg & 1;
    if (kg + 1 < NKG) rd(kg + 1, b ^ 1);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp[b][i]),
                                                               __builtin_bit_cast(bf16x8, fq[b][j]), acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fp[b][i][e]), __uint_as_float(fq[b][j][e]),
                                                              acc[i][j], 0, 0, 0);
    }
    if constexpr (MidEvery<Mid>::value) mid(kg);
    else if (kg == midkg) mid();
  }
}
