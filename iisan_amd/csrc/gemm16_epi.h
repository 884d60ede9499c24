// Shared epilogue of the persistent 256x256 GEMMs (gemm16_p256.hip, gemm16_s256.hip): 16-bit outputs go through a
// small per-wave LDS transpose so that every global store instruction writes whole 64-byte row segments.
//
// Why: in the MFMA C layout a lane owns 16 consecutive columns of one row (32 bytes as two 16-byte halves), so a
// wave-wide dwordx4 store touches 32 rows x two 16-byte pieces.  Ablation: those scattered stores
// cost ~20 % of the QKV GEMM although they are asynchronous — the CU's memory pipeline processes them (and the LDS-DMA
// loads queued behind them) piece by piece.  After the transpose a store instruction covers 16 rows x 64 contiguous
// bytes (4 lanes per row): 4x fewer pieces for the same bytes.
#pragma once
#include "common.h"

constexpr int EPI_ROW_BYTES = 144;                        // 128 data bytes (64 columns) + 16 pad
constexpr int EPI_SCRATCH_PER_WAVE = 16 * EPI_ROW_BYTES;  // 2304 B: 16 rows x 64 columns at a time
constexpr int EPI_SCRATCH_BYTES = 8 * EPI_SCRATCH_PER_WAVE;

// One 32x32 accumulator block (v = acc + bias, already in registers: lane (frow, fh) holds row frow, columns 16*fh..+15)
// -> LDS -> global.  `dst_row(r)` returns the output pointer of block row r at the block's first column.
template <typename T, bool GELU, typename RowPtr>
__device__ __forceinline__ void epi_store_block16(const float (&v)[16], char* scratch, int lane, int rows_valid, RowPtr dst_row) {
    typedef typename T::v8 V8;
    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = T::from_f32(GELU ? gelu_erf_fast(v[8 * h2 + e]) : v[8 * h2 + e]);
        *(V8*)(scratch + frow * EPI_ROW_BYTES + fh * 32 + h2 * 16) = o;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int id = p * 64 + lane;
        const int row = id >> 2, c16 = id & 3;
        const V8 o = *(const V8*)(scratch + row * EPI_ROW_BYTES + c16 * 16);
        if (row < rows_valid) *(V8*)(dst_row(row) + c16 * 8) = o;
    }
    __builtin_amdgcn_wave_barrier();
}


// Full-line variant: one 32-row x 64-column strip of a wave (both 32-column accumulator blocks v0 | v1) goes out as
// two passes of 16 rows; after the transpose 8 consecutive lanes hold one row's 128 contiguous bytes, so every store
// instruction writes 8 WHOLE 128-byte lines (partial-line writes make the L2 fetch the line first).
template <typename T, bool GELU, typename RowPtr>
__device__ __forceinline__ void epi_store_strip16(const float (&v0)[16], const float (&v1)[16], char* scratch, int lane,
                                                  int rows_valid, RowPtr dst_row) {
    typedef typename T::v8 V8;
    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((frow >> 4) == half) {
            char* rowp = scratch + (frow & 15) * EPI_ROW_BYTES + fh * 32;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                V8 a, b;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    a[e] = T::from_f32(GELU ? gelu_erf_fast(v0[8 * h2 + e]) : v0[8 * h2 + e]);
                    b[e] = T::from_f32(GELU ? gelu_erf_fast(v1[8 * h2 + e]) : v1[8 * h2 + e]);
                }
                *(V8*)(rowp + h2 * 16) = a;             // columns  0..31 of the strip
                *(V8*)(rowp + 64 + h2 * 16) = b;        // columns 32..63
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int id = p * 64 + lane;
            const int row = id >> 3, c16 = id & 7;
            const V8 o = *(const V8*)(scratch + row * EPI_ROW_BYTES + c16 * 16);
            if (half * 16 + row < rows_valid) *(V8*)(dst_row(half * 16 + row) + c16 * 8) = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
