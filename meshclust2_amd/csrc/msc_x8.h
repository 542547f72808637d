// msc_x8.h -- layout of the x8 mirror (msc_pair_gemm.hip): one byte per bin = min(count - 1, 127), the B operand of the int8 matrix
// product of the Q x M pass. Shared by the kernels that stream it (msc_pair_gemm.hip) and the epilogue that looks single bins up
// (pair_features.hip).
//
// Slots are blocked by 32 (the columns of one v_mfma_i32_32x32x32_i8): block = [32-bin chunk][slot % 32][32 bytes], so the
// 32 candidates x 32 bins one MFMA consumes are ONE contiguous KiB and a wave's operand load (lane l: candidate l % 32, bytes
// 16 (l / 32) .. + 15 of its 32-byte run) is one fully coalesced instruction; a block's chunks follow each other, and a KiB of padding
// keeps blocks from sitting a power of two apart (workgroups walk their blocks in step: with 4^k * 32 bytes between them every wave
// of the chip would be on the same HBM channels at the same time).
// Bin order inside a slot = the set's own tile-permuted order (msc_layout.h): a product only needs both operands to share it, and the
// lists of large bins below name bins by the same physical position.
#pragma once
#include <stdint.h>

#define MSC_X8_CAP 127u      // largest excess count a byte holds (int8 operand); larger ones are clamped and corrected from the lists

__host__ __device__ __forceinline__ uint64_t msc_x8_block_bytes(uint64_t nbins) { return nbins * 32 + 1024; }
// byte offset of bin `bin` of `slot`
__host__ __device__ __forceinline__ uint64_t msc_x8_offset(uint64_t slot, uint64_t bin, uint64_t nbins) {
	return (slot >> 5) * msc_x8_block_bytes(nbins) + (bin >> 5) * 1024 + (slot & 31) * 32 + (bin & 31);
}
