// msc_kbits.h -- layout of the k-mer presence mirror (msc_pair_gemm.hip): ONE BIT per bin = [count >= 2], i.e. "this k-mer occurs in the
// sequence" -- the B operand of the int8 matrix product of the Q x M pass, expanded to bytes in registers. 32 KiB per histogram at
// k = 9 where the bins themselves take 1 MiB (uint32_t): BASELINE cfg2's 100 000 histograms stream as 3.3 GB.
//
// Slots are blocked by 32 (the columns of one v_mfma_i32_32x32x32_i8): block = [256-bin super-step][slot % 32][32 bytes] + a KiB of
// padding, so the 16 bytes a lane loads per super-step (lane l: candidate l % 32, half l / 32) make one contiguous KiB per wave and
// blocks do not sit a power of two apart. Inside a slot's 32 bytes of a super-step: 16 halfwords, halfword 8 h + j = the bits of
// bins 32 j + 16 h .. + 15 of the super-step (bit i = bin 32 j + 16 h + i) -- the 16 bins lane half h feeds MFMA k-chunk j.
// Bin order inside a slot = the set's own tile-permuted order (msc_layout.h): a product only needs both operands to share it, and the
// lists of large bins name bins by the same physical position.
#pragma once
#include <stdint.h>

// The queries' transposed image of a block (k_kb_gather): per bin two planes of qn bits (each padded to 16 bytes): plane 0 = the query's
// presence bit, plane 1 = "its excess count is >= 2" (the exact count then sits in the query's list of large bins).
// -> index of the 32-bit word that holds row `row` of plane `plane` of bin `bin`; its bit: row & 31
__host__ __device__ __forceinline__ uint64_t msc_qt_word(uint64_t bin, uint32_t plane, uint32_t row, uint32_t qn) {
	const uint32_t wpp = qn < 128 ? 4 : qn / 32;          // words per plane
	return (bin * 2 + plane) * wpp + (row >> 5);
}

__host__ __device__ __forceinline__ uint64_t msc_kb_block_bytes(uint64_t nbins) { return nbins * 4 + 1024; }
// byte offset of the halfword that holds bin `bin` of `slot` (its bit: bin & 15)
__host__ __device__ __forceinline__ uint64_t msc_kb_offset(uint64_t slot, uint64_t bin, uint64_t nbins) {
	return (slot >> 5) * msc_kb_block_bytes(nbins) + (bin >> 8) * 1024 + (slot & 31) * 32 + (8 * ((bin >> 4) & 1) + ((bin >> 5) & 7)) * 2;
}
