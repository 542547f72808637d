// msc_wave.h -- wavefront-level device primitives (gfx950, wave64) shared by the pair kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

// ---------------------------------------------------------------------------------------- wave primitives (DPP)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
	return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// inclusive prefix sum over the 64 lanes: row_shr 1,2,4,8 then row_bcast 15 / 31 (gfx9 DPP)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
	v = dpp_add<0x111, 0xf>(v);
	v = dpp_add<0x112, 0xf>(v);
	v = dpp_add<0x114, 0xf>(v);
	v = dpp_add<0x118, 0xf>(v);
	v = dpp_add<0x142, 0xa>(v);
	v = dpp_add<0x143, 0xc>(v);
	return v;
}
__device__ __forceinline__ uint32_t wave_total_u32(uint32_t v) {      // valid in every lane (SGPR broadcast)
	return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63);
}
// per-lane values < 2^32 whose wave total may exceed 32 bits: add the 16-bit halves separately
__device__ __forceinline__ uint64_t wave_total_u64(uint32_t v) {
	const uint32_t lo = wave_total_u32(v & 0xffffu);
	const uint32_t hi = wave_total_u32(v >> 16);
	return (uint64_t)lo + ((uint64_t)hi << 16);
}

// |a - b| + c in ONE VALU op. hipcc lowers __usad to v_max/v_min/v_sub/v_add, so the instruction is named explicitly.
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c) {
	uint32_t d;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Arithmetic on the PACKED 32-bit words a lane holds after its 16-byte loads. NW words cover the lane's R bins in
// logical order; bins are never unpacked for the order-independent reductions (v_sad_u8/u16, v_dot4/dot2 work on the
// packed word), only the prefix statistic touches single bins (byte/half selects fold into SDWA operands).
// Four wave sums for the price of ~1.7: the two gfx950 lane-swap instructions fold the values pairwise ("transposed"
// reduction), so one register ends up holding all four results in its four 16-lane rows.
//   v_permlane32_swap a, b : a[32..63] <-> b[0..31]      -> a+b = [a_lo+a_hi | b_lo+b_hi]
//   v_permlane16_swap x, y : x rows 1,3 <-> y rows 0,2   -> x+y = rows [x0+x1 | y0+y1 | x2+x3 | y2+y3]
// then 4 row_shr DPP adds leave each row's total in its last lane: lane 15 = sum(a), 31 = sum(c), 47 = sum(b), 63 = sum(d).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t wave_sum4_rows(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
	const u32x2 ab = __builtin_amdgcn_permlane32_swap(a, b, false, false);
	const u32x2 cd = __builtin_amdgcn_permlane32_swap(c, d, false, false);
	const uint32_t x = ab.x + ab.y, y = cd.x + cd.y;
	const u32x2 xy = __builtin_amdgcn_permlane16_swap(x, y, false, false);
	uint32_t v = xy.x + xy.y;
	v = dpp_add<0x111, 0xf>(v);
	v = dpp_add<0x112, 0xf>(v);
	v = dpp_add<0x114, 0xf>(v);
	v = dpp_add<0x118, 0xf>(v);
	return v;
}
#define MSC_ROW_A(v) ((uint32_t)__builtin_amdgcn_readlane((int)(v), 15))
#define MSC_ROW_C(v) ((uint32_t)__builtin_amdgcn_readlane((int)(v), 31))
#define MSC_ROW_B(v) ((uint32_t)__builtin_amdgcn_readlane((int)(v), 47))
#define MSC_ROW_D(v) ((uint32_t)__builtin_amdgcn_readlane((int)(v), 63))

__device__ __forceinline__ uint64_t shfl_sum_u64_early(uint64_t v) {
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace
