// msc_dot_gemm.hip -- the Q x M pass on the matrix cores (gfx950 MFMA): the products of the counts and, through their thermometer
// levels, the Manhattan distance.
//
// Of the three integer reductions the all-pairs pass needs per pair (pair_digest.hip), dot(q, c) = sum over bins of q_i c_i for every
// (query, candidate) is the matrix product  D[64 x M] = Q[64 x 4^k] . C[4^k x M]  -- normalized_vectors, pearson, euclidean and simratio
// (predict/Feature.cpp:1171-1184,795-811,1113-1124,829-841) all derive from it. The digest kernel spent a quarter of its arithmetic on it
// (8 of 32 VALU operations per query and 32 bins: v_dot4_u32_u8) while VALU-bound at 86 % busy (r02 counters), with the matrix cores idle.
// Here it is an int8 GEMM: v_mfma_i32_16x16x64_i8, exact in int32 (counts <= 127 and count x sum < 2^31: host-checked per set).
// sum |q_i - c_i| is not bilinear in the counts but is in their levels [count - 1 >= t] (see `level_of` below): with every count of both
// sets <= 16 it comes out of the same pass over the same bytes as a second accumulator, and the digest kernel does not run at all (the
// third reduction, the earth mover's distance, comes from msc_emd_ranks.hip).
//
//   count8 mirror  one byte per bin in the set's own (tile-permuted) bin order -- any order serves a product as long as both
//                  operands share it --, BLOCKED for the B operand: slots in blocks of 16, a block = [64-bin chunk][slot % 16][64 bytes]
//                  + a KiB of padding, so that the 16 candidates x 64 bytes a wave loads per MFMA are ONE contiguous KiB, the chunks of
//                  a block follow each other (the first version read 64 bytes from each of 16 rows 256 KiB apart: 2.1 TB/s) and blocks
//                  are not a power of two apart. 1 byte per bin (1/4 of a 32-bit set), refreshed with the other mirrors' stale range.
//   k_dot_gemm_i8  workgroup = 64 candidates x 64 queries x one slice of the bins; wave w owns candidates 16 w .. 16 w + 15 and
//                  all 64 queries: 4 (+ 4) accumulators of 16 x 16. Candidate bytes go from HBM straight into the B operand registers
//                  (lane l: candidate l % 16, 16 bins of block l / 16: a 64-byte run per candidate and load, the next load takes
//                  the other half of the line); the query tile of a 256-bin step is staged once per workgroup in LDS, 16-byte
//                  segments XOR-swizzled by row so that the 16 rows an A operand touches sit in 16 different bank groups.
//                  Roofline: HBM -- every candidate byte is read once per 64 queries (1 byte per bin): 5.1 ms per 100 000 x 64 at
//                  k = 9 = 5.1 TB/s.
//   output         int32 [slice][candidate][64] per accumulator: the epilogue adds the slices (no atomics: deterministic).
#include "msc_internal.h"
#include "msc_wave.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr uint32_t kStep = 256;          // bins per k-step

// byte offset of bins [at, at + 16) (at a multiple of 16) of `slot` in the blocked mirror
// (a block takes one KiB more than its chunks: blocks 4^k * 16 bytes apart -- a power of two -- would put the same chunk of every block
// on the same HBM channels, and the workgroups of a slice walk their blocks nearly in step)
__host__ __device__ __forceinline__ uint64_t c8_block_bytes(uint64_t nbins) { return ((nbins >> 6) + 1) * 1024; }
__device__ __forceinline__ uint64_t c8_offset(uint64_t slot, uint64_t at, uint64_t nbins) {
	return (slot >> 4) * c8_block_bytes(nbins) + (at >> 6) * 1024 + (slot & 15) * 64 + (at & 63);
}

template <typename T>
__global__ void __launch_bounds__(256) k_count8_build(const T* __restrict__ bins, uint64_t slot_elems, uint8_t* __restrict__ count8, uint64_t first_slot, uint64_t n_slots,
                                                      int32_t* __restrict__ has_zero) {
	const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;          // 16 bins per thread
	if (i >= n_slots * slot_elems) return;
	const uint64_t slot = first_slot + i / slot_elems, at = i % slot_elems;
	const T* src = bins + slot * slot_elems + at;
	uint32_t w[4];
#pragma unroll
	for (int j = 0; j < 4; j++) w[j] = (uint32_t)src[4 * j] | ((uint32_t)src[4 * j + 1] << 8) | ((uint32_t)src[4 * j + 2] << 16) | ((uint32_t)src[4 * j + 3] << 24);
	*reinterpret_cast<uint4*>(count8 + c8_offset(slot, at, slot_elems)) = make_uint4(w[0], w[1], w[2], w[3]);
	bool zero = false;
#pragma unroll
	for (int j = 0; j < 16; j++) zero |= src[j] == 0;
	if (zero) atomicOr(has_zero, 1);
}

// rows of the query operand: q8[r] = the bins of slot q_slots[r] (r < n_q) as one row, zeros for the rows up to 64
__global__ void __launch_bounds__(256) k_gather_rows8(const uint8_t* __restrict__ count8, const uint32_t* __restrict__ q_slots, uint32_t n_q, uint64_t nbins,
                                                      uint8_t* __restrict__ q8) {
	const uint32_t r = blockIdx.y;
	const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
	if (i >= nbins) return;
	uint4 v = make_uint4(0, 0, 0, 0);
	if (r < n_q) v = *reinterpret_cast<const uint4*>(count8 + c8_offset(q_slots[r], i, nbins));
	*reinterpret_cast<uint4*>(q8 + (uint64_t)r * nbins + i) = v;
}

// The thermometer levels of four bins' excess counts e = count - 1 (one byte each): byte b of lv[t - 1] = [e_b >= t], t = 1 .. 2^LB - 1, for
// e < 2^LB (host-checked: the sets' largest count). min(e, e') = sum over t of [e >= t][e' >= t], so the products of the level bytes of two
// histograms, added over levels and bins, are sum min(e_i, e'_i) -- and sum |e_i - e'_i| = sum e + sum e' - 2 sum min: the Manhattan
// distance, the one statistic of the Q x M pass that is not bilinear in the counts, IS bilinear in their levels.
// byte b of the result = [e_b >= T] for the four excess counts e_b < 8 of a word. v_perm_b32 is a byte-wise table lookup: selector bytes
// 0..7 pick from an 8-byte table -- with the excess counts as selectors and the table [j >= T] (j = 0..7), one instruction per level and
// word (bit logic on the three bits of e took 12 per word for 7 levels).
template <int T>
__device__ __forceinline__ v4i level_of(const v4i e) {
	uint32_t lo = 0, hi = 0;
#pragma unroll
	for (int j = 0; j < 4; j++) { lo |= (uint32_t)(j >= T) << (8 * j); hi |= (uint32_t)(j + 4 >= T) << (8 * j); }
	v4i r;
#pragma unroll
	for (int c = 0; c < 4; c++) r[c] = (int)__builtin_amdgcn_perm(hi, lo, (uint32_t)e[c]);
	return r;
}
// the same with the level in a register (the path for tiles that hold repeats: a loop, not seven copies of its body)
__device__ __forceinline__ v4i level_at(const v4i e, uint32_t T) {
	const uint64_t tab = 0x0101010101010101ull << (8 * T);          // byte j = [j >= T]
	v4i r;
#pragma unroll
	for (int c = 0; c < 4; c++) r[c] = (int)__builtin_amdgcn_perm((uint32_t)(tab >> 32), (uint32_t)tab, (uint32_t)e[c]);
	return r;
}

// Excess counts of four bits (LB = 4: counts up to 16): the selector of v_perm_b32 only reaches an 8-byte table, so the low three bits are
// looked up and the fourth decides -- [e >= T] = hi | [lo >= T] below 8, hi at 8, hi & [lo >= T - 8] above.
template <int LB>
__device__ __forceinline__ v4i level_any(const v4i e, uint32_t T) {
	if constexpr (LB < 4) return level_at(e, T);
	else {
		const v4i m7 = {0x07070707, 0x07070707, 0x07070707, 0x07070707}, m1 = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
		const v4i lo = e & m7, hi = (e >> 3) & m1;
		if (T == 8) return hi;
		const v4i l = level_at(lo, T < 8 ? T : T - 8);
		return T < 8 ? (hi | l) : (hi & l);
	}
}

// LB > 0: besides the products of the counts, sum min(e, e') from the level bytes (out_min, same layout): 2^LB - 1 more MFMAs per tile,
// operands derived in registers from the same bytes -- the candidates are still read once.
// NCB: blocks of 16 candidates per wave. The level bytes of a query operand serve every candidate block of the wave (and a candidate
// operand's all four query blocks): with one block a step derived 20 operands for 16 MFMA tiles and the kernel was bound by those
// v_perm_b32, with two it derives 24 for 32.
// (Held to 128 registers -- 4 waves per SIMD instead of 3 -- the compiler spills the operand loads: 12.4 ms against 5.2.)
template <int LB, int NCB>
__global__ void __launch_bounds__(256) k_dot_gemm_i8(const uint8_t* __restrict__ cand8, const uint32_t* __restrict__ cand_slots, uint64_t first, uint32_t m,
                                                     const uint8_t* __restrict__ q8, uint64_t nbins, uint32_t k_slices, int32_t* __restrict__ out,
                                                     int32_t* __restrict__ out_min, bool nt, bool lockstep) {
	constexpr int NL = LB ? (1 << LB) - 1 : 0;
	__shared__ v4i sA[2][64][16];          // [buffer][query row][16-byte segment ^ (row & 15)]: 32 KiB
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t ks = blockIdx.y;
	const uint64_t per = nbins / k_slices, k0 = (uint64_t)ks * per;
	uint32_t ci[NCB];
	bool valid[NCB];
	const uint8_t* brow[NCB];
#pragma unroll
	for (int cb = 0; cb < NCB; cb++) {
		ci[cb] = (blockIdx.x * 4 + wave) * (16 * NCB) + 16 * cb + (lane & 15);
		valid[cb] = ci[cb] < m;
		const uint32_t cc = valid[cb] ? ci[cb] : m - 1;
		const uint64_t slot = cand_slots ? cand_slots[cc] : first + cc;
		// lane l: candidate l % 16 of the block, bytes 16 (l / 16) .. + 15 of every 64-bin chunk: consecutive slots of one block make the
		// wave's load one contiguous KiB
		brow[cb] = cand8 + (slot >> 4) * c8_block_bytes(nbins) + (slot & 15) * 64 + (lane >> 4) * 16;
	}
	const uint32_t arow = tid >> 2, aseg0 = (tid & 3) * 4;
	const uint8_t* asrc = q8 + (uint64_t)arow * nbins + aseg0 * 16;
	v4i acc[NCB][4], acc_min[LB ? NCB : 1][4];
#pragma unroll
	for (int cb = 0; cb < NCB; cb++)
#pragma unroll
		for (int rb = 0; rb < 4; rb++) acc[cb][rb] = v4i{0, 0, 0, 0};
#pragma unroll
	for (int cb = 0; cb < (LB ? NCB : 1); cb++)
#pragma unroll
		for (int rb = 0; rb < 4; rb++) acc_min[cb][rb] = v4i{0, 0, 0, 0};
	v4i a_reg[4], b0[NCB][4], b1[NCB][4];          // plain vectors: HIP's uint4 struct kept these arrays in scratch
	// Rows of the operands a power of two apart (4^k bytes): every workgroup walking its steps in the same order put the whole chip on the
	// same HBM channels at the same time (first version: 1.7 TB/s), so each workgroup started its walk somewhere else (MSC_GEMM_NO_LOCKSTEP
	// still does). Since the blocks of the mirror are a KiB more than a power of two apart the workgroups may walk in step again, and
	// then the queries' tile of a step is hot in L2 for all of them: 4-5 % faster than the scattered walk.
	const uint32_t steps = (uint32_t)(per / kStep);
	const uint32_t rot = lockstep ? 0u : (blockIdx.x * 37u + ks * 11u) % steps;
	// operands of step i (the last step once more past the end: a load nobody uses is cheaper than a branch around it -- with the
	// loads under `if (more)` the compiler parked them in scratch and so waited for each as soon as it was issued). One step ahead is
	// enough: a third register set, two steps ahead, changed nothing (6.58 -> 6.51 ms) -- what held this kernel at 4 TB/s was the queries'
	// side falling out of L2 (msc_dot_gemm_slices).
	auto fetch = [&](uint32_t i, v4i (&b)[NCB][4]) {
		const uint32_t j = (i < steps ? i : steps - 1) + rot;
		const uint64_t k = k0 + (uint64_t)(j >= steps ? j - steps : j) * kStep;
#pragma unroll
		for (int t = 0; t < 4; t++) a_reg[t] = *reinterpret_cast<const v4i*>(asrc + k + 16 * t);
#pragma unroll
		for (int cb = 0; cb < NCB; cb++)
#pragma unroll
			for (int kc = 0; kc < 4; kc++) {
				const v4i* src = reinterpret_cast<const v4i*>(brow[cb] + ((k >> 6) + kc) * 1024);
				b[cb][kc] = nt ? __builtin_nontemporal_load(src) : *src;
			}
	};
	auto park = [&](uint32_t buf) {
#pragma unroll
		for (int t = 0; t < 4; t++) sA[buf][arow][(aseg0 + t) ^ (arow & 15)] = a_reg[t];
	};
	// The levels a tile of candidates does not reach cost nothing: their products are zero whatever the queries hold (min(e, e') <= e), so
	// a wave looks at its candidates' bytes first. No excess count above 1 in the tile (1 kb sequences at k = 9: 99 % of the tiles) --
	// the candidates' level-1 bytes are the excess counts themselves and only the queries' level 1 is derived: 2 MFMAs per tile instead
	// of 2^LB. Otherwise the levels present are taken one by one until one is absent from every candidate of the tile (they nest).
	auto multiply = [&](uint32_t buf, const v4i (&b)[NCB][4]) {
		const v4i one = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
#pragma unroll
		for (int kc = 0; kc < 4; kc++) {
			if constexpr (LB == 0) {
#pragma unroll
				for (int rb = 0; rb < 4; rb++) {
					// A operand: lane l = query 16 rb + l % 16, bins of block l / 16 of this 64-bin chunk
					const v4i A = sA[buf][16 * rb + (lane & 15)][(4 * kc + (lane >> 4)) ^ (lane & 15)];
#pragma unroll
					for (int cb = 0; cb < NCB; cb++) acc[cb][rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, b[cb][kc], acc[cb][rb], 0, 0, 0);
				}
			} else {
				v4i E[NCB];          // excess counts of the candidates (every count >= 1, the ranks mirror's build checked: no borrow between bytes)
				uint32_t above = 0;
#pragma unroll
				for (int cb = 0; cb < NCB; cb++) {
					E[cb] = b[cb][kc] - one;
					above |= (uint32_t)(E[cb].x | E[cb].y | E[cb].z | E[cb].w);
				}
				if (__builtin_amdgcn_ballot_w64((above & 0xfefefefeu) != 0) == 0) {
#pragma unroll
					for (int rb = 0; rb < 4; rb++) {
						const v4i A = sA[buf][16 * rb + (lane & 15)][(4 * kc + (lane >> 4)) ^ (lane & 15)];
						const v4i A1 = LB < 4 ? level_of<1>(A - one) : level_any<LB>(A - one, 1);
#pragma unroll
						for (int cb = 0; cb < NCB; cb++) {
							acc[cb][rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, b[cb][kc], acc[cb][rb], 0, 0, 0);
							acc_min[cb][rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, E[cb], acc_min[cb][rb], 0, 0, 0);
						}
					}
				} else {
					// how many levels the tile reaches (they nest: the first one no candidate byte reaches ends the walk)
					uint32_t reach = 1;
#pragma unroll 1
					for (uint32_t T = 2; T <= (uint32_t)NL; T++) {
						uint32_t any = 0;
#pragma unroll
						for (int cb = 0; cb < NCB; cb++) { const v4i l = level_any<LB>(E[cb], T); any |= (uint32_t)(l.x | l.y | l.z | l.w); }
						if (__builtin_amdgcn_ballot_w64(any != 0) == 0) break;
						reach = T;
					}
#pragma unroll
					for (int rb = 0; rb < 4; rb++) {
						const v4i A = sA[buf][16 * rb + (lane & 15)][(4 * kc + (lane >> 4)) ^ (lane & 15)];
						const v4i AE = A - one;
#pragma unroll
						for (int cb = 0; cb < NCB; cb++) acc[cb][rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, b[cb][kc], acc[cb][rb], 0, 0, 0);
#pragma unroll 1
						for (uint32_t T = 1; T <= reach; T++) {
							const v4i al = level_any<LB>(AE, T);
#pragma unroll
							for (int cb = 0; cb < NCB; cb++) acc_min[cb][rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(al, level_any<LB>(E[cb], T), acc_min[cb][rb], 0, 0, 0);
						}
					}
				}
			}
		}
	};
	fetch(0, b0);
	park(0);
	__syncthreads();
	for (uint32_t i = 0; i < steps; i += 2) {          // two steps per turn: the operand registers and LDS halves swap roles by name
		fetch(i + 1, b1);
		multiply(0, b0);
		park(1);
		__syncthreads();
		if (i + 1 >= steps) break;
		fetch(i + 2, b0);
		multiply(1, b1);
		park(0);
		__syncthreads();
	}
	// D: lane l holds column l % 16 (its candidate), rows 4 (l / 16) .. + 3 of each 16-query block
#pragma unroll
	for (int cb = 0; cb < NCB; cb++) {
		if (!valid[cb]) continue;
		int32_t* o = out + ((uint64_t)ks * m + ci[cb]) * 64 + 4 * (lane >> 4);
#pragma unroll
		for (int rb = 0; rb < 4; rb++) *reinterpret_cast<v4i*>(o + 16 * rb) = acc[cb][rb];
		if constexpr (LB > 0) {
			int32_t* o2 = out_min + ((uint64_t)ks * m + ci[cb]) * 64 + 4 * (lane >> 4);
#pragma unroll
			for (int rb = 0; rb < 4; rb++) *reinterpret_cast<v4i*>(o2 + 16 * rb) = acc_min[cb][rb];
		}
	}
}

}  // namespace

// bytes of the blocked mirror of a set of `capacity` slots
uint64_t msc_count8_bytes(const MscLayout& L, uint64_t capacity) { return (capacity + 15) / 16 * c8_block_bytes(L.padded_bins); }

hipError_t msc_launch_count8_build(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, uint8_t* count8, uint64_t first_slot, uint64_t n_slots, int32_t* has_zero) {
	if (n_slots == 0) return hipSuccess;
	const uint64_t threads = n_slots * L.padded_bins / 16;
	const dim3 grid((unsigned)((threads + 255) / 256));
	if (dtype == 8) k_count8_build<uint8_t><<<grid, dim3(256), 0, st>>>((const uint8_t*)bins, L.padded_bins, count8, first_slot, n_slots, has_zero);
	else if (dtype == 16) k_count8_build<uint16_t><<<grid, dim3(256), 0, st>>>((const uint16_t*)bins, L.padded_bins, count8, first_slot, n_slots, has_zero);
	else if (dtype == 32) k_count8_build<uint32_t><<<grid, dim3(256), 0, st>>>((const uint32_t*)bins, L.padded_bins, count8, first_slot, n_slots, has_zero);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

uint32_t msc_dot_gemm_slices(uint64_t nbins, uint32_t m, int num_cus) {
	// enough workgroups for a few rounds of the chip; a slice is a whole number of 256-bin steps
	uint32_t s = 1;
	while (s < 64 && (uint64_t)((m + 63) / 64) * s < (uint64_t)num_cus * 10 && nbins / (2 * s) >= kStep && nbins % (2 * s * kStep) == 0) s *= 2;
	// and slices short enough that the queries' side of ONE slice (64 rows) stays in an XCD's 4 MiB L2 while the workgroups of that
	// slice -- dispatched together -- walk it in their different orders: with two slices of 8 MiB every step re-read its 16 KiB of queries
	// from beyond L2, as many bytes again as the candidates' (4.0 TB/s of candidate bytes against 5.5 with 1 MiB slices for the plain
	// products; with the level products 8 slices of 2 MiB measure best: 5.21 ms against 5.46 with 16 and 5.84 with 4, and the epilogue
	// adds half as many)
	static const uint64_t a_bytes = [] { const char* e = getenv("MSC_GEMM_A_KIB"); return (uint64_t)(e ? std::max(64, atoi(e)) : 2048) << 10; }();
	while (s < 64 && 64 * (nbins / s) > a_bytes && nbins / (2 * s) >= kStep && nbins % (2 * s * kStep) == 0) s *= 2;
	return s;
}

// dots[slice][candidate][64] of n_q <= 64 queries (rows q_slots of q_count8) against m candidates (slot list, or slots first .. first + m - 1)
// level_bits 2 / 3 / 4 (every excess count below 4 / 8 / 16): out_min[slice][candidate][64] = sum min(e, e') as well
hipError_t msc_launch_dot_gemm(hipStream_t st, uint64_t nbins, const uint8_t* cand_count8, const uint32_t* cand_slots, uint64_t first, uint32_t m,
                               const uint8_t* q_count8, const uint32_t* q_slots_dev, uint32_t n_q, uint8_t* q8_scratch, uint32_t k_slices, int32_t* out,
                               int level_bits, int32_t* out_min) {
	if (m == 0 || n_q == 0) return hipSuccess;
	if (n_q > 64 || nbins % (16 * 16) || nbins % ((uint64_t)k_slices * kStep)) return hipErrorInvalidValue;
	if (level_bits != 0 && (level_bits < 2 || level_bits > 4 || !out_min)) return hipErrorInvalidValue;
	k_gather_rows8<<<dim3((unsigned)((nbins / 16 + 255) / 256), 64), dim3(256), 0, st>>>(q_count8, q_slots_dev, n_q, nbins, q8_scratch);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	static const bool nt = getenv("MSC_GEMM_NT") != nullptr;
	static const bool lockstep = getenv("MSC_GEMM_NO_LOCKSTEP") == nullptr;
	static const int ncb_env = [] { const char* e = getenv("MSC_GEMM_NCB"); return e ? atoi(e) : 0; }();
	const int ncb = ncb_env == 2 ? 2 : 1;          // (two blocks paid while every level was multiplied out; with the levels a tile does not reach skipped, one is faster: 5.25 against 5.44 ms)
	const dim3 grid((m + 64 * ncb - 1) / (64 * ncb), k_slices);
#define MSC_GEMM_GO(LB, NCB) k_dot_gemm_i8<LB, NCB><<<grid, dim3(256), 0, st>>>(cand_count8, cand_slots, first, m, q8_scratch, nbins, k_slices, out, out_min, nt, lockstep)
	if (level_bits == 2) { if (ncb == 2) MSC_GEMM_GO(2, 2); else MSC_GEMM_GO(2, 1); }
	else if (level_bits == 3) { if (ncb == 2) MSC_GEMM_GO(3, 2); else MSC_GEMM_GO(3, 1); }
	else if (level_bits == 4) MSC_GEMM_GO(4, 1);
	else { if (ncb == 2) MSC_GEMM_GO(0, 2); else MSC_GEMM_GO(0, 1); }
#undef MSC_GEMM_GO
	return hipGetLastError();
}
