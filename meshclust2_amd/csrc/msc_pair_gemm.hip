// msc_pair_gemm.hip -- the Q x M pass on the matrix cores (gfx950 MFMA): ONE FP4 matrix product per tile of bins over ONE BIT per bin,
// exact for ANY counts.
//
// The pass needs three integer reductions per (query q, candidate c) (pair_features.hip): sum |q_i - c_i| (manhattan, intersection,
// kulczynski2: predict/Feature.cpp:859-871,764-777,682-695), sum q_i c_i (euclidean, normalized_vectors, pearson, simratio:
// :1113-1124,1171-1184,795-811,829-841) and the earth mover's distance (:1505-1518; msc_emd_ranks.hip). With excess counts
// e = count - 1 (every bin starts at the pseudocount 1, nonltr/KmerHashTable.cpp:69-72):
//     sum |q_i - c_i| = sum e_q + sum e_c - 2 sum min(e_q, e_c)          sum q_i c_i = 4^k + sum e_q + sum e_c + sum e_q e_c
// An L-base sequence touches at most L of the 4^k bins, almost all of them ONCE (a 1 kb sequence at k = 9 has ~2 bins with e >= 2), and
// where both excesses are 0 or 1, min(e_q, e_c) = e_q e_c = f_q f_c with the presence bits f = [e >= 1]. Write e = f + g (g = e - 1
// on the "large bins" e >= 2, else 0). Then, exactly:
//     P1(q, c) = sum_i f_q(i) f_c(i)        the number of shared k-mers: a matrix product over bits
//     sum min(e_q, e_c) = P1 + sum over bins large in BOTH of [ min(e_q, e_c) - 1 ]
//     sum e_q e_c       = P1 + sum_i g_q(i) f_c(i)  +  sum over the candidate's large bins of g_c e_q
//                              ^ P2: the queries' large bins      ^ walked by the epilogue
// The corrections run over SHORT LISTS, not over bins: per slot the (bin, e) pairs with e >= 2 (~2 per 1 kb sequence at k = 9, one more per
// repeat unit), kept beside the mirror. The candidate's list is walked by the epilogue (k_pair_epilogue_bits* in pair_features.hip: per entry
// one read of the queries' bits at that bin); the queries' lists become a per-step "hot list" the product consults while the candidates'
// bits of that step sit in its registers (P2: one atomic add per (entry, candidate that holds the k-mer) -- 0.4 % of the candidates at
// k = 9). Exact in integers for any counts of the narrow range (<= 8191).
//
//   kb mirror            msc_kbits.h: a bit per bin, slots blocked by 32; + the lists of large bins (mb, pitch entries per slot, unordered)
//   k_kb_gather          the queries' side of a block of <= QN queries: their bits as the nibble tiles the product copies into LDS, and
//                        transposed as two bit planes per bin ("the query holds this k-mer", "... more than once") for the epilogue
//   k_hot_*              the queries' large bins bucketed by 128-bin step: (bin, query row, e - 1); the fill also sets the second plane
//   k_pair_gemm_fp4_dma  the product on v_mfma_f32_32x32x64_f8f6f4 with E2M1 operands: workgroup = 128 candidates x QN queries, wave = 32
//                        candidates x QN queries (QN / 32 accumulators of 32 x 32); both operands arrive in LDS by LDS-DMA. Roofline: the
//                        FP4 matrix pipe (~10 POPS dense at 2.4 GHz; 8.7 measured back to back, tools/ubench/mfma_rate.hip) --
//                        2 x 4^k operations per pair; HBM sees 4^k / 8 bytes per candidate and block of QN queries.
//   output               int32 P1 [slice][candidate][QN] (plain stores; the tiles a whole number of rounds of the chip leaves over add
//                        theirs piece by piece), int32 P2 [candidate][QN] (atomics)
// r03 - r04 forms that are gone (git history): thermometer levels over a byte per bin (counts <= 16 only), the int8 product
// (k_pair_gemm_bits, 2.25 ms per block), the FP4 product with the queries' tile expanded from bits by every workgroup (k_pair_gemm_fp4:
// no slower alone, slower beside the tail kernels), its 64-candidate-per-wave deal, 8-wave workgroups, 256-row blocks and the
// work-dealing grid (k_pair_gemm_fp4_sk) -- every one measured slower than what stayed (profiles/r04_notes.md section 1).
#include "msc_internal.h"
#include "msc_wave.h"
#include "msc_kbits.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr uint32_t kStep = 128;          // bins per step of the queries' tile; the candidates' bits are loaded per 256 (two steps)

// ------------------------------------------------------------------------------------------------ the mirror and its lists
// 16 bins per thread = one halfword of the mirror. flags[0] |= 1 when a zero count is met (the identities above need every count >= 1:
// KmerHashTable's initial value, and a mean of such histograms too); flags[1] = the longest list seen (the host re-lays the lists out
// when it passes the pitch).
template <typename T>
__global__ void __launch_bounds__(256) k_kb_build(const T* __restrict__ bins, uint64_t nbins, uint8_t* __restrict__ kb, uint64_t first_slot, uint64_t n_slots,
                                                  uint2* __restrict__ mb, uint32_t* __restrict__ mb_n, uint32_t pitch, int32_t* __restrict__ flags) {
	const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
	if (i >= n_slots * nbins) return;
	const uint64_t slot = first_slot + i / nbins, at = i % nbins;
	constexpr int NV = sizeof(T);          // 16-byte loads for 16 bins
	uint32_t raw[4 * NV];
	const uint4* src = reinterpret_cast<const uint4*>(bins + slot * nbins + at);
#pragma unroll
	for (int v = 0; v < NV; v++) { const uint4 q = src[v]; raw[4 * v] = q.x; raw[4 * v + 1] = q.y; raw[4 * v + 2] = q.z; raw[4 * v + 3] = q.w; }
	uint32_t hw = 0;
	bool zero = false;
#pragma unroll
	for (int j = 0; j < 16; j++) {
		uint32_t c;
		if constexpr (sizeof(T) == 1) c = (raw[j >> 2] >> (8 * (j & 3))) & 0xffu;
		else if constexpr (sizeof(T) == 2) c = (raw[j >> 1] >> (16 * (j & 1))) & 0xffffu;
		else c = raw[j];
		if (c == 0) { zero = true; continue; }
		const uint32_t e = c - 1;
		if (e) hw |= 1u << j;
		if (e >= 2) {
			const uint32_t pos = atomicAdd(&mb_n[slot], 1u);
			if (pos < pitch) mb[slot * pitch + pos] = make_uint2((uint32_t)at + j, e);
			atomicMax(&flags[1], (int32_t)(pos + 1));
		}
	}
	*reinterpret_cast<uint16_t*>(kb + msc_kb_offset(slot, at, nbins)) = (uint16_t)hw;
	if (zero) atomicOr(&flags[0], 1);
}

// ------------------------------------------------------------------------------------------------ the queries' side of a block
// One workgroup per 128-bin step, one thread per query row (rows past n_q are zero). anib = the queries' tile of each 256-bin super-step
// as k_pair_gemm_fp4_dma copies it into LDS: [super-step][row][16-byte segment (2 t + h) ^ ((row >> 1) & 7)] = the nibbles lane half h
// feeds the super-step's product t. qT = the queries' presence transposed as BITS, see below (r04: a byte per (bin, row), 33.5 MB per
// block of 128 at k = 9 and most of this kernel's 0.22 ms; now 2 x 16 bytes per bin).
template <int QN>
__global__ void __launch_bounds__(256) k_kb_gather(const uint8_t* __restrict__ kb, const uint32_t* __restrict__ q_slots, uint32_t n_q, uint64_t nbins,
                                                   uint8_t* __restrict__ qT, uint8_t* __restrict__ anib) {
	__shared__ uint32_t qbits[kStep * (QN / 32)];          // [bin of the step][word of 32 rows]
	const uint32_t step = blockIdx.x, row = threadIdx.x;
	uint32_t hwv_keep[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	if (row < QN) {
		uint32_t hwv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		if (row < n_q) {
			// the mirror keeps a 256-bin super-step as [h][8 halfwords j]: this step is j = 4 (step & 1) .. + 3 of both halves
			const uint8_t* src = kb + msc_kb_offset(q_slots[row], (uint64_t)(step >> 1) * 256, nbins) + 8 * (step & 1);
			const v2i lo = *reinterpret_cast<const v2i*>(src), hi = *reinterpret_cast<const v2i*>(src + 16);
#pragma unroll
			for (int kc = 0; kc < 4; kc++) {
				hwv[2 * kc] = ((uint32_t)lo[kc >> 1] >> (16 * (kc & 1))) & 0xffffu;
				hwv[2 * kc + 1] = ((uint32_t)hi[kc >> 1] >> (16 * (kc & 1))) & 0xffffu;
			}
		}
		{          // the queries' tile as k_pair_gemm_fp4_dma copies it into LDS: nibbles, segments already swizzled
			v4i* tile_row = reinterpret_cast<v4i*>(anib) + ((uint64_t)(step >> 1) * QN + row) * 8;
#pragma unroll
			for (uint32_t x = 0; x < 4; x++) {          // x = 2 (t & 1) + h of this step's two products
				const uint32_t h = x & 1, t = 2 * (step & 1) + (x >> 1), w = hwv[4 * (x >> 1) + h] | (hwv[4 * (x >> 1) + 2 + h] << 16);
				tile_row[(2 * t + h) ^ ((row >> 1) & 7)] = v4i{(int)((w << 2) & 0x44444444u), (int)(w & 0x22222222u), (int)((w >> 2) & 0x11111111u), (int)((w >> 1) & 0x44444444u)};
			}
		}
#pragma unroll
		for (int sg = 0; sg < 8; sg++) hwv_keep[sg] = hwv[sg];
	}
	// transposed, as BITS: bin p of the step (halfword p / 16 of the row, bit p % 16) -> one ballot per bin and wave of 64 rows; lane l of a
	// wave keeps the ballots of bins l and l + 64. Record of a bin = [plane][QN / 32 words]: plane 0 = "the query holds this k-mer"
	// (e >= 1), plane 1 = "... more than once" (e >= 2: set by k_hot_fill; the epilogue then reads the count in the query's own list).
	if (row < ((QN + 63u) & ~63u)) {          // (whole waves: a lane past QN holds zeros and keeps the ballots of its two bins like any other)
		const uint32_t lane = row & 63, wv = row >> 6;
		uint32_t lo[2] = {0, 0}, hi[2] = {0, 0};
#pragma unroll
		for (uint32_t pth = 0; pth < kStep; pth++) {
			const unsigned long long b = __builtin_amdgcn_ballot_w64((hwv_keep[pth >> 4] >> (pth & 15)) & 1u);
			// (a select, not v_writelane from inline asm: the compiler pads no wait states between the compare that writes the ballot's
			// scalar registers and an asm statement that reads them, and the lane then keeps the PREVIOUS bin's ballot now and then)
			const bool mine = lane == (pth & 63);
			lo[pth >> 6] = mine ? (uint32_t)b : lo[pth >> 6];
			hi[pth >> 6] = mine ? (uint32_t)(b >> 32) : hi[pth >> 6];
		}
#pragma unroll
		for (uint32_t hf = 0; hf < 2; hf++) {
			qbits[(64 * hf + lane) * (QN / 32) + 2 * wv] = lo[hf];
			if (QN >= 64) qbits[(64 * hf + lane) * (QN / 32) + 2 * wv + 1] = hi[hf];
		}
	}
	__syncthreads();
	// one 16-byte piece per thread: (bin, plane, piece) in the record's own order; plane 1 leaves here as zeros
	constexpr uint32_t PPB = QN / 128 ? QN / 128 : 1;          // 16-byte pieces per plane of a record (QN < 128: the record is padded to 16 bytes per plane)
	for (uint32_t it = threadIdx.x; it < kStep * 2 * PPB; it += 256) {
		const uint32_t pbin = it / (2 * PPB), plane = (it / PPB) & 1, piece = it % PPB;
		uint4 v = make_uint4(0, 0, 0, 0);
		if (plane == 0) {
			const uint32_t* src = qbits + pbin * (QN / 32) + 4 * piece;
			v.x = src[0];
			if (QN >= 64) v.y = src[1];
			if (QN >= 128) { v.z = src[2]; v.w = src[3]; }
		}
		*reinterpret_cast<uint4*>(qT + (((uint64_t)step * kStep + pbin) * 2 * PPB + plane * PPB + piece) * 16) = v;
	}
}

// The queries' large bins, bucketed by step: count, exclusive scan, fill. An entry = (bin, row << 16 | e - 1); the fill also puts
// min(e, 127) at [bin][row] of the transposed image (behind k_kb_gather on the stream).
__global__ void __launch_bounds__(256) k_hot_count(const uint2* __restrict__ mb, const uint32_t* __restrict__ mb_n, uint32_t pitch, const uint32_t* __restrict__ q_slots,
                                                   uint32_t n_q, uint32_t* __restrict__ cnt) {
	const uint32_t row = blockIdx.x;
	if (row >= n_q) return;
	const uint32_t slot = q_slots[row];
	const uint32_t n = mb_n[slot] < pitch ? mb_n[slot] : pitch;
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&cnt[mb[(uint64_t)slot * pitch + i].x / kStep], 1u);
}
__global__ void __launch_bounds__(1024) k_hot_scan(const uint32_t* __restrict__ cnt, uint32_t nsteps, uint32_t* __restrict__ ptr, uint32_t* __restrict__ cursor) {
	__shared__ uint32_t part[1024];
	const uint32_t per = (nsteps + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < nsteps ? lo + per : nsteps;
	uint32_t s = 0;
	for (uint32_t i = lo; i < hi; i++) s += cnt[i];
	part[threadIdx.x] = s;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) {
		const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	uint32_t run = part[threadIdx.x] - s;
	for (uint32_t i = lo; i < hi; i++) { ptr[i] = run; cursor[i] = run; run += cnt[i]; }
	if (threadIdx.x == 1023) ptr[nsteps] = part[1023];
}
__global__ void __launch_bounds__(256) k_hot_fill(const uint2* __restrict__ mb, const uint32_t* __restrict__ mb_n, uint32_t pitch, const uint32_t* __restrict__ q_slots,
                                                  uint32_t n_q, uint32_t qn, uint32_t* __restrict__ cursor, uint2* __restrict__ hot, uint8_t* __restrict__ qT) {
	const uint32_t row = blockIdx.x;
	if (row >= n_q) return;
	const uint32_t slot = q_slots[row];
	const uint32_t n = mb_n[slot] < pitch ? mb_n[slot] : pitch;
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
		const uint2 en = mb[(uint64_t)slot * pitch + i];
		const uint32_t g = en.y - 1 > 0xffffu ? 0xffffu : en.y - 1;          // (counts of the narrow range are <= 8191)
		hot[atomicAdd(&cursor[en.x / kStep], 1u)] = make_uint2(en.x, (row << 16) | g);
		atomicOr(reinterpret_cast<uint32_t*>(qT) + msc_qt_word(en.x, 1, row, qn), 1u << (row & 31));
	}
}

// ------------------------------------------------------------------------------------------------ the product
// v_mfma_f32_32x32x64_f8f6f4 with both operands in FP4 (E2M1): 64 bins per instruction (MI355X: ~10 POPS dense), and an expansion of bits
// that costs a v_and per EIGHT bins: the product only needs both operands to agree on which bin sits in which K position, so nibble i of
// operand dword d takes bit 4 i + d of the 32-bit word -- a mask. The nibble then holds 1, 2, 4 (or, shifted down, 1): as E2M1 0.5, 1.0, 2.0. The candidates' side uses
// (0.5, 1, 2, 0.5) for d = 0 .. 3 and the queries' side (2, 1, 0.5, 2), so every product of two set bits is exactly 1.0; the sums (at most
// the number of k-mers of a sequence) are exact in the f32 accumulators and are converted to int32 on the way out. (tools/probes/
// fp4_mfma_probe.hip checks the identity D[i][j] = popcount(a_i & b_j) on the card; the non-scaled opcode -- both scale operands the
// constant 0 -- scales by 1.) One tile step = 256 bins = one 16-byte load of a candidate = four products per row block.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr uint32_t kM1 = 0x11111111u, kM2 = 0x22222222u, kM4 = 0x44444444u;

// The queries' tile comes into LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU) from the image k_kb_gather wrote once per
// block in the tile's own byte order (nibbles, swizzled segments). A wave copies its QN / NW rows of the next super-step (1 KiB per
// instruction) while the current one is multiplied.
__device__ __forceinline__ void lds_dma_1k(const uint8_t* lane_src, uint32_t lds_wave_base) {
	uint32_t keep;      // m0 is the compiler's: saved and restored around the copy
	asm volatile(
	    "s_mov_b32 %0, m0\n\t"
	    "s_mov_b32 m0, %2\n\t"
	    "s_nop 0\n\t"
	    "global_load_lds_dwordx4 %1, off\n\t"
	    "s_mov_b32 m0, %0"
	    : "=&s"(keep)
	    : "v"(lane_src), "s"(lds_wave_base)
	    : "memory");
}

// the same with a wave-uniform 64-bit base in scalar registers + a 32-bit offset per lane
__device__ __forceinline__ void lds_dma_1k_s(uint64_t sbase, uint32_t lane_off, uint32_t lds_wave_base) {
	uint32_t keep;
	asm volatile(
	    "s_mov_b32 %0, m0\n\t"
	    "s_mov_b32 m0, %3\n\t"
	    "s_nop 0\n\t"
	    "global_load_lds_dwordx4 %1, %2\n\t"
	    "s_mov_b32 m0, %0"
	    : "=&s"(keep)
	    : "v"(lane_off), "s"(sbase), "s"(lds_wave_base)
	    : "memory");
}

// The candidates' 16 bytes per lane and super-step come by LDS-DMA too, into a ring of kBDepth KiB per wave, kBDepth - 1 super-steps
// ahead: a register destination can only be one super-step ahead (the wait in front of the barrier would have to let exactly that load
// through, and a register still in flight cannot be handed on), and one super-step -- 0.2-0.5 us -- is less than a trip to HBM.
// Order of issue per super-step: the tile's pieces, THEN the candidates' KiB; vector-memory operations retire in order, so
// s_waitcnt vmcnt(1) in front of the barrier lets only that youngest copy stay in flight.
template <int NRB, int NW>
__global__ void __launch_bounds__(64 * NW, 3) k_pair_gemm_fp4_dma(const uint8_t* __restrict__ cand_kb, const uint32_t* __restrict__ cand_slots, uint64_t first, uint32_t m,
                                                                         const uint8_t* __restrict__ anib, uint64_t nbins, uint32_t k_slices, const uint32_t* __restrict__ hot_ptr,
                                                                         const uint2* __restrict__ hot, int32_t* __restrict__ out_min, int32_t* __restrict__ out_diff, uint32_t n_whole,
                                                                         uint32_t piece_ss) {
	constexpr int QN = 32 * NRB;
	constexpr uint32_t kTile = QN * 128;              // bytes of the queries' tile of a super-step
	constexpr uint32_t kPieces = kTile / NW / 1024;   // KiB a wave copies per super-step
	constexpr uint32_t kBDepth = 4;                   // ring slots of the candidates' bits per wave
	static_assert(kPieces >= 1 && kPieces * NW * 1024 == kTile, "tile copy");
	__shared__ v4i sA[2][QN * 8];
	__shared__ v4i sB[NW][kBDepth][64];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// Workgroups 0 .. n_whole - 1 take a whole tile of 128 candidates each (one slice of the bins: blockIdx.y). With piece_ss > 0 the tiles
	// past n_whole -- what is left when the tiles do not fill the chip's places a whole number of times -- are cut into pieces of piece_ss
	// super-steps, one workgroup each, which ADD their sums into rows the host zeroed: 782 tiles on 768 places ran a second round of 14
	// workgroups on an otherwise idle chip (1.60 ms against 1.06 for 768 tiles); as ~730 short pieces they are one more twentieth of a round.
	const uint32_t ks = blockIdx.y;
	const uint64_t per = nbins / k_slices;
	uint32_t tile = blockIdx.x, n_ss = (uint32_t)(per / 256), gss0 = (uint32_t)((uint64_t)ks * per / 256);
	const bool piece = piece_ss && blockIdx.x >= n_whole;
	if (piece) {
		const uint32_t ppt = (n_ss + piece_ss - 1) / piece_ss, p = blockIdx.x - n_whole;
		tile = n_whole + p / ppt;
		gss0 = (p % ppt) * piece_ss;
		n_ss = n_ss - gss0 < piece_ss ? n_ss - gss0 : piece_ss;
	}
	const uint32_t ci = (tile * NW + wave) * 32 + (lane & 31);
	const bool valid = ci < m;
	const uint32_t cc = valid ? ci : m - 1;
	const uint64_t slot = cand_slots ? cand_slots[cc] : first + cc;
	const uint8_t* brow = cand_kb + (slot >> 5) * msc_kb_block_bytes(nbins) + (slot & 31) * 32 + (lane >> 5) * 16 + (uint64_t)gss0 * 1024;
	const uint64_t abase = (uint64_t)(anib + (uint64_t)gss0 * kTile + __builtin_amdgcn_readfirstlane(wave) * (kPieces * 1024));          // wave-uniform: the pieces' base in scalar registers
	const uint32_t aoff = lane * 16;
	const uint32_t lds_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&sA[0][0] + wave * (kPieces * 1024));
	const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&sB[wave][0][0]);
	v16f acc[NRB];
#pragma unroll
	for (int rb = 0; rb < NRB; rb++)
#pragma unroll
		for (int i = 0; i < 16; i++) acc[rb][i] = 0.f;
	// (a tile piece = scalar base + one 32-bit offset per lane: no 64-bit vector adds per piece; 0.99 against 1.05 ms per block)
	auto copy_a = [&](uint32_t buf, uint32_t ss) {
		const uint64_t pv = abase + (uint64_t)(ss < n_ss ? ss : n_ss - 1) * kTile;
		const uint64_t p = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pv >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);          // (wave-uniform by construction; said so)
#pragma unroll
		for (uint32_t i = 0; i < kPieces; i++) lds_dma_1k_s(p + i * 1024, aoff, lds_a + buf * kTile + i * 1024);
	};
	// a lane's source is its own candidate's 16 bytes: one wave-instruction gathers the 64 of them into one KiB of the ring
	auto copy_b = [&](uint32_t ss) { lds_dma_1k(brow + (uint64_t)(ss < n_ss ? ss : n_ss - 1) * 1024, lds_b + (ss % kBDepth) * 1024); };
	// prologue: the candidates' bits of the first kBDepth - 1 super-steps, the first tile; everything has landed behind the wait
#pragma unroll
	for (uint32_t d = 0; d + 1 < kBDepth; d++) copy_b(d);
	copy_a(0, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	for (uint32_t ss = 0; ss < n_ss; ss++) {
		const uint32_t buf = ss & 1;
		copy_a(buf ^ 1, ss + 1);          // (the buffer the previous super-step read: every wave is past that super-step's barrier)
		copy_b(ss + kBDepth - 1);         // (the slot super-step ss - 1 read)
		const v4i bcur = sB[wave][ss % kBDepth][lane];
		{
			const uint32_t r = lane & 31, sw = (r >> 1) & 7, hh = lane >> 5;
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const uint32_t w = (uint32_t)bcur[t];
				const v8i B = {(int)(w & kM1), (int)(w & kM2), (int)(w & kM4), (int)((w >> 3) & kM1), 0, 0, 0, 0};
#pragma unroll
				for (int rb = 0; rb < NRB; rb++) {
					const v4i a = sA[buf][(32 * rb + r) * 8 + ((2 * t + hh) ^ sw)];
					const v8i A = {a.x, a.y, a.z, a.w, 0, 0, 0, 0};
					acc[rb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[rb], 4, 4, 0, 0, 0, 0);
				}
			}
		}
		if (hot_ptr) {          // P2 (see k_pair_gemm_fp4)
			const uint32_t h0 = __builtin_amdgcn_readfirstlane(hot_ptr[2 * (gss0 + ss)]), h1 = __builtin_amdgcn_readfirstlane(hot_ptr[2 * (gss0 + ss) + 2]);
			for (uint32_t e = h0; e < h1; e++) {
				const uint2 en = hot[e];
				const uint32_t bin = __builtin_amdgcn_readfirstlane(en.x), rg = __builtin_amdgcn_readfirstlane(en.y);
				const uint32_t j = (bin >> 5) & 7, wsel = j >> 1;
				const uint32_t w = (uint32_t)(wsel == 0 ? bcur.x : wsel == 1 ? bcur.y : wsel == 2 ? bcur.z : bcur.w);
				const uint32_t bit = (w >> (16 * (j & 1) + (bin & 15))) & 1u;
				if (valid && (lane >> 5) == ((bin >> 4) & 1) && bit) atomicAdd(out_diff + (uint64_t)ci * QN + (rg >> 16), (int32_t)(rg & 0xffffu));
			}
		}
		asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
		__syncthreads();
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (nothing of this wave may still be writing LDS when the workgroup ends)
	if (!valid) return;
	int32_t* o = out_min + ((uint64_t)ks * m + ci) * QN + 4 * (lane >> 5);
	if (piece) {          // (most pairs of unrelated sequences share no k-mer inside a piece)
#pragma unroll
		for (int rb = 0; rb < NRB; rb++)
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const int v = (int)acc[rb][i];
				if (v) atomicAdd(o + 32 * rb + 8 * (i >> 2) + (i & 3), v);
			}
		return;
	}
#pragma unroll
	for (int rb = 0; rb < NRB; rb++)
#pragma unroll
		for (int g = 0; g < 4; g++) *reinterpret_cast<v4i*>(o + 32 * rb + 8 * g) = v4i{(int)acc[rb][4 * g], (int)acc[rb][4 * g + 1], (int)acc[rb][4 * g + 2], (int)acc[rb][4 * g + 3]};
}

}  // namespace

const char* msc_pair_gemm_kernel_name() { return "k_pair_gemm_fp4_dma"; }

uint64_t msc_kb_bytes(const MscLayout& L, uint64_t capacity) { return (capacity + 31) / 32 * msc_kb_block_bytes(L.padded_bins); }

// flags (device, two int32 zeroed by the caller): [0] a zero count was met, [1] the longest list of large bins
hipError_t msc_launch_kb_build(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, uint8_t* kb, uint64_t first_slot, uint64_t n_slots, void* mb,
                               uint32_t* mb_n, uint32_t pitch, int32_t* flags) {
	if (n_slots == 0) return hipSuccess;
	if (L.padded_bins % 256) return hipErrorInvalidValue;
	hipError_t e = hipMemsetAsync(mb_n + first_slot, 0, n_slots * sizeof(uint32_t), st);
	if (e != hipSuccess) return e;
	const uint64_t threads = n_slots * L.padded_bins / 16;
	const dim3 grid((unsigned)((threads + 255) / 256));
	if (dtype == 8) k_kb_build<uint8_t><<<grid, dim3(256), 0, st>>>((const uint8_t*)bins, L.padded_bins, kb, first_slot, n_slots, (uint2*)mb, mb_n, pitch, flags);
	else if (dtype == 16) k_kb_build<uint16_t><<<grid, dim3(256), 0, st>>>((const uint16_t*)bins, L.padded_bins, kb, first_slot, n_slots, (uint2*)mb, mb_n, pitch, flags);
	else if (dtype == 32) k_kb_build<uint32_t><<<grid, dim3(256), 0, st>>>((const uint32_t*)bins, L.padded_bins, kb, first_slot, n_slots, (uint2*)mb, mb_n, pitch, flags);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

// rows of queries one pass serves for a block of n_q <= 128: 32, 64 or 128
uint32_t msc_pair_gemm_rows(uint32_t n_q) { return n_q <= 32 ? 32 : n_q <= 64 ? 64 : 128; }

uint32_t msc_pair_gemm_slices(uint64_t nbins, uint32_t m, uint32_t qn, int num_cus) {
	// As few slices as give every CU about three workgroups (12 waves of its 16): the first rule here asked for six ROUNDS of the chip
	// (24 workgroups per CU) and paid for them twice -- every slice is another int32 per pair written by the product and read and added
	// by the epilogue, and a workgroup's set-up is spread over fewer steps. Measured (G pairs/s, cfg2's sequences, FP4 form):
	//   100 000 candidates (782 workgroups per slice): 1 / 2 / 4 / 8 slices -> 5.10 / 5.08 / 4.97 / 4.76
	//    50 000: 2 / 4 -> 4.88 / 4.77        25 000: 2 / 4 / 8 -> 3.55 / 4.32 / 4.16        12 500: 4 / 8 / 16 / 32 / 64 -> 3.29 / 3.72 / 3.62 / 3.12 / 2.40
	// -- the best is always the count that brings the grid to ~780 workgroups. A slice is an even number of 128-bin steps.
	// MSC_GEMM_MAX_SLICES / MSC_GEMM_SLICES bound the count from above / below for A/B runs.
	(void)qn;
	const uint32_t per_wg = 128;
	uint32_t s = 1;
	static const uint32_t s_max = [] { const char* e = getenv("MSC_GEMM_MAX_SLICES"); return (uint32_t)(e && atoi(e) > 0 ? atoi(e) : 64); }();
	auto can_split = [&] { return s < s_max && nbins / (2 * s) >= 2 * kStep && nbins % (2 * s * 2 * kStep) == 0; };
	while (can_split() && (uint64_t)((m + per_wg - 1) / per_wg) * s < (uint64_t)num_cus * 3) s *= 2;
	static const uint32_t s_min = [] { const char* e = getenv("MSC_GEMM_SLICES"); return (uint32_t)(e ? std::max(1, atoi(e)) : 1); }();
	while (can_split() && s < s_min) s *= 2;
	return s;
}

// bytes of the queries' side of a block: the transposed bit planes ...
uint64_t msc_pair_gemm_qt_bytes(uint64_t nbins, uint32_t qn) { return nbins * 2 * (qn < 128 ? 16 : qn / 8); }
// ... and the nibble image of the queries' tiles
uint64_t msc_pair_gemm_anib_bytes(uint64_t nbins, uint32_t qn) { return nbins / 2 * qn; }

// The queries' side: anib and qT of rows q_slots[0 .. n_q) of the mirror q_kb; when n_hot > 0 also the hot list (hot: n_hot entries;
// hot_ptr, hot_cursor, hot_cnt: nbins / 128 + 1 words each) from the queries' lists of large bins.
hipError_t msc_launch_pair_gemm_queries(hipStream_t st, uint64_t nbins, const uint8_t* q_kb, const void* q_mb, const uint32_t* q_mb_n, uint32_t q_pitch,
                                        const uint32_t* q_slots_dev, uint32_t n_q, uint32_t qn, uint8_t* qT, uint64_t n_hot, void* hot, uint32_t* hot_ptr,
                                        uint32_t* hot_cursor, uint32_t* hot_cnt, uint8_t* anib) {
	if (n_q == 0 || n_q > qn || nbins % 256 || !anib || !qT) return hipErrorInvalidValue;
	const uint32_t nsteps = (uint32_t)(nbins / kStep);
	if (qn == 32) k_kb_gather<32><<<dim3(nsteps), dim3(256), 0, st>>>(q_kb, q_slots_dev, n_q, nbins, qT, anib);
	else if (qn == 64) k_kb_gather<64><<<dim3(nsteps), dim3(256), 0, st>>>(q_kb, q_slots_dev, n_q, nbins, qT, anib);
	else if (qn == 128) k_kb_gather<128><<<dim3(nsteps), dim3(256), 0, st>>>(q_kb, q_slots_dev, n_q, nbins, qT, anib);
	else return hipErrorInvalidValue;
	hipError_t e = hipGetLastError();
	if (e != hipSuccess || n_hot == 0) return e;
	if ((e = hipMemsetAsync(hot_cnt, 0, (nsteps + 1) * sizeof(uint32_t), st)) != hipSuccess) return e;
	k_hot_count<<<dim3(n_q), dim3(256), 0, st>>>((const uint2*)q_mb, q_mb_n, q_pitch, q_slots_dev, n_q, hot_cnt);
	k_hot_scan<<<dim3(1), dim3(1024), 0, st>>>(hot_cnt, nsteps, hot_ptr, hot_cursor);
	k_hot_fill<<<dim3(n_q), dim3(256), 0, st>>>((const uint2*)q_mb, q_mb_n, q_pitch, q_slots_dev, n_q, qn, hot_cursor, (uint2*)hot, qT);
	return hipGetLastError();
}

// P1 [k_slices][m][qn] and, with a hot list, P2 [m][qn] (zeroed here) of the block's queries against m candidates (slot list, or slots
// first .. first + m - 1) of the mirror cand_kb
hipError_t msc_launch_pair_gemm(hipStream_t st, uint64_t nbins, const uint8_t* cand_kb, const uint32_t* cand_slots, uint64_t first, uint32_t m, uint32_t qn,
                                uint32_t k_slices, const uint32_t* hot_ptr, const void* hot, int32_t* out_min, int32_t* out_diff, const uint8_t* anib) {
	if (m == 0) return hipSuccess;
	if (k_slices == 0 || nbins % ((uint64_t)k_slices * 2 * kStep)) return hipErrorInvalidValue;
	if (hot_ptr) {
		const hipError_t e = hipMemsetAsync(out_diff, 0, (size_t)m * qn * sizeof(int32_t), st);
		if (e != hipSuccess) return e;
	}
	if (!anib) return hipErrorInvalidValue;
	{
		// the tiles that do not fill the chip's places a whole number of times go as short pieces (see the kernel); one slice only
		static const bool no_pieces = getenv("MSC_GEMM_NO_PIECES") != nullptr;
		static const int num_cus = [] { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess || p.multiProcessorCount <= 0) return 256; return p.multiProcessorCount; }();
		const uint32_t tiles = (m + 127) / 128, places = (uint32_t)num_cus * 3, n_ss_all = (uint32_t)(nbins / 256);
		uint32_t n_whole = tiles, piece_ss = 0, grid_x = tiles;
		if (k_slices == 1 && !no_pieces && tiles > places && tiles % places && (tiles % places) * 4 <= places) {          // (a leftover of more than a quarter round is better off as a round of whole tiles: 1.87 against 1.95 ms at 150 000 candidates)
			n_whole = tiles / places * places;
			const uint32_t left = tiles - n_whole;
			piece_ss = std::max<uint32_t>(16, (uint32_t)(((uint64_t)left * n_ss_all + places - 1) / places));
			if (piece_ss >= n_ss_all) { n_whole = tiles; piece_ss = 0; }
			else {
				grid_x = n_whole + left * ((n_ss_all + piece_ss - 1) / piece_ss);
				const hipError_t e = hipMemsetAsync(out_min + (size_t)n_whole * 128 * qn, 0, (size_t)(m - n_whole * 128) * qn * sizeof(int32_t), st);
				if (e != hipSuccess) return e;
			}
		}
		const dim3 grid(grid_x, k_slices);
#define MSC_DMA_GO(NRB) k_pair_gemm_fp4_dma<NRB, 4><<<grid, dim3(256), 0, st>>>(cand_kb, cand_slots, first, m, anib, nbins, k_slices, hot_ptr, (const uint2*)hot, out_min, out_diff, n_whole, piece_ss)
		if (qn == 32) MSC_DMA_GO(1);
		else if (qn == 64) MSC_DMA_GO(2);
		else if (qn == 128) MSC_DMA_GO(4);
		else return hipErrorInvalidValue;
#undef MSC_DMA_GO
		return hipGetLastError();
	}
}
