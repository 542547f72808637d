// msc_emd_ranks.hip -- the earth mover's distance of the Q x M pass from SORTED K-MER RANKS instead of per-bin prefixes (gfx950).
//
// The statistic (predict/Feature.cpp emd; the integer the epilogue of pair_features.hip takes) is
//     emd = sum over bins i of | P_i - Q_i |,          P_i = sum_{j <= i} (count_j - 1)   (the common pseudocount cancels)
// Both prefixes are monotone integer staircases, so with a_t = the bin of the t-th counted k-mer of P in bin order (a bin with excess
// e appears e times; t = 1 .. nP) and b_t likewise for Q
//     [P_i >= t] = [i >= a_t]     =>     sum_i min(P_i, Q_i) = sum_{t <= min(nP, nQ)} (nbins - max(a_t, b_t)),    sum_i P_i = sum_t (nbins - a_t)
//     emd = sum_i (P_i + Q_i - 2 min(P_i, Q_i)) = sum_{t <= min(nP, nQ)} | a_t - b_t |  +  sum_{t > min} (nbins - x_t)   (x = the longer list)
// -- the textbook form of the 1-D transport distance, exact in integers. It costs max(nP, nQ) operations per pair where the digest
// kernel spends one per BIN: 1 000 against 262 144 for BASELINE cfg2 (1 kb sequences, k = 9), where the prefix half was two thirds of
// k_pair_digest_multi's arithmetic and half of its bytes.
//
//   ranks mirror   per slot `pitch` uint32 (pitch = the set's longest list rounded up to 256) and its length n: a_1 .. a_n, then nbins repeated -- with
//                  that padding | a_t - b_t | IS the tail term when one list has ended and 0 when both have, so the kernel has no cases.
//                  Built from the slots' bins (the tile prefixes of the scalar record say where each bin's copies go), refreshed with the
//                  other mirrors' stale range.
//                  Needs every count >= 1 (the reference's histograms start at 1, KmerHashTable's initial value; a mean of such too):
//                  a zero bin would make the prefix non-monotone -- the build reports it and the caller keeps the digest's prefix form.
//   k_emd_ranks    a wave holds 1 024 ranks of each of four candidates in registers; the workgroup moves 1 024 ranks of 8 queries at a
//                  time into LDS (LDS-DMA, a two-slot ring) and every wave walks its candidates past them (v_sad_u32 per rank), one
//                  transposed fold per candidate and 16 queries.
#include <type_traits>

#include "msc_internal.h"
#include "msc_wave.h"

namespace {

constexpr uint32_t kTileBytes = 4096;      // a raw tile: 64 lanes x 4 loads of 16 bytes (msc_layout.h, LPT = 4)

// One wave per RAW tile of a dense slot (64 lanes x R = 64 / 32 / 16 bins of uint8 / uint16 / uint32, the lane's R bins logically
// consecutive -- the walk k_digest_build takes): the tile prefix of the scalar record + a wave scan place the lane's run, and every bin
// is written as many times as it was counted.
template <typename T>
__global__ void __launch_bounds__(256) k_ranks_build(const uint8_t* __restrict__ bins, uint64_t slot_bytes, const uint8_t* __restrict__ scalars, uint64_t scalar_stride,
                                                     uint32_t* __restrict__ ranks, uint32_t* __restrict__ n_of, uint64_t pitch, uint64_t nbins, uint64_t first_slot,
                                                     uint64_t n_slots, uint32_t S, int32_t* __restrict__ bad) {
	constexpr int R = 64 / sizeof(T);
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t W = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (W >= n_slots * S) return;
	const uint64_t slot = first_slot + W / S;
	const uint32_t s = (uint32_t)(W % S);
	const u32x4* src = reinterpret_cast<const u32x4*>(bins + slot * slot_bytes + (uint64_t)s * kTileBytes) + lane;
	u32x4 v[4];
#pragma unroll
	for (int l = 0; l < 4; l++) v[l] = src[64 * l];
	const T* w = reinterpret_cast<const T*>(v);
	uint32_t t = 0;
#pragma unroll
	for (int r = 0; r < R; r++) t += (uint32_t)w[r];
	const uint64_t* prefix = reinterpret_cast<const uint64_t*>(scalars + slot * scalar_stride + sizeof(MscSlotScalars));
	const uint32_t first_bin = s * (64u * R) + lane * R;
	uint32_t run = (uint32_t)prefix[s] + wave_incl_scan(t) - t - first_bin;      // counted k-mers in the bins before this lane's run
	uint32_t* out = ranks + slot * pitch;
	bool zero = false;
#pragma unroll
	for (int r = 0; r < R; r++) {
		const uint32_t c = (uint32_t)w[r];
		if (c == 0) { zero = true; continue; }
		for (uint32_t e = c - 1; e > 0; e--, run++) if (run < pitch) out[run] = first_bin + r;
	}
	if (zero) atomicOr(bad, 1);
	if (s == S - 1) {          // behind the last counted k-mer: nbins up to the pitch
		const uint32_t n = __builtin_amdgcn_readlane(run, 63);
		if (lane == 0) { n_of[slot] = n; if (n > pitch) atomicOr(bad, 2); }          // (the host sized the pitch from the set's largest sum)
		for (uint64_t i = n + lane; i < pitch; i += 64) out[i] = (uint32_t)nbins;
	}
}

typedef int v4i_ __attribute__((ext_vector_type(4)));

// 16 per-lane sums (one per query of a group) -> one register: lane (row r, bank b) holds the wave total of query {0, 2, 1, 3}[b] + 4 r
// in all four lanes of the bank. permlane32/16 swaps take 16 -> 8 -> 4 registers (row r of register i then holds 16 partial sums of
// query i + 4 r), DPP adds and bank-masked merges the rest: 35 operations for 16 totals.
__device__ __forceinline__ uint32_t fold16q(const uint32_t (&s)[16]) {
	auto fold32 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false); return r.x + r.y; };
	auto fold16 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false); return r.x + r.y; };
	uint32_t r2[4];
#pragma unroll
	for (int i = 0; i < 4; i++) r2[i] = fold16(fold32(s[i], s[i + 8]), fold32(s[i + 4], s[i + 12]));
	uint32_t X[4];
#pragma unroll
	for (int i = 0; i < 4; i++) X[i] = r2[i] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r2[i], 0x128, 0xf, 0xf, false);      // row_ror:8
	uint32_t P = (uint32_t)__builtin_amdgcn_update_dpp((int)X[0], (int)X[1], 0xe4, 0xf, 0xc, false);      // lanes 8-15 <- register 1
	uint32_t Q = (uint32_t)__builtin_amdgcn_update_dpp((int)X[2], (int)X[3], 0xe4, 0xf, 0xc, false);      // lanes 8-15 <- register 3
	P += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x141, 0xf, 0xf, false);                        // row_half_mirror
	Q += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Q, 0x141, 0xf, 0xf, false);
	uint32_t R = (uint32_t)__builtin_amdgcn_update_dpp((int)P, (int)Q, 0xe4, 0xf, 0xa, false);            // banks 1 and 3 <- registers 2 and 3
	R += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)R, 0xb1, 0xf, 0xf, false);
	R += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)R, 0x4e, 0xf, 0xf, false);
	return R;
}

// 8 per-lane sums -> one register: lane l ends up with the wave total of query 2 (l / 16) + (l / 8) % 2 in all eight lanes of its half
// row. permlane32 / 16 swaps take 8 -> 4 -> 2 registers (row r of register i then holds 16 partial sums of query i + 2 r), DPP adds and
// one bank-masked merge the rest: 18 operations for 8 totals.
__device__ __forceinline__ uint32_t fold8q(const uint32_t (&s)[8]) {
	auto fold32 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false); return r.x + r.y; };
	auto fold16 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false); return r.x + r.y; };
	uint32_t w[2];
#pragma unroll
	for (int i = 0; i < 2; i++) w[i] = fold16(fold32(s[i], s[i + 4]), fold32(s[i + 2], s[i + 6]));
	uint32_t X[2];
#pragma unroll
	for (int i = 0; i < 2; i++) X[i] = w[i] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[i], 0x128, 0xf, 0xf, false);      // row_ror:8
	uint32_t P = (uint32_t)__builtin_amdgcn_update_dpp((int)X[0], (int)X[1], 0xe4, 0xf, 0xc, false);      // lanes 8-15 of every row <- register 1
	P += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x141, 0xf, 0xf, false);                        // row_half_mirror
	P += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0xb1, 0xf, 0xf, false);
	P += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x4e, 0xf, 0xf, false);
	return P;
}

// one wave-instruction of LDS-DMA: 64 lanes x 16 bytes from global memory (wave-uniform base in SGPRs + per-lane byte offset) straight
// into LDS at lds_dst + 16 lane (no destination VGPRs); m0 is the compiler's: saved and restored
__device__ __forceinline__ void dma_piece(uint64_t sbase, uint32_t lane_off, uint32_t lds_dst) {
	uint32_t keep;
	asm volatile(
	    "s_mov_b32 %0, m0\n\t"
	    "s_mov_b32 m0, %3\n\t"
	    "s_nop 0\n\t"
	    "global_load_lds_dwordx4 %1, %2\n\t"
	    "s_mov_b32 m0, %0"
	    : "=&s"(keep)
	    : "v"(lane_off), "s"(sbase), "s"(lds_dst)
	    : "memory");
}

// One workgroup = 8 waves = 16 candidates (2 per wave, their ranks of the current round held in registers: 32 VGPRs) x ALL queries of the
// block, taken 8 at a time through a two-slot LDS ring (2 x 32 KiB) filled by LDS-DMA one half-group ahead: while a wave walks the 8
// lists of one slot past its two candidates (per query 4 ds_read_b128 + 32 v_sad_u32), the next 8 lists land in the other. One barrier
// per half-group; one transposed fold per candidate and 16 queries. ~100 registers: four waves per SIMD (two workgroups per CU by LDS).
// (r03 held ONE candidate per wave and gave every group of 16 queries its own workgroups: every candidate list was fetched once per
// 16 queries -- 3.3 GB per 128-query block of cfg2 -- and every LDS word served one pair. Forms tried on the way here, per pair of cfg2:
// r03 92 ps; 4 candidates per wave + 16 lists staged through registers between two barriers 85 ps; the same with the DMA ring 83 ps --
// at two waves per SIMD the waits were never covered; this form: see DESIGN.md.)
// Lists longer than a round (1 kb sequences: one round) add their rounds up in `out`.
constexpr uint32_t kRound = 1024, kQGroup = 16, kQHalf = 8, kCandPerWave = 2, kWaves = 8;
__global__ void __launch_bounds__(512, 2) k_emd_ranks(const uint32_t* __restrict__ c_rk, uint64_t c_pitch, const uint32_t* __restrict__ c_n, const uint32_t* __restrict__ cand_slots,
                                                      uint64_t first, uint32_t m, const uint32_t* __restrict__ q_rk, uint64_t q_pitch, const uint32_t* __restrict__ q_n,
                                                      const uint32_t* __restrict__ q_slots, uint32_t n_q, uint32_t nbins, uint64_t* __restrict__ out, uint32_t out_stride) {
	__shared__ v4i_ sQ[2][kQHalf][kRound / 4];          // 2 x 32 KiB
	const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t c0 = (blockIdx.x * kWaves + wave) * kCandPerWave;
	const uint32_t n_groups = (n_q + kQGroup - 1) / kQGroup;
	uint64_t slot[kCandPerWave];
	uint32_t nc[kCandPerWave];
#pragma unroll
	for (uint32_t c = 0; c < kCandPerWave; c++) {
		const uint32_t ci = c0 + c < m ? c0 + c : m - 1;
		slot[c] = cand_slots ? cand_slots[ci] : first + ci;
		nc[c] = c_n[slot[c]];
	}
	const uint64_t rounds_end = c_pitch > q_pitch ? c_pitch : q_pitch;      // workgroup-uniform (the barriers below)
	// which query's total a lane ends up with (fold16q), and whether this lane stores it
	const uint32_t my_q = ((0x3120u >> (4 * ((lane >> 2) & 3))) & 3) + 4 * (lane >> 4);
	const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&sQ[0][0][0]);
	const v4i_ pad = {(int)nbins, (int)nbins, (int)nbins, (int)nbins};
	for (uint64_t base = 0; base < rounds_end; base += kRound) {
		v4i_ a[kCandPerWave][4];          // lane l: ranks 256 j + 4 l .. + 3 of the round
#pragma unroll
		for (uint32_t c = 0; c < kCandPerWave; c++)
#pragma unroll
			for (int j = 0; j < 4; j++) {
				a[c][j] = pad;
				if (base + 256 * j < c_pitch) a[c][j] = *reinterpret_cast<const v4i_*>(c_rk + slot[c] * c_pitch + base + 256 * j + 4 * lane);
			}
		// the compiler must see these loads complete HERE: its scoreboard knows nothing of the DMA pieces issued from inline asm, and a
		// wait of its own inside the loop would drain them
#pragma unroll
		for (uint32_t c = 0; c < kCandPerWave; c++)
#pragma unroll
			for (int j = 0; j < 4; j++) asm volatile("" : "+v"(a[c][j]));
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		// half-group h = queries 8 h .. 8 h + 7 -> ring slot h % 2: each wave moves 4 of its 32 pieces of 1 KiB = one list's round
		auto stage = [&](uint32_t h) {
			const uint32_t q = wave, qi = h * kQHalf + q;
#pragma unroll
			for (uint32_t part = 0; part < 4; part++) {
				if (qi < n_q && base + 256 * part < q_pitch)
					dma_piece((uint64_t)(q_rk + (uint64_t)q_slots[qi] * q_pitch + base + 256 * part), lane * 16u, lds0 + (((h & 1) * kQHalf + q) * (kRound / 4) + 64 * part) * 16);
				else
					sQ[h & 1][q][64 * part + lane] = pad;          // (queries past n_q: never stored; parts past the pitch: | x - nbins | is the tail term)
			}
		};
		// the 8 lists of ring slot `buf` past this wave's candidates: sums of queries 8 `half` .. + 7 of the group
		uint32_t sum[kCandPerWave][kQGroup];
		auto walk = [&](uint32_t buf, auto half) {
			constexpr uint32_t H = decltype(half)::value;
#pragma unroll
			for (uint32_t q = 0; q < kQHalf; q++) {
				v4i_ b[4];
#pragma unroll
				for (int j = 0; j < 4; j++) b[j] = sQ[buf][q][64 * j + lane];
#pragma unroll
				for (uint32_t c = 0; c < kCandPerWave; c++) {
					uint32_t t = 0;          // nbins <= 2^20: 16 terms fit 32 bits, and so do the 64 lanes' in the fold
#pragma unroll
					for (int j = 0; j < 4; j++) {
						asm("v_sad_u32 %0, %1, %2, %0" : "+v"(t) : "v"(a[c][j].x), "v"(b[j].x));
						asm("v_sad_u32 %0, %1, %2, %0" : "+v"(t) : "v"(a[c][j].y), "v"(b[j].y));
						asm("v_sad_u32 %0, %1, %2, %0" : "+v"(t) : "v"(a[c][j].z), "v"(b[j].z));
						asm("v_sad_u32 %0, %1, %2, %0" : "+v"(t) : "v"(a[c][j].w), "v"(b[j].w));
					}
					sum[c][H * kQHalf + q] = t;
				}
			}
		};
		auto landed = [&] {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces have landed (and its stores have left)
			__syncthreads();                                           // everybody's have, and everybody is done with the slot refilled next
		};
		stage(0);
		for (uint32_t g = 0; g < n_groups; g++) {          // two half-groups per turn: ring slots 0 and 1
			uint32_t nq_max = 0;
			for (uint32_t q = 0; q < kQGroup; q++) {
				const uint32_t qi = g * kQGroup + q;
				const uint32_t v = qi < n_q ? q_n[q_slots[qi]] : 0;
				nq_max = v > nq_max ? v : nq_max;
			}
			// a later round that lies past every list of this wave's candidates and of the group is all | nbins - nbins |: the wave
			// keeps the ring's pace (barriers, its share of the pieces) and skips the walk -- one long list in a set makes every pass
			// two rounds, and all but a few pairs end in the first
			uint32_t reach = nq_max;
#pragma unroll
			for (uint32_t c = 0; c < kCandPerWave; c++) reach = nc[c] > reach ? nc[c] : reach;
			const bool idle = base != 0 && base >= reach;
			landed();
			stage(2 * g + 1);
			if (!idle) walk(0, std::integral_constant<uint32_t, 0>());
			landed();
			if (g + 1 < n_groups) stage(2 * g + 2);
			if (idle) continue;
			walk(1, std::integral_constant<uint32_t, 1>());
			const uint32_t q0 = g * kQGroup;
			const bool owner = (lane & 3) == 0 && q0 + my_q < n_q;
#pragma unroll
			for (uint32_t c = 0; c < kCandPerWave; c++) {
				const uint32_t tot = fold16q(sum[c]);
				// past every list of this candidate and group all terms are | nbins - nbins |: nothing to add
				if (owner && c0 + c < m && (base == 0 || base < (nc[c] > nq_max ? nc[c] : nq_max))) {
					uint64_t* o = out + (uint64_t)(c0 + c) * out_stride + q0 + my_q;
					*o = base ? *o + tot : (uint64_t)tot;
				}
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();          // the ring is free for the next round's first half-group
	}
}


// ------------------------------------------------------------------------------------------------ the 16-bit form (r05)
// | a_t - b_t | does not change when the same number is taken from both ranks. With base_t = floor(t * nbins / pitch) -- where the t-th
// k-mer of a sequence whose k-mers were spread evenly over the bins would sit -- the reduced rank a_t - base_t + 32768 of ordinary
// sequences fits 16 bits (a 1 kb sequence at k = 9 strays a few thousand bins from the even spread; the padding nbins - base_t fits
// once a list fills 7/8 of the pitch; up to k = 7 everything fits), and v_sad_u16 then takes TWO ranks per instruction and lane: half
// the vector instructions, half the LDS bytes and half the registers per candidate of the 32-bit walk above, so a wave holds four
// candidates and one LDS read of a query's ranks serves four of them. Exact: the sums are the same integers. A set whose every slot
// fits carries the mirror (k_ranks16_build reports a slot that does not, and the set then keeps the 32-bit walk); both sets of a pass
// must share one pitch (one base_t).
__global__ void __launch_bounds__(256) k_ranks16_build(const uint32_t* __restrict__ ranks, uint64_t pitch, uint16_t* __restrict__ out, uint64_t nbins, uint64_t first_slot,
                                                       uint64_t n_slots, int32_t* __restrict__ bad) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_slots * pitch) return;
	const uint64_t slot = first_slot + i / pitch, t = i % pitch;
	const int64_t v = (int64_t)ranks[slot * pitch + t] - (int64_t)(t * nbins / pitch) + 32768;
	if (v < 0 || v > 65535) atomicOr(bad, 1);
	out[slot * pitch + t] = (uint16_t)v;
}

constexpr uint32_t kCand16 = 4;          // candidates per wave
// NWV waves per workgroup (NWV x 4 candidates), QS queries per ring slot (2 slots of QS x 2 KiB). <8, 8>: 32 KiB of LDS, the form in use.
// (<4, 4> -- 16 KiB and 4 waves of 96 registers, what is left on a CU that holds three workgroups of the product kernel -- does run
// beside the product (0.37 -> 0.68 ms, the product 1.1 -> 1.6 - 2.2 ms), but the step comes out the same, 14.36 against 14.16 ms per
// 1 024 queries: the two kernels share issue slots and LDS ports, not only places. profiles/r05_notes.md.)
template <uint32_t NWV, uint32_t QS>
__global__ void __launch_bounds__(64 * NWV, NWV == 8 ? 2 : 4) k_emd_ranks16(const uint16_t* __restrict__ c_rk, uint64_t pitch, const uint32_t* __restrict__ c_n, const uint32_t* __restrict__ cand_slots,
                                                        uint64_t first, uint32_t m, const uint16_t* __restrict__ q_rk, const uint32_t* __restrict__ q_n,
                                                        const uint32_t* __restrict__ q_slots, uint32_t n_q, uint32_t nbins, uint64_t* __restrict__ out, uint32_t out_stride) {
	static_assert(QS % NWV == 0 && 8 % QS == 0, "ring slots");
	constexpr uint32_t SPG = 8 / QS;          // ring slots per group of 8 queries (one fold)
	__shared__ v4i_ sQ[2][QS][kRound / 8];          // a list's round = 1 024 reduced ranks = 2 KiB
	const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t c0 = (blockIdx.x * NWV + wave) * kCand16;
	uint64_t slot[kCand16];
	uint32_t nc[kCand16];
#pragma unroll
	for (uint32_t c = 0; c < kCand16; c++) {
		const uint32_t ci = c0 + c < m ? c0 + c : m - 1;
		slot[c] = cand_slots ? cand_slots[ci] : first + ci;
		nc[c] = c_n[slot[c]];
	}
	const uint32_t my_q = 2 * (lane >> 4) + ((lane >> 3) & 1);          // the query of a group whose total fold8q leaves in this lane
	const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&sQ[0][0][0]);
	const uint32_t n_slots = (n_q + QS - 1) / QS;
	for (uint64_t base = 0; base < pitch; base += kRound) {
		v4i_ a[kCand16][2];          // lane l: reduced ranks 512 j + 8 l .. + 7 of the round, two per register
#pragma unroll
		for (uint32_t c = 0; c < kCand16; c++)
#pragma unroll
			for (int j = 0; j < 2; j++) a[c][j] = *reinterpret_cast<const v4i_*>(c_rk + slot[c] * pitch + base + 512 * j + 8 * lane);          // (pitch is a multiple of 1 024 here)
#pragma unroll
		for (uint32_t c = 0; c < kCand16; c++)
#pragma unroll
			for (int j = 0; j < 2; j++) asm volatile("" : "+v"(a[c][j]));
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		auto stage = [&](uint32_t hs) {          // ring slot hs % 2 <- queries QS hs .. + QS - 1: a wave moves QS / NWV lists' rounds, two pieces of 1 KiB each
#pragma unroll
			for (uint32_t q = wave; q < QS; q += NWV) {
				const uint32_t qi = hs * QS + q;
				const uint32_t qs = q_slots[qi < n_q ? qi : n_q - 1];          // (rows past n_q: some list, never stored)
#pragma unroll
				for (uint32_t part = 0; part < 2; part++)
					dma_piece((uint64_t)(q_rk + (uint64_t)qs * pitch + base + 512 * part), lane * 16u, lds0 + (((hs & 1) * QS + q) * (kRound / 8) + 64 * part) * 16);
			}
		};
		stage(0);
		uint32_t sum[kCand16][8];
		uint32_t nq_max = 0;
		bool idle = false;
		for (uint32_t hs = 0; hs < n_slots; hs++) {          // a group of 8 queries = SPG ring slots; its 8 totals per candidate are folded and stored at once
			if (hs % SPG == 0) {
				nq_max = 0;
				for (uint32_t q = 0; q < 8; q++) {
					const uint32_t qi = (hs / SPG) * 8 + q;
					const uint32_t v = qi < n_q ? q_n[q_slots[qi]] : 0;
					nq_max = v > nq_max ? v : nq_max;
				}
				uint32_t reach = nq_max;
#pragma unroll
				for (uint32_t c = 0; c < kCand16; c++) reach = nc[c] > reach ? nc[c] : reach;
				idle = base != 0 && base >= reach;          // (a later round past every list of the wave and the group: all terms | pad - pad |)
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces have landed (and its stores have left)
			__syncthreads();                                       // everybody's have, and everybody is done with the slot refilled next
			if (hs + 1 < n_slots) stage(hs + 1);
			if (idle) continue;
			const uint32_t buf = hs & 1, sub = (hs % SPG) * QS;
#pragma unroll
			for (uint32_t q = 0; q < QS; q++) {
				v4i_ b[2];
#pragma unroll
				for (int j = 0; j < 2; j++) b[j] = sQ[buf][q][64 * j + lane];
#pragma unroll
				for (uint32_t c = 0; c < kCand16; c++) {
					uint32_t t = 0;          // 16 terms of < 2^17 per lane; 64 lanes in the fold: < 2^27
#pragma unroll
					for (int j = 0; j < 2; j++) {
						asm("v_sad_u16 %0, %1, %2, %0" : "+v"(t) : "v"(a[c][j].x), "v"(b[j].x));
						asm("v_sad_u16 %0, %1, %2, %0" : "+v"(t) : "v"(a[c][j].y), "v"(b[j].y));
						asm("v_sad_u16 %0, %1, %2, %0" : "+v"(t) : "v"(a[c][j].z), "v"(b[j].z));
						asm("v_sad_u16 %0, %1, %2, %0" : "+v"(t) : "v"(a[c][j].w), "v"(b[j].w));
					}
					if constexpr (SPG == 1) sum[c][q] = t;
					else {          // (sub is not a compile-time constant: a select per register instead of an indexed array in scratch)
#pragma unroll
						for (uint32_t z = 0; z < 8; z++) sum[c][z] = z == sub + q ? t : sum[c][z];
					}
				}
			}
			if ((hs + 1) % SPG && hs + 1 < n_slots) continue;
			const uint32_t q0 = (hs / SPG) * 8;
			const bool owner = (lane & 7) == 0 && q0 + my_q < n_q;
#pragma unroll
			for (uint32_t c = 0; c < kCand16; c++) {
				const uint32_t tot = fold8q(sum[c]);
				if (owner && c0 + c < m && (base == 0 || base < (nc[c] > nq_max ? nc[c] : nq_max))) {
					uint64_t* o = out + (uint64_t)(c0 + c) * out_stride + q0 + my_q;
					*o = base ? *o + tot : (uint64_t)tot;
				}
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();          // the ring is free for the next round's first slot
	}
}

}  // namespace

// ranks per slot of a set whose longest list holds `max_excess` k-mers
uint64_t msc_ranks_pitch(uint64_t max_excess) { return std::max<uint64_t>(256, (max_excess + 255) / 256 * 256); }      // (k_emd_ranks loads 256 ranks per wave instruction)

// *bad (device int32, zeroed by the caller) is set when a slot holds a zero count: the ranks of that set are then not usable
hipError_t msc_launch_ranks_build(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, const uint8_t* scalars, uint32_t* ranks, uint32_t* n_of, uint64_t pitch,
                                  uint64_t first_slot, uint64_t n_slots, int32_t* bad) {
	if (n_slots == 0) return hipSuccess;
	if (L.LPT != 4 || L.nbins != L.padded_bins) return hipErrorInvalidValue;
	const uint64_t waves = n_slots * L.S;
	const dim3 grid((unsigned)((waves + 3) / 4));
	const uint64_t ss = msc_scalar_stride(L.S);
	if (dtype == 8) k_ranks_build<uint8_t><<<grid, dim3(256), 0, st>>>(bins, L.slot_bytes, scalars, ss, ranks, n_of, pitch, L.nbins, first_slot, n_slots, L.S, bad);
	else if (dtype == 16) k_ranks_build<uint16_t><<<grid, dim3(256), 0, st>>>(bins, L.slot_bytes, scalars, ss, ranks, n_of, pitch, L.nbins, first_slot, n_slots, L.S, bad);
	else if (dtype == 32) k_ranks_build<uint32_t><<<grid, dim3(256), 0, st>>>(bins, L.slot_bytes, scalars, ss, ranks, n_of, pitch, L.nbins, first_slot, n_slots, L.S, bad);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

// out[candidate][out_stride]: emd of (candidate, query q) at [q], q < n_q <= out_stride; nbins <= 2^20
hipError_t msc_launch_emd_ranks(hipStream_t st, uint64_t nbins, const uint32_t* c_ranks, uint64_t c_pitch, const uint32_t* c_n, const uint32_t* cand_slots, uint64_t first,
                                uint32_t m, const uint32_t* q_ranks, uint64_t q_pitch, const uint32_t* q_n, const uint32_t* q_slots_dev, uint32_t n_q, uint64_t* out,
                                uint32_t out_stride) {
	if (m == 0 || n_q == 0) return hipSuccess;
	if (n_q > out_stride || nbins > (1u << 20) || c_pitch % 256 || q_pitch % 256) return hipErrorInvalidValue;
	k_emd_ranks<<<dim3((m + kWaves * kCandPerWave - 1) / (kWaves * kCandPerWave)), dim3(64 * kWaves), 0, st>>>(c_ranks, c_pitch, c_n, cand_slots, first, m, q_ranks, q_pitch, q_n, q_slots_dev, n_q, (uint32_t)nbins, out, out_stride);
	return hipGetLastError();
}

// the 16-bit mirror of slots [first_slot, first_slot + n_slots) from their ranks; *bad |= 1 when a reduced rank does not fit
hipError_t msc_launch_ranks16_build(hipStream_t st, uint64_t nbins, const uint32_t* ranks, uint16_t* ranks16, uint64_t pitch, uint64_t first_slot, uint64_t n_slots, int32_t* bad) {
	if (n_slots == 0) return hipSuccess;
	const uint64_t n = n_slots * pitch;
	k_ranks16_build<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(ranks, pitch, ranks16, nbins, first_slot, n_slots, bad);
	return hipGetLastError();
}

// as msc_launch_emd_ranks over the 16-bit mirrors of two sets that share one pitch (a multiple of 1 024)
hipError_t msc_launch_emd_ranks16(hipStream_t st, uint64_t nbins, const uint16_t* c_ranks, uint64_t pitch, const uint32_t* c_n, const uint32_t* cand_slots, uint64_t first,
                                  uint32_t m, const uint16_t* q_ranks, const uint32_t* q_n, const uint32_t* q_slots_dev, uint32_t n_q, uint64_t* out, uint32_t out_stride) {
	if (m == 0 || n_q == 0) return hipSuccess;
	if (n_q > out_stride || nbins > (1u << 20) || pitch % kRound) return hipErrorInvalidValue;
	k_emd_ranks16<8, 8><<<dim3((m + 8 * kCand16 - 1) / (8 * kCand16)), dim3(512), 0, st>>>(c_ranks, pitch, c_n, cand_slots, first, m, q_ranks, q_n, q_slots_dev, n_q, (uint32_t)nbins, out, out_stride);
	return hipGetLastError();
}
