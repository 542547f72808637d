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
//   ranks mirror   per slot `pitch` uint32 (pitch = the set's longest list rounded up to 64) and its length n: a_1 .. a_n, then nbins repeated -- with
//                  that padding | a_t - b_t | IS the tail term when one list has ended and 0 when both have, so the kernel has no cases.
//                  Built from the digest mirror (whose prefix words say where each bin's copies go), refreshed with its stale range.
//                  Needs every count >= 1 (the reference's histograms start at 1, KmerHashTable's initial value; a mean of such too):
//                  a zero bin would make the prefix non-monotone -- the build reports it and the caller keeps the digest's prefix form.
//   k_emd_ranks    one wave per candidate: 1 024 ranks of it in registers per round, every query's list streamed past them (L2-resident:
//                  64 lists of 4 KiB), v_sad_u32 per rank, one wave sum per (pair, round), the pair's total kept in lane q.
#include "msc_internal.h"
#include "msc_wave.h"

namespace {

constexpr uint32_t kTileBytes = 4096;      // a digest tile: 1024 bins = 64 lanes x 16 words (pair_digest.hip)

// one wave per digest tile: lane l holds 16 consecutive bins (words 0..7: two counts each) and their inclusive excess prefixes (words 8..15)
__global__ void __launch_bounds__(256) k_ranks_build(const uint8_t* __restrict__ digest, uint64_t dg_slot_bytes, uint32_t* __restrict__ ranks, uint32_t* __restrict__ n_of,
                                                     uint64_t pitch, uint64_t nbins, uint64_t first_slot, uint64_t n_slots, uint32_t S, int32_t* __restrict__ bad) {
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t W = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (W >= n_slots * S) return;
	const uint64_t slot = first_slot + W / S;
	const uint32_t s = (uint32_t)(W % S);
	const u32x4* src = reinterpret_cast<const u32x4*>(digest + slot * dg_slot_bytes + (uint64_t)s * kTileBytes) + lane;
	u32x4 v[4];
#pragma unroll
	for (int l = 0; l < 4; l++) v[l] = src[64 * l];
	const uint32_t* w = reinterpret_cast<const uint32_t*>(v);
	uint32_t* out = ranks + slot * pitch;
	const uint32_t bin0 = s * 1024u + lane * 16u;
	bool zero = false;
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const uint32_t c = (w[r >> 1] >> (16 * (r & 1))) & 0xffffu;
		const uint32_t E = (w[8 + (r >> 1)] >> (16 * (r & 1))) & 0xffffu;      // excess prefix up to and including this bin
		if (c == 0) { zero = true; continue; }
		for (uint32_t e = c - 1; e > 0; e--) if (E - e < pitch) out[E - e] = bin0 + r;
	}
	if (zero) atomicOr(bad, 1);
	if (s == S - 1) {          // behind the last counted k-mer (the prefix of the last bin says how many there are): nbins up to the pitch
		const uint32_t n = __builtin_amdgcn_readlane(w[15] >> 16, 63);
		if (lane == 0) { n_of[slot] = n; if (n > pitch) atomicOr(bad, 2); }          // (the host sized the pitch from the set's largest sum)
		for (uint64_t t = n + lane; t < pitch; t += 64) out[t] = (uint32_t)nbins;
	}
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

__global__ void __launch_bounds__(256) k_emd_ranks(const uint32_t* __restrict__ c_rk, uint64_t c_pitch, const uint32_t* __restrict__ c_n, const uint32_t* __restrict__ cand_slots,
                                                   uint64_t first, uint32_t m, const uint32_t* __restrict__ q_rk, uint64_t q_pitch, const uint32_t* __restrict__ q_n,
                                                   const uint32_t* __restrict__ q_slots, uint32_t n_q, uint32_t nbins, uint64_t* __restrict__ out) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t ci = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (ci >= m) return;
	const uint64_t slot = cand_slots ? cand_slots[ci] : first + ci;
	const uint32_t nc = c_n[slot];
	const uint32_t* A = c_rk + slot * c_pitch;
	const uint32_t my_q = lane < n_q ? q_slots[lane] : 0;
	const uint32_t nq_l = lane < n_q ? q_n[my_q] : 0;
	uint32_t n_all = nq_l;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = __shfl_xor(n_all, off, 64); n_all = o > n_all ? o : n_all; }
	n_all = n_all > nc ? n_all : nc;
	uint64_t tot = 0;          // lane q: the total of (candidate, query q)
	for (uint32_t base = 0; base < n_all; base += 1024) {
		uint32_t a[16];
#pragma unroll
		for (int j = 0; j < 16; j++) a[j] = base + 64 * j < c_pitch ? A[base + 64 * j + lane] : nbins;
		for (uint32_t q = 0; q < n_q; q++) {
			const uint32_t nq = __builtin_amdgcn_readlane(nq_l, q);
			const uint32_t lim = nc > nq ? nc : nq;          // past both lists every term is | nbins - nbins |
			if (base >= lim) continue;
			const uint32_t* B = q_rk + (uint64_t)__builtin_amdgcn_readlane(my_q, q) * q_pitch;
			uint32_t sum = 0;
#pragma unroll
			for (int j = 0; j < 16; j++) {
				const uint32_t t0 = base + 64 * j;
				if (t0 < lim) {
					const uint32_t b = t0 < q_pitch ? B[t0 + lane] : nbins;
					asm("v_sad_u32 %0, %1, %2, %0" : "+v"(sum) : "v"(a[j]), "v"(b));      // nbins <= 2^20: 16 terms fit, and so do the 64 lanes' below
				}
			}
			sum = wave_sum_u32(sum);
			if (lane == q) tot += sum;
		}
	}
	if (lane < n_q) out[(uint64_t)ci * 64 + lane] = tot;
}

}  // namespace

// ranks per slot of a set whose longest list holds `max_excess` k-mers
uint64_t msc_ranks_pitch(uint64_t max_excess) { return std::max<uint64_t>(64, (max_excess + 63) / 64 * 64); }

// *bad (device int32, zeroed by the caller) is set when a slot holds a zero count: the ranks of that set are then not usable
hipError_t msc_launch_ranks_build(hipStream_t st, const MscLayout& L, const uint8_t* digest, uint32_t* ranks, uint32_t* n_of, uint64_t pitch, uint64_t first_slot,
                                  uint64_t n_slots, int32_t* bad) {
	if (n_slots == 0) return hipSuccess;
	const uint32_t S = (uint32_t)(L.nbins / 1024);
	const uint64_t waves = n_slots * S;
	k_ranks_build<<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st>>>(digest, msc_digest_slot_bytes(L), ranks, n_of, pitch, L.nbins, first_slot, n_slots, S, bad);
	return hipGetLastError();
}

// out[candidate][64]: emd of (candidate, query q) at [q], q < n_q <= 64; nbins <= 2^20
hipError_t msc_launch_emd_ranks(hipStream_t st, uint64_t nbins, const uint32_t* c_ranks, uint64_t c_pitch, const uint32_t* c_n, const uint32_t* cand_slots, uint64_t first,
                                uint32_t m, const uint32_t* q_ranks, uint64_t q_pitch, const uint32_t* q_n, const uint32_t* q_slots_dev, uint32_t n_q, uint64_t* out) {
	if (m == 0 || n_q == 0) return hipSuccess;
	if (n_q > 64 || nbins > (1u << 20) || c_pitch % 64 || q_pitch % 64) return hipErrorInvalidValue;
	k_emd_ranks<<<dim3((m + 3) / 4), dim3(256), 0, st>>>(c_ranks, c_pitch, c_n, cand_slots, first, m, q_ranks, q_pitch, q_n, q_slots_dev, n_q, (uint32_t)nbins, out);
	return hipGetLastError();
}
