// hist_build.hip -- gfx950 kernels that build k-mer histograms in HBM.
//
// Replaces, for a whole batch of sequences at once,
//   KmerHashTable::initialize fill             nonltr/KmerHashTable.cpp:69-72
//   KmerHashTable::hash + wholesaleIncrementNoOverflow   nonltr/KmerHashTable.cpp:134-160,236-256
//   Loader::fill_table copy, DivergencePoint ctor copy + mag, stddev loop
//                                             clutil/Loader.cpp:73-77,158-171; clutil/DivergencePoint.cpp:99-110
// All HBM-bound byte/integer work (no MFMA). Three builders, chosen per batch by msc_hist_build_packed:
//   k_build_lds  : 4^k <= 16384 -- the whole histogram in LDS, slot written once
//   k_build_sort : larger k, <= 32768 k-mers per sequence -- the sequence's k-mer indices sorted in LDS, every scalar derived
//                  from the runs, slot written once as a stream of ones patched per tile
//   k_fill + k_count + k_finalize + k_prefix : everything else (long sequences, ungrouped segment lists)
//     k_fill      : coalesced 16-byte stores of the pseudocount (1) over the batch's slots
//     k_count     : one thread per k-mer; 2k bits pulled from the packed 2-bit stream, bit-reversed into the
//                   reference's "first base most significant" index, mapped through the tile permutation
//                   (msc_layout.h) and added with one global atomic (saturating CAS only when a sequence is
//                   long enough to saturate T)
//     k_finalize  : one workgroup per slot re-reads it once: sum, sum of squares, max, per-tile sums;
//     k_prefix    : exclusive scan of the tile sums -> the tile carries the pair kernel's prefix statistic needs.
#include "msc_internal.h"

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------ fill
template <typename T>
__global__ void __launch_bounds__(kBlock) k_fill(T* __restrict__ bins, uint64_t first_slot, uint64_t n_slots,
                                                 uint64_t slot_elems, uint64_t nbins, uint32_t E, uint32_t R) {
	// one 16-byte chunk per thread-iteration
	const uint64_t chunks_per_slot = slot_elems / E;
	const uint64_t total = n_slots * chunks_per_slot;
	const bool padded = nbins < slot_elems;
	T ones[16 / sizeof(T)];
#pragma unroll
	for (uint32_t j = 0; j < 16 / sizeof(T); j++) ones[j] = (T)1;
	const uint4 one_v = *reinterpret_cast<const uint4*>(ones);
	uint4* base = reinterpret_cast<uint4*>(bins + first_slot * slot_elems);
	for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (uint64_t)gridDim.x * blockDim.x) {
		if (!padded) {
			base[c] = one_v;
		} else {
			// only histograms smaller than one tile are padded (S == 1): zero the bins past 4^k
			uint64_t in_slot = c % chunks_per_slot;
			uint32_t t = (uint32_t)(in_slot / 64), lane = (uint32_t)(in_slot % 64);
			T v[16 / sizeof(T)];
#pragma unroll
			for (uint32_t j = 0; j < 16 / sizeof(T); j++) {
				uint64_t logical = (uint64_t)lane * R + t * E + j;
				v[j] = logical < nbins ? (T)1 : (T)0;
			}
			base[c] = *reinterpret_cast<const uint4*>(v);
		}
	}
}

// ------------------------------------------------------------------------------------------------ count
// reverse the order of the 2-bit groups of x (32 bits = 16 bases)
__device__ __forceinline__ uint32_t rev2(uint32_t x) {
	x = __brev(x);
	return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

template <typename T, bool SAT>
__device__ __forceinline__ void bump(T* bins, uint64_t elem, uint64_t* overflow_word) {
	if constexpr (sizeof(T) == 8) {
		unsigned long long* a = reinterpret_cast<unsigned long long*>(bins) + elem;
		if constexpr (SAT) {
			unsigned long long old = *a;
			while (true) {
				if (old == ~0ull) { *overflow_word = 1; return; }
				unsigned long long prev = atomicCAS(a, old, old + 1);
				if (prev == old) return;
				old = prev;
			}
		} else {
			atomicAdd(a, 1ull);
		}
	} else {
		// 8/16/32-bit bins live inside a 32-bit word; counts cannot carry into a neighbour unless they
		// saturate, which the SAT variant excludes with a compare-and-swap.
		const uint64_t byte_off = elem * sizeof(T);
		uint32_t* a = reinterpret_cast<uint32_t*>(bins) + (byte_off >> 2);
		const uint32_t shift = (uint32_t)(byte_off & 3) * 8;
		const uint32_t mask = sizeof(T) == 4 ? 0xffffffffu : ((1u << (8 * sizeof(T))) - 1u);
		if constexpr (SAT) {
			uint32_t old = *a;
			while (true) {
				if (((old >> shift) & mask) == mask) { *overflow_word = 1; return; }
				uint32_t prev = atomicCAS(a, old, old + (1u << shift));
				if (prev == old) return;
				old = prev;
			}
		} else {
			atomicAdd(a, 1u << shift);
		}
	}
}

template <typename T, bool SAT>
__global__ void __launch_bounds__(kBlock) k_count(T* __restrict__ bins, uint8_t* __restrict__ scalars, uint64_t scalar_stride,
                                                  uint64_t slot_elems, uint64_t first_slot, int k, uint32_t E, uint32_t R,
                                                  const uint32_t* __restrict__ packed, const uint32_t* __restrict__ seg_seq,
                                                  const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ kmer_off,
                                                  uint64_t n_segs, uint64_t total_kmers) {
	for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total_kmers; g += (uint64_t)gridDim.x * blockDim.x) {
		// segment j with kmer_off[j] <= g < kmer_off[j+1]
		uint64_t lo = 0, hi = n_segs;
		while (hi - lo > 1) {
			uint64_t mid = (lo + hi) >> 1;
			if (kmer_off[mid] <= g) lo = mid; else hi = mid;
		}
		const uint64_t pos = seg_start[lo] + (g - kmer_off[lo]);     // global base offset of the k-mer's first base
		const uint64_t w = pos >> 4;
		const uint32_t sh = (uint32_t)(pos & 15) * 2;
		uint64_t window = (uint64_t)packed[w] | ((uint64_t)packed[w + 1] << 32);
		uint32_t bits = (uint32_t)(window >> sh);
		if (2 * k < 32) bits &= (1u << (2 * k)) - 1u;
		// base i sits at bits [2i, 2i+1]; the reference index has base 0 most significant (KmerHashTable.cpp:108-131)
		const uint32_t idx = rev2(bits) >> (32 - 2 * k);
		const uint64_t slot = first_slot + seg_seq[lo];
		MscSlotScalars* sc = reinterpret_cast<MscSlotScalars*>(scalars + slot * scalar_stride);
		bump<T, SAT>(bins + slot * slot_elems, msc_phys_index(idx, E, R), &sc->overflow);
	}
}

// ------------------------------------------------------------------------------------------------ finalize
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) { uint64_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
	return v;
}

template <typename T>
__global__ void __launch_bounds__(kBlock) k_finalize(const T* __restrict__ bins, uint8_t* __restrict__ scalars, uint64_t scalar_stride,
                                                     uint64_t slot_elems, uint64_t first_slot, uint32_t S, uint32_t tile_bins,
                                                     uint64_t nbins, int keep_mag) {
	const uint64_t slot = first_slot + blockIdx.x;
	const T* h = bins + slot * slot_elems;
	MscSlotScalars* sc = reinterpret_cast<MscSlotScalars*>(scalars + slot * scalar_stride);
	uint64_t* tile_sum = reinterpret_cast<uint64_t*>(scalars + slot * scalar_stride + sizeof(MscSlotScalars));
	constexpr uint32_t E = 16 / sizeof(T);
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t loads = tile_bins / (64 * E);
	uint64_t sum = 0, sq = 0, mx = 0;
	for (uint32_t t = wave; t < S; t += kBlock / 64) {
		uint64_t ts = 0;
		for (uint32_t l = 0; l < loads; l++) {
			const uint4 v = *reinterpret_cast<const uint4*>(h + (uint64_t)t * tile_bins + (uint64_t)l * 64 * E + lane * E);
			const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
			for (uint32_t j = 0; j < E; j++) {
				uint64_t p = e[j];
				ts += p;
				sq += p * p;
				mx = p > mx ? p : mx;
			}
		}
		sum += ts;
		ts = wave_sum_u64(ts);
		if (lane == 0) tile_sum[t] = ts;       // turned into an exclusive prefix by k_prefix
	}
	__shared__ uint64_t s_sum[kBlock / 64], s_sq[kBlock / 64], s_mx[kBlock / 64];
	sum = wave_sum_u64(sum);
	sq = wave_sum_u64(sq);
	mx = wave_max_u64(mx);
	if (lane == 0) { s_sum[wave] = sum; s_sq[wave] = sq; s_mx[wave] = mx; }
	__syncthreads();
	if (threadIdx.x == 0) {
		uint64_t a = 0, b = 0, c = 0;
		for (int i = 0; i < kBlock / 64; i++) { a += s_sum[i]; b += s_sq[i]; c = s_mx[i] > c ? s_mx[i] : c; }
		sc->sum = a;
		sc->sum_sq = b;
		sc->max_count = c;
		if (!keep_mag) sc->mag = a;       // DivergencePoint(pts, len) ctor sums mag (clutil/DivergencePoint.cpp:99-110)
		// stddev = sqrt(sum (p - a/N)^2 / N), clutil/Loader.cpp:158-171, from the exact integer moments
		const double N = (double)nbins;
		const double aq = (double)a / N;
		double var = ((double)b - 2.0 * aq * (double)a + N * aq * aq) / N;
		sc->stddev = sqrt(var > 0 ? var : 0);
	}
}

__global__ void __launch_bounds__(64) k_prefix(uint8_t* __restrict__ scalars, uint64_t scalar_stride, uint64_t first_slot, uint32_t S) {
	const uint64_t slot = first_slot + blockIdx.x;
	uint64_t* p = reinterpret_cast<uint64_t*>(scalars + slot * scalar_stride + sizeof(MscSlotScalars));
	const uint32_t lane = threadIdx.x;
	uint64_t carry = 0;
	for (uint32_t base = 0; base < S; base += 64) {
		const uint32_t i = base + lane;
		uint64_t v = i < S ? p[i] : 0;
		uint64_t inc = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			uint64_t o = __shfl_up(inc, off, 64);
			if ((int)lane >= off) inc += o;
		}
		if (i < S) p[i] = carry + inc - v;
		carry += __shfl(inc, 63, 64);
	}
}

// k_finalize for a FEW large slots (the rounded mean of msc_mean_nearest is one slot: a single workgroup re-reading 1 MiB took
// 107 us): one WAVE per tile writes the tile's sum into the record's tile array and its sum of squares / maximum into a scratch
// pair; k_prefix_record (one wave per slot) scans the tile sums and folds the record.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_finalize_tiles(const T* __restrict__ bins, uint8_t* __restrict__ scalars, uint64_t scalar_stride, uint64_t slot_elems,
                                                           uint64_t first_slot, uint64_t n_slots, uint32_t S, uint32_t tile_bins, uint64_t* __restrict__ sq_max) {
	constexpr uint32_t E = 16 / sizeof(T);
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t W = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	if (W >= n_slots * S) return;
	const uint64_t rel = W / S, slot = first_slot + rel;
	const uint32_t t = (uint32_t)(W % S);
	const T* h = bins + slot * slot_elems + (uint64_t)t * tile_bins;
	uint64_t ts = 0, sq = 0, mx = 0;
	for (uint32_t l = 0; l < tile_bins / (64 * E); l++) {
		const uint4 v = *reinterpret_cast<const uint4*>(h + (uint64_t)l * 64 * E + lane * E);
		const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
		for (uint32_t j = 0; j < E; j++) { const uint64_t p = e[j]; ts += p; sq += p * p; mx = p > mx ? p : mx; }
	}
	ts = wave_sum_u64(ts);
	sq = wave_sum_u64(sq);
	mx = wave_max_u64(mx);
	if (lane == 0) {
		reinterpret_cast<uint64_t*>(scalars + slot * scalar_stride + sizeof(MscSlotScalars))[t] = ts;
		sq_max[(rel * S + t) * 2] = sq;
		sq_max[(rel * S + t) * 2 + 1] = mx;
	}
}
__global__ void __launch_bounds__(64) k_prefix_record(uint8_t* __restrict__ scalars, uint64_t scalar_stride, uint64_t first_slot, uint32_t S, uint64_t nbins,
                                                      int keep_mag, const uint64_t* __restrict__ sq_max) {
	const uint64_t slot = first_slot + blockIdx.x;
	MscSlotScalars* sc = reinterpret_cast<MscSlotScalars*>(scalars + slot * scalar_stride);
	uint64_t* p = reinterpret_cast<uint64_t*>(scalars + slot * scalar_stride + sizeof(MscSlotScalars));
	const uint32_t lane = threadIdx.x;
	uint64_t carry = 0, sq = 0, mx = 0;
	for (uint32_t base = 0; base < S; base += 64) {
		const uint32_t i = base + lane;
		const uint64_t v = i < S ? p[i] : 0;
		if (i < S) { sq += sq_max[((uint64_t)blockIdx.x * S + i) * 2]; const uint64_t m_ = sq_max[((uint64_t)blockIdx.x * S + i) * 2 + 1]; mx = m_ > mx ? m_ : mx; }
		uint64_t inc = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const uint64_t o = __shfl_up(inc, off, 64);
			if ((int)lane >= off) inc += o;
		}
		if (i < S) p[i] = carry + inc - v;
		carry += __shfl(inc, 63, 64);
	}
	sq = wave_sum_u64(sq);
	mx = wave_max_u64(mx);
	if (lane == 0) {
		const uint64_t a = carry;
		sc->sum = a;
		sc->sum_sq = sq;
		sc->max_count = mx;
		if (!keep_mag) sc->mag = a;
		const double N = (double)nbins;                      // (the same expressions as k_finalize)
		const double aq = (double)a / N;
		double var = ((double)sq - 2.0 * aq * (double)a + N * aq * aq) / N;
		sc->stddev = sqrt(var > 0 ? var : 0);
	}
}

// ------------------------------------------------------------------------------------------------ small k: one fused LDS pass
// 4^k <= 16384 bins (k <= 7): the whole histogram lives in LDS as 32-bit counters. One workgroup per sequence streams
// its packed 2-bit k-mers, counts them with LDS atomics, then writes the slot ONCE in the tile-permuted layout
// (pseudocount added, saturated at max(T)) while folding sum / sum of squares / max / tile sums with wave reductions.
// No fill pass, no global atomics, no finalize re-read: N*sizeof(T) bytes written + L/4 read per sequence.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_lds(T* __restrict__ bins, uint8_t* __restrict__ scalars, uint64_t scalar_stride,
                                                     uint64_t slot_elems, uint64_t first_slot, int k, uint32_t E, uint32_t R, uint32_t S,
                                                     uint64_t nbins, const uint32_t* __restrict__ packed,
                                                     const uint64_t* __restrict__ seg_start, const uint64_t* __restrict__ kmer_off,
                                                     const uint64_t* __restrict__ seq_seg_begin) {
	extern __shared__ uint32_t s_cnt[];                  // nbins counters
	__shared__ uint64_t s_tile[16];
	__shared__ uint64_t s_red[3][kBlock / 64];
	const uint32_t seq = blockIdx.x;
	const uint64_t slot = first_slot + seq;
	for (uint32_t i = threadIdx.x; i < nbins; i += kBlock) s_cnt[i] = 0;
	if (threadIdx.x < 16) s_tile[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t sb = seq_seg_begin[seq], se = seq_seg_begin[seq + 1];
	for (uint64_t j = sb; j < se; j++) {
		const uint64_t nk = kmer_off[j + 1] - kmer_off[j];
		const uint64_t base = seg_start[j];
		for (uint64_t t = threadIdx.x; t < nk; t += kBlock) {
			const uint64_t pos = base + t;
			const uint64_t w = pos >> 4;
			const uint32_t sh = (uint32_t)(pos & 15) * 2;
			const uint64_t window = (uint64_t)packed[w] | ((uint64_t)packed[w + 1] << 32);
			uint32_t bits = (uint32_t)(window >> sh);
			bits &= (1u << (2 * k)) - 1u;
			atomicAdd(&s_cnt[rev2(bits) >> (32 - 2 * k)], 1u);
		}
	}
	__syncthreads();
	// write-out in physical order, 16 bytes per thread-iteration
	constexpr uint32_t EL = 16 / sizeof(T);
	const uint64_t tmax = sizeof(T) == 8 ? ~0ull : ((1ull << (8 * sizeof(T))) - 1);
	const uint32_t tile_bins = 64 * R;
	uint64_t sum = 0, sq = 0, mx = 0, ovf = 0;
	T* out = bins + slot * slot_elems;
	for (uint32_t c = threadIdx.x; c < slot_elems / EL; c += kBlock) {
		const uint32_t tile = (c * EL) / tile_bins;
		const uint32_t in_tile = (c * EL) % tile_bins;
		const uint32_t t = in_tile / (64 * E), lane = (in_tile % (64 * E)) / E;
		T v[EL];
		uint64_t ts = 0;
#pragma unroll
		for (uint32_t e = 0; e < EL; e++) {
			const uint64_t logical = (uint64_t)tile * tile_bins + (uint64_t)lane * R + t * E + e;
			uint64_t val = 0;
			if (logical < nbins) {
				val = 1ull + s_cnt[logical];                          // pseudocount 1 + occurrences
				if (val > tmax) { val = tmax; ovf = 1; }              // wholesaleIncrementNoOverflow stops at max(T)
			}
			v[e] = (T)val;
			ts += val; sq += val * val; mx = val > mx ? val : mx;
		}
		*reinterpret_cast<uint4*>(out + (uint64_t)c * EL) = *reinterpret_cast<const uint4*>(v);
		sum += ts;
		ts = wave_sum_u64(ts);      // a wave's 64 consecutive chunks never straddle a tile (tiles are >= 64 chunks)
		if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)&s_tile[tile], (unsigned long long)ts);
	}
	sum = wave_sum_u64(sum); sq = wave_sum_u64(sq); mx = wave_max_u64(mx); ovf = wave_max_u64(ovf);
	const uint32_t wave = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) { s_red[0][wave] = sum; s_red[1][wave] = sq; s_red[2][wave] = mx | (ovf << 63); }
	__syncthreads();
	if (threadIdx.x == 0) {
		uint64_t a = 0, b = 0, cmx = 0, o = 0;
		for (int i = 0; i < kBlock / 64; i++) { a += s_red[0][i]; b += s_red[1][i]; const uint64_t m_ = s_red[2][i] & ~(1ull << 63); cmx = m_ > cmx ? m_ : cmx; o |= s_red[2][i] >> 63; }
		MscSlotScalars* sc = reinterpret_cast<MscSlotScalars*>(scalars + slot * scalar_stride);
		sc->sum = a; sc->sum_sq = b; sc->max_count = cmx; sc->mag = a; sc->overflow = o;
		const double N = (double)nbins;
		const double aq = (double)a / N;
		const double var = ((double)b - 2.0 * aq * (double)a + N * aq * aq) / N;
		sc->stddev = sqrt(var > 0 ? var : 0);
		uint64_t* prefix = reinterpret_cast<uint64_t*>(scalars + slot * scalar_stride + sizeof(MscSlotScalars));
		uint64_t run = 0;
		for (uint32_t i = 0; i < S; i++) { prefix[i] = run; run += s_tile[i]; }
	}
}

// ------------------------------------------------------------------------------------------------ large k: sort + one streaming write
// 4^k bins do not fit LDS, but a sequence's k-mers do: an L-base sequence touches at most L bins and every other bin is the
// pseudocount. One workgroup per sequence loads its k-mer indices into LDS, bitonic-sorts them, derives every scalar of the
// record from the runs of equal indices (sum = N + k-mers, sum of squares, max, saturation, tile prefixes by binary search)
// and then writes the slot ONCE, tile by tile in the tile-permuted layout: a tile without k-mers is four 16-byte stores of
// ones per lane, a tile with k-mers is patched in a 4 KiB wave-private LDS stage first. No fill pass, no global atomics, no
// finalize re-read: N*sizeof(T) bytes written + L/4 read per sequence (the fill + count + finalize path moves 3x that).
constexpr uint32_t kSortMaxKeys = 32768;
constexpr uint32_t kStageWords = 1024;               // one 4 KiB tile (LPT == 4)
constexpr uint32_t kMaxSat = 128;                    // runs that saturate T: >= 255 equal k-mers each, so <= 128 of 32768

template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_sort(T* __restrict__ bins, uint8_t* __restrict__ scalars, uint64_t scalar_stride,
                                                       uint64_t slot_elems, uint64_t first_slot, int k, uint32_t R, uint32_t S, uint64_t nbins,
                                                       const uint32_t* __restrict__ packed, const uint64_t* __restrict__ seg_start,
                                                       const uint64_t* __restrict__ kmer_off, const uint64_t* __restrict__ seq_seg_begin,
                                                       const uint32_t* __restrict__ seq_ids, uint32_t P, const uint64_t* __restrict__ seq_meta,
                                                       unsigned long long* __restrict__ bounds) {
	extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];       // [4][kStageWords] tile stages, then P keys
	__shared__ uint64_t s_red[4][kBlock / 64];
	__shared__ uint2 s_sat[kMaxSat];                  // (bin, occurrences dropped by saturation)
	__shared__ uint32_t s_nsat;
	constexpr uint32_t E = 16 / sizeof(T);
	uint32_t* keys = s_mem + (kBlock / 64) * kStageWords;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t seq = seq_ids[blockIdx.x];
	const uint64_t slot = first_slot + seq;
	const uint32_t tile_bins = 64 * R;
	// 1. k-mer indices of this sequence (first base most significant, nonltr/KmerHashTable.cpp:108-131)
	const uint64_t sb = seq_seg_begin[seq], se = seq_seg_begin[seq + 1];
	const uint64_t k0 = kmer_off[sb];
	const uint32_t n = (uint32_t)(kmer_off[se] - k0);
	for (uint64_t j = sb; j < se; j++) {
		const uint64_t nk = kmer_off[j + 1] - kmer_off[j], base = seg_start[j];
		const uint32_t o = (uint32_t)(kmer_off[j] - k0);
		for (uint64_t t = tid; t < nk; t += kBlock) {
			const uint64_t pos = base + t;
			const uint64_t window = (uint64_t)packed[pos >> 4] | ((uint64_t)packed[(pos >> 4) + 1] << 32);
			uint32_t bits = (uint32_t)(window >> ((pos & 15) * 2));
			if (2 * k < 32) bits &= (1u << (2 * k)) - 1u;
			keys[o + t] = rev2(bits) >> (32 - 2 * k);
		}
	}
	for (uint32_t i = n + tid; i < P; i += kBlock) keys[i] = 0xffffffffu;
	{	// every stage starts as a tile of ones and is restored to that after each use
		T ones[E];
#pragma unroll
		for (uint32_t j = 0; j < E; j++) ones[j] = (T)1;
		uint4* st = reinterpret_cast<uint4*>(s_mem + wave * kStageWords);
#pragma unroll
		for (int i = 0; i < 4; i++) st[i * 64 + lane] = *reinterpret_cast<const uint4*>(ones);
	}
	if (tid == 0) s_nsat = 0;
	__syncthreads();
	// 2. bitonic sort
	for (uint32_t size = 2; size <= P; size <<= 1) {
		for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
			for (uint32_t t = tid; t < P / 2; t += kBlock) {
				const uint32_t lo = 2 * t - (t & (stride - 1));
				const uint32_t hi = lo + stride;
				const bool up = (lo & size) == 0;
				const uint32_t a = keys[lo], b = keys[hi];
				if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
			}
			__syncthreads();
		}
	}
	// 3. runs of equal keys -> the scalar record. `i` is a run head: length = upper_bound(keys[i]) - i
	const uint64_t tmax = sizeof(T) == 8 ? ~0ull : ((1ull << (8 * sizeof(T))) - 1);
	auto run_value = [&](uint32_t i, uint64_t* dropped) -> uint64_t {
		const uint32_t key = keys[i];
		uint32_t lo = i + 1, hi = n;
		if (lo < n && keys[lo] == key)                       // most k-mers of a sequence occur once: one read settles those
			while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] <= key) lo = mid + 1; else hi = mid; }
		uint64_t v = 1ull + (lo - i);                        // pseudocount + occurrences
		*dropped = 0;
		if (v > tmax) { *dropped = v - tmax; v = tmax; }     // wholesaleIncrementNoOverflow stops at max(T)
		return v;
	};
	{
		const uint32_t C = P / kBlock;                       // P >= kBlock
		const uint32_t c0 = tid * C, c1 = c0 + C < n ? c0 + C : (c0 < n ? n : c0);
		uint64_t ex = 0, sq = 0, mx = 1, ovf = 0;
		for (uint32_t i = c0; i < c1; i++) {
			if (i == 0 || keys[i] != keys[i - 1]) {
				uint64_t dropped;
				const uint64_t v = run_value(i, &dropped);
				if (dropped) {
					ovf = 1;
					const uint32_t w = atomicAdd(&s_nsat, 1u);
					if (w < kMaxSat) s_sat[w] = make_uint2(keys[i], (uint32_t)dropped);
				}
				ex += v - 1; sq += v * v - 1; mx = v > mx ? v : mx;
			}
		}
		ex = wave_sum_u64(ex); sq = wave_sum_u64(sq); mx = wave_max_u64(mx); ovf = wave_max_u64(ovf);
		if (lane == 0) { s_red[0][wave] = ex; s_red[1][wave] = sq; s_red[2][wave] = mx; s_red[3][wave] = ovf; }
	}
	__syncthreads();
	if (tid == 0) {
		uint64_t ex = 0, sq = 0, mx = 1, ovf = 0;
		for (int i = 0; i < kBlock / 64; i++) { ex += s_red[0][i]; sq += s_red[1][i]; mx = s_red[2][i] > mx ? s_red[2][i] : mx; ovf |= s_red[3][i]; }
		// the whole record is written here (length, 1-mer table and k-mer count arrive as 6 words per sequence), and the batch's
		// range bounds are folded on the device: the host neither uploads nor re-reads one record per slot
		const uint64_t* meta = seq_meta + 6ull * seq;
		const uint64_t sum = nbins + ex, sum_sq = nbins + sq;
		MscSlotScalars rec;
		rec.mag = sum; rec.length = meta[0]; rec.sum = sum; rec.sum_sq = sum_sq; rec.max_count = mx;
		rec.one_mers[0] = meta[1]; rec.one_mers[1] = meta[2]; rec.one_mers[2] = meta[3]; rec.one_mers[3] = meta[4];
		const double N = (double)nbins, aq = (double)sum / N;
		const double var = ((double)sum_sq - 2.0 * aq * (double)sum + N * aq * aq) / N;
		rec.stddev = sqrt(var > 0 ? var : 0);
		rec.overflow = ovf; rec.id = 0; rec.n_kmers = meta[5];
		rec.reserved[0] = rec.reserved[1] = rec.reserved[2] = 0;
		*reinterpret_cast<MscSlotScalars*>(scalars + slot * scalar_stride) = rec;
		atomicMax(&bounds[0], (unsigned long long)mx);
		atomicMax(&bounds[1], (unsigned long long)sum);
	}
	// 4. tile prefixes: bins before tile t sum to t * tile_bins + (k-mers with a smaller index) - (occurrences saturation dropped)
	const uint32_t nsat = s_nsat < kMaxSat ? s_nsat : kMaxSat;
	auto lower_bound = [&](uint64_t bound) -> uint32_t {
		uint32_t lo = 0, hi = n;
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)keys[mid] < bound) lo = mid + 1; else hi = mid; }
		return lo;
	};
	{
		uint64_t* prefix = reinterpret_cast<uint64_t*>(scalars + slot * scalar_stride + sizeof(MscSlotScalars));
		for (uint32_t t = tid; t < S; t += kBlock) {
			const uint64_t bound = (uint64_t)t * tile_bins;
			uint64_t v = bound + lower_bound(bound);
			for (uint32_t i = 0; i < nsat; i++) if ((uint64_t)s_sat[i].x < bound) v -= s_sat[i].y;
			prefix[t] = v;
		}
	}
	// 5. the slot, written once: wave w streams tiles [w * S/4, (w+1) * S/4)
	T* stage = reinterpret_cast<T*>(s_mem + wave * kStageWords);
	typedef uint32_t v4u __attribute__((ext_vector_type(4)));
	const v4u* stage4 = reinterpret_cast<const v4u*>(stage);
	v4u* out = reinterpret_cast<v4u*>(bins + slot * slot_elems);
	T ones[E];
#pragma unroll
	for (uint32_t j = 0; j < E; j++) ones[j] = (T)1;
	const v4u one_v = *reinterpret_cast<const v4u*>(ones);
	const uint32_t per_wave = S / (kBlock / 64);          // S is a power of four >= 16 here
	const uint32_t t0 = wave * per_wave, t1 = t0 + per_wave;
	uint32_t lo = lower_bound((uint64_t)t0 * tile_bins);
	for (uint32_t t = t0; t < t1; t++) {
		const uint64_t bound_hi = (uint64_t)(t + 1) * tile_bins;
		uint32_t cnt = 0;                                  // k-mers of this tile: keys[lo, lo + cnt)
		while (true) {
			const uint32_t i = lo + cnt + lane;
			const uint32_t key = i < n ? keys[i] : 0xffffffffu;
			const uint32_t c = (uint32_t)__popcll(__ballot(i < n && (uint64_t)key < bound_hi));
			cnt += c;
			if (c < 64) break;
		}
		v4u* dst = out + (uint64_t)t * 256 + lane;
		if (cnt == 0) {
#pragma unroll
			for (int i = 0; i < 4; i++) __builtin_nontemporal_store(one_v, dst + i * 64);
		} else {
			for (int pass = 0; pass < 2; pass++) {         // 0: patch the run values in, 1: restore the ones
				for (uint32_t b = 0; b < cnt; b += 64) {
					const uint32_t i = lo + b + lane;
					if (b + lane < cnt && (i == 0 || keys[i] != keys[i - 1])) {
						const uint32_t e = (uint32_t)(keys[i] % tile_bins);
						const uint32_t ln = e / R, r = e % R;
						const uint32_t phys = (r / E) * (64 * E) + ln * E + (r % E);
						uint64_t dropped;
						stage[phys] = pass == 0 ? (T)run_value(i, &dropped) : (T)1;
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // LDS operations of one wave execute in order; this
				__builtin_amdgcn_wave_barrier();                             // pins the compiler's order to the program's
				if (pass == 0) {
#pragma unroll
					for (int i = 0; i < 4; i++) __builtin_nontemporal_store(stage4[i * 64 + lane], dst + i * 64);
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
					__builtin_amdgcn_wave_barrier();
				}
			}
		}
		lo += cnt;
	}
}

// ------------------------------------------------------------------------------------------------ permute (upload / download)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_permute(const T* __restrict__ src, T* __restrict__ dst, uint64_t nbins, uint64_t padded,
                                                    uint32_t E, uint32_t R, int to_physical) {
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += (uint64_t)gridDim.x * blockDim.x) {
		if (to_physical) {
			// i = logical bin (or pad)
			dst[msc_phys_index(i, E, R)] = i < nbins ? src[i] : (T)0;
		} else {
			if (i < nbins) dst[i] = src[msc_phys_index(i, E, R)];
		}
	}
}

inline int grid_for(uint64_t work_items, int cap) {
	uint64_t b = (work_items + kBlock - 1) / kBlock;
	if (b < 1) b = 1;
	if (b > (uint64_t)cap) b = cap;
	return (int)b;
}

}  // namespace

template <typename F>
static inline void by_dtype(int dtype, F&& f) {
	switch (dtype) {
	case 8: f(uint8_t{}); break;
	case 16: f(uint16_t{}); break;
	case 32: f(uint32_t{}); break;
	default: f(uint64_t{}); break;
	}
}

hipError_t msc_launch_fill(hipStream_t st, void* bins, const MscLayout& L, uint64_t first_slot, uint64_t n_slots) {
	if (n_slots == 0) return hipSuccess;
	const uint64_t chunks = n_slots * (L.padded_bins / L.E);
	const int grid = grid_for(chunks, 256 * 16);
	by_dtype((int)L.esz * 8, [&](auto tag) {
		using T = decltype(tag);
		k_fill<T><<<dim3(grid), dim3(kBlock), 0, st>>>((T*)bins, first_slot, n_slots, L.padded_bins, L.nbins, L.E, L.R);
	});
	return hipGetLastError();
}

hipError_t msc_launch_count(hipStream_t st, void* bins, uint8_t* scalars, const MscLayout& L, int k, int dtype,
                            uint64_t first_slot, const uint32_t* packed_words, const uint32_t* seg_seq,
                            const uint64_t* seg_start, const uint64_t* kmer_off, uint64_t n_segs,
                            uint64_t total_kmers, bool saturating) {
	if (total_kmers == 0 || n_segs == 0) return hipSuccess;
	const int grid = grid_for(total_kmers, 256 * 16);
	const uint64_t stride = msc_scalar_stride(L.S);
	by_dtype(dtype, [&](auto tag) {
		using T = decltype(tag);
		if (saturating)
			k_count<T, true><<<dim3(grid), dim3(kBlock), 0, st>>>((T*)bins, scalars, stride, L.padded_bins, first_slot, k, L.E, L.R,
			                                                       packed_words, seg_seq, seg_start, kmer_off, n_segs, total_kmers);
		else
			k_count<T, false><<<dim3(grid), dim3(kBlock), 0, st>>>((T*)bins, scalars, stride, L.padded_bins, first_slot, k, L.E, L.R,
			                                                        packed_words, seg_seq, seg_start, kmer_off, n_segs, total_kmers);
	});
	return hipGetLastError();
}

bool msc_lds_build_supported(const MscLayout& L) { return L.nbins <= 16384 && L.S <= 16; }

hipError_t msc_launch_build_lds(hipStream_t st, void* bins, uint8_t* scalars, const MscLayout& L, int k, int dtype, uint64_t first_slot,
                                uint64_t n_seqs, const uint32_t* packed_words, const uint64_t* seg_start, const uint64_t* kmer_off,
                                const uint64_t* seq_seg_begin) {
	if (n_seqs == 0) return hipSuccess;
	const uint64_t stride = msc_scalar_stride(L.S);
	by_dtype(dtype, [&](auto tag) {
		using T = decltype(tag);
		k_build_lds<T><<<dim3((unsigned)n_seqs), dim3(kBlock), L.nbins * sizeof(uint32_t), st>>>((T*)bins, scalars, stride, L.padded_bins, first_slot, k,
		                                                                                         L.E, L.R, L.S, L.nbins, packed_words, seg_start, kmer_off,
		                                                                                         seq_seg_begin);
	});
	return hipGetLastError();
}

bool msc_sort_build_supported(const MscLayout& L, int k) { return L.LPT == 4 && L.S >= 16 && L.S % 4 == 0 && k <= 15; }
uint32_t msc_sort_build_max_kmers() { return kSortMaxKeys; }

// seq_ids: n sequence numbers (relative to first_slot) whose k-mer counts are all <= P; P a power of two in [256, 32768].
// seq_meta: 6 words per sequence of the batch (effective length, the four 1-mer counts, k-mers); bounds: two words the kernel
// folds max(max_count) and max(sum) into (zeroed by the caller).
hipError_t msc_launch_build_sort(hipStream_t st, void* bins, uint8_t* scalars, const MscLayout& L, int k, int dtype, uint64_t first_slot,
                                 const uint32_t* seq_ids, uint64_t n, uint32_t P, const uint32_t* packed_words, const uint64_t* seg_start,
                                 const uint64_t* kmer_off, const uint64_t* seq_seg_begin, const uint64_t* seq_meta, uint64_t* bounds) {
	if (n == 0) return hipSuccess;
	if (!msc_sort_build_supported(L, k) || P < (uint32_t)kBlock || P > kSortMaxKeys || (P & (P - 1))) return hipErrorInvalidValue;
	const uint64_t stride = msc_scalar_stride(L.S);
	const size_t lds = ((size_t)(kBlock / 64) * kStageWords + P) * sizeof(uint32_t);
	hipError_t e = hipSuccess;
	by_dtype(dtype, [&](auto tag) {
		using T = decltype(tag);
		if (lds > 48 * 1024) e = hipFuncSetAttribute((const void*)k_build_sort<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return;
		k_build_sort<T><<<dim3((unsigned)n), dim3(kBlock), lds, st>>>((T*)bins, scalars, stride, L.padded_bins, first_slot, k, L.R, L.S, L.nbins,
		                                                              packed_words, seg_start, kmer_off, seq_seg_begin, seq_ids, P, seq_meta,
		                                                              (unsigned long long*)bounds);
	});
	return e != hipSuccess ? e : hipGetLastError();
}

// tile_scratch (optional, 16 * n_slots * S bytes): the wave-per-tile form, for a few large slots
hipError_t msc_launch_finalize(hipStream_t st, const void* bins, uint8_t* scalars, const MscLayout& L, int dtype,
                               uint64_t first_slot, uint64_t n_slots, bool keep_mag, uint64_t* tile_scratch) {
	if (n_slots == 0) return hipSuccess;
	const uint64_t stride = msc_scalar_stride(L.S);
	if (tile_scratch) {
		const uint64_t waves = n_slots * L.S;
		by_dtype(dtype, [&](auto tag) {
			using T = decltype(tag);
			k_finalize_tiles<T><<<dim3((unsigned)((waves + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, st>>>((const T*)bins, scalars, stride, L.padded_bins, first_slot,
			                                                                                                          n_slots, L.S, L.tile_bins, tile_scratch);
		});
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
		k_prefix_record<<<dim3((unsigned)n_slots), dim3(64), 0, st>>>(scalars, stride, first_slot, L.S, L.nbins, keep_mag ? 1 : 0, tile_scratch);
		return hipGetLastError();
	}
	by_dtype(dtype, [&](auto tag) {
		using T = decltype(tag);
		k_finalize<T><<<dim3((unsigned)n_slots), dim3(kBlock), 0, st>>>((const T*)bins, scalars, stride, L.padded_bins, first_slot, L.S,
		                                                                 L.tile_bins, L.nbins, keep_mag ? 1 : 0);
	});
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	k_prefix<<<dim3((unsigned)n_slots), dim3(64), 0, st>>>(scalars, stride, first_slot, L.S);
	return hipGetLastError();
}

hipError_t msc_launch_permute(hipStream_t st, const void* src, void* dst, const MscLayout& L, int dtype, bool to_physical) {
	const int grid = grid_for(L.padded_bins, 256 * 4);
	by_dtype(dtype, [&](auto tag) {
		using T = decltype(tag);
		k_permute<T><<<dim3(grid), dim3(kBlock), 0, st>>>((const T*)src, (T*)dst, L.nbins, L.padded_bins, L.E, L.R, to_physical ? 1 : 0);
	});
	return hipGetLastError();
}
