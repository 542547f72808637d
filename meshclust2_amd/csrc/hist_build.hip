// hist_build.hip -- gfx950 kernels that build k-mer histograms in HBM.
//
// Replaces, for a whole batch of sequences at once,
//   KmerHashTable::initialize fill             nonltr/KmerHashTable.cpp:69-72
//   KmerHashTable::hash + wholesaleIncrementNoOverflow   nonltr/KmerHashTable.cpp:134-160,236-256
//   Loader::fill_table copy, DivergencePoint ctor copy + mag, stddev loop
//                                             clutil/Loader.cpp:73-77,158-171; clutil/DivergencePoint.cpp:99-110
// Three launches, all HBM-bound byte/integer work (no MFMA):
//   k_fill      : coalesced 16-byte stores of the pseudocount (1) over the batch's slots
//   k_count     : one thread per k-mer; 2k bits pulled from the packed 2-bit stream, bit-reversed into the
//                 reference's "first base most significant" index, mapped through the tile permutation
//                 (msc_layout.h) and added with one global atomic (saturating CAS only when a sequence is
//                 long enough to saturate T)
//   k_finalize  : one workgroup per slot re-reads it once: sum, sum of squares, max, per-tile sums;
//   k_prefix    : exclusive scan of the tile sums -> the tile carries the pair kernel's prefix statistic needs.
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

hipError_t msc_launch_finalize(hipStream_t st, const void* bins, uint8_t* scalars, const MscLayout& L, int dtype,
                               uint64_t first_slot, uint64_t n_slots, bool keep_mag) {
	if (n_slots == 0) return hipSuccess;
	const uint64_t stride = msc_scalar_stride(L.S);
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
