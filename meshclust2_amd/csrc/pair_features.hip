// pair_features.hip -- the pairwise-statistics kernels (1 query x m candidates), gfx950.
//
// Replaces Feature<T>::compute_all_raw and the raw statistics it calls (predict/Feature.cpp:156-171 and
// :682-695,764-777,795-811,829-841,859-871,1113-1124,1171-1184,1505-1518), normalize_cache (:137-154),
// Feature::operator() (predict/Feature.h:205-239), Trainer::classify (cluster/Trainer.cpp:112-120) and the
// pmax / && reductions of Trainer::get_close (cluster/Trainer.cpp:41-64).
//
// The reference makes one full pass over both histograms PER statistic (9-11 passes, twice per get_close
// pair). Here ONE streaming pass produces three integer reductions per (candidate, tile):
//     manh = sum |p-q|      dot = sum p*q      emd = sum |prefix(p) - prefix(q)|
// and every in-scope statistic is a closed form of those plus per-histogram constants (sum, sum of squares,
// stored mag, length) that hist_build keeps beside each slot:
//     sum min(p,q) = (sum p + sum q - manh)/2        sum (p-q)^2 = sum p^2 + sum q^2 - 2 dot
// so the kernel is pure HBM streaming: each candidate byte is read exactly once, the query tile lives in
// registers for the whole launch. Memory-bound integer work: no MFMA, no LDS.
//
// k_pair_tiles   : wave w owns tile s = w % S of the query and walks candidates g, g+G, ... (G = waves / S),
//                  reading LPT coalesced 16-byte loads per lane per candidate, software-pipelined one candidate ahead.
//                  The tile-permuted layout (msc_layout.h) gives each lane R consecutive bins, so the prefix
//                  statistic is a register scan + one 6-step DPP wave scan per tile.
// k_pair_epilogue: one wave (S > 4) or one thread per candidate folds the S partial records and evaluates the
//                  statistics / normalisation / combos / GLM in FP64 with the reference's expression order.
// k_pair_reduce  : single workgroup; arg-max of combo 0 with the serial tie order, close flags, counts.
//
// Integer ranges ("narrow" path, checked on the host before launch): largest bin <= MSC_NARROW_MAX_COUNT,
// bin sum < 2^31. Within that range every 32-bit product of the reference is exact, so results are
// bit-identical to the reference's integer accumulators.
#include <algorithm>
#include <cstddef>

#include "msc_groups.h"
#include "msc_internal.h"
#include "msc_kbits.h"
#include "msc_wave.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;

template <typename T> struct Packed;
template <> struct Packed<uint8_t> {
	static constexpr int EPW = 4, STEP = 1;
	static __device__ __forceinline__ uint32_t elem(uint32_t w, int i) { return (w >> (8 * i)) & 0xffu; }
	static __device__ __forceinline__ uint32_t sum(uint32_t w, uint32_t acc) { return __builtin_amdgcn_sad_u8(w, 0u, acc); }
	static __device__ __forceinline__ uint32_t manh(uint32_t p, uint32_t q, uint32_t acc) { return __builtin_amdgcn_sad_u8(p, q, acc); }
	static __device__ __forceinline__ uint32_t dot(uint32_t p, uint32_t q, uint32_t acc) { return __builtin_amdgcn_udot4(p, q, acc, false); }
};
template <> struct Packed<uint16_t> {
	static constexpr int EPW = 2, STEP = 1;
	static __device__ __forceinline__ uint32_t elem(uint32_t w, int i) { return (w >> (16 * i)) & 0xffffu; }
	static __device__ __forceinline__ uint32_t sum(uint32_t w, uint32_t acc) { return __builtin_amdgcn_sad_u16(w, 0u, acc); }
	static __device__ __forceinline__ uint32_t manh(uint32_t p, uint32_t q, uint32_t acc) { return __builtin_amdgcn_sad_u16(p, q, acc); }
	static __device__ __forceinline__ uint32_t dot(uint32_t p, uint32_t q, uint32_t acc) {
		return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, p), __builtin_bit_cast(u16x2, q), acc, false);
	}
};
template <> struct Packed<uint32_t> {
	static constexpr int EPW = 1, STEP = 1;
	static __device__ __forceinline__ uint32_t elem(uint32_t w, int) { return w; }
	static __device__ __forceinline__ uint32_t sum(uint32_t w, uint32_t acc) { return acc + w; }
	static __device__ __forceinline__ uint32_t manh(uint32_t p, uint32_t q, uint32_t acc) { return sad_u32(p, q, acc); }
	static __device__ __forceinline__ uint32_t dot(uint32_t p, uint32_t q, uint32_t acc) { return __umul24(p, q) + acc; }
};
template <> struct Packed<uint64_t> {   // narrow path: the high dword is zero by the host-side range check -> every other word
	static constexpr int EPW = 1, STEP = 2;
	static __device__ __forceinline__ uint32_t elem(uint32_t w, int) { return w; }
	static __device__ __forceinline__ uint32_t sum(uint32_t w, uint32_t acc) { return acc + w; }
	static __device__ __forceinline__ uint32_t manh(uint32_t p, uint32_t q, uint32_t acc) { return sad_u32(p, q, acc); }
	static __device__ __forceinline__ uint32_t dot(uint32_t p, uint32_t q, uint32_t acc) { return __umul24(p, q) + acc; }
};

// sum of the lane's run
template <typename T, int NW>
__device__ __forceinline__ uint32_t run_sum(const uint32_t* w) {
	uint32_t t = 0;
#pragma unroll
	for (int i = 0; i < NW; i += Packed<T>::STEP) t = Packed<T>::sum(w[i], t);
	return t;
}

// ---------------------------------------------------------------------------------------- streaming kernel
template <int LPT>
struct TileRegs {
	u32x4 v[LPT];
};

// LPT coalesced 16-byte loads per lane (1 KiB per wave-instruction); candidates are read exactly once -> nontemporal
template <int LPT>
__device__ __forceinline__ void load_tile(TileRegs<LPT>& t, const uint8_t* tile_base, uint32_t lane) {
	const u32x4* p = reinterpret_cast<const u32x4*>(tile_base) + lane;
#pragma unroll
	for (int l = 0; l < LPT; l++) t.v[l] = __builtin_nontemporal_load(p + 64 * l);
}

// ---- divergence statistics (`--feat slow`: jefferey_divergence, jensen_shannon; predict/Feature.cpp:1231-1263,984-1009)
// Both are sums over bins of a function of the two COUNTS (a, b) and of the two stored magnitudes only. Counts are
// small integers, so per candidate a TB x TB table of the exact per-bin terms is evaluated once (k_div_tables, 2-3
// FP64 logs per entry instead of per bin) and the streaming pass turns each bin into one LDS lookup + two FP64 adds.
// Bins with a count >= TB (rare) are re-read from memory and evaluated directly with the same expression.
struct DivTerm {
	double jd, js;
};

__device__ __forceinline__ DivTerm div_term(uint32_t cand_count, uint32_t q_count, double cand_mag, double q_mag, int order) {
	DivTerm t{0.0, 0.0};
	if (cand_count == 0 || q_count == 0) return t;        // pad bins of tiny histograms
	const bool cf = order == MSC_ORDER_CAND_FIRST;
	const double pp = cf ? (double)cand_count / cand_mag : (double)q_count / q_mag;     // (double)p.points[i] / mp
	const double pq = cf ? (double)q_count / q_mag : (double)cand_count / cand_mag;
	const double diff = pp - pq;
	t.jd = diff * log(pp / pq);
	const double avg = 0.5 * (pp + pq);
	t.js = pp * log(pp / avg) + pq * log(pq / avg);
	return t;
}

template <int TB>
__global__ void __launch_bounds__(TB * TB) k_div_tables(const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
                                                       const uint32_t* __restrict__ cand_slots, uint32_t m,
                                                       const uint8_t* __restrict__ q_scalars, int order, DivTerm* __restrict__ tables) {
	const uint32_t c = blockIdx.x;
	if (c >= m) return;
	const uint32_t slot = cand_slots ? cand_slots[c] : c;
	const double cm = (double)reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride)->mag;
	const double qm = (double)reinterpret_cast<const MscSlotScalars*>(q_scalars)->mag;
	const uint32_t j = threadIdx.x;
	tables[(uint64_t)c * TB * TB + j] = div_term(j / TB, j % TB, cm, qm, order);
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

struct MscPartialDiv {
	double jd, js;
};

// body shared by the 1 x M kernel and its batched form (one list of candidates per query, k_pair_tiles_batch); W = the wave's
// index inside its own (query, candidate list) problem
template <typename T, int LPT, bool PADDED, int TB>
__device__ __forceinline__ void pair_tiles_body(
    const uint8_t* __restrict__ cand_bins, uint64_t slot_bytes, const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
    const uint32_t* __restrict__ cand_slots, uint32_t m, const uint8_t* __restrict__ q_bins, const uint8_t* __restrict__ q_scalars,
    uint32_t S, uint32_t G, uint32_t nvalid /* bins of the (single) tile that are real */,
    int use_window, uint64_t min_len, uint64_t max_len, MscPartial* __restrict__ partials,
    const DivTerm* __restrict__ div_tables, MscPartialDiv* __restrict__ div_partials, int order, uint32_t W) {
	constexpr int E = 16 / sizeof(T);
	constexpr int R = LPT * E;
	constexpr uint32_t tile_bytes = 1024u * LPT;
	constexpr bool DIV = TB > 0;
	constexpr int TPL = DIV ? (TB * TB) / 64 : 1;          // table entries per lane

	__shared__ DivTerm s_tbl[DIV ? kWavesPerBlock * TB * TB : 1];

	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave_in_block = threadIdx.x >> 6;
	const uint32_t s = W % S;
	const uint32_t g = W / S;
	if (g >= G) return;
	DivTerm* my_tbl = s_tbl + (DIV ? wave_in_block * TB * TB : 0);

	// ---- query tile: registers for the whole launch
	TileRegs<LPT> qt;
	load_tile<LPT>(qt, q_bins + (uint64_t)s * tile_bytes, lane);
	const uint32_t* qw0 = reinterpret_cast<const uint32_t*>(&qt);
	const uint64_t* q_prefix = reinterpret_cast<const uint64_t*>(q_scalars + sizeof(MscSlotScalars));
	constexpr int NW = 4 * LPT;
	using P = Packed<T>;
	const uint32_t tq = run_sum<T, NW>(qw0);
	const uint32_t cq0 = (uint32_t)q_prefix[s] + wave_incl_scan(tq) - tq;     // prefix(q) just before this lane's run

	auto slot_of = [&](uint32_t c) -> uint32_t { return cand_slots ? cand_slots[c] : c; };
	auto in_window = [&](uint32_t slot) -> bool {
		if (!use_window) return true;
		const uint64_t len = reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride)->length;
		return len >= min_len && len <= max_len;
	};

	// ---- software pipeline, one candidate ahead
	TileRegs<LPT> nxt;
	DivTerm nxt_tbl[TPL];
	uint32_t nxt_carry = 0, nxt_slot = 0;
	bool nxt_ok = false;
	auto prefetch = [&](uint32_t cand) {
		const uint32_t slot = slot_of(cand);
		nxt_slot = slot;
		nxt_ok = in_window(slot);
		if (nxt_ok) {
			load_tile<LPT>(nxt, cand_bins + (uint64_t)slot * slot_bytes + (uint64_t)s * tile_bytes, lane);
			nxt_carry = (uint32_t)reinterpret_cast<const uint64_t*>(cand_scalars + (uint64_t)slot * scalar_stride + sizeof(MscSlotScalars))[s];
			if constexpr (DIV) {
#pragma unroll
				for (int i = 0; i < TPL; i++) nxt_tbl[i] = div_tables[(uint64_t)cand * TB * TB + i * 64 + lane];
			}
		}
	};
	uint32_t c = g;
	if (c < m) prefetch(c);
	for (; c < m; c += G) {
		const TileRegs<LPT> cur = nxt;
		const uint32_t carry = nxt_carry;
		const uint32_t cur_slot = nxt_slot;
		const bool ok = nxt_ok;
		if constexpr (DIV) {
			if (ok) {
#pragma unroll
				for (int i = 0; i < TPL; i++) my_tbl[i * 64 + lane] = nxt_tbl[i];      // wave-private: LDS ops of one wave are ordered
			}
		}
		if (c + G < m) prefetch(c + G);
		if (!ok) continue;

		const uint32_t* pw = reinterpret_cast<const uint32_t*>(&cur);
		const uint32_t tp = run_sum<T, NW>(pw);
		uint32_t cp = carry + wave_incl_scan(tp) - tp;
		uint32_t cq = cq0;
		// launder the packed query words of sub-dword types: otherwise the compiler unpacks all R query bins (and
		// their running prefix) into registers once per launch; in place, the byte/half selects are free SDWA operands
		uint32_t qw[NW];
#pragma unroll
		for (int w = 0; w < NW; w++) {
			qw[w] = qw0[w];
			if constexpr (sizeof(T) < 4) asm volatile("" : "+v"(qw[w]));
		}
		uint32_t manh = 0, dot = 0, emd = 0;
		double jd = 0.0, js = 0.0;
		uint32_t any_big = 0;
#pragma unroll
		for (int w = 0; w < NW; w += P::STEP) {
			manh = P::manh(pw[w], qw[w], manh);
			dot = P::dot(pw[w], qw[w], dot);
#pragma unroll
			for (int i = 0; i < P::EPW; i++) {
				const uint32_t p = P::elem(pw[w], i);
				const uint32_t q = P::elem(qw[w], i);
				cp += p;
				cq += q;
				if constexpr (PADDED) {
					const int r = (w / P::STEP) * P::EPW + i;
					const uint32_t e2 = sad_u32(cp, cq, emd);
					emd = (lane * R + r < nvalid) ? e2 : emd;
				} else {
					emd = sad_u32(cp, cq, emd);
				}
				if constexpr (DIV) {
					const uint32_t big = (p | q) >= (uint32_t)TB ? 1u : 0u;
					const DivTerm t = my_tbl[big ? 0u : p * TB + q];          // entry (0,0) is {0,0}
					jd += t.jd;
					js += t.js;
					any_big |= big;
				}
			}
		}
		if constexpr (DIV) {
			if (__any(any_big)) {
				// rare: some count >= TB. Re-read this lane's run from memory (no dynamic register indexing) and
				// evaluate those bins directly with the reference's expression.
				const double cm = (double)reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)cur_slot * scalar_stride)->mag;
				const double qm = (double)reinterpret_cast<const MscSlotScalars*>(q_scalars)->mag;
				const T* pt = reinterpret_cast<const T*>(cand_bins + (uint64_t)cur_slot * slot_bytes + (uint64_t)s * tile_bytes);
				const T* qtp = reinterpret_cast<const T*>(q_bins + (uint64_t)s * tile_bytes);
#pragma unroll 1
				for (int r = 0; r < R; r++) {
					const uint32_t off = (uint32_t)(r / E) * (64 * E) + lane * E + (uint32_t)(r % E);
					const uint32_t p = (uint32_t)pt[off], q = (uint32_t)qtp[off];
					if ((p | q) >= (uint32_t)TB) {
						const DivTerm t = div_term(p, q, cm, qm, order);
						jd += t.jd;
						js += t.js;
					}
				}
			}
		}
		const uint32_t manh_t = wave_total_u32(manh);
		const uint64_t dot_t = wave_total_u64(dot);
		const uint64_t emd_t = wave_total_u64(emd);
		if constexpr (DIV) {
			jd = wave_sum_f64(jd);
			js = wave_sum_f64(js);
		}
		if (lane == 0) {
			MscPartial out;
			out.manh = manh_t;
			out.dot = dot_t;
			out.emd = emd_t;
			partials[(uint64_t)c * S + s] = out;
			if constexpr (DIV) div_partials[(uint64_t)c * S + s] = MscPartialDiv{jd, js};
		}
	}
}

template <typename T, int LPT, bool PADDED, int TB>
__global__ void __launch_bounds__(kBlock) k_pair_tiles(
    const uint8_t* __restrict__ cand_bins, uint64_t slot_bytes, const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
    const uint32_t* __restrict__ cand_slots, uint32_t m, const uint8_t* __restrict__ q_bins, const uint8_t* __restrict__ q_scalars,
    uint32_t S, uint32_t G, uint32_t nvalid, int use_window, uint64_t min_len, uint64_t max_len, MscPartial* __restrict__ partials,
    const DivTerm* __restrict__ div_tables, MscPartialDiv* __restrict__ div_partials, int order) {
	const uint32_t W = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
	pair_tiles_body<T, LPT, PADDED, TB>(cand_bins, slot_bytes, cand_scalars, scalar_stride, cand_slots, m, q_bins, q_scalars, S, G, nvalid, use_window, min_len,
	                                    max_len, partials, div_tables, div_partials, order, W);
}

// Many small 1 x M problems in one launch (the update stage of the mean-shift driver: every centre against the points of
// its neighbouring clusters, cluster/ClusterFactory.cpp:288-335). Segment i = {query slot, first candidate, count, window};
// `bps` consecutive workgroups serve one segment. Partials are indexed by the candidate's position in the concatenated list.
template <typename T, int LPT, bool PADDED>
__global__ void __launch_bounds__(kBlock) k_pair_tiles_batch(
    const uint8_t* __restrict__ cand_bins, uint64_t slot_bytes, const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
    const uint32_t* __restrict__ cand_slots, const MscBatchSeg* __restrict__ segs, uint32_t bps, const uint8_t* __restrict__ qset_bins,
    uint64_t q_slot_bytes, const uint8_t* __restrict__ qset_scalars, uint64_t q_scalar_stride, uint32_t S, uint32_t G, uint32_t nvalid,
    int use_window, MscPartial* __restrict__ partials, int order) {
	const MscBatchSeg seg = segs[blockIdx.x / bps];
	if (seg.m == 0) return;
	const uint32_t W = __builtin_amdgcn_readfirstlane((blockIdx.x % bps) * kWavesPerBlock + (threadIdx.x >> 6));
	pair_tiles_body<T, LPT, PADDED, 0>(cand_bins, slot_bytes, cand_scalars, scalar_stride, cand_slots + seg.first, seg.m, qset_bins + (uint64_t)seg.q_slot * q_slot_bytes,
	                                   qset_scalars + (uint64_t)seg.q_slot * q_scalar_stride, S, G < seg.m ? G : seg.m, nvalid, use_window, seg.min_len, seg.max_len,
	                                   partials + (uint64_t)seg.first * S, nullptr, nullptr, order, W);
}

// ---------------------------------------------------------------------------------------- wide (64-bit) streaming kernel
// Fallback for histograms outside the 32-bit range of k_pair_tiles (a bin > 8191 or a bin sum >= 2^31: long
// homopolymer / microsatellite sequences). Same work decomposition and the same three reductions, but every bin is
// widened to 64 bits and every accumulator is 64-bit (mathematical values mod 2^64; the reference's own 32-bit
// products wrap in this range). Rolled loops straight from memory: correctness path, not tuned.
__device__ __forceinline__ uint64_t wave_excl_scan_u64(uint64_t v, uint32_t lane) {
	uint64_t inc = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint64_t o = __shfl_up(inc, off, 64);
		if ((int)lane >= off) inc += o;
	}
	return inc - v;
}

template <typename T, int LPT, bool DIV>
__global__ void __launch_bounds__(kBlock) k_pair_tiles_wide(
    const uint8_t* __restrict__ cand_bins, uint64_t slot_bytes, const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
    const uint32_t* __restrict__ cand_slots, uint32_t m, const uint8_t* __restrict__ q_bins, const uint8_t* __restrict__ q_scalars,
    uint32_t S, uint32_t G, uint32_t nvalid, int use_window, uint64_t min_len, uint64_t max_len, MscPartial* __restrict__ partials,
    MscPartialDiv* __restrict__ div_partials, int order) {
	constexpr int E = 16 / sizeof(T);
	constexpr int R = LPT * E;
	constexpr uint32_t tile_bytes = 1024u * LPT;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t W = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
	const uint32_t s = W % S;
	const uint32_t g = W / S;
	if (g >= G) return;
	const T* qt = reinterpret_cast<const T*>(q_bins + (uint64_t)s * tile_bytes);
	auto off_of = [&](int r) -> uint32_t { return (uint32_t)(r / E) * (64 * E) + lane * E + (uint32_t)(r % E); };
	uint64_t tq = 0;
	for (int r = 0; r < R; r++) tq += (uint64_t)qt[off_of(r)];
	const uint64_t cq0 = reinterpret_cast<const uint64_t*>(q_scalars + sizeof(MscSlotScalars))[s] + wave_excl_scan_u64(tq, lane);
	const double qm = (double)reinterpret_cast<const MscSlotScalars*>(q_scalars)->mag;
	for (uint32_t c = g; c < m; c += G) {
		const uint32_t slot = cand_slots ? cand_slots[c] : c;
		const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride);
		if (use_window && (cs->length < min_len || cs->length > max_len)) continue;
		const T* pt = reinterpret_cast<const T*>(cand_bins + (uint64_t)slot * slot_bytes + (uint64_t)s * tile_bytes);
		uint64_t tp = 0;
		for (int r = 0; r < R; r++) tp += (uint64_t)pt[off_of(r)];
		uint64_t cp = reinterpret_cast<const uint64_t*>(cand_scalars + (uint64_t)slot * scalar_stride + sizeof(MscSlotScalars))[s] + wave_excl_scan_u64(tp, lane);
		uint64_t cq = cq0, manh = 0, dot = 0, emd = 0;
		double jd = 0.0, js = 0.0;
		const double cm = (double)cs->mag;
#pragma unroll 1
		for (int r = 0; r < R; r++) {
			const uint64_t p = (uint64_t)pt[off_of(r)], q = (uint64_t)qt[off_of(r)];
			cp += p;
			cq += q;
			if (lane * R + r < nvalid) emd += cp > cq ? cp - cq : cq - cp;
			manh += p > q ? p - q : q - p;
			dot += p * q;
			if constexpr (DIV) {
				if (p != 0 && q != 0) {
					const bool cf = order == MSC_ORDER_CAND_FIRST;
					const double pp = cf ? (double)p / cm : (double)q / qm;
					const double pq = cf ? (double)q / qm : (double)p / cm;
					jd += (pp - pq) * log(pp / pq);
					const double avg = 0.5 * (pp + pq);
					js += pp * log(pp / avg) + pq * log(pq / avg);
				}
			}
		}
		manh = shfl_sum_u64_early(manh);
		dot = shfl_sum_u64_early(dot);
		emd = shfl_sum_u64_early(emd);
		if constexpr (DIV) { jd = wave_sum_f64(jd); js = wave_sum_f64(js); }
		if (lane == 0) {
			MscPartial out;
			out.manh = manh; out.dot = dot; out.emd = emd;
			partials[(uint64_t)c * S + s] = out;
			if constexpr (DIV) div_partials[(uint64_t)c * S + s] = MscPartialDiv{jd, js};
		}
	}
}

// ---------------------------------------------------------------------------------------- Q x M streaming kernel
// All-pairs shape (fastcar work(), fastcar/FC_Runner.cpp:426-471; the training table, predict/FeatureSelector.cpp:23-33):
// a wave keeps the SAME tile of TQ different queries in registers and scores every candidate tile it loads against all
// of them, so a candidate byte read from HBM serves TQ pairs (algorithmic floor N*sizeof(T)/TQ bytes per pair).
template <typename T, int LPT, int TQ, bool COMPACT>
__global__ void __launch_bounds__(kBlock) k_pair_tiles_multi(
    const uint8_t* __restrict__ cand_bins, uint64_t slot_bytes, const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
    const uint32_t* __restrict__ cand_slots, uint32_t m, const uint8_t* __restrict__ qset_bins, uint64_t q_slot_bytes,
    const uint8_t* __restrict__ qset_scalars, uint64_t q_scalar_stride, const uint32_t* __restrict__ q_slots, uint32_t n_q,
    uint32_t S, uint32_t G, MscPartial* __restrict__ partials) {
	constexpr int E = 16 / sizeof(T);
	constexpr int R = LPT * E;
	constexpr uint32_t tile_bytes = 1024u * LPT;
	constexpr bool HOIST = R <= 32;          // keep prefix(p) of the candidate run in registers, shared by the TQ queries
	constexpr int NW = 4 * LPT;
	using P = Packed<T>;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t W = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
	const uint32_t s = W % S;
	const uint32_t rest = W / S;
	const uint32_t g = rest % G;
	const uint32_t qb = rest / G;
	if (qb * TQ >= n_q) return;

	TileRegs<LPT> qt[TQ];
	uint32_t cq0[TQ];
	bool qvalid[TQ];
#pragma unroll
	for (int j = 0; j < TQ; j++) {
		const uint32_t qi = qb * TQ + j;
		qvalid[j] = qi < n_q;
		const uint32_t qslot = q_slots[qvalid[j] ? qi : qb * TQ];
		load_tile<LPT>(qt[j], qset_bins + (uint64_t)qslot * q_slot_bytes + (uint64_t)s * tile_bytes, lane);
		const uint32_t tq = run_sum<T, NW>(reinterpret_cast<const uint32_t*>(&qt[j]));
		const uint64_t* q_prefix = reinterpret_cast<const uint64_t*>(qset_scalars + (uint64_t)qslot * q_scalar_stride + sizeof(MscSlotScalars));
		cq0[j] = (uint32_t)q_prefix[s] + wave_incl_scan(tq) - tq;
	}

	TileRegs<LPT> nxt;
	uint32_t nxt_carry = 0;
	auto prefetch = [&](uint32_t cand) {
		const uint32_t slot = cand_slots ? cand_slots[cand] : cand;
		load_tile<LPT>(nxt, cand_bins + (uint64_t)slot * slot_bytes + (uint64_t)s * tile_bytes, lane);
		nxt_carry = (uint32_t)reinterpret_cast<const uint64_t*>(cand_scalars + (uint64_t)slot * scalar_stride + sizeof(MscSlotScalars))[s];
	};
	uint32_t c = g;
	if (c < m) prefetch(c);
	for (; c < m; c += G) {
		const TileRegs<LPT> cur = nxt;
		const uint32_t carry = nxt_carry;
		if (c + G < m) prefetch(c + G);
		const uint32_t* pw = reinterpret_cast<const uint32_t*>(&cur);
		const uint32_t tp = run_sum<T, NW>(pw);
		const uint32_t cp0 = carry + wave_incl_scan(tp) - tp;
		uint32_t cpv[HOIST ? R : 1];
		if constexpr (HOIST) {
			uint32_t run = cp0;
#pragma unroll
			for (int w = 0; w < NW; w += P::STEP) {
#pragma unroll
				for (int i = 0; i < P::EPW; i++) { run += P::elem(pw[w], i); cpv[(w / P::STEP) * P::EPW + i] = run; }
			}
		}
#pragma unroll
		for (int j = 0; j < TQ; j++) {
			// launder the packed query words: stops the compiler from hoisting per-bin unpacking / running prefixes of all
			// TQ query tiles out of the candidate loop (R*TQ live registers)
			uint32_t qw[NW];
#pragma unroll
			for (int w = 0; w < NW; w++) {
				qw[w] = reinterpret_cast<const uint32_t*>(&qt[j])[w];
				asm volatile("" : "+v"(qw[w]));
			}
			uint32_t cp = cp0, cq = cq0[j];
			uint32_t manh = 0, dot = 0, emd = 0;
#pragma unroll
			for (int w = 0; w < NW; w += P::STEP) {
				manh = P::manh(pw[w], qw[w], manh);
				dot = P::dot(pw[w], qw[w], dot);
#pragma unroll
				for (int i = 0; i < P::EPW; i++) {
					cq += P::elem(qw[w], i);
					if constexpr (HOIST) {
						emd = sad_u32(cpv[(w / P::STEP) * P::EPW + i], cq, emd);
					} else {
						cp += P::elem(pw[w], i);
						emd = sad_u32(cp, cq, emd);
					}
				}
			}
			const uint32_t manh_t = wave_total_u32(manh);
			uint64_t dot_t, emd_t;
			if constexpr (COMPACT) {          // host checked that the wave totals fit 32 bits
				dot_t = wave_total_u32(dot);
				emd_t = wave_total_u32(emd);
			} else {
				dot_t = wave_total_u64(dot);
				emd_t = wave_total_u64(emd);
			}
			if (lane == 0 && qvalid[j]) {
				MscPartial out;
				out.manh = manh_t;
				out.dot = dot_t;
				out.emd = emd_t;
				partials[((uint64_t)(qb * TQ + j) * m + c) * S + s] = out;
			}
			__builtin_amdgcn_sched_barrier(0);      // one query at a time: keeps the live set (and the VGPR count) small
		}
	}
}

// 32/64-bit bins (narrow range: every count < 2^16): the query side is kept PRE-DIGESTED in registers -- per query the
// absolute prefix of its run (R registers) and its bins re-packed two per word (R/2 registers). Per (candidate, query)
// that leaves one v_sad_u32 per bin for the prefix statistic and one v_sad_u16 + one v_dot2_u32_u16 per TWO bins for
// manhattan / dot: 2 VALU ops per bin instead of 4.
template <typename T, int LPT, int TQ, bool COMPACT>
__global__ void __launch_bounds__(kBlock) k_pair_tiles_multi32(
    const uint8_t* __restrict__ cand_bins, uint64_t slot_bytes, const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
    const uint32_t* __restrict__ cand_slots, uint32_t m, const uint8_t* __restrict__ qset_bins, uint64_t q_slot_bytes,
    const uint8_t* __restrict__ qset_scalars, uint64_t q_scalar_stride, const uint32_t* __restrict__ q_slots, uint32_t n_q,
    uint32_t S, uint32_t G, MscPartial* __restrict__ partials) {
	static_assert(sizeof(T) >= 4, "32/64-bit bins only");
	constexpr int E = 16 / sizeof(T);
	constexpr int R = LPT * E;
	constexpr int NW = 4 * LPT;
	constexpr int STEP = sizeof(T) / 4;
	constexpr uint32_t tile_bytes = 1024u * LPT;
	static_assert(R % 2 == 0, "bins are packed in pairs");
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t W = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
	const uint32_t s = W % S;
	const uint32_t rest = W / S;
	const uint32_t g = rest % G;
	const uint32_t qb = rest / G;
	if (qb * TQ >= n_q) return;

	uint32_t cqv[TQ][R];        // absolute inclusive prefix of each query's run
	uint32_t qpk[TQ][R / 2];    // bins 2i | 2i+1 << 16
	bool qvalid[TQ];
#pragma unroll
	for (int j = 0; j < TQ; j++) {
		const uint32_t qi = qb * TQ + j;
		qvalid[j] = qi < n_q;
		const uint32_t qslot = q_slots[qvalid[j] ? qi : qb * TQ];
		TileRegs<LPT> qt;
		load_tile<LPT>(qt, qset_bins + (uint64_t)qslot * q_slot_bytes + (uint64_t)s * tile_bytes, lane);
		const uint32_t* qw = reinterpret_cast<const uint32_t*>(&qt);
		const uint32_t tq = run_sum<T, NW>(qw);
		const uint64_t* q_prefix = reinterpret_cast<const uint64_t*>(qset_scalars + (uint64_t)qslot * q_scalar_stride + sizeof(MscSlotScalars));
		uint32_t run = (uint32_t)q_prefix[s] + wave_incl_scan(tq) - tq;
#pragma unroll
		for (int r = 0; r < R; r++) { run += qw[r * STEP]; cqv[j][r] = run; }
#pragma unroll
		for (int r = 0; r < R / 2; r++) qpk[j][r] = qw[2 * r * STEP] | (qw[(2 * r + 1) * STEP] << 16);
	}

	TileRegs<LPT> nxt;
	uint32_t nxt_carry = 0;
	auto prefetch = [&](uint32_t cand) {
		const uint32_t slot = cand_slots ? cand_slots[cand] : cand;
		load_tile<LPT>(nxt, cand_bins + (uint64_t)slot * slot_bytes + (uint64_t)s * tile_bytes, lane);
		nxt_carry = (uint32_t)reinterpret_cast<const uint64_t*>(cand_scalars + (uint64_t)slot * scalar_stride + sizeof(MscSlotScalars))[s];
	};
	uint32_t c = g;
	if (c < m) prefetch(c);
	for (; c < m; c += G) {
		const TileRegs<LPT> cur = nxt;
		const uint32_t carry = nxt_carry;
		if (c + G < m) prefetch(c + G);
		const uint32_t* pw = reinterpret_cast<const uint32_t*>(&cur);
		const uint32_t tp = run_sum<T, NW>(pw);
		uint32_t run = carry + wave_incl_scan(tp) - tp;
		uint32_t cpv[R], ppk[R / 2];
#pragma unroll
		for (int r = 0; r < R; r++) { run += pw[r * STEP]; cpv[r] = run; }
#pragma unroll
		for (int r = 0; r < R / 2; r++) ppk[r] = pw[2 * r * STEP] | (pw[(2 * r + 1) * STEP] << 16);
		// per-lane sums for a group of QG queries first, then their 3*QG wave reductions back to back: the DPP chains are
		// independent, so their latency overlaps instead of stalling a wave that has only 1-2 partners on its SIMD
		constexpr int QG = TQ < 4 ? TQ : 4;
#pragma unroll
		for (int j0 = 0; j0 < TQ; j0 += QG) {
			uint32_t manh[QG], dot[QG], emd[QG];
#pragma unroll
			for (int jj = 0; jj < QG; jj++) {
				const int j = j0 + jj;
				manh[jj] = 0; dot[jj] = 0; emd[jj] = 0;
#pragma unroll
				for (int r = 0; r < R / 2; r++) {
					manh[jj] = __builtin_amdgcn_sad_u16(ppk[r], qpk[j][r], manh[jj]);
					dot[jj] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, ppk[r]), __builtin_bit_cast(u16x2, qpk[j][r]), dot[jj], false);
					emd[jj] = sad_u32(cpv[2 * r], cqv[j][2 * r], emd[jj]);
					emd[jj] = sad_u32(cpv[2 * r + 1], cqv[j][2 * r + 1], emd[jj]);
				}
			}
			uint32_t manh_t[QG];
			uint64_t dot_t[QG], emd_t[QG];
			if constexpr (COMPACT && QG == 4) {
				const uint32_t ra = wave_sum4_rows(manh[0], manh[1], manh[2], manh[3]);
				const uint32_t rb = wave_sum4_rows(dot[0], dot[1], dot[2], dot[3]);
				const uint32_t rd = wave_sum4_rows(emd[0], emd[1], emd[2], emd[3]);
				manh_t[0] = MSC_ROW_A(ra); manh_t[1] = MSC_ROW_B(ra); manh_t[2] = MSC_ROW_C(ra); manh_t[3] = MSC_ROW_D(ra);
				dot_t[0] = MSC_ROW_A(rb); dot_t[1] = MSC_ROW_B(rb); dot_t[2] = MSC_ROW_C(rb); dot_t[3] = MSC_ROW_D(rb);
				emd_t[0] = MSC_ROW_A(rd); emd_t[1] = MSC_ROW_B(rd); emd_t[2] = MSC_ROW_C(rd); emd_t[3] = MSC_ROW_D(rd);
			} else if constexpr (COMPACT) {
				uint32_t a[QG], b[QG], d[QG];
#pragma unroll
				for (int jj = 0; jj < QG; jj++) { a[jj] = wave_incl_scan(manh[jj]); b[jj] = wave_incl_scan(dot[jj]); d[jj] = wave_incl_scan(emd[jj]); }
#pragma unroll
				for (int jj = 0; jj < QG; jj++) {
					manh_t[jj] = (uint32_t)__builtin_amdgcn_readlane((int)a[jj], 63);
					dot_t[jj] = (uint32_t)__builtin_amdgcn_readlane((int)b[jj], 63);
					emd_t[jj] = (uint32_t)__builtin_amdgcn_readlane((int)d[jj], 63);
				}
			} else {
#pragma unroll
				for (int jj = 0; jj < QG; jj++) { manh_t[jj] = wave_total_u32(manh[jj]); dot_t[jj] = wave_total_u64(dot[jj]); emd_t[jj] = wave_total_u64(emd[jj]); }
			}
			if (lane == 0) {
#pragma unroll
				for (int jj = 0; jj < QG; jj++) {
					if (qvalid[j0 + jj]) {
						MscPartial out;
						out.manh = manh_t[jj];
						out.dot = dot_t[jj];
						out.emd = emd_t[jj];
						partials[((uint64_t)(qb * TQ + j0 + jj) * m + c) * S + s] = out;
					}
				}
			}
			if constexpr (TQ > QG) __builtin_amdgcn_sched_barrier(0);
		}
	}
}

// ---------------------------------------------------------------------------------------- Q x M kernel, LDS-DMA ring
// Same arithmetic as k_pair_tiles_multi32<.., COMPACT = true>, different data movement. With the query digests in
// registers a wave has room for one candidate tile in flight (2-3 waves per SIMD -> 8-12 KiB in flight per SIMD, which
// does not cover the HBM latency). Here the candidate tiles arrive by LDS-DMA (global_load_lds_dwordx4: no destination
// VGPRs) into a wave-private ring of NB slots, NB-1 tiles ahead of the one being scored; the wave never shares its ring,
// so there is no barrier, only a counted s_waitcnt. Vector-memory operations retire in issue order (loads, stores and
// LDS-DMA on one counter), so "tile i has landed" == "at most YOUNGER(i) operations outstanding", where YOUNGER counts
// the DMA pieces issued after tile i's pieces. They are issued from inline asm in a fixed number per iteration (tail
// iterations re-fetch the last tile instead of skipping the fetch), which keeps that count exact; the compiler sees
// no vector load in the loop, so it adds no wait of its own. Partial records are 16 bytes here (three u32 totals + pad): one store per group of four queries, issued by
// the four lanes that own the row totals after wave_sum4_rows.
constexpr int kRingPieces = 5;          // VM operations per tile fetch: 4 x 1 KiB of bins + the tile's prefix carry
constexpr uint32_t kRingSlotBytes = 4096 + 256;

__device__ __forceinline__ void ring_fetch(const uint8_t* lane_src, const uint8_t* carry_src, uint32_t lds_slot) {
	uint32_t keep;      // m0 is the compiler's: saved and restored around the DMA pieces
	asm volatile(
	    "s_mov_b32 %0, m0\n\t"
	    "s_mov_b32 m0, %3\n\t"
	    "s_nop 0\n\t"
	    "global_load_lds_dwordx4 %1, off\n\t"
	    "global_load_lds_dwordx4 %1, off offset:1024\n\t"
	    "global_load_lds_dwordx4 %1, off offset:2048\n\t"
	    "global_load_lds_dwordx4 %1, off offset:3072\n\t"
	    "s_add_u32 m0, m0, 4096\n\t"
	    "s_nop 0\n\t"
	    "global_load_lds_dword %2, off\n\t"
	    "s_mov_b32 m0, %0"
	    : "=&s"(keep)
	    : "v"(lane_src), "v"(carry_src), "s"(lds_slot)
	    : "memory", "scc");
}

template <typename T, int TQ, int NB, bool P16>
__global__ void __launch_bounds__(kBlock) k_pair_tiles_multi32_ring(
    const uint8_t* __restrict__ cand_bins, uint64_t slot_bytes, const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
    const uint32_t* __restrict__ cand_slots, uint32_t m, const uint8_t* __restrict__ qset_bins, uint64_t q_slot_bytes,
    const uint8_t* __restrict__ qset_scalars, uint64_t q_scalar_stride, const uint32_t* __restrict__ q_slots, uint32_t n_q,
    uint32_t S, uint32_t G, uint32_t nqb, u32x4* __restrict__ partials16) {
	static_assert(sizeof(T) >= 4 && TQ % 4 == 0 && NB >= 2 && NB <= 4, "32/64-bit bins, query groups of four, ramp-up waits written for <= 3 tiles ahead");
	constexpr int LPT = 4;
	constexpr int E = 16 / sizeof(T);
	constexpr int R = LPT * E;
	constexpr int NW = 4 * LPT;
	constexpr int STEP = sizeof(T) / 4;
	constexpr uint32_t tile_bytes = 4096;
	constexpr int D = NB - 1;                          // tiles in flight ahead of the one being scored
	constexpr int SPI = TQ / 4;                        // partial stores per iteration
	extern __shared__ __attribute__((aligned(16))) uint8_t s_ring[];      // [waves per block][NB][kRingSlotBytes]
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = threadIdx.x >> 6;
	const uint32_t W = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + wib);
	// query blocks fastest: the waves that stream the SAME candidate tiles (one per query block) sit side by side in one
	// workgroup, so the second one finds the tile in the CU's L1 / the XCD's L2 instead of fetching it from HBM again;
	// the follower runs faster than the leader (hits instead of misses), which keeps the two from drifting apart
	const uint32_t qb = W % nqb;
	const uint32_t rest = W / nqb;
	const uint32_t s = rest % S;
	const uint32_t g = rest / S;
	if (g >= G) return;

	// prefix statistic: |prefix(candidate) - prefix(query)| per bin. The pseudocount baseline (bin index + 1) is common to
	// both sides, so the prefixes of the EXCESS counts (count - 1) give the same differences; they are bounded by the
	// number of k-mers of the sequence, and when that is < 2^16 for every histogram involved (P16, host-checked) two of
	// them share a register and one v_sad_u16 covers two bins.
	constexpr int NP = P16 ? R / 2 : R;
	uint32_t cqv[TQ][NP];       // inclusive prefix of each query's run: absolute (32-bit) or excess, packed 2i | 2i+1 << 16
	uint32_t qpk[TQ][R / 2];    // bins 2i | 2i+1 << 16
#pragma unroll
	for (int j = 0; j < TQ; j++) {
		const uint32_t qi = qb * TQ + j;
		const uint32_t qslot = q_slots[qi < n_q ? qi : qb * TQ];       // padded queries score a valid slot; their records are ignored
		TileRegs<LPT> qt;
		load_tile<LPT>(qt, qset_bins + (uint64_t)qslot * q_slot_bytes + (uint64_t)s * tile_bytes, lane);
		const uint32_t* qw = reinterpret_cast<const uint32_t*>(&qt);
		const uint32_t tq = run_sum<T, NW>(qw);
		const uint64_t* q_prefix = reinterpret_cast<const uint64_t*>(qset_scalars + (uint64_t)qslot * q_scalar_stride + sizeof(MscSlotScalars));
		uint32_t run = (uint32_t)q_prefix[s] + wave_incl_scan(tq) - tq;
		if constexpr (P16) {
			run -= s * (uint32_t)(64 * R) + lane * R;      // bins before this lane's run: the baseline they contribute
			uint32_t ev[R];
#pragma unroll
			for (int r = 0; r < R; r++) { run += qw[r * STEP] - 1u; ev[r] = run; }
#pragma unroll
			for (int r = 0; r < R / 2; r++) cqv[j][r] = ev[2 * r] | (ev[2 * r + 1] << 16);
		} else {
#pragma unroll
			for (int r = 0; r < R; r++) { run += qw[r * STEP]; cqv[j][r] = run; }
		}
#pragma unroll
		for (int r = 0; r < R / 2; r++) qpk[j][r] = qw[2 * r * STEP] | (qw[(2 * r + 1) * STEP] << 16);
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the counted waits below start from an empty queue

	uint8_t* my_ring = s_ring + (size_t)wib * NB * kRingSlotBytes;
	const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)my_ring);
	const uint32_t n_iter = (m - g + G - 1) / G;       // g < G <= m
	auto fetch = [&](uint32_t it, uint32_t slot_idx) {
		uint32_t cand = g + (it < n_iter ? it : n_iter - 1) * G;          // past the end: re-fetch the last tile (keeps the count fixed)
		const uint32_t slot = cand_slots ? cand_slots[cand] : cand;
		const uint8_t* src = cand_bins + (uint64_t)slot * slot_bytes + (uint64_t)s * tile_bytes + lane * 16u;
		const uint8_t* csrc = cand_scalars + (uint64_t)slot * scalar_stride + sizeof(MscSlotScalars) + 8ull * s;
		ring_fetch(src, csrc, ring_lds + slot_idx * kRingSlotBytes);
	};
#pragma unroll
	for (int d = 0; d < D; d++) fetch((uint32_t)d, (uint32_t)d);

	// the four lanes holding row totals after wave_sum4_rows: lane 15 -> query 0, 31 -> 2, 47 -> 1, 63 -> 3 of a group
	const uint32_t row = lane >> 4;
	const uint32_t jrow = ((row & 1) << 1) | (row >> 1);
	const bool owner = (lane & 15) == 15;
	u32x4* out_ptr[SPI];
#pragma unroll
	for (int gq = 0; gq < SPI; gq++) out_ptr[gq] = partials16 + ((uint64_t)(qb * TQ + gq * 4 + jrow) * m + g) * S + s;
	const uint64_t out_step = (uint64_t)G * S;

	uint32_t slot_idx = 0, fetch_idx = D % NB;
	for (uint32_t it = 0; it < n_iter; it++) {
		fetch(it + D, fetch_idx);                 // overwrites the slot scored in the previous iteration (its reads are complete)
		fetch_idx = fetch_idx + 1 == NB ? 0 : fetch_idx + 1;
		// Tile `it` has landed once at most D * kRingPieces operations are outstanding: D fetches were issued after it and
		// operations retire in issue order. The partial-record stores issued in between are left out of the count, which
		// keeps it valid however many records an iteration writes (and during ramp-up); the price is that up to D * SPI
		// pieces of the NEXT tile are waited for as well -- pieces requested a whole iteration ago.
		wait_vm<D * kRingPieces>();
		const uint8_t* slot_base = my_ring + slot_idx * kRingSlotBytes;
		slot_idx = slot_idx + 1 == NB ? 0 : slot_idx + 1;
		TileRegs<LPT> cur;
#pragma unroll
		for (int l = 0; l < LPT; l++) cur.v[l] = reinterpret_cast<const u32x4*>(slot_base)[64 * l + lane];
		const uint32_t carry = reinterpret_cast<const uint32_t*>(slot_base + 4096)[lane];
		const uint32_t* pw = reinterpret_cast<const uint32_t*>(&cur);
		const uint32_t tp = run_sum<T, NW>(pw);
		uint32_t run = carry + wave_incl_scan(tp) - tp;
		uint32_t cpv[NP], ppk[R / 2];
		if constexpr (P16) {
			run -= s * (uint32_t)(64 * R) + lane * R;
			uint32_t ev[R];
#pragma unroll
			for (int r = 0; r < R; r++) { run += pw[r * STEP] - 1u; ev[r] = run; }
#pragma unroll
			for (int r = 0; r < R / 2; r++) cpv[r] = ev[2 * r] | (ev[2 * r + 1] << 16);
		} else {
#pragma unroll
			for (int r = 0; r < R; r++) { run += pw[r * STEP]; cpv[r] = run; }
		}
#pragma unroll
		for (int r = 0; r < R / 2; r++) ppk[r] = pw[2 * r * STEP] | (pw[(2 * r + 1) * STEP] << 16);
#pragma unroll
		for (int gq = 0; gq < SPI; gq++) {
			uint32_t manh[4], dot[4], emd[4];
#pragma unroll
			for (int jj = 0; jj < 4; jj++) {
				const int j = gq * 4 + jj;
				manh[jj] = 0; dot[jj] = 0; emd[jj] = 0;
#pragma unroll
				for (int r = 0; r < R / 2; r++) {
					manh[jj] = __builtin_amdgcn_sad_u16(ppk[r], qpk[j][r], manh[jj]);
					dot[jj] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, ppk[r]), __builtin_bit_cast(u16x2, qpk[j][r]), dot[jj], false);
					if constexpr (P16) {
						emd[jj] = __builtin_amdgcn_sad_u16(cpv[r], cqv[j][r], emd[jj]);
					} else {
						emd[jj] = sad_u32(cpv[2 * r], cqv[j][2 * r], emd[jj]);
						emd[jj] = sad_u32(cpv[2 * r + 1], cqv[j][2 * r + 1], emd[jj]);
					}
				}
			}
			u32x4 rec;
			rec.x = wave_sum4_rows(manh[0], manh[1], manh[2], manh[3]);
			rec.y = wave_sum4_rows(dot[0], dot[1], dot[2], dot[3]);
			rec.z = wave_sum4_rows(emd[0], emd[1], emd[2], emd[3]);
			rec.w = 0;
			// s_nop: a store of more than 8 bytes reads its data registers after issue; the compiler's hazard recognizer does
			// not look inside an asm statement and may overwrite `rec` in the very next instruction
			if (owner) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(out_ptr[gq]), "v"(rec) : "memory");
			out_ptr[gq] += out_step;
			if constexpr (SPI > 1) __builtin_amdgcn_sched_barrier(0);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring must not be released with fetches in flight
}

// ---------------------------------------------------------------------------------------- epilogue
struct PairTotals {
	uint64_t manh, dot, emd;
	double jd, js;
	double markov_ab, markov_aa, markov_bb, rre;      // 4-bin group statistics (a = first argument of the reference call, b = second)
};

struct Side {
	uint64_t mag, len, sum, sq;
};

// One raw statistic from the integer reductions. `a` is the FIRST argument of the reference call, `b` the second.
__device__ double raw_stat(uint64_t flag, const PairTotals& t, const Side& a, const Side& b, uint64_t nbins, int dtype, int* err) {
	const double N = (double)nbins;
	switch (flag) {
	case MSC_FEAT_MANHATTAN:           // `int sum`, predict/Feature.cpp:864-870
		return (double)(int32_t)(uint32_t)t.manh;
	case MSC_FEAT_EUCLIDEAN:           // :1118-1123
		return sqrt((double)(a.sq + b.sq - 2 * t.dot));
	case MSC_FEAT_NORMALIZED_VECTORS:  // :1176-1183, d1*d2 in uintmax_t
		return (double)t.dot / sqrt((double)(a.sq * b.sq));
	case MSC_FEAT_PEARSON_COEFF: {     // :800-810 with dap/daq from the STORED mag
		const double dap = (double)a.mag / N;
		const double daq = (double)b.mag / N;
		const double np = (double)a.sq - 2.0 * dap * (double)a.sum + N * dap * dap;
		const double nq = (double)b.sq - 2.0 * daq * (double)b.sum + N * daq * daq;
		const double dt = (double)t.dot - daq * (double)a.sum - dap * (double)b.sum + N * dap * daq;
		return dt / sqrt(np * nq);
	}
	case MSC_FEAT_INTERSECTION: {      // :769-776
		const uint64_t min_sum = (a.sum + b.sum - t.manh) >> 1;
		return (double)(2 * min_sum) / (double)(a.mag + b.mag);
	}
	case MSC_FEAT_EMD:                 // :1510-1517
		return (double)t.emd;
	case MSC_FEAT_LENGTHD:             // :878-886
		if (a.len == 0 || b.len == 0) { *err = MSC_ERR_ZERO_LENGTH; return NAN; }
		return (double)(a.len > b.len ? a.len - b.len : b.len - a.len);
	case MSC_FEAT_KULCZYNSKI2: {       // :686-694
		const uint64_t min_sum = (a.sum + b.sum - t.manh) >> 1;
		const double ap = (double)a.mag / N;
		const double aq = (double)b.mag / N;
		const double coeff = N * (ap + aq) / (2 * ap * aq);
		return coeff * (double)min_sum;
	}
	case MSC_FEAT_SIMRATIO: {          // :834-840
		uint64_t norm2 = a.sq + b.sq - 2 * t.dot;
		if (dtype == 32) {
			// `intmax_t diff = p - q` wraps mod 2^32 for uint32_t bins BEFORE widening: every bin with
			// p < q contributes (2^32 - (q-p))^2 = (q-p)^2 - 2^33 (q-p)  (mod 2^64). Reproduced exactly:
			const int64_t sdiff = (int64_t)a.sum - (int64_t)b.sum;             // sum(p-q)
			const uint64_t neg = (uint64_t)(((int64_t)t.manh - sdiff) >> 1);   // sum over p<q of (q-p)
			norm2 -= neg << 33;
		}
		return (double)t.dot / ((double)t.dot + sqrt((double)norm2));
	}
	case MSC_FEAT_RRE_K_R:             // :1029-1062, `0.5 * (op + oq)`
		return 0.5 * t.rre;
	case MSC_FEAT_SIM_MM: {            // :1429-1455: d_markov(a, b) = log(markov(b, a) / markov(b, b)) / b.getRealMagnitude()
		const double d_ab = log(t.markov_ab / t.markov_bb) / (double)(b.mag - nbins);
		const double d_ba = log(t.markov_ab / t.markov_aa) / (double)(a.mag - nbins);
		return 1 - exp(0.5 * (d_ab + d_ba));
	}
	case MSC_FEAT_JEFFEREY_DIV:        // :1235-1262
		return t.jd;
	case MSC_FEAT_JENSEN_SHANNON:      // :988-1008, `return sum / 2`
		return t.js / 2;
	default:
		*err = MSC_ERR_UNSUPPORTED;
		return NAN;
	}
}

__device__ __forceinline__ uint64_t shfl_sum_u64(uint64_t v) {
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

// The evaluation of one pair from its integer reductions and the two histograms' scalars: raw statistics -> normalize_cache ->
// combos -> weighted sum -> logistic + bias -> close (predict/Feature.cpp:137-171, predict/Feature.h:205-239, cluster/Trainer.cpp:112-120).
// c = the pair's index in the outputs; ci / qi = its candidate / query position (group and divergence records)
__device__ void epilogue_eval(const MscEpilogueArgs& a, uint32_t c, uint32_t ci, uint32_t qi, const Side& cand, const Side& qry, uint64_t min_len, uint64_t max_len,
                              const PairTotals& t_in, MscPairOut* ret = nullptr) {
	PairTotals t = t_in;
	const Side& first = a.order == MSC_ORDER_CAND_FIRST ? cand : qry;
	const Side& second = a.order == MSC_ORDER_CAND_FIRST ? qry : cand;
	// k_pair_sparse summed over the union of stored bins only; every other bin is (1, 1)
	if (a.sparse_base) t.dot += a.sparse_base;
	if (a.div_direct) {
		// The divergence statistics of EVERY route come from the sparse merge kernels (a dense set is scored through its sparse
		// mirror): one evaluation order whatever the layout or the batching, so equal pairs give bit-equal values.
		const double* d = a.div_direct + (uint64_t)c * a.div_direct_n * 2;
		double jd = 0.0, js = 0.0;
		for (uint32_t r = 0; r < a.div_direct_n; r++) { jd += d[2 * r]; js += d[2 * r + 1]; }
		const double pp = 1.0 / (double)first.mag, pq = 1.0 / (double)second.mag;
		const double avg = 0.5 * (pp + pq);
		t.jd = jd + (double)a.div_base * ((pp - pq) * log(pp / pq));
		t.js = js + (double)a.div_base * (pp * log(pp / avg) + pq * log(pq / avg));
	}

	if (a.grp_pairs) {
		// 16 sub-range records per pair / per histogram, summed in sub-range order; markov(x, y) = total / 2 (predict/Feature.cpp:1392)
		const double* g = a.grp_pairs + (uint64_t)c * 32;
		const double* sc = a.grp_self_c + (uint64_t)ci * 16;
		const double* sq = a.grp_self_q + (uint64_t)qi * 16;
		double mk = 0.0, rre = 0.0, mc = 0.0, mq = 0.0;
		for (int r = 0; r < 16; r++) { mk += g[2 * r]; rre += g[2 * r + 1]; mc += sc[r]; mq += sq[r]; }
		t.markov_ab = mk / 2;
		t.rre = rre;
		t.markov_aa = a.order == MSC_ORDER_CAND_FIRST ? mc : mq;
		t.markov_bb = a.order == MSC_ORDER_CAND_FIRST ? mq : mc;
	}

	MscPairOut po;
	po.sum = NAN; po.csum = NAN; po.combo0 = NAN; po.status = 0; po.close = 0;
	const bool skipped = a.use_window && (cand.len < min_len || cand.len > max_len);
	int nf = __popcll(a.feat_mask);
	if (skipped) {
		po.status = 1;
		if (a.raw_out) for (int i = 0; i < nf; i++) a.raw_out[(uint64_t)c * nf + i] = NAN;
		if (a.model && a.singles_out) for (int i = 0; i < a.model->n_singles; i++) a.singles_out[(uint64_t)c * a.model->n_singles + i] = NAN;
		if (a.model && a.combos_out) for (int i = 0; i < a.model->n_combos; i++) a.combos_out[(uint64_t)c * a.model->n_combos + i] = NAN;
		if (a.pair_out) a.pair_out[c] = po;
		if (a.sum_soa) a.sum_soa[c] = NAN;
		if (a.csum_soa) a.csum_soa[c] = NAN;
		if (a.close_soa) a.close_soa[c] = 0;
		if (ret) *ret = po;
		return;
	}
	int err = 0;
	if (a.raw_out) {
		int i = 0;
		for (uint64_t f = a.feat_mask; f; f &= f - 1, i++) {
			const uint64_t bit = f & (~f + 1);
			a.raw_out[(uint64_t)c * nf + i] = raw_stat(bit, t, first, second, a.nbins, a.dtype, &err);
		}
	}
	if (a.model) {
		const MscDevModel& md = *a.model;
		double v[MSC_MAX_SINGLES];
		for (int i = 0; i < md.n_singles; i++) v[i] = raw_stat(md.single_flag[i], t, first, second, a.nbins, a.dtype, &err);
		// Feature::normalize_cache, predict/Feature.cpp:137-154
		for (int i = 0; i < md.n_singles; i++) {
			double val = (v[i] - md.mins[i]) / (md.maxs[i] - md.mins[i]);
			if (isnan(val) && err == 0) err = MSC_ERR_NAN;
			v[i] = md.is_sim[i] ? val : 1 - val;
			if (a.singles_out) a.singles_out[(uint64_t)c * md.n_singles + i] = v[i];
		}
		// Feature::operator(), predict/Feature.h:205-239 + Trainer::classify, cluster/Trainer.cpp:112-120
		double sum = md.weights[0];
		for (int col = 0; col < md.n_combos; col++) {
			const int i0 = md.combo_idx[col][0], i1 = md.combo_idx[col][1], n = md.combo_n[col];
			double d;
			switch (md.combo_kind[col]) {
			case MSC_COMBO_XY: { d = 1; d *= v[i0]; if (n == 2) d *= v[i1]; } break;
			case MSC_COMBO_X2Y2: { d = 1; d *= v[i0] * v[i0]; if (n == 2) d *= v[i1] * v[i1]; } break;
			case MSC_COMBO_XY2: d = v[i0] * v[i1] * v[i1]; break;
			default: d = v[i0] * v[i0] * v[i1]; break;
			}
			if (col == 0) po.combo0 = d;
			if (a.combos_out) a.combos_out[(uint64_t)c * md.n_combos + col] = d;
			sum += md.weights[col + 1] * d;
		}
		po.sum = sum;
		po.csum = 1.0 / (1 + exp(-sum)) + md.bias;      // GLM::logistic + _bias, predict/GLM.cpp:26-29, Predictor.cpp:315-320
		po.close = round(po.csum) > 0 ? 1 : 0;
		if (err) { po.sum = NAN; po.csum = NAN; po.combo0 = NAN; po.close = 0; }
	}
	po.status = err;
	if (a.pair_out) a.pair_out[c] = po;
	if (a.sum_soa) a.sum_soa[c] = po.sum;
	if (a.csum_soa) a.csum_soa[c] = po.csum;
	if (a.close_soa) a.close_soa[c] = (uint8_t)(err == 0 && po.close);
	if (err < 0 && a.error_word) atomicMin(a.error_word, err);
	if (ret) *ret = po;
}

// ---------------------------------------------------------------------------------------- the close flag in f32, with an error bound
// When a Q x M call wants nothing but the close flags (fastcar's Predictor::close loop, fastcar/FC_Runner.cpp:446-458; the bench's step),
// the FP64 evaluation above -- ~1 200 vector instructions per pair, most of them the Newton sequences of FP64 divide / sqrt / exp -- is
// far more than the decision needs: close <=> round(1 / (1 + exp(-s)) + 0) > 0 <=> s >= 0 up to FP64 rounding of s itself
// (predict/GLM.cpp:26-29, cluster/Trainer.cpp:112-120). screen_close evaluates s in f32 from the same EXACT integer reductions and carries
// an upper bound B of |s_f32 - s| along: every raw statistic is a short chain of well-conditioned f32 operations on exactly formed
// integers (every difference is taken in integer or FP64 arithmetic BEFORE the conversion), so its relative error is below
// kScreenEps = 2^-20 (16 units of f32 roundoff; the chains have at most 7 roundings of <= 1 ulp each); the bound then follows
// normalisation, products and the weighted sum by ordinary interval arithmetic. s_f32 > B: close. s_f32 < -B: not close. Otherwise
// (|s| within B of 0, a NaN or an infinity anywhere -- comparisons with NaN are false) the pair is UNDECIDED and the caller evaluates it
// in FP64 as before, so the flags are those of the FP64 path, pair for pair (tests/test_gpu_qxm_direct.py compares them).
constexpr float kScreenEps = 9.5367431640625e-07f;      // 2^-20

__device__ __forceinline__ float screen_raw(uint64_t flag, const PairTotals& t, const Side& a, const Side& b, uint64_t nbins, int dtype) {
	const float N = (float)nbins;
	switch (flag) {
	case MSC_FEAT_MANHATTAN:
		return (float)(int32_t)(uint32_t)t.manh;
	case MSC_FEAT_EUCLIDEAN:
		return __builtin_amdgcn_sqrtf((float)(a.sq + b.sq - 2 * t.dot));
	case MSC_FEAT_NORMALIZED_VECTORS:
		return (float)t.dot * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf((float)(a.sq * b.sq)));
	case MSC_FEAT_PEARSON_COEFF: {     // the three cancelling sums in FP64 as raw_stat forms them (adds and multiplies only), then f32
		const double Nd = (double)nbins, dap = (double)a.mag / Nd, daq = (double)b.mag / Nd;
		const double np = (double)a.sq - 2.0 * dap * (double)a.sum + Nd * dap * dap;
		const double nq = (double)b.sq - 2.0 * daq * (double)b.sum + Nd * daq * daq;
		const double dt = (double)t.dot - daq * (double)a.sum - dap * (double)b.sum + Nd * dap * daq;
		return (float)dt * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf((float)np * (float)nq));
	}
	case MSC_FEAT_INTERSECTION:
		return (float)(2 * ((a.sum + b.sum - t.manh) >> 1)) * __builtin_amdgcn_rcpf((float)(a.mag + b.mag));
	case MSC_FEAT_EMD:
		return (float)t.emd;
	case MSC_FEAT_LENGTHD:
		return (float)(a.len > b.len ? a.len - b.len : b.len - a.len);
	case MSC_FEAT_KULCZYNSKI2: {
		const float ap = (float)a.mag * __builtin_amdgcn_rcpf(N), aq = (float)b.mag * __builtin_amdgcn_rcpf(N);
		return N * (ap + aq) * __builtin_amdgcn_rcpf(2.f * ap * aq) * (float)((a.sum + b.sum - t.manh) >> 1);
	}
	case MSC_FEAT_SIMRATIO: {
		uint64_t norm2 = a.sq + b.sq - 2 * t.dot;
		if (dtype == 32) {          // (the wrap of raw_stat, in exact integers)
			const int64_t sdiff = (int64_t)a.sum - (int64_t)b.sum;
			norm2 -= (uint64_t)(((int64_t)t.manh - sdiff) >> 1) << 33;
		}
		const float d = (float)t.dot;
		return d * __builtin_amdgcn_rcpf(d + __builtin_amdgcn_sqrtf((float)norm2));
	}
	default:
		return NAN;          // (msc_model_create gives such a model no f32 image; a NaN leaves every pair undecided)
	}
}

// -> 1 close, 0 not close, -1 undecided. a / b = first / second argument of the reference call. Combo by combo, each one's one or two
// normalised statistics formed on the spot (no array of them: indexed by a model's run-time positions it becomes chains of selects --
// 350 v_cndmask in the first form of this function, more than its arithmetic; a statistic two combos share is formed twice).
__device__ __forceinline__ int screen_close(const MscDevModel& md, const PairTotals& t, const Side& a, const Side& b, uint64_t nbins, int dtype) {
	if (a.len == 0 || b.len == 0) return -1;          // (length_difference throws: the FP64 path reports it)
	auto single = [&](int i, float& v, float& dv) {          // normalised statistic i and the bound of its error
		const float r = screen_raw(md.single_flag[i], t, a, b, nbins, dtype);
		const float u = (r - md.s_min[i]) * md.s_inv[i];
		const float du = (kScreenEps * (fabsf(r) + fabsf(md.s_min[i])) + 1e-30f) * fabsf(md.s_inv[i]) + kScreenEps * fabsf(u);
		v = md.is_sim[i] ? u : 1.f - u;
		dv = du + kScreenEps * fabsf(v);
	};
	float s = md.s_w[0], B = kScreenEps * fabsf(md.s_w[0]) + 1e-12f;
	for (int col = 0; col < md.n_combos; col++) {
		const int kind = md.combo_kind[col];
		const bool two = md.combo_n[col] == 2 || kind == MSC_COMBO_XY2 || kind == MSC_COMBO_X2Y;
		float x, dx, y = 1.f, dy = 0.f;
		single(md.combo_idx[col][0], x, dx);
		if (two) single(md.combo_idx[col][1], y, dy);
		const float ax = fabsf(x), hx = ax + dx, ay = fabsf(y), hy = ay + dy;
		float d, p, h;          // the product, of absolute values, of their upper bounds
		switch (kind) {
		case MSC_COMBO_XY:   d = x * y;         p = ax * ay;           h = hx * hy; break;
		case MSC_COMBO_X2Y2: d = x * x * y * y; p = ax * ax * ay * ay; h = hx * hx * hy * hy; break;
		case MSC_COMBO_XY2:  d = x * y * y;     p = ax * ay * ay;      h = hx * hy * hy; break;
		default:             d = x * x * y;     p = ax * ax * ay;      h = hx * hx * hy; break;
		}
		const float w = md.s_w[col + 1];
		s += w * d;
		B += fabsf(w) * ((h - p) + 2.f * kScreenEps * h);
	}
	B += kScreenEps * fabsf(s);
	if (s > B) return 1;
	if (s < -B) return 0;
	return -1;
}

__device__ void epilogue_one(const MscEpilogueArgs& a, uint32_t c, const PairTotals& t, MscPairOut* ret = nullptr) {
	// c is a virtual index: query-major [n_queries][m_per_query] when several queries were scored in one launch
	uint32_t ci = c, qi = 0;
	if (a.n_queries > 1) { qi = c / a.m_per_query; ci = c % a.m_per_query; }
	const uint32_t slot = a.cand_slots ? a.cand_slots[ci] : ci;
	const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(a.cand_scalars + (uint64_t)slot * a.cand_scalar_stride);
	uint64_t min_len = a.min_len, max_len = a.max_len;
	const MscSlotScalars* qs = a.n_queries > 1
	    ? reinterpret_cast<const MscSlotScalars*>(a.qset_scalars + (uint64_t)a.q_slots[qi] * a.q_scalar_stride)
	    : reinterpret_cast<const MscSlotScalars*>(a.q_scalars);
	if (a.segs) {
		const MscBatchSeg sg = a.segs[a.pair_seg[c]];
		qs = reinterpret_cast<const MscSlotScalars*>(a.qset_scalars + (uint64_t)sg.q_slot * a.q_scalar_stride);
		min_len = sg.min_len;
		max_len = sg.max_len;
	}
	const Side cand{cs->mag, cs->length, cs->sum, cs->sum_sq};
	const Side qry{qs->mag, qs->length, qs->sum, qs->sum_sq};
	epilogue_eval(a, c, ci, qi, cand, qry, min_len, max_len, t, ret);
}

__global__ void __launch_bounds__(kBlock) k_pair_epilogue_wave(const MscEpilogueArgs a) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t c = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
	if (c >= a.m) return;
	PairTotals t{0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
	for (uint32_t s = lane; s < a.S; s += 64) {
		if (a.partials16) {
			const u32x4 p = reinterpret_cast<const u32x4*>(a.partials16)[(uint64_t)c * a.S + s];
			t.manh += p.x; t.dot += p.y; t.emd += p.z;
		} else {
			const MscPartial p = a.partials[(uint64_t)c * a.S + s];
			t.manh += p.manh; t.dot += p.dot; t.emd += p.emd;
		}
		if (a.div_partials) {
			const MscPartialDiv d = reinterpret_cast<const MscPartialDiv*>(a.div_partials)[(uint64_t)c * a.S + s];
			t.jd += d.jd; t.js += d.js;
		}
	}
	t.manh = shfl_sum_u64(t.manh);
	t.dot = shfl_sum_u64(t.dot);
	t.emd = shfl_sum_u64(t.emd);
	if (a.div_partials) { t.jd = wave_sum_f64(t.jd); t.js = wave_sum_f64(t.js); }
	if (lane == 0) epilogue_one(a, c, t);
}

// Records laid out [candidate][query group of 16][tile][query in group] (k_pair_digest_multi): one wave per (candidate,
// query group); a wave load covers four tiles x sixteen queries = 1 KiB contiguous, lane l always sees query l % 16.
__global__ void __launch_bounds__(kBlock) k_pair_epilogue_cq(const MscEpilogueArgs a) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t w = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
	const uint32_t grp = a.cq_group;
	const uint32_t nqg = (a.n_queries + grp - 1) / grp;
	if (w >= a.m_per_query * nqg) return;
	const uint32_t ci = w / nqg, qg = w % nqg;
	PairTotals t{0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
	if (grp == 32) {           // 4-byte records (manh), 32 queries per group: lane l always sees query l % 32
		const uint32_t* rec = reinterpret_cast<const uint32_t*>(a.partials_cq) + ((uint64_t)ci * nqg + qg) * a.S * 32;
		for (uint32_t i = lane; i < a.S * 32; i += 64) t.manh += rec[i];
		t.manh += __shfl_xor(t.manh, 32, 64);
	} else if (a.dot_gemm) {   // 8-byte records (manh, emd): the digest kernel left the products to the matrix cores
		const u32x2* rec = reinterpret_cast<const u32x2*>(a.partials_cq) + ((uint64_t)ci * nqg + qg) * a.S * 16;
		for (uint32_t i = lane; i < a.S * 16; i += 64) {
			const u32x2 p = rec[i];
			t.manh += p.x; t.emd += p.y;
		}
	} else {
		const u32x4* rec = reinterpret_cast<const u32x4*>(a.partials_cq) + ((uint64_t)ci * nqg + qg) * a.S * 16;
		for (uint32_t i = lane; i < a.S * 16; i += 64) {
			const u32x4 p = rec[i];
			t.manh += p.x; t.dot += p.y; t.emd += p.z;
		}
	}
	if (grp == 16) {
#pragma unroll
		for (int off = 16; off <= 32; off <<= 1) {
			t.manh += __shfl_xor(t.manh, off, 64);
			t.dot += __shfl_xor(t.dot, off, 64);
			t.emd += __shfl_xor(t.emd, off, 64);
		}
	}
	const uint32_t q = qg * grp + lane;
	if (lane < grp && q < a.n_queries) {
		if (a.dot_gemm) {          // the products were computed on the matrix cores: add the slices of the bins
			uint64_t d = 0;
			for (uint32_t s_ = 0; s_ < a.dot_slices; s_++) d += (uint64_t)(uint32_t)a.dot_gemm[((uint64_t)s_ * a.m_per_query + ci) * a.dot_stride + q];
			t.dot = d;
		}
		if (a.emd_ranks) t.emd = a.emd_ranks[(uint64_t)ci * 64 + q];
		epilogue_one(a, q * a.m_per_query + ci, t);
	}
}

// The r04 pass on the matrix cores (msc_pair_gemm.hip): P1 = the shared k-mers per slice (sum of products of presence bits), P2 = sum of
// (e_q - 1) over the queries' large bins the candidate holds, and what the bits leave out, taken from the candidate's list of large bins
// (e = count - 1 >= 2) -- see the head of that file. One wave per (candidate, 64 queries), lane = query: the list is walked by the wave
// (entries are wave-uniform, the queries' counts at an entry's bin are one coalesced 64-byte read of the transposed image).
// Everything in exact 64-bit integers.
// The integer reductions of one (candidate, query) pair from the product's output and the two lists of large bins. Wave-uniform: ci, slot,
// cmb, c_n (the candidate's list goes through the scalar cache); per lane: qq (its query row), q_slot.
__device__ __forceinline__ void bits_pair_totals(const MscEpilogueArgs& a, uint32_t ci, const uint2* cmb, uint32_t c_n, uint32_t qq, uint32_t q_slot, int64_t min_e, int64_t diff_e,
                                                 uint64_t emd, const Side& cand, const Side& qry, PairTotals& t) {
	int64_t dot_e = min_e + diff_e;
	// the candidate's large bins, four at a time: e_q (e_c - 1) for the products; min(e_q, e_c) - 1 where the query's bin is large too
	for (uint32_t i0 = 0; i0 < c_n; i0 += 4) {
		uint2 en[4];
		uint32_t x_q[4];
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) en[j] = cmb[i0 + j < c_n ? i0 + j : c_n - 1];
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {          // bit 0: the query holds the k-mer; bit 1: more than once
			const uint32_t* qt = reinterpret_cast<const uint32_t*>(a.kb_qT);
			x_q[j] = ((qt[msc_qt_word(en[j].x, 0, qq, a.kb_qn)] >> (qq & 31)) & 1u) | (((qt[msc_qt_word(en[j].x, 1, qq, a.kb_qn)] >> (qq & 31)) & 1u) << 1);
		}
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			if (i0 + j >= c_n) break;
			const int64_t e_c = en[j].y;
			int64_t e_q = x_q[j] & 1u;
			if (x_q[j] & 2u) {          // a large bin of the query too: its own list has the count
				const uint2* qmb = reinterpret_cast<const uint2*>(a.kb_q_mb) + (uint64_t)q_slot * a.kb_q_pitch;
				const uint32_t q_n = a.kb_q_mb_n[q_slot] < a.kb_q_pitch ? a.kb_q_mb_n[q_slot] : a.kb_q_pitch;
				for (uint32_t t_ = 0; t_ < q_n; t_++) if (qmb[t_].x == en[j].x) { e_q = qmb[t_].y; break; }
			}
			dot_e += (e_c - 1) * e_q;
			if (e_q >= 2) min_e += (e_q < e_c ? e_q : e_c) - 1;
		}
	}
	const uint64_t ex_c = cand.sum - a.nbins, ex_q = qry.sum - a.nbins;          // sum of the excess counts
	t.manh = ex_c + ex_q - 2 * (uint64_t)min_e;
	t.dot = a.nbins + ex_c + ex_q + (uint64_t)dot_e;
	t.emd = emd;
}

__global__ void __launch_bounds__(kBlock) k_pair_epilogue_bits(const MscEpilogueArgs a) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t groups = (a.n_queries + 63) / 64;
	const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));          // wave-uniform: the candidate's record and list go through the scalar cache
	if (w >= a.m_per_query * groups) return;
	const uint32_t ci = w / groups, q = (w % groups) * 64 + lane;
	const bool live = q < a.n_queries;
	const uint32_t qq = live ? q : 0;
	const uint32_t slot_rel = a.cand_slots ? a.cand_slots[ci] : ci;
	const uint64_t slot = a.cand_slots ? (uint64_t)slot_rel : a.kb_first + ci;
	// every load that depends on nothing else first: a wave has few instructions between its loads, so what is not in flight together
	// is a round trip each (the first form of this kernel spent two thirds of its time waiting)
	const uint32_t q_slot = a.q_slots[qq];
	const uint2* cmb = reinterpret_cast<const uint2*>(a.kb_c_mb) + slot * a.kb_c_pitch;
	const uint32_t c_n = a.kb_c_mb_n[slot] < a.kb_c_pitch ? a.kb_c_mb_n[slot] : a.kb_c_pitch;
	const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(a.cand_scalars + (uint64_t)slot_rel * a.cand_scalar_stride);
	const Side cand{cs->mag, cs->length, cs->sum, cs->sum_sq};
	int32_t part[16];
	const uint32_t ns = a.kb_slices;          // (at most 64; summed 16 at a time)
	int64_t min_e = 0;
	for (uint32_t s0 = 0; s0 < ns; s0 += 16) {
#pragma unroll
		for (uint32_t j = 0; j < 16; j++) part[j] = s0 + j < ns ? a.kb_min[((uint64_t)(s0 + j) * a.m_per_query + ci) * a.kb_qn + qq] : 0;
#pragma unroll
		for (uint32_t j = 0; j < 16; j++) min_e += part[j];
	}
	const int64_t diff_e = a.kb_diff ? a.kb_diff[(uint64_t)ci * a.kb_qn + qq] : 0;
	const uint64_t emd = a.emd_ranks ? a.emd_ranks[(uint64_t)ci * (a.emd_stride ? a.emd_stride : 64) + qq] : 0;
	const MscSlotScalars* qs = reinterpret_cast<const MscSlotScalars*>(a.qset_scalars + (uint64_t)q_slot * a.q_scalar_stride);
	const Side qry{qs->mag, qs->length, qs->sum, qs->sum_sq};
	PairTotals t{0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
	bits_pair_totals(a, ci, cmb, c_n, qq, q_slot, min_e, diff_e, emd, cand, qry, t);
	if (!live) return;
	epilogue_eval(a, q * a.m_per_query + ci, ci, q, cand, qry, a.min_len, a.max_len, t);
}

// The same pass when only the close flags are wanted and the model has an f32 image (screen_close above): a wave takes kScreenChunk
// CONSECUTIVE candidates x 64 queries (lane = query), so that the queries' side -- slot, scalar record, everything the compiler derives
// from them -- is fetched once per chunk instead of once per candidate, and a lane's flags of the chunk leave as ONE 16-byte store into its
// query's row (a byte per (lane, candidate) was 64 partial lines per store instruction). A pair the screen leaves undecided gets the
// flag byte kScreenOpen; k_pair_epilogue_bits_open, queued behind this kernel, finds those bytes and evaluates their pairs in FP64 --
// as its own launch, because the FP64 evaluation inlined here took the kernel from 131 to 212 registers for one pair in thousands.
constexpr uint32_t kScreenOpen = 2;          // flag byte of a pair the screen left undecided, until k_pair_epilogue_bits_open has evaluated it
template <uint32_t kScreenChunk>
__global__ void __launch_bounds__(kBlock, 4) k_pair_epilogue_bits_screen(const MscEpilogueArgs a) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t groups = (a.n_queries + 63) / 64;
	const uint32_t chunks = (a.m_per_query + kScreenChunk - 1) / kScreenChunk;
	const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
	if (w >= chunks * groups) return;
	const uint32_t c0 = (w / groups) * kScreenChunk, q = (w % groups) * 64 + lane;
	const uint32_t nc = a.m_per_query - c0 < kScreenChunk ? a.m_per_query - c0 : kScreenChunk;
	const bool live = q < a.n_queries;
	const uint32_t qq = live ? q : 0;
	const uint32_t q_slot = a.q_slots[qq];
	const MscSlotScalars* qs = reinterpret_cast<const MscSlotScalars*>(a.qset_scalars + (uint64_t)q_slot * a.q_scalar_stride);
	const Side qry{qs->mag, qs->length, qs->sum, qs->sum_sq};
	const uint32_t ns = a.kb_slices, estride = a.emd_stride ? a.emd_stride : 64;
	uint32_t fl[4] = {0, 0, 0, 0};
	// one candidate's loads and integer reductions (the candidate's record and list are wave-uniform: scalar cache)
	auto totals_of = [&](uint32_t ci, int64_t min_e, int64_t diff_e, uint64_t emd, Side& cand, PairTotals& t) {
		const uint32_t slot_rel = a.cand_slots ? a.cand_slots[ci] : ci;
		const uint64_t slot = a.cand_slots ? (uint64_t)slot_rel : a.kb_first + ci;
		const uint2* cmb = reinterpret_cast<const uint2*>(a.kb_c_mb) + slot * a.kb_c_pitch;
		const uint32_t c_n = a.kb_c_mb_n[slot] < a.kb_c_pitch ? a.kb_c_mb_n[slot] : a.kb_c_pitch;
		const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(a.cand_scalars + (uint64_t)slot_rel * a.cand_scalar_stride);
		cand = Side{cs->mag, cs->length, cs->sum, cs->sum_sq};
		bits_pair_totals(a, ci, cmb, c_n, qq, q_slot, min_e, diff_e, emd, cand, qry, t);
	};
	auto sums_of = [&](uint32_t ci, int64_t& min_e, int64_t& diff_e, uint64_t& emd) {
		int64_t v = a.kb_min[(uint64_t)ci * a.kb_qn + qq];
		for (uint32_t s_ = 1; s_ < ns; s_++) v += a.kb_min[((uint64_t)s_ * a.m_per_query + ci) * a.kb_qn + qq];
		min_e = v;
		diff_e = a.kb_diff ? a.kb_diff[(uint64_t)ci * a.kb_qn + qq] : 0;
		emd = a.emd_ranks ? a.emd_ranks[(uint64_t)ci * estride + qq] : 0;
	};
#pragma unroll 1
	for (uint32_t j0 = 0; j0 < nc; j0 += 4) {
		int64_t min4[4], diff4[4];          // four candidates' loads in flight together
		uint64_t emd4[4];
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) sums_of(c0 + (j0 + j < nc ? j0 + j : nc - 1), min4[j], diff4[j], emd4[j]);
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			if (j0 + j >= nc) break;
			Side cand;
			PairTotals t{0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
			totals_of(c0 + j0 + j, min4[j], diff4[j], emd4[j], cand, t);
			const int verdict = a.order == MSC_ORDER_CAND_FIRST ? screen_close(*a.model, t, cand, qry, a.nbins, a.dtype) : screen_close(*a.model, t, qry, cand, a.nbins, a.dtype);
			fl[(j0 + j) >> 2] |= (uint32_t)(verdict < 0 ? kScreenOpen : verdict) << (8 * ((j0 + j) & 3));
			__builtin_amdgcn_sched_barrier(0);          // (one candidate's evaluation at a time: interleaved, the four take 212 registers)
		}
	}
	if (!live) return;
	uint8_t* row = a.close_soa + (uint64_t)q * a.m_per_query + c0;
	if (kScreenChunk == 16 && nc == kScreenChunk && ((uintptr_t)row & 15) == 0) *reinterpret_cast<uint4*>(row) = make_uint4(fl[0], fl[1], fl[2], fl[3]);
	else if (kScreenChunk == 8 && nc == kScreenChunk && ((uintptr_t)row & 7) == 0) *reinterpret_cast<uint2*>(row) = make_uint2(fl[0], fl[1]);
	else if (kScreenChunk == 4 && nc == kScreenChunk && ((uintptr_t)row & 3) == 0) *reinterpret_cast<uint32_t*>(row) = fl[0];
	else for (uint32_t j = 0; j < nc; j++) row[j] = (uint8_t)((fl[j >> 2] >> (8 * (j & 3))) & 0xffu);
}

// Behind k_pair_epilogue_bits_screen: a thread reads 16 flags of one query's row; where a byte says kScreenOpen, the pair's integer
// reductions are formed again and evaluated in FP64 (epilogue_eval writes the flag).
// (five waves per SIMD asked of the compiler = at most 96 registers: with the FP64 evaluation inlined the kernel took 256, and a wave of
// 256 registers finds no room on a CU that holds three workgroups of the product kernel -- queued beside it, this 20 us scan waited a
// millisecond for one to leave; the evaluation spills, and runs for one pair in thousands)
__global__ void __launch_bounds__(kBlock, 5) k_pair_epilogue_bits_open(const MscEpilogueArgs a) {
	const uint64_t total = (uint64_t)a.n_queries * a.m_per_query;
	const uint64_t at = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
	if (at >= total) return;
	const uint32_t n = total - at < 16 ? (uint32_t)(total - at) : 16;
	uint32_t w[4] = {0, 0, 0, 0};
	if (n == 16 && ((uintptr_t)(a.close_soa + at) & 15) == 0) {
		const uint4 v = *reinterpret_cast<const uint4*>(a.close_soa + at);
		w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
	} else for (uint32_t j = 0; j < n; j++) w[j >> 2] |= (uint32_t)a.close_soa[at + j] << (8 * (j & 3));
	if (!(((w[0] | w[1] | w[2] | w[3]) >> 1) & 0x01010101u)) return;          // no byte holds 2
	const uint32_t ns = a.kb_slices, estride = a.emd_stride ? a.emd_stride : 64;
	for (uint32_t j = 0; j < n; j++) {
		if (((w[j >> 2] >> (8 * (j & 3))) & 0xffu) != kScreenOpen) continue;
		const uint32_t q = (uint32_t)((at + j) / a.m_per_query), ci = (uint32_t)((at + j) % a.m_per_query);
		const uint32_t q_slot = a.q_slots[q];
		const uint32_t slot_rel = a.cand_slots ? a.cand_slots[ci] : ci;
		const uint64_t slot = a.cand_slots ? (uint64_t)slot_rel : a.kb_first + ci;
		const uint2* cmb = reinterpret_cast<const uint2*>(a.kb_c_mb) + slot * a.kb_c_pitch;
		const uint32_t c_n = a.kb_c_mb_n[slot] < a.kb_c_pitch ? a.kb_c_mb_n[slot] : a.kb_c_pitch;
		const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(a.cand_scalars + (uint64_t)slot_rel * a.cand_scalar_stride);
		const MscSlotScalars* qs = reinterpret_cast<const MscSlotScalars*>(a.qset_scalars + (uint64_t)q_slot * a.q_scalar_stride);
		const Side cand{cs->mag, cs->length, cs->sum, cs->sum_sq}, qry{qs->mag, qs->length, qs->sum, qs->sum_sq};
		int64_t min_e = 0;
		for (uint32_t s_ = 0; s_ < ns; s_++) min_e += a.kb_min[((uint64_t)s_ * a.m_per_query + ci) * a.kb_qn + q];
		const int64_t diff_e = a.kb_diff ? a.kb_diff[(uint64_t)ci * a.kb_qn + q] : 0;
		const uint64_t emd = a.emd_ranks ? a.emd_ranks[(uint64_t)ci * estride + q] : 0;
		PairTotals t{0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
		bits_pair_totals(a, ci, cmb, c_n, q, q_slot, min_e, diff_e, emd, cand, qry, t);
		epilogue_eval(a, q * a.m_per_query + ci, ci, q, cand, qry, a.min_len, a.max_len, t);
	}
}

__global__ void __launch_bounds__(kBlock) k_pair_epilogue_thread(const MscEpilogueArgs a) {
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= a.m) return;
	PairTotals t{0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
	for (uint32_t s = 0; s < a.S; s++) {
		if (a.partials16) {
			const u32x4 p = reinterpret_cast<const u32x4*>(a.partials16)[(uint64_t)c * a.S + s];
			t.manh += p.x; t.dot += p.y; t.emd += p.z;
		} else {
			const MscPartial p = a.partials[(uint64_t)c * a.S + s];
			t.manh += p.manh; t.dot += p.dot; t.emd += p.emd;
		}
		if (a.div_partials) {
			const MscPartialDiv d = reinterpret_cast<const MscPartialDiv*>(a.div_partials)[(uint64_t)c * a.S + s];
			t.jd += d.jd; t.js += d.js;
		}
	}
	epilogue_one(a, c, t);
}

// ---------------------------------------------------------------------------------------- reduce
// get_close: arg-max of combo0 over candidates that were scored, strict '>' in window order (first maximum wins),
//            starting from (-1, none)                                  cluster/Trainer.cpp:26-37,59
// merge    : among close candidates, later index wins ties, starting from (0, DBL_MIN)   cluster/Trainer.cpp:76-79,103-105
struct Best { double sim; int64_t pos; };

__device__ __forceinline__ Best better(Best x, Best y, int mode) {
	if (mode == MSC_REDUCE_GET_CLOSE) {
		if (y.sim > x.sim || (y.sim == x.sim && y.pos >= 0 && (x.pos < 0 || y.pos < x.pos))) return y;
		return x;
	}
	if (y.pos < 0) return x;
	if (x.pos < 0) return y;
	if (y.sim > x.sim || (y.sim == x.sim && y.pos > x.pos)) return y;
	return x;
}

// (key, optional: what decides between equal similarities instead of the index -- a window pass hands in the candidates' window positions,
// its list being in no particular order; best_pos then IS that position)
__global__ void __launch_bounds__(1024) k_pair_reduce(const MscPairOut* __restrict__ po, uint32_t m, int mode, int64_t begin,
                                                      uint8_t* __restrict__ flags_out, MscReduceOut* __restrict__ out, const uint32_t* __restrict__ key) {
	__shared__ Best s_best[1024];
	__shared__ unsigned long long s_nclose;
	__shared__ int s_err;
	if (threadIdx.x == 0) { s_nclose = 0; s_err = 0; }
	__syncthreads();
	Best b{mode == MSC_REDUCE_GET_CLOSE ? -1.0 : 0.0, -1};
	unsigned long long nclose = 0;
	int err = 0;
	for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
		const MscPairOut p = po[i];
		const bool scored = p.status == 0;
		if (p.status < 0 && p.status < err) err = p.status;
		if (flags_out) flags_out[i] = (scored && p.close) ? 1 : 0;
		if (!scored) continue;
		nclose += p.close ? 1 : 0;
		if (mode == MSC_REDUCE_GET_CLOSE) {
			if (p.combo0 > -1.0) b = better(b, Best{p.combo0, key ? (int64_t)key[i] : (int64_t)i}, mode);
		} else {
			// best = best.second > dist ? best : (i, dist), starting at DBL_MIN
			if (p.close && !(2.2250738585072014e-308 > p.combo0)) b = better(b, Best{p.combo0, begin + (int64_t)i}, mode);
		}
	}
	s_best[threadIdx.x] = b;
	if (nclose) atomicAdd(&s_nclose, nclose);
	if (err) atomicMin(&s_err, err);
	__syncthreads();
	for (int stride = 512; stride >= 1; stride >>= 1) {
		if ((int)threadIdx.x < stride) s_best[threadIdx.x] = better(s_best[threadIdx.x], s_best[threadIdx.x + stride], mode);
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		MscReduceOut r;
		r.best_sim = s_best[0].pos >= 0 ? s_best[0].sim : (mode == MSC_REDUCE_GET_CLOSE ? -1.0 : 2.2250738585072014e-308);
		r.best_pos = s_best[0].pos >= 0 ? s_best[0].pos : (mode == MSC_REDUCE_GET_CLOSE ? -1 : 0);
		r.any_close = s_nclose > 0;
		r.n_close = s_nclose;
		r.first_error = s_err;
		*out = r;
	}
}

// The same in two stages for long windows (one workgroup walking 100 000 records and writing their flags took 50 us behind a 320 us
// kernel): workgroup b folds records [b * per, (b + 1) * per) into parts[b]; k_pair_reduce_fold folds the parts. `better` breaks
// ties by position, so the order of folding does not matter.
__global__ void __launch_bounds__(1024) k_pair_reduce_part(const MscPairOut* __restrict__ po, uint32_t m, uint32_t per, int mode, int64_t begin,
                                                           uint8_t* __restrict__ flags_out, ReducePart* __restrict__ parts, const uint32_t* __restrict__ key) {
	__shared__ Best s_best[1024];
	__shared__ unsigned long long s_nclose;
	__shared__ int s_err;
	if (threadIdx.x == 0) { s_nclose = 0; s_err = 0; }
	__syncthreads();
	Best b{mode == MSC_REDUCE_GET_CLOSE ? -1.0 : 0.0, -1};
	unsigned long long nclose = 0;
	int err = 0;
	const uint32_t lo = blockIdx.x * per, hi = lo + per < m ? lo + per : m;
	for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
		const MscPairOut p = po[i];
		const bool scored = p.status == 0;
		if (p.status < 0 && p.status < err) err = p.status;
		if (flags_out) flags_out[i] = (scored && p.close) ? 1 : 0;
		if (!scored) continue;
		nclose += p.close ? 1 : 0;
		if (mode == MSC_REDUCE_GET_CLOSE) {
			if (p.combo0 > -1.0) b = better(b, Best{p.combo0, key ? (int64_t)key[i] : (int64_t)i}, mode);
		} else {
			if (p.close && !(2.2250738585072014e-308 > p.combo0)) b = better(b, Best{p.combo0, begin + (int64_t)i}, mode);
		}
	}
	s_best[threadIdx.x] = b;
	if (nclose) atomicAdd(&s_nclose, nclose);
	if (err) atomicMin(&s_err, err);
	__syncthreads();
	for (int stride = 512; stride >= 1; stride >>= 1) {
		if ((int)threadIdx.x < stride) s_best[threadIdx.x] = better(s_best[threadIdx.x], s_best[threadIdx.x + stride], mode);
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		ReducePart r;
		r.sim = s_best[0].sim; r.pos = s_best[0].pos; r.nclose = s_nclose; r.err = s_err; r.wpos = 0;
		parts[blockIdx.x] = r;
	}
}
__global__ void __launch_bounds__(256) k_pair_reduce_fold(const ReducePart* __restrict__ parts, uint32_t n_parts, int mode, MscReduceOut* __restrict__ out) {
	__shared__ Best s_best[256];
	__shared__ unsigned long long s_nclose;
	__shared__ int s_err;
	if (threadIdx.x == 0) { s_nclose = 0; s_err = 0; }
	__syncthreads();
	Best b{mode == MSC_REDUCE_GET_CLOSE ? -1.0 : 0.0, -1};
	if (threadIdx.x < n_parts) {
		const ReducePart p = parts[threadIdx.x];
		b = Best{p.sim, p.pos};
		if (p.nclose) atomicAdd(&s_nclose, p.nclose);
		if (p.err) atomicMin(&s_err, p.err);
	}
	s_best[threadIdx.x] = b;
	__syncthreads();
	for (int stride = 128; stride >= 1; stride >>= 1) {
		if ((int)threadIdx.x < stride) s_best[threadIdx.x] = better(s_best[threadIdx.x], s_best[threadIdx.x + stride], mode);
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		MscReduceOut r;
		r.best_sim = s_best[0].pos >= 0 ? s_best[0].sim : (mode == MSC_REDUCE_GET_CLOSE ? -1.0 : 2.2250738585072014e-308);
		r.best_pos = s_best[0].pos >= 0 ? s_best[0].pos : (mode == MSC_REDUCE_GET_CLOSE ? -1 : 0);
		r.any_close = s_nclose > 0;
		r.n_close = s_nclose;
		r.first_error = s_err;
		*out = r;
	}
}

// Epilogue + first reduce stage (+ the window's close pass) in ONE kernel, for the step-serial loop: a get_close pass over a window is a
// chain of small kernels the host waits for once per step (profiles/r04_notes.md 5), and k_pair_epilogue_thread -> k_pair_reduce_part ->
// k_window_close were three of them that all make one pass over the same m pairs. A thread evaluates its pair (epilogue_one), keeps the
// record in registers, folds it into its workgroup's part and -- cl.pos != nullptr: the window's bookkeeping, msc_window.hip -- takes
// a close candidate's position off the alive list and appends it to the host's list. k_pair_reduce_fold2 folds the parts and turns the
// best candidate's index into a position.
__global__ void __launch_bounds__(kBlock) k_pair_epilogue_reduce_part(const MscEpilogueArgs a, int mode, int64_t begin, uint8_t* __restrict__ flags_out,
                                                                      ReducePart* __restrict__ parts, const MscCloseList cl) {
	__shared__ Best s_best[kBlock];
	__shared__ unsigned long long s_nclose;
	__shared__ int s_err;
	if (threadIdx.x == 0) { s_nclose = 0; s_err = 0; }
	__syncthreads();
	Best b{mode == MSC_REDUCE_GET_CLOSE ? -1.0 : 0.0, -1};
	unsigned long long nclose = 0;
	int err = 0;
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < a.m; c += gridDim.x * blockDim.x) {
		PairTotals t{0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
		for (uint32_t s = 0; s < a.S; s++) {
			const MscPartial p = a.partials[(uint64_t)c * a.S + s];
			t.manh += p.manh; t.dot += p.dot; t.emd += p.emd;
			if (a.div_partials) {
				const MscPartialDiv d = reinterpret_cast<const MscPartialDiv*>(a.div_partials)[(uint64_t)c * a.S + s];
				t.jd += d.jd; t.js += d.js;
			}
		}
		MscPairOut p;
		epilogue_one(a, c, t, &p);
		const bool scored = p.status == 0;
		if (p.status < 0 && p.status < err) err = p.status;
		const bool close = scored && p.close;
		if (flags_out) flags_out[c] = close ? 1 : 0;
		if (close && cl.pos) {
			const uint32_t at = cl.pos[c];
			cl.alive[at] = 0;
			cl.out[2 + atomicAdd(cl.counter, 1u)] = at;
		}
		if (!scored) continue;
		nclose += p.close ? 1 : 0;
		if (mode == MSC_REDUCE_GET_CLOSE) {
			// (a window pass: ties go to the lower WINDOW POSITION -- the list of candidates is in no particular order, msc_window.hip)
			if (p.combo0 > -1.0) b = better(b, Best{p.combo0, cl.pos ? (int64_t)cl.pos[c] : (int64_t)c}, mode);
		} else {
			if (p.close && !(2.2250738585072014e-308 > p.combo0)) b = better(b, Best{p.combo0, begin + (int64_t)c}, mode);
		}
	}
	s_best[threadIdx.x] = b;
	if (nclose) atomicAdd(&s_nclose, nclose);
	if (err) atomicMin(&s_err, err);
	__syncthreads();
	for (int stride = kBlock / 2; stride >= 1; stride >>= 1) {
		if ((int)threadIdx.x < stride) s_best[threadIdx.x] = better(s_best[threadIdx.x], s_best[threadIdx.x + stride], mode);
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		ReducePart r;
		r.sim = s_best[0].sim; r.pos = s_best[0].pos; r.nclose = s_nclose; r.err = s_err;
		r.wpos = cl.pos && s_best[0].pos >= 0 ? (uint32_t)s_best[0].pos + 1u : 0u;          // (a window pass: pos IS the window position)
		parts[blockIdx.x] = r;
	}
}
__global__ void __launch_bounds__(256) k_pair_reduce_fold2(const ReducePart* __restrict__ parts, uint32_t n_parts, int mode, MscReduceOut* __restrict__ out, const MscCloseList cl) {
	__shared__ Best s_best[256];
	__shared__ unsigned long long s_nclose;
	__shared__ int s_err;
	if (threadIdx.x == 0) { s_nclose = 0; s_err = 0; }
	__syncthreads();
	Best b{mode == MSC_REDUCE_GET_CLOSE ? -1.0 : 0.0, -1};
	unsigned long long nclose = 0;
	int err = 0;
	for (uint32_t i = threadIdx.x; i < n_parts; i += 256) {
		const ReducePart p = parts[i];
		b = better(b, Best{p.sim, p.pos}, mode);
		nclose += p.nclose;
		if (p.err < err) err = p.err;
	}
	if (nclose) atomicAdd(&s_nclose, nclose);
	if (err) atomicMin(&s_err, err);
	s_best[threadIdx.x] = b;
	__syncthreads();
	for (int stride = 128; stride >= 1; stride >>= 1) {
		if ((int)threadIdx.x < stride) s_best[threadIdx.x] = better(s_best[threadIdx.x], s_best[threadIdx.x + stride], mode);
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		MscReduceOut r;
		r.best_sim = s_best[0].pos >= 0 ? s_best[0].sim : (mode == MSC_REDUCE_GET_CLOSE ? -1.0 : 2.2250738585072014e-308);
		r.best_pos = s_best[0].pos >= 0 ? s_best[0].pos : (mode == MSC_REDUCE_GET_CLOSE ? -1 : 0);
		r.any_close = s_nclose > 0;
		r.n_close = s_nclose;
		r.first_error = s_err;
		*out = r;
		if (cl.pos) cl.out[0] = s_best[0].pos >= 0 ? (uint32_t)s_best[0].pos + 1u : 0u;          // (a window pass: the parts' pos is the window position)
	}
}

// distance_d for the m members against the rounded mean (clutil/DivergencePoint.cpp:55-66):
//   dist = sum 2*min(p, (T)round(c))  =  sum p + sum r - manh(p, r)
//   mag  = sum_i floor-accumulated (p_i + c_i) = sum p + sum floor(c_i)      (uint64 += double truncates every step)
__global__ void __launch_bounds__(1024) k_distance_d(const MscPartial* __restrict__ partials, uint32_t S, uint32_t m,
                                                     const uint8_t* __restrict__ scalars, uint64_t scalar_stride,
                                                     const uint32_t* __restrict__ member_slots, const uint8_t* __restrict__ r_scalars,
                                                     const uint64_t* __restrict__ floor_sum, double* __restrict__ dist_out,
                                                     MscReduceOut* __restrict__ out) {
	__shared__ Best s_best[1024];
	const MscSlotScalars* rs = reinterpret_cast<const MscSlotScalars*>(r_scalars);
	Best b{0.0, -1};
	for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
		uint64_t manh = 0;
		for (uint32_t s = 0; s < S; s++) manh += partials[(uint64_t)i * S + s].manh;
		const uint32_t slot = member_slots ? member_slots[i] : i;
		const MscSlotScalars* ps = reinterpret_cast<const MscSlotScalars*>(scalars + (uint64_t)slot * scalar_stride);
		const uint64_t dist = ps->sum + rs->sum - manh;
		const uint64_t mag = ps->sum + *floor_sum;
		const double frac = (double)dist / (double)mag;
		const double d = 10000.0 * (1.0 - frac * frac);
		if (dist_out) dist_out[i] = d;
		// first minimum wins (cluster/Trainer.cpp:150-153, ClusterFactory.cpp:369-373 serial order)
		if (b.pos < 0 || d < b.sim) b = Best{d, (int64_t)i};
	}
	s_best[threadIdx.x] = b;
	__syncthreads();
	for (int stride = 512; stride >= 1; stride >>= 1) {
		if ((int)threadIdx.x < stride) {
			Best x = s_best[threadIdx.x], y = s_best[threadIdx.x + stride];
			if (y.pos >= 0 && (x.pos < 0 || y.sim < x.sim || (y.sim == x.sim && y.pos < x.pos))) s_best[threadIdx.x] = y;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		MscReduceOut r;
		r.best_sim = s_best[0].sim;
		r.best_pos = s_best[0].pos;
		r.any_close = 0;
		r.n_close = 0;
		r.first_error = 0;
		*out = r;
	}
}

// the same, shared out: one wave per member (its S records are consecutive), then the first minimum over the distances
__global__ void __launch_bounds__(kBlock) k_distance_members(const MscPartial* __restrict__ partials, uint32_t S, uint32_t m, const uint8_t* __restrict__ scalars,
                                                             uint64_t scalar_stride, const uint32_t* __restrict__ member_slots, const uint8_t* __restrict__ r_scalars,
                                                             const uint64_t* __restrict__ floor_sum, double* __restrict__ dist_out) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	if (i >= m) return;
	uint64_t manh = 0;
	for (uint32_t s = lane; s < S; s += 64) manh += partials[(uint64_t)i * S + s].manh;
	manh = shfl_sum_u64(manh);
	if (lane == 0) {
		const MscSlotScalars* rs = reinterpret_cast<const MscSlotScalars*>(r_scalars);
		const uint32_t slot = member_slots ? member_slots[i] : i;
		const MscSlotScalars* ps = reinterpret_cast<const MscSlotScalars*>(scalars + (uint64_t)slot * scalar_stride);
		const uint64_t dist = ps->sum + rs->sum - manh;
		const uint64_t mag = ps->sum + *floor_sum;
		const double frac = (double)dist / (double)mag;
		dist_out[i] = 10000.0 * (1.0 - frac * frac);
	}
}
__global__ void __launch_bounds__(1024) k_first_minimum(const double* __restrict__ dist, uint32_t m, MscReduceOut* __restrict__ out) {
	__shared__ Best s_best[1024];
	Best b{0.0, -1};
	for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
		const double d = dist[i];
		if (b.pos < 0 || d < b.sim) b = Best{d, (int64_t)i};          // first minimum wins (cluster/Trainer.cpp:150-153)
	}
	s_best[threadIdx.x] = b;
	__syncthreads();
	for (int stride = 512; stride >= 1; stride >>= 1) {
		if ((int)threadIdx.x < stride) {
			Best x = s_best[threadIdx.x], y = s_best[threadIdx.x + stride];
			if (y.pos >= 0 && (x.pos < 0 || y.sim < x.sim || (y.sim == x.sim && y.pos < x.pos))) s_best[threadIdx.x] = y;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		MscReduceOut r;
		r.best_sim = s_best[0].sim;
		r.best_pos = s_best[0].pos;
		r.any_close = 0;
		r.n_close = 0;
		r.first_error = 0;
		*out = r;
	}
}

// ---------------------------------------------------------------------------------------- column sums (mean)
// get_mean / mean_shift_update mean (cluster/ClusterFactory.cpp:338-357,297-326): mean_i = (sum_members p_i) / m in FP64.
// Sums of integers are exact below 2^53, so integer column sums + one division reproduce the reference's
// sequential double accumulation bit for bit. Emits r = (T)round(mean) in the same tile-permuted layout (it is
// scored against the members by k_pair_tiles), the FP64 mean, and sum floor(mean_i) over the REAL bins.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_colsum(const T* __restrict__ bins, uint64_t slot_elems, const uint32_t* __restrict__ member_slots,
                                                   uint32_t m, uint64_t padded, uint64_t nbins, uint32_t R, T* __restrict__ rounded,
                                                   double* __restrict__ mean_out, unsigned long long* __restrict__ floor_sum) {
	constexpr uint32_t E = 16 / sizeof(T);
	const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long fl = 0;
	if (chunk * E < padded) {
		uint64_t acc[E];
#pragma unroll
		for (uint32_t j = 0; j < E; j++) acc[j] = 0;
		// eight members' loads in flight per thread: with one 16-byte column per thread a 1 MiB histogram is only 1 024 waves, so the
		// walk over the members is latency-bound unless every wave keeps several loads outstanding (r02 profile: 2.15 TB/s with one)
		constexpr uint32_t U = 8;
		uint32_t i = 0;
		for (; i + U <= m; i += U) {
			u32x4 v[U];
#pragma unroll
			for (uint32_t u = 0; u < U; u++) {
				const uint32_t slot = member_slots ? member_slots[i + u] : i + u;
				v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(bins + (uint64_t)slot * slot_elems + chunk * E));
			}
#pragma unroll
			for (uint32_t u = 0; u < U; u++) {
				const T* e = reinterpret_cast<const T*>(&v[u]);
#pragma unroll
				for (uint32_t j = 0; j < E; j++) acc[j] += e[j];
			}
		}
		for (; i < m; i++) {
			const uint32_t slot = member_slots ? member_slots[i] : i;
			const uint4 v = *reinterpret_cast<const uint4*>(bins + (uint64_t)slot * slot_elems + chunk * E);
			const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
			for (uint32_t j = 0; j < E; j++) acc[j] += e[j];
		}
		// logical index of element j of this chunk (to exclude the zero pad bins of tiny histograms)
		const uint32_t tile_bins = 64 * R;
		const uint64_t tile = (chunk * E) / tile_bins;
		const uint32_t in_tile = (uint32_t)((chunk * E) % tile_bins);
		const uint32_t t = in_tile / (64 * E), lane = (in_tile % (64 * E)) / E;
		T rv[E];
#pragma unroll
		for (uint32_t j = 0; j < E; j++) {
			const uint64_t logical = tile * tile_bins + (uint64_t)lane * R + t * E + j;
			const double mean = (double)acc[j] / (double)m;
			if (mean_out) mean_out[chunk * E + j] = mean;
			rv[j] = (T)round(mean);
			if (logical < nbins) fl += (unsigned long long)floor(mean);
		}
		*reinterpret_cast<uint4*>(rounded + chunk * E) = *reinterpret_cast<const uint4*>(rv);
	}
	fl = shfl_sum_u64(fl);
	if ((threadIdx.x & 63) == 0 && fl) atomicAdd(floor_sum, fl);
}

// k_colsum for many member lists at once: `cpb` consecutive workgroups serve segment i (members = member_slots[first .. first+m)),
// whose rounded mean goes to slot i of `rounded` and whose floor sum to floor_sum[i] (zeroed by the caller).
template <typename T>
__global__ void __launch_bounds__(kBlock) k_colsum_batch(const T* __restrict__ bins, uint64_t slot_elems, const uint32_t* __restrict__ member_slots,
                                                         const MscBatchSeg* __restrict__ segs, uint32_t cpb, uint64_t padded, uint64_t nbins, uint32_t R,
                                                         T* __restrict__ rounded, unsigned long long* __restrict__ floor_sum) {
	constexpr uint32_t E = 16 / sizeof(T);
	const uint32_t si = blockIdx.x / cpb;
	const MscBatchSeg seg = segs[si];
	if (seg.m == 0) return;
	const uint64_t chunk = (uint64_t)(blockIdx.x % cpb) * blockDim.x + threadIdx.x;
	unsigned long long fl = 0;
	if (chunk * E < padded) {
		uint64_t acc[E];
#pragma unroll
		for (uint32_t j = 0; j < E; j++) acc[j] = 0;
		constexpr uint32_t U = 8;          // eight members' loads in flight per thread (see k_colsum)
		uint32_t i = 0;
		for (; i + U <= seg.m; i += U) {
			u32x4 v[U];
#pragma unroll
			for (uint32_t u = 0; u < U; u++)
				v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(bins + (uint64_t)member_slots[seg.first + i + u] * slot_elems + chunk * E));
#pragma unroll
			for (uint32_t u = 0; u < U; u++) {
				const T* e = reinterpret_cast<const T*>(&v[u]);
#pragma unroll
				for (uint32_t j = 0; j < E; j++) acc[j] += e[j];
			}
		}
		for (; i < seg.m; i++) {
			const uint32_t slot = member_slots[seg.first + i];
			const uint4 v = *reinterpret_cast<const uint4*>(bins + (uint64_t)slot * slot_elems + chunk * E);
			const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
			for (uint32_t j = 0; j < E; j++) acc[j] += e[j];
		}
		const uint32_t tile_bins = 64 * R;
		const uint64_t tile = (chunk * E) / tile_bins;
		const uint32_t in_tile = (uint32_t)((chunk * E) % tile_bins);
		const uint32_t t = in_tile / (64 * E), lane = (in_tile % (64 * E)) / E;
		T rv[E];
#pragma unroll
		for (uint32_t j = 0; j < E; j++) {
			const uint64_t logical = tile * tile_bins + (uint64_t)lane * R + t * E + j;
			const double mean = (double)acc[j] / (double)seg.m;
			rv[j] = (T)round(mean);
			if (logical < nbins) fl += (unsigned long long)floor(mean);
		}
		*reinterpret_cast<uint4*>(rounded + (uint64_t)si * slot_elems + chunk * E) = *reinterpret_cast<const uint4*>(rv);
	}
	fl = shfl_sum_u64(fl);
	if ((threadIdx.x & 63) == 0 && fl) atomicAdd(floor_sum + si, fl);
}

// The two halves of k_colsum_batch for a mean whose members are spread over several GPUs ("all-reduce (sum) of partial
// column sums"): k_colsum_sums leaves the exact integer column sums of each segment's members in sums[si][padded] (what the ranks
// add up), k_mean_from_sums turns the added-up sums of m_total[si] members into the rounded mean slot and its floor sum -- the same
// division, rounding and floor as k_colsum_batch, so a sharded get_mean is the single-GPU one bit for bit.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_colsum_sums(const T* __restrict__ bins, uint64_t slot_elems, const uint32_t* __restrict__ member_slots,
                                                        const MscBatchSeg* __restrict__ segs, uint32_t cpb, uint64_t padded, unsigned long long* __restrict__ sums) {
	constexpr uint32_t E = 16 / sizeof(T);
	const uint32_t si = blockIdx.x / cpb;
	const MscBatchSeg seg = segs[si];
	const uint64_t chunk = (uint64_t)(blockIdx.x % cpb) * blockDim.x + threadIdx.x;
	if (chunk * E >= padded) return;
	uint64_t acc[E];
#pragma unroll
	for (uint32_t j = 0; j < E; j++) acc[j] = 0;
	constexpr uint32_t U = 8;          // eight members' loads in flight per thread (see k_colsum)
	uint32_t i = 0;
	for (; i + U <= seg.m; i += U) {
		u32x4 v[U];
#pragma unroll
		for (uint32_t u = 0; u < U; u++)
			v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(bins + (uint64_t)member_slots[seg.first + i + u] * slot_elems + chunk * E));
#pragma unroll
		for (uint32_t u = 0; u < U; u++) {
			const T* e = reinterpret_cast<const T*>(&v[u]);
#pragma unroll
			for (uint32_t j = 0; j < E; j++) acc[j] += e[j];
		}
	}
	for (; i < seg.m; i++) {
		const uint4 v = *reinterpret_cast<const uint4*>(bins + (uint64_t)member_slots[seg.first + i] * slot_elems + chunk * E);
		const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
		for (uint32_t j = 0; j < E; j++) acc[j] += e[j];
	}
#pragma unroll
	for (uint32_t j = 0; j < E; j++) sums[(uint64_t)si * padded + chunk * E + j] = acc[j];
}

template <typename T>
__global__ void __launch_bounds__(kBlock) k_mean_from_sums(const unsigned long long* __restrict__ sums, const unsigned long long* __restrict__ m_total, uint32_t cpb,
                                                           uint64_t padded, uint64_t nbins, uint32_t R, uint64_t slot_elems, T* __restrict__ rounded,
                                                           unsigned long long* __restrict__ floor_sum) {
	constexpr uint32_t E = 16 / sizeof(T);
	const uint32_t si = blockIdx.x / cpb;
	const unsigned long long m = m_total[si];
	if (m == 0) return;
	const uint64_t chunk = (uint64_t)(blockIdx.x % cpb) * blockDim.x + threadIdx.x;
	unsigned long long fl = 0;
	if (chunk * E < padded) {
		const uint32_t tile_bins = 64 * R;
		const uint64_t tile = (chunk * E) / tile_bins;
		const uint32_t in_tile = (uint32_t)((chunk * E) % tile_bins);
		const uint32_t t = in_tile / (64 * E), lane = (in_tile % (64 * E)) / E;
		T rv[E];
#pragma unroll
		for (uint32_t j = 0; j < E; j++) {
			const uint64_t logical = tile * tile_bins + (uint64_t)lane * R + t * E + j;
			const double mean = (double)sums[(uint64_t)si * padded + chunk * E + j] / (double)m;
			rv[j] = (T)round(mean);
			if (logical < nbins) fl += (unsigned long long)floor(mean);
		}
		*reinterpret_cast<uint4*>(rounded + (uint64_t)si * slot_elems + chunk * E) = *reinterpret_cast<const uint4*>(rv);
	}
	fl = shfl_sum_u64(fl);
	if ((threadIdx.x & 63) == 0 && fl) atomicAdd(floor_sum + si, fl);
}

// distance_d of every member to the rounded mean of ITS segment (see k_distance_d); the per-segment arg-min is left to the host
__global__ void __launch_bounds__(kBlock) k_distance_batch(const MscPartial* __restrict__ partials, uint32_t S, uint32_t n, const uint8_t* __restrict__ scalars,
                                                           uint64_t scalar_stride, const uint32_t* __restrict__ member_slots, const uint32_t* __restrict__ pair_seg,
                                                           const uint8_t* __restrict__ r_scalars, uint64_t r_stride, const uint64_t* __restrict__ floor_sum,
                                                           double* __restrict__ dist_out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint64_t manh = 0;
	for (uint32_t s = 0; s < S; s++) manh += partials[(uint64_t)i * S + s].manh;
	const uint32_t sg = pair_seg[i];
	const MscSlotScalars* ps = reinterpret_cast<const MscSlotScalars*>(scalars + (uint64_t)member_slots[i] * scalar_stride);
	const MscSlotScalars* rs = reinterpret_cast<const MscSlotScalars*>(r_scalars + (uint64_t)sg * r_stride);
	const uint64_t dist = ps->sum + rs->sum - manh;
	const uint64_t mag = ps->sum + floor_sum[sg];
	const double frac = (double)dist / (double)mag;
	dist_out[i] = 10000.0 * (1.0 - frac * frac);
}

}  // namespace

// ======================================================================================== launchers
template <typename T, int LPT, int TB>
static hipError_t launch_tiles_t(hipStream_t st, const MscLayout& L, const uint8_t* cand_bins, const uint8_t* cand_scalars,
                                 const uint32_t* cand_slots, uint32_t m, const uint8_t* q_bins, const uint8_t* q_scalars,
                                 int use_window, uint64_t min_len, uint64_t max_len, MscPartial* partials, int num_cus,
                                 void* div_tables, void* div_partials, int order) {
	const uint32_t S = L.S;
	// enough waves to fill the chip (8 per SIMD), at least one candidate group, at most one group per candidate
	const uint64_t target_waves = (uint64_t)num_cus * 32;
	uint64_t G = (target_waves + S - 1) / S;
	if (G < 1) G = 1;
	if (G > m) G = m;
	const uint64_t waves = (uint64_t)S * G;
	const unsigned blocks = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
	const bool padded = L.nbins < L.padded_bins;
	const uint64_t stride = msc_scalar_stride(S);
	if constexpr (TB > 0) {
		k_div_tables<TB><<<dim3(m), dim3(TB * TB), 0, st>>>(cand_scalars, stride, cand_slots, m, q_scalars, order, (DivTerm*)div_tables);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	if (padded) {
		k_pair_tiles<T, LPT, true, TB><<<dim3(blocks), dim3(kBlock), 0, st>>>(cand_bins, L.slot_bytes, cand_scalars, stride, cand_slots, m, q_bins,
		    q_scalars, S, (uint32_t)G, (uint32_t)L.nbins, use_window, min_len, max_len, partials, (const DivTerm*)div_tables, (MscPartialDiv*)div_partials, order);
	} else {
		k_pair_tiles<T, LPT, false, TB><<<dim3(blocks), dim3(kBlock), 0, st>>>(cand_bins, L.slot_bytes, cand_scalars, stride, cand_slots, m, q_bins,
		    q_scalars, S, (uint32_t)G, (uint32_t)L.tile_bins, use_window, min_len, max_len, partials, (const DivTerm*)div_tables, (MscPartialDiv*)div_partials, order);
	}
	return hipGetLastError();
}

template <typename T>
static hipError_t launch_tiles_lpt(hipStream_t st, const MscLayout& L, const uint8_t* cb, const uint8_t* cs, const uint32_t* sl, uint32_t m,
                                   const uint8_t* qb, const uint8_t* qs, int uw, uint64_t mn, uint64_t mx, MscPartial* p, int cus,
                                   int tb, void* dt, void* dp, int order) {
	if (tb == 0) {
		switch (L.LPT) {
		case 1: return launch_tiles_t<T, 1, 0>(st, L, cb, cs, sl, m, qb, qs, uw, mn, mx, p, cus, dt, dp, order);
		case 2: return launch_tiles_t<T, 2, 0>(st, L, cb, cs, sl, m, qb, qs, uw, mn, mx, p, cus, dt, dp, order);
		default: return launch_tiles_t<T, 4, 0>(st, L, cb, cs, sl, m, qb, qs, uw, mn, mx, p, cus, dt, dp, order);
		}
	}
	if (tb == 8) return launch_tiles_t<T, 4, 8>(st, L, cb, cs, sl, m, qb, qs, uw, mn, mx, p, cus, dt, dp, order);
	switch (L.LPT) {
	case 1: return launch_tiles_t<T, 1, 16>(st, L, cb, cs, sl, m, qb, qs, uw, mn, mx, p, cus, dt, dp, order);
	case 2: return launch_tiles_t<T, 2, 16>(st, L, cb, cs, sl, m, qb, qs, uw, mn, mx, p, cus, dt, dp, order);
	default: return launch_tiles_t<T, 4, 16>(st, L, cb, cs, sl, m, qb, qs, uw, mn, mx, p, cus, dt, dp, order);
	}
}

template <typename T>
static hipError_t launch_wide_t(hipStream_t st, const MscLayout& L, const uint8_t* cb, const uint8_t* cs, const uint32_t* sl, uint32_t m,
                                const uint8_t* qb, const uint8_t* qs, int uw, uint64_t mn, uint64_t mx, MscPartial* p, int cus, void* dp, int order) {
	const uint32_t S = L.S;
	uint64_t G = ((uint64_t)cus * 32 + S - 1) / S;
	if (G < 1) G = 1;
	if (G > m) G = m;
	const unsigned blocks = (unsigned)(((uint64_t)S * G + kWavesPerBlock - 1) / kWavesPerBlock);
	const uint64_t stride = msc_scalar_stride(S);
	const uint32_t nvalid = L.nbins < L.padded_bins ? (uint32_t)L.nbins : L.tile_bins;
#define MSC_WIDE(LPTV)                                                                                                                         \
	(dp ? k_pair_tiles_wide<T, LPTV, true><<<dim3(blocks), dim3(kBlock), 0, st>>>(cb, L.slot_bytes, cs, stride, sl, m, qb, qs, S, (uint32_t)G,   \
	                                                                              nvalid, uw, mn, mx, p, (MscPartialDiv*)dp, order)          \
	    : k_pair_tiles_wide<T, LPTV, false><<<dim3(blocks), dim3(kBlock), 0, st>>>(cb, L.slot_bytes, cs, stride, sl, m, qb, qs, S, (uint32_t)G,  \
	                                                                               nvalid, uw, mn, mx, p, (MscPartialDiv*)dp, order))
	switch (L.LPT) {
	case 1: MSC_WIDE(1); break;
	case 2: MSC_WIDE(2); break;
	default: MSC_WIDE(4); break;
	}
#undef MSC_WIDE
	return hipGetLastError();
}

hipError_t msc_launch_pair_tiles_wide(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* cand_bins, const uint8_t* cand_scalars,
                                      const uint32_t* cand_slots, uint32_t m, const uint8_t* q_bins_slot, const uint8_t* q_scalars_slot,
                                      int use_window, uint64_t min_len, uint64_t max_len, MscPartial* partials, int num_cus,
                                      void* div_partials, int order) {
	if (m == 0) return hipSuccess;
	switch (dtype) {
	case 8: return launch_wide_t<uint8_t>(st, L, cand_bins, cand_scalars, cand_slots, m, q_bins_slot, q_scalars_slot, use_window, min_len, max_len, partials, num_cus, div_partials, order);
	case 16: return launch_wide_t<uint16_t>(st, L, cand_bins, cand_scalars, cand_slots, m, q_bins_slot, q_scalars_slot, use_window, min_len, max_len, partials, num_cus, div_partials, order);
	case 32: return launch_wide_t<uint32_t>(st, L, cand_bins, cand_scalars, cand_slots, m, q_bins_slot, q_scalars_slot, use_window, min_len, max_len, partials, num_cus, div_partials, order);
	default: return launch_wide_t<uint64_t>(st, L, cand_bins, cand_scalars, cand_slots, m, q_bins_slot, q_scalars_slot, use_window, min_len, max_len, partials, num_cus, div_partials, order);
	}
}

int msc_div_table_dim(const MscLayout& L) { return (L.LPT == 4 && L.S >= 4) ? 8 : 16; }

hipError_t msc_launch_pair_tiles(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* cand_bins, const uint8_t* cand_scalars,
                                 const uint32_t* cand_slots, uint32_t m, const uint8_t* q_bins_slot, const uint8_t* q_scalars_slot,
                                 int use_window, uint64_t min_len, uint64_t max_len, MscPartial* partials, int num_cus,
                                 void* div_tables, void* div_partials, int order) {
	if (m == 0) return hipSuccess;
	const int tb = div_tables ? msc_div_table_dim(L) : 0;
	switch (dtype) {
	case 8: return launch_tiles_lpt<uint8_t>(st, L, cand_bins, cand_scalars, cand_slots, m, q_bins_slot, q_scalars_slot, use_window, min_len, max_len, partials, num_cus, tb, div_tables, div_partials, order);
	case 16: return launch_tiles_lpt<uint16_t>(st, L, cand_bins, cand_scalars, cand_slots, m, q_bins_slot, q_scalars_slot, use_window, min_len, max_len, partials, num_cus, tb, div_tables, div_partials, order);
	case 32: return launch_tiles_lpt<uint32_t>(st, L, cand_bins, cand_scalars, cand_slots, m, q_bins_slot, q_scalars_slot, use_window, min_len, max_len, partials, num_cus, tb, div_tables, div_partials, order);
	default: return launch_tiles_lpt<uint64_t>(st, L, cand_bins, cand_scalars, cand_slots, m, q_bins_slot, q_scalars_slot, use_window, min_len, max_len, partials, num_cus, tb, div_tables, div_partials, order);
	}
}

template <typename T, int TQ, bool COMPACT>
static hipError_t launch_multi_t(hipStream_t st, const MscLayout& L, const uint8_t* cb, const uint8_t* cs, const uint32_t* sl, uint32_t m,
                                 const uint8_t* qb, uint64_t qsb, const uint8_t* qs, uint64_t qss, const uint32_t* qslots, uint32_t nq,
                                 MscPartial* partials, int num_cus) {
	const uint32_t S = L.S;
	const uint32_t nqb = (nq + TQ - 1) / TQ;
	// Grid policy. "round" (default): exactly ONE resident round of equal-length waves, sized from the kernel's real
	// occupancy, so no partially filled tail round exists. "fine": ~8 rounds of short waves (>= 32 candidates each).
	int blocks_per_cu = 0;
	const void* fn = nullptr;
	if constexpr (sizeof(T) >= 4) { if (L.LPT == 4) fn = (const void*)k_pair_tiles_multi32<T, 4, TQ, COMPACT>; }
	if (!fn) {
		if constexpr (TQ <= 4) {
			fn = L.LPT == 1 ? (const void*)k_pair_tiles_multi<T, 1, TQ, COMPACT> : L.LPT == 2 ? (const void*)k_pair_tiles_multi<T, 2, TQ, COMPACT>
			                                                                                   : (const void*)k_pair_tiles_multi<T, 4, TQ, COMPACT>;
		}
	}
	if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, fn, kBlock, 0) != hipSuccess || blocks_per_cu < 1) blocks_per_cu = 2;
	const uint64_t target_waves = (uint64_t)num_cus * blocks_per_cu * kWavesPerBlock;
	uint64_t G = target_waves / ((uint64_t)S * nqb);
	if (G < 1) G = 1;
	if (G > m) G = m;
	const uint64_t waves = (uint64_t)S * G * nqb;
	const unsigned blocks = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
	const uint64_t stride = msc_scalar_stride(S);
	if constexpr (sizeof(T) >= 4) {
		if (L.LPT == 4) {
			k_pair_tiles_multi32<T, 4, TQ, COMPACT><<<dim3(blocks), dim3(kBlock), 0, st>>>(cb, L.slot_bytes, cs, stride, sl, m, qb, qsb, qs, qss, qslots, nq, S, (uint32_t)G, partials);
			return hipGetLastError();
		}
	}
	if constexpr (TQ > 4) return hipErrorInvalidValue;       // the generic form keeps raw query tiles in registers: TQ <= 4
	else
	switch (L.LPT) {
	case 1: k_pair_tiles_multi<T, 1, TQ, COMPACT><<<dim3(blocks), dim3(kBlock), 0, st>>>(cb, L.slot_bytes, cs, stride, sl, m, qb, qsb, qs, qss, qslots, nq, S, (uint32_t)G, partials); break;
	case 2: k_pair_tiles_multi<T, 2, TQ, COMPACT><<<dim3(blocks), dim3(kBlock), 0, st>>>(cb, L.slot_bytes, cs, stride, sl, m, qb, qsb, qs, qss, qslots, nq, S, (uint32_t)G, partials); break;
	default: k_pair_tiles_multi<T, 4, TQ, COMPACT><<<dim3(blocks), dim3(kBlock), 0, st>>>(cb, L.slot_bytes, cs, stride, sl, m, qb, qsb, qs, qss, qslots, nq, S, (uint32_t)G, partials); break;
	}
	return hipGetLastError();
}

hipError_t msc_launch_pair_tiles_multi(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* cand_bins, const uint8_t* cand_scalars,
                                       const uint32_t* cand_slots, uint32_t m, const uint8_t* qset_bins, uint64_t q_slot_bytes,
                                       const uint8_t* qset_scalars, uint64_t q_scalar_stride, const uint32_t* q_slots, uint32_t n_q,
                                       int tq, bool compact, MscPartial* partials, int num_cus) {
	if (m == 0 || n_q == 0) return hipSuccess;
#define MSC_MULTI_ARGS st, L, cand_bins, cand_scalars, cand_slots, m, qset_bins, q_slot_bytes, qset_scalars, q_scalar_stride, q_slots, n_q, partials, num_cus
#define MSC_MULTI(T)                                                                                          \
	((tq >= 8 && sizeof(T) >= 4 && L.LPT == 4)                                                                      \
	     ? (compact ? launch_multi_t<T, 8, true>(MSC_MULTI_ARGS) : launch_multi_t<T, 8, false>(MSC_MULTI_ARGS))  \
	 : tq >= 4 ? (compact ? launch_multi_t<T, 4, true>(MSC_MULTI_ARGS) : launch_multi_t<T, 4, false>(MSC_MULTI_ARGS)) \
	           : (compact ? launch_multi_t<T, 2, true>(MSC_MULTI_ARGS) : launch_multi_t<T, 2, false>(MSC_MULTI_ARGS)))
	switch (dtype) {
	case 8: return MSC_MULTI(uint8_t);
	case 16: return MSC_MULTI(uint16_t);
	case 32: return MSC_MULTI(uint32_t);
	default: return MSC_MULTI(uint64_t);
	}
#undef MSC_MULTI
#undef MSC_MULTI_ARGS
}

template <typename T, int TQ, int NB, bool P16>
static hipError_t launch_multi_ring_t(hipStream_t st, const MscLayout& L, const uint8_t* cb, const uint8_t* cs, const uint32_t* sl, uint32_t m,
                                      const uint8_t* qb, uint64_t qsb, const uint8_t* qs, uint64_t qss, const uint32_t* qslots, uint32_t nq,
                                      void* partials16, int num_cus) {
	const uint32_t S = L.S;
	const uint32_t nqb = (nq + TQ - 1) / TQ;
	const size_t lds = (size_t)kWavesPerBlock * NB * kRingSlotBytes;
	const void* fn = (const void*)k_pair_tiles_multi32_ring<T, TQ, NB, P16>;
	static bool attr_done = false;
	if (!attr_done) {
		hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return e;
		attr_done = true;
	}
	int blocks_per_cu = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, fn, kBlock, lds) != hipSuccess || blocks_per_cu < 1) blocks_per_cu = 1;
	// one resident round of equal-length waves (same policy as launch_multi_t)
	uint64_t G = (uint64_t)num_cus * blocks_per_cu * kWavesPerBlock / ((uint64_t)S * nqb);
	if (G < 1) G = 1;
	if (G > m) G = m;
	const uint64_t waves = (uint64_t)S * G * nqb;
	const unsigned blocks = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
	k_pair_tiles_multi32_ring<T, TQ, NB, P16><<<dim3(blocks), dim3(kBlock), lds, st>>>(cb, L.slot_bytes, cs, msc_scalar_stride(S), sl, m, qb, qsb, qs, qss, qslots, nq, S,
	                                                                               (uint32_t)G, nqb, (u32x4*)partials16);
	return hipGetLastError();
}

hipError_t msc_launch_pair_tiles_multi_ring(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* cand_bins, const uint8_t* cand_scalars,
                                            const uint32_t* cand_slots, uint32_t m, const uint8_t* qset_bins, uint64_t q_slot_bytes,
                                            const uint8_t* qset_scalars, uint64_t q_scalar_stride, const uint32_t* q_slots, uint32_t n_q, int tq,
                                            bool prefix16, void* partials16, int num_cus) {
	if (m == 0 || n_q == 0) return hipSuccess;
	if (L.LPT != 4 || (dtype != 32 && dtype != 64) || (tq != 4 && tq != 8)) return hipErrorInvalidValue;
	static const int nb_env = [] { const char* e = getenv("MSC_RING_SLOTS"); return e ? atoi(e) : 0; }();
#define MSC_RING_ARGS st, L, cand_bins, cand_scalars, cand_slots, m, qset_bins, q_slot_bytes, qset_scalars, q_scalar_stride, q_slots, n_q, partials16, num_cus
#define MSC_RING_NB(T, Q, P)                                                                                     \
	(nb_env == 2 ? launch_multi_ring_t<T, Q, 2, P>(MSC_RING_ARGS) : nb_env == 3 ? launch_multi_ring_t<T, Q, 3, P>(MSC_RING_ARGS) \
	                                                                          : launch_multi_ring_t<T, Q, 4, P>(MSC_RING_ARGS))
#define MSC_RING(T)                                                                                              \
	(tq == 8 ? (prefix16 ? MSC_RING_NB(T, 8, true) : MSC_RING_NB(T, 8, false)) : (prefix16 ? MSC_RING_NB(T, 4, true) : MSC_RING_NB(T, 4, false)))
	return dtype == 32 ? MSC_RING(uint32_t) : MSC_RING(uint64_t);
#undef MSC_RING_NB
#undef MSC_RING
#undef MSC_RING_ARGS
}

// counts[q] += number of close flags in row q of flags[n_q][m] (one workgroup per 4 096 flags of a row)
__global__ void __launch_bounds__(256) k_close_counts(const uint8_t* __restrict__ flags, uint32_t m, unsigned long long* __restrict__ counts) {
	const uint32_t q = blockIdx.y;
	const uint8_t* row = flags + (uint64_t)q * m;
	uint32_t n = 0;
	for (uint32_t i = blockIdx.x * 4096 + threadIdx.x; i < m && i < (blockIdx.x + 1) * 4096; i += 256) n += row[i] != 0;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off, 64);
	if ((threadIdx.x & 63) == 0 && n) atomicAdd(counts + q, (unsigned long long)n);
}
hipError_t msc_launch_close_counts(hipStream_t st, const uint8_t* flags, uint32_t n_q, uint32_t m, uint64_t* counts) {
	if (!n_q || !m) return hipSuccess;
	k_close_counts<<<dim3((m + 4095) / 4096, n_q), dim3(256), 0, st>>>(flags, m, (unsigned long long*)counts);
	return hipGetLastError();
}

hipError_t msc_launch_epilogue(hipStream_t st, const MscEpilogueArgs& a) {
	if (a.m == 0) return hipSuccess;
	if (a.kb_min) {
		if (a.n_queries < 2 || a.n_queries > a.kb_qn || !a.kb_c_mb || !a.kb_q_mb || !a.kb_qT) return hipErrorInvalidValue;      // (epilogue_one's query-major index needs n_queries > 1)
		const uint64_t waves = (uint64_t)a.m_per_query * ((a.n_queries + 63) / 64);
		const bool screen = a.screen && a.model && a.close_soa && !a.sum_soa && !a.csum_soa && !a.raw_out && !a.singles_out && !a.combos_out && !a.pair_out && !a.use_window &&
		                    !a.div_direct && !a.grp_pairs && !a.sparse_base;
		if (screen) {
			constexpr uint32_t chunk = 4;          // candidates per wave: 1 / 4 / 8 / 16 took 265 / 274 / 609 / 678 us per block before the screen was slimmed; 4 keeps the queries' side per wave
			const uint64_t cw = (uint64_t)((a.m_per_query + chunk - 1) / chunk) * ((a.n_queries + 63) / 64);
			hipLaunchKernelGGL(k_pair_epilogue_bits_screen<chunk>, dim3((unsigned)((cw + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, st, a);
			const uint64_t threads = ((uint64_t)a.n_queries * a.m_per_query + 15) / 16;
			hipLaunchKernelGGL(k_pair_epilogue_bits_open, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a);
		} else hipLaunchKernelGGL(k_pair_epilogue_bits, dim3((unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, st, a);
	} else if (a.partials_cq) {
		if (a.cq_group != 16 && (a.cq_group != 32 || !a.dot_gemm)) return hipErrorInvalidValue;      // (manh-only records: the products must come from the GEMM)
		const unsigned waves = a.m_per_query * ((a.n_queries + a.cq_group - 1) / a.cq_group);
		hipLaunchKernelGGL(k_pair_epilogue_cq, dim3((waves + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, st, a);
	} else if (a.S > 4) {
		const unsigned blocks = (a.m + kWavesPerBlock - 1) / kWavesPerBlock;
		hipLaunchKernelGGL(k_pair_epilogue_wave, dim3(blocks), dim3(kBlock), 0, st, a);
	} else {
		const unsigned blocks = (a.m + kBlock - 1) / kBlock;
		hipLaunchKernelGGL(k_pair_epilogue_thread, dim3(blocks), dim3(kBlock), 0, st, a);
	}
	return hipGetLastError();
}

// epilogue + reduction (+ the window's close pass) of a 1 x M pass whose records are MscPartial (a.S of them per pair, <= 4) in two
// launches; parts_scratch: msc_reduce_scratch_bytes()
hipError_t msc_launch_epilogue_reduce(hipStream_t st, const MscEpilogueArgs& a, int mode, int64_t begin, uint8_t* flags_out, MscReduceOut* out, void* parts_scratch,
                                      const MscCloseList& cl, ReducePart* host_parts, uint32_t* n_parts_out) {
	if (a.m == 0 || a.S > 4 || a.partials16 || a.partials_cq || a.kb_min || !parts_scratch) return hipErrorInvalidValue;
	const uint32_t n_parts = std::min<uint32_t>(1024, (a.m + kBlock - 1) / kBlock);
	if (n_parts_out) *n_parts_out = n_parts;
	k_pair_epilogue_reduce_part<<<dim3(n_parts), dim3(kBlock), 0, st>>>(a, mode, begin, flags_out, host_parts ? host_parts : (ReducePart*)parts_scratch, cl);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess || host_parts) return e;
	k_pair_reduce_fold2<<<dim3(1), dim3(256), 0, st>>>((const ReducePart*)parts_scratch, n_parts, mode, out, cl);
	return hipGetLastError();
}

void msc_reduce_fold_host(const ReducePart* parts, uint32_t n_parts, int mode, MscReduceOut* out, uint32_t* wpos_out) {
	double sim = mode == MSC_REDUCE_GET_CLOSE ? -1.0 : 0.0;
	int64_t pos = -1;
	uint32_t wpos = 0;
	unsigned long long nclose = 0;
	int err = 0;
	for (uint32_t i = 0; i < n_parts; i++) {          // (better() of the device code, part by part)
		const ReducePart& p = parts[i];
		bool take;
		if (mode == MSC_REDUCE_GET_CLOSE) take = p.sim > sim || (p.sim == sim && p.pos >= 0 && (pos < 0 || p.pos < pos));
		else take = p.pos >= 0 && (pos < 0 || p.sim > sim || (p.sim == sim && p.pos > pos));
		if (take) { sim = p.sim; pos = p.pos; wpos = p.wpos; }
		nclose += p.nclose;
		if (p.err < err) err = p.err;
	}
	out->best_sim = pos >= 0 ? sim : (mode == MSC_REDUCE_GET_CLOSE ? -1.0 : 2.2250738585072014e-308);
	out->best_pos = pos >= 0 ? pos : (mode == MSC_REDUCE_GET_CLOSE ? -1 : 0);
	out->any_close = nclose > 0;
	out->n_close = nclose;
	out->first_error = err;
	if (wpos_out) *wpos_out = pos >= 0 ? wpos : 0u;
}

// parts_scratch (optional, msc_reduce_scratch_bytes()): long windows are folded by up to 256 workgroups first (1 024 in the fused form)
size_t msc_reduce_scratch_bytes() { return 1024 * sizeof(ReducePart); }
hipError_t msc_launch_reduce(hipStream_t st, const MscPairOut* pair_out, uint32_t m, int mode, int64_t begin, uint8_t* flags_out,
                             MscReduceOut* out, void* parts_scratch, const uint32_t* key) {
	if (parts_scratch && m > 8192) {
		const uint32_t n_parts = std::min<uint32_t>(256, (m + 4095) / 4096);
		const uint32_t per = (m + n_parts - 1) / n_parts;
		k_pair_reduce_part<<<dim3(n_parts), dim3(1024), 0, st>>>(pair_out, m, per, mode, begin, flags_out, (ReducePart*)parts_scratch, key);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
		k_pair_reduce_fold<<<dim3(1), dim3(256), 0, st>>>((const ReducePart*)parts_scratch, n_parts, mode, out);
		return hipGetLastError();
	}
	hipLaunchKernelGGL(k_pair_reduce, dim3(1), dim3(1024), 0, st, pair_out, m, mode, begin, flags_out, out, key);
	return hipGetLastError();
}

hipError_t msc_launch_colsum(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, const uint32_t* member_slots,
                             uint32_t m, void* rounded_out, double* mean_out, uint64_t* floor_sum_out, uint64_t* scratch) {
	(void)scratch;
	hipError_t e = hipMemsetAsync(floor_sum_out, 0, sizeof(uint64_t), st);
	if (e != hipSuccess) return e;
	const uint64_t chunks = L.padded_bins / L.E;
	const unsigned blocks = (unsigned)((chunks + kBlock - 1) / kBlock);
	switch (dtype) {
	case 8: hipLaunchKernelGGL(k_colsum<uint8_t>, dim3(blocks), dim3(kBlock), 0, st, (const uint8_t*)bins, L.padded_bins, member_slots, m, L.padded_bins, L.nbins, L.R, (uint8_t*)rounded_out, mean_out, (unsigned long long*)floor_sum_out); break;
	case 16: hipLaunchKernelGGL(k_colsum<uint16_t>, dim3(blocks), dim3(kBlock), 0, st, (const uint16_t*)bins, L.padded_bins, member_slots, m, L.padded_bins, L.nbins, L.R, (uint16_t*)rounded_out, mean_out, (unsigned long long*)floor_sum_out); break;
	case 32: hipLaunchKernelGGL(k_colsum<uint32_t>, dim3(blocks), dim3(kBlock), 0, st, (const uint32_t*)bins, L.padded_bins, member_slots, m, L.padded_bins, L.nbins, L.R, (uint32_t*)rounded_out, mean_out, (unsigned long long*)floor_sum_out); break;
	default: hipLaunchKernelGGL(k_colsum<uint64_t>, dim3(blocks), dim3(kBlock), 0, st, (const uint64_t*)bins, L.padded_bins, member_slots, m, L.padded_bins, L.nbins, L.R, (uint64_t*)rounded_out, mean_out, (unsigned long long*)floor_sum_out); break;
	}
	return hipGetLastError();
}

// dist_out != nullptr and more than a handful of records: one wave per member folds its S records (coalesced) and writes the
// distance, a single workgroup then takes the first minimum; otherwise the one-workgroup kernel does both (a thread per member
// walking its records took 150 us for 1 000 members of 256 tiles)
hipError_t msc_launch_distance_d(hipStream_t st, const MscPartial* partials, uint32_t S, uint32_t m, const uint8_t* scalars,
                                 uint64_t scalar_stride, const uint32_t* member_slots, const uint8_t* r_scalars,
                                 const uint64_t* floor_sum, double* dist_out, MscReduceOut* out) {
	if (dist_out && (uint64_t)m * S >= 4096) {
		k_distance_members<<<dim3((m + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st>>>(partials, S, m, scalars, scalar_stride, member_slots, r_scalars, floor_sum, dist_out);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
		k_first_minimum<<<dim3(1), dim3(1024), 0, st>>>(dist_out, m, out);
		return hipGetLastError();
	}
	hipLaunchKernelGGL(k_distance_d, dim3(1), dim3(1024), 0, st, partials, S, m, scalars, scalar_stride, member_slots, r_scalars,
	                   floor_sum, dist_out, out);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------- batched center->set(*next)
namespace {
// DivergencePoint::set for many (destination, source) slot pairs (clutil/DivergencePoint.cpp:182-190): bins, then length / derived
// sums / id / tile prefixes of the scalar record -- NOT mag, NOT stddev, NOT the 1-mers (msc_hist_assign, one pair at a time)
__global__ void __launch_bounds__(kBlock) k_assign_bins(uint4* __restrict__ dst, const uint4* __restrict__ src, const uint32_t* __restrict__ ds,
                                                       const uint32_t* __restrict__ ss, uint64_t per_slot, uint64_t total) {
	for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t pair = e / per_slot, off = e % per_slot;
		dst[(uint64_t)ds[pair] * per_slot + off] = src[(uint64_t)ss[pair] * per_slot + off];
	}
}
__global__ void __launch_bounds__(kBlock) k_assign_scalars(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, const uint32_t* __restrict__ ds,
                                                          const uint32_t* __restrict__ ss, uint32_t words_per_slot, uint32_t S, uint64_t total, int every_word) {
	for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t pair = e / words_per_slot;
		const uint32_t w = (uint32_t)(e % words_per_slot);
		// words of MscSlotScalars: 0 mag | 1 length 2 sum 3 sum_sq 4 max_count | 5-8 one_mers | 9 stddev 10 overflow | 11 id | 12.. ; 16.. tile prefixes
		// (every_word 1: an exact copy of the record, msc_hist_copy_batch; 2: DivergencePoint::clone, msc_hist_clone_batch -- an exact
		// copy whose magnitude is re-summed from the bins, i.e. word 0 <- word 2)
		const bool copied = every_word || (w >= 1 && w <= 4) || w == 11 || (w >= 16 && w < 16 + S);
		if (copied) dst[(uint64_t)ds[pair] * words_per_slot + w] = src[(uint64_t)ss[pair] * words_per_slot + (every_word == 2 && w == 0 ? 2u : w)];
	}
}
}  // namespace

hipError_t msc_launch_assign_batch(hipStream_t st, const MscLayout& L, uint8_t* dst_bins, uint8_t* dst_scalars, const uint8_t* src_bins,
                                   const uint8_t* src_scalars, const uint32_t* dst_slots, const uint32_t* src_slots, uint32_t n, int exact) {
	if (n == 0) return hipSuccess;
	static_assert(offsetof(MscSlotScalars, length) == 8 && offsetof(MscSlotScalars, max_count) == 32 && offsetof(MscSlotScalars, id) == 88 && sizeof(MscSlotScalars) == 128,
	              "k_assign_scalars hard-codes the record's word positions");
	const uint64_t per = L.slot_bytes / 16, total = per * n;
	k_assign_bins<<<dim3((unsigned)std::min<uint64_t>((total + kBlock - 1) / kBlock, 1u << 20)), dim3(kBlock), 0, st>>>((uint4*)dst_bins, (const uint4*)src_bins, dst_slots, src_slots,
	                                                                                                                  per, total);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	const uint32_t words = (uint32_t)(msc_scalar_stride(L.S) / 8);
	const uint64_t stotal = (uint64_t)words * n;
	k_assign_scalars<<<dim3((unsigned)std::min<uint64_t>((stotal + kBlock - 1) / kBlock, 1u << 20)), dim3(kBlock), 0, st>>>((uint64_t*)dst_scalars, (const uint64_t*)src_scalars,
	                                                                                                                      dst_slots, src_slots, words, L.S, stotal, exact);
	return hipGetLastError();
}

// the scalar half of msc_launch_assign_batch on its own (sparse stores: 128-byte records, no tile prefixes)
hipError_t msc_launch_assign_scalars(hipStream_t st, uint8_t* dst_scalars, const uint8_t* src_scalars, uint64_t stride_bytes, const uint32_t* dst_slots,
                                     const uint32_t* src_slots, uint32_t n, int exact) {
	if (n == 0) return hipSuccess;
	const uint32_t words = (uint32_t)(stride_bytes / 8);
	const uint64_t stotal = (uint64_t)words * n;
	k_assign_scalars<<<dim3((unsigned)std::min<uint64_t>((stotal + kBlock - 1) / kBlock, 1u << 20)), dim3(kBlock), 0, st>>>((uint64_t*)dst_scalars, (const uint64_t*)src_scalars,
	                                                                                                                      dst_slots, src_slots, words, 0, stotal, exact);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------ 4-bin group statistics of small dense sets
// sim_mm (through markov) and rre_k_r on histograms too small for the sparse layout (4^k * sizeof(T) < 64 KiB, where the list
// form of sparse.hip does not exist): the same per-group terms (msc_groups.h), the same 16 index sub-ranges per candidate and the
// same records as k_pair_sparse_groups / k_sparse_self_markov, read straight from the tile-permuted dense slots. One lane per
// (candidate, sub-range) walks its groups in index order; the slots are a few KiB and stay in L2.
template <typename T>
__device__ __forceinline__ void load_group(const T* __restrict__ h, uint64_t g, uint32_t E, uint32_t R, uint32_t (&v)[4]) {
#pragma unroll
	for (int j = 0; j < 4; j++) v[j] = (uint32_t)h[msc_phys_index(4 * g + j, E, R)];
}
__device__ __forceinline__ void group_span(uint64_t nbins, uint32_t r, uint64_t& g0, uint64_t& g1) {
	const uint64_t G = nbins / 4, per = (G + 15) / 16;
	g0 = per * r < G ? per * r : G;
	g1 = g0 + per < G ? g0 + per : G;
}
template <typename T>
__global__ void __launch_bounds__(256) k_pair_groups_dense(const uint8_t* __restrict__ c_bins, uint64_t slot_bytes, const uint8_t* __restrict__ cand_scalars,
                                                           uint64_t scalar_stride, const uint32_t* __restrict__ cand_slots, uint32_t m,
                                                           const uint8_t* __restrict__ q_bins, uint32_t E, uint32_t R, uint64_t nbins, int use_window,
                                                           uint64_t min_len, uint64_t max_len, double* __restrict__ out) {
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t c = (uint32_t)(t / 16), r = (uint32_t)(t % 16);
	if (c >= m) return;
	const uint32_t slot = cand_slots ? cand_slots[c] : c;
	const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride);
	if (use_window && (cs->length < min_len || cs->length > max_len)) return;
	const T* P = reinterpret_cast<const T*>(c_bins + (uint64_t)slot * slot_bytes);
	const T* Q = reinterpret_cast<const T*>(q_bins);
	uint64_t g0, g1;
	group_span(nbins, r, g0, g1);
	double markov = 0.0, rre = 0.0;
	for (uint64_t g = g0; g < g1; g++) {
		uint32_t p[4], q[4];
		load_group(P, g, E, R, p);
		load_group(Q, g, E, R, q);
		if ((p[0] & p[1] & p[2] & p[3] & q[0] & q[1] & q[2] & q[3]) == 1u && (p[0] | p[1] | p[2] | p[3] | q[0] | q[1] | q[2] | q[3]) == 1u) continue;
		msc_group_terms(p, q, markov, rre);
	}
	out[((uint64_t)c * 16 + r) * 2] = markov;
	out[((uint64_t)c * 16 + r) * 2 + 1] = rre;
}
template <typename T>
__global__ void __launch_bounds__(256) k_self_markov_dense(const uint8_t* __restrict__ bins, uint64_t slot_bytes, const uint32_t* __restrict__ slots,
                                                           uint64_t first_slot, uint32_t m, uint32_t E, uint32_t R, uint64_t nbins, double* __restrict__ out) {
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t c = (uint32_t)(t / 16), r = (uint32_t)(t % 16);
	if (c >= m) return;
	const T* P = reinterpret_cast<const T*>(bins + (slots ? (uint64_t)slots[c] : first_slot + c) * slot_bytes);
	uint64_t g0, g1;
	group_span(nbins, r, g0, g1);
	double total = 0.0;
	for (uint64_t g = g0; g < g1; g++) {
		uint32_t v[4];
		load_group(P, g, E, R, v);
		if ((v[0] | v[1] | v[2] | v[3]) == 1u && (v[0] & v[1] & v[2] & v[3]) == 1u) continue;
		msc_group_self(v, total);
	}
	out[(uint64_t)c * 16 + r] = total;
}

// c_bins / cand_scalars: base of the set when cand_slots != nullptr, else of the first candidate of the chunk
hipError_t msc_launch_pair_groups_dense(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* c_bins, const uint8_t* cand_scalars, uint64_t scalar_stride,
                                        const uint32_t* cand_slots, uint32_t m, const uint8_t* q_bins, int use_window, uint64_t min_len, uint64_t max_len, double* out) {
	if (m == 0) return hipSuccess;
	const dim3 grid((unsigned)(((uint64_t)m * 16 + 255) / 256));
#define MSC_GD(TT) k_pair_groups_dense<TT><<<grid, dim3(256), 0, st>>>(c_bins, L.slot_bytes, cand_scalars, scalar_stride, cand_slots, m, q_bins, L.E, L.R, L.nbins, use_window, min_len, max_len, out)
	switch (dtype) {
	case 8: MSC_GD(uint8_t); break;
	case 16: MSC_GD(uint16_t); break;
	case 32: MSC_GD(uint32_t); break;
	default: MSC_GD(uint64_t); break;
	}
#undef MSC_GD
	return hipGetLastError();
}
// bins: base of the set; the slots are slots[c] or first_slot + c
hipError_t msc_launch_self_markov_dense(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, const uint32_t* slots, uint64_t first_slot, uint32_t m,
                                        double* out) {
	if (m == 0) return hipSuccess;
	const dim3 grid((unsigned)(((uint64_t)m * 16 + 255) / 256));
#define MSC_SD(TT) k_self_markov_dense<TT><<<grid, dim3(256), 0, st>>>(bins, L.slot_bytes, slots, first_slot, m, L.E, L.R, L.nbins, out)
	switch (dtype) {
	case 8: MSC_SD(uint8_t); break;
	case 16: MSC_SD(uint16_t); break;
	case 32: MSC_SD(uint32_t); break;
	default: MSC_SD(uint64_t); break;
	}
#undef MSC_SD
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------- batched launchers (msc_update_centres)
bool msc_batch_tiles_supported(const MscLayout& L, int dtype) { (void)L; return dtype == 8 || dtype == 16 || dtype == 32 || dtype == 64; }

template <typename T>
static hipError_t launch_tiles_batch_t(hipStream_t st, const MscLayout& L, const uint8_t* cb, const uint8_t* cs, const uint32_t* sl, const MscBatchSeg* segs,
                                       uint32_t n_segs, uint32_t max_m, const uint8_t* qb, uint64_t qsb, const uint8_t* qs, uint64_t qss, int use_window,
                                       MscPartial* partials, int order) {
	const uint32_t S = L.S;
	// short lists: a few candidate groups per segment are enough to cover the latency; every segment gets the same grid slice
	uint32_t G = max_m < 4 ? (max_m ? max_m : 1) : 4;
	const uint32_t bps = (S * G + kWavesPerBlock - 1) / kWavesPerBlock;
	const bool padded = L.nbins < L.padded_bins;
	const uint64_t stride = msc_scalar_stride(S);
	const uint32_t nvalid = padded ? (uint32_t)L.nbins : L.tile_bins;
	const dim3 grid((unsigned)((uint64_t)n_segs * bps));
#define MSC_TB(LPTV, PAD) k_pair_tiles_batch<T, LPTV, PAD><<<grid, dim3(kBlock), 0, st>>>(cb, L.slot_bytes, cs, stride, sl, segs, bps, qb, qsb, qs, qss, S, G, nvalid, use_window, partials, order)
	switch (L.LPT) {
	case 1: if (padded) MSC_TB(1, true); else MSC_TB(1, false); break;
	case 2: if (padded) MSC_TB(2, true); else MSC_TB(2, false); break;
	default: if (padded) MSC_TB(4, true); else MSC_TB(4, false); break;
	}
#undef MSC_TB
	return hipGetLastError();
}

hipError_t msc_launch_pair_tiles_batch(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* cand_bins, const uint8_t* cand_scalars,
                                       const uint32_t* cand_slots, const MscBatchSeg* segs, uint32_t n_segs, uint32_t max_m, const uint8_t* qset_bins,
                                       uint64_t q_slot_bytes, const uint8_t* qset_scalars, uint64_t q_scalar_stride, int use_window, MscPartial* partials,
                                       int order) {
	if (n_segs == 0) return hipSuccess;
	switch (dtype) {
	case 8: return launch_tiles_batch_t<uint8_t>(st, L, cand_bins, cand_scalars, cand_slots, segs, n_segs, max_m, qset_bins, q_slot_bytes, qset_scalars, q_scalar_stride, use_window, partials, order);
	case 16: return launch_tiles_batch_t<uint16_t>(st, L, cand_bins, cand_scalars, cand_slots, segs, n_segs, max_m, qset_bins, q_slot_bytes, qset_scalars, q_scalar_stride, use_window, partials, order);
	case 32: return launch_tiles_batch_t<uint32_t>(st, L, cand_bins, cand_scalars, cand_slots, segs, n_segs, max_m, qset_bins, q_slot_bytes, qset_scalars, q_scalar_stride, use_window, partials, order);
	default: return launch_tiles_batch_t<uint64_t>(st, L, cand_bins, cand_scalars, cand_slots, segs, n_segs, max_m, qset_bins, q_slot_bytes, qset_scalars, q_scalar_stride, use_window, partials, order);
	}
}

hipError_t msc_launch_colsum_batch(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, const uint32_t* member_slots, const MscBatchSeg* segs,
                                   uint32_t n_segs, void* rounded_out, uint64_t* floor_sum_out) {
	if (n_segs == 0) return hipSuccess;
	hipError_t e = hipMemsetAsync(floor_sum_out, 0, sizeof(uint64_t) * n_segs, st);
	if (e != hipSuccess) return e;
	const uint64_t chunks = L.slot_bytes / 16;
	const uint32_t cpb = (uint32_t)((chunks + kBlock - 1) / kBlock);
	const dim3 grid((unsigned)((uint64_t)n_segs * cpb));
	unsigned long long* fs = reinterpret_cast<unsigned long long*>(floor_sum_out);
	switch (dtype) {
	case 8: k_colsum_batch<uint8_t><<<grid, dim3(kBlock), 0, st>>>((const uint8_t*)bins, L.padded_bins, member_slots, segs, cpb, L.padded_bins, L.nbins, L.R, (uint8_t*)rounded_out, fs); break;
	case 16: k_colsum_batch<uint16_t><<<grid, dim3(kBlock), 0, st>>>((const uint16_t*)bins, L.padded_bins, member_slots, segs, cpb, L.padded_bins, L.nbins, L.R, (uint16_t*)rounded_out, fs); break;
	case 32: k_colsum_batch<uint32_t><<<grid, dim3(kBlock), 0, st>>>((const uint32_t*)bins, L.padded_bins, member_slots, segs, cpb, L.padded_bins, L.nbins, L.R, (uint32_t*)rounded_out, fs); break;
	default: k_colsum_batch<uint64_t><<<grid, dim3(kBlock), 0, st>>>((const uint64_t*)bins, L.padded_bins, member_slots, segs, cpb, L.padded_bins, L.nbins, L.R, (uint64_t*)rounded_out, fs); break;
	}
	return hipGetLastError();
}

hipError_t msc_launch_colsum_sums(hipStream_t st, const MscLayout& L, int dtype, const uint8_t* bins, const uint32_t* member_slots, const MscBatchSeg* segs, uint32_t n_segs,
                                  uint64_t* sums_out) {
	if (n_segs == 0) return hipSuccess;
	const uint64_t chunks = L.padded_bins / L.E;
	const uint32_t cpb = (uint32_t)((chunks + kBlock - 1) / kBlock);
	const dim3 grid(n_segs * cpb);
	switch (dtype) {
	case 8: k_colsum_sums<uint8_t><<<grid, dim3(kBlock), 0, st>>>((const uint8_t*)bins, L.padded_bins, member_slots, segs, cpb, L.padded_bins, (unsigned long long*)sums_out); break;
	case 16: k_colsum_sums<uint16_t><<<grid, dim3(kBlock), 0, st>>>((const uint16_t*)bins, L.padded_bins, member_slots, segs, cpb, L.padded_bins, (unsigned long long*)sums_out); break;
	case 32: k_colsum_sums<uint32_t><<<grid, dim3(kBlock), 0, st>>>((const uint32_t*)bins, L.padded_bins, member_slots, segs, cpb, L.padded_bins, (unsigned long long*)sums_out); break;
	default: k_colsum_sums<uint64_t><<<grid, dim3(kBlock), 0, st>>>((const uint64_t*)bins, L.padded_bins, member_slots, segs, cpb, L.padded_bins, (unsigned long long*)sums_out); break;
	}
	return hipGetLastError();
}
// sums[n_segs][padded] of m_total[si] members -> rounded mean slots 0 .. n_segs-1 behind rounded_out and floor_sum_out[si] (zeroed here)
hipError_t msc_launch_mean_from_sums(hipStream_t st, const MscLayout& L, int dtype, const uint64_t* sums, const uint64_t* m_total, uint32_t n_segs, void* rounded_out,
                                     uint64_t* floor_sum_out) {
	if (n_segs == 0) return hipSuccess;
	hipError_t e = hipMemsetAsync(floor_sum_out, 0, (size_t)n_segs * sizeof(uint64_t), st);
	if (e != hipSuccess) return e;
	const uint64_t chunks = L.padded_bins / L.E;
	const uint32_t cpb = (uint32_t)((chunks + kBlock - 1) / kBlock);
	const dim3 grid(n_segs * cpb);
	const unsigned long long* su = (const unsigned long long*)sums;
	const unsigned long long* mt = (const unsigned long long*)m_total;
	unsigned long long* fs = (unsigned long long*)floor_sum_out;
	switch (dtype) {
	case 8: k_mean_from_sums<uint8_t><<<grid, dim3(kBlock), 0, st>>>(su, mt, cpb, L.padded_bins, L.nbins, L.R, L.padded_bins, (uint8_t*)rounded_out, fs); break;
	case 16: k_mean_from_sums<uint16_t><<<grid, dim3(kBlock), 0, st>>>(su, mt, cpb, L.padded_bins, L.nbins, L.R, L.padded_bins, (uint16_t*)rounded_out, fs); break;
	case 32: k_mean_from_sums<uint32_t><<<grid, dim3(kBlock), 0, st>>>(su, mt, cpb, L.padded_bins, L.nbins, L.R, L.padded_bins, (uint32_t*)rounded_out, fs); break;
	default: k_mean_from_sums<uint64_t><<<grid, dim3(kBlock), 0, st>>>(su, mt, cpb, L.padded_bins, L.nbins, L.R, L.padded_bins, (uint64_t*)rounded_out, fs); break;
	}
	return hipGetLastError();
}

hipError_t msc_launch_distance_batch(hipStream_t st, const MscPartial* partials, uint32_t S, uint32_t n, const uint8_t* scalars, uint64_t scalar_stride,
                                     const uint32_t* member_slots, const uint32_t* pair_seg, const uint8_t* r_scalars, uint64_t r_stride, const uint64_t* floor_sum,
                                     double* dist_out) {
	if (n == 0) return hipSuccess;
	k_distance_batch<<<dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st>>>(partials, S, n, scalars, scalar_stride, member_slots, pair_seg, r_scalars, r_stride, floor_sum, dist_out);
	return hipGetLastError();
}
