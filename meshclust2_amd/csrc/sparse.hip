// sparse.hip -- sparse histogram representation and the merge-based pair kernel (gfx950).
//
// A 4^k histogram of a sequence of L bases holds at most L bins above the pseudocount; at k = 13 (67 M bins) the dense
// form is 64-512 MiB per sequence and cannot exist for a data set, and already at k = 9 / 1 kb 99.6 % of the
// dense bytes are the constant 1. A SPARSE slot keeps only the bins with value >= 2:
//     ent[t] = (bin index, value)   sorted by index          (value = min(1 + occurrences, max(T)), exactly the dense bin)
//     cum[t] = sum_{s<=t} (value_s - 1)                      inclusive "excess" prefix
//     split[0..16]                                           entry offsets of 16 equal index sub-ranges
// Every in-scope statistic is recovered EXACTLY from the sorted union U of two such lists, because outside U both bins are 1:
//     manh = sum_U |p-q|        dot = N + sum_U (p*q - 1)        sum min, sum (p-q)^2 ... as in the dense path
//     emd  = sum over maximal stretches between events of |D| * length,  D = cumP - cumQ  (prefix difference is piecewise constant)
//     jefferey / jensen-shannon = sum_U term(p,q) + (N - |U|) * term(1,1)
// so the dense epilogue (pair_features.hip) is reused unchanged on 16 partial records per candidate.
//
// k_pair_sparse_mp (default): one WAVE per candidate walks the merged order of the two lists in LDS-staged chunks whose
// boundaries, and the lanes' shares inside a chunk, come from merge-path co-rank searches (see the kernel).
// k_pair_sparse (wide arithmetic range): one LANE per (candidate, index sub-range): a branch-light two-pointer merge of the two
// sorted lists restricted to the sub-range, straight from global memory; the prefix difference at a sub-range start comes from the
// stored cum arrays. k_pair_sparse_lds: the whole-list LDS variant the merge-path kernel replaced (MSC_SPARSE_LDS=1).
// Bytes per pair: 8 B per stored bin of the candidate list instead of 4^k * sizeof(T) per histogram.
//
// Build: the dense builder (hist_build.hip) fills a scratch slot per sequence of the batch, then k_sparse_count /
// k_sparse_write compact it IN INDEX ORDER: in the tile-permuted layout every lane already holds a logically consecutive
// run, so one wave scan per tile yields ordered output with coalesced reads.
#include <algorithm>

#include "msc_groups.h"
#include "msc_internal.h"
#include "msc_wave.h"

namespace {

constexpr int kBlockC = 1024;                 // 16 waves = the 16 index sub-ranges of one sequence
constexpr int kSub = MSC_SPARSE_SUB;

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

// ------------------------------------------------------------------------------------------------ compaction, pass 1
// counts[seq][w] = {entries, excess sum} of index sub-range w (wave w streams tiles [w*S/16, (w+1)*S/16))
template <typename T>
__global__ void __launch_bounds__(kBlockC) k_sparse_count(const T* __restrict__ bins, uint64_t slot_elems, uint32_t S, uint32_t R,
                                                         uint64_t* __restrict__ counts /* [n][16][2] */) {
	constexpr uint32_t E = 16 / sizeof(T);
	const uint32_t seq = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t loads = R / E, tile_bins = 64 * R;
	const T* h = bins + (uint64_t)seq * slot_elems;
	const uint32_t t0 = (uint32_t)((uint64_t)wave * S / kSub), t1 = (uint32_t)((uint64_t)(wave + 1) * S / kSub);
	uint64_t n = 0, ex = 0;
	for (uint32_t t = t0; t < t1; t++) {
		for (uint32_t l = 0; l < loads; l++) {
			const uint4 v = *reinterpret_cast<const uint4*>(h + (uint64_t)t * tile_bins + (uint64_t)l * 64 * E + lane * E);
			const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
			for (uint32_t j = 0; j < E; j++) {
				const uint64_t p = e[j];
				if (p > 1) { n++; ex += p - 1; }
			}
		}
	}
	n = wave_sum_u64(n);
	ex = wave_sum_u64(ex);
	if (lane == 0) { counts[((uint64_t)seq * kSub + wave) * 2] = n; counts[((uint64_t)seq * kSub + wave) * 2 + 1] = ex; }
}

// ------------------------------------------------------------------------------------------------ compaction, pass 2
// wave w writes its sub-range's entries at hdr.off + hdr.split[w], ordered by index, with the running excess prefix
template <typename T>
__global__ void __launch_bounds__(kBlockC) k_sparse_write(const T* __restrict__ bins, uint64_t slot_elems, uint32_t S, uint32_t R,
                                                         const MscSparseHdr* __restrict__ hdr, uint64_t first_slot,
                                                         const uint64_t* __restrict__ cum_base /* [n][16] */, uint2* __restrict__ ent,
                                                         uint32_t* __restrict__ cum) {
	constexpr uint32_t E = 16 / sizeof(T);
	const uint32_t seq = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t loads = R / E, tile_bins = 64 * R;
	const T* h = bins + (uint64_t)seq * slot_elems;
	const MscSparseHdr& hd = hdr[first_slot + seq];
	uint64_t o = hd.off + hd.split[wave];
	uint32_t run = (uint32_t)cum_base[(uint64_t)seq * kSub + wave];
	const uint32_t t0 = (uint32_t)((uint64_t)wave * S / kSub), t1 = (uint32_t)((uint64_t)(wave + 1) * S / kSub);
	for (uint32_t t = t0; t < t1; t++) {
		// this lane's logically consecutive run of R bins
		uint32_t cnt = 0, ex = 0;
		for (uint32_t l = 0; l < loads; l++) {
			const uint4 v = *reinterpret_cast<const uint4*>(h + (uint64_t)t * tile_bins + (uint64_t)l * 64 * E + lane * E);
			const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
			for (uint32_t j = 0; j < E; j++) { const uint64_t p = e[j]; if (p > 1) { cnt++; ex += (uint32_t)(p - 1); } }
		}
		const uint32_t cnt_incl = wave_incl_scan(cnt), ex_incl = wave_incl_scan(ex);
		const uint32_t tile_cnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt_incl, 63);
		const uint32_t tile_ex = (uint32_t)__builtin_amdgcn_readlane((int)ex_incl, 63);
		if (tile_cnt) {
			uint64_t w = o + (cnt_incl - cnt);
			uint32_t c = run + (ex_incl - ex);
			if (cnt) {
				for (uint32_t l = 0; l < loads; l++) {
					const uint4 v = *reinterpret_cast<const uint4*>(h + (uint64_t)t * tile_bins + (uint64_t)l * 64 * E + lane * E);
					const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
					for (uint32_t j = 0; j < E; j++) {
						const uint64_t p = e[j];
						if (p > 1) {
							c += (uint32_t)(p - 1);
							ent[w] = make_uint2(t * tile_bins + lane * R + l * E + j, (uint32_t)p);
							cum[w] = c;
							w++;
						}
					}
				}
			}
		}
		o += tile_cnt;
		run += tile_ex;
	}
}

// ------------------------------------------------------------------------------------------------ direct sparse build (sort)
// One workgroup per sequence: its k-mer indices go to LDS, are bitonic-sorted there, and the runs of equal indices become
// the (bin, value) entries -- no dense scratch slot at all (a k = 13 scratch slot is 64-512 MiB). Sequences with more than
// 32768 k-mers fall back to the scratch + compaction path above.
__device__ __forceinline__ uint32_t rev2_sp(uint32_t x) {
	x = __brev(x);
	return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

constexpr int kSortBlock = 256;
__global__ void __launch_bounds__(kSortBlock) k_sparse_build_sort(
    int k, int dtype, uint64_t nbins, uint64_t first_slot, const uint32_t* __restrict__ packed, const uint64_t* __restrict__ seg_start,
    const uint64_t* __restrict__ kmer_off, const uint64_t* __restrict__ seq_seg_begin, const uint64_t* __restrict__ seq_arena_off,
    uint32_t P /* LDS keys, power of two >= k-mers of the longest sequence of the launch */, uint8_t* __restrict__ scalars, uint64_t scalar_stride,
    MscSparseHdr* __restrict__ hdr, uint2* __restrict__ ent, uint32_t* __restrict__ cum) {
	extern __shared__ uint32_t keys[];                 // P keys
	__shared__ uint64_t s_cnt[kSortBlock], s_ex[kSortBlock];
	__shared__ uint64_t s_red[4][kSortBlock / 64];
	__shared__ uint32_t s_split[MSC_SPARSE_SUB + 1];
	const uint32_t seq = blockIdx.x, tid = threadIdx.x;
	const uint64_t slot = first_slot + seq;
	// 1. this sequence's k-mer indices
	const uint64_t sb = seq_seg_begin[seq], se = seq_seg_begin[seq + 1];
	const uint64_t k0 = kmer_off[sb];
	const uint32_t n = (uint32_t)(kmer_off[se] - k0);
	for (uint64_t j = sb; j < se; j++) {
		const uint64_t nk = kmer_off[j + 1] - kmer_off[j], base = seg_start[j];
		const uint32_t o = (uint32_t)(kmer_off[j] - k0);
		for (uint64_t t = tid; t < nk; t += kSortBlock) {
			const uint64_t pos = base + t;
			const uint64_t window = (uint64_t)packed[pos >> 4] | ((uint64_t)packed[(pos >> 4) + 1] << 32);
			uint32_t bits = (uint32_t)(window >> ((pos & 15) * 2));
			if (2 * k < 32) bits &= (1u << (2 * k)) - 1u;
			keys[o + t] = rev2_sp(bits) >> (32 - 2 * k);
		}
	}
	for (uint32_t i = n + tid; i < P; i += kSortBlock) keys[i] = 0xffffffffu;
	__syncthreads();
	// 2. bitonic sort of P keys
	for (uint32_t size = 2; size <= P; size <<= 1) {
		for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
			for (uint32_t t = tid; t < P / 2; t += kSortBlock) {
				const uint32_t lo = 2 * t - (t & (stride - 1));
				const uint32_t hi = lo + stride;
				const bool up = (lo & size) == 0;
				const uint32_t a = keys[lo], b = keys[hi];
				if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
			}
			__syncthreads();
		}
	}
	// 3. runs of equal keys -> entries. Thread t owns the consecutive chunk [t*C, (t+1)*C) of the sorted array.
	const uint64_t tmax = dtype == 64 ? ~0ull : ((1ull << dtype) - 1);
	const uint32_t C = P / kSortBlock > 0 ? P / kSortBlock : 1;
	const uint32_t c0 = tid * C, c1 = c0 + C < n ? c0 + C : (c0 < n ? n : c0);
	auto run_len = [&](uint32_t i) -> uint32_t {          // i is a run head: length = upper_bound(keys[i]) - i
		const uint32_t key = keys[i];
		uint32_t lo = i + 1, hi = n;
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] <= key) lo = mid + 1; else hi = mid; }
		return lo - i;
	};
	uint64_t cnt = 0, ex = 0, sq = 0, mx = 0, ovf = 0;
	for (uint32_t i = c0; i < c1; i++) {
		if (i == 0 || keys[i] != keys[i - 1]) {
			uint64_t v = 1ull + run_len(i);
			if (v > tmax) { v = tmax; ovf = 1; }
			cnt++; ex += v - 1; sq += v * v - 1; mx = v > mx ? v : mx;
		}
	}
	s_cnt[tid] = cnt;
	s_ex[tid] = ex;
	__syncthreads();
	if (tid == 0) {                                       // 256-element exclusive scans: serial is fine
		uint64_t a = 0, b = 0;
		for (int i = 0; i < kSortBlock; i++) { const uint64_t x = s_cnt[i], y = s_ex[i]; s_cnt[i] = a; s_ex[i] = b; a += x; b += y; }
		s_red[0][0] = a;                                  // total entries
		s_red[1][0] = b;                                  // total excess
	}
	__syncthreads();
	const uint64_t arena = seq_arena_off[seq];
	{
		uint64_t o = arena + s_cnt[tid];
		uint32_t run = (uint32_t)s_ex[tid];
		for (uint32_t i = c0; i < c1; i++) {
			if (i == 0 || keys[i] != keys[i - 1]) {
				uint64_t v = 1ull + run_len(i);
				if (v > tmax) v = tmax;
				run += (uint32_t)(v - 1);
				ent[o] = make_uint2(keys[i], (uint32_t)v);
				cum[o] = run;
				o++;
			}
		}
	}
	// 4. sub-range offsets: entries with index < w * N/16  ==  run heads before the first key >= that boundary
	if (tid <= MSC_SPARSE_SUB) {
		const uint64_t bound = nbins / MSC_SPARSE_SUB * tid;
		uint32_t lo = 0, hi = n;
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)keys[mid] < bound) lo = mid + 1; else hi = mid; }
		const uint32_t owner = lo / C < (uint32_t)kSortBlock ? lo / C : kSortBlock - 1;
		uint32_t heads = (uint32_t)s_cnt[owner];
		for (uint32_t i = owner * C; i < lo; i++) if (i == 0 || keys[i] != keys[i - 1]) heads++;
		s_split[tid] = tid == MSC_SPARSE_SUB ? (uint32_t)s_red[0][0] : heads;
	}
	// 5. block reductions for the scalar record
	sq = wave_sum_u64(sq);
	uint64_t m2 = mx, o2 = ovf;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) { const uint64_t a = __shfl_xor(m2, off, 64); m2 = a > m2 ? a : m2; o2 |= __shfl_xor(o2, off, 64); }
	if ((tid & 63) == 0) { s_red[2][tid >> 6] = sq; s_red[3][tid >> 6] = m2 | (o2 << 63); }
	__syncthreads();
	if (tid == 0) {
		uint64_t sqt = 0, mxt = 1, ov = 0;
		for (int i = 0; i < kSortBlock / 64; i++) { sqt += s_red[2][i]; const uint64_t m_ = s_red[3][i] & ~(1ull << 63); mxt = m_ > mxt ? m_ : mxt; ov |= s_red[3][i] >> 63; }
		MscSlotScalars* sc = reinterpret_cast<MscSlotScalars*>(scalars + slot * scalar_stride);
		const uint64_t sum = nbins + s_red[1][0], sum_sq = nbins + sqt;
		sc->sum = sum; sc->sum_sq = sum_sq; sc->max_count = mxt; sc->mag = sum; sc->overflow = ov;
		const double N = (double)nbins, aq = (double)sum / N;
		const double var = ((double)sum_sq - 2.0 * aq * (double)sum + N * aq * aq) / N;
		sc->stddev = sqrt(var > 0 ? var : 0);
		MscSparseHdr h;
		h.off = arena;
		h.nnz = (uint32_t)s_red[0][0];
		for (int w = 0; w <= MSC_SPARSE_SUB; w++) h.split[w] = s_split[w];
		h.pad_[0] = h.pad_[1] = 0;
		hdr[slot] = h;
	}
}

// ------------------------------------------------------------------------------------------------ pair kernel
struct DivTerm { double jd, js; };
__device__ __forceinline__ DivTerm div_term_sp(uint32_t cand_count, uint32_t q_count, double cand_mag, double q_mag, int order) {
	DivTerm t;
	const bool cf = order == MSC_ORDER_CAND_FIRST;
	const double pp = cf ? (double)cand_count / cand_mag : (double)q_count / q_mag;
	const double pq = cf ? (double)q_count / q_mag : (double)cand_count / cand_mag;
	t.jd = (pp - pq) * log(pp / pq);
	const double avg = 0.5 * (pp + pq);
	t.js = pp * log(pp / avg) + pq * log(pq / avg);
	return t;
}

// the same as a call: the merge-path kernel evaluates a term directly only off its hot path (once per pair, and for the counts its
// table does not hold), and three FP64 logs inlined there cost the walk its registers
__device__ __noinline__ DivTerm div_term_call(uint32_t cand_count, uint32_t q_count, double cand_mag, double q_mag, int order) {
	return div_term_sp(cand_count, q_count, cand_mag, q_mag, order);
}

// per-candidate 16 x 16 table of the exact per-bin divergence terms (same idea as k_div_tables of the dense path)
// (segs != nullptr: the pair-list form -- the query of pair c is slot segs[pair_seg[c]].q_slot behind q_scalars, stride q_stride)
__global__ void __launch_bounds__(256) k_sparse_div_tables(const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
                                                           const uint32_t* __restrict__ cand_slots, uint32_t m,
                                                           const uint8_t* __restrict__ q_scalars, int order, DivTerm* __restrict__ tables,
                                                           const MscBatchSeg* __restrict__ segs = nullptr, const uint32_t* __restrict__ pair_seg = nullptr,
                                                           uint64_t q_stride = 0) {
	const uint32_t c = blockIdx.x;
	if (c >= m) return;
	const uint32_t slot = cand_slots ? cand_slots[c] : c;
	const double cm = (double)reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride)->mag;
	const double qm = (double)reinterpret_cast<const MscSlotScalars*>(q_scalars + (segs ? (uint64_t)segs[pair_seg[c]].q_slot * q_stride : 0))->mag;
	const uint32_t j = threadIdx.x;
	DivTerm t{0.0, 0.0};
	if (j / 16 && j % 16) t = div_term_sp(j / 16, j % 16, cm, qm, order);
	tables[(uint64_t)c * 256 + j] = t;
}

template <bool DIV>
__global__ void __launch_bounds__(256) k_pair_sparse(
    const uint2* __restrict__ c_ent, const uint32_t* __restrict__ c_cum, const MscSparseHdr* __restrict__ c_hdr,
    const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride, const uint32_t* __restrict__ cand_slots, uint32_t m,
    const uint2* __restrict__ q_ent, const uint32_t* __restrict__ q_cum, const MscSparseHdr* __restrict__ q_hdr_p,
    const uint8_t* __restrict__ q_scalars, uint64_t nbins, int use_window, uint64_t min_len, uint64_t max_len,
    MscPartial* __restrict__ partials, const DivTerm* __restrict__ div_tables, double* __restrict__ div_partials, int order) {
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t c = (uint32_t)(t / kSub), r = (uint32_t)(t % kSub);
	if (c >= m) return;
	const uint32_t slot = cand_slots ? cand_slots[c] : c;
	const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride);
	if (use_window && (cs->length < min_len || cs->length > max_len)) return;
	const MscSparseHdr ch = c_hdr[slot];
	const MscSparseHdr qh = *q_hdr_p;
	const uint2* P = c_ent + ch.off;
	const uint32_t* CP = c_cum + ch.off;
	const uint2* Q = q_ent + qh.off;
	const uint32_t* CQ = q_cum + qh.off;
	uint32_t i = ch.split[r], iend = ch.split[r + 1];
	uint32_t j = qh.split[r], jend = qh.split[r + 1];
	const uint64_t range_begin = nbins / kSub * r, range_end = nbins / kSub * (r + 1);
	int64_t D = (int64_t)(i ? CP[i - 1] : 0u) - (int64_t)(j ? CQ[j - 1] : 0u);     // prefix difference entering the sub-range
	uint64_t pos = range_begin, manh = 0, dotx = 0, emd = 0;
	double jd = 0.0, js = 0.0;
	double cm = 0.0, qm = 0.0;
	DivTerm t11{0.0, 0.0};
	if constexpr (DIV) {
		cm = (double)cs->mag;
		qm = (double)reinterpret_cast<const MscSlotScalars*>(q_scalars)->mag;
		t11 = div_term_sp(1, 1, cm, qm, order);
	}
	const uint32_t kInf = 0xffffffffu;
	uint2 a = i < iend ? P[i] : make_uint2(kInf, 1u);
	uint2 b = j < jend ? Q[j] : make_uint2(kInf, 1u);
	while (i < iend || j < jend) {
		const uint32_t e = a.x < b.x ? a.x : b.x;
		const bool ta = a.x == e, tb = b.x == e;
		const uint64_t absD = (uint64_t)(D < 0 ? -D : D);
		emd += absD * ((uint64_t)e - pos);                 // bins [pos, e) all carry the prefix difference D
		const uint32_t pv = ta ? a.y : 1u, qv = tb ? b.y : 1u;
		manh += pv > qv ? pv - qv : qv - pv;
		dotx += (uint64_t)pv * qv - 1;
		D += (int64_t)pv - (int64_t)qv;
		if constexpr (DIV) {
			DivTerm tt;
			if ((pv | qv) < 16u) tt = div_tables[(uint64_t)c * 256 + pv * 16 + qv];
			else tt = div_term_sp(pv, qv, cm, qm, order);
			jd += tt.jd - t11.jd;
			js += tt.js - t11.js;
		}
		pos = e;
		if (ta) { i++; a = i < iend ? P[i] : make_uint2(kInf, 1u); }
		if (tb) { j++; b = j < jend ? Q[j] : make_uint2(kInf, 1u); }
	}
	{
		const uint64_t absD = (uint64_t)(D < 0 ? -D : D);
		emd += absD * (range_end - pos);
	}
	MscPartial out;
	out.manh = manh;
	out.dot = dotx;
	out.emd = emd;
	partials[(uint64_t)c * kSub + r] = out;
	if constexpr (DIV) { div_partials[((uint64_t)c * kSub + r) * 2] = jd; div_partials[((uint64_t)c * kSub + r) * 2 + 1] = js; }
}

// ------------------------------------------------------------------------------------------------ statistics over 4-bin groups
// Feature<T>::markov (-> sim_mm) and rre_k_r (predict/Feature.cpp:1367-1393,1429-1455,1029-1062) are sums over the groups of four
// neighbouring bins that share a (k-1)-mer prefix. A group in which both histograms hold only pseudocounts contributes exactly 0
// to either (every factor (count - 1) vanishes; the conditional distributions are both uniform, log 1 = 0), so the sums run over
// the groups that hold a stored bin of either list: one lane per (candidate, index sub-range) merges the two lists as
// k_pair_sparse does -- a group never straddles a sub-range -- and closes a group when the merged order leaves it.
struct GroupState {
	uint32_t g;            // group (bin >> 2) being assembled, 0xffffffff = none
	uint32_t p[4], q[4];
	double markov, rre;
};
__device__ __forceinline__ void group_close(GroupState& st) {
	if (st.g == 0xffffffffu) return;
	msc_group_terms(st.p, st.q, st.markov, st.rre);
	st.g = 0xffffffffu;
}
__device__ __forceinline__ void group_put(GroupState& st, uint32_t bin, uint32_t pv, uint32_t qv) {
	const uint32_t g = bin >> 2;
	if (g != st.g) {
		group_close(st);
		st.g = g;
#pragma unroll
		for (int j = 0; j < 4; j++) { st.p[j] = 1u; st.q[j] = 1u; }
	}
	const uint32_t j = bin & 3u;
	// (no dynamic register indexing)
	st.p[0] = j == 0 ? pv : st.p[0]; st.p[1] = j == 1 ? pv : st.p[1]; st.p[2] = j == 2 ? pv : st.p[2]; st.p[3] = j == 3 ? pv : st.p[3];
	st.q[0] = j == 0 ? qv : st.q[0]; st.q[1] = j == 1 ? qv : st.q[1]; st.q[2] = j == 2 ? qv : st.q[2]; st.q[3] = j == 3 ? qv : st.q[3];
}

// out[(c * 16 + r) * 2 + {0, 1}] = {markov total (before the reference's / 2), rre op + oq} of sub-range r of candidate c against the query
__global__ void __launch_bounds__(256) k_pair_sparse_groups(
    const uint2* __restrict__ c_ent, const MscSparseHdr* __restrict__ c_hdr, const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
    const uint32_t* __restrict__ cand_slots, uint32_t m, const uint2* __restrict__ q_ent, const MscSparseHdr* __restrict__ q_hdr_p, int use_window,
    uint64_t min_len, uint64_t max_len, double* __restrict__ out) {
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t c = (uint32_t)(t / kSub), r = (uint32_t)(t % kSub);
	if (c >= m) return;
	const uint32_t slot = cand_slots ? cand_slots[c] : c;
	const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride);
	if (use_window && (cs->length < min_len || cs->length > max_len)) return;
	const MscSparseHdr ch = c_hdr[slot];
	const MscSparseHdr qh = *q_hdr_p;
	const uint2* P = c_ent + ch.off;
	const uint2* Q = q_ent + qh.off;
	uint32_t i = ch.split[r], iend = ch.split[r + 1];
	uint32_t j = qh.split[r], jend = qh.split[r + 1];
	GroupState st;
	st.g = 0xffffffffu; st.markov = 0.0; st.rre = 0.0;
	const uint32_t kInf = 0xffffffffu;
	uint2 a = i < iend ? P[i] : make_uint2(kInf, 1u);
	uint2 b = j < jend ? Q[j] : make_uint2(kInf, 1u);
	while (i < iend || j < jend) {
		const uint32_t e = a.x < b.x ? a.x : b.x;
		const bool ta = a.x == e, tb = b.x == e;
		group_put(st, e, ta ? a.y : 1u, tb ? b.y : 1u);
		if (ta) { i++; a = i < iend ? P[i] : make_uint2(kInf, 1u); }
		if (tb) { j++; b = j < jend ? Q[j] : make_uint2(kInf, 1u); }
	}
	group_close(st);
	out[((uint64_t)c * kSub + r) * 2] = st.markov;
	out[((uint64_t)c * kSub + r) * 2 + 1] = st.rre;
}

// MSC_PROFILE_CALLS: stored bins of the candidates of a pass that lie inside its length window, added to *acc (list bytes the pass
// reads = 8 x that) -- the byte count behind a roofline figure of the merge kernels in a whole clustering run
__global__ void __launch_bounds__(256) k_sparse_nnz_sum(const MscSparseHdr* __restrict__ hdr, const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
                                                        const uint32_t* __restrict__ slots, uint64_t first_slot, uint32_t m, int use_window, uint64_t min_len,
                                                        uint64_t max_len, unsigned long long* __restrict__ acc) {
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long v = 0, n = 0;
	if (c < m) {
		const uint64_t slot = slots ? slots[c] : first_slot + c;
		const uint64_t len = reinterpret_cast<const MscSlotScalars*>(cand_scalars + slot * scalar_stride)->length;
		if (!use_window || (len >= min_len && len <= max_len)) { v = hdr[slot].nnz; n = 1; }
	}
	v = wave_sum_u64(v);
	n = wave_sum_u64(n);
	if ((threadIdx.x & 63) == 0 && n) { atomicAdd(acc, v); atomicAdd(acc + 1, n); }
}

// markov(a, a) of single histograms (the denominators of d_markov): out[c * 16 + r] = sum over the groups of sub-range r of
// sum_j (a_j - 1) (log a_j - log group sum)  (= Feature<T>::markov(a, a), whose two equal terms per bin are halved at the end)
__global__ void __launch_bounds__(256) k_sparse_self_markov(const uint2* __restrict__ ent, const MscSparseHdr* __restrict__ hdr,
                                                            const uint32_t* __restrict__ slots, uint64_t first_slot, uint32_t m, double* __restrict__ out) {
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t c = (uint32_t)(t / kSub), r = (uint32_t)(t % kSub);
	if (c >= m) return;
	const MscSparseHdr h = hdr[slots ? slots[c] : first_slot + c];
	const uint2* P = ent + h.off;
	double total = 0.0;
	uint32_t g = 0xffffffffu, v[4] = {1u, 1u, 1u, 1u};
	auto close = [&]() {
		if (g == 0xffffffffu) return;
		msc_group_self(v, total);
	};
	for (uint32_t i = h.split[r]; i < h.split[r + 1]; i++) {
		const uint2 e = P[i];
		if ((e.x >> 2) != g) {
			close();
			g = e.x >> 2;
			v[0] = v[1] = v[2] = v[3] = 1u;
		}
		const uint32_t j = e.x & 3u;
		v[0] = j == 0 ? e.y : v[0]; v[1] = j == 1 ? e.y : v[1]; v[2] = j == 2 ? e.y : v[2]; v[3] = j == 3 ? e.y : v[3];
	}
	close();
	out[(uint64_t)c * kSub + r] = total;
}

// center->set(*next) for many centres of a sparse store at once: pair i appends a copy of src slot ss[i]'s entry list at
// dst_off[i] of the destination arena and points dst slot ds[i]'s header at it (same sub-range offsets, same length)
__global__ void __launch_bounds__(256) k_sparse_assign_batch(uint2* __restrict__ d_ent, uint32_t* __restrict__ d_cum, MscSparseHdr* __restrict__ d_hdr,
                                                             const uint2* __restrict__ s_ent, const uint32_t* __restrict__ s_cum, const MscSparseHdr* __restrict__ s_hdr,
                                                             const uint32_t* __restrict__ ds, const uint32_t* __restrict__ ss, const uint64_t* __restrict__ dst_off, uint32_t n) {
	const uint32_t i = blockIdx.x;
	if (i >= n) return;
	const MscSparseHdr h = s_hdr[ss[i]];
	const uint64_t o = dst_off[i];
	for (uint32_t t = threadIdx.x; t < h.nnz; t += blockDim.x) {
		d_ent[o + t] = s_ent[h.off + t];
		d_cum[o + t] = s_cum[h.off + t];
	}
	if (threadIdx.x == 0) {
		MscSparseHdr nh = h;
		nh.off = o;
		d_hdr[ds[i]] = nh;
	}
}

}  // namespace

// ------------------------------------------------------------------------------------------------ mean of sparse members
// get_mean / closest on sparse slots (cluster/ClusterFactory.cpp:338-380, cluster/Trainer.cpp:144-157). Outside the union of
// the members' stored bins every member holds 1, so the FP64 mean is exactly 1 there; inside, the column sum is m + (sum of
// excesses). The excesses are scatter-added into a dense u32 accumulator (4 * 4^k bytes, kept zero between calls), which is
// then swept in index order: chunk counts -> host prefix -> ordered write of the rounded mean as a sparse slot.
__global__ void __launch_bounds__(256) k_sparse_scatter(const uint2* __restrict__ ent, const MscSparseHdr* __restrict__ hdr,
                                                        const uint32_t* __restrict__ slots, uint32_t m, uint32_t* __restrict__ acc) {
	// (a member's list is shared by gridDim.y workgroups: with one per member the two or three long members of a cfg5 cluster were forty
	// dependent load -> atomic turns each, 25 us of every `closest` call)
	const uint32_t j = blockIdx.x;
	if (j >= m) return;
	const MscSparseHdr h = hdr[slots ? slots[j] : j];
	for (uint32_t t = blockIdx.y * blockDim.x + threadIdx.x; t < h.nnz; t += gridDim.y * blockDim.x) {
		const uint2 e = ent[h.off + t];
		atomicAdd(&acc[e.x], e.y - 1u);
	}
}

struct MeanBin { uint32_t r; uint64_t fl; };
template <typename T>
__device__ __forceinline__ MeanBin mean_bin(uint32_t E, uint32_t m) {
	MeanBin b;
	const double mean = (double)((uint64_t)m + E) / (double)m;       // (sum of the m bins) / m, cluster/ClusterFactory.cpp:349-357
	b.r = (uint32_t)(T)round(mean);                                   // (T)round(c.points[i]), clutil/DivergencePoint.cpp:61
	b.fl = (uint64_t)floor(mean);                                     // uint64 += double truncates every step (:62)
	return b;
}

// one wave per chunk of bins; counts[chunk] = {entries with r >= 2, sum (r-1), sum (floor-1)}
template <typename T>
__global__ void __launch_bounds__(64) k_sparse_mean_count(const uint32_t* __restrict__ acc, uint64_t chunk_bins, uint32_t m,
                                                          uint64_t* __restrict__ counts) {
	const uint64_t base = (uint64_t)blockIdx.x * chunk_bins;
	uint64_t n = 0, ex = 0, fl = 0;
	for (uint64_t i = threadIdx.x; i < chunk_bins; i += 64) {
		const uint32_t E = acc[base + i];
		if (E) {
			const MeanBin b = mean_bin<T>(E, m);
			if (b.r >= 2) { n++; ex += b.r - 1; }
			fl += b.fl - 1;
		}
	}
	n = wave_sum_u64(n); ex = wave_sum_u64(ex); fl = wave_sum_u64(fl);
	if (threadIdx.x == 0) { counts[blockIdx.x * 3ull] = n; counts[blockIdx.x * 3ull + 1] = ex; counts[blockIdx.x * 3ull + 2] = fl; }
}

template <typename T>
__global__ void __launch_bounds__(64) k_sparse_mean_write(uint32_t* __restrict__ acc, uint64_t chunk_bins, uint32_t m,
                                                          const uint64_t* __restrict__ chunk_off, const uint64_t* __restrict__ chunk_cum,
                                                          uint2* __restrict__ ent, uint32_t* __restrict__ cum) {
	const uint64_t base = (uint64_t)blockIdx.x * chunk_bins;
	uint64_t o = chunk_off[blockIdx.x];
	uint32_t run = (uint32_t)chunk_cum[blockIdx.x];
	const uint32_t lane = threadIdx.x;
	for (uint64_t i0 = 0; i0 < chunk_bins; i0 += 64) {
		const uint64_t i = i0 + lane;
		uint32_t E = 0;
		if (i < chunk_bins) { E = acc[base + i]; if (E) acc[base + i] = 0; }       // leave the accumulator clean for the next call
		uint32_t r = 1;
		if (E) r = mean_bin<T>(E, m).r;
		const bool emit = r >= 2;
		const unsigned long long mask = __ballot(emit);
		const uint32_t ex = emit ? r - 1 : 0;
		const uint32_t ex_incl = wave_incl_scan(ex);
		if (emit) {
			const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
			ent[o + rank] = make_uint2((uint32_t)(base + i), r);
			cum[o + rank] = run + ex_incl;
		}
		o += (uint64_t)__popcll(mask);
		run += (uint32_t)__builtin_amdgcn_readlane((int)ex_incl, 63);
	}
}

// ---- the same three steps for MANY centres at once (the update stage of a mean-shift round on sparse sets): member j belongs to
// centre seg[j] of the chunk; accumulator, counts and offsets carry a centre dimension (acc[centre][bin]); m differs per centre.
// touched (optional): one bit per group of 16 bins and centre, word w of a centre = groups 32 w .. 32 w + 31 (the grouped sweeps below)
__global__ void __launch_bounds__(256) k_sparse_scatter_batch(const uint2* __restrict__ ent, const MscSparseHdr* __restrict__ hdr,
                                                              const uint32_t* __restrict__ slots, const uint32_t* __restrict__ seg, uint32_t n_members,
                                                              uint64_t nbins, uint32_t* __restrict__ acc, uint32_t* __restrict__ touched) {
	const uint32_t j = blockIdx.x;
	if (j >= n_members) return;
	const MscSparseHdr h = hdr[slots[j]];
	uint32_t* a = acc + (uint64_t)seg[j] * nbins;
	uint32_t* tw = touched ? touched + (uint64_t)seg[j] * (nbins >> 9) : nullptr;
	for (uint32_t t = threadIdx.x; t < h.nnz; t += blockDim.x) {
		const uint2 e = ent[h.off + t];
		atomicAdd(&a[e.x], e.y - 1u);
		if (tw) {
			const uint32_t bit = 1u << ((e.x >> 4) & 31u);
			if (!(__builtin_nontemporal_load(&tw[e.x >> 9]) & bit)) atomicOr(&tw[e.x >> 9], bit);      // (most groups are hit by several members)
		}
	}
}

// (Both sweeps read four bins per lane and instruction: almost every group of 256 bins of an accumulator is zero -- a centre's
// neighbourhood touches a per cent or two of the 4^k bins --, so the common iteration is one 1 KiB load and a vote. With one bin per lane
// the two sweeps ran at 0.43 TB/s and were half of the device time of the update stage of a 200 000-sequence run.)
template <typename T>
__global__ void __launch_bounds__(64) k_sparse_mean_count_batch(const uint32_t* __restrict__ acc, uint64_t nbins, uint64_t chunk_bins,
                                                                const uint32_t* __restrict__ m_of, uint64_t* __restrict__ counts) {
	const uint32_t ci = blockIdx.y, n_chunks = gridDim.x;
	const uint32_t m = m_of[ci];
	const uint64_t base = (uint64_t)ci * nbins + (uint64_t)blockIdx.x * chunk_bins;
	uint64_t n = 0, ex = 0, fl = 0;
	if (m) {
		auto one = [&](uint32_t E) {
			if (E) {
				const MeanBin b = mean_bin<T>(E, m);
				if (b.r >= 2) { n++; ex += b.r - 1; }
				fl += b.fl - 1;
			}
		};
		if (chunk_bins % 256 == 0) {
			const uint4* a4 = reinterpret_cast<const uint4*>(acc + base);
			for (uint64_t i = threadIdx.x; i < chunk_bins / 4; i += 64) {
				const uint4 E = a4[i];
				if (E.x | E.y | E.z | E.w) { one(E.x); one(E.y); one(E.z); one(E.w); }
			}
		} else {
			for (uint64_t i = threadIdx.x; i < chunk_bins; i += 64) one(acc[base + i]);
		}
	}
	n = wave_sum_u64(n); ex = wave_sum_u64(ex); fl = wave_sum_u64(fl);
	if (threadIdx.x == 0) {
		uint64_t* o = counts + ((uint64_t)ci * n_chunks + blockIdx.x) * 3;
		o[0] = n; o[1] = ex; o[2] = fl;
	}
}

template <typename T>
__global__ void __launch_bounds__(64) k_sparse_mean_write_batch(uint32_t* __restrict__ acc, uint64_t nbins, uint64_t chunk_bins, const uint32_t* __restrict__ m_of,
                                                                const uint64_t* __restrict__ chunk_off, const uint64_t* __restrict__ chunk_cum,
                                                                uint2* __restrict__ ent, uint32_t* __restrict__ cum) {
	const uint32_t ci = blockIdx.y, n_chunks = gridDim.x;
	const uint32_t m = m_of[ci];
	if (m == 0) return;
	const uint64_t bin0 = (uint64_t)blockIdx.x * chunk_bins, base = (uint64_t)ci * nbins + bin0;
	uint64_t o = chunk_off[(uint64_t)ci * n_chunks + blockIdx.x];
	uint32_t run = (uint32_t)chunk_cum[(uint64_t)ci * n_chunks + blockIdx.x];
	const uint32_t lane = threadIdx.x;
	if (chunk_bins % 256 == 0) {
		// lane l holds bins i0 + 4 l .. + 3: entries leave in bin order = lane order, then the order inside the lane
		uint4* a4 = reinterpret_cast<uint4*>(acc + base);
		for (uint64_t i0 = 0; i0 < chunk_bins; i0 += 256) {
			const uint4 E4 = a4[i0 / 4 + lane];
			const bool any = (E4.x | E4.y | E4.z | E4.w) != 0;
			if (__ballot(any) == 0) continue;
			if (any) a4[i0 / 4 + lane] = make_uint4(0, 0, 0, 0);       // leave the accumulator clean for the next chunk of centres
			const uint32_t E[4] = {E4.x, E4.y, E4.z, E4.w};
			uint32_t r[4], cnt = 0, exs = 0;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				r[j] = E[j] ? mean_bin<T>(E[j], m).r : 1u;
				if (r[j] >= 2) { cnt++; exs += r[j] - 1; }
			}
			const uint32_t cnt_incl = wave_incl_scan(cnt), ex_incl = wave_incl_scan(exs);
			uint32_t at = cnt_incl - cnt, c = run + ex_incl - exs;
#pragma unroll
			for (int j = 0; j < 4; j++)
				if (r[j] >= 2) {
					c += r[j] - 1;
					ent[o + at] = make_uint2((uint32_t)(bin0 + i0 + 4 * lane + j), r[j]);
					cum[o + at] = c;
					at++;
				}
			o += (uint64_t)__builtin_amdgcn_readlane((int)cnt_incl, 63);
			run += (uint32_t)__builtin_amdgcn_readlane((int)ex_incl, 63);
		}
		return;
	}
	for (uint64_t i0 = 0; i0 < chunk_bins; i0 += 64) {
		const uint64_t i = i0 + lane;
		uint32_t E = 0;
		if (i < chunk_bins) { E = acc[base + i]; if (E) acc[base + i] = 0; }
		uint32_t r = 1;
		if (E) r = mean_bin<T>(E, m).r;
		const bool emit = r >= 2;
		const unsigned long long mask = __ballot(emit);
		const uint32_t ex = emit ? r - 1 : 0;
		const uint32_t ex_incl = wave_incl_scan(ex);
		if (emit) {
			const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
			ent[o + rank] = make_uint2((uint32_t)(bin0 + i), r);
			cum[o + rank] = run + ex_incl;
		}
		o += (uint64_t)__popcll(mask);
		run += (uint32_t)__builtin_amdgcn_readlane((int)ex_incl, 63);
	}
}

// ---- the two sweeps over TOUCHED groups only (large k). At k = 13 a centre's accumulator is 256 MiB and the ~1 M entries of its
// neighbourhood touch a few per cent of it; sweeping all of it twice per centre and round made the update stage of 20 kb sequences
// bandwidth-bound on zeros. The scatter marks every group of 16 bins (one 64-byte line) it touches; a wave expands 64 words of that
// bitmap into the ascending list of touched groups (LDS) and every lane then owns ONE group: its 16 bins are one line, the order of
// lanes is the order of bins. A chunk of bins is whole bitmap words (chunk_bins % 512 == 0: host-checked).
constexpr uint32_t kGroupList = 64 * 32;
__device__ __forceinline__ uint32_t expand_touched(const uint32_t* __restrict__ words, uint32_t w0, uint32_t nw, uint32_t lane, uint32_t* s_list) {
	uint32_t word = lane < nw ? words[w0 + lane] : 0u;
	const uint32_t cnt = (uint32_t)__popc(word);
	const uint32_t incl = wave_incl_scan(cnt);
	uint32_t off = incl - cnt;
	while (word) {
		const uint32_t b = (uint32_t)__ffs((int)word) - 1u;
		word &= word - 1u;
		s_list[off++] = (w0 + lane) * 32u + b;
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
	return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}

template <typename T>
__global__ void __launch_bounds__(64) k_sparse_mean_count_groups(const uint32_t* __restrict__ acc, const uint32_t* __restrict__ touched, uint64_t nbins,
                                                                 uint64_t chunk_bins, const uint32_t* __restrict__ m_of, uint64_t* __restrict__ counts) {
	__shared__ uint32_t s_list[kGroupList];
	const uint32_t ci = blockIdx.y, n_chunks = gridDim.x, lane = threadIdx.x;
	const uint32_t m = m_of[ci];
	const uint32_t* a = acc + (uint64_t)ci * nbins;
	const uint32_t* words = touched + (uint64_t)ci * (nbins >> 9);
	const uint32_t w_lo = (uint32_t)(((uint64_t)blockIdx.x * chunk_bins) >> 9), w_hi = (uint32_t)(((uint64_t)(blockIdx.x + 1) * chunk_bins) >> 9);
	uint64_t n = 0, ex = 0, fl = 0;
	if (m) {
		for (uint32_t w0 = w_lo; w0 < w_hi; w0 += 64) {
			const uint32_t T_ = expand_touched(words, w0, w_hi - w0 < 64 ? w_hi - w0 : 64, lane, s_list);
			for (uint32_t g0 = 0; g0 < T_; g0 += 64) {
				if (g0 + lane < T_) {
					const uint4* p = reinterpret_cast<const uint4*>(a + (uint64_t)s_list[g0 + lane] * 16);
					uint32_t E[16];
#pragma unroll
					for (int q = 0; q < 4; q++) { const uint4 v = p[q]; E[4 * q] = v.x; E[4 * q + 1] = v.y; E[4 * q + 2] = v.z; E[4 * q + 3] = v.w; }
#pragma unroll
					for (int q = 0; q < 16; q++)
						if (E[q]) {
							const MeanBin b = mean_bin<T>(E[q], m);
							if (b.r >= 2) { n++; ex += b.r - 1; }
							fl += b.fl - 1;
						}
				}
			}
			__builtin_amdgcn_wave_barrier();          // the list is rewritten by the next window
		}
	}
	n = wave_sum_u64(n); ex = wave_sum_u64(ex); fl = wave_sum_u64(fl);
	if (lane == 0) {
		uint64_t* o = counts + ((uint64_t)ci * n_chunks + blockIdx.x) * 3;
		o[0] = n; o[1] = ex; o[2] = fl;
	}
}

template <typename T>
__global__ void __launch_bounds__(64) k_sparse_mean_write_groups(uint32_t* __restrict__ acc, uint32_t* __restrict__ touched, uint64_t nbins, uint64_t chunk_bins,
                                                                 const uint32_t* __restrict__ m_of, const uint64_t* __restrict__ chunk_off,
                                                                 const uint64_t* __restrict__ chunk_cum, uint2* __restrict__ ent, uint32_t* __restrict__ cum) {
	__shared__ uint32_t s_list[kGroupList];
	const uint32_t ci = blockIdx.y, n_chunks = gridDim.x, lane = threadIdx.x;
	const uint32_t m = m_of[ci];
	if (m == 0) return;
	uint32_t* a = acc + (uint64_t)ci * nbins;
	uint32_t* words = touched + (uint64_t)ci * (nbins >> 9);
	const uint32_t w_lo = (uint32_t)(((uint64_t)blockIdx.x * chunk_bins) >> 9), w_hi = (uint32_t)(((uint64_t)(blockIdx.x + 1) * chunk_bins) >> 9);
	uint64_t o = chunk_off[(uint64_t)ci * n_chunks + blockIdx.x];
	uint32_t run = (uint32_t)chunk_cum[(uint64_t)ci * n_chunks + blockIdx.x];
	for (uint32_t w0 = w_lo; w0 < w_hi; w0 += 64) {
		const uint32_t nw = w_hi - w0 < 64 ? w_hi - w0 : 64;
		const uint32_t T_ = expand_touched(words, w0, nw, lane, s_list);
		for (uint32_t g0 = 0; g0 < T_; g0 += 64) {
			uint32_t r[16], n_emit = 0, ex_sum = 0, grp = 0;
			const bool have = g0 + lane < T_;
			if (have) {
				grp = s_list[g0 + lane];
				uint4* p = reinterpret_cast<uint4*>(a + (uint64_t)grp * 16);
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const uint4 v = p[q];
					r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
					p[q] = make_uint4(0u, 0u, 0u, 0u);          // leave the accumulator clean for the next chunk of centres
				}
#pragma unroll
				for (int q = 0; q < 16; q++) {
					r[q] = r[q] ? mean_bin<T>(r[q], m).r : 1u;
					if (r[q] >= 2) { n_emit++; ex_sum += r[q] - 1u; }
				}
			}
			const uint32_t e_incl = wave_incl_scan(n_emit), x_incl = wave_incl_scan(ex_sum);
			if (n_emit) {
				uint64_t at = o + (e_incl - n_emit);
				uint32_t c_run = run + (x_incl - ex_sum);
#pragma unroll
				for (int q = 0; q < 16; q++)
					if (r[q] >= 2) {
						c_run += r[q] - 1u;
						ent[at] = make_uint2(grp * 16u + (uint32_t)q, r[q]);
						cum[at] = c_run;
						at++;
					}
			}
			o += (uint64_t)__builtin_amdgcn_readlane((int)e_incl, 63);
			run += (uint32_t)__builtin_amdgcn_readlane((int)x_incl, 63);
		}
		__builtin_amdgcn_wave_barrier();
		if (lane < nw && words[w0 + lane]) words[w0 + lane] = 0u;          // ... and the bitmap
	}
}

// ------------------------------------------------------------------------------------------------ LDS-staged merge kernel
// k_pair_sparse is bound by the texture-address unit: 64 lanes chase 64 private pointers, ~30 cache lines per wave load.
// Here a WAVE owns one candidate: its entry list is staged into LDS with coalesced loads (the query list once per
// workgroup), lane r merges the index sub-range [r, r+1) * 4^k / 64 out of LDS, and the 64 partial results are folded in
// the wave, so one record per candidate reaches the epilogue. 32-bit running values (counts < 2^16, sums < 2^31: the host
// checks and otherwise keeps the general kernel). Lists longer than the LDS capacity also stay on the general kernel.
__device__ __forceinline__ uint32_t lower_bound_lds(const uint2* l, uint32_t n, uint32_t key) {
	uint32_t lo = 0, hi = n;
	while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (l[mid].x < key) lo = mid + 1; else hi = mid; }
	return lo;
}

template <bool DIV>
__global__ void __launch_bounds__(256) k_pair_sparse_lds(
    const uint2* __restrict__ c_ent, const uint32_t* __restrict__ c_cum, const MscSparseHdr* __restrict__ c_hdr,
    const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride, const uint32_t* __restrict__ cand_slots, uint32_t m,
    const uint2* __restrict__ q_ent, const uint32_t* __restrict__ q_cum, const MscSparseHdr* __restrict__ q_hdr_p,
    const uint8_t* __restrict__ q_scalars, uint64_t nbins, uint32_t qcap, uint32_t ccap, int use_window, uint64_t min_len, uint64_t max_len,
    MscPartial* __restrict__ partials, const DivTerm* __restrict__ div_tables, double* __restrict__ div_partials, int order) {
	extern __shared__ uint2 s_lists[];                  // [qcap] query entries | 4 x [ccap] candidate entries
	__shared__ uint32_t s_qsplit[65];
	__shared__ uint32_t s_qcum0[64];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const MscSparseHdr qh = *q_hdr_p;
	uint2* qlist = s_lists;
	uint2* clist = s_lists + qcap + (size_t)wave * ccap;
	for (uint32_t t = threadIdx.x; t < qh.nnz; t += blockDim.x) qlist[t] = q_ent[qh.off + t];
	__syncthreads();
	const uint32_t sub = (uint32_t)(nbins / 64);
	if (threadIdx.x <= 64) {
		const uint32_t pos = threadIdx.x == 64 ? qh.nnz : lower_bound_lds(qlist, qh.nnz, threadIdx.x * sub);
		s_qsplit[threadIdx.x] = pos;
		if (threadIdx.x < 64) s_qcum0[threadIdx.x] = pos ? q_cum[qh.off + pos - 1] : 0u;
	}
	__syncthreads();
	const uint32_t j0 = s_qsplit[lane], j1 = s_qsplit[lane + 1];
	const uint32_t cq0 = s_qcum0[lane];
	const double qm = DIV ? (double)reinterpret_cast<const MscSlotScalars*>(q_scalars)->mag : 0.0;
	const uint32_t total_waves = gridDim.x * (blockDim.x >> 6);
	for (uint32_t c = blockIdx.x * (blockDim.x >> 6) + wave; c < m; c += total_waves) {
		const uint32_t slot = cand_slots ? cand_slots[c] : c;
		const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride);
		if (use_window && (cs->length < min_len || cs->length > max_len)) continue;
		const MscSparseHdr ch = c_hdr[slot];
		__builtin_amdgcn_wave_barrier();                 // every lane is done reading the previous candidate's list
		for (uint32_t t = lane; t < ch.nnz; t += 64) clist[t] = c_ent[ch.off + t];       // coalesced: 512 B per wave load
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();                 // LDS operations of one wave complete in order: the reads below see the writes
		const uint32_t i0 = lower_bound_lds(clist, ch.nnz, lane * sub);
		uint32_t i1 = __shfl_down(i0, 1, 64);
		if (lane == 63) i1 = ch.nnz;
		uint32_t i = i0, j = j0;
		int32_t D = (int32_t)(i ? c_cum[ch.off + i - 1] : 0u) - (int32_t)cq0;           // prefix difference entering the sub-range
		uint32_t pos = lane * sub, manh = 0;
		uint64_t dotx = 0, emd = 0;
		double jd = 0.0, js = 0.0, cm = 0.0;
		DivTerm t11{0.0, 0.0};
		if constexpr (DIV) { cm = (double)cs->mag; t11 = div_term_sp(1, 1, cm, qm, order); }
		const uint32_t kInf = 0xffffffffu;
		uint2 a = i < i1 ? clist[i] : make_uint2(kInf, 1u);
		uint2 b = j < j1 ? qlist[j] : make_uint2(kInf, 1u);
		while (i < i1 || j < j1) {
			const uint32_t e = a.x < b.x ? a.x : b.x;
			const bool ta = a.x == e, tb = b.x == e;
			const uint32_t absD = (uint32_t)(D < 0 ? -D : D);
			emd += (uint64_t)absD * (e - pos);
			const uint32_t pv = ta ? a.y : 1u, qv = tb ? b.y : 1u;
			manh += pv > qv ? pv - qv : qv - pv;
			dotx += (uint64_t)(pv * qv - 1u);               // counts < 2^16: the product fits 32 bits
			D += (int32_t)pv - (int32_t)qv;
			if constexpr (DIV) {
				DivTerm tt;
				if ((pv | qv) < 16u) tt = div_tables[(uint64_t)c * 256 + pv * 16 + qv];
				else tt = div_term_sp(pv, qv, cm, qm, order);
				jd += tt.jd - t11.jd;
				js += tt.js - t11.js;
			}
			pos = e;
			if (ta) { i++; a = i < i1 ? clist[i] : make_uint2(kInf, 1u); }
			if (tb) { j++; b = j < j1 ? qlist[j] : make_uint2(kInf, 1u); }
		}
		{
			const uint32_t absD = (uint32_t)(D < 0 ? -D : D);
			emd += (uint64_t)absD * ((lane + 1) * sub - pos);
		}
		const uint64_t manh_t = wave_sum_u64(manh), dot_t = wave_sum_u64(dotx), emd_t = wave_sum_u64(emd);
		if constexpr (DIV) {
#pragma unroll
			for (int off = 32; off >= 1; off >>= 1) { jd += __shfl_xor(jd, off, 64); js += __shfl_xor(js, off, 64); }
		}
		if (lane == 0) {
			MscPartial out;
			out.manh = manh_t; out.dot = dot_t; out.emd = emd_t;
			partials[c] = out;
			if constexpr (DIV) { div_partials[2ull * c] = jd; div_partials[2ull * c + 1] = js; }
		}
	}
}

// ------------------------------------------------------------------------------------------------ merge-path kernel (long lists)
// Lists that do not fit LDS whole (20 kb sequences at k = 13: ~20 000 entries, 160 KB each). A WAVE owns one candidate and
// walks the MERGED order of the two lists in chunks of kMpT entries: the chunk boundaries are co-ranks on the merge path
// (how many candidate entries the first t * kMpT merged entries hold; 64 lanes search the next 63 boundaries at once in global memory),
// each chunk's two pieces are staged into wave-private LDS with coalesced loads, and a second co-rank search in LDS gives
// every lane an equal share of the chunk -- balanced whatever the index distribution, which equal index sub-ranges are not.
// A lane enters its share with the prefix difference and the last event position of everything before it (stored cum arrays,
// the predecessor entries staged with the chunk), so no cross-lane dependency exists; ties (the same bin in both lists) are
// never split across a boundary. One record per candidate, 32-bit running values as in the LDS kernel (host-checked).
// kMpT entries of the merged order per chunk; LDS per wave: predecessor + candidate piece, predecessor + query piece (the pieces
// sum to <= kMpT + 1) = kMpT + 8 entries

// co-rank of diagonal d: how many of the first d merged entries come from list C (C first on equal indices); cx(i) / qx(j) read
// the bin index of entry i / j. Afterwards a tie that the diagonal would split is pulled into the earlier side.
template <typename CX, typename QX>
__device__ __forceinline__ void mp_split(uint32_t d, uint32_t nc, uint32_t nq, CX cx, QX qx, uint32_t& i_out, uint32_t& j_out) {
	uint32_t lo = d > nq ? d - nq : 0, hi = d < nc ? d : nc;
	while (lo < hi) {
		const uint32_t i = (lo + hi) >> 1;
		if (cx(i) <= qx(d - i - 1)) lo = i + 1; else hi = i;
	}
	uint32_t j = d - lo;
	if (lo > 0 && j < nq && cx(lo - 1) == qx(j)) j++;
	i_out = lo;
	j_out = j;
}

// PAIRS (the batched update stage on sparse sets, msc_update_centres / msc_merge_all): every candidate has its OWN query -- slot
// segs[pair_seg[c]].q_slot behind q_hdr_p, with that segment's length window -- instead of the one query of a 1 x M pass.
// `count` 8-byte entries from global memory (src: wave-uniform, 8-byte aligned) straight into LDS at byte address lds_dst, two per lane
// and instruction (global_load_lds_dwordx4: no destination registers, no load -> store round trip through the VALU); an odd count
// copies one entry more. The caller waits (vmcnt) before it reads.
__device__ __forceinline__ void mp_stage_dma(const uint2* src, uint32_t count, uint32_t lds_dst, uint32_t lane) {
	// (wave-uniform values the compiler cannot prove uniform: through readfirstlane into scalar registers)
	const uint64_t sbase = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)src) |
	                       ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)src >> 32)) << 32);
	const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst);
	count = (uint32_t)__builtin_amdgcn_readfirstlane((int)count);
	for (uint32_t t = 0; t * 128 < count; t++) {
		if (t * 128 + lane * 2 < count) {
			const uint32_t off = t * 2048 + lane * 16, dst = lds0 + t * 2048;
			uint32_t keep;
			asm volatile(
			    "s_mov_b32 %0, m0\n\t"
			    "s_mov_b32 m0, %3\n\t"
			    "s_nop 0\n\t"
			    "global_load_lds_dwordx4 %1, %2\n\t"
			    "s_mov_b32 m0, %0"
			    : "=&s"(keep)
			    : "v"(off), "s"(sbase), "s"(dst)
			    : "memory");
		}
	}
}

// A piece and its predecessor (entries first - 1 .. first + n - 1 of a list; a neutral entry when first == 0) into the stage dst[0 .. n].
// pairs: two entries per lane and instruction -- one 16-byte load (list entries are 8-byte aligned; the hardware asks a multi-dword
// global load for dword alignment only) and one ds_write_b128 (dst is 16-byte aligned) instead of two of each; the lane that takes the
// last entry may copy one entry more (the arena keeps two entries of slack; the caller writes its end marker afterwards).
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ void stage_pairs(const uint2* __restrict__ list, uint32_t first, uint32_t n, uint2* dst, uint32_t lane, bool pairs) {
	if (!pairs || first == 0) {
		for (uint32_t k = lane; k <= n; k += 64) dst[k] = (first + k) ? list[first + k - 1] : make_uint2(0u, 1u);
		return;
	}
	const uint2* src = list + first - 1;
	for (uint32_t k = 2 * lane; k <= n; k += 128) {
		const u32x4_a8 v = *reinterpret_cast<const u32x4_a8*>(src + k);
		*reinterpret_cast<u32x4*>(dst + k) = u32x4{v.x, v.y, v.z, v.w};
	}
}

constexpr uint32_t kMpDivGran = 4;      // chunks per divergence record (see the DIV comment inside the kernel)
constexpr uint32_t kMpTab = 8;          // side of the divergence-term table in LDS
constexpr uint32_t kMpDivMinGran = 4;   // granules a part of a pair keeps at least (16 chunks, 8 192 merged entries)
template <bool DIV, uint32_t kMpT, bool PAIRS = false, int WPE = 1>
__global__ void __launch_bounds__(256, WPE) k_pair_sparse_mp(
    const uint2* __restrict__ c_ent, const uint32_t* __restrict__ c_cum, const MscSparseHdr* __restrict__ c_hdr,
    const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride, const uint32_t* __restrict__ cand_slots, uint32_t m,
    const uint2* __restrict__ q_ent, const uint32_t* __restrict__ q_cum, const MscSparseHdr* __restrict__ q_hdr_p,
    const uint8_t* __restrict__ q_scalars, uint64_t nbins, int use_window, uint64_t min_len, uint64_t max_len,
    MscPartial* __restrict__ partials, const DivTerm* __restrict__ div_tables, double* __restrict__ div_partials, int order,
    const MscBatchSeg* __restrict__ segs = nullptr, const uint32_t* __restrict__ pair_seg = nullptr, uint32_t parts = 1, uint64_t q_scalar_stride = 0,
    uint32_t div_stride = 1, bool dma = false, bool pairs = false) {
	constexpr uint32_t kMpBuf = (kMpT + 9) & ~1u;      // (a wave's stage starts on a 16-byte boundary: the staging writes two entries at a time)
	__shared__ uint2 s_buf[4][kMpBuf];
	// DIV: the 8 x 8 corner of the candidate's table of terms (counts below 8: all but the k-mers of repeats), already relative to the
	// (1, 1) term, in wave-private LDS -- the walk looks one entry up per event, and a lookup in global memory (a dependent L2 round
	// trip per step of a serial walk) was what bounded this form (r03: 124 G merged entries/s against 415 G for the integer form).
	// 1 KiB per wave keeps 6-7 workgroups per CU; WPE = the waves per SIMD the register allocator is held to (the direct evaluation
	// of a large count -- three FP64 logs -- is what inflates the kernel to 106 VGPRs; at 80 it spills a few values around that
	// cold path instead)
	__shared__ DivTerm s_tab[DIV ? 4 : 1][DIV ? kMpTab * kMpTab : 1];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint2* buf = s_buf[wave];
	MscSparseHdr qh = PAIRS ? MscSparseHdr{} : *q_hdr_p;
	const uint2* Q = q_ent + qh.off;
	const uint32_t* CQ = q_cum + qh.off;
	uint32_t nq_all = qh.nnz;
	double qm = DIV && !PAIRS ? (double)reinterpret_cast<const MscSlotScalars*>(q_scalars)->mag : 0.0;
	const uint32_t total_waves = gridDim.x * (blockDim.x >> 6);
	const uint32_t kInf = 0xffffffffu;
	// `parts` waves share a candidate (a short window would leave most of the chip idle, and a get_close step is as slow as its
	// slowest wave): part p walks chunks [p n / parts, (p + 1) n / parts) of the merged order and writes record c * parts + p
	// the records of a pair the walk does not reach (and all of them for a candidate outside the length window) read as zero: written
	// here by the pair's first part -- a memset of the whole array ahead of every pass was one more launch per step
	auto zero_records = [&](uint32_t c, uint32_t from) {
		if constexpr (DIV) {
			double* rec = div_partials + 2ull * (uint64_t)c * div_stride;
			for (uint32_t g = from + lane; g < div_stride; g += 64) { rec[2 * g] = 0.0; rec[2 * g + 1] = 0.0; }
		}
	};
	for (uint32_t w = blockIdx.x * (blockDim.x >> 6) + wave; w < m * parts; w += total_waves) {
		const uint32_t c = w / parts, part = w - c * parts;
		const uint32_t slot = cand_slots ? cand_slots[c] : c;
		const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride);
		if constexpr (PAIRS) {
			const MscBatchSeg sg = segs[pair_seg[c]];
			if (use_window && (cs->length < sg.min_len || cs->length > sg.max_len)) { if (part == 0) zero_records(c, 0); continue; }
			qh = q_hdr_p[sg.q_slot];
			Q = q_ent + qh.off;
			CQ = q_cum + qh.off;
			nq_all = qh.nnz;
			if constexpr (DIV) qm = (double)reinterpret_cast<const MscSlotScalars*>(q_scalars + (uint64_t)sg.q_slot * q_scalar_stride)->mag;
		} else {
			if (use_window && (cs->length < min_len || cs->length > max_len)) { if (part == 0) zero_records(c, 0); continue; }
		}
		const MscSparseHdr ch = c_hdr[slot];
		const uint2* P = c_ent + ch.off;
		const uint32_t* CP = c_cum + ch.off;
		const uint32_t nc_all = ch.nnz;
		const uint32_t total = nc_all + nq_all;
		const uint32_t n_chunks = (total + kMpT - 1) / kMpT;
		uint32_t manh = 0;
		uint64_t dotx = 0, emd = 0;
		double jd = 0.0, js = 0.0, cm = 0.0;
		DivTerm t11{0.0, 0.0};
		if constexpr (DIV) {
			cm = (double)cs->mag; t11 = div_term_call(1, 1, cm, qm, order);
			__builtin_amdgcn_wave_barrier();          // the previous candidate's walk is over
			static_assert(kMpTab * kMpTab == 64, "one table entry per lane");
			{
				// (computed here, one term per lane: a table kernel of its own ahead of every pass was 10 us of a 500 us step)
				const uint32_t a = lane / kMpTab, b = lane % kMpTab;
				DivTerm g{0.0, 0.0};
				if (a && b) g = div_term_call(a, b, cm, qm, order);
				s_tab[wave][lane] = DivTerm{g.jd - t11.jd, g.js - t11.js};
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
		uint32_t ci = 0, qj = 0, dchunk = 0;
		// DIV: the two FP64 sums leave the wave once per GRANULE of kMpDivGran chunks, as record [c][granule] of div_partials, and the
		// epilogue adds a pair's records in granule order -- so the value of a pair depends on the pair alone, not on how many waves
		// shared it (parts cut the merged order between granules) nor on the route that scored it
		uint32_t t_begin, t_end;
		if constexpr (DIV) {
			// `parts` is the most a pair of this launch may be cut into; THIS pair takes as many as it has stretches of kMpDivMinGran
			// granules (a window of mixed lengths: the long pairs are shared out, the short ones stay whole) -- the other parts of its
			// record group stay empty
			const uint32_t n_gran = (n_chunks + kMpDivGran - 1) / kMpDivGran;
			if (part == 0) zero_records(c, n_gran);
			uint32_t parts_c = n_gran / kMpDivMinGran;
			parts_c = parts_c < 1 ? 1 : parts_c > parts ? parts : parts_c;
			if (part >= parts_c) {
				if (lane == 0) partials[w] = MscPartial{0, 0, 0};
				continue;
			}
			t_begin = (uint32_t)((uint64_t)n_gran * part / parts_c) * kMpDivGran;
			t_end = (uint32_t)((uint64_t)n_gran * (part + 1) / parts_c) * kMpDivGran;
			if (t_end > n_chunks) t_end = n_chunks;
		} else {
			t_begin = (uint32_t)((uint64_t)n_chunks * part / parts);
			t_end = (uint32_t)((uint64_t)n_chunks * (part + 1) / parts);
		}
		for (uint32_t t = t_begin; t < t_end; t++) {
			const uint32_t tl = (t - t_begin) % 63;
			if (tl == 0) {                       // boundaries of the next 63 chunks: lane l searches diagonal (t + l) * kMpT in global memory
				const uint64_t dd = (uint64_t)(t + lane) * kMpT;
				const uint32_t d = dd < total ? (uint32_t)dd : total;
				mp_split(d, nc_all, nq_all, [&](uint32_t i) { return P[i].x; }, [&](uint32_t j) { return Q[j].x; }, ci, qj);
				dchunk = (ci ? CP[ci - 1] : 0u) - (qj ? CQ[qj - 1] : 0u);      // prefix difference entering that chunk (two's complement)
			}
			// the chunk's boundaries are wave-uniform: scalar registers
			const uint32_t c0 = __builtin_amdgcn_readlane(ci, tl), c1 = __builtin_amdgcn_readlane(ci, tl + 1);
			const uint32_t q0 = __builtin_amdgcn_readlane(qj, tl), q1 = __builtin_amdgcn_readlane(qj, tl + 1);
			const uint32_t nc = c1 - c0, nq = q1 - q0;
			uint2* cl = buf;                     // cl[0] = predecessor of the piece (or a neutral entry), cl[1 + k] = P[c0 + k]
			uint2* ql = buf + ((nc + 3) & ~1u);  // (behind cl[0 .. nc + 1], on a 16-byte boundary like buf: the DMA writes 16 bytes per lane)
			__builtin_amdgcn_wave_barrier();     // every lane is done with the previous chunk's entries
			// (a plain load / store loop per piece: the kernel is bound by vector-instruction issue at 8 waves per SIMD, which hide the
			// latency of these loads; staging through registers with every load of the chunk in flight, a chunk ahead, shortens a lone
			// wave's chunk by 20 % but costs more instructions and two waves per SIMD -- 12 % slower once the chip is full)
			// LDS-DMA: a piece and its predecessor entry go from the list to the stage without passing through registers (r02: the load ->
			// store loops were 37 % of this kernel at k = 13). A piece that STARTS its list has no predecessor in memory (and its entries
			// would land 8 bytes off the DMA's 16-byte grid): the first chunk of either list keeps the loop.
			if (dma && c0) mp_stage_dma(P + c0 - 1, nc + 1, (uint32_t)(uintptr_t)cl, lane);
			else stage_pairs(P, c0, nc, cl, lane, pairs);
			if (dma && q0) mp_stage_dma(Q + q0 - 1, nq + 1, (uint32_t)(uintptr_t)ql, lane);
			else stage_pairs(Q, q0, nq, ql, lane, pairs);
			if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			// one entry past either piece: a bin index no share reaches, so a lane that runs off the end of a piece stops by itself
			if (lane == 0) { cl[nc + 1] = make_uint2(kInf, 1u); ql[nq + 1] = make_uint2(kInf, 1u); }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__builtin_amdgcn_wave_barrier();     // LDS operations of one wave complete in order: the reads below see the writes
			// equal shares of `seg` merged entries; seg is kept ODD: lane l starts near byte 8 * seg * l / 2 of either piece, and an even
			// seg (8 for a full chunk) lines the 64 lanes up on a quarter of the LDS banks (bank-conflict cycles were 2.4x the active ones)
			const uint32_t n = nc + nq, seg = ((n + 63) >> 6) | 1u;
			uint32_t i, j;
			{
				const uint32_t d = lane * seg < n ? lane * seg : n;
				mp_split(d, nc, nq, [&](uint32_t a) { return cl[a + 1].x; }, [&](uint32_t b) { return ql[b + 1].x; }, i, j);
			}
			uint32_t i1 = __shfl_down(i, 1, 64), j1 = __shfl_down(j, 1, 64);
			if (lane == 63) { i1 = nc; j1 = nq; }
			// prefix difference entering each share = the chunk's + what the earlier lanes' shares add: a pass over the share's values
			// and one wave scan instead of two scattered reads of the cum arrays per lane
			uint32_t delta = (j1 - j) - (i1 - i);                              // sum of (value - 1) = sum of values - count
			for (uint32_t a = i; a < i1; a++) delta += cl[a + 1].y;
			for (uint32_t b = j; b < j1; b++) delta -= ql[b + 1].y;
			const uint32_t d_in = __builtin_amdgcn_readlane(dchunk, tl) + wave_incl_scan(delta) - delta;
			// The share: every bin of it lies below the first bin of the next lane's share (the merged order is split between bins,
			// ties kept together), so the walk needs no per-list bound -- it reloads both heads every step (an entry of the next share,
			// or the end marker, simply never becomes the minimum) and stops at that bin.
			// both heads with one ds_read_b64 each (the compiler reads .x and .y separately, and the LDS pipe is the busiest unit here):
			// pa / pb = LDS byte addresses of the heads; the wait ties the two results so nothing uses them early
			auto heads = [](uint32_t pa, uint32_t pb, uint2& a, uint2& b) {
				uint64_t wa, wb;
				asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(wa), "=&v"(wb) : "v"(pa), "v"(pb) : "memory");
				a = make_uint2((uint32_t)wa, (uint32_t)(wa >> 32));
				b = make_uint2((uint32_t)wb, (uint32_t)(wb >> 32));
			};
			uint32_t pa = (uint32_t)(uintptr_t)(cl + i + 1), pb = (uint32_t)(uintptr_t)(ql + j + 1);
			uint2 a, b;
			heads(pa, pb, a, b);
			uint32_t e = a.x < b.x ? a.x : b.x;
			uint32_t e_end = __shfl_down(e, 1, 64);
			if (lane == 63) e_end = kInf;                                      // (a lane behind the last share reads the end markers: its e is kInf)
			{
				const uint32_t pc = cl[i].x, pq = ql[j].x;                      // predecessors (neutral entries have index 0)
				uint32_t pos = pc > pq ? pc : pq;
				int32_t D = (int32_t)d_in;
				uint32_t events = 0;
				while (e < e_end) {
					const bool ta = a.x == e, tb = b.x == e;
					const uint32_t absD = (uint32_t)(D < 0 ? -D : D);
					emd += (uint64_t)absD * (e - pos);
					const uint32_t pv = ta ? a.y : 1u, qv = tb ? b.y : 1u;
					manh += pv > qv ? pv - qv : qv - pv;
					dotx += (uint64_t)pv * qv;
					events++;
					D += (int32_t)pv - (int32_t)qv;
					if constexpr (DIV) {
						DivTerm tt;
						if ((pv | qv) < kMpTab) tt = s_tab[wave][pv * kMpTab + qv];
						else { tt = div_term_call(pv, qv, cm, qm, order); tt.jd -= t11.jd; tt.js -= t11.js; }
						jd += tt.jd;
						js += tt.js;
					}
					pos = e;
					pa += ta ? 8u : 0u;
					pb += tb ? 8u : 0u;
					heads(pa, pb, a, b);
					e = a.x < b.x ? a.x : b.x;
				}
				dotx -= events;                                                 // sum of (p q - 1) over the events
			}
			if constexpr (DIV) {
				if ((t + 1) % kMpDivGran == 0 || t + 1 == t_end) {              // (t_begin is a multiple of the granule)
#pragma unroll
					for (int off = 32; off >= 1; off >>= 1) { jd += __shfl_xor(jd, off, 64); js += __shfl_xor(js, off, 64); }
					if (lane == 0) {
						double* rec = div_partials + 2ull * ((uint64_t)c * div_stride + t / kMpDivGran);
						rec[0] = jd; rec[1] = js;
					}
					jd = 0.0; js = 0.0;
				}
			}
		}
		if (lane == 0 && t_end == n_chunks && (t_begin < t_end || part == 0)) {       // the stretch behind the last event of either list: the part that walks the last chunk
			const uint32_t lc = nc_all ? P[nc_all - 1].x : 0u, lq = nq_all ? Q[nq_all - 1].x : 0u;
			const int64_t D = (int64_t)(nc_all ? CP[nc_all - 1] : 0u) - (int64_t)(nq_all ? CQ[nq_all - 1] : 0u);
			emd += (uint64_t)(D < 0 ? -D : D) * (nbins - (uint64_t)(lc > lq ? lc : lq));
		}
		const uint64_t manh_t = wave_sum_u64(manh), dot_t = wave_sum_u64(dotx), emd_t = wave_sum_u64(emd);
		if (lane == 0) {
			MscPartial out;
			out.manh = manh_t; out.dot = dot_t; out.emd = emd_t;
			partials[w] = out;
		}
	}
}

// ------------------------------------------------------------------------------------------------ whole-list kernel (short lists)
// 1 x M over lists short enough to sit in LDS whole (1 kb sequences: ~1 000 entries, 8 KB): the QUERY list is staged once per
// workgroup and shared by its four waves for every candidate they walk; a wave stages its candidate's list, takes ONE merge-path
// split of the two whole lists (k_pair_sparse_mp repeats staging of both pieces, the split and the entry sums per chunk of 512,
// behind a co-rank search in global memory) and every lane walks an equal share of the merged order with the end-marker walk of
// the chunked kernel. The prefix difference entering a share is two reads of the stored cum arrays. Integer statistics only: the
// divergence form stays with the chunked kernel, so its FP64 sums keep one evaluation order whatever the list lengths of a set.
// LDS: [query: nq + 2][wave 0: c_cap + 2] ... [wave 3: c_cap + 2] entries; entry 0 of a list = the neutral predecessor (bin 0, value
// 1), entry n + 1 = the end marker.
__global__ void __launch_bounds__(256) k_pair_sparse_wl(
    const uint2* __restrict__ c_ent, const uint32_t* __restrict__ c_cum, const MscSparseHdr* __restrict__ c_hdr,
    const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride, const uint32_t* __restrict__ cand_slots, uint32_t m,
    const uint2* __restrict__ q_ent, const uint32_t* __restrict__ q_cum, const MscSparseHdr* __restrict__ q_hdr_p, uint32_t c_cap,
    uint64_t nbins, int use_window, uint64_t min_len, uint64_t max_len, MscPartial* __restrict__ partials) {
	extern __shared__ __attribute__((aligned(16))) uint2 s_wl[];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t kInf = 0xffffffffu;
	const MscSparseHdr qh = *q_hdr_p;
	const uint32_t nq = qh.nnz;
	const uint2* Q = q_ent + qh.off;
	const uint32_t* CQ = q_cum + qh.off;
	uint2* ql = s_wl;                                      // ql[0] predecessor, ql[1 + j] = Q[j], ql[nq + 1] marker
	uint2* cl = s_wl + (nq + 2) + wave * (c_cap + 2);
	for (uint32_t j = threadIdx.x; j < nq; j += 256) ql[1 + j] = Q[j];
	if (threadIdx.x == 0) { ql[0] = make_uint2(0u, 1u); ql[nq + 1] = make_uint2(kInf, 1u); }
	__syncthreads();
	auto heads = [](uint32_t pa, uint32_t pb, uint2& a, uint2& b) {       // one ds_read_b64 per head (see k_pair_sparse_mp)
		uint64_t wa, wb;
		asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(wa), "=&v"(wb) : "v"(pa), "v"(pb) : "memory");
		a = make_uint2((uint32_t)wa, (uint32_t)(wa >> 32));
		b = make_uint2((uint32_t)wb, (uint32_t)(wb >> 32));
	};
	const uint32_t total_waves = gridDim.x * 4;
	for (uint32_t c = blockIdx.x * 4 + wave; c < m; c += total_waves) {
		const uint32_t slot = cand_slots ? cand_slots[c] : c;
		const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride);
		if (use_window && (cs->length < min_len || cs->length > max_len)) continue;
		const MscSparseHdr ch = c_hdr[slot];
		const uint2* P = c_ent + ch.off;
		const uint32_t* CP = c_cum + ch.off;
		const uint32_t nc = ch.nnz;
		__builtin_amdgcn_wave_barrier();                       // every lane is done with the previous candidate's entries
		for (uint32_t k0 = lane; k0 < nc; k0 += 256) {         // four loads in flight per lane
			uint2 v[4];
#pragma unroll
			for (uint32_t u = 0; u < 4; u++) v[u] = k0 + 64 * u < nc ? P[k0 + 64 * u] : make_uint2(kInf, 1u);
#pragma unroll
			for (uint32_t u = 0; u < 4; u++) if (k0 + 64 * u < nc) cl[1 + k0 + 64 * u] = v[u];
		}
		if (lane == 0) { cl[0] = make_uint2(0u, 1u); cl[nc + 1] = make_uint2(kInf, 1u); }
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
		const uint32_t n = nc + nq, seg = ((n + 63) >> 6) | 1u;             // odd share length: the lanes' heads spread over the LDS banks
		uint32_t i, j;
		{
			const uint32_t d = lane * seg < n ? lane * seg : n;
			mp_split(d, nc, nq, [&](uint32_t a) { return cl[a + 1].x; }, [&](uint32_t b) { return ql[b + 1].x; }, i, j);
		}
		// prefix difference entering the share: the stored inclusive sums of the excess counts before it
		const uint32_t d_in = (i ? CP[i - 1] : 0u) - (j ? CQ[j - 1] : 0u);
		uint32_t pa = (uint32_t)(uintptr_t)(cl + i + 1), pb = (uint32_t)(uintptr_t)(ql + j + 1);
		uint2 a, b;
		heads(pa, pb, a, b);
		uint32_t e = a.x < b.x ? a.x : b.x;
		uint32_t e_end = __shfl_down(e, 1, 64);
		if (lane == 63) e_end = kInf;
		uint32_t manh = 0;
		uint64_t dotx = 0, emd = 0;
		{
			const uint32_t pc = cl[i].x, pq = ql[j].x;                         // predecessors (the neutral entries have bin 0)
			uint32_t pos = pc > pq ? pc : pq;
			int32_t D = (int32_t)d_in;
			uint32_t events = 0;
			while (e < e_end) {
				const bool ta = a.x == e, tb = b.x == e;
				const uint32_t absD = (uint32_t)(D < 0 ? -D : D);
				emd += (uint64_t)absD * (e - pos);
				const uint32_t pv = ta ? a.y : 1u, qv = tb ? b.y : 1u;
				manh += pv > qv ? pv - qv : qv - pv;
				dotx += (uint64_t)pv * qv;
				events++;
				D += (int32_t)pv - (int32_t)qv;
				pos = e;
				pa += ta ? 8u : 0u;
				pb += tb ? 8u : 0u;
				heads(pa, pb, a, b);
				e = a.x < b.x ? a.x : b.x;
			}
			dotx -= events;
		}
		if (lane == 0) {      // the stretch behind the last event of either list
			const uint32_t lc = cl[nc].x, lq = ql[nq].x;                       // (bin 0 of the neutral entry when a list is empty)
			const int64_t D = (int64_t)(nc ? CP[nc - 1] : 0u) - (int64_t)(nq ? CQ[nq - 1] : 0u);
			emd += (uint64_t)(D < 0 ? -D : D) * (nbins - (uint64_t)(lc > lq ? lc : lq));
		}
		const uint64_t manh_t = wave_sum_u64(manh), dot_t = wave_sum_u64(dotx), emd_t = wave_sum_u64(emd);
		if (lane == 0) {
			MscPartial out;
			out.manh = manh_t; out.dot = dot_t; out.emd = emd_t;
			partials[c] = out;
		}
	}
}

// ================================================================================================ launchers
hipError_t msc_launch_sparse_count(hipStream_t st, const void* scratch_bins, const MscLayout& L, int dtype, uint32_t n, uint64_t* counts) {
	if (n == 0) return hipSuccess;
	switch (dtype) {
	case 8: k_sparse_count<uint8_t><<<dim3(n), dim3(kBlockC), 0, st>>>((const uint8_t*)scratch_bins, L.padded_bins, L.S, L.R, counts); break;
	case 16: k_sparse_count<uint16_t><<<dim3(n), dim3(kBlockC), 0, st>>>((const uint16_t*)scratch_bins, L.padded_bins, L.S, L.R, counts); break;
	case 32: k_sparse_count<uint32_t><<<dim3(n), dim3(kBlockC), 0, st>>>((const uint32_t*)scratch_bins, L.padded_bins, L.S, L.R, counts); break;
	default: k_sparse_count<uint64_t><<<dim3(n), dim3(kBlockC), 0, st>>>((const uint64_t*)scratch_bins, L.padded_bins, L.S, L.R, counts); break;
	}
	return hipGetLastError();
}

hipError_t msc_launch_sparse_write(hipStream_t st, const void* scratch_bins, const MscLayout& L, int dtype, uint32_t n, const MscSparseHdr* hdr,
                                   uint64_t first_slot, const uint64_t* cum_base, void* ent, uint32_t* cum) {
	if (n == 0) return hipSuccess;
	switch (dtype) {
	case 8: k_sparse_write<uint8_t><<<dim3(n), dim3(kBlockC), 0, st>>>((const uint8_t*)scratch_bins, L.padded_bins, L.S, L.R, hdr, first_slot, cum_base, (uint2*)ent, cum); break;
	case 16: k_sparse_write<uint16_t><<<dim3(n), dim3(kBlockC), 0, st>>>((const uint16_t*)scratch_bins, L.padded_bins, L.S, L.R, hdr, first_slot, cum_base, (uint2*)ent, cum); break;
	case 32: k_sparse_write<uint32_t><<<dim3(n), dim3(kBlockC), 0, st>>>((const uint32_t*)scratch_bins, L.padded_bins, L.S, L.R, hdr, first_slot, cum_base, (uint2*)ent, cum); break;
	default: k_sparse_write<uint64_t><<<dim3(n), dim3(kBlockC), 0, st>>>((const uint64_t*)scratch_bins, L.padded_bins, L.S, L.R, hdr, first_slot, cum_base, (uint2*)ent, cum); break;
	}
	return hipGetLastError();
}

hipError_t msc_launch_pair_sparse(hipStream_t st, const void* c_ent, const uint32_t* c_cum, const MscSparseHdr* c_hdr, const uint8_t* cand_scalars,
                                  uint64_t scalar_stride, const uint32_t* cand_slots, uint32_t m, const void* q_ent, const uint32_t* q_cum,
                                  const MscSparseHdr* q_hdr, const uint8_t* q_scalars, uint64_t nbins, int use_window, uint64_t min_len,
                                  uint64_t max_len, MscPartial* partials, void* div_tables, void* div_partials, int order) {
	if (m == 0) return hipSuccess;
	const uint64_t threads = (uint64_t)m * kSub;
	const unsigned blocks = (unsigned)((threads + 255) / 256);
	if (div_tables) {
		k_sparse_div_tables<<<dim3(m), dim3(256), 0, st>>>(cand_scalars, scalar_stride, cand_slots, m, q_scalars, order, (DivTerm*)div_tables);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
		k_pair_sparse<true><<<dim3(blocks), dim3(256), 0, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, (const uint2*)q_ent,
		                                                        q_cum, q_hdr, q_scalars, nbins, use_window, min_len, max_len, partials,
		                                                        (const DivTerm*)div_tables, (double*)div_partials, order);
	} else {
		k_pair_sparse<false><<<dim3(blocks), dim3(256), 0, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, (const uint2*)q_ent,
		                                                         q_cum, q_hdr, q_scalars, nbins, use_window, min_len, max_len, partials, nullptr,
		                                                         nullptr, order);
	}
	return hipGetLastError();
}

hipError_t msc_launch_sparse_scatter(hipStream_t st, const void* ent, const MscSparseHdr* hdr, const uint32_t* slots, uint32_t m, uint32_t* acc) {
	if (m == 0) return hipSuccess;
	const uint32_t parts = std::max<uint32_t>(1, std::min<uint32_t>(32, 512 / m));          // (about 512 workgroups: few members -> many parts each)
	k_sparse_scatter<<<dim3(m, parts), dim3(256), 0, st>>>((const uint2*)ent, hdr, slots, m, acc);
	return hipGetLastError();
}

hipError_t msc_launch_sparse_mean_count(hipStream_t st, int dtype, const uint32_t* acc, uint32_t n_chunks, uint64_t chunk_bins, uint32_t m, uint64_t* counts) {
	switch (dtype) {
	case 8: k_sparse_mean_count<uint8_t><<<dim3(n_chunks), dim3(64), 0, st>>>(acc, chunk_bins, m, counts); break;
	case 16: k_sparse_mean_count<uint16_t><<<dim3(n_chunks), dim3(64), 0, st>>>(acc, chunk_bins, m, counts); break;
	case 32: k_sparse_mean_count<uint32_t><<<dim3(n_chunks), dim3(64), 0, st>>>(acc, chunk_bins, m, counts); break;
	default: k_sparse_mean_count<uint64_t><<<dim3(n_chunks), dim3(64), 0, st>>>(acc, chunk_bins, m, counts); break;
	}
	return hipGetLastError();
}

hipError_t msc_launch_sparse_mean_write(hipStream_t st, int dtype, uint32_t* acc, uint32_t n_chunks, uint64_t chunk_bins, uint32_t m,
                                        const uint64_t* chunk_off, const uint64_t* chunk_cum, void* ent, uint32_t* cum) {
	switch (dtype) {
	case 8: k_sparse_mean_write<uint8_t><<<dim3(n_chunks), dim3(64), 0, st>>>(acc, chunk_bins, m, chunk_off, chunk_cum, (uint2*)ent, cum); break;
	case 16: k_sparse_mean_write<uint16_t><<<dim3(n_chunks), dim3(64), 0, st>>>(acc, chunk_bins, m, chunk_off, chunk_cum, (uint2*)ent, cum); break;
	case 32: k_sparse_mean_write<uint32_t><<<dim3(n_chunks), dim3(64), 0, st>>>(acc, chunk_bins, m, chunk_off, chunk_cum, (uint2*)ent, cum); break;
	default: k_sparse_mean_write<uint64_t><<<dim3(n_chunks), dim3(64), 0, st>>>(acc, chunk_bins, m, chunk_off, chunk_cum, (uint2*)ent, cum); break;
	}
	return hipGetLastError();
}

hipError_t msc_launch_sparse_build_sort(hipStream_t st, int k, int dtype, uint64_t nbins, uint64_t first_slot, uint32_t n_seqs, const uint32_t* packed,
                                        const uint64_t* seg_start, const uint64_t* kmer_off, const uint64_t* seq_seg_begin, const uint64_t* seq_arena_off,
                                        uint32_t P, uint8_t* scalars, uint64_t scalar_stride, MscSparseHdr* hdr, void* ent, uint32_t* cum) {
	if (n_seqs == 0) return hipSuccess;
	const size_t lds = (size_t)P * sizeof(uint32_t);
	static bool attr_set = false;
	if (!attr_set) {
		hipError_t e = hipFuncSetAttribute((const void*)k_sparse_build_sort, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4);
		if (e != hipSuccess) return e;
		attr_set = true;
	}
	k_sparse_build_sort<<<dim3(n_seqs), dim3(kSortBlock), lds, st>>>(k, dtype, nbins, first_slot, packed, seg_start, kmer_off, seq_seg_begin, seq_arena_off, P, scalars,
	                                                                   scalar_stride, hdr, (uint2*)ent, cum);
	return hipGetLastError();
}

// returns hipErrorInvalidValue when the lists do not fit the LDS budget (caller then uses msc_launch_pair_sparse)
hipError_t msc_launch_pair_sparse_lds(hipStream_t st, const void* c_ent, const uint32_t* c_cum, const MscSparseHdr* c_hdr, const uint8_t* cand_scalars,
                                      uint64_t scalar_stride, const uint32_t* cand_slots, uint32_t m, const void* q_ent, const uint32_t* q_cum,
                                      const MscSparseHdr* q_hdr, const uint8_t* q_scalars, uint64_t nbins, uint32_t q_nnz, uint32_t max_c_nnz, int use_window,
                                      uint64_t min_len, uint64_t max_len, MscPartial* partials, void* div_tables, void* div_partials, int order, int num_cus) {
	if (m == 0) return hipSuccess;
	const uint32_t qcap = (q_nnz + 63) / 64 * 64 + 64, ccap = (max_c_nnz + 63) / 64 * 64 + 64;
	const size_t lds = ((size_t)qcap + 4ull * ccap) * sizeof(uint2);
	if (lds > 96 * 1024 || nbins < 64) return hipErrorInvalidValue;
	static bool attr_set = false;
	if (!attr_set) {
		hipError_t e = hipFuncSetAttribute((const void*)k_pair_sparse_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
		if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_pair_sparse_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
		if (e != hipSuccess) return e;
		attr_set = true;
	}
	// resident workgroups: LDS-limited (160 KiB per CU), at most 8 per CU; each wave then walks several candidates
	uint32_t per_cu = (uint32_t)std::min<size_t>(8, (160 * 1024) / (lds + 1024));
	if (per_cu < 1) per_cu = 1;
	uint32_t blocks = (uint32_t)num_cus * per_cu;
	if (blocks > (m + 3) / 4) blocks = (m + 3) / 4;
	if (div_tables) {
		k_sparse_div_tables<<<dim3(m), dim3(256), 0, st>>>(cand_scalars, scalar_stride, cand_slots, m, q_scalars, order, (DivTerm*)div_tables);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
		k_pair_sparse_lds<true><<<dim3(blocks), dim3(256), lds, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, (const uint2*)q_ent,
		                                                               q_cum, q_hdr, q_scalars, nbins, qcap, ccap, use_window, min_len, max_len, partials,
		                                                               (const DivTerm*)div_tables, (double*)div_partials, order);
	} else {
		k_pair_sparse_lds<false><<<dim3(blocks), dim3(256), lds, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, (const uint2*)q_ent,
		                                                                q_cum, q_hdr, q_scalars, nbins, qcap, ccap, use_window, min_len, max_len, partials, nullptr,
		                                                                nullptr, order);
	}
	return hipGetLastError();
}

int msc_sparse_div_waves();
bool msc_sparse_mp_pairs();
// MSC_SPARSE_MP_DMA=1: the chunks of the merge-path kernel are staged by LDS-DMA instead of the load -> store loop. OFF by default:
// measured r03 (k = 13, 20 kb lists, 8 000 candidates) 2.89 ms per launch against 0.78 ms for the loop, and results that differ -- list
// pieces start on 8-byte, not 16-byte, boundaries of global memory, which global_load_lds_dwordx4 does not take (profiles/r03_notes.md).
// two list entries per lane and instruction when a chunk is staged (MSC_SPARSE_MP_NO_PAIRS for A/B runs)
bool msc_sparse_mp_pairs() {
	static const bool off = getenv("MSC_SPARSE_MP_NO_PAIRS") != nullptr;
	return !off;
}
bool msc_sparse_mp_dma() {
	static const bool on = getenv("MSC_SPARSE_MP_DMA") != nullptr;
	return on;
}
uint32_t msc_sparse_mp_max_entries() { return 0x7fffffffu; }      // both lists together (32-bit merged positions)

// 512 merged entries per chunk keep 8 waves per SIMD resident, which is what this merge wants (r01: k=9/5 kb lists 43 M pairs/s at 512,
// 32 M at 1024, 20 M at 2048). A chunk is walked in 64 shares of an ODD number of entries (see the kernel): 9 for 512, which leaves 7
// lanes without a share. 575 gives every lane its 9 (575, not 576: a chunk holds one entry more when a tie straddles its end, and 577
// would make the share 11). r03, same box: cfg5's accumulate stage -5 % at 20 000 sequences and -7 % at full size (get_close 46.1 ->
// 42.9 s); but a window of 8 000 EQUAL lists (one and a third fills of the grid, k = 9, 11 or 13 alike) takes 0.87 ms instead of
// 0.67 -- and so does the 512 build as soon as the records are dealt to the waves in any other order, so it is the placement of
// that window's second helping, not the chunk, that the 0.67 depends on (profiles/r03_notes.md). Until that is understood the wide
// chunk is for the sets it was measured to help, k <= 11; a set's chunk never changes, so a pair's FP64 sums do not either.
constexpr uint32_t kMpChunk = 512, kMpChunkWide = 575;
static bool mp_wide(uint64_t nbins) {
	static const int force = [] { const char* e = getenv("MSC_SPARSE_MP_CHUNK"); return e ? atoi(e) : 0; }();
	return force == 575 ? true : force == 512 ? false : nbins <= (1ull << 22);
}

// lists of any length up to msc_sparse_mp_max_entries() together; same arithmetic range as the LDS kernel (caller checks).
// parts (1 .. 16): waves per candidate, each writing its own record -- partials[c * parts + p]. The divergence form writes its two
// FP64 sums once per granule of the merged order (div_partials[c][div_stride][2], zeroed by the caller; the epilogue adds a pair's
// records in order), so they do not depend on parts either.
hipError_t msc_launch_pair_sparse_mp(hipStream_t st, const void* c_ent, const uint32_t* c_cum, const MscSparseHdr* c_hdr, const uint8_t* cand_scalars,
                                     uint64_t scalar_stride, const uint32_t* cand_slots, uint32_t m, const void* q_ent, const uint32_t* q_cum,
                                     const MscSparseHdr* q_hdr, const uint8_t* q_scalars, uint64_t nbins, int use_window, uint64_t min_len,
                                     uint64_t max_len, MscPartial* partials, void* div_tables, void* div_partials, int order, int num_cus, uint32_t max_total,
                                     uint32_t parts, uint32_t q_nnz, uint32_t c_max_nnz, uint32_t div_stride) {
	if (m == 0) return hipSuccess;
	if (max_total > msc_sparse_mp_max_entries() || parts < 1 || parts > 16 || (uint64_t)m * parts > 0xffffffffull) return hipErrorInvalidValue;
	if (div_tables && div_stride < msc_sparse_mp_div_records(max_total)) return hipErrorInvalidValue;
	// short lists, integer statistics: the whole-list kernel (the caller passes q_nnz / c_max_nnz; 0 = unknown)
	if (!div_tables && parts == 1 && msc_sparse_wl_fits(q_nnz, c_max_nnz)) {
		const size_t lds = ((size_t)q_nnz + 2 + 4 * ((size_t)c_max_nnz + 2)) * sizeof(uint2);
		const uint32_t blocks_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 256)));
		uint32_t blocks = (uint32_t)num_cus * blocks_per_cu;
		if (blocks > (m + 3) / 4) blocks = (m + 3) / 4;
		k_pair_sparse_wl<<<dim3(blocks), dim3(256), lds, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, (const uint2*)q_ent, q_cum, q_hdr,
		                                                     c_max_nnz, nbins, use_window, min_len, max_len, partials);
		return hipGetLastError();
	}
	const int wpe = msc_sparse_div_waves();
	const bool wide = mp_wide(nbins);
	const uint32_t per_cu = div_tables ? (uint32_t)wpe : std::min<uint32_t>(8, (160 * 1024) / (4 * (kMpChunkWide + 9) * 8 + 512));      // LDS-limited residency; every wave walks several candidates
	const uint64_t waves = (uint64_t)m * parts;
	uint32_t blocks = (uint32_t)num_cus * per_cu;
	if (blocks > (waves + 3) / 4) blocks = (uint32_t)((waves + 3) / 4);
	if (div_tables) {
#define MSC_MP_DIV(W) if (wide) MSC_MP_DIV_T(kMpChunkWide, W); else MSC_MP_DIV_T(kMpChunk, W)
#define MSC_MP_DIV_T(T, W) k_pair_sparse_mp<true, T, false, W><<<dim3(blocks), dim3(256), 0, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, \
		(const uint2*)q_ent, q_cum, q_hdr, q_scalars, nbins, use_window, min_len, max_len, partials, (const DivTerm*)div_tables, (double*)div_partials, order, nullptr, nullptr, parts, 0, div_stride, msc_sparse_mp_dma(), msc_sparse_mp_pairs())
		MSC_MP_DIV(6);
#undef MSC_MP_DIV
#undef MSC_MP_DIV_T
	} else if (wide) {
		k_pair_sparse_mp<false, kMpChunkWide><<<dim3(blocks), dim3(256), 0, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, (const uint2*)q_ent,
		                                                                          q_cum, q_hdr, q_scalars, nbins, use_window, min_len, max_len, partials, nullptr, nullptr, order,
		                                                                          nullptr, nullptr, parts, 0, 1, msc_sparse_mp_dma(), msc_sparse_mp_pairs());
	} else {
		k_pair_sparse_mp<false, kMpChunk><<<dim3(blocks), dim3(256), 0, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, (const uint2*)q_ent,
		                                                                      q_cum, q_hdr, q_scalars, nbins, use_window, min_len, max_len, partials, nullptr, nullptr, order,
		                                                                      nullptr, nullptr, parts, 0, 1, msc_sparse_mp_dma(), msc_sparse_mp_pairs());
	}
	return hipGetLastError();
}

// how many waves should share a candidate of a window of m (about `entries` merged entries each): fill the resident wave slots, keep
// at least two chunks per part
bool msc_sparse_wl_fits(uint32_t q_nnz, uint32_t c_max_nnz) {
	static const bool no_wl = getenv("MSC_SPARSE_NO_WL") != nullptr;
	return !no_wl && c_max_nnz && (uint64_t)q_nnz + 4ull * c_max_nnz + 10 <= 8192;
}
// waves per SIMD the divergence form of the merge-path kernel is compiled for and launched at (r03: 6 against 4 and 7)
int msc_sparse_div_waves() { return 6; }
uint32_t msc_sparse_mp_parts(uint32_t m, uint64_t entries, int num_cus, bool div) {
	static const bool off = getenv("MSC_SPARSE_MP_NO_PARTS") != nullptr;
	if (off || m == 0) return 1;
	const uint64_t slots = (uint64_t)num_cus * (div ? 4 * msc_sparse_div_waves() : 32), chunks = entries / kMpChunk;
	uint32_t parts = 1;
	if (div) {
		// the divergence form cuts between granules, each pair by its OWN length (k_pair_sparse_mp): the launch only sets the most a
		// pair may be cut into -- whatever the longest pair can use, unless the window alone fills the chip several times over
		const uint32_t cap = 16;
		const uint64_t fill = 1;
		if ((uint64_t)m >= fill * slots) return 1;
		while (parts < cap && (uint64_t)parts * 2 * kMpDivGran * kMpDivMinGran <= chunks && (uint64_t)m * parts * 2 <= 2 * fill * slots) parts *= 2;
		return parts;
	}
	while (parts < 16 && (uint64_t)m * parts * 2 <= slots && (uint64_t)parts * 2 * 2 <= chunks) parts *= 2;
	return parts;
}
// divergence records per pair of the merge-path kernel for lists of up to `entries` entries together
uint32_t msc_sparse_mp_div_records(uint64_t entries) {
	// (counted in the shorter of the two chunk sizes: enough for either; the kernel zeroes the records a pair does not reach)
	const uint64_t chunks = (entries + kMpChunk - 1) / kMpChunk;
	return (uint32_t)std::max<uint64_t>(1, (chunks + kMpDivGran - 1) / kMpDivGran);
}

// candidates (slot list or the first m slots behind c_hdr) against one query list: 16 x {markov, rre} partials per candidate
hipError_t msc_launch_pair_sparse_groups(hipStream_t st, const void* c_ent, const MscSparseHdr* c_hdr, const uint8_t* cand_scalars, uint64_t scalar_stride,
                                         const uint32_t* cand_slots, uint32_t m, const void* q_ent, const MscSparseHdr* q_hdr, int use_window, uint64_t min_len,
                                         uint64_t max_len, double* out) {
	if (m == 0) return hipSuccess;
	const uint64_t threads = (uint64_t)m * kSub;
	k_pair_sparse_groups<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>((const uint2*)c_ent, c_hdr, cand_scalars, scalar_stride, cand_slots, m,
	                                                                                    (const uint2*)q_ent, q_hdr, use_window, min_len, max_len, out);
	return hipGetLastError();
}

hipError_t msc_launch_sparse_nnz_sum(hipStream_t st, const MscSparseHdr* hdr, const uint8_t* cand_scalars, uint64_t scalar_stride, const uint32_t* slots, uint64_t first_slot,
                                     uint32_t m, int use_window, uint64_t min_len, uint64_t max_len, uint64_t* acc) {
	if (m == 0) return hipSuccess;
	k_sparse_nnz_sum<<<dim3((m + 255) / 256), dim3(256), 0, st>>>(hdr, cand_scalars, scalar_stride, slots, first_slot, m, use_window, min_len, max_len, (unsigned long long*)acc);
	return hipGetLastError();
}

hipError_t msc_launch_sparse_self_markov(hipStream_t st, const void* ent, const MscSparseHdr* hdr, const uint32_t* slots, uint64_t first_slot, uint32_t m, double* out) {
	if (m == 0) return hipSuccess;
	const uint64_t threads = (uint64_t)m * kSub;
	k_sparse_self_markov<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>((const uint2*)ent, hdr, slots, first_slot, m, out);
	return hipGetLastError();
}

// the pair-list form of the merge-path kernel: candidate c is scored against slot segs[pair_seg[c]].q_slot of the query set, inside
// that segment's length window (use_window); one record per candidate (and, with div_tables, its two divergence sums)
hipError_t msc_launch_pair_sparse_mp_pairs(hipStream_t st, const void* c_ent, const uint32_t* c_cum, const MscSparseHdr* c_hdr, const uint8_t* cand_scalars,
                                           uint64_t scalar_stride, const uint32_t* cand_slots, uint32_t m, const void* q_ent, const uint32_t* q_cum,
                                           const MscSparseHdr* q_hdr, uint64_t nbins, int use_window, const MscBatchSeg* segs, const uint32_t* pair_seg,
                                           MscPartial* partials, int order, int num_cus, const uint8_t* q_scalars, uint64_t q_scalar_stride, void* div_tables,
                                           void* div_partials, uint32_t div_stride) {
	if (m == 0) return hipSuccess;
	const bool wide = mp_wide(nbins);
	const uint32_t per_cu = std::min<uint32_t>(8, (160 * 1024) / (4 * (kMpChunkWide + 9) * 8 + 512));
	uint32_t blocks = (uint32_t)num_cus * per_cu;
	if (blocks > (m + 3) / 4) blocks = (m + 3) / 4;
	if (div_tables) {      // the divergence sums of every pair as well: the same walk, hence the same bits, as the 1 x M divergence form
		if (!q_scalars || !div_partials) return hipErrorInvalidValue;
		const int wpe = msc_sparse_div_waves();
		blocks = (uint32_t)num_cus * (uint32_t)wpe;
		if (blocks > (m + 3) / 4) blocks = (m + 3) / 4;
#define MSC_MP_DIVP(W) if (wide) MSC_MP_DIVP_T(kMpChunkWide, W); else MSC_MP_DIVP_T(kMpChunk, W)
#define MSC_MP_DIVP_T(T, W) k_pair_sparse_mp<true, T, true, W><<<dim3(blocks), dim3(256), 0, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, (const uint2*)q_ent, \
		q_cum, q_hdr, q_scalars, nbins, use_window, 0, ~0ull, partials, (const DivTerm*)div_tables, (double*)div_partials, order, segs, pair_seg, 1, q_scalar_stride, div_stride, msc_sparse_mp_dma(), msc_sparse_mp_pairs())
		MSC_MP_DIVP(6);
#undef MSC_MP_DIVP
#undef MSC_MP_DIVP_T
		return hipGetLastError();
	}
	if (wide) {
		k_pair_sparse_mp<false, kMpChunkWide, true><<<dim3(blocks), dim3(256), 0, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, (const uint2*)q_ent, q_cum,
		                                                                               q_hdr, nullptr, nbins, use_window, 0, ~0ull, partials, nullptr, nullptr, order, segs, pair_seg, 1, 0, 1, msc_sparse_mp_dma(), msc_sparse_mp_pairs());
		return hipGetLastError();
	}
	k_pair_sparse_mp<false, kMpChunk, true><<<dim3(blocks), dim3(256), 0, st>>>((const uint2*)c_ent, c_cum, c_hdr, cand_scalars, scalar_stride, cand_slots, m, (const uint2*)q_ent, q_cum,
	                                                                    q_hdr, nullptr, nbins, use_window, 0, ~0ull, partials, nullptr, nullptr, order, segs, pair_seg, 1, 0, 1, msc_sparse_mp_dma(), msc_sparse_mp_pairs());
	return hipGetLastError();
}

hipError_t msc_launch_sparse_scatter_batch(hipStream_t st, const void* ent, const MscSparseHdr* hdr, const uint32_t* slots, const uint32_t* seg, uint32_t n_members,
                                           uint64_t nbins, uint32_t* acc, uint32_t* touched) {
	if (n_members == 0) return hipSuccess;
	k_sparse_scatter_batch<<<dim3(n_members), dim3(256), 0, st>>>((const uint2*)ent, hdr, slots, seg, n_members, nbins, acc, touched);
	return hipGetLastError();
}

hipError_t msc_launch_sparse_mean_count_batch(hipStream_t st, int dtype, const uint32_t* acc, uint64_t nbins, uint32_t n_chunks, uint64_t chunk_bins, uint32_t n_centres,
                                              const uint32_t* m_of, uint64_t* counts, const uint32_t* touched) {
	if (n_centres == 0) return hipSuccess;
	const dim3 grid(n_chunks, n_centres);
	if (touched) {
		if (chunk_bins % 512) return hipErrorInvalidValue;
		switch (dtype) {
		case 8: k_sparse_mean_count_groups<uint8_t><<<grid, dim3(64), 0, st>>>(acc, touched, nbins, chunk_bins, m_of, counts); break;
		case 16: k_sparse_mean_count_groups<uint16_t><<<grid, dim3(64), 0, st>>>(acc, touched, nbins, chunk_bins, m_of, counts); break;
		case 32: k_sparse_mean_count_groups<uint32_t><<<grid, dim3(64), 0, st>>>(acc, touched, nbins, chunk_bins, m_of, counts); break;
		default: k_sparse_mean_count_groups<uint64_t><<<grid, dim3(64), 0, st>>>(acc, touched, nbins, chunk_bins, m_of, counts); break;
		}
		return hipGetLastError();
	}
	switch (dtype) {
	case 8: k_sparse_mean_count_batch<uint8_t><<<grid, dim3(64), 0, st>>>(acc, nbins, chunk_bins, m_of, counts); break;
	case 16: k_sparse_mean_count_batch<uint16_t><<<grid, dim3(64), 0, st>>>(acc, nbins, chunk_bins, m_of, counts); break;
	case 32: k_sparse_mean_count_batch<uint32_t><<<grid, dim3(64), 0, st>>>(acc, nbins, chunk_bins, m_of, counts); break;
	default: k_sparse_mean_count_batch<uint64_t><<<grid, dim3(64), 0, st>>>(acc, nbins, chunk_bins, m_of, counts); break;
	}
	return hipGetLastError();
}

hipError_t msc_launch_sparse_mean_write_batch(hipStream_t st, int dtype, uint32_t* acc, uint64_t nbins, uint32_t n_chunks, uint64_t chunk_bins, uint32_t n_centres,
                                              const uint32_t* m_of, const uint64_t* chunk_off, const uint64_t* chunk_cum, void* ent, uint32_t* cum, uint32_t* touched) {
	if (n_centres == 0) return hipSuccess;
	const dim3 grid(n_chunks, n_centres);
	if (touched) {
		if (chunk_bins % 512) return hipErrorInvalidValue;
		switch (dtype) {
		case 8: k_sparse_mean_write_groups<uint8_t><<<grid, dim3(64), 0, st>>>(acc, touched, nbins, chunk_bins, m_of, chunk_off, chunk_cum, (uint2*)ent, cum); break;
		case 16: k_sparse_mean_write_groups<uint16_t><<<grid, dim3(64), 0, st>>>(acc, touched, nbins, chunk_bins, m_of, chunk_off, chunk_cum, (uint2*)ent, cum); break;
		case 32: k_sparse_mean_write_groups<uint32_t><<<grid, dim3(64), 0, st>>>(acc, touched, nbins, chunk_bins, m_of, chunk_off, chunk_cum, (uint2*)ent, cum); break;
		default: k_sparse_mean_write_groups<uint64_t><<<grid, dim3(64), 0, st>>>(acc, touched, nbins, chunk_bins, m_of, chunk_off, chunk_cum, (uint2*)ent, cum); break;
		}
		return hipGetLastError();
	}
	switch (dtype) {
	case 8: k_sparse_mean_write_batch<uint8_t><<<grid, dim3(64), 0, st>>>(acc, nbins, chunk_bins, m_of, chunk_off, chunk_cum, (uint2*)ent, cum); break;
	case 16: k_sparse_mean_write_batch<uint16_t><<<grid, dim3(64), 0, st>>>(acc, nbins, chunk_bins, m_of, chunk_off, chunk_cum, (uint2*)ent, cum); break;
	case 32: k_sparse_mean_write_batch<uint32_t><<<grid, dim3(64), 0, st>>>(acc, nbins, chunk_bins, m_of, chunk_off, chunk_cum, (uint2*)ent, cum); break;
	default: k_sparse_mean_write_batch<uint64_t><<<grid, dim3(64), 0, st>>>(acc, nbins, chunk_bins, m_of, chunk_off, chunk_cum, (uint2*)ent, cum); break;
	}
	return hipGetLastError();
}

hipError_t msc_launch_sparse_assign_batch(hipStream_t st, void* d_ent, uint32_t* d_cum, MscSparseHdr* d_hdr, const void* s_ent, const uint32_t* s_cum, const MscSparseHdr* s_hdr,
                                          const uint32_t* ds, const uint32_t* ss, const uint64_t* dst_off, uint32_t n) {
	if (n == 0) return hipSuccess;
	k_sparse_assign_batch<<<dim3(n), dim3(256), 0, st>>>((uint2*)d_ent, d_cum, d_hdr, (const uint2*)s_ent, s_cum, s_hdr, ds, ss, dst_off, n);
	return hipGetLastError();
}
