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
// k_pair_sparse: one LANE per (candidate, index sub-range): a branch-light two-pointer merge of the two sorted lists
// restricted to the sub-range. No cross-lane communication (the prefix difference at a sub-range start comes from the
// stored cum arrays), so 64 independent merges run per wavefront and the kernel is bound by the candidate lists' bytes
// (12 B per stored bin instead of 4^k * sizeof(T) per histogram).
//
// Build: the dense builder (hist_build.hip) fills a scratch slot per sequence of the batch, then k_sparse_count /
// k_sparse_write compact it IN INDEX ORDER: in the tile-permuted layout every lane already holds a logically consecutive
// run, so one wave scan per tile yields ordered output with coalesced reads.
#include "msc_internal.h"

namespace {

constexpr int kBlockC = 1024;                 // 16 waves = the 16 index sub-ranges of one sequence
constexpr int kSub = MSC_SPARSE_SUB;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
	return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
	v = dpp_add<0x111, 0xf>(v);
	v = dpp_add<0x112, 0xf>(v);
	v = dpp_add<0x114, 0xf>(v);
	v = dpp_add<0x118, 0xf>(v);
	v = dpp_add<0x142, 0xa>(v);
	v = dpp_add<0x143, 0xc>(v);
	return v;
}
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

// per-candidate 16 x 16 table of the exact per-bin divergence terms (same idea as k_div_tables of the dense path)
__global__ void __launch_bounds__(256) k_sparse_div_tables(const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
                                                           const uint32_t* __restrict__ cand_slots, uint32_t m,
                                                           const uint8_t* __restrict__ q_scalars, int order, DivTerm* __restrict__ tables) {
	const uint32_t c = blockIdx.x;
	if (c >= m) return;
	const uint32_t slot = cand_slots ? cand_slots[c] : c;
	const double cm = (double)reinterpret_cast<const MscSlotScalars*>(cand_scalars + (uint64_t)slot * scalar_stride)->mag;
	const double qm = (double)reinterpret_cast<const MscSlotScalars*>(q_scalars)->mag;
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

}  // namespace

// ------------------------------------------------------------------------------------------------ mean of sparse members
// get_mean / closest on sparse slots (cluster/ClusterFactory.cpp:338-380, cluster/Trainer.cpp:144-157). Outside the union of
// the members' stored bins every member holds 1, so the FP64 mean is exactly 1 there; inside, the column sum is m + (sum of
// excesses). The excesses are scatter-added into a dense u32 accumulator (4 * 4^k bytes, kept zero between calls), which is
// then swept in index order: chunk counts -> host prefix -> ordered write of the rounded mean as a sparse slot.
__global__ void __launch_bounds__(256) k_sparse_scatter(const uint2* __restrict__ ent, const MscSparseHdr* __restrict__ hdr,
                                                        const uint32_t* __restrict__ slots, uint32_t m, uint32_t* __restrict__ acc) {
	const uint32_t j = blockIdx.x;
	if (j >= m) return;
	const MscSparseHdr h = hdr[slots ? slots[j] : j];
	for (uint32_t t = threadIdx.x; t < h.nnz; t += blockDim.x) {
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
	k_sparse_scatter<<<dim3(m), dim3(256), 0, st>>>((const uint2*)ent, hdr, slots, m, acc);
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
