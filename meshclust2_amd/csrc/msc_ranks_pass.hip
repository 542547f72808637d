// msc_ranks_pass.hip -- the 1 x M pass of Trainer::get_close / filter (cluster/Trainer.cpp:26-61) over RANK LISTS, for histograms of up to
// 4^9 bins: no merge of two sorted lists. r04; r05: the long-list / divergence form (k_pair_ranks_items, below) walks first copies only.
//
// A rank list = the bins of a histogram's counted k-mers in bin order, a bin with count c listed e = c - 1 times (every bin starts at the
// pseudocount 1, nonltr/KmerHashTable.cpp:69-72): 4 bytes per k-mer where the (bin, value) lists of sparse.hip take 8 bytes per stored bin
// plus their prefix words, and -- what matters -- a form in which the three integer reductions of a pair (pair_features.hip: manhattan,
// the dot product, the earth mover's distance; predict/Feature.cpp:859-871, 1113-1124, 1505-1518) need no co-ordination between the lists:
//     emd           = sum_t |a_t - b_t|                     the t-th k-mer of either histogram, both lists padded with 4^k (msc_emd_ranks.hip)
//     sum e_c e_q   = sum over the candidate's entries of e_q(bin)
//     sum min(e_c, e_q) = sum over the candidate's entries, the r-th copy of a bin (r = 0, 1, ..) counting [r < e_q(bin)]
// e_q(bin) is a LOOKUP: the query's histogram sits in LDS as two bits per bin -- e_q = 0, 1, 2 or "three and more" --, one word per 16 bins, shared by the
// sixteen waves of a workgroup for every candidate they walk. An entry whose bin is not large in the query (all but a handful: a 1 kb
// sequence at k = 9 has ~2 large bins) contributes its present bit to the product and, if it is the first copy of its bin, to the minimum;
// the rare entry that hits a large bin looks e_q up in the query's own rank list (LDS, binary search) and its copy number r in the
// candidate's (global memory, binary search). Exact in integers; records as k_pair_sparse_wl writes them (manh, dot over the union of stored
// bins, emd), so the epilogue and everything behind it are shared.
//
// Per candidate a wave reads its list once, 16 bytes per lane and load, coalesced: 4 KB for a 1 kb sequence (the merge kernels: 8 KB of
// entries + 4 KB of prefix words, staged through LDS and walked by a data-dependent two-pointer loop). Roofline: HBM, 4 bytes per k-mer.
#include "msc_internal.h"
#include "msc_wave.h"

namespace {

// wave totals by DPP (msc_wave.h: wave_total_u32, six row operations + a readlane): a __shfl_xor butterfly is twelve ds_bpermute_b32 per
// 64-bit value, LDS round trips one behind the other -- five such totals were a third of what a 1 kb candidate costs its wave.
// 64-bit per-lane values: the four 16-bit pieces are totalled separately.
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
	const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
	const uint64_t a = wave_total_u32(lo & 0xffffu), b = wave_total_u32(lo >> 16);
	uint64_t r = a + (b << 16);
	if (__builtin_amdgcn_readfirstlane((int)(__ballot(hi != 0) != 0))) {          // (rarely: a lane past 2^32)
		const uint64_t c = wave_total_u32(hi & 0xffffu), d = wave_total_u32(hi >> 16);
		r += (c << 32) + (d << 48);
	}
	return r;
}
// the two FP64 spot sums of a wave: a butterfly only when some lane holds one (fixed order: the butterfly's)
__device__ __forceinline__ void wave_sum_f64_pair(double& x, double& y) {
	if (__ballot(x != 0.0 || y != 0.0) == 0) return;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) { x += __shfl_xor(x, off, 64); y += __shfl_xor(y, off, 64); }
}

// neighbours in the wave by DPP (gfx9 wave shifts): __shfl_up / __shfl_down compile to ds_bpermute_b32, an LDS round trip each, and the
// walks below take a neighbour or a prefix for every 256 entries. Lanes without a source read 0.
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }          // wave_shr:1: lane l <- lane l - 1
__device__ __forceinline__ uint32_t lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }          // wave_shl:1: lane l <- lane l + 1
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max(uint32_t v) {
	const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
	return o > v ? o : v;
}
// inclusive prefix maximum over the 64 lanes (row_shr 1, 2, 4, 8, then row_bcast 15 / 31: the steps of wave_incl_scan)
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) {
	v = dpp_max<0x111, 0xf>(v);
	v = dpp_max<0x112, 0xf>(v);
	v = dpp_max<0x114, 0xf>(v);
	v = dpp_max<0x118, 0xf>(v);
	v = dpp_max<0x142, 0xa>(v);
	v = dpp_max<0x143, 0xc>(v);
	return v;
}

constexpr uint32_t kRpBlock = 1024;          // sixteen waves share the query's tables
constexpr uint32_t kRpQCap = 8192;           // longest query rank list (host-checked against the set's bound)

// ------------------------------------------------------------------------------------------------ rank lists of a sparse set
// n[slot] = the slot's k-mers = the last inclusive prefix of its excess counts; lists are padded to a multiple of four entries
__global__ void __launch_bounds__(256) k_rkl_sizes(const MscSparseHdr* __restrict__ hdr, const uint32_t* __restrict__ cum, uint64_t capacity, uint32_t* __restrict__ n) {
	const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= capacity) return;
	const MscSparseHdr* h = hdr + s;
	n[s] = h->nnz ? cum[h->off + h->nnz - 1] : 0u;
}
// exclusive scan of the padded sizes by one workgroup (a set is scanned once per build); off[capacity] = the total
__global__ void __launch_bounds__(1024) k_rkl_scan(const uint32_t* __restrict__ n, uint64_t capacity, uint64_t* __restrict__ off) {
	__shared__ uint64_t s_part[1024];
	const uint64_t per = (capacity + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < capacity ? lo + per : capacity;
	uint64_t s = 0;
	for (uint64_t i = lo; i < hi; i++) s += (n[i] + 3u) & ~3u;
	s_part[threadIdx.x] = s;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) {
		const uint64_t v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
		__syncthreads();
		s_part[threadIdx.x] += v;
		__syncthreads();
	}
	uint64_t run = s_part[threadIdx.x] - s;
	for (uint64_t i = lo; i < hi; i++) { off[i] = run; run += (n[i] + 3u) & ~3u; }
	if (threadIdx.x == 1023) off[capacity] = s_part[1023];
}
// one wave per slot: entry j = (bin, value) puts value - 1 copies of bin at [cum[j] - (value - 1), cum[j]); the tail is padded with 4^k
__global__ void __launch_bounds__(256) k_rkl_fill(const uint2* __restrict__ ent, const uint32_t* __restrict__ cum, const MscSparseHdr* __restrict__ hdr, uint64_t capacity,
                                                  const uint32_t* __restrict__ n, const uint64_t* __restrict__ off, uint32_t nbins, uint32_t* __restrict__ out) {
	const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const uint32_t lane = threadIdx.x & 63;
	if (s >= capacity) return;
	const MscSparseHdr* h = hdr + s;
	uint32_t* o = out + off[s];
	const uint32_t tot = n[s], pad = (tot + 3u) & ~3u;
	for (uint32_t j = lane; j < h->nnz; j += 64) {
		const uint2 en = ent[h->off + j];
		const uint32_t e = en.y ? en.y - 1u : 0u, end = cum[h->off + j];
		for (uint32_t t = end - e; t < end; t++) o[t] = en.x;
	}
	if (lane < pad - tot) o[tot + lane] = nbins;
}

// the rank list of ONE slot into out[0 .. n), padded with 4^k to a multiple of 256 (the query of a pass when it is too long for LDS)
__global__ void __launch_bounds__(1024) k_rank_expand_one(const uint2* __restrict__ ent, const uint32_t* __restrict__ cum, const MscSparseHdr* __restrict__ hdr_p, uint32_t nbins,
                                                          uint32_t* __restrict__ out, uint32_t cap, uint32_t* __restrict__ guard) {
	const MscSparseHdr h = *hdr_p;
	const uint32_t tot = h.nnz ? cum[h.off + h.nnz - 1] : 0u, pad = (tot + 255u) & ~255u;
	if (pad > cap) {          // the set's bound did not hold: say so and write nothing
		if (blockIdx.x == 0 && threadIdx.x == 0 && guard) atomicOr(guard, 1u);
		return;
	}
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < h.nnz; j += gridDim.x * blockDim.x) {
		const uint2 en = ent[h.off + j];
		const uint32_t e = en.y ? en.y - 1u : 0u, end = cum[h.off + j];
		for (uint32_t t = end - e; t < end; t++) out[t] = en.x;
	}
	if (blockIdx.x == 0) for (uint32_t i = tot + threadIdx.x; i < pad; i += blockDim.x) out[i] = nbins;
}

// ------------------------------------------------------------------------------------------------ the divergence statistics in rank form
// jefferey_divergence / jensen_shannon (predict/Feature.cpp:1235-1262, 988-1008) are sums over bins of a term that depends on the pair of
// counts (c, q) of the bin and on the two magnitudes only; every bin outside the union of the two lists holds (1, 1). The merge kernels add
// term(c, q) - term(1, 1) per bin of the union in FP64 as they walk it. Here the candidate's rank list is streamed without a walk of the
// query's, so the sum is re-arranged by COPIES: the r-th copy (r = 0, 1, ..) of a bin whose count in the query is b contributes
//     r = 0:  F(2, b)                      r >= 1:  F(r + 2, b) - F(r + 1, b)              with F(x, b) = term(x, b) - term(1, 1)
// (they telescope to F(e_c + 1, b)), and the query's bins the candidate does not hold contribute F(1, b). The pass only COUNTS copies per
// cell -- integers --, FP64 on the spot for counts of 9 and more in the candidate or 8 and more in the query, and k_rank_items_finish evaluates each
// candidate's sums from its cell counts in one fixed order. (r05: the long-list pass counts BINS per cell (count in the candidate, count in the
// query) -- every first copy as a bin held once, the candidate's repeated bins moved to their true cell from a list of their own; the
// short-list kernel k_pair_ranks_1xm no longer carries a divergence form: such passes take the items kernel whatever their length.)
struct RkDivTerm { double jd, js; };
__device__ __forceinline__ RkDivTerm rk_div_term(uint32_t cand_count, uint32_t q_count, double cand_mag, double q_mag, int order) {          // = div_term_sp of sparse.hip
	RkDivTerm t;
	const bool cf = order == MSC_ORDER_CAND_FIRST;
	const double pp = cf ? (double)cand_count / cand_mag : (double)q_count / q_mag;
	const double pq = cf ? (double)q_count / q_mag : (double)cand_count / cand_mag;
	t.jd = (pp - pq) * log(pp / pq);
	const double avg = 0.5 * (pp + pq);
	t.js = pp * log(pp / avg) + pq * log(pq / avg);
	return t;
}
__device__ __noinline__ RkDivTerm rk_div_term_call(uint32_t cand_count, uint32_t q_count, double cand_mag, double q_mag, int order) {
	return rk_div_term(cand_count, q_count, cand_mag, q_mag, order);
}
constexpr uint32_t kRkCells = 64;          // cell (r, b) at r * 8 + b, r < 7 (7: unused), 1 <= b < 8 (0: unused)

// ------------------------------------------------------------------------------------------------ the pass
// first index in the sorted list v[0 .. n) whose value is >= x
template <typename Load>
__device__ __forceinline__ uint32_t lower_bound_u32(Load v, uint32_t n, uint32_t x) {
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (v(mid) < x) lo = mid + 1; else hi = mid;
	}
	return lo;
}

// LDS: [table: nbins / 16 + 1 words, two bits per bin (present, large)][query ranks: padded to 256 with nbins]
//
// A wave walks its candidates three deep: while candidate i is scored, the first KiB-entries of candidate i + 1 are on their way (its
// list's place and length arrived an iteration earlier), the place and length of candidate i + 2 are being fetched (its slot arrived an
// iteration earlier) and the slot of candidate i + 3 is read from the window's list. Without that every candidate cost its wave a chain of
// three dependent round trips plus one per 256 entries -- 5 us per candidate, 0.16 ms per 100 000 however short the lists.
template <bool QG>
__global__ void __launch_bounds__(kRpBlock) k_pair_ranks_1xm(const uint32_t* __restrict__ c_rk, const uint64_t* __restrict__ c_off, const uint32_t* __restrict__ c_n,
                                                              const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride, const uint32_t* __restrict__ cand_slots, uint64_t first,
                                                              uint32_t m, const uint2* __restrict__ q_ent, const uint32_t* __restrict__ q_cum, const MscSparseHdr* __restrict__ q_hdr_p,
                                                              uint32_t nbins, int use_window, uint64_t min_len, uint64_t max_len, MscPartial* __restrict__ partials, uint32_t q_cap,
                                                              uint32_t* __restrict__ guard, const uint32_t* __restrict__ q_ranks_g) {
	extern __shared__ __attribute__((aligned(16))) uint32_t s_rp[];
	const uint32_t words = nbins / 16 + 1;
	uint32_t* sb = s_rp;
	// the query's own rank list: in LDS, or (QG: a query of more than kRpQCap k-mers) the copy k_rank_expand_one left in global memory,
	// read through L2 by every wave alike
	uint32_t* rq_l = s_rp + ((words + 3u) & ~3u);          // 16-byte aligned
	const uint32_t* rq = QG ? q_ranks_g : rq_l;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const MscSparseHdr qh = *q_hdr_p;
	const uint32_t nq = qh.nnz;
	const uint2* Q = q_ent + qh.off;
	const uint32_t* CQ = q_cum + qh.off;
	const uint32_t nq_tot = nq ? CQ[nq - 1] : 0u;          // (<= q_cap: the host's bound on the set, which sized the LDS)
	if (nq_tot > q_cap) {          // the bound did not hold: say so (the host fails the call) and touch nothing
		if (threadIdx.x == 0) atomicOr(guard, 1u);
		return;
	}
	const uint32_t nq_pad = (nq_tot + 255u) & ~255u;
	const uint32_t bd = blockDim.x;
	__shared__ uint32_t s_ticket;
	if (threadIdx.x == 0) s_ticket = (uint32_t)((uint64_t)m * blockIdx.x / gridDim.x) + 3 * (bd / 64);          // (two barriers stand between this and the first ticket)
	for (uint32_t i = threadIdx.x; i < words; i += bd) sb[i] = 0u;
	if constexpr (!QG) for (uint32_t i = nq_tot + threadIdx.x; i < (nq_pad > 1024u ? nq_pad : 1024u); i += bd) rq_l[i] = nbins;          // (at least the four chunks a wave reads without asking)
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < nq; j += bd) {
		const uint2 en = Q[j];
		const uint32_t e = en.y ? en.y - 1u : 0u, end = CQ[j];
		if (e >= 1) atomicOr(&sb[en.x >> 4], (e >= 3 ? 3u : e) << (2 * (en.x & 15)));          // two bits per bin: e_q = 0, 1, 2, or 3 = "three and more: look it up"
		if constexpr (!QG) for (uint32_t t = end - e; t < end; t++) rq_l[t] = en.x;
	}
	__syncthreads();
	// WHO scores WHICH candidate: the workgroup owns a contiguous share of the window, [s0, s1), and its waves take the share's
	// candidates as they come free -- a counter in LDS, read three candidates ahead like the window's list (a wave's first three are its
	// own by position). r05, until the end of the round: every wave took candidates w, w + waves, .. and two workgroups per CU were
	// launched of which the registers admit one -- the second set rebuilt the tables (3.9 us) behind the first.
	const uint32_t s0 = (uint32_t)((uint64_t)m * blockIdx.x / gridDim.x), s1 = (uint32_t)((uint64_t)m * (blockIdx.x + 1) / gridDim.x);
	const uint32_t nw = bd / 64;
	const uint32_t c0 = s0 + wave;
	if (c0 >= s1) return;
	// stage 1: the slot of a candidate (identity without a slot list)
	// (every load of the three stages is UNCONDITIONAL, from a clamped place, and what must not count is dropped when it is used: behind a
	// load under a branch the compiler cannot count what is in flight and waits for all of it -- vmcnt(0) in front of every candidate, the
	// next one's entries included: the walk was three deep on paper only, r05)
	auto slot_of = [&](uint32_t c) -> uint64_t { const uint32_t cc = c < m ? c : m - 1; return cand_slots ? (uint64_t)cand_slots[cc] : first + cc; };
	// stage 2: where its rank list sits, how long it is, whether the length window keeps it (n = 0xffffffff: not scored)
	// (... as it arrives: whether the candidate counts is decided by whoever USES the record, an iteration later -- a comparison here makes the
	// compiler wait for the loads it has just issued)
	struct Meta { uint64_t off, len; uint32_t n; bool inside; };
	auto meta_of = [&](uint32_t c, uint64_t slot) -> Meta {
		Meta mt;
		const uint32_t cc = c < m ? c : m - 1;
		const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(cand_scalars + (cand_slots ? slot : (uint64_t)cc) * scalar_stride);
		mt.len = cs->length;
		mt.off = c_off[slot];
		mt.n = c_n[slot];
		mt.inside = c < s1;
		return mt;
	};
	auto scored_n = [&](const Meta& mt) -> uint32_t { return !mt.inside || (use_window && (mt.len < min_len || mt.len > max_len)) ? 0xffffffffu : mt.n; };
	// stage 3: the first four chunks of 256 entries (16 bytes per lane each)
	auto data_of = [&](const Meta& mt, uint4 (&d)[4]) {
		const uint32_t n_of = scored_n(mt);
		const uint32_t n_pad = n_of == 0xffffffffu ? 0u : (n_of + 3u) & ~3u;
		const uint32_t last4 = n_pad ? n_pad - 4u : 0u;          // (a list is padded to four entries in the arena; an empty one reads its first four words, whatever they are)
		const uint32_t* P = c_rk + mt.off;
#pragma unroll
		for (uint32_t u = 0; u < 4; u++) {
			const uint32_t t = 256 * u + 4 * lane;
			d[u] = *reinterpret_cast<const uint4*>(P + (t < last4 ? t : last4));
		}
	};
	uint32_t c_1 = c0 + nw, c_2 = c0 + 2 * nw, c_3 = 0;
	uint64_t slot2 = slot_of(c_2);
	Meta meta1 = meta_of(c_1, slot_of(c_1));
	Meta meta0 = meta_of(c0, slot_of(c0));
	uint4 d0[4];
	data_of(meta0, d0);
	for (uint32_t c = c0; c < s1; c = c_1, c_1 = c_2, c_2 = c_3) {
		uint32_t tk = 0;
		if (lane == 0) tk = atomicAdd(&s_ticket, 1u);
		c_3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
		const uint64_t slot3 = slot_of(c_3);
		const Meta meta2 = meta_of(c_2, slot2);
		uint4 d1[4];
		data_of(meta1, d1);
		const uint32_t n0 = scored_n(meta0);
		if (n0 != 0xffffffffu) {
			const uint32_t nc = n0, nc_pad = (nc + 3u) & ~3u;
			const uint32_t* P = c_rk + meta0.off;
			const uint32_t T = nc > nq_tot ? nc : nq_tot;
			uint64_t emd = 0;
			uint32_t prod = 0, mins = 0;          // sum e_c e_q and sum min(e_c, e_q) over the candidate's entries: what the packed counter does not hold
			uint32_t carry = 0xffffffffu;         // the entry in front of this chunk's first
			// An entry costs a table look-up, a comparison with the entry in front of it and ONE add: `pk` counts the entries whose bin the
			// query holds (low half) and the first copies among them (high half). What the query holds twice and more -- rare: a 1 kb
			// sequence has a handful of such bins -- adds e_q - 1 to the product and, for a further copy of the bin, its share of the
			// minimum; a chunk in which any lane met one walks its four entries again for that (r05: every entry went through those
			// branches, and every chunk through six lane shifts for copy indices hardly any entry asks for: 370 instructions per
			// 1 kb candidate, 80 registers -- one workgroup per CU instead of two).
			uint32_t pk = 0;
			auto chunk = [&](uint32_t t0, const uint4& a, const uint4& a_before) {          // a_before: the chunk in front (unused at t0 = 0)
				const uint32_t t = t0 + 4 * lane;
				uint4 b = make_uint4(nbins, nbins, nbins, nbins);
				if (!QG && t0 < 1024) b = *reinterpret_cast<const uint4*>(rq + t);          // (the list in LDS is filled up to 1 024 entries)
				else if (t < nq_pad) b = *reinterpret_cast<const uint4*>(rq + t);
				uint32_t d = sad_u32(a.x, b.x, 0u);
				d = sad_u32(a.y, b.y, d);
				d = sad_u32(a.z, b.z, d);
				d = sad_u32(a.w, b.w, d);
				emd += d;
				uint32_t before = lane_prev(a.w);
				if (lane == 0) before = carry;
				carry = (uint32_t)__builtin_amdgcn_readlane((int)a.w, 63);
				const uint32_t av[4] = {a.x, a.y, a.z, a.w};
				const uint32_t pv[4] = {before, a.x, a.y, a.z};
				uint32_t two4[4];
				bool rare = false;
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const uint32_t bin = av[j];
					const uint32_t two = __builtin_amdgcn_ubfe(sb[bin >> 4], 2 * bin, 2);          // (v_bfe_u32 takes the offset modulo 32; the padding value 4^k reads the zero word behind the table)
					two4[j] = two;
					const uint32_t present = two ? 1u : 0u;
					pk += present + (bin != pv[j] ? present << 16 : 0u);
					rare |= two >= 2u;
				}
				if (__ballot(rare)) {
					uint32_t seq[8] = {0, 0, 0, before, a.x, a.y, a.z, a.w};          // the seven entries in front of a lane's last: an entry's copy index
					seq[0] = lane_prev(a.x); seq[1] = lane_prev(a.y); seq[2] = lane_prev(a.z);
					if (lane == 0) {
						seq[0] = t0 ? (uint32_t)__builtin_amdgcn_readlane((int)a_before.x, 63) : 0xffffffffu;
						seq[1] = t0 ? (uint32_t)__builtin_amdgcn_readlane((int)a_before.y, 63) : 0xffffffffu;
						seq[2] = t0 ? (uint32_t)__builtin_amdgcn_readlane((int)a_before.z, 63) : 0xffffffffu;
					}
#pragma unroll
					for (int j = 0; j < 4; j++) {
						const uint32_t bin = av[j];
						if (two4[j] < 2u) continue;
						uint32_t e_q = two4[j];
						if (e_q == 3u) {          // the query holds this k-mer three times or more -- how often, says its own rank list
							const uint32_t lo = lower_bound_u32([&](uint32_t i) { return rq[i]; }, nq_tot, bin);
							e_q = lower_bound_u32([&](uint32_t i) { return rq[i]; }, nq_tot, bin + 1) - lo;          // (two searches: a homopolymer run is thousands of copies)
						}
						prod += e_q - 1;
						if (bin < nbins && bin == pv[j]) {          // a further copy of a bin (not the padding behind the list's end): its copy index from the entries in front of it
							uint32_t r = 0;
							for (int i = 3 + j; i >= 0; i--) { if (seq[i] != bin) break; r++; }
							if (r == (uint32_t)(4 + j)) r = t + j - lower_bound_u32([&](uint32_t i) { return P[i]; }, nc, bin);          // (a run of eight and more)
							if (r < e_q) mins += 1;
						}
					}
				}
			};
			const uint4 fill = make_uint4(nbins, nbins, nbins, nbins);
#pragma unroll
			for (uint32_t u = 0; u < 4; u++) if (256 * (u + 1) > nc_pad && 256 * u + 4 * lane >= nc_pad) d0[u] = fill;          // (behind the list's end: what the clamped load brought is not the list's)
#pragma unroll
			for (uint32_t u = 0; u < 4; u++) if (256 * u < T) chunk(256 * u, d0[u], d0[u ? u - 1 : 0]);          // the chunks that were fetched ahead
			prod += pk & 0xffffu;
			mins += pk >> 16;
			if (T > 1024) {          // longer lists: the rest as it comes
				uint4 a_before = d0[3];
				for (uint32_t t0 = 1024; t0 < T; t0 += 256) {
					const uint32_t t = t0 + 4 * lane;
					uint4 a = make_uint4(nbins, nbins, nbins, nbins);
					if (t < nc_pad) a = *reinterpret_cast<const uint4*>(P + t);
					pk = 0;
					chunk(t0, a, a_before);
					prod += pk & 0xffffu;
					mins += pk >> 16;
					a_before = a;
				}
			}
			uint64_t emd_t, prod_t, mins_t;
			if (T <= 4096) {          // the three totals in one pass over the wave (msc_wave.h): each below 2^32 -- emd < 4 096 x 4^k, sum e_c e_q <= 4 096^2
				const uint32_t rows = wave_sum4_rows((uint32_t)emd, prod, mins, 0u);
				emd_t = MSC_ROW_A(rows); prod_t = MSC_ROW_B(rows); mins_t = MSC_ROW_C(rows);
			} else {
				emd_t = wave_sum_u64(emd);
				prod_t = wave_sum_u64(prod);
				mins_t = wave_sum_u64(mins);
			}
			if (lane == 0) {
				MscPartial out;
				out.manh = (uint64_t)nc + nq_tot - 2 * mins_t;          // sum |e_c - e_q|
				out.dot = prod_t + nc + nq_tot;                          // sum (c q - 1) over the union of stored bins (the epilogue adds 4^k)
				out.emd = emd_t;
				partials[c] = out;
			}
		}
		meta0 = meta1;
		meta1 = meta2;
		slot2 = slot3;
#pragma unroll
		for (int u = 0; u < 4; u++) d0[u] = d1[u];
	}
}

// ------------------------------------------------------------------------------------------------ long lists: (candidate, round) items
// k_pair_ranks_1xm gives a candidate to ONE wave. With lists of tens of thousands of k-mers and windows of a few thousand candidates
// (BASELINE cfg5: sequences of 500 b - 50 kb) that is forty dependent loads per wave and most of the chip idle. Here the unit of work is a
// ROUND of a candidate -- kRiRound consecutive entries of its rank list -- and any wave takes any round. r05: the walk knows no runs.
//   * Every FIRST copy of a bin (an entry that differs from the one in front of it: the round reads one entry before its own) is accounted
//     for as if the candidate held the bin ONCE: the query's e_q from the two-bit table in LDS, and a packed counter per value of it --
//     twelve instructions per entry, no branch but the rare "three and more" (a hash table in LDS of the query's few such bins).
//     sum e_c e_q, sum min(e_c, e_q) and the divergence cells (1, e_q) of the round are sums of those counters.
//   * What a bin held MORE than once adds to that comes from the candidate's list of REPEATED bins -- (bin, e_c) for e_c >= 2, one in
//     twenty of a random 25 kb sequence's bins, a homopolymer run of thousands of copies ONE entry -- in items of their own, kRiMulti
//     entries each, every lane at work: (e_c - 1) e_q, min(e_c, e_q) - min(1, e_q), the bin moved from cell (1, e_q) to (e_c, e_q).
//     (Until r05 every bin was accounted for at its LAST copy with its copy index from a prefix maximum over the round and a binary search
//     for runs that began before it: ~60 instructions per entry and lane, 0.08 of HBM.)
//   * a round's integer sums go into its candidate's accumulators with atomics (integers: any order), an item's divergence spot terms into a
//     slot of its own; k_rank_items_finish turns accumulators, cells and slots into the candidate's records in a fixed order.
constexpr uint32_t kRiRound = 1024;
constexpr uint32_t kRiMulti = 256;          // entries of a candidate's repeated-bin list per item (at most 2 x rounds items: a repeated bin is two k-mers and more)
constexpr uint32_t kRiHash = 1024;          // slots of the LDS hash of the query's bins with e_q >= 3 (at most half are used)
constexpr uint32_t kRiMultiFlag = 0x80000000u;
constexpr uint32_t kRiQueue = 512;          // pairs of counts a wave puts aside before it evaluates them (256 entries of the list bring at most 256; emptied before the next 256 would not fit)
constexpr uint32_t kRiListCands = 64;           // candidates per workgroup of k_rank_pass_prep's list part
constexpr uint32_t kRiPrepBlocks = 32;          // workgroups of k_rank_pass_prep that take the query's side (one entry per thread for lists of up to 32 768 stored bins)

// n[slot] = its stored bins with two and more counted k-mers (value >= 3: every bin starts at 1); one wave per slot
__global__ void __launch_bounds__(256) k_rkm_sizes(const uint2* __restrict__ ent, const MscSparseHdr* __restrict__ hdr, uint64_t capacity, uint32_t* __restrict__ n) {
	const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const uint32_t lane = threadIdx.x & 63;
	if (s >= capacity) return;
	const MscSparseHdr h = hdr[s];
	uint32_t cnt = 0;
	for (uint32_t j = lane; j < h.nnz; j += 64) cnt += ent[h.off + j].y >= 3u ? 1u : 0u;
	cnt = wave_total_u32(cnt);
	if (lane == 0) n[s] = cnt;
}
// ... and the list itself, (bin, e = value - 1) in bin order (the lanes' places from a ballot: the same list on every run)
__global__ void __launch_bounds__(256) k_rkm_fill(const uint2* __restrict__ ent, const MscSparseHdr* __restrict__ hdr, uint64_t capacity, const uint64_t* __restrict__ off,
                                                  uint2* __restrict__ out) {
	const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const uint32_t lane = threadIdx.x & 63;
	if (s >= capacity) return;
	const MscSparseHdr h = hdr[s];
	uint2* o = out + off[s];
	uint32_t base = 0;
	for (uint32_t j0 = 0; j0 < h.nnz; j0 += 64) {
		const uint32_t j = j0 + lane;
		uint2 en = make_uint2(0u, 0u);
		if (j < h.nnz) en = ent[h.off + j];
		const bool keep = en.y >= 3u;
		const uint64_t mask = __ballot(keep);
		if (keep) o[base + __popcll(mask & ((1ull << lane) - 1ull))] = make_uint2(en.x, en.y - 1u);
		base += (uint32_t)__popcll(mask);
	}
}

// The items of a pass, as a list: a candidate's rounds and repeated-bin items are counted (0 for a candidate the length window drops) and
// written out -- the pass then walks real items only (with the host's bound, rounds per candidate = that of the LONGEST list of the set: on
// mixed lengths four items in five were empty, and finding that out cost a wave three dependent loads each). Any order will do (integer
// atomics, a slot per item for what is FP64), so every workgroup claims its candidates' places with one atomic.
struct RkItemMeta { uint64_t off, moff; double mag; uint32_t n, nm, rounds, mrounds, first, pad; };          // first: the candidate's place in the list of items
// the query's side of a pass (its rank list when it has none in place, its counts of counts, its tables: blocks [0, kRiPrepBlocks)) and the list of
// items (the blocks behind them) in one launch
__global__ void __launch_bounds__(1024) k_rank_pass_prep(const uint2* __restrict__ ent, const uint32_t* __restrict__ cum, const MscSparseHdr* __restrict__ hdr_p, uint32_t nbins,
                                                         uint32_t* __restrict__ out, uint32_t cap, uint32_t* __restrict__ guard, uint32_t* __restrict__ hq, uint32_t* __restrict__ big,
                                                         RkItemMeta* __restrict__ meta, uint32_t m, uint32_t* __restrict__ n_items, uint2* __restrict__ items,
                                                         const uint64_t* __restrict__ c_off, const uint32_t* __restrict__ c_n, const uint64_t* __restrict__ m_off,
                                                         const uint32_t* __restrict__ m_n, const uint8_t* __restrict__ cand_scalars, uint64_t scalar_stride,
                                                         const uint32_t* __restrict__ cand_slots, uint64_t first, int use_window, uint64_t min_len, uint64_t max_len, uint32_t* __restrict__ zero_next,
                                                         uint32_t* __restrict__ tab, uint32_t* __restrict__ tab_next, uint32_t tab_words, uint32_t* __restrict__ n_big3) {
	extern __shared__ __attribute__((aligned(16))) uint32_t s_tab[];          // (the query's workgroups: up to the whole two-bit table)
	if (blockIdx.x == 0 && threadIdx.x < 16) zero_next[threadIdx.x] = 0u;          // the counters of the NEXT pass (two sets, used in turn)
	if (blockIdx.x < kRiPrepBlocks)          // ... and its tables
		for (uint32_t i = 4 * (blockIdx.x * blockDim.x + threadIdx.x); i < tab_words; i += 4 * kRiPrepBlocks * blockDim.x) *reinterpret_cast<uint4*>(tab_next + i) = make_uint4(0u, 0u, 0u, 0u);
	const MscSparseHdr h = *hdr_p;
	const uint32_t tot = h.nnz ? cum[h.off + h.nnz - 1] : 0u, pad = (tot + 255u) & ~255u;
	if (blockIdx.x < kRiPrepBlocks) {
		if (out && pad > cap) {          // the set's bound did not hold: say so and write nothing
			if (blockIdx.x == 0 && threadIdx.x == 0 && guard) atomicOr(guard, 1u);
			return;
		}
		uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		// A workgroup takes 1 024 consecutive entries at a time: their words of the two-bit table are put together in LDS and written out
		// once each -- the entries are in bin order, so a word belongs to one workgroup but for the first and the last of its span (those
		// two by atomics). (One global atomic per entry, eighty of them aimed at the same 128-byte line at once, was 8 .. 20 us of a pass.)
		for (uint32_t j0 = blockIdx.x * blockDim.x; j0 < h.nnz; j0 += kRiPrepBlocks * blockDim.x) {
			const uint32_t j = j0 + threadIdx.x, j_last = (j0 + blockDim.x < h.nnz ? j0 + blockDim.x : h.nnz) - 1;
			const uint32_t w_lo = ent[h.off + j0].x >> 4, w_hi = ent[h.off + j_last].x >> 4, span = w_hi - w_lo + 1;
			for (uint32_t i = threadIdx.x; i < span; i += blockDim.x) s_tab[i] = 0u;
			__syncthreads();
			uint2 en = make_uint2(0u, 0u);
			if (j < h.nnz) en = ent[h.off + j];
			const uint32_t e = en.y ? en.y - 1u : 0u;
			if (j < h.nnz) {
				if (out) { const uint32_t end = cum[h.off + j]; for (uint32_t t = end - e; t < end; t++) out[t] = en.x; }          // (out == nullptr: the query's list is one of the set's own)
				// the query's histogram as the pass's workgroups will hold it in LDS: two bits per bin -- e_q = 0, 1, 2, "three and more" --
				if (e >= 1) atomicOr(&s_tab[(en.x >> 4) - w_lo], (e >= 3 ? 3u : e) << (2 * (en.x & 15)));
			}
			// ... and, behind them, a hash table of the few bins of the last kind (key = bin + 1: the tables start out zero). Places are claimed
			// a wave at a time: a hundred atomics with a return on ONE word, one behind the other, were 10 us of a long query's pass.
			{
				const uint32_t ln = threadIdx.x & 63;
				const uint64_t m3 = __ballot(e >= 3);
				if (m3) {
					uint32_t base = 0;
					if (ln == (uint32_t)__builtin_ctzll(m3)) base = atomicAdd(n_big3, (uint32_t)__popcll(m3));
					base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m3));
					if (e >= 3 && base + (uint32_t)__popcll(m3 & ((1ull << ln) - 1ull)) < kRiHash / 2) {          // (more than that: the look-up falls back to the query's rank list)
						uint32_t *keys = tab + (tab_words - 2 * kRiHash), *vals = keys + kRiHash;
						uint32_t hsh = (en.x * 2654435761u) >> 22;
						for (;;) {
							const uint32_t old = atomicCAS(&keys[hsh], 0u, en.x + 1u);
							if (old == 0u || old == en.x + 1u) { vals[hsh] = e; break; }
							hsh = (hsh + 1) & (kRiHash - 1);
						}
					}
				}
				if (hq) {
#pragma unroll
					for (uint32_t x = 2; x < 8; x++) cnt[x] += en.y == x ? 1u : 0u;
					const uint64_t m8 = __ballot(en.y >= 8);
					if (m8) {
						uint32_t base = 0;
						if (ln == (uint32_t)__builtin_ctzll(m8)) base = atomicAdd(&hq[8], (uint32_t)__popcll(m8));
						base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m8));
						if (en.y >= 8) big[base + (uint32_t)__popcll(m8 & ((1ull << ln) - 1ull))] = en.y;
					}
				}
			}
			__syncthreads();
			for (uint32_t i = threadIdx.x; i < span; i += blockDim.x) {
				const uint32_t v = s_tab[i];
				if (!v) continue;
				if (i == 0 || i == span - 1) atomicOr(&tab[w_lo + i], v);
				else tab[w_lo + i] = v;
			}
			__syncthreads();
		}
		if (blockIdx.x == 0 && out) for (uint32_t i = tot + threadIdx.x; i < pad; i += blockDim.x) out[i] = nbins;
		if (hq) {
#pragma unroll
			for (uint32_t x = 2; x < 8; x++) {
				const uint32_t t = (uint32_t)wave_sum_u64(cnt[x]);
				if ((threadIdx.x & 63) == 0 && t) atomicAdd(&hq[x], t);
			}
		}
		return;
	}
	// the list: kRiListCands candidates per workgroup -- its first wave reads their records, counts their items and claims the places (a
	// DPP scan, one atomic), then every wave writes the items of every sixteenth candidate, 64 at a time (r05: with a candidate per
	// thread, a thousand threads of ONE workgroup wrote up to sixty scattered items each -- 7 .. 29 us, as long as the pass itself)
	__shared__ uint32_t s_first[kRiListCands], s_rounds[kRiListCands], s_mrounds[kRiListCands];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t c_base = (blockIdx.x - kRiPrepBlocks) * kRiListCands;
	if (wave == 0) {
		const uint32_t c = c_base + lane;
		RkItemMeta mt{0, 0, 0.0, 0, 0, 0, 0, 0, 0};
		if (c < m) {
			const uint64_t slot = cand_slots ? (uint64_t)cand_slots[c] : first + c;
			const MscSlotScalars* cs = reinterpret_cast<const MscSlotScalars*>(cand_scalars + (cand_slots ? slot : (uint64_t)c) * scalar_stride);
			const uint64_t len = cs->length;
			mt.mag = (double)cs->mag;
			mt.off = c_off[slot]; mt.n = c_n[slot];
			mt.moff = m_off[slot]; mt.nm = m_n[slot];
			if (!(use_window && (len < min_len || len > max_len))) {
				const uint32_t T = mt.n > tot ? mt.n : tot;
				mt.rounds = (T + kRiRound - 1) / kRiRound;
				mt.mrounds = (mt.nm + kRiMulti - 1) / kRiMulti;
			}
		}
		const uint32_t mine = mt.rounds + mt.mrounds;
		const uint32_t incl = wave_incl_scan(mine);
		uint32_t base = 0;
		if (lane == 63 && incl) base = atomicAdd(n_items, incl);
		base = (uint32_t)__builtin_amdgcn_readlane((int)base, 63);
		mt.first = base + incl - mine;
		if (c < m) meta[c] = mt;
		s_first[lane] = mt.first; s_rounds[lane] = mt.rounds; s_mrounds[lane] = mt.mrounds;
	}
	__syncthreads();
	for (uint32_t i = wave; i < kRiListCands; i += kRpBlock / 64) {
		const uint32_t c = c_base + i, r = s_rounds[i], mr = s_mrounds[i];
		uint2* o = items + s_first[i];
		for (uint32_t rd = lane; rd < r + mr; rd += 64) o[rd] = make_uint2(c, rd < r ? rd : (rd - r) | kRiMultiFlag);
	}
}

template <bool DIV>
__global__ void __launch_bounds__(kRpBlock) k_pair_ranks_items(const uint32_t* __restrict__ c_rk, const uint2* __restrict__ c_rm, uint32_t m, const uint2* __restrict__ q_ent,
                                                                const uint32_t* __restrict__ q_cum, const MscSparseHdr* __restrict__ q_hdr_p, uint32_t nbins,
                                                                const uint32_t* __restrict__ rq, uint32_t* __restrict__ rec, double* __restrict__ extras, const uint8_t* __restrict__ q_scalars, int order, const RkItemMeta* __restrict__ meta,
                                                                const uint2* __restrict__ items, const uint32_t* __restrict__ cnt_p, const uint32_t* __restrict__ tab, uint32_t* __restrict__ big) {
	extern __shared__ __attribute__((aligned(16))) uint32_t s_rp[];          // [two bits per bin: nbins / 16 + 1 words, padded to four][hash keys][hash values]
	__shared__ uint32_t s_cnt[DIV ? kRpBlock / 64 : 1][DIV ? kRkCells : 1];          // DIV: a wave's cell counts of the item in hand, those no register counts
	__shared__ uint2 s_q[DIV ? kRpBlock / 64 : 1][DIV ? kRiQueue : 1];               // DIV: the pairs of counts whose terms are evaluated on the spot, put aside (below)
	const uint32_t words_pad = (nbins / 16 + 4) & ~3u;
	uint32_t *sb = s_rp, *s_key = s_rp + words_pad, *s_val = s_key + kRiHash;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const MscSparseHdr qh = *q_hdr_p;
	const uint32_t nq = qh.nnz;
	const uint32_t nq_tot = nq ? q_cum[qh.off + nq - 1] : 0u;
	const uint32_t nq_pad4 = (nq_tot + 3u) & ~3u;          // (the query's list is padded with 4^k to four entries at least: the set's own list, or the pass's expansion)
	const bool hash_ok = cnt_p[13] <= kRiHash / 2;
	auto e_q_of = [&](uint32_t bin) -> uint32_t {          // a bin the table marks "three and more"
		if (hash_ok) {
			uint32_t h = (bin * 2654435761u) >> 22;
			while (s_key[h] != bin + 1u) h = (h + 1) & (kRiHash - 1);
			return s_val[h];
		}
		const uint32_t lo = lower_bound_u32([&](uint32_t i) { return rq[i]; }, nq_tot, bin);
		return lower_bound_u32([&](uint32_t i) { return rq[i]; }, nq_tot, bin + 1) - lo;
	};
	double qm = 0.0;
	if constexpr (DIV) {
		qm = (double)reinterpret_cast<const MscSlotScalars*>(q_scalars)->mag;
		s_cnt[wave][lane] = 0u;
	}
	const uint32_t n_items = cnt_p[12];
	(void)m; (void)q_ent;
	// SPOT TERMS. A bin the candidate holds 8 times and more, or the query 7 and more, has no cell: its term is two FP64 logarithms on the
	// spot. Evaluated where they turn up -- one lane in sixty-four, four times over in an unrolled loop -- they were a third of the kernel
	// on cfg5's shape (runs of thousands of copies in every sequence) and the reason some waves took three times as long as others. The
	// walk only puts the pair of counts aside, at a place a ballot gives (the same on every run); when the item is done -- or 256 are in --
	// every lane takes one: term(ca, cb) - term(lo, cb), lo = 1 (b >= 8: the finish adds F(1, b) for every such bin of the query),
	// 2 (a repeated bin's correction) or, for b < 8, - term(1, 1).
	uint32_t qn = 0;          // (uniform) pairs in the wave's queue
	double xjd = 0.0, xjs = 0.0;
	double cm = 0.0;
	auto q_push = [&](bool want, uint32_t ca, uint32_t cb, uint32_t lo) {          // called by all lanes of the wave at once
		if constexpr (DIV) {
			const uint64_t mask = __ballot(want);
			if (want) s_q[wave][qn + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = make_uint2(ca | (lo << 30), cb);
			qn += (uint32_t)__popcll(mask);
		}
	};
	auto q_drain = [&]() {
		if constexpr (DIV) {
			if (qn == 0) return;
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__builtin_amdgcn_wave_barrier();
			RkDivTerm t11{0.0, 0.0};
			bool have11 = false;
			for (uint32_t b0 = 0; b0 < qn; b0 += 64) {
				const bool mine = b0 + lane < qn;
				const uint2 en = mine ? s_q[wave][b0 + lane] : make_uint2(1u, 1u);
				const uint32_t ca = en.x & 0x3fffffffu, lo = en.x >> 30, cb = en.y;
				if (!have11 && __ballot(mine && cb < 8)) { t11 = rk_div_term_call(1, 1, cm, qm, order); have11 = true; }
				if (mine) {
					const RkDivTerm hi = rk_div_term_call(ca, cb, cm, qm, order);
					RkDivTerm l = t11;
					if (cb >= 8) l = rk_div_term_call(lo, cb, cm, qm, order);
					xjd += hi.jd - l.jd;
					xjs += hi.js - l.js;
				}
			}
			qn = 0;
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
	};
	// The loop is laid out around the memory pipe's one counter (loads and the acknowledgements of atomics and stores are waited for in
	// issue order): an item's loads go out BEFORE the flush of the item in front of it -- into the registers that item has just finished
	// with --, so that the wait for them does not include the round trip of atomics that twenty waves aim at the same candidate; the list
	// of items is read three ahead, the candidates' records two ahead (r05: with the wait for the next record at the loop's head, then the
	// loads, then the flush, an item cost its wave 11 us whatever it held).
	auto issue = [&](const uint2 item, const RkItemMeta& mt, uint4 (&a)[4], uint4 (&b)[4], uint32_t& before_round) {
		const uint32_t rd = item.y & ~kRiMultiFlag;
		if (!(item.y & kRiMultiFlag)) {
			const uint32_t nc = mt.n, nc_pad = (nc + 3u) & ~3u;
			const uint32_t t0 = rd * kRiRound;
			const uint32_t* P = c_rk + mt.off;
#pragma unroll
			for (uint32_t u = 0; u < 4; u++) {
				const uint32_t t = t0 + 256 * u + 4 * lane;
				a[u] = make_uint4(nbins, nbins, nbins, nbins);
				b[u] = a[u];
				if (t < nc_pad) a[u] = *reinterpret_cast<const uint4*>(P + t);
				if (t < nq_pad4) b[u] = *reinterpret_cast<const uint4*>(rq + t);
			}
			before_round = t0 && t0 - 1 < nc ? P[t0 - 1] : 0xffffffffu;          // the entry in front of the round
		} else {          // repeated bins [rd * kRiMulti, ..) of the candidate: four per lane, (bin, e_c) each
			const uint2* M = c_rm + mt.moff;
			const uint32_t j0 = rd * kRiMulti + 4 * lane;
			a[0] = make_uint4(0u, 0u, 0u, 0u);
			a[1] = a[0];
			if (j0 < mt.nm) a[0] = *reinterpret_cast<const uint4*>(M + j0);
			if (j0 + 2 < mt.nm) a[1] = *reinterpret_cast<const uint4*>(M + j0 + 2);
		}
	};
	// WHO takes WHICH item. The workgroup owns a contiguous share of the list, [s0, s1); its sixteen waves take the share's items as they
	// come free, by a counter in LDS -- three ahead, as the list is read (a wave's first three are its own by position). r05, until the
	// end of the round every wave took items w, w + 4 096, ..: the same NUMBER each, and the slowest wave of a launch left 31 us behind a
	// mean of 23 (items differ: rare bins, spot terms, repeated-bin items) -- the chip a quarter idle. Records are written by item, so
	// nothing depends on who walked what.
	__shared__ uint32_t s_ticket;
	const uint32_t s0 = (uint32_t)((uint64_t)n_items * blockIdx.x / gridDim.x), s1 = (uint32_t)((uint64_t)n_items * (blockIdx.x + 1) / gridDim.x);
	constexpr uint32_t kWaves = kRpBlock / 64;
	if (threadIdx.x == 0) s_ticket = s0 + 3 * kWaves;          // (the barrier behind the tables' copy stands between this and the first ticket)
	uint32_t it = s0 + wave, it_1 = it + kWaves, it_2 = it + 2 * kWaves;
	uint2 item_n = it < s1 ? items[it] : make_uint2(0u, 0u);
	uint2 item_n2 = it_1 < s1 ? items[it_1] : make_uint2(0u, 0u);
	uint2 item_n3 = it_2 < s1 ? items[it_2] : make_uint2(0u, 0u);
	RkItemMeta mt_n = meta[item_n.x];
	RkItemMeta mt_n2 = meta[item_n2.x];
	uint4 a[4], b[4];
	uint32_t before_round = 0xffffffffu;
	uint32_t it_3 = 0;
	if (it < s1) issue(item_n, mt_n, a, b, before_round);
	// The query's counts >= 8 were appended in whatever order k_rank_pass_prep's waves came by; the finish kernel adds an FP64 term per
	// entry in list order, and a sum must not depend on that: the first workgroup sorts the list (a bitonic network in the LDS the
	// tables are about to take; lists longer than that -- 16 384 at k = 9 -- stay as they are). Nothing in this kernel reads it.
	if constexpr (DIV) {
		if (blockIdx.x == 0) {
			const uint32_t nb = cnt_p[8];
			uint32_t p2 = 2;
			while (p2 < nb && p2 < (1u << 20)) p2 <<= 1;
			if (nb > 1 && p2 >= nb && p2 <= words_pad) {
				for (uint32_t i = threadIdx.x; i < p2; i += kRpBlock) s_rp[i] = i < nb ? big[i] : 0xffffffffu;
				__syncthreads();
				for (uint32_t k2 = 2; k2 <= p2; k2 <<= 1)
					for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
						for (uint32_t i = threadIdx.x; i < p2; i += kRpBlock) {
							const uint32_t l = i ^ j2;
							if (l > i) {
								const uint32_t x = s_rp[i], y = s_rp[l];
								if (((i & k2) == 0) == (x > y)) { s_rp[i] = y; s_rp[l] = x; }
							}
						}
						__syncthreads();
					}
				for (uint32_t i = threadIdx.x; i < nb; i += kRpBlock) big[i] = s_rp[i];
				__syncthreads();
			}
		}
	}
	// the query's tables, as k_rank_pass_prep left them in global memory (r05: every workgroup used to build them from the query's list -- 16 000
	// clocks of a kernel of 47 000); the wave's first loads are on their way meanwhile
	for (uint32_t i = 4 * threadIdx.x; i < words_pad + 2 * kRiHash; i += 4 * kRpBlock) *reinterpret_cast<uint4*>(s_rp + i) = *reinterpret_cast<const uint4*>(tab + i);
	__syncthreads();
	for (; it < s1; it = it_1, it_1 = it_2, it_2 = it_3) {
		const uint2 item = item_n;
		const RkItemMeta mt = mt_n;
		item_n = item_n2; mt_n = mt_n2; item_n2 = item_n3;
		mt_n2 = meta[item_n2.x];          // (its item arrived an iteration ago)
		uint32_t tk = 0;
		if (lane == 0) tk = atomicAdd(&s_ticket, 1u);
		it_3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
		if (it_3 < s1) item_n3 = items[it_3];
		const bool has_next = it_1 < s1;
		const uint32_t rd = item.y & ~kRiMultiFlag;
		const bool multi = (item.y & kRiMultiFlag) != 0;
		cm = mt.mag;
		uint32_t emd = 0;
		uint64_t prod = 0;          // prod, mins: what the packed counters do not hold
		uint32_t mins = 0;
		uint32_t pk_a = 0, pk_b = 0;          // two pairs of 16-bit counters (see below)
		xjd = 0.0; xjs = 0.0;
		uint32_t spurious = 0;
		bool cnt_dirty = multi;          // (uniform) the wave's LDS cell counts were touched: a repeated-bin item nearly always, a round only through its rare bins
		if (!multi) {
			const uint32_t nc = mt.n;
			const uint32_t T = nc > nq_tot ? nc : nq_tot;
			const uint32_t t0 = rd * kRiRound;
			// the padding behind the list's end reads as ONE first copy of a bin the query does not hold (4^k: the zero word behind the table)
			const uint32_t walked = T - t0 >= kRiRound ? kRiRound : (T - t0 + 255u) & ~255u;
			spurious = nc < t0 + walked ? 1u : 0u;
			uint32_t pk = 0;          // four byte counters: first copies with e_q = 0, 1, 2, three and more
			const uint8_t* sb8 = reinterpret_cast<const uint8_t*>(sb);          // (a byte of the table: four bins)
#pragma unroll
			for (uint32_t u = 0; u < 4; u++) {
				if (t0 + 256 * u >= T) break;
				const uint4 av = a[u], bv = b[u];
				uint32_t d = sad_u32(av.x, bv.x, 0u);
				d = sad_u32(av.y, bv.y, d);
				d = sad_u32(av.z, bv.z, d);
				d = sad_u32(av.w, bv.w, d);
				emd += d;
				uint32_t prev = lane_prev(av.w);
				if (lane == 0) prev = u ? (uint32_t)__builtin_amdgcn_readlane((int)a[u ? u - 1 : 0].w, 63) : before_round;
				const uint32_t e4[4] = {av.x, av.y, av.z, av.w};
				const uint32_t p4[4] = {prev, av.x, av.y, av.z};
				uint32_t two[4];
				const uint32_t pk0 = pk;
#pragma unroll
				for (int j = 0; j < 4; j++) {          // ten instructions per entry: a compare, the look-up (five), a shift and a shifted add, the round's |a - b|
					const uint32_t bin = e4[j];
					two[j] = __builtin_amdgcn_ubfe((uint32_t)sb8[bin >> 2], (bin & 3u) << 1, 2u);
					pk += (bin != p4[j] ? 1u : 0u) << (two[j] << 3);
				}
				if (__ballot((pk - pk0) >> 24)) {          // rare: among the four of some lane, the first copy of a bin the query holds three times and more
#pragma unroll
					for (int j = 0; j < 4; j++) {
						const uint32_t bin = e4[j];
						const bool hit = bin != p4[j] && two[j] == 3u;
						uint32_t cb = 0;
						if (hit) {
							const uint32_t e_q = e_q_of(bin);
							prod += e_q;
							mins += 1u;
							cb = e_q + 1;
							if constexpr (DIV) { if (cb < 8) atomicAdd(&s_cnt[wave][cb], 1u); }
						}
						q_push(hit && cb >= 8, 2u, cb, 1u);
					}
					cnt_dirty = true;
					if (qn > kRiQueue - 256) q_drain();
				}
			}
			pk_a = (pk & 255u) | (((pk >> 8) & 255u) << 16);
			pk_b = (pk >> 16) & 255u;
		} else {
			const uint32_t j0 = rd * kRiMulti + 4 * lane;
			const uint4 m0 = a[0], m1 = a[1];
			const uint32_t mb[4] = {m0.x, m0.z, m1.x, m1.z}, me[4] = {m0.y, m0.w, m1.y, m1.w};
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const bool live = j0 + i < mt.nm;
				uint32_t ca = 0, cb = 0;
				if (live) {
					const uint32_t bin = mb[i], e_c = me[i];
					const uint32_t two = __builtin_amdgcn_ubfe(sb[bin >> 4], 2u * (bin & 15u), 2u);
					const uint32_t e_q = two < 3 ? two : e_q_of(bin);
					prod += (uint64_t)(e_c - 1u) * e_q;
					mins += (e_c < e_q ? e_c : e_q) - (e_q ? 1u : 0u);
					ca = e_c + 1; cb = e_q + 1;          // the bin's counts in the candidate and in the query
					if constexpr (DIV) {
						if (cb < 8) {
							// out of cell (1 copy, cb), into cell (e_c copies, cb); [row 7: counted as held, its term among the spot terms]
							if (cb <= 2) pk_a += cb == 1 ? 1u : 0x10000u;
							else atomicAdd(&s_cnt[wave][cb], 0xffffffffu);
							if (ca == 3 && cb <= 2) pk_b += cb == 1 ? 1u : 0x10000u;
							else atomicAdd(&s_cnt[wave][(ca < 9 ? ca - 2 : 7) * 8 + cb], 1u);
						}
					}
				}
				q_push(live && (ca >= 9 || cb >= 8), ca, cb, 2u);
			}
		}
		if (has_next) issue(item_n, mt_n, a, b, before_round);          // the next item's loads, in front of this item's spot terms and flush
		q_drain();
		// the item's three totals in one pass over the wave (msc_wave.h: four sums for the price of ~1.7)
		const uint32_t rows = wave_sum4_rows(pk_a, pk_b, emd, 0u);
		const uint32_t ta = MSC_ROW_A(rows), tb = MSC_ROW_B(rows);
		uint64_t prod_t = 0;
		uint32_t mins_t = 0;
		uint64_t emd_t = 0;
		if (__ballot(prod != 0 || mins != 0)) { prod_t = wave_sum_u64(prod); mins_t = (uint32_t)wave_sum_u64(mins); }
		uint32_t v1, v2, v3 = 0, v9 = 0, v10 = 0;          // what the counters add to cells (0, 1), (0, 2), (0, 3), (1, 1), (1, 2) (modulo 2^32: the repeated bins take away)
		if (!multi) {
			const uint32_t c01 = (ta & 0xffffu) - spurious, c02 = ta >> 16, c03 = tb;
			v1 = c01; v2 = c02; v3 = c03;
			prod_t += c02 + 2 * (uint64_t)c03;
			mins_t += c02 + c03;
			emd_t = MSC_ROW_C(rows);          // (a round's sum: 1 024 x 4^k <= 2^28 up to k = 9, the kernel's bound)
		} else {
			v1 = 0u - (ta & 0xffffu); v2 = 0u - (ta >> 16);
			v9 = tb & 0xffffu; v10 = tb >> 16;
		}
		// The item's RECORD, 64 words, one per lane, at the item's place in the list: the cell counts (r, b) at r * 8 + b -- b = 0 is no cell:
		// words 0 and 32 hold the halves of the round's emd, 8 and 16 those of sum e_c e_q, 24 sum min(e_c, e_q). No atomics: until r05 the items of a
		// candidate added into ITS accumulators and cells, twenty waves at the same words at the same time, and that wait was half the kernel.
		uint32_t v = 0;
		if constexpr (DIV) {
			wave_sum_f64_pair(xjd, xjs);
			if (cnt_dirty) {
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
				__builtin_amdgcn_wave_barrier();
				v = s_cnt[wave][lane];
				s_cnt[wave][lane] = 0u;
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
		}
		v += lane == 1 ? v1 : lane == 2 ? v2 : lane == 3 ? v3 : lane == 9 ? v9 : lane == 10 ? v10 : 0u;
		v = lane == 0 ? (uint32_t)emd_t : lane == 32 ? (uint32_t)(emd_t >> 32) : lane == 8 ? (uint32_t)prod_t : lane == 16 ? (uint32_t)(prod_t >> 32) : lane == 24 ? mins_t : v;
		rec[(uint64_t)it * kRkCells + lane] = v;
		if constexpr (DIV) {
			if (lane == 0) *reinterpret_cast<double2*>(extras + 2 * (uint64_t)it) = make_double2(xjd, xjs);
		}
	}
}

// A WAVE per candidate (four to a workgroup): lane l adds up word l of its items' records (rounds, then repeated-bin items, as the list
// holds them; eight records in flight); the integer record from words 0 / 32 / 8 / 16 / 24; DIV: lane (r, b) evaluates cell (r, b) -- row
// a' = candidate count a' + 2, row 7 = bins evaluated on the spot, counted as held: its lanes take the query's bins with count b that the
// candidate does not hold --, every lane its share of the query's bins with a count >= 8 and of the items' spot terms (lane l: entries l,
// l + 64, ..); one butterfly adds the 64 partial sums in a fixed order. (r05: a WORKGROUP per candidate, its four waves sharing the
// records, was 34 us per pass of 8 000 candidates at cfg5's full size -- three of four waves idle through the FP64 part, five rounds of
// workgroups over the chip; a candidate has eight or nine items.)
template <bool DIV>
__global__ void __launch_bounds__(256) k_rank_items_finish(const uint32_t* __restrict__ rec, const double* __restrict__ extras, uint32_t rounds, const uint32_t* __restrict__ hq,
                                                           const uint32_t* __restrict__ big, const RkItemMeta* __restrict__ meta, uint32_t m, const uint32_t* __restrict__ q_cum,
                                                           const MscSparseHdr* __restrict__ q_hdr_p, const uint8_t* __restrict__ q_scalars, int order, MscPartial* __restrict__ partials,
                                                           double* __restrict__ div_out, uint32_t* __restrict__ guard) {
	const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (c >= m) return;
	const RkItemMeta mt = meta[c];
	// (what does not hang on the candidate's record goes out beside it: the wave is one chain of round trips, and the chip holds all of a
	// pass's waves at once or nearly -- the kernel's time is the chain's)
	const MscSparseHdr qh = *q_hdr_p;
	const uint32_t hq_b = DIV ? hq[lane & 7] : 0u, n_big = DIV ? hq[8] : 0u;
	const double qm = DIV ? (double)reinterpret_cast<const MscSlotScalars*>(q_scalars)->mag : 0.0;
	const uint64_t nq_tot = qh.nnz ? q_cum[qh.off + qh.nnz - 1] : 0u;
	const uint32_t n_it = mt.rounds + mt.mrounds;
	const uint32_t* R = rec + (uint64_t)mt.first * kRkCells + lane;
	uint64_t sum = 0;
	// sixteen records in flight, a short tail's from a clamped place (r05: the records behind the last full eight came one by one)
	for (uint32_t s = 0; s < n_it; s += 16) {
		uint32_t x[16];
#pragma unroll
		for (uint32_t i = 0; i < 16; i++) x[i] = R[(uint64_t)(s + i < n_it ? s + i : n_it - 1) * kRkCells];
#pragma unroll
		for (uint32_t i = 0; i < 16; i++) sum += s + i < n_it ? x[i] : 0u;
	}
	double2 ex0 = make_double2(0.0, 0.0);          // the lane's first pair of spot-term sums (a candidate seldom has more than 64 items)
	if constexpr (DIV) { if (n_it) ex0 = (reinterpret_cast<const double2*>(extras) + mt.first)[lane < n_it ? lane : n_it - 1]; }
	{
		const uint64_t emd = sum + (__shfl(sum, 32, 64) << 32), p_lo = __shfl(sum, 8, 64), p_hi = __shfl(sum, 16, 64), mins = __shfl(sum, 24, 64);
		if (lane == 0) {
			const uint64_t nc = mt.n;
			// a list longer than the pass's bound: the host fails the call (a candidate the length window dropped took no round: any length)
			const bool dropped = mt.rounds == 0;
			if ((!dropped && nc > (uint64_t)rounds * kRiRound) || nq_tot > (uint64_t)rounds * kRiRound) atomicOr(guard, 1u);
			MscPartial out;
			out.manh = nc + nq_tot - 2 * mins;
			out.dot = p_lo + (p_hi << 32) + nc + nq_tot;
			out.emd = emd;
			partials[c] = out;
		}
	}
	if constexpr (DIV) {
		const double cm = mt.mag;
		const uint32_t b = lane & 7, r = lane >> 3;
		const uint32_t k = b ? (uint32_t)sum : 0u;          // the cell's count (modulo 2^32, as the items wrote it)
		uint32_t held = k;                                  // the candidate's bins the query holds b - 1 times: the column's sum
		held += __shfl_xor(held, 8, 64);
		held += __shfl_xor(held, 16, 64);
		held += __shfl_xor(held, 32, 64);
		// ONE evaluation per wave, a pair of counts per lane: lane 0 term(1, 1); lane (r, b), r < 7: term(r + 2, b) for its cell; lane (7, b): term(1, b)
		// for the query's bins with count b that the candidate does not hold (three calls one behind the other -- the FP64 logarithms of this
		// kernel -- made it 30 us per pass of 8 000 candidates at cfg5's full size)
		uint32_t n_of = 0, a1 = 1, a2 = 1;
		if (b >= 1 && r < 7) { n_of = k; a1 = r + 2; a2 = b; }
		if (b >= 2 && r == 7) { n_of = hq_b - held; a2 = b; }
		RkDivTerm f{0.0, 0.0};
		if (lane == 0 || n_of) f = rk_div_term_call(a1, a2, cm, qm, order);
		const RkDivTerm t11{__shfl(f.jd, 0, 64), __shfl(f.js, 0, 64)};
		double jd = 0.0, js = 0.0;
		if (n_of) {
			jd = (double)n_of * (f.jd - t11.jd);
			js = (double)n_of * (f.js - t11.js);
		}
		for (uint32_t i = lane; i < n_big; i += 64) {
			const RkDivTerm f = rk_div_term_call(1, big[i], cm, qm, order);
			jd += f.jd - t11.jd;
			js += f.js - t11.js;
		}
		const double2* x = reinterpret_cast<const double2*>(extras) + mt.first;
		if (lane < n_it) { jd += ex0.x; js += ex0.y; }
		for (uint32_t i = lane + 64; i < n_it; i += 64) { const double2 v = x[i]; jd += v.x; js += v.y; }
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) { jd += __shfl_xor(jd, off, 64); js += __shfl_xor(js, off, 64); }
		if (lane == 0) { div_out[2 * (uint64_t)c] = jd; div_out[2 * (uint64_t)c + 1] = js; }
	}
}

}  // namespace

// bytes of dynamic LDS the pass needs for 4^k = nbins and query lists of up to q_kmers entries; 0 when the histogram is too large for it
size_t msc_ranks_pass_lds(uint64_t nbins, uint64_t q_kmers) {
	if (nbins > (1ull << 18) || nbins % 32 || q_kmers > (1ull << 26)) return 0;
	const size_t words = nbins / 16 + 1;
	return (words + 4) * 4 + (q_kmers > kRpQCap ? 0 : std::max<size_t>(1024, ((size_t)q_kmers + 255) & ~(size_t)255) * 4);          // (a longer query's ranks stay in global memory; 1 024 entries at least: k_pair_ranks_1xm reads four chunks without asking)
}
// entries of global scratch a pass needs for the query's rank list (0: the list fits LDS)
uint64_t msc_ranks_pass_query_scratch(uint64_t q_kmers) { return q_kmers > kRpQCap ? ((q_kmers + 255) & ~255ull) : 0; }
uint32_t msc_ranks_pass_query_cap() { return 1u << 26; }

// sizes (n: capacity words) and offsets (off: capacity + 1 words) of the rank lists of a sparse set; the caller reads off[capacity], allocates, fills
hipError_t msc_launch_rank_lists_sizes(hipStream_t st, const MscSparseHdr* hdr, const uint32_t* cum, uint64_t capacity, uint32_t* n, uint64_t* off) {
	if (capacity == 0) return hipSuccess;
	k_rkl_sizes<<<dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, st>>>(hdr, cum, capacity, n);
	k_rkl_scan<<<dim3(1), dim3(1024), 0, st>>>(n, capacity, off);
	return hipGetLastError();
}
hipError_t msc_launch_rank_lists_fill(hipStream_t st, const void* ent, const uint32_t* cum, const MscSparseHdr* hdr, uint64_t capacity, const uint32_t* n, const uint64_t* off,
                                      uint64_t nbins, uint32_t* out) {
	if (capacity == 0) return hipSuccess;
	k_rkl_fill<<<dim3((unsigned)((capacity + 3) / 4)), dim3(256), 0, st>>>((const uint2*)ent, cum, hdr, capacity, n, off, (uint32_t)nbins, out);
	return hipGetLastError();
}

// candidates [first, first + m) (or the device slot list cand_slots; cand_scalars then is the set's base) against the query list
hipError_t msc_launch_pair_ranks_1xm(hipStream_t st, const uint32_t* c_rk, const uint64_t* c_off, const uint32_t* c_n, const uint8_t* cand_scalars, uint64_t scalar_stride,
                                     const uint32_t* cand_slots, uint64_t first, uint32_t m, const void* q_ent, const uint32_t* q_cum, const MscSparseHdr* q_hdr, uint64_t nbins,
                                     int use_window, uint64_t min_len, uint64_t max_len, MscPartial* partials, int num_cus, uint64_t q_kmers, uint32_t* guard, uint32_t* q_scratch) {
	if (m == 0) return hipSuccess;
	const size_t lds = msc_ranks_pass_lds(nbins, q_kmers);
	if (!lds) return hipErrorInvalidValue;
	const bool qg = q_kmers > kRpQCap;
	if (qg && !q_scratch) return hipErrorInvalidValue;
	static bool attr_set = false;
	if (!attr_set) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_ranks_1xm<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
		if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_ranks_1xm<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
		if (e != hipSuccess) return e;
		attr_set = true;
	}
	if (qg) k_rank_expand_one<<<dim3(8), dim3(1024), 0, st>>>((const uint2*)q_ent, q_cum, q_hdr, (uint32_t)nbins, q_scratch, (uint32_t)((q_kmers + 255) & ~255ull), guard);
	// as many workgroups as fit the chip at once (their tables take 64 KiB + the query's list of a CU's 160 KiB of LDS: two per CU for
	// 1 kb sequences at k = 9); fewer when the window is short: a workgroup's set-up is ~2 us
	// (r05: at 86 registers -- five waves per SIMD -- only one sixteen-wave workgroup is resident per CU, the second set of 256 follows the
	// first. Measured on 100 000 x 1 kb: 1 024 threads x 2 per CU 96.9 us, x 1 98.6; 768 x 2 114.5; 640 x 2 125; 512 x 2 99.7, x 3 121;
	// capped at 64 registers, two resident: 18 spilled, 158 us.)
	const uint32_t per_cu = (uint32_t)std::min<size_t>(2, (160 * 1024) / lds);
	const uint32_t block = kRpBlock;
	const uint32_t per_wg = block / 64;
	// (at least n candidates per wave, i.e. fewer workgroups for a short window, was measured on a window-bearing run,
	// 13 300 candidates per pass on average: 31.5 / 37.7 / 46.8 us per pass for n = 1 / 4 / 8 -- spreading wins)
	uint32_t blocks = (m + per_wg - 1) / per_wg;
	(void)per_cu;
	if (blocks > (uint32_t)num_cus) blocks = (uint32_t)num_cus;          // (one resident workgroup per CU: see above)
	const uint32_t q_cap = (uint32_t)((q_kmers + 255) & ~255ull);
	if (qg) k_pair_ranks_1xm<true><<<dim3(blocks), dim3(block), lds, st>>>(c_rk, c_off, c_n, cand_scalars, scalar_stride, cand_slots, first, m, (const uint2*)q_ent, q_cum, q_hdr, (uint32_t)nbins,
	                                                                           use_window, min_len, max_len, partials, q_cap, guard, q_scratch);
	else k_pair_ranks_1xm<false><<<dim3(blocks), dim3(block), lds, st>>>(c_rk, c_off, c_n, cand_scalars, scalar_stride, cand_slots, first, m, (const uint2*)q_ent, q_cum, q_hdr, (uint32_t)nbins,
	                                                                         use_window, min_len, max_len, partials, q_cap, guard, nullptr);
	return hipGetLastError();
}

// The pass over long lists: (candidate, round) items. rounds = rounds of kRiRound entries that cover the longest list involved (host
// bound); q_scratch: ((the query set's k-mer bound + 255) & ~255) entries; rec_scratch: msc_ranks_items_rec_bytes -- a 64-word record and
// a spot-term slot per item, nothing to clear; item_scratch: msc_ranks_items_list_bytes -- [m records][the items]; counters: 2 x 16 words,
// zero at the first pass -- the passes use the two halves in turn (`turn`) and each clears the other's. Of dv the launch takes big,
// q_scalars, order and div_out. q_rk: the query's own rank list when it is a slot of a set that holds them (no expansion then), or nullptr.
uint32_t msc_ranks_items_round() { return kRiRound; }
uint32_t msc_ranks_items_table_words(uint64_t nbins) { return (uint32_t)((nbins / 16 + 4) & ~3ull) + 2 * kRiHash; }          // one set of the query's tables (the caller holds two, zero at first)
size_t msc_ranks_items_rec_bytes(uint64_t m, uint32_t rounds) { return m * 3 * (size_t)rounds * (kRkCells * sizeof(uint32_t) + 2 * sizeof(double)) + 64; }
size_t msc_ranks_items_list_bytes(uint64_t m, uint32_t rounds) { return m * sizeof(RkItemMeta) + m * 3 * (size_t)rounds * sizeof(uint2) + 64; }
hipError_t msc_launch_pair_ranks_items(hipStream_t st, const uint32_t* c_rk, const uint64_t* c_off, const uint32_t* c_n, const void* c_rm, const uint64_t* c_rm_off,
                                       const uint32_t* c_rm_n, const uint8_t* cand_scalars, uint64_t scalar_stride, const uint32_t* cand_slots, uint64_t first, uint32_t m,
                                       const void* q_ent, const uint32_t* q_cum, const MscSparseHdr* q_hdr, uint64_t nbins, int use_window, uint64_t min_len, uint64_t max_len,
                                       MscPartial* partials, int num_cus, uint32_t* q_scratch, uint32_t rounds, void* rec_scratch, const MscRankDiv* dv, uint64_t q_kmers,
                                       uint32_t* guard, void* item_scratch, uint32_t* counters, uint32_t* tables, int turn, const uint32_t* q_rk) {
	if (m == 0) return hipSuccess;
	if (nbins > (1ull << 18) || nbins % 32 || !q_scratch || !rec_scratch || !item_scratch || !c_rm || !counters || !tables || rounds == 0) return hipErrorInvalidValue;
	const uint32_t tab_words = msc_ranks_items_table_words(nbins);
	const size_t lds = (size_t)tab_words * 4;
	uint32_t *tab = tables + (size_t)tab_words * (turn & 1), *tab_next = tables + (size_t)tab_words * ((turn & 1) ^ 1);
	const uint64_t items = (uint64_t)m * 3 * rounds;          // (at most: the grid is sized for the bound, the kernel walks the list)
	uint32_t* rec = (uint32_t*)rec_scratch;
	double* extras = (double*)((uint8_t*)rec_scratch + items * kRkCells * sizeof(uint32_t));
	uint32_t *tail = counters + 16 * (turn & 1), *tail_next = counters + 16 * ((turn & 1) ^ 1);
	uint32_t *hq = tail, *n_items = tail + 12;
	hipError_t e;
	static bool attr_set = false;
	if (!attr_set) {
		e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_pass_prep), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
		if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_ranks_items<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
		if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_ranks_items<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 88 * 1024);
		if (e != hipSuccess) return e;
		attr_set = true;
	}
	RkItemMeta* meta = reinterpret_cast<RkItemMeta*>(item_scratch);
	uint2* items_list = reinterpret_cast<uint2*>((uint8_t*)item_scratch + (size_t)m * sizeof(RkItemMeta));
	k_rank_pass_prep<<<dim3(kRiPrepBlocks + (m + kRiListCands - 1) / kRiListCands), dim3(1024), (size_t)(tab_words - 2 * kRiHash) * 4, st>>>((const uint2*)q_ent, q_cum, q_hdr, (uint32_t)nbins, q_rk ? nullptr : q_scratch, (uint32_t)((q_kmers + 255) & ~255ull), guard,
	                                                                      dv ? hq : nullptr, dv ? dv->big : nullptr, meta, m, n_items, items_list, c_off, c_n, c_rm_off, c_rm_n,
	                                                                      cand_scalars, scalar_stride, cand_slots, first, use_window, min_len, max_len, tail_next, tab, tab_next, tab_words, tail + 13);          // (tail + 14: the query's workgroups done)
	// (one workgroup of the divergence form per CU: sixteen waves of it fill a CU's register file; MSC_RANKS_ITEMS_PER_CU)
	static const uint32_t per_cu_env = [] { const char* e = getenv("MSC_RANKS_ITEMS_PER_CU"); return (uint32_t)(e && atoi(e) > 0 ? atoi(e) : 0); }();
	const uint32_t per_cu = per_cu_env ? per_cu_env : dv ? 1u : (uint32_t)std::min<size_t>(2, (150 * 1024) / lds);
	uint32_t blocks = (uint32_t)std::min<uint64_t>((items + kRpBlock / 64 - 1) / (kRpBlock / 64), (uint64_t)num_cus * per_cu);
	if (dv) k_pair_ranks_items<true><<<dim3(blocks), dim3(kRpBlock), lds, st>>>(c_rk, (const uint2*)c_rm, m, (const uint2*)q_ent, q_cum, q_hdr, (uint32_t)nbins, q_rk ? q_rk : q_scratch, rec, extras,
	                                                                             dv->q_scalars, dv->order, meta, items_list, tail, tab, dv->big);
	else k_pair_ranks_items<false><<<dim3(blocks), dim3(kRpBlock), lds, st>>>(c_rk, (const uint2*)c_rm, m, (const uint2*)q_ent, q_cum, q_hdr, (uint32_t)nbins, q_rk ? q_rk : q_scratch, rec, nullptr, nullptr,
	                                                                          0, meta, items_list, tail, tab, nullptr);
	if ((e = hipGetLastError()) != hipSuccess) return e;
	const dim3 fgrid((m + 3) / 4);
	if (dv) k_rank_items_finish<true><<<fgrid, dim3(256), 0, st>>>(rec, extras, rounds, hq, dv->big, meta, m, q_cum, q_hdr, dv->q_scalars, dv->order, partials, dv->div_out, guard);
	else k_rank_items_finish<false><<<fgrid, dim3(256), 0, st>>>(rec, nullptr, rounds, nullptr, nullptr, meta, m, q_cum, q_hdr, nullptr, 0, partials, nullptr, guard);
	return hipGetLastError();
}

// the repeated-bin lists of a sparse set: sizes (n: capacity words) and offsets (off: capacity + 1 words, lists padded to four entries); the caller reads off[capacity], allocates, fills
hipError_t msc_launch_rank_multi_sizes(hipStream_t st, const void* ent, const MscSparseHdr* hdr, uint64_t capacity, uint32_t* n, uint64_t* off) {
	if (capacity == 0) return hipSuccess;
	k_rkm_sizes<<<dim3((unsigned)((capacity + 3) / 4)), dim3(256), 0, st>>>((const uint2*)ent, hdr, capacity, n);
	k_rkl_scan<<<dim3(1), dim3(1024), 0, st>>>(n, capacity, off);
	return hipGetLastError();
}
hipError_t msc_launch_rank_multi_fill(hipStream_t st, const void* ent, const MscSparseHdr* hdr, uint64_t capacity, const uint64_t* off, void* out) {
	if (capacity == 0) return hipSuccess;
	k_rkm_fill<<<dim3((unsigned)((capacity + 3) / 4)), dim3(256), 0, st>>>((const uint2*)ent, hdr, capacity, off, (uint2*)out);
	return hipGetLastError();
}
