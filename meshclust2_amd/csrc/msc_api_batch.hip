// msc_api_batch.hip -- the batched update stage behind msc_update_centres / msc_filter_batch / msc_merge_all (cluster/ClusterFactory.cpp:288-335,
// 612-690). Split from msc_api.hip in r05.
#include <algorithm>
#include <cctype>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "msc_internal.h"

#include "msc_objects.h"
#include "msc_api_private.h"

// ================================================================================================ batched update stage
// mean_shift_update for MANY centres in three launches instead of ~6 launches and ~5 host round trips per centre
// (cluster/ClusterFactory.cpp:288-335; the reference runs the centres of a round under `omp parallel for`, :639, so they are
// independent by construction): Trainer::filter of every centre's neighbourhood list, the FP64 mean of the survivors, and the
// survivor nearest that mean (Trainer::closest). Results are those of msc_filter + msc_mean_nearest per centre.
static int update_centres_one_by_one(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots,
                                     uint64_t n_centres, const msc_hist_set* pts, const uint32_t* pt_slots, const uint64_t* offsets, int64_t* nearest_pos,
                                     uint64_t* n_kept) {
	std::vector<uint8_t> keep;
	std::vector<uint32_t> kept;
	std::vector<uint64_t> where;
	for (uint64_t c = 0; c < n_centres; c++) {
		const uint64_t m = offsets[c + 1] - offsets[c];
		keep.assign(m, 0);
		uint64_t n = 0;
		int r = msc_filter(ctx, model, cutoff, centres, centre_slots[c], pts, pt_slots + offsets[c], m, keep.data(), &n);
		if (r) return r;
		kept.clear();
		where.clear();
		for (uint64_t i = 0; i < m; i++) if (keep[i]) { kept.push_back(pt_slots[offsets[c] + i]); where.push_back(i); }
		if (n_kept) n_kept[c] = kept.size();
		nearest_pos[c] = -1;
		if (!kept.empty()) {
			int64_t pos = -1;
			if ((r = msc_mean_nearest(ctx, pts, kept.data(), kept.size(), &pos, nullptr, nullptr))) return r;
			nearest_pos[c] = (int64_t)where[(size_t)pos];
		}
	}
	return MSC_OK;
}

// Step 3 of the batched update stage on SPARSE sets: the rounded mean of every centre's survivors as a sparse slot of a scratch set
// (scatter-add of the members' excesses into one 32-bit column accumulator per centre, swept in index order: the kernels of
// mean_nearest_sparse with a centre dimension), then distance_d of every survivor to the mean of its centre -> ctx->dist[pair].
// segs[c] = {q_slot = c, first, m} over `members` (device copies are made here), pair_seg[j] = centre of member j.
// (a) the accumulators: one 32-bit column array of 4^k bins per list, zero between calls (the write sweep re-zeroes what it read), and
// for large k a bit per 16 bins and list that the scatter sets and the sweeps follow
// chunks of bins per list the two sweeps of sparse_acc_sweep are cut into: a wave per (list, chunk) -- many lists bring their own
// parallelism (about 65 536 waves in all; at least MSC_SPARSE_SUB chunks, one per index sub-range; a power of two)
static uint32_t sparse_sweep_chunks(const MscLayout& L, uint32_t nc) {
	uint32_t n_chunks = (uint32_t)std::min<uint64_t>(1024, L.nbins / 256);      // a multiple of 16 for every sparse-capable k
	while (n_chunks > MSC_SPARSE_SUB && (uint64_t)n_chunks * nc > 65536) n_chunks /= 2;
	return n_chunks;
}
int sparse_acc_prepare(msc_ctx* ctx, const MscLayout& L, uint32_t nc, uint32_t** touched_out) {
	int r;
	const uint32_t n_chunks = sparse_sweep_chunks(L, nc);
	const uint64_t chunk_bins = L.nbins / n_chunks;
	const size_t acc_bytes = (size_t)nc * L.nbins * sizeof(uint32_t);
	if (acc_bytes > ctx->sp_acc_batch.cap) {
		if ((r = ensure(ctx, ctx->sp_acc_batch, acc_bytes))) return r;
		HIP_TRY(ctx, hipMemsetAsync(ctx->sp_acc_batch.p, 0, ctx->sp_acc_batch.cap, ctx->stream));
	}
	static const bool no_groups = getenv("MSC_SPARSE_MEAN_NO_GROUPS") != nullptr;
	const bool grouped = !no_groups && L.nbins >= msc_sparse_groups_min_bins() && chunk_bins % 512 == 0;
	if (grouped) {
		const size_t tb = (size_t)nc * (L.nbins >> 9) * sizeof(uint32_t);
		if (tb > ctx->sp_touched.cap) {
			if ((r = ensure(ctx, ctx->sp_touched, tb))) return r;
			HIP_TRY(ctx, hipMemsetAsync(ctx->sp_touched.p, 0, ctx->sp_touched.cap, ctx->stream));
		}
	}
	*touched_out = grouped ? (uint32_t*)ctx->sp_touched.p : nullptr;
	return MSC_OK;
}

// (b) the excesses of P lists of `src` (slots[j] belongs to accumulator seg[j]) are added in
int sparse_acc_scatter(msc_ctx* ctx, const msc_hist_set* src, const uint32_t* slots, const uint32_t* seg, uint64_t P, uint32_t* touched) {
	if (P == 0) return MSC_OK;
	int r;
	if ((r = ensure(ctx, ctx->slots, P * sizeof(uint32_t))) || (r = ensure(ctx, ctx->pair_seg, P * sizeof(uint32_t)))) return r;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, slots, P * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->pair_seg.p, seg, P * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, msc_launch_sparse_scatter_batch(ctx->stream, src->ent, src->hdr, (const uint32_t*)ctx->slots.p, (const uint32_t*)ctx->pair_seg.p, (uint32_t)P, src->L.nbins,
	                                             (uint32_t*)ctx->sp_acc_batch.p, touched));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // (slots / seg are the caller's, and ctx->slots is reused below)
	return MSC_OK;
}

// (c) the accumulators of nc lists are swept into the sparse slots 0 .. nc-1 of ctx->sparse_mean_batch: list c's rounded mean over
// m_of[c] members (value_bits = the set's bin type), or -- m_of[c] = 1 and value_bits = 32 -- its summed excesses + 1, the column sums
// a rank sends to the others (msc_colsum_partial). floor_sum_out[c] = sum of floor(mean) (nullable). The accumulators are zero again.
int sparse_acc_sweep(msc_ctx* ctx, const msc_hist_set* pts, uint32_t nc, const uint32_t* m_of, int value_bits, uint32_t* touched, uint64_t* floor_sum_out) {
	const MscLayout& L = pts->L;
	int r;
	// a wave per (list, chunk of bins): many lists bring their own parallelism, and every chunk costs 24 bytes of counts to the host and
	// 16 bytes of offsets back -- with 1024 chunks each, a round over a million centres (BASELINE cfg3) moved 40 GB over PCIe and spent
	// its time in the loops below (r03 profile, 200 000 x 1 kb: 24 s of update stage around 2.7 s of kernels). About 65 536 waves in
	// all; at least 16 chunks (one per index sub-range), a power of two.
	const uint32_t n_chunks = sparse_sweep_chunks(L, nc);
	const uint64_t chunk_bins = L.nbins / n_chunks;
	const uint32_t per_sub = n_chunks / MSC_SPARSE_SUB;
	if ((r = ensure(ctx, ctx->qslots, nc * sizeof(uint32_t))) || (r = ensure(ctx, ctx->sp_counts, (size_t)nc * n_chunks * 3 * sizeof(uint64_t))) ||
	    (r = ensure(ctx, ctx->sp_chunk_off, (size_t)nc * n_chunks * sizeof(uint64_t))) || (r = ensure(ctx, ctx->sp_chunk_cum, (size_t)nc * n_chunks * sizeof(uint64_t))) ||
	    (r = ensure(ctx, ctx->floor_sum, nc * sizeof(uint64_t))))
		return r;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->qslots.p, m_of, nc * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, msc_launch_sparse_mean_count_batch(ctx->stream, value_bits, (const uint32_t*)ctx->sp_acc_batch.p, L.nbins, n_chunks, chunk_bins, nc,
	                                                (const uint32_t*)ctx->qslots.p, (uint64_t*)ctx->sp_counts.p, touched));
	std::vector<uint64_t> counts((size_t)nc * n_chunks * 3);
	HIP_TRY(ctx, hipMemcpyAsync(counts.data(), ctx->sp_counts.p, counts.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	// headers, scalar records, floor sums and the chunks' write offsets of every mean
	std::vector<MscSparseHdr> hdr(nc);
	std::vector<MscSlotScalars> sc(nc);
	std::vector<uint64_t> floor_sum(nc), off((size_t)nc * n_chunks), cb((size_t)nc * n_chunks);
	memset(sc.data(), 0, sizeof(MscSlotScalars) * nc);
	uint64_t used = 0, max_mean_sum = 0;
	uint32_t max_nnz = 0;
	for (uint32_t c = 0; c < nc; c++) {
		MscSparseHdr h{};
		uint64_t n = 0, ex = 0, fl = 0;
		for (uint32_t ch = 0; ch < n_chunks; ch++) {
			if (ch % per_sub == 0) h.split[ch / per_sub] = (uint32_t)n;
			const uint64_t* cnt = &counts[((size_t)c * n_chunks + ch) * 3];
			off[(size_t)c * n_chunks + ch] = used + n;
			cb[(size_t)c * n_chunks + ch] = ex;
			n += cnt[0]; ex += cnt[1]; fl += cnt[2];
		}
		h.split[MSC_SPARSE_SUB] = (uint32_t)n;
		h.nnz = (uint32_t)n;
		h.off = used;
		used += n;
		max_nnz = std::max(max_nnz, h.nnz);
		hdr[c] = h;
		sc[c].sum = L.nbins + ex;          // sum of the rounded mean's bins
		max_mean_sum = std::max<uint64_t>(max_mean_sum, L.nbins + ex);
		sc[c].mag = sc[c].sum;
		sc[c].length = 1;
		floor_sum[c] = L.nbins + fl;
		if (floor_sum_out) floor_sum_out[c] = floor_sum[c];
	}
	msc_hist_set*& ms = ctx->sparse_mean_batch;
	if (!ms || ms->k != pts->k || ms->dtype != pts->dtype || ms->capacity < nc || ms->ent_capacity < used + 1) {
		const uint64_t cap = ms && ms->k == pts->k && ms->dtype == pts->dtype ? std::max<uint64_t>(ms->capacity, nc) : std::max<uint64_t>(nc, 256);
		const uint64_t arena = std::max<uint64_t>(used + used / 2 + 1, ms ? ms->ent_capacity : (1u << 20));
		if (ms) { msc_hist_set_destroy(ms); ms = nullptr; }
		if ((r = msc_hist_set_create_sparse(ctx, pts->k, pts->dtype, cap, arena, &ms))) return r;
	}
	ms->ent_used = used;
	ms->list_epoch++;
	ms->max_nnz = std::max(ms->max_nnz, max_nnz);
	ms->max_sum = std::max(ms->max_sum, max_mean_sum);          // (a rounded mean can hold more k-mers than any member: the bound travels with msc_hist_assign*)
	for (uint32_t c = 0; c < nc; c++) ms->hdr_host[c] = hdr[c];
	HIP_TRY(ctx, hipMemcpyAsync(ms->hdr, hdr.data(), nc * sizeof(MscSparseHdr), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ms->scalars, sc.data(), nc * sizeof(MscSlotScalars), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->floor_sum.p, floor_sum.data(), nc * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_chunk_off.p, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_chunk_cum.p, cb.data(), cb.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, msc_launch_sparse_mean_write_batch(ctx->stream, value_bits, (uint32_t*)ctx->sp_acc_batch.p, L.nbins, n_chunks, chunk_bins, nc, (const uint32_t*)ctx->qslots.p,
	                                                (const uint64_t*)ctx->sp_chunk_off.p, (const uint64_t*)ctx->sp_chunk_cum.p, ms->ent, ms->cum, touched));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // hdr, sc, floor_sum, off, cb live on this frame
	return MSC_OK;
}

// (d) distance_d of every member to the rounded mean of ITS list (slot segs[].q_slot of ctx->sparse_mean_batch) -> ctx->dist[pair]
int sparse_distances_to_means(msc_ctx* ctx, const msc_hist_set* pts, const std::vector<MscBatchSeg>& segs, const std::vector<uint32_t>& pair_seg,
                              const std::vector<uint32_t>& members, uint32_t nc) {
	const MscLayout& L = pts->L;
	const uint64_t P2 = members.size();
	msc_hist_set* ms = ctx->sparse_mean_batch;
	int r;
	if ((r = ensure(ctx, ctx->slots, P2 * sizeof(uint32_t))) || (r = ensure(ctx, ctx->pair_seg, P2 * sizeof(uint32_t))) || (r = ensure(ctx, ctx->segs, nc * sizeof(MscBatchSeg))) ||
	    (r = ensure(ctx, ctx->partials, P2 * sizeof(MscPartial))) || (r = ensure(ctx, ctx->dist, P2 * sizeof(double))))
		return r;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, members.data(), P2 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->pair_seg.p, pair_seg.data(), P2 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->segs.p, segs.data(), nc * sizeof(MscBatchSeg), hipMemcpyHostToDevice, ctx->stream));
	// survivors against the rounded mean of their own centre: only the |p - r| reduction of the merge kernel is used
	HIP_TRY(ctx, msc_launch_pair_sparse_mp_pairs(ctx->stream, pts->ent, pts->cum, pts->hdr, pts->scalars, pts->scalar_stride, (const uint32_t*)ctx->slots.p, (uint32_t)P2,
	                                             ms->ent, ms->cum, ms->hdr, L.nbins, 0, (const MscBatchSeg*)ctx->segs.p, (const uint32_t*)ctx->pair_seg.p,
	                                             (MscPartial*)ctx->partials.p, MSC_ORDER_CAND_FIRST, ctx->num_cus));
	HIP_TRY(ctx, msc_launch_distance_batch(ctx->stream, (const MscPartial*)ctx->partials.p, 1, (uint32_t)P2, pts->scalars, pts->scalar_stride, (const uint32_t*)ctx->slots.p,
	                                       (const uint32_t*)ctx->pair_seg.p, ms->scalars, ms->scalar_stride, (const uint64_t*)ctx->floor_sum.p, (double*)ctx->dist.p));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MSC_OK;
}

static int sparse_means_and_distances(msc_ctx* ctx, const msc_hist_set* pts, const std::vector<MscBatchSeg>& segs, const std::vector<uint32_t>& pair_seg,
                                      const std::vector<uint32_t>& members, uint32_t nc) {
	int r;
	uint32_t* touched = nullptr;
	if ((r = sparse_acc_prepare(ctx, pts->L, nc, &touched))) return r;
	if ((r = sparse_acc_scatter(ctx, pts, members.data(), pair_seg.data(), members.size(), touched))) return r;
	std::vector<uint32_t> m_of(nc);
	for (uint32_t c = 0; c < nc; c++) m_of[c] = segs[c].m;
	if ((r = sparse_acc_sweep(ctx, pts, nc, m_of.data(), pts->dtype, touched, nullptr))) return r;
	return sparse_distances_to_means(ctx, pts, segs, pair_seg, members, nc);
}

// The two divergence sums of a pair list inside the batched entry points: which lists to merge (the sets themselves, or the sparse
// mirrors of dense sets) -- or nothing (*ok = false: the caller goes centre by centre) when a 1 x M call on these sets would NOT take
// the chunked merge kernel, because a pair must get the same kernel, hence the same evaluation order, in every route.
static int batch_div_lists(msc_ctx* ctx, const msc_hist_set* cands, const msc_hist_set* queries, uint64_t any_q_slot, const msc_hist_set** c_sp,
                           const msc_hist_set** q_sp, bool* ok) {
	*ok = false;
	int r;
	if ((r = ensure_sparse_mirror(ctx, cands, c_sp)) || (r = ensure_sparse_mirror(ctx, queries, q_sp))) return r;
	if (!*c_sp || !*q_sp) return MSC_OK;
	*ok = pick_sparse_kernel(*c_sp, *q_sp, any_q_slot, std::max(cands->max_count, queries->max_count), false) == SPK_MP;
	return MSC_OK;
}
// ... and the pass itself, for P pairs already described by ctx->slots / ctx->segs / ctx->pair_seg: sums -> ctx->div_partials[2 * pair].
// A sparse pair of sets gets its integer records from the same launch (partials); dense sets have theirs from k_pair_tiles_batch.
static int batch_div_pass(msc_ctx* ctx, const msc_hist_set* cands, const msc_hist_set* queries, const msc_hist_set* c_sp, const msc_hist_set* q_sp, uint64_t P,
                          int order, MscPartial* partials, uint32_t* div_n) {
	int r;
	const uint32_t dvn = msc_sparse_mp_div_records((uint64_t)c_sp->max_nnz + q_sp->max_nnz);      // records per pair (the 1 x M form's granules)
	*div_n = dvn;
	if ((r = ensure(ctx, ctx->div_tables, P * 256 * 16)) || (r = ensure(ctx, ctx->div_partials, P * dvn * 16))) return r;
	HIP_TRY(ctx, msc_launch_pair_sparse_mp_pairs(ctx->stream, c_sp->ent, c_sp->cum, c_sp->hdr, cands->scalars, cands->scalar_stride, (const uint32_t*)ctx->slots.p, (uint32_t)P,
	                                             q_sp->ent, q_sp->cum, q_sp->hdr, cands->L.nbins, 1, (const MscBatchSeg*)ctx->segs.p, (const uint32_t*)ctx->pair_seg.p, partials,
	                                             order, ctx->num_cus, queries->scalars, queries->scalar_stride, ctx->div_tables.p, ctx->div_partials.p, dvn));
	return MSC_OK;
}

// keep_only != nullptr: Trainer::filter of every list and nothing else -- keep_only[i] = 1 iff pt_slots[i] survives the filter of its
// centre (msc_filter_batch: the rank-local half of a sharded update round, whose means need the other ranks' column sums)
static int update_centres_impl(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots,
                               uint64_t n_centres, const msc_hist_set* pts, const uint32_t* pt_slots, const uint64_t* offsets, int64_t* nearest_pos,
                               uint64_t* n_kept, uint8_t* keep_only) {
	if (!ctx || !model || model->ctx != ctx || !centres || !pts || centres->ctx != ctx || pts->ctx != ctx) return MSC_ERR_INVALID_ARG;
	if (n_centres == 0) return MSC_OK;
	if (!centre_slots || !offsets || (!nearest_pos && !keep_only)) return MSC_ERR_INVALID_ARG;
	if (centres->k != pts->k || centres->dtype != pts->dtype) return fail(ctx, MSC_ERR_INVALID_ARG, "sets differ in k or dtype");
	const uint64_t total = offsets[n_centres];
	if (total && !pt_slots) return MSC_ERR_INVALID_ARG;
	for (uint64_t c = 0; c < n_centres; c++) {
		if (centre_slots[c] >= centres->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "centre slot out of range");
		if (offsets[c + 1] < offsets[c] || offsets[c + 1] - offsets[c] > 0x7fffffffull) return fail(ctx, MSC_ERR_INVALID_ARG, "offsets must be non-decreasing");
	}
	for (uint64_t i = 0; i < total; i++) if (pt_slots[i] >= pts->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "point slot out of range");
	uint64_t want = 0;
	for (int i = 0; i < model->h.n_singles; i++) want |= model->h.single_flag[i];
	static const bool no_batch = getenv("MSC_NO_BATCH_UPDATE") != nullptr;
	// sparse sets (both): the pair-list form of the merge-path kernel takes the place of k_pair_tiles_batch, and the rounded means of a
	// chunk of centres are built as sparse slots by the scatter / count / write kernels with a centre dimension (32-bit range)
	const bool sp = pts->sparse && centres->sparse;
	auto one_by_one = [&]() -> int {
		if (!keep_only) return update_centres_one_by_one(ctx, model, cutoff, centres, centre_slots, n_centres, pts, pt_slots, offsets, nearest_pos, n_kept);
		for (uint64_t c = 0; c < n_centres; c++) {
			uint64_t n = 0;
			const int rr = msc_filter(ctx, model, cutoff, centres, centre_slots[c], pts, pt_slots + offsets[c], offsets[c + 1] - offsets[c], keep_only + offsets[c], &n);
			if (rr) return rr;
		}
		return MSC_OK;
	};
	if (no_batch || (pts->sparse != centres->sparse) || (sp && std::max(pts->max_count, centres->max_count) >= 65536) || (want & MSC_FEAT_GROUPS) ||
	    needs_wide(pts, centres))
		return one_by_one();
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const MscLayout& L = pts->L;
	int r;
	// a `--feat slow` model: the filter's two divergence sums from a pair-list pass of the chunked merge kernel over the lists
	const bool want_div = (want & MSC_FEAT_DIV) != 0;
	const msc_hist_set *c_sp = nullptr, *q_sp = nullptr;
	if (want_div) {
		bool ok = false;
		if ((r = batch_div_lists(ctx, pts, centres, centre_slots[0], &c_sp, &q_sp, &ok))) return r;
		if (!ok) return one_by_one();
	}
	// lengths of every centre slot in one strided copy (Trainer::filter's window is relative to the centre's length)
	std::vector<uint64_t> clen(centres->capacity);
	HIP_TRY(ctx, hipMemcpy2DAsync(clen.data(), 8, centres->scalars + offsetof(MscSlotScalars, length), centres->scalar_stride, 8, centres->capacity,
	                              hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	const double id = trainer_get_id(cutoff);
	// chunks of centres: their rounded means share one scratch set (<= 4 GiB; sparse: <= 1 GiB of 32-bit column accumulators) and
	// their pair counts stay 32-bit
	const uint32_t PS = sp ? 1 : L.S;          // partial records per pair
	// sparse: one 32-bit column accumulator of 4^k bins per centre of a chunk. Every chunk costs a handful of launches and two host
	// round trips, so at k = 13 (256 MiB per accumulator) a 1 GiB budget -- 4 centres per chunk -- made the update stage launch-bound
	// (2 000 x 20 kb: 1.6 s); the budget is a quarter of the free device memory, between 1 and 16 GiB
	uint64_t acc_budget = 1024ull << 20;
	if (sp) {
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) acc_budget = std::min<uint64_t>(16384ull << 20, std::max<uint64_t>(acc_budget, (free_b + ctx->sp_acc_batch.cap) / 4));
		else (void)hipGetLastError();
	}
	const uint64_t max_chunk_centres = sp ? std::max<uint64_t>(1, std::min<uint64_t>(4096, acc_budget / (L.nbins * 4)))
	                                      : std::max<uint64_t>(1, (4096ull << 20) / L.slot_bytes);
	const uint64_t max_chunk_pairs = std::min<uint64_t>(std::max<uint64_t>(1, (2048ull << 20) / ((uint64_t)PS * sizeof(MscPartial))),
	                                                    want_div ? (1024ull << 20) / 4096 : ~0ull);      // (a 4 KiB table of divergence terms per pair)
	std::vector<MscBatchSeg> segs;
	std::vector<uint32_t> pair_seg, members, where;
	std::vector<uint64_t> seg_centre;
	std::vector<uint8_t> keep;
	std::vector<double> dist;
	for (uint64_t c0 = 0; c0 < n_centres;) {
		uint64_t c1 = c0;
		// (sparse sets: the accumulators bound step 3 only -- it takes its centres in sub-chunks; the filter of step 1 takes as many as its pair
		// budget holds. r05: at k = 13 a chunk of 64 centres -- 256 MiB of accumulator each -- was the unit of the WHOLE loop: 414 chunks per
		// round of BASELINE cfg4, a handful of launches and host round trips each, 4.8 s of a 14.5 s run)
		const uint64_t outer_cap = sp ? std::max<uint64_t>(max_chunk_centres, 65536) : max_chunk_centres;
		while (c1 < n_centres && c1 - c0 < outer_cap && (c1 == c0 || offsets[c1 + 1] - offsets[c0] <= max_chunk_pairs)) c1++;
		const uint64_t nc = c1 - c0, base = offsets[c0], P = offsets[c1] - base;
		// ---- 1. filter: every centre against its list
		segs.resize(nc);
		pair_seg.resize(P);
		uint32_t max_m = 0;
		for (uint64_t c = c0; c < c1; c++) {
			MscBatchSeg& sg = segs[c - c0];
			sg.q_slot = centre_slots[c];
			sg.first = (uint32_t)(offsets[c] - base);
			sg.m = (uint32_t)(offsets[c + 1] - offsets[c]);
			sg.pad_ = 0;
			const uint64_t len = clen[centre_slots[c]];
			sg.min_len = (uint64_t)((double)len * id);      // cluster/Trainer.cpp:126-127
			sg.max_len = (uint64_t)((double)len / id);
			max_m = std::max(max_m, sg.m);
			for (uint32_t i = 0; i < sg.m; i++) pair_seg[sg.first + i] = (uint32_t)(c - c0);
		}
		keep.assign(P, 0);
		if (P) {
			if ((r = ensure(ctx, ctx->segs, nc * sizeof(MscBatchSeg))) || (r = ensure(ctx, ctx->pair_seg, P * sizeof(uint32_t))) ||
			    (r = ensure(ctx, ctx->slots, P * sizeof(uint32_t))) || (r = ensure(ctx, ctx->partials, P * PS * sizeof(MscPartial))) ||
			    (r = ensure(ctx, ctx->soa_close, P)) || (r = ensure(ctx, ctx->err_word, sizeof(int32_t))))
				return r;
			HIP_TRY(ctx, hipMemcpyAsync(ctx->segs.p, segs.data(), nc * sizeof(MscBatchSeg), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(ctx->pair_seg.p, pair_seg.data(), P * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, pt_slots + base, P * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemsetAsync(ctx->err_word.p, 0, sizeof(int32_t), ctx->stream));
			uint32_t dvn = 1;          // {jd, js} records per pair
			if (sp && want_div) {
				if ((r = batch_div_pass(ctx, pts, centres, c_sp, q_sp, P, MSC_ORDER_QUERY_FIRST, (MscPartial*)ctx->partials.p, &dvn))) return r;
			} else if (sp)
				HIP_TRY(ctx, msc_launch_pair_sparse_mp_pairs(ctx->stream, pts->ent, pts->cum, pts->hdr, pts->scalars, pts->scalar_stride, (const uint32_t*)ctx->slots.p,
				                                             (uint32_t)P, centres->ent, centres->cum, centres->hdr, L.nbins, 1, (const MscBatchSeg*)ctx->segs.p,
				                                             (const uint32_t*)ctx->pair_seg.p, (MscPartial*)ctx->partials.p, MSC_ORDER_QUERY_FIRST, ctx->num_cus));
			else {
				HIP_TRY(ctx, msc_launch_pair_tiles_batch(ctx->stream, L, pts->dtype, pts->bins, pts->scalars, (const uint32_t*)ctx->slots.p, (const MscBatchSeg*)ctx->segs.p,
				                                         (uint32_t)nc, max_m, centres->bins, centres->L.slot_bytes, centres->scalars, centres->scalar_stride, 1,
				                                         (MscPartial*)ctx->partials.p, MSC_ORDER_QUERY_FIRST));
				if (want_div) {          // the mirrors' lists, the dense sets' scalar records (a mirror has none of its own)
					if ((r = ensure(ctx, ctx->sp_partials, P * sizeof(MscPartial)))) return r;
					if ((r = batch_div_pass(ctx, pts, centres, c_sp, q_sp, P, MSC_ORDER_QUERY_FIRST, (MscPartial*)ctx->sp_partials.p, &dvn))) return r;
				}
			}
			MscEpilogueArgs ea;
			memset(&ea, 0, sizeof ea);
			ea.partials = (const MscPartial*)ctx->partials.p;
			if (want_div) { ea.div_direct = (const double*)ctx->div_partials.p; ea.div_direct_n = dvn; ea.div_base = L.nbins; }
			ea.S = PS;
			ea.sparse_base = sp ? L.nbins : 0;
			ea.m = (uint32_t)P;
			ea.cand_scalars = pts->scalars;
			ea.cand_scalar_stride = pts->scalar_stride;
			ea.cand_slots = (const uint32_t*)ctx->slots.p;
			ea.q_scalars = centres->scalars;
			ea.qset_scalars = centres->scalars;
			ea.q_scalar_stride = centres->scalar_stride;
			ea.nbins = L.nbins;
			ea.dtype = pts->dtype;
			ea.order = MSC_ORDER_QUERY_FIRST;      // classify(p, pt.first), cluster/Trainer.cpp:133
			ea.use_window = 1;
			ea.model = model->d;
			ea.close_soa = (uint8_t*)ctx->soa_close.p;
			ea.error_word = (int32_t*)ctx->err_word.p;
			ea.segs = (const MscBatchSeg*)ctx->segs.p;
			ea.pair_seg = (const uint32_t*)ctx->pair_seg.p;
			HIP_TRY(ctx, msc_launch_epilogue(ctx->stream, ea));
			int32_t first_err = 0;
			HIP_TRY(ctx, hipMemcpyAsync(keep.data(), ctx->soa_close.p, P, hipMemcpyDeviceToHost, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(&first_err, ctx->err_word.p, sizeof first_err, hipMemcpyDeviceToHost, ctx->stream));
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
			if (first_err == MSC_ERR_ZERO_LENGTH) return fail(ctx, first_err, "length_difference: a point has length 0 (the reference throws 123, predict/Feature.cpp:878-886)");
			if (first_err == MSC_ERR_NAN) return fail(ctx, first_err, "normalisation produced NaN (the reference throws, predict/Feature.cpp:143-146)");
			if (first_err < 0) return fail(ctx, first_err, "feature evaluation failed with status %d", first_err);
		}
		if (keep_only) {
			if (P) memcpy(keep_only + base, keep.data(), P);
			c0 = c1;
			continue;
		}
		// ---- 2. survivors per centre. A centre with ONE survivor needs no mean: the survivor is the nearest (Trainer::closest over one
		// point, cluster/Trainer.cpp:144-157) -- on BASELINE cfg3 (10^6 x 1 kb) five centres in six, round after round; only centres with
		// two survivors and more go through step 3 (their means sit in slots 0 .. nc2 - 1 of the scratch set; seg_centre maps them back)
		members.clear();
		where.clear();
		pair_seg.clear();
		seg_centre.clear();
		uint32_t max_m2 = 0, nc2 = 0;
		for (uint64_t c = c0; c < c1; c++) {
			const MscBatchSeg old = segs[c - c0];
			const uint32_t at = (uint32_t)members.size();
			for (uint32_t i = 0; i < old.m; i++)
				if (keep[old.first + i]) { members.push_back(pt_slots[base + old.first + i]); where.push_back(i); pair_seg.push_back(nc2); }
			const uint32_t m2 = (uint32_t)members.size() - at;
			if (n_kept) n_kept[c] = m2;
			nearest_pos[c] = -1;
			if (m2 == 1) nearest_pos[c] = (int64_t)where[at];
			if (m2 < 2) { members.resize(at); where.resize(at); pair_seg.resize(at); continue; }
			MscBatchSeg& sg = segs[nc2];          // (nc2 <= c - c0: the slot has been read already)
			sg.q_slot = nc2;                      // slot of this centre's rounded mean in the scratch set
			sg.first = at;
			sg.m = m2;
			sg.pad_ = 0;
			sg.min_len = 0;
			sg.max_len = ~0ull;
			max_m2 = std::max(max_m2, m2);
			seg_centre.push_back(c);
			nc2++;
		}
		segs.resize(nc2);
		const uint64_t P2 = members.size();
		if (P2 == 0) { c0 = c1; continue; }
		// ---- 3. means of the survivors (exact integer column sums), rounded means as slots of a scratch set, distance_d of every survivor
		dist.resize(P2);
		if (sp) {
			std::vector<MscBatchSeg> sub_segs;
			std::vector<uint32_t> sub_pair, sub_members;
			for (uint32_t j0 = 0; j0 < nc2;) {
				const uint32_t j1 = (uint32_t)std::min<uint64_t>(nc2, j0 + max_chunk_centres);
				const uint32_t f0 = segs[j0].first, f1 = j1 < nc2 ? segs[j1].first : (uint32_t)P2;
				sub_segs.assign(segs.begin() + j0, segs.begin() + j1);
				for (MscBatchSeg& sg : sub_segs) { sg.q_slot -= j0; sg.first -= f0; }
				sub_members.assign(members.begin() + f0, members.begin() + f1);
				sub_pair.assign(pair_seg.begin() + f0, pair_seg.begin() + f1);
				for (uint32_t& x : sub_pair) x -= j0;
				if ((r = sparse_means_and_distances(ctx, pts, sub_segs, sub_pair, sub_members, j1 - j0))) return r;
				HIP_TRY(ctx, hipMemcpyAsync(dist.data() + f0, ctx->dist.p, (size_t)(f1 - f0) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
				HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
				j0 = j1;
			}
		} else {
			if (!ctx->batch_scratch || ctx->batch_scratch->k != pts->k || ctx->batch_scratch->dtype != pts->dtype || ctx->batch_scratch->capacity < nc2) {
				if (ctx->batch_scratch) { msc_hist_set_destroy(ctx->batch_scratch); ctx->batch_scratch = nullptr; }
				if ((r = msc_hist_set_create(ctx, pts->k, pts->dtype, std::min<uint64_t>(max_chunk_centres, std::max<uint64_t>(nc, 256)), &ctx->batch_scratch))) return r;
			}
			msc_hist_set* rs = ctx->batch_scratch;
			if ((r = ensure(ctx, ctx->floor_sum, nc2 * sizeof(uint64_t))) || (r = ensure(ctx, ctx->slots, P2 * sizeof(uint32_t))) ||
			    (r = ensure(ctx, ctx->pair_seg, P2 * sizeof(uint32_t))) || (r = ensure(ctx, ctx->partials, P2 * L.S * sizeof(MscPartial))) ||
			    (r = ensure(ctx, ctx->dist, P2 * sizeof(double))))
				return r;
			HIP_TRY(ctx, hipMemcpyAsync(ctx->segs.p, segs.data(), nc2 * sizeof(MscBatchSeg), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(ctx->pair_seg.p, pair_seg.data(), P2 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, members.data(), P2 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, msc_launch_colsum_batch(ctx->stream, L, pts->dtype, pts->bins, (const uint32_t*)ctx->slots.p, (const MscBatchSeg*)ctx->segs.p, nc2,
			                                     rs->bins, (uint64_t*)ctx->floor_sum.p));
			HIP_TRY(ctx, hipMemsetAsync(rs->scalars, 0, rs->scalar_stride * nc2, ctx->stream));
			HIP_TRY(ctx, msc_launch_finalize(ctx->stream, rs->bins, rs->scalars, L, pts->dtype, 0, nc2, false));
			HIP_TRY(ctx, msc_launch_pair_tiles_batch(ctx->stream, L, pts->dtype, pts->bins, pts->scalars, (const uint32_t*)ctx->slots.p, (const MscBatchSeg*)ctx->segs.p,
			                                         nc2, max_m2, rs->bins, rs->L.slot_bytes, rs->scalars, rs->scalar_stride, 0, (MscPartial*)ctx->partials.p,
			                                         MSC_ORDER_CAND_FIRST));
			HIP_TRY(ctx, msc_launch_distance_batch(ctx->stream, (const MscPartial*)ctx->partials.p, L.S, (uint32_t)P2, pts->scalars, pts->scalar_stride,
			                                       (const uint32_t*)ctx->slots.p, (const uint32_t*)ctx->pair_seg.p, rs->scalars, rs->scalar_stride,
			                                       (const uint64_t*)ctx->floor_sum.p, (double*)ctx->dist.p));
			HIP_TRY(ctx, hipMemcpyAsync(dist.data(), ctx->dist.p, P2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		}
		// first minimum wins (cluster/Trainer.cpp:150-153)
		for (uint32_t j = 0; j < nc2; j++) {
			const MscBatchSeg& sg = segs[j];
			uint32_t best = 0;
			for (uint32_t i = 1; i < sg.m; i++) if (dist[sg.first + i] < dist[sg.first + best]) best = i;
			nearest_pos[seg_centre[j]] = (int64_t)where[sg.first + best];
		}
		c0 = c1;
	}
	return MSC_OK;
}

extern "C" int msc_update_centres(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots,
                                  uint64_t n_centres, const msc_hist_set* pts, const uint32_t* pt_slots, const uint64_t* offsets, int64_t* nearest_pos,
                                  uint64_t* n_kept) {
	if (!nearest_pos && n_centres) return MSC_ERR_INVALID_ARG;
	return update_centres_impl(ctx, model, cutoff, centres, centre_slots, n_centres, pts, pt_slots, offsets, nearest_pos, n_kept, nullptr);
}

extern "C" int msc_filter_batch(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots, uint64_t n_centres,
                                const msc_hist_set* pts, const uint32_t* pt_slots, const uint64_t* offsets, uint8_t* keep) {
	if (n_centres && offsets && offsets[n_centres] && !keep) return MSC_ERR_INVALID_ARG;
	return update_centres_impl(ctx, model, cutoff, centres, centre_slots, n_centres, pts, pt_slots, offsets, nullptr, nullptr, keep);
}

// Trainer::merge for EVERY centre of the serial merge loop in one launch (cluster/ClusterFactory.cpp:383-401 calls
// trn.merge(centers, i, i + 1, min(n - 1, i + delta)) for i = 0 .. n-1; no call changes a histogram, so the calls are independent).
// best_out[i] = what msc_merge(..., current = i, begin = i + 1, last = min(n - 1, i + delta)) returns.
// which == nullptr: every centre, best_out[i] for centre i; else best_out[w] for centre which[w] (ascending or not: the calls are independent)
static int merge_impl(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots, uint64_t n, int delta,
                      const uint64_t* which, uint64_t n_which, int64_t* best_out) {
	if (!ctx || !model || model->ctx != ctx || !centres || centres->ctx != ctx || (n && !centre_slots) || (n_which && !best_out) || delta < 0) return MSC_ERR_INVALID_ARG;
	if (n == 0 || n_which == 0) return MSC_OK;
	for (uint64_t i = 0; i < n; i++) if (centre_slots[i] >= centres->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "centre slot out of range");
	if (which) for (uint64_t w = 0; w < n_which; w++) if (which[w] >= n) return fail(ctx, MSC_ERR_INVALID_ARG, "merge: centre index out of range");
	auto centre_of = [&](uint64_t w) -> uint64_t { return which ? which[w] : w; };
	uint64_t want = 0;
	for (int i = 0; i < model->h.n_singles; i++) want |= model->h.single_flag[i];
	static const bool no_batch = getenv("MSC_NO_BATCH_UPDATE") != nullptr;
	// sparse centres: the pair-list form of the merge-path kernel (32-bit range) takes the place of k_pair_tiles_batch
	const bool sp = centres->sparse;
	const bool want_div = (want & MSC_FEAT_DIV) != 0;
	const msc_hist_set *c_sp = nullptr, *q_sp = nullptr;
	bool one_by_one = no_batch || (sp && centres->max_count >= 65536) || (want & MSC_FEAT_GROUPS) || needs_wide(centres, centres) || n > 0x7fffffffull;
	if (!one_by_one && want_div) {
		bool ok = false;
		int r0 = hipSetDevice(ctx->device) == hipSuccess ? batch_div_lists(ctx, centres, centres, centre_slots[0], &c_sp, &q_sp, &ok) : MSC_ERR_HIP;
		if (r0) return r0;
		one_by_one = !ok;
	}
	if (one_by_one) {
		for (uint64_t w = 0; w < n_which; w++) {
			const uint64_t i = centre_of(w);
			int r = msc_merge(ctx, model, cutoff, centres, centre_slots, n, (int64_t)i, (int64_t)i + 1, (int64_t)std::min<uint64_t>(n - 1, i + (uint64_t)delta), &best_out[w]);
			if (r) return r;
		}
		return MSC_OK;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const MscLayout& L = centres->L;
	int r;
	std::vector<uint64_t> clen(centres->capacity);
	HIP_TRY(ctx, hipMemcpy2DAsync(clen.data(), 8, centres->scalars + offsetof(MscSlotScalars, length), centres->scalar_stride, 8, centres->capacity,
	                              hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	const double id = trainer_get_id(cutoff);
	const uint32_t PS = sp ? 1 : L.S;          // partial records per pair
	const uint64_t max_chunk_pairs = std::min<uint64_t>(std::max<uint64_t>(1, (2048ull << 20) / ((uint64_t)PS * sizeof(MscPartial))),
	                                                    want_div ? (1024ull << 20) / 4096 : ~0ull);
	std::vector<MscBatchSeg> segs;
	std::vector<uint32_t> pair_seg, cand;
	std::vector<MscPairOut> po;
	// (c0, c1: positions in the list of centres ASKED about; ci: the centre's index among all)
	for (uint64_t c0 = 0; c0 < n_which;) {
		segs.clear(); pair_seg.clear(); cand.clear();
		uint64_t c1 = c0;
		uint32_t max_m = 0;
		while (c1 < n_which && (c1 == c0 || cand.size() + (uint64_t)delta <= max_chunk_pairs)) {
			const uint64_t ci = centre_of(c1);
			MscBatchSeg sg;
			sg.q_slot = centre_slots[ci];
			sg.first = (uint32_t)cand.size();
			const uint64_t last = std::min<uint64_t>(n - 1, ci + (uint64_t)delta);
			for (uint64_t j = ci + 1; j <= last; j++) { cand.push_back(centre_slots[j]); pair_seg.push_back((uint32_t)(c1 - c0)); }
			sg.m = (uint32_t)cand.size() - sg.first;
			sg.pad_ = 0;
			const uint64_t len = clen[centre_slots[ci]];
			sg.min_len = (uint64_t)((double)len * id);      // cluster/Trainer.cpp:80-81
			sg.max_len = (uint64_t)((double)len / id);
			max_m = std::max(max_m, sg.m);
			segs.push_back(sg);
			c1++;
		}
		const uint64_t nc = c1 - c0, P = cand.size();
		for (uint64_t i = c0; i < c1; i++) best_out[i] = 0;
		if (P) {
			if ((r = ensure(ctx, ctx->segs, nc * sizeof(MscBatchSeg))) || (r = ensure(ctx, ctx->pair_seg, P * sizeof(uint32_t))) ||
			    (r = ensure(ctx, ctx->slots, P * sizeof(uint32_t))) || (r = ensure(ctx, ctx->partials, P * PS * sizeof(MscPartial))) ||
			    (r = ensure(ctx, ctx->pair_out, P * sizeof(MscPairOut))) || (r = ensure(ctx, ctx->err_word, sizeof(int32_t))))
				return r;
			HIP_TRY(ctx, hipMemcpyAsync(ctx->segs.p, segs.data(), nc * sizeof(MscBatchSeg), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(ctx->pair_seg.p, pair_seg.data(), P * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, cand.data(), P * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemsetAsync(ctx->err_word.p, 0, sizeof(int32_t), ctx->stream));
			uint32_t dvn = 1;          // {jd, js} records per pair
			if (sp && want_div) {
				if ((r = batch_div_pass(ctx, centres, centres, c_sp, q_sp, P, MSC_ORDER_CAND_FIRST, (MscPartial*)ctx->partials.p, &dvn))) return r;
			} else if (sp)
				HIP_TRY(ctx, msc_launch_pair_sparse_mp_pairs(ctx->stream, centres->ent, centres->cum, centres->hdr, centres->scalars, centres->scalar_stride,
				                                             (const uint32_t*)ctx->slots.p, (uint32_t)P, centres->ent, centres->cum, centres->hdr, L.nbins, 1,
				                                             (const MscBatchSeg*)ctx->segs.p, (const uint32_t*)ctx->pair_seg.p, (MscPartial*)ctx->partials.p,
				                                             MSC_ORDER_CAND_FIRST, ctx->num_cus));
			else {
				HIP_TRY(ctx, msc_launch_pair_tiles_batch(ctx->stream, L, centres->dtype, centres->bins, centres->scalars, (const uint32_t*)ctx->slots.p,
				                                         (const MscBatchSeg*)ctx->segs.p, (uint32_t)nc, max_m, centres->bins, L.slot_bytes, centres->scalars,
				                                         centres->scalar_stride, 1, (MscPartial*)ctx->partials.p, MSC_ORDER_CAND_FIRST));
				if (want_div) {
					if ((r = ensure(ctx, ctx->sp_partials, P * sizeof(MscPartial)))) return r;
					if ((r = batch_div_pass(ctx, centres, centres, c_sp, q_sp, P, MSC_ORDER_CAND_FIRST, (MscPartial*)ctx->sp_partials.p, &dvn))) return r;
				}
			}
			MscEpilogueArgs ea;
			memset(&ea, 0, sizeof ea);
			ea.partials = (const MscPartial*)ctx->partials.p;
			if (want_div) { ea.div_direct = (const double*)ctx->div_partials.p; ea.div_direct_n = dvn; ea.div_base = L.nbins; }
			ea.S = PS;
			ea.sparse_base = sp ? L.nbins : 0;
			ea.m = (uint32_t)P;
			ea.cand_scalars = centres->scalars;
			ea.cand_scalar_stride = centres->scalar_stride;
			ea.cand_slots = (const uint32_t*)ctx->slots.p;
			ea.q_scalars = centres->scalars;
			ea.qset_scalars = centres->scalars;
			ea.q_scalar_stride = centres->scalar_stride;
			ea.nbins = L.nbins;
			ea.dtype = centres->dtype;
			ea.order = MSC_ORDER_CAND_FIRST;       // feat->compute(*cen, *p), cluster/Trainer.cpp:93
			ea.use_window = 1;
			ea.model = model->d;
			ea.pair_out = (MscPairOut*)ctx->pair_out.p;
			ea.error_word = (int32_t*)ctx->err_word.p;
			ea.segs = (const MscBatchSeg*)ctx->segs.p;
			ea.pair_seg = (const uint32_t*)ctx->pair_seg.p;
			HIP_TRY(ctx, msc_launch_epilogue(ctx->stream, ea));
			po.resize(P);
			int32_t first_err = 0;
			HIP_TRY(ctx, hipMemcpyAsync(po.data(), ctx->pair_out.p, P * sizeof(MscPairOut), hipMemcpyDeviceToHost, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(&first_err, ctx->err_word.p, sizeof first_err, hipMemcpyDeviceToHost, ctx->stream));
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
			if (first_err == MSC_ERR_ZERO_LENGTH) return fail(ctx, first_err, "length_difference: a point has length 0 (the reference throws 123, predict/Feature.cpp:878-886)");
			if (first_err == MSC_ERR_NAN) return fail(ctx, first_err, "normalisation produced NaN (the reference throws, predict/Feature.cpp:143-146)");
			if (first_err < 0) return fail(ctx, first_err, "feature evaluation failed with status %d", first_err);
			// best = best.second > dist ? best : (i, dist), from (0, DBL_MIN): among in-window centres that classify close, the
			// largest combo 0, the LATER index on ties (cluster/Trainer.cpp:79-105)
			for (uint64_t c = c0; c < c1; c++) {
				const MscBatchSeg& sg = segs[c - c0];
				double best_sim = 2.2250738585072014e-308;
				int64_t best = 0;
				for (uint32_t i = 0; i < sg.m; i++) {
					const MscPairOut& p = po[sg.first + i];
					if (p.status != 0 || !p.close) continue;
					if (!(best_sim > p.combo0)) { best_sim = p.combo0; best = (int64_t)(centre_of(c) + 1 + i); }
				}
				best_out[c] = best;
			}
		}
		c0 = c1;
	}
	return MSC_OK;
}

extern "C" int msc_merge_all(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots, uint64_t n,
                             int delta, int64_t* best_out) {
	return merge_impl(ctx, model, cutoff, centres, centre_slots, n, delta, nullptr, n, best_out);
}

// ... for SOME of the centres: best_out[w] = what msc_merge(..., current = which[w], begin = which[w] + 1, last = min(n - 1, which[w] + delta)) returns.
// A round of the serial merge loop repeats most of the round before it once the clusters have settled; the driver asks only about the
// centres whose delta + 1 histograms changed (msc_driver.hpp: merge_round).
extern "C" int msc_merge_some(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots, uint64_t n,
                              int delta, const uint64_t* which, uint64_t n_which, int64_t* best_out) {
	if (n_which && !which) return MSC_ERR_INVALID_ARG;
	return merge_impl(ctx, model, cutoff, centres, centre_slots, n, delta, which, n_which, best_out);
}
