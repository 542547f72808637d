// msc_api_score.hip -- the scoring driver behind the C ABI's 1 x M calls (include/meshclust2_hip.h): run_score -- which kernel takes a pass, its
// scratch, the epilogue and the reduction -- and msc_pair_features_raw / msc_score / msc_get_close / msc_filter / msc_merge / msc_search /
// msc_mean_nearest (Trainer::get_close / filter / merge / closest, cluster/Trainer.cpp:23-157). Split from msc_api.hip in r05.
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

// ================================================================================================ scoring driver


// integer range of the fast streaming kernels (pair_features.hip header); outside it the 64-bit kernel runs
static bool needs_wide_impl(const msc_hist_set* a, const msc_hist_set* b) {
	const uint64_t mc = std::max(a->max_count, b->max_count), ms = std::max(a->max_sum, b->max_sum);
	// a lane adds the |prefix difference| of its R bins of a tile in 32 bits; a prefix difference is at most the larger excess
	// (k-mer) total, so R * excess must stay below 2^32 (only sequences of >= 2^26 k-mers can break it)
	const uint64_t excess = ms > a->L.nbins ? ms - a->L.nbins : 0;
	return mc > kNarrowMaxCount || ms > kNarrowMaxSum || (uint64_t)a->L.R * excess >= (1ull << 32);
}

int validate_pair(msc_ctx* ctx, const msc_hist_set* cands, const msc_hist_set* qset, uint64_t q_slot, const uint32_t* slots, uint64_t m) {
	if (!ctx || !cands || !qset || cands->ctx != ctx || qset->ctx != ctx) return MSC_ERR_INVALID_ARG;
	if (cands->k != qset->k || cands->dtype != qset->dtype || cands->sparse != qset->sparse)
		return fail(ctx, MSC_ERR_INVALID_ARG, "query and candidate sets differ in k, dtype or layout");
	if (q_slot >= qset->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "query slot out of range");
	if (m > 0xfffffff0ull) return fail(ctx, MSC_ERR_INVALID_ARG, "too many candidates in one call");
	if (slots) { for (uint64_t i = 0; i < m; i++) if (slots[i] >= cands->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "candidate slot %u out of range", slots[i]); }
	else if (m > cands->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "m exceeds capacity");
	return MSC_OK;
}

// Which merge kernel scores a query list against candidate lists -- one rule for sparse sets and for the sparse mirrors of dense
// sets, so a pair gets the same kernel (hence the same evaluation order of the FP64 divergence sums) in every route:
//   SPK_LDS     whole lists in LDS (MSC_SPARSE_LDS=1; kept for comparison), 32-bit range, one record per candidate
//   SPK_MP      merge-path chunks, 32-bit range (counts < 2^16, sums < 2^31), one record per candidate
//   SPK_GENERIC lane per index sub-range straight from global memory, 64-bit running values, 16 records per candidate
SparseKernel pick_sparse_kernel(const msc_hist_set* c_sp, const msc_hist_set* q_sp, uint64_t q_slot, uint64_t max_count, bool wide) {
	static const bool want_lds = getenv("MSC_SPARSE_LDS") != nullptr;
	static const bool no_mp = getenv("MSC_SPARSE_NO_MP") != nullptr;
	const uint64_t q_nnz = q_sp->hdr_host[q_slot].nnz;
	if (want_lds && !wide && max_count < 65536 && c_sp->L.nbins >= 64 && ((size_t)(q_nnz + 128) + 4ull * (c_sp->max_nnz + 128)) * 8 <= 96 * 1024) return SPK_LDS;
	if (!no_mp && !wide && max_count < 65536 && q_nnz + c_sp->max_nnz <= msc_sparse_mp_max_entries()) return SPK_MP;
	return SPK_GENERIC;
}
uint32_t sparse_records(SparseKernel k, uint32_t mp_parts) { return k == SPK_GENERIC ? MSC_SPARSE_SUB : k == SPK_MP ? mp_parts : 1; }
// {jd, js} records per pair: the merge-path kernel leaves one per granule of the merged order (lists of up to `entries` together)
uint32_t div_records(SparseKernel k, uint64_t entries) { return k == SPK_GENERIC ? MSC_SPARSE_SUB : k == SPK_MP ? msc_sparse_mp_div_records(entries) : 1; }
const char* sparse_kernel_name(SparseKernel k) { return k == SPK_LDS ? "k_pair_sparse_lds" : k == SPK_MP ? "k_pair_sparse_mp" : "k_pair_sparse"; }

// candidates [off, off + mc) (or the device slot list d_slots) of the sparse set / mirror c_sp against slot q_slot of q_sp; the
// scalar records are those of the sets the lists belong to (a mirror has none of its own)
hipError_t launch_sparse_pass(msc_ctx* ctx, SparseKernel k, const msc_hist_set* c_sp, const uint8_t* c_scalars, uint64_t c_stride, const uint32_t* d_slots,
                              uint64_t off, uint32_t mc, const msc_hist_set* q_sp, uint64_t q_slot, const uint8_t* q_scal, uint64_t nbins, int use_window,
                              uint64_t min_len, uint64_t max_len, MscPartial* partials, void* div_tables, void* div_partials, int order, uint32_t parts,
                              uint32_t div_stride) {
	const MscSparseHdr* c_hdr = c_sp->hdr + (d_slots ? 0 : off);
	const uint8_t* c_scal = c_scalars + (d_slots ? 0 : off * c_stride);
	const uint32_t q_nnz = q_sp->hdr_host[q_slot].nnz;
	if (k == SPK_LDS)
		return msc_launch_pair_sparse_lds(ctx->stream, c_sp->ent, c_sp->cum, c_hdr, c_scal, c_stride, d_slots, mc, q_sp->ent, q_sp->cum, q_sp->hdr + q_slot, q_scal, nbins,
		                                  q_nnz, c_sp->max_nnz, use_window, min_len, max_len, partials, div_tables, div_partials, order, ctx->num_cus);
	if (k == SPK_MP)
		return msc_launch_pair_sparse_mp(ctx->stream, c_sp->ent, c_sp->cum, c_hdr, c_scal, c_stride, d_slots, mc, q_sp->ent, q_sp->cum, q_sp->hdr + q_slot, q_scal, nbins,
		                                 use_window, min_len, max_len, partials, div_tables, div_partials, order, ctx->num_cus,
		                                 (uint32_t)std::min<uint64_t>(0x7fffffffull, (uint64_t)q_nnz + c_sp->max_nnz), parts, q_nnz, c_sp->max_nnz, div_stride);
	return msc_launch_pair_sparse(ctx->stream, c_sp->ent, c_sp->cum, c_hdr, c_scal, c_stride, d_slots, mc, q_sp->ent, q_sp->cum, q_sp->hdr + q_slot, q_scal, nbins,
	                              use_window, min_len, max_len, partials, div_tables, div_partials, order);
}

// The rank lists of the sparse set (or sparse mirror) `s`, for the 1 x M pass of msc_ranks_pass.hip: true when they are current. Built only
// once the same state of the set has been asked for three times (msc_objects.h).
bool rank_lists_ready(msc_ctx* ctx, const msc_hist_set* s, int* err, bool eager) {
	*err = MSC_OK;
	if (!s->sparse || s->rkl_unavailable) return false;
	if (s->rkl && s->rkl_epoch == s->list_epoch) return true;
	if (s->rkl_seen_epoch != s->list_epoch) { s->rkl_seen_epoch = s->list_epoch; s->rkl_seen = 0; }
	// (MSC_RANKS_1XM_AFTER=n: build at the n-th request instead of the third; read on every call so that a test can switch it)
	const char* after_env = getenv("MSC_RANKS_1XM_AFTER");
	const uint32_t after = after_env && atoi(after_env) > 0 ? (uint32_t)atoi(after_env) : 3u;
	// eager: the step-serial loop's own call (msc_get_close_window over a sealed store) builds at its FIRST pass -- which kernel scores a
	// candidate must not depend on how many passes its rank has seen (a rank whose window was empty for a step would otherwise switch a
	// step later than the others, and two identical sequences on two ranks would differ in the last bit of a divergence sum)
	if (!eager && ++s->rkl_seen < after) return false;
	// (no memory for the lists: the merge kernels stay -- except in the step-serial loop's own call, where it is an error: which kernel scores
	// a pass there must not depend on what one rank's allocator had left)
	auto give_up = [&]() {
		(void)hipGetLastError();
		if (eager) { *err = fail(ctx, MSC_ERR_OOM, "rank lists: out of device memory"); return false; }
		s->rkl_unavailable = true;
		return false;
	};
	if (!s->rkl_off && (hipMalloc((void**)&s->rkl_off, (s->capacity + 1) * sizeof(uint64_t)) != hipSuccess || hipMalloc((void**)&s->rkl_n, s->capacity * sizeof(uint32_t)) != hipSuccess))
		return give_up();
	if (msc_launch_rank_lists_sizes(ctx->stream, s->hdr, s->cum, s->capacity, s->rkl_n, s->rkl_off) != hipSuccess) { *err = fail(ctx, MSC_ERR_HIP, "rank lists: size pass failed"); return false; }
	uint64_t total = 0;
	if (hipMemcpyAsync(&total, s->rkl_off + s->capacity, sizeof total, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
		*err = fail(ctx, MSC_ERR_HIP, "rank lists: size read-back failed");
		return false;
	}
	if (total + 4 > s->rkl_entries) {
		if (s->rkl) (void)hipFree(s->rkl);
		s->rkl = nullptr;
		s->rkl_entries = total + total / 8 + 1024;
		if (hipMalloc((void**)&s->rkl, s->rkl_entries * sizeof(uint32_t)) != hipSuccess) { s->rkl_entries = 0; return give_up(); }
	}
	if (msc_launch_rank_lists_fill(ctx->stream, s->ent, s->cum, s->hdr, s->capacity, s->rkl_n, s->rkl_off, s->L.nbins, s->rkl) != hipSuccess) {
		*err = fail(ctx, MSC_ERR_HIP, "rank lists: fill failed");
		return false;
	}
	s->rkl_off_host.resize(s->capacity + 1);
	if (hipMemcpyAsync(s->rkl_off_host.data(), s->rkl_off, (s->capacity + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
		*err = fail(ctx, MSC_ERR_HIP, "rank lists: offsets read-back failed");
		return false;
	}
	s->rkl_epoch = s->list_epoch;
	return true;
}

// ... and its repeated-bin lists, which the long-list pass (k_pair_ranks_items) reads beside them: built at the first such pass of an epoch
int rank_multi_ready(msc_ctx* ctx, const msc_hist_set* s) {
	if (s->rkm && s->rkm_epoch == s->list_epoch) return MSC_OK;
	if (!s->rkm_off && (hipMalloc((void**)&s->rkm_off, (s->capacity + 1) * sizeof(uint64_t)) != hipSuccess || hipMalloc((void**)&s->rkm_n, s->capacity * sizeof(uint32_t)) != hipSuccess)) {
		(void)hipGetLastError();
		return fail(ctx, MSC_ERR_OOM, "repeated-bin lists: out of device memory");
	}
	HIP_TRY(ctx, msc_launch_rank_multi_sizes(ctx->stream, s->ent, s->hdr, s->capacity, s->rkm_n, s->rkm_off));
	uint64_t total = 0;
	HIP_TRY(ctx, hipMemcpyAsync(&total, s->rkm_off + s->capacity, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (total + 4 > s->rkm_entries) {
		if (s->rkm) (void)hipFree(s->rkm);
		s->rkm = nullptr;
		s->rkm_entries = total + total / 8 + 1024;
		if (hipMalloc((void**)&s->rkm, s->rkm_entries * sizeof(uint2)) != hipSuccess) {
			(void)hipGetLastError();
			s->rkm_entries = 0;
			return fail(ctx, MSC_ERR_OOM, "repeated-bin lists: out of device memory");
		}
	}
	HIP_TRY(ctx, msc_launch_rank_multi_fill(ctx->stream, s->ent, s->hdr, s->capacity, s->rkm_off, s->rkm));
	s->rkm_epoch = s->list_epoch;
	return MSC_OK;
}

// Streams the candidates once, then folds / evaluates per candidate. Chunked so the partial records stay <= 256 MiB.
bool needs_wide(const msc_hist_set* a, const msc_hist_set* b) { return needs_wide_impl(a, b); }
int run_score(msc_ctx* ctx, ScoreRequest& rq) {
	const double t_call = g_profile_calls ? now_s() : 0;
	int r = validate_pair(ctx, rq.cands, rq.qset, rq.q_slot, rq.cand_slots, rq.dev_slots ? 0 : rq.m);
	if (r) return r;
	if (rq.dev_slots && (rq.cand_slots || rq.reduce_mode < 0 || rq.m > 0xfffffff0ull)) return fail(ctx, MSC_ERR_INVALID_ARG, "run_score: a device slot list goes with a reduction only");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const msc_hist_set* cs = rq.cands;
	const MscLayout& L = cs->L;
	const uint64_t m = rq.m;
	const int nf = __builtin_popcountll(rq.feat_mask);
	const int ns = rq.model ? rq.model->h.n_singles : 0;
	const int nc = rq.model ? rq.model->h.n_combos : 0;
	uint64_t want = rq.feat_mask;
	if (rq.model) for (int i = 0; i < ns; i++) want |= rq.model->h.single_flag[i];
	const bool need_div = (want & MSC_FEAT_DIV) != 0 && !rq.only_tiles;
	const bool need_grp = (want & MSC_FEAT_GROUPS) != 0 && !rq.only_tiles;      // sim_mm / rre_k_r: 4-bin group statistics
	const int tb = msc_div_table_dim(L);
	const bool wide = needs_wide(rq.cands, rq.qset);
	ctx->tiles_ms_accum = 0.f;
	ctx->tiles_launches = 0;
	ctx->have_timing = false;
	ctx->last_kernel = cs->sparse ? "k_pair_sparse" : wide ? "k_pair_tiles_wide" : "k_pair_tiles";
	ctx->last_query_tile = 1;
	if (m == 0) {
		if (rq.reduce_host) { rq.reduce_host->best_pos = rq.reduce_mode == MSC_REDUCE_GET_CLOSE ? -1 : 0; rq.reduce_host->best_sim = rq.reduce_mode == MSC_REDUCE_GET_CLOSE ? -1.0 : DBL_MIN;
		                      rq.reduce_host->any_close = 0; rq.reduce_host->n_close = 0; rq.reduce_host->first_error = 0; }
		return MSC_OK;
	}
	const bool sp = cs->sparse;
	// The divergence statistics of a DENSE set are scored on its sparse mirror by the same merge kernels a sparse set uses (one
	// evaluation order in every route); the dense streaming kernel then only produces the integer reductions. Histograms too small
	// for the sparse layout (< 64 KiB) keep the table form inside the streaming kernel.
	const msc_hist_set *c_sp = nullptr, *q_sp = nullptr;
	// r04: a DENSE set's 1 x M pass runs over its sparse mirror too whenever the mirror exists (histograms of 64 KiB and more, narrow
	// range): the merge kernels read 8 bytes per counted k-mer where k_pair_tiles streams 4^k bins (cfg2: ~8 KB against 1 MiB per
	// candidate), and return the same integer reductions bit for bit (test_sparse_sets_equal_dense_sets). The streaming kernel keeps
	// the histograms without a list form, the wide range, and msc_mean_nearest's pass against a mean (only_tiles).
	static const bool no_mirror_env = getenv("MSC_NO_SPARSE_MIRROR") != nullptr;
	const bool no_mirror_pass = no_mirror_env || !ctx->mirror_pass;
	bool via = false;
	if (sp) { c_sp = cs; q_sp = rq.qset; }
	else if (need_div || need_grp || (!wide && !rq.only_tiles && !no_mirror_pass && L.nbins == L.padded_bins)) {
		if ((r = ensure_sparse_mirror(ctx, cs, &c_sp)) || (r = ensure_sparse_mirror(ctx, rq.qset, &q_sp))) return r;
		if (!c_sp || !q_sp) c_sp = q_sp = nullptr;
		via = c_sp != nullptr && !wide && !rq.only_tiles && !no_mirror_pass;
	}
	const bool lists = sp || via;          // the pass is a merge of two lists
	// sim_mm / rre_k_r: from the lists where they exist, else (histograms under 64 KiB) by the dense group kernels -- a given (k, dtype)
	// always takes the same one of the two, so a pair has one evaluation order in every route
	const bool grp_dense = need_grp && !c_sp;
	if (grp_dense && std::max(rq.cands->max_count, rq.qset->max_count) > 0xffffffffull)
		return fail(ctx, MSC_ERR_UNSUPPORTED, "sim_mm / rre_k_r: counts above 2^32 - 1 are not supported");
	const bool mirror_div = need_div && !lists && c_sp != nullptr;      // (the mirror may be here for the group statistics alone)
	const bool inline_div = need_div && !lists && !mirror_div;       // table form inside k_pair_tiles / direct form inside the wide kernel
	const SparseKernel spk = c_sp ? pick_sparse_kernel(c_sp, q_sp, rq.q_slot, std::max(rq.cands->max_count, rq.qset->max_count), wide) : SPK_GENERIC;
	if (lists) ctx->last_kernel = spk == SPK_MP && !need_div && msc_sparse_wl_fits(q_sp->hdr_host[rq.q_slot].nnz, c_sp->max_nnz) ? "k_pair_sparse_wl" : sparse_kernel_name(spk);
	// r04: up to k = 9 the integer statistics of a list pass come from RANK lists -- no merge (msc_ranks_pass.hip): the query's histogram as
	// two bits per bin in LDS, the candidates' k-mers streamed at 4 bytes each. Same records as the merge kernels, bit for bit.
	const bool no_rank_pass = getenv("MSC_NO_RANKS_1XM") != nullptr;          // (read on every call: tests compare both routes in one process)
	bool rank_pass = false;
	const uint64_t q_kmers = rq.qset->max_sum >= L.nbins ? rq.qset->max_sum - L.nbins : ~0ull;          // bound on the k-mers of any histogram of the query's set
	// (the divergence statistics too: bins counted per cell of (candidate's count, query's count) by k_pair_ranks_items, evaluated per
	// candidate in one fixed order by k_rank_items_finish -- msc_ranks_pass.hip; MSC_NO_RANKS_DIV keeps such passes on the merge kernel)
	const bool no_rank_div = getenv("MSC_NO_RANKS_DIV") != nullptr;
	// ... and only in the step-serial loop's own call (msc_get_close_window: rq.close_list) unless MSC_RANKS_DIV asks for it everywhere: the
	// two FP64 sums of the rank form add the same terms in another order than the merge kernel's (they agree to ~1e-15 relative), and every
	// OTHER route -- 1 x M by slot list, Q x M, the batched update stage, dense or sparse -- keeps returning bit-identical values for a pair
	// (test_divergence_statistics_are_the_same_in_every_route). Within a window pass all candidates come from one kernel, so
	// ties among them are decided as before.
	const bool rank_div_wanted = rq.close_list.pos != nullptr || getenv("MSC_RANKS_DIV") != nullptr;
	// Long lists (more than 8 192 k-mers on either side) go through k_pair_ranks_items: the unit of work is a round of 1 024 entries of a
	// candidate, so a window of a few thousand long candidates still fills the chip (MSC_NO_RANKS_ITEMS: such passes stay on the merge kernel
	// when they carry divergence statistics, on k_pair_ranks_1xm otherwise).
	const uint64_t c_kmers = c_sp && c_sp->max_sum >= L.nbins ? c_sp->max_sum - L.nbins : ~0ull;          // bound on the k-mers of any candidate
	// (which of the two rank kernels takes a pass is decided by the QUERY alone -- its stored bins, the same number on every rank of a sharded
	// run and in a one-rank run -- never by a bound of the set or shard at hand: the two kernels add the divergence terms in different orders,
	// and two identical sequences scored for one query must not come out one bit apart because they sit on different ranks)
	// r05: a pass that carries the divergence statistics takes the items kernel whatever the query's length (cfg5's shape, the 3 461 passes
	// of short queries among 14 719: 22 us against 137 + 31 + 9 of k_pair_ranks_1xm's divergence form, its finish and the query's two
	// preparation kernels); without them a short query's pass stays with k_pair_ranks_1xm (1 kb +- 100, 13 300 candidates per pass: 31 us
	// against 35 + 10 + 6.5). (MSC_RANKS_ITEMS_FROM=n: the bound for both.)
	static const int items_from = [] { const char* e = getenv("MSC_RANKS_ITEMS_FROM"); return e ? atoi(e) : -1; }();
	const bool long_lists = q_sp && q_sp->hdr_host[rq.q_slot].nnz > (uint32_t)(items_from >= 0 ? items_from : need_div ? 0 : 2000);
	// Rounds of 1 024 entries that cover the longest list THIS PASS can meet: a candidate the length window drops takes no
	// round, so inside a window no list is longer than max_len k-mers (nor the query's own); only without a window does the set's bound
	// count. One 50 Mb scaffold among short sequences used to size -- and clear, every step -- the accumulators of every pass for 50 000
	// rounds per candidate. And a pass whose scratch would still pass 2 GiB stays with the merge kernels instead of failing with OOM.
	uint64_t pass_kmers = std::max(q_kmers, c_kmers);
	if (rq.use_window && long_lists) {
		uint64_t q_len = 0;
		if ((r = slot_length(ctx, rq.qset, rq.q_slot, &q_len))) return r;
		pass_kmers = std::min(pass_kmers, std::max(rq.max_len, q_len));
	}
	const uint64_t pass_rounds = (pass_kmers + msc_ranks_items_round() - 1) / msc_ranks_items_round();
	const bool items_ok = getenv("MSC_NO_RANKS_ITEMS") == nullptr && c_kmers < (1ull << 26) && pass_rounds < (1ull << 26) && m * (64 + pass_rounds * 3 * 280) <= (2048ull << 20);
	const bool div_fits = !need_div || (!no_rank_div && rank_div_wanted && long_lists && items_ok);          // (only the items kernel carries the divergence statistics)
	bool rank_items = false;
	uint32_t rank_rounds = 0;
	if (lists && div_fits && !rq.only_tiles && spk == SPK_MP && !no_rank_pass && q_kmers <= msc_ranks_pass_query_cap() && msc_ranks_pass_lds(L.nbins, q_kmers) != 0) {
		int e = MSC_OK;
		rank_pass = rank_lists_ready(ctx, c_sp, &e, rq.close_list.pos != nullptr);
		if (e) return e;
		if (rank_pass && !ctx->rk_guard) {
			HIP_TRY(ctx, hipHostMalloc((void**)&ctx->rk_guard, 64, hipHostMallocDefault));
			*ctx->rk_guard = 0;
		}
		rank_items = rank_pass && long_lists && items_ok;
		if (rank_items) {
			if ((r = rank_multi_ready(ctx, c_sp))) return r;
			rank_rounds = (uint32_t)pass_rounds;
			if ((r = ensure(ctx, ctx->rk_q, ((q_kmers + 255) & ~255ull) * sizeof(uint32_t) + 1024))) return r;
		} else if (rank_pass && msc_ranks_pass_query_scratch(q_kmers) && (r = ensure(ctx, ctx->rk_q, msc_ranks_pass_query_scratch(q_kmers) * sizeof(uint32_t)))) return r;
		if (rank_pass) ctx->last_kernel = rank_items ? "k_pair_ranks_items" : "k_pair_ranks_1xm";
	}
	// a sparse set's integer statistics through the merge-path kernel: a short window is shared out, several waves per candidate
	// (never the divergence form: its FP64 sums keep one evaluation order whatever the window)
	// ... and so is the divergence form (sparse sets and the mirror pass of dense ones): its FP64 sums leave per granule of the merged
	// order and are added in granule order by the epilogue, whatever the number of waves that shared a pair
	const uint64_t mp_entries = c_sp ? (uint64_t)q_sp->hdr_host[rq.q_slot].nnz + c_sp->max_nnz : 0;
	const uint32_t mp_parts = c_sp && !rank_pass && spk == SPK_MP && (need_div ? true : lists && !msc_sparse_wl_fits(q_sp->hdr_host[rq.q_slot].nnz, c_sp->max_nnz))
	                              ? msc_sparse_mp_parts((uint32_t)std::min<uint64_t>(m, 0xffffffffu), mp_entries, ctx->num_cus, need_div) : 1;
	const uint32_t SPN = sparse_records(spk, mp_parts);               // records per candidate the merge kernel writes
	const bool rank_div = rank_pass && need_div;
	const uint32_t DVN = rank_div ? 1 : div_records(spk, mp_entries);                // ... and {jd, js} records per candidate
	const uint32_t PS = lists ? SPN : L.S;                            // partial records per candidate
	ctx->last_partial_stride = PS;
	uint64_t chunk = (256ull << 20) / ((uint64_t)PS * sizeof(MscPartial));
	chunk = std::max<uint64_t>(chunk, 1024);
	if (rq.reduce_mode >= 0 || rq.only_tiles) chunk = m;      // reductions run over the whole window in one piece
	chunk = std::min(chunk, m);

	if ((r = ensure(ctx, ctx->partials, chunk * PS * sizeof(MscPartial))) != MSC_OK) return r;
	if (rq.cand_slots && !(rq.slots_uploaded && ctx->slots.cap >= m * sizeof(uint32_t))) {
		if ((r = ensure(ctx, ctx->slots, m * sizeof(uint32_t))) != MSC_OK) return r;
		if ((r = ensure_pinned(ctx, ctx->pin_up, m * sizeof(uint32_t))) != MSC_OK) return r;
		memcpy(ctx->pin_up.p, rq.cand_slots, m * sizeof(uint32_t));      // the previous call's copy has completed: every call ends in a sync
		HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, ctx->pin_up.p, m * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	}
	// (long lists: a record and a spot-term slot per item, nothing to clear; the pass's counters in two sets used in turn, each pass clearing the other's)
	if (rank_items) {
		if ((r = ensure(ctx, ctx->rk_acc, msc_ranks_items_rec_bytes(chunk, rank_rounds))) || (r = ensure(ctx, ctx->rk_items, msc_ranks_items_list_bytes(chunk, rank_rounds)))) return r;
		if (!ctx->rk_counters.p) {
			if ((r = ensure(ctx, ctx->rk_counters, 32 * sizeof(uint32_t)))) return r;
			HIP_TRY(ctx, hipMemsetAsync(ctx->rk_counters.p, 0, 32 * sizeof(uint32_t), ctx->stream));
		}
		if (ctx->rk_table_words != msc_ranks_items_table_words(L.nbins)) {          // (another k: both sets start out zero again)
			ctx->rk_table_words = msc_ranks_items_table_words(L.nbins);
			if ((r = ensure(ctx, ctx->rk_tables, 2 * (size_t)ctx->rk_table_words * sizeof(uint32_t)))) return r;
			HIP_TRY(ctx, hipMemsetAsync(ctx->rk_tables.p, 0, 2 * (size_t)ctx->rk_table_words * sizeof(uint32_t), ctx->stream));
		}
	}
	if (rank_div && (r = ensure(ctx, ctx->rk_big, ((size_t)q_sp->hdr_host[rq.q_slot].nnz + 1) * sizeof(uint32_t)))) return r;
	if (need_div) {
		if (!rank_div && (r = ensure(ctx, ctx->div_tables, chunk * (c_sp ? 256 : tb * tb) * 16)) != MSC_OK) return r;
		if ((r = ensure(ctx, ctx->div_partials, chunk * (c_sp ? DVN : PS) * 16)) != MSC_OK) return r;
		if (mirror_div && (r = ensure(ctx, ctx->sp_partials, chunk * SPN * sizeof(MscPartial))) != MSC_OK) return r;
	}
	if (need_grp) {
		if ((r = ensure(ctx, ctx->grp_pairs, chunk * 32 * sizeof(double))) != MSC_OK) return r;
		if ((r = ensure(ctx, ctx->grp_self, (chunk + 1) * 16 * sizeof(double))) != MSC_OK) return r;      // [candidates][16] then the query's 16
	}
	if (!rq.only_tiles) {
		if ((r = ensure(ctx, ctx->pair_out, chunk * sizeof(MscPairOut))) != MSC_OK) return r;
		if (rq.raw_out && (r = ensure(ctx, ctx->raw, chunk * nf * sizeof(double))) != MSC_OK) return r;
		if (rq.singles_out && (r = ensure(ctx, ctx->singles, chunk * ns * sizeof(double))) != MSC_OK) return r;
		if (rq.combos_out && (r = ensure(ctx, ctx->combos, chunk * nc * sizeof(double))) != MSC_OK) return r;
		if ((r = ensure(ctx, ctx->flags, 64 + chunk)) != MSC_OK) return r;      // [reduce record (64 B)][close flags]: one copy back
		if ((r = ensure(ctx, ctx->reduce_out, sizeof(MscReduceOut))) != MSC_OK) return r;
	}
	const uint8_t* q_bins = sp ? nullptr : rq.qset->bins + rq.q_slot * rq.qset->L.slot_bytes;
	const uint8_t* q_scal = rq.qset->scalars + rq.q_slot * rq.qset->scalar_stride;
	std::vector<MscPairOut> po_host;
	int first_err = 0;
	if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_all0, ctx->stream));
	if (g_profile_calls) { ctx->prof_calls++; ctx->prof_cands += m; ctx->prof_prep += now_s() - t_call; }
	bool fold_on_host = false;
	uint32_t n_parts_host = 0;
	for (uint64_t off = 0; off < m; off += chunk) {
		const double t_issue = g_profile_calls ? now_s() : 0;
		const uint32_t mc = (uint32_t)std::min(chunk, m - off);
		const uint32_t* d_slots = rq.dev_slots ? rq.dev_slots : rq.cand_slots ? (const uint32_t*)ctx->slots.p + off : nullptr;
		const uint8_t* c_bins = sp ? nullptr : cs->bins + (d_slots ? 0 : off * L.slot_bytes);
		const uint8_t* c_scal = cs->scalars + (d_slots ? 0 : off * cs->scalar_stride);
		if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_tiles0, ctx->stream));
		if (lists && g_profile_calls) {
			if (!ctx->prof_nnz.p) { if ((r = ensure(ctx, ctx->prof_nnz, 32))) return r; HIP_TRY(ctx, hipMemsetAsync(ctx->prof_nnz.p, 0, 32, ctx->stream)); }
			HIP_TRY(ctx, msc_launch_sparse_nnz_sum(ctx->stream, c_sp->hdr, cs->scalars, cs->scalar_stride, d_slots, off, mc, rq.use_window, rq.min_len, rq.max_len,
			                                       (uint64_t*)ctx->prof_nnz.p + (rank_pass ? 2 : 0)));
			ctx->prof_q_nnz += q_sp->hdr_host[rq.q_slot].nnz;
		}
		if (lists && rank_items) {
			MscRankDiv dv{(uint32_t*)ctx->rk_big.p, q_scal, rq.order, (double*)ctx->div_partials.p};
			HIP_TRY(ctx, msc_launch_pair_ranks_items(ctx->stream, c_sp->rkl, c_sp->rkl_off, c_sp->rkl_n, c_sp->rkm, c_sp->rkm_off, c_sp->rkm_n, cs->scalars + (d_slots ? 0 : off * cs->scalar_stride),
			                                         cs->scalar_stride, d_slots, off, mc, q_sp->ent, q_sp->cum, q_sp->hdr + rq.q_slot, L.nbins, rq.use_window, rq.min_len, rq.max_len,
			                                         (MscPartial*)ctx->partials.p, ctx->num_cus, (uint32_t*)ctx->rk_q.p, rank_rounds, ctx->rk_acc.p, rank_div ? &dv : nullptr, q_kmers, ctx->rk_guard,
			                                         ctx->rk_items.p, (uint32_t*)ctx->rk_counters.p, (uint32_t*)ctx->rk_tables.p, (int)(ctx->rk_turn++ & 1u),
			                                         q_sp == c_sp && rq.q_slot < c_sp->capacity ? c_sp->rkl + c_sp->rkl_off_host[rq.q_slot] : nullptr));
		} else if (lists && rank_pass) {
			HIP_TRY(ctx, msc_launch_pair_ranks_1xm(ctx->stream, c_sp->rkl, c_sp->rkl_off, c_sp->rkl_n, cs->scalars + (d_slots ? 0 : off * cs->scalar_stride), cs->scalar_stride, d_slots, off, mc,
			                                       q_sp->ent, q_sp->cum, q_sp->hdr + rq.q_slot, L.nbins, rq.use_window, rq.min_len, rq.max_len, (MscPartial*)ctx->partials.p, ctx->num_cus, q_kmers, ctx->rk_guard, (uint32_t*)ctx->rk_q.p));
		} else if (lists) {
			HIP_TRY(ctx, launch_sparse_pass(ctx, spk, c_sp, cs->scalars, cs->scalar_stride, d_slots, off, mc, q_sp, rq.q_slot, q_scal, L.nbins, rq.use_window, rq.min_len,
			                                rq.max_len, (MscPartial*)ctx->partials.p, need_div ? ctx->div_tables.p : nullptr, need_div ? ctx->div_partials.p : nullptr, rq.order,
			                                mp_parts, DVN));
		} else if (wide) {
			HIP_TRY(ctx, msc_launch_pair_tiles_wide(ctx->stream, L, cs->dtype, c_bins, c_scal, d_slots, mc, q_bins, q_scal, rq.use_window, rq.min_len,
			                                        rq.max_len, (MscPartial*)ctx->partials.p, ctx->num_cus, inline_div ? ctx->div_partials.p : nullptr, rq.order));
		} else {
			HIP_TRY(ctx, msc_launch_pair_tiles(ctx->stream, L, cs->dtype, c_bins, c_scal, d_slots, mc, q_bins, q_scal, rq.use_window, rq.min_len,
			                                   rq.max_len, (MscPartial*)ctx->partials.p, ctx->num_cus, inline_div ? ctx->div_tables.p : nullptr,
			                                   inline_div ? ctx->div_partials.p : nullptr, rq.order));
		}
		if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_tiles1, ctx->stream));
		if (mirror_div)        // the divergence sums of this chunk, from the lists of the same slots (outside the streaming kernel's timing)
			HIP_TRY(ctx, launch_sparse_pass(ctx, spk, c_sp, cs->scalars, cs->scalar_stride, d_slots, off, mc, q_sp, rq.q_slot, q_scal, L.nbins, rq.use_window, rq.min_len,
			                                rq.max_len, (MscPartial*)ctx->sp_partials.p, ctx->div_tables.p, ctx->div_partials.p, rq.order, mp_parts, DVN));
		if (grp_dense) {
			HIP_TRY(ctx, msc_launch_pair_groups_dense(ctx->stream, L, cs->dtype, c_bins, c_scal, cs->scalar_stride, d_slots, mc, q_bins, rq.use_window, rq.min_len, rq.max_len,
			                                          (double*)ctx->grp_pairs.p));
			HIP_TRY(ctx, msc_launch_self_markov_dense(ctx->stream, L, cs->dtype, cs->bins, d_slots, off, mc, (double*)ctx->grp_self.p));
			HIP_TRY(ctx, msc_launch_self_markov_dense(ctx->stream, rq.qset->L, rq.qset->dtype, rq.qset->bins, nullptr, rq.q_slot, 1, (double*)ctx->grp_self.p + (uint64_t)chunk * 16));
		} else if (need_grp) {
			HIP_TRY(ctx, msc_launch_pair_sparse_groups(ctx->stream, c_sp->ent, c_sp->hdr + (d_slots ? 0 : off), cs->scalars + (d_slots ? 0 : off * cs->scalar_stride),
			                                           cs->scalar_stride, d_slots, mc, q_sp->ent, q_sp->hdr + rq.q_slot, rq.use_window, rq.min_len, rq.max_len,
			                                           (double*)ctx->grp_pairs.p));
			HIP_TRY(ctx, msc_launch_sparse_self_markov(ctx->stream, c_sp->ent, c_sp->hdr, d_slots, off, mc, (double*)ctx->grp_self.p));
			HIP_TRY(ctx, msc_launch_sparse_self_markov(ctx->stream, q_sp->ent, q_sp->hdr, nullptr, rq.q_slot, 1, (double*)ctx->grp_self.p + (uint64_t)chunk * 16));
		}
		if (rq.only_tiles) {
			if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_all1, ctx->stream));
			break;
		}
		MscEpilogueArgs ea;
		memset(&ea, 0, sizeof ea);
		ea.partials = (const MscPartial*)ctx->partials.p;
		ea.div_partials = inline_div ? ctx->div_partials.p : nullptr;
		if (need_div && c_sp) { ea.div_direct = (const double*)ctx->div_partials.p; ea.div_direct_n = DVN; ea.div_base = L.nbins; }
		if (need_grp) { ea.grp_pairs = (const double*)ctx->grp_pairs.p; ea.grp_self_c = (const double*)ctx->grp_self.p; ea.grp_self_q = (const double*)ctx->grp_self.p + (uint64_t)chunk * 16; }
		ea.S = PS;
		ea.sparse_base = lists ? L.nbins : 0;
		ea.m = mc;
		ea.cand_scalars = c_scal;
		ea.cand_scalar_stride = cs->scalar_stride;
		ea.cand_slots = d_slots;
		ea.q_scalars = q_scal;
		ea.nbins = L.nbins;
		ea.dtype = cs->dtype;
		ea.order = rq.order;
		ea.use_window = rq.use_window;
		ea.min_len = rq.min_len;
		ea.max_len = rq.max_len;
		ea.feat_mask = rq.feat_mask;
		ea.raw_out = rq.raw_out ? (double*)ctx->raw.p : nullptr;
		ea.model = rq.model ? rq.model->d : nullptr;
		ea.singles_out = rq.singles_out ? (double*)ctx->singles.p : nullptr;
		ea.combos_out = rq.combos_out ? (double*)ctx->combos.p : nullptr;
		ea.pair_out = (MscPairOut*)ctx->pair_out.p;
		const bool need_po = rq.sum_out || rq.csum_out || rq.combo0_out || rq.status_out || (rq.flags_out && rq.reduce_mode < 0) || rq.reduce_mode < 0;
		// a reduced pass that returns nothing per pair: epilogue, first reduce stage and the window's close pass in one kernel
		const bool fused = rq.reduce_mode >= 0 && !need_po && !rq.raw_out && !rq.singles_out && !rq.combos_out && PS <= 4 && getenv("MSC_NO_FUSED_REDUCE") == nullptr;
		if (fused) ea.pair_out = nullptr;
		else HIP_TRY(ctx, msc_launch_epilogue(ctx->stream, ea));
		if (rq.reduce_mode >= 0) {
			// the reduce kernel writes its record and the close flags straight into page-locked host memory the device can address
			// (no copy command behind the kernel): [reduce record (64 B)][close flags]
			constexpr size_t kRo = 64;
			static_assert(sizeof(MscReduceOut) <= kRo, "the reduce record sits in front of the flags");
			if ((r = ensure_pinned(ctx, ctx->pin_down, kRo + mc)) != MSC_OK) return r;
			uint8_t* down = nullptr;
			HIP_TRY(ctx, hipHostGetDevicePointer((void**)&down, ctx->pin_down.p, 0));
			if ((r = ensure(ctx, ctx->reduce_parts, msc_reduce_scratch_bytes())) != MSC_OK) return r;
			uint8_t* d_flags = rq.dev_flags_out ? rq.dev_flags_out : rq.flags_out ? down + kRo : nullptr;
			if (fused) {
				// (the step-serial loop's own call: the workgroups' parts go to page-locked memory and are folded here, behind the wait for the
				// stream -- k_pair_reduce_fold2 was one of a pass's launches)
				ReducePart* host_parts = nullptr;
				if (rq.close_list.pos && rq.close_list.out_host && off == 0 && mc == m) {
					if ((r = ensure_pinned(ctx, ctx->pin_parts, 1024 * sizeof(ReducePart))) != MSC_OK) return r;
					HIP_TRY(ctx, hipHostGetDevicePointer((void**)&host_parts, ctx->pin_parts.p, 0));
				}
				fold_on_host = host_parts != nullptr;
				HIP_TRY(ctx, msc_launch_epilogue_reduce(ctx->stream, ea, rq.reduce_mode, rq.reduce_begin, d_flags, (MscReduceOut*)down, ctx->reduce_parts.p, rq.close_list, host_parts, &n_parts_host));
				if (rq.after_reduce && !rq.close_list.pos) HIP_TRY(ctx, rq.after_reduce((const MscReduceOut*)down));
			} else {
				HIP_TRY(ctx, msc_launch_reduce(ctx->stream, (const MscPairOut*)ctx->pair_out.p, mc, rq.reduce_mode, rq.reduce_begin, d_flags, (MscReduceOut*)down, ctx->reduce_parts.p,
				                               rq.reduce_mode == MSC_REDUCE_GET_CLOSE ? rq.close_list.pos : nullptr));          // (a window pass: ties by window position)
				if (rq.after_reduce) HIP_TRY(ctx, rq.after_reduce((const MscReduceOut*)down));
			}
		}
		if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_all1, ctx->stream));
		if (rq.raw_out) HIP_TRY(ctx, hipMemcpyAsync(rq.raw_out + off * nf, ctx->raw.p, (size_t)mc * nf * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
		if (rq.singles_out) HIP_TRY(ctx, hipMemcpyAsync(rq.singles_out + off * ns, ctx->singles.p, (size_t)mc * ns * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
		if (rq.combos_out) HIP_TRY(ctx, hipMemcpyAsync(rq.combos_out + off * nc, ctx->combos.p, (size_t)mc * nc * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
		if (need_po) {
			po_host.resize(mc);
			HIP_TRY(ctx, hipMemcpyAsync(po_host.data(), ctx->pair_out.p, (size_t)mc * sizeof(MscPairOut), hipMemcpyDeviceToHost, ctx->stream));
		}
		const double t_wait = g_profile_calls ? now_s() : 0;
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (g_profile_calls) { ctx->prof_issue += t_wait - t_issue; ctx->prof_wait += now_s() - t_wait; }
		if (rq.reduce_mode >= 0 && fold_on_host) {
			uint32_t wpos = 0;
			msc_reduce_fold_host((const ReducePart*)ctx->pin_parts.p, n_parts_host, rq.reduce_mode, rq.reduce_host, &wpos);
			rq.close_list.out_host[0] = wpos;
			if (rq.flags_out) memcpy(rq.flags_out, (const uint8_t*)ctx->pin_down.p + 64, mc);
		} else if (rq.reduce_mode >= 0) {
			constexpr size_t kRo = 64;
			memcpy(rq.reduce_host, ctx->pin_down.p, sizeof(MscReduceOut));
			if (rq.flags_out) memcpy(rq.flags_out, (const uint8_t*)ctx->pin_down.p + kRo, mc);
		}
		float t = 0;
		if (ctx->timing && hipEventElapsedTime(&t, ctx->ev_tiles0, ctx->ev_tiles1) == hipSuccess) { ctx->tiles_ms_accum += t; ctx->tiles_launches++; ctx->have_timing = true; }
		if (need_po) {
			for (uint32_t i = 0; i < mc; i++) {
				const MscPairOut& p = po_host[i];
				if (rq.sum_out) rq.sum_out[off + i] = p.sum;
				if (rq.csum_out) rq.csum_out[off + i] = p.csum;
				if (rq.combo0_out) rq.combo0_out[off + i] = p.combo0;
				if (rq.status_out) rq.status_out[off + i] = p.status;
				if (rq.flags_out && rq.reduce_mode < 0) rq.flags_out[off + i] = (p.status == 0 && p.close) ? 1 : 0;
				if (p.status < 0 && p.status < first_err) first_err = p.status;
			}
		}
	}
	if (rq.only_tiles) return MSC_OK;
	if (rank_pass && *ctx->rk_guard) {
		*ctx->rk_guard = 0;
		return fail(ctx, MSC_ERR_HIP, "rank pass: the query's list is longer than its set's bound (max_sum not maintained by a writer of that set)");
	}
	if (rq.reduce_host && rq.reduce_host->first_error < first_err) first_err = rq.reduce_host->first_error;
	if (first_err == MSC_ERR_ZERO_LENGTH) return fail(ctx, first_err, "length_difference: a point has length 0 (the reference throws 123, predict/Feature.cpp:878-886)");
	if (first_err == MSC_ERR_NAN) return fail(ctx, first_err, "normalisation produced NaN (the reference throws, predict/Feature.cpp:143-146)");
	if (first_err < 0) return fail(ctx, first_err, "feature evaluation failed with status %d", first_err);
	return MSC_OK;
}


static int run_score_fwd(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, uint64_t m, const msc_hist_set* rs) {
	ScoreRequest rq;
	rq.cands = set; rq.cand_slots = member_slots; rq.m = m; rq.qset = rs; rq.q_slot = 0; rq.only_tiles = true;
	rq.slots_uploaded = member_slots != nullptr;          // (mean_nearest_sparse copied them at its head)
	return run_score(ctx, rq);
}

extern "C" int msc_pair_features_raw(msc_ctx* ctx, const msc_hist_set* cands, const uint32_t* cand_slots, uint64_t m, const msc_hist_set* qset,
                                     uint64_t q_slot, int order, uint64_t feat_mask, double* raw_out) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	if (!raw_out && m) return fail(ctx, MSC_ERR_INVALID_ARG, "raw_out is NULL");
	if (feat_mask == 0 || (feat_mask & ~kSupportedFeats))
		return fail(ctx, MSC_ERR_UNSUPPORTED, "feat_mask 0x%llx holds statistics outside the GPU path (supported 0x%llx)", (unsigned long long)feat_mask,
		            (unsigned long long)kSupportedFeats);
	ScoreRequest rq;
	rq.cands = cands; rq.cand_slots = cand_slots; rq.m = m; rq.qset = qset; rq.q_slot = q_slot; rq.order = order;
	rq.feat_mask = feat_mask; rq.raw_out = raw_out;
	return run_score(ctx, rq);
}

extern "C" int msc_score(msc_ctx* ctx, const msc_model* model, const msc_hist_set* cands, const uint32_t* cand_slots, uint64_t m,
                         const msc_hist_set* qset, uint64_t q_slot, int order, double* singles_out, double* combos_out, double* sum_out,
                         double* csum_out) {
	if (!ctx || !model || model->ctx != ctx) return MSC_ERR_INVALID_ARG;
	ScoreRequest rq;
	rq.model = model; rq.cands = cands; rq.cand_slots = cand_slots; rq.m = m; rq.qset = qset; rq.q_slot = q_slot; rq.order = order;
	rq.singles_out = singles_out; rq.combos_out = combos_out; rq.sum_out = sum_out; rq.csum_out = csum_out;
	return run_score(ctx, rq);
}

extern "C" int msc_get_close(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* cands, const uint32_t* cand_slots,
                             uint64_t m, const msc_hist_set* qset, uint64_t q_slot, uint8_t* close_flags, int64_t* best_pos, double* best_sim,
                             int* is_min) {
	if (!ctx || !model || model->ctx != ctx || !qset) return MSC_ERR_INVALID_ARG;
	if (m && !close_flags) return fail(ctx, MSC_ERR_INVALID_ARG, "close_flags is NULL");
	int r = check_slot(ctx, qset, q_slot);
	if (r) return r;
	uint64_t q_len = 0;
	if ((r = slot_length(ctx, qset, q_slot, &q_len))) return r;
	ScoreRequest rq;
	rq.model = model; rq.cands = cands; rq.cand_slots = cand_slots; rq.m = m; rq.qset = qset; rq.q_slot = q_slot;
	rq.order = MSC_ORDER_CAND_FIRST;                              // feat->compute(*pt, *p), cluster/Trainer.cpp:49
	rq.use_window = 1;
	rq.min_len = (uint64_t)((double)q_len * cutoff);              // uint64_t min_len = p->get_length() * cutoff;  :39
	rq.max_len = (uint64_t)((double)q_len / cutoff);              // uint64_t max_len = p->get_length() / cutoff;  :40
	rq.flags_out = close_flags;
	rq.reduce_mode = MSC_REDUCE_GET_CLOSE;
	MscReduceOut ro;
	memset(&ro, 0, sizeof ro);
	rq.reduce_host = &ro;
	if ((r = run_score(ctx, rq))) return r;
	if (best_pos) *best_pos = ro.best_pos;
	if (best_sim) *best_sim = ro.best_sim;
	if (is_min) *is_min = ro.any_close ? 0 : 1;
	return MSC_OK;
}

extern "C" int msc_filter(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centre_set, uint64_t centre_slot,
                          const msc_hist_set* pts, const uint32_t* pt_slots, uint64_t m, uint8_t* keep, uint64_t* n_kept) {
	if (!ctx || !model || model->ctx != ctx || !centre_set) return MSC_ERR_INVALID_ARG;
	if (m && !keep) return fail(ctx, MSC_ERR_INVALID_ARG, "keep is NULL");
	int r = check_slot(ctx, centre_set, centre_slot);
	if (r) return r;
	uint64_t c_len = 0;
	if ((r = slot_length(ctx, centre_set, centre_slot, &c_len))) return r;
	ScoreRequest rq;
	rq.model = model; rq.cands = pts; rq.cand_slots = pt_slots; rq.m = m; rq.qset = centre_set; rq.q_slot = centre_slot;
	rq.order = MSC_ORDER_QUERY_FIRST;                             // classify(p, pt.first), cluster/Trainer.cpp:133
	rq.use_window = 1;
	rq.min_len = (uint64_t)((double)c_len * trainer_get_id(cutoff));          // :126-127
	rq.max_len = (uint64_t)((double)c_len / trainer_get_id(cutoff));
	rq.flags_out = keep;                                          // kept  <=>  in window && round(classify) != 0
	if ((r = run_score(ctx, rq))) return r;
	if (n_kept) { uint64_t n = 0; for (uint64_t i = 0; i < m; i++) n += keep[i]; *n_kept = n; }
	return MSC_OK;
}

extern "C" int msc_merge(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots,
                         uint64_t n, int64_t current, int64_t begin, int64_t last, int64_t* best_out) {
	if (!ctx || !model || model->ctx != ctx || !centres || !best_out) return MSC_ERR_INVALID_ARG;
	if (current < 0 || (uint64_t)current >= n) return fail(ctx, MSC_ERR_INVALID_ARG, "current out of range");
	*best_out = 0;
	if (begin > last) return MSC_OK;
	if (begin < 0 || (uint64_t)last >= n) return fail(ctx, MSC_ERR_INVALID_ARG, "[begin,last] out of range");
	const uint64_t cur_slot = centre_slots ? centre_slots[current] : (uint64_t)current;
	int r = check_slot(ctx, centres, cur_slot);
	if (r) return r;
	uint64_t c_len = 0;
	if ((r = slot_length(ctx, centres, cur_slot, &c_len))) return r;
	std::vector<uint32_t> slots((size_t)(last - begin + 1));
	for (int64_t i = begin; i <= last; i++) slots[(size_t)(i - begin)] = centre_slots ? centre_slots[i] : (uint32_t)i;
	std::vector<uint8_t> flags(slots.size());
	ScoreRequest rq;
	rq.model = model; rq.cands = centres; rq.cand_slots = slots.data(); rq.m = slots.size(); rq.qset = centres; rq.q_slot = cur_slot;
	rq.order = MSC_ORDER_CAND_FIRST;                              // feat->compute(*cen, *p), cluster/Trainer.cpp:93
	rq.use_window = 1;
	rq.min_len = (uint64_t)((double)c_len * trainer_get_id(cutoff));
	rq.max_len = (uint64_t)((double)c_len / trainer_get_id(cutoff));
	rq.flags_out = flags.data();
	rq.reduce_mode = MSC_REDUCE_MERGE;
	rq.reduce_begin = begin;
	MscReduceOut ro;
	memset(&ro, 0, sizeof ro);
	rq.reduce_host = &ro;
	if ((r = run_score(ctx, rq))) return r;
	*best_out = ro.best_pos;
	return MSC_OK;
}

extern "C" int msc_search(msc_ctx* ctx, const msc_model* cls, const msc_model* reg, const msc_hist_set* db, const uint32_t* db_slots,
                          uint64_t m, const msc_hist_set* qset, uint64_t q_slot, uint8_t* close_out, double* sim_out) {
	if (!ctx || (cls && cls->ctx != ctx) || (reg && reg->ctx != ctx)) return MSC_ERR_INVALID_ARG;
	if (!cls && !reg) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_search needs a classification or a regression model");
	// work() follows pred->get_mode() (fastcar/FC_Runner.cpp:432,446-458): without a classification block every pair of the
	// window counts as close, without a regression block the similarity of a close pair is 1
	int r;
	if (cls) {
		// pred->close(pts[i], query) then pred->similarity(pts[i], query): fastcar/FC_Runner.cpp:449-455
		ScoreRequest a;
		a.model = cls; a.cands = db; a.cand_slots = db_slots; a.m = m; a.qset = qset; a.q_slot = q_slot; a.order = MSC_ORDER_CAND_FIRST;
		a.flags_out = close_out;
		if ((r = run_score(ctx, a))) return r;
	} else {
		if ((r = validate_pair(ctx, db, qset, q_slot, db_slots, m))) return r;
		if (close_out) memset(close_out, 1, m);
	}
	if (reg && sim_out) {
		ScoreRequest b;
		b.model = reg; b.cands = db; b.cand_slots = db_slots; b.m = m; b.qset = qset; b.q_slot = q_slot; b.order = MSC_ORDER_CAND_FIRST;
		b.sum_out = sim_out;
		if ((r = run_score(ctx, b))) return r;
		for (uint64_t i = 0; i < m; i++) {           // p_predict clamps to [0,1], predict/Predictor.cpp:293-298
			if (sim_out[i] < 0) sim_out[i] = 0; else if (sim_out[i] > 1) sim_out[i] = 1;
		}
	} else if (sim_out) {
		for (uint64_t i = 0; i < m; i++) sim_out[i] = 1.0;
	}
	return MSC_OK;
}

// ================================================================================================ mean + nearest

// msc_mean_nearest for sparse members (kernels and the derivation in sparse.hip)
static int run_score_fwd(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, uint64_t m, const msc_hist_set* rs);
namespace {
__global__ void k_put_words(uint32_t* __restrict__ a, const uint32_t* __restrict__ a_src, uint32_t na, uint32_t* __restrict__ b, const uint32_t* __restrict__ b_src, uint32_t nb) {
	for (uint32_t i = threadIdx.x; i < na; i += blockDim.x) a[i] = a_src[i];
	for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) b[i] = b_src[i];
}
}  // namespace
static int mean_nearest_sparse(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, uint64_t m, int64_t* nearest_pos, double* dist_out,
                               double* mean_out) {
	if (mean_out) return fail(ctx, MSC_ERR_UNSUPPORTED, "mean_out is not available for sparse sets");
	if (m > 0xfffffff0ull) return MSC_ERR_INVALID_ARG;
	if (member_slots) { for (uint64_t i = 0; i < m; i++) if (member_slots[i] >= set->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "member slot out of range"); }
	else if (m > set->capacity) return MSC_ERR_INVALID_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const MscLayout& L = set->L;
	int r;
	if (ctx->sp_acc_bins != L.nbins) {          // dense u32 accumulator, zero between calls
		if ((r = ensure(ctx, ctx->sp_acc, L.nbins * sizeof(uint32_t)))) return r;
		HIP_TRY(ctx, hipMemsetAsync(ctx->sp_acc.p, 0, L.nbins * sizeof(uint32_t), ctx->stream));
		ctx->sp_acc_bins = L.nbins;
	}
	uint64_t upper = 0;                           // the rounded mean cannot have more stored bins than the members together
	for (uint64_t i = 0; i < m; i++) upper += set->hdr_host[member_slots ? member_slots[i] : i].nnz;
	upper = std::min<uint64_t>(upper, L.nbins);
	if (!ctx->sparse_mean_set || ctx->sparse_mean_set->k != set->k || ctx->sparse_mean_set->dtype != set->dtype || ctx->sparse_mean_set->ent_capacity < upper + 1) {
		if (ctx->sparse_mean_set) { msc_hist_set_destroy(ctx->sparse_mean_set); ctx->sparse_mean_set = nullptr; }
		if ((r = msc_hist_set_create_sparse(ctx, set->k, set->dtype, 1, std::max<uint64_t>(upper + 1, 1 << 16), &ctx->sparse_mean_set))) return r;
	}
	msc_hist_set* rs = ctx->sparse_mean_set;
	const uint32_t n_chunks = (uint32_t)std::min<uint64_t>(1024, L.nbins / 256);      // multiple of 16 for every k >= 6
	const uint64_t chunk_bins = L.nbins / n_chunks;
	// r05: what the call hands to and fro in small pieces -- the chunk counts (device -> host), the chunks' offsets and running sums,
	// the floor sum, the mean's header and scalar record (host -> device), the reduce record (device -> host) -- sits in ONE page-locked
	// block the kernels address directly; eight staged copies of a few bytes to a few KiB each were a quarter of a `closest` call
	// (cfg5's shape: 141 us per call, 2.5 s of the run). Only the header and the scalar record, which the set's own arrays must hold,
	// and the member slots are still copied (from page-locked memory: no staging).
	struct PinLayout { size_t counts, off, cb, floor_sum, ro, hdr, sc, slots, bytes; } pl;
	pl.counts = 0;
	pl.off = pl.counts + (size_t)n_chunks * 3 * sizeof(uint64_t);
	pl.cb = pl.off + (size_t)n_chunks * sizeof(uint64_t);
	pl.floor_sum = pl.cb + (size_t)n_chunks * sizeof(uint64_t);
	pl.ro = pl.floor_sum + 64;
	pl.hdr = pl.ro + 64;
	pl.sc = pl.hdr + ((sizeof(MscSparseHdr) + 63) & ~(size_t)63);
	pl.slots = pl.sc + ((sizeof(MscSlotScalars) + 63) & ~(size_t)63);
	pl.bytes = pl.slots + (member_slots ? m * sizeof(uint32_t) : 0);
	if ((r = ensure_pinned(ctx, ctx->pin_mean, pl.bytes))) return r;
	uint8_t *pin_h = (uint8_t*)ctx->pin_mean.p, *pin_d = nullptr;
	HIP_TRY(ctx, hipHostGetDevicePointer((void**)&pin_d, ctx->pin_mean.p, 0));
	if (member_slots) {
		if ((r = ensure(ctx, ctx->slots, m * sizeof(uint32_t)))) return r;
		memcpy(pin_h + pl.slots, member_slots, m * sizeof(uint32_t));          // (the previous call's copy has completed: every call ends in a wait for the stream)
		HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, pin_h + pl.slots, m * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	}
	const uint32_t* d_slots = member_slots ? (const uint32_t*)ctx->slots.p : nullptr;
	uint64_t* d_counts = (uint64_t*)(pin_d + pl.counts);
	// k >= 11: the kernels of the batched form with one centre, whose sweeps visit touched 64-byte lines only
	static const bool no_groups = getenv("MSC_SPARSE_MEAN_NO_GROUPS") != nullptr;
	const bool grouped = !no_groups && L.nbins >= msc_sparse_groups_min_bins() && chunk_bins % 512 == 0 && member_slots;
	if (grouped) {
		const size_t tb = (size_t)(L.nbins >> 9) * sizeof(uint32_t);
		if (tb > ctx->sp_touched.cap) {
			if ((r = ensure(ctx, ctx->sp_touched, tb))) return r;
			HIP_TRY(ctx, hipMemsetAsync(ctx->sp_touched.p, 0, ctx->sp_touched.cap, ctx->stream));
		}
		if ((r = ensure(ctx, ctx->pair_seg, m * sizeof(uint32_t))) || (r = ensure(ctx, ctx->qslots, sizeof(uint32_t)))) return r;
		const uint32_t m32 = (uint32_t)m;
		HIP_TRY(ctx, hipMemsetAsync(ctx->pair_seg.p, 0, m * sizeof(uint32_t), ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->qslots.p, &m32, sizeof m32, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, msc_launch_sparse_scatter_batch(ctx->stream, set->ent, set->hdr, d_slots, (const uint32_t*)ctx->pair_seg.p, m32, L.nbins, (uint32_t*)ctx->sp_acc.p,
		                                             (uint32_t*)ctx->sp_touched.p));
		HIP_TRY(ctx, msc_launch_sparse_mean_count_batch(ctx->stream, set->dtype, (const uint32_t*)ctx->sp_acc.p, L.nbins, n_chunks, chunk_bins, 1, (const uint32_t*)ctx->qslots.p,
		                                                d_counts, (const uint32_t*)ctx->sp_touched.p));
	} else {
	HIP_TRY(ctx, msc_launch_sparse_scatter(ctx->stream, set->ent, set->hdr, d_slots, (uint32_t)m, (uint32_t*)ctx->sp_acc.p));
	HIP_TRY(ctx, msc_launch_sparse_mean_count(ctx->stream, set->dtype, (const uint32_t*)ctx->sp_acc.p, n_chunks, chunk_bins, (uint32_t)m, d_counts));
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	const uint64_t* counts = (const uint64_t*)(pin_h + pl.counts);
	uint64_t *off = (uint64_t*)(pin_h + pl.off), *cb = (uint64_t*)(pin_h + pl.cb);
	MscSparseHdr h{};
	uint64_t n = 0, ex = 0, fl = 0;
	const uint32_t per_sub = n_chunks / MSC_SPARSE_SUB;
	for (uint32_t c = 0; c < n_chunks; c++) {
		if (c % per_sub == 0) h.split[c / per_sub] = (uint32_t)n;
		off[c] = n;
		cb[c] = ex;
		n += counts[c * 3ull]; ex += counts[c * 3ull + 1]; fl += counts[c * 3ull + 2];
	}
	h.split[MSC_SPARSE_SUB] = (uint32_t)n;
	h.nnz = (uint32_t)n;
	h.off = 0;
	rs->ent_used = n;
	rs->hdr_host[0] = h;
	rs->list_epoch++;
	rs->max_nnz = std::max(rs->max_nnz, h.nnz);          // (every writer of hdr_host keeps max_nnz >= each list: the whole-list kernel sizes its LDS by it)
	MscSlotScalars sc;
	memset(&sc, 0, sizeof sc);
	sc.sum = L.nbins + ex;          // sum of the rounded mean's bins
	rs->max_sum = std::max<uint64_t>(rs->max_sum, sc.sum);
	sc.mag = sc.sum;
	sc.length = 1;
	*(uint64_t*)(pin_h + pl.floor_sum) = L.nbins + fl;
	memcpy(pin_h + pl.hdr, &h, sizeof h);
	memcpy(pin_h + pl.sc, &sc, sizeof sc);
	// (the header and the scalar record into the set's own arrays by ONE small kernel that reads the page-locked block: two staged copies
	// of a few dozen bytes were 9 us of blit kernels per call, a fifth of the call's kernel time)
	static_assert(sizeof(MscSparseHdr) % 4 == 0 && sizeof(MscSlotScalars) % 4 == 0, "copied word by word");
	k_put_words<<<dim3(1), dim3(64), 0, ctx->stream>>>((uint32_t*)rs->hdr, (const uint32_t*)(pin_d + pl.hdr), (uint32_t)(sizeof h / 4), (uint32_t*)rs->scalars,
	                                                  (const uint32_t*)(pin_d + pl.sc), (uint32_t)(sizeof sc / 4));
	HIP_TRY(ctx, hipGetLastError());
	const uint64_t *d_off = (const uint64_t*)(pin_d + pl.off), *d_cb = (const uint64_t*)(pin_d + pl.cb), *d_floor = (const uint64_t*)(pin_d + pl.floor_sum);
	if (grouped)
		HIP_TRY(ctx, msc_launch_sparse_mean_write_batch(ctx->stream, set->dtype, (uint32_t*)ctx->sp_acc.p, L.nbins, n_chunks, chunk_bins, 1, (const uint32_t*)ctx->qslots.p,
		                                                d_off, d_cb, rs->ent, rs->cum, (uint32_t*)ctx->sp_touched.p));
	else
	HIP_TRY(ctx, msc_launch_sparse_mean_write(ctx->stream, set->dtype, (uint32_t*)ctx->sp_acc.p, n_chunks, chunk_bins, (uint32_t)m, d_off,
	                                          d_cb, rs->ent, rs->cum));
	// members vs the rounded mean: only the |p - r| reduction of the merge kernel is used
	if ((r = run_score_fwd(ctx, set, member_slots, m, rs))) return r;
	if ((r = ensure(ctx, ctx->raw, m * sizeof(double)))) return r;
	HIP_TRY(ctx, msc_launch_distance_d(ctx->stream, (const MscPartial*)ctx->partials.p, ctx->last_partial_stride, (uint32_t)m, set->scalars, set->scalar_stride, d_slots,
	                                   rs->scalars, d_floor, (double*)ctx->raw.p, (MscReduceOut*)(pin_d + pl.ro)));
	if (dist_out) HIP_TRY(ctx, hipMemcpyAsync(dist_out, ctx->raw.p, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	*nearest_pos = ((const MscReduceOut*)(pin_h + pl.ro))->best_pos;
	return MSC_OK;
}

extern "C" int msc_mean_nearest(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, uint64_t m, int64_t* nearest_pos,
                                double* dist_out, double* mean_out) {
	if (!ctx || !set || set->ctx != ctx || !nearest_pos) return MSC_ERR_INVALID_ARG;
	if (m == 0) return fail(ctx, MSC_ERR_INVALID_ARG, "N cannot be 0 (cluster/ClusterFactory.cpp:346-348 throws)");
	if (set->sparse) return mean_nearest_sparse(ctx, set, member_slots, m, nearest_pos, dist_out, mean_out);
	if (m > 0xfffffff0ull) return MSC_ERR_INVALID_ARG;
	if (member_slots) { for (uint64_t i = 0; i < m; i++) if (member_slots[i] >= set->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "member slot out of range"); }
	else if (m > set->capacity) return MSC_ERR_INVALID_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int r;
	if (!ctx->scratch_set || ctx->scratch_set->k != set->k || ctx->scratch_set->dtype != set->dtype) {
		if (ctx->scratch_set) { msc_hist_set_destroy(ctx->scratch_set); ctx->scratch_set = nullptr; }
		if ((r = msc_hist_set_create(ctx, set->k, set->dtype, 1, &ctx->scratch_set))) return r;
	}
	msc_hist_set* rs = ctx->scratch_set;
	const MscLayout& L = set->L;
	if ((r = ensure(ctx, ctx->floor_sum, 8))) return r;
	if (mean_out && (r = ensure(ctx, ctx->mean, L.padded_bins * sizeof(double)))) return r;
	if (member_slots) {
		if ((r = ensure(ctx, ctx->slots, m * sizeof(uint32_t)))) return r;
		HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, member_slots, m * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	}
	const uint32_t* d_slots = member_slots ? (const uint32_t*)ctx->slots.p : nullptr;
	HIP_TRY(ctx, msc_launch_colsum(ctx->stream, L, set->dtype, set->bins, d_slots, (uint32_t)m, rs->bins, mean_out ? (double*)ctx->mean.p : nullptr,
	                               (uint64_t*)ctx->floor_sum.p, nullptr));
	HIP_TRY(ctx, hipMemsetAsync(rs->scalars, 0, sizeof(MscSlotScalars), ctx->stream));
	if ((r = ensure(ctx, ctx->tile_scratch, (size_t)L.S * 2 * sizeof(uint64_t)))) return r;
	HIP_TRY(ctx, msc_launch_finalize(ctx->stream, rs->bins, rs->scalars, L, set->dtype, 0, 1, false, (uint64_t*)ctx->tile_scratch.p));      // one wave per tile
	if ((r = refresh_bounds(ctx, rs, 0, 1))) return r;
	// members vs the rounded mean through the streaming kernel (only the |p - r| reduction is used)
	ScoreRequest rq;
	rq.cands = set; rq.cand_slots = member_slots; rq.m = m; rq.qset = rs; rq.q_slot = 0; rq.only_tiles = true;
	rq.slots_uploaded = member_slots != nullptr;          // (copied above, on this stream)
	if ((r = run_score(ctx, rq))) return r;
	if ((r = ensure(ctx, ctx->reduce_out, sizeof(MscReduceOut)))) return r;
	if ((r = ensure(ctx, ctx->raw, m * sizeof(double)))) return r;
	HIP_TRY(ctx, msc_launch_distance_d(ctx->stream, (const MscPartial*)ctx->partials.p, L.S, (uint32_t)m, set->scalars, set->scalar_stride, d_slots,
	                                   rs->scalars, (const uint64_t*)ctx->floor_sum.p, (double*)ctx->raw.p, (MscReduceOut*)ctx->reduce_out.p));
	MscReduceOut ro;
	HIP_TRY(ctx, hipMemcpyAsync(&ro, ctx->reduce_out.p, sizeof ro, hipMemcpyDeviceToHost, ctx->stream));
	if (dist_out) HIP_TRY(ctx, hipMemcpyAsync(dist_out, ctx->raw.p, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	if (mean_out) {
		if ((r = ensure(ctx, ctx->nat, L.padded_bins * sizeof(double)))) return r;
		MscLayout L64 = L;       // move 8-byte elements through the same bin permutation
		HIP_TRY(ctx, msc_launch_permute(ctx->stream, ctx->mean.p, ctx->nat.p, L64, 64, false));
		HIP_TRY(ctx, hipMemcpyAsync(mean_out, ctx->nat.p, L.nbins * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	float t = 0;
	if (ctx->timing && hipEventElapsedTime(&t, ctx->ev_tiles0, ctx->ev_tiles1) == hipSuccess) { ctx->tiles_ms_accum = t; ctx->tiles_launches = 1; ctx->have_timing = true; }
	*nearest_pos = ro.best_pos;
	return MSC_OK;
}

