// msc_api_multi.hip -- msc_score_multi, the Q x M all-pairs call (fastcar's loop, fastcar/FC_Runner.cpp:426-471): the mirrors of a dense set it
// reads (digest, presence bits, ranks), the routes, the block pipe over three streams. Split from msc_api.hip in r05.
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

// The digest mirror of a dense 32-bit set (pair_digest.hip): allocated on first use, refreshed for the slots written since.
// Returns MSC_OK with set->digest == nullptr when the mirror cannot be had (no memory): the caller then streams the raw bins.
static int ensure_digest(msc_ctx* ctx, const msc_hist_set* set) {
	if (set->sparse || !msc_digest_supported(set->L) || set->digest_unavailable) return MSC_OK;
	if (!set->digest) {
		void* p = nullptr;
		if (hipMalloc(&p, msc_digest_slot_bytes(set->L) * set->capacity) != hipSuccess) {
			(void)hipGetLastError();
			set->digest_unavailable = true;
			return MSC_OK;
		}
		set->digest = (uint8_t*)p;
		set->dg_lo = 0;
		set->dg_hi = set->capacity;
	}
	if (set->dg_lo < set->dg_hi) {
		HIP_TRY(ctx, msc_launch_digest_build(ctx->stream, set->L, set->bins, set->scalars, set->digest, set->dg_lo, set->dg_hi - set->dg_lo));
		set->dg_lo = set->dg_hi = 0;
	}
	return MSC_OK;
}

// The presence-bit mirror of a dense set and its lists of large bins (msc_pair_gemm.hip): the operands of the int8 product that takes the
// Q x M pass. MSC_OK with set->kb == nullptr when it cannot be had (no memory): the older routes then run.
static int ensure_kb(msc_ctx* ctx, const msc_hist_set* set) {
	if (set->sparse || set->kb_unavailable || set->dtype == 64) return MSC_OK;
	auto give_up = [&] {
		(void)hipGetLastError();
		if (set->kb) (void)hipFree(set->kb);
		if (set->mb) (void)hipFree(set->mb);
		if (set->mb_n) (void)hipFree(set->mb_n);
		set->kb = nullptr; set->mb = nullptr; set->mb_n = nullptr;
		set->kb_unavailable = true;
		return MSC_OK;
	};
	if (!set->kb) {
		void *p = nullptr, *pm = nullptr, *pn = nullptr;
		set->mb_pitch = 16;
		if (hipMalloc(&p, msc_kb_bytes(set->L, set->capacity)) != hipSuccess) return give_up();
		set->kb = (uint8_t*)p;
		if (hipMalloc(&pm, (size_t)set->capacity * set->mb_pitch * 8) != hipSuccess) return give_up();
		set->mb = pm;
		if (hipMalloc(&pn, (size_t)set->capacity * 4) != hipSuccess) return give_up();
		set->mb_n = (uint32_t*)pn;
		HIP_TRY(ctx, hipMemsetAsync(set->mb_n, 0, (size_t)set->capacity * 4, ctx->stream));
		set->mb_n_host.assign(set->capacity, 0);
		set->kb_lo = 0;
		set->kb_hi = set->capacity;
	}
	while (set->kb_lo < set->kb_hi) {
		// runs of slots that hold a histogram; the build reports a zero count (sticky: the pass's identities take count - 1 of every
		// bin) and the longest list of large bins it met: past the pitch, the lists are laid out again and every written slot rebuilt
		int r;
		if ((r = ensure(ctx, ctx->rk_bad, 2 * sizeof(int32_t)))) return r;
		HIP_TRY(ctx, hipMemsetAsync(ctx->rk_bad.p, 0, 2 * sizeof(int32_t), ctx->stream));
		const uint64_t lo = set->kb_lo, hi = std::min<uint64_t>(set->kb_hi, set->written.size());
		for (uint64_t i = lo; i < hi;) {
			if (!set->written[i]) { i++; continue; }
			uint64_t j = i;
			while (j < hi && set->written[j]) j++;
			HIP_TRY(ctx, msc_launch_kb_build(ctx->stream, set->L, set->dtype, set->bins, set->kb, i, j - i, set->mb, set->mb_n, set->mb_pitch, (int32_t*)ctx->rk_bad.p));
			i = j;
		}
		int32_t flags[2] = {0, 0};
		HIP_TRY(ctx, hipMemcpyAsync(flags, ctx->rk_bad.p, sizeof flags, hipMemcpyDeviceToHost, ctx->stream));
		if (hi > lo) HIP_TRY(ctx, hipMemcpyAsync(set->mb_n_host.data() + lo, set->mb_n + lo, (hi - lo) * 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (flags[0]) set->kb_has_zero = true;
		set->kb_lo = set->kb_hi = 0;
		if ((uint32_t)flags[1] > set->mb_pitch) {
			const uint32_t pitch = ((uint32_t)flags[1] + 15) / 16 * 16;
			void* pm = nullptr;
			(void)hipFree(set->mb);
			set->mb = nullptr;
			if (hipMalloc(&pm, (size_t)set->capacity * pitch * 8) != hipSuccess) return give_up();
			set->mb = pm;
			set->mb_pitch = pitch;
			set->kb_lo = 0;
			set->kb_hi = set->capacity;
		}
	}
	return MSC_OK;
}

// The ranks mirror of a dense set (msc_emd_ranks.hip), from its bins. MSC_OK with set->ranks == nullptr when it cannot be had (no
// memory, or a slot holds a zero count): the digest kernel then keeps the prefixes.
static int ensure_ranks(msc_ctx* ctx, const msc_hist_set* set) {
	if (set->sparse || set->dtype == 64 || !msc_digest_supported(set->L) || set->ranks_unavailable || set->max_sum < set->L.nbins) return MSC_OK;
	const uint64_t pitch = msc_ranks_pitch(set->max_sum - set->L.nbins);
	if (set->ranks && pitch > set->rk_pitch) {          // a longer list than any before: lay the mirror out again
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		(void)hipFree(set->ranks);
		set->ranks = nullptr;
		if (set->ranks16) { (void)hipFree(set->ranks16); set->ranks16 = nullptr; }
		set->rk16_off = false;
	}
	if (!set->ranks) {
		void *p = nullptr, *pn = set->rk_n;
		if (hipMalloc(&p, pitch * 4 * set->capacity) != hipSuccess || (!pn && hipMalloc(&pn, 4 * set->capacity) != hipSuccess)) {
			(void)hipGetLastError();
			if (p) (void)hipFree(p);
			set->ranks_unavailable = true;
			return MSC_OK;
		}
		set->ranks = (uint32_t*)p;
		set->rk_n = (uint32_t*)pn;
		set->rk_pitch = pitch;
		set->rk_lo = 0;
		set->rk_hi = set->capacity;
	}
	static const bool no_rk16 = getenv("MSC_NO_RANKS16") != nullptr;
	if (!set->ranks16 && !set->rk16_off && !no_rk16 && set->rk_pitch % 1024 == 0) {          // the 16-bit form beside it (k_emd_ranks16)
		void* p16 = nullptr;
		if (hipMalloc(&p16, set->rk_pitch * 2 * set->capacity) != hipSuccess) { (void)hipGetLastError(); set->rk16_off = true; }
		else { set->ranks16 = (uint16_t*)p16; set->rk_lo = 0; set->rk_hi = set->capacity; }
	}
	if (set->rk_lo < set->rk_hi) {
		int r;
		if ((r = ensure(ctx, ctx->rk_bad, 2 * sizeof(int32_t)))) return r;
		HIP_TRY(ctx, hipMemsetAsync(ctx->rk_bad.p, 0, 2 * sizeof(int32_t), ctx->stream));
		// runs of slots that hold a histogram (an unwritten slot's digest is whatever the allocation held)
		const uint64_t hi = std::min<uint64_t>(set->rk_hi, set->written.size());
		for (uint64_t i = set->rk_lo; i < hi;) {
			if (!set->written[i]) { i++; continue; }
			uint64_t j = i;
			while (j < hi && set->written[j]) j++;
			HIP_TRY(ctx, msc_launch_ranks_build(ctx->stream, set->L, set->dtype, set->bins, set->scalars, set->ranks, set->rk_n, set->rk_pitch, i, j - i, (int32_t*)ctx->rk_bad.p));
			if (set->ranks16) HIP_TRY(ctx, msc_launch_ranks16_build(ctx->stream, set->L.nbins, set->ranks, set->ranks16, set->rk_pitch, i, j - i, (int32_t*)ctx->rk_bad.p + 1));
			i = j;
		}
		int32_t bad[2] = {0, 0};
		HIP_TRY(ctx, hipMemcpyAsync(bad, ctx->rk_bad.p, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		set->rk_lo = set->rk_hi = 0;
		if (bad[0]) {
			(void)hipFree(set->ranks);
			set->ranks = nullptr;
			set->ranks_unavailable = true;
		}
		if ((bad[0] || bad[1]) && set->ranks16) {          // a reduced rank that does not fit 16 bits: this set keeps the 32-bit walk
			(void)hipFree(set->ranks16);
			set->ranks16 = nullptr;
			set->rk16_off = true;
		}
	}
	return MSC_OK;
}

// Whether the pass on the matrix cores (msc_pair_gemm.hip) can take a Q x M call over these sets -- host-side bounds only: dense 8/16/32-bit
// sets of the narrow range whose histograms are whole 4 KiB tiles, P1 / P2 within int32 and, when
// the earth mover's distance is wanted, lists short enough for the ranks mirror (msc_emd_ranks.hip).
static bool kb_route_fits(const msc_hist_set* cands, const msc_hist_set* qset, bool need_emd) {
	static const bool off = getenv("MSC_MULTI_NO_GEMM") != nullptr;
	static const bool no_ranks = getenv("MSC_MULTI_NO_RANKS") != nullptr;
	const MscLayout& L = cands->L;
	if (off || cands->sparse || qset->sparse || cands->dtype == 64 || L.nbins != L.padded_bins || !msc_digest_supported(L) || needs_wide(cands, qset)) return false;
	const uint64_t ms_ = std::max(cands->max_sum, qset->max_sum);
	if (ms_ < L.nbins || ms_ - L.nbins >= (1ull << 24)) return false;          // (P1 <= the k-mers of either sequence is summed in f32: exact below 2^24; the corrections stay within int32)
	if (need_emd && (no_ranks || L.nbins > (1ull << 20) || (ms_ - L.nbins) * 4 > L.nbins)) return false;
	return true;
}

static int score_multi_impl(msc_ctx* ctx, const msc_model* model, const msc_hist_set* cands, const uint32_t* cand_slots, uint64_t m,
                            const msc_hist_set* qset, const uint32_t* q_slots, uint64_t n_q, int order, double* sum_out, double* csum_out,
                            uint8_t* close_out, uint64_t feat_mask, double* raw_out);

// the epilogue's error word (the stream is idle): the first failing pair's status
static int read_error_word(msc_ctx* ctx) {
	int32_t first_err = 0;
	HIP_TRY(ctx, hipMemcpy(&first_err, ctx->err_word.p, sizeof first_err, hipMemcpyDeviceToHost));
	if (first_err == MSC_ERR_ZERO_LENGTH) return fail(ctx, first_err, "length_difference: a point has length 0 (the reference throws 123, predict/Feature.cpp:878-886)");
	if (first_err == MSC_ERR_NAN) return fail(ctx, first_err, "normalisation produced NaN (the reference throws, predict/Feature.cpp:143-146)");
	if (first_err < 0) return fail(ctx, first_err, "feature evaluation failed with status %d", first_err);
	return MSC_OK;
}

// timing events of queued blocks (two per launch of the streaming kernel), kept for the life of the context
static int pool_event(msc_ctx* ctx, hipEvent_t* e) {
	if (ctx->ev_used == ctx->ev_pool.size()) {
		hipEvent_t n = nullptr;
		HIP_TRY(ctx, hipEventCreate(&n));
		ctx->ev_pool.push_back(n);
	}
	*e = ctx->ev_pool[ctx->ev_used++];
	return MSC_OK;
}

// the queued blocks of msc_score_multi: wait for them, add up their kernel times, read the error word they share
static int flush_deferred(msc_ctx* ctx) {
	if (ctx->defer != 2) return MSC_OK;
	ctx->defer = 1;
	hipError_t e = hipStreamSynchronize(ctx->stream);
	if (ctx->tail_used) {          // (the epilogues of the queued blocks run on the second stream)
		const hipError_t e2 = hipStreamSynchronize(ctx->tail_stream);
		if (e == hipSuccess) e = e2;
		ctx->tail_used = false;
		ctx->tail_busy[0] = ctx->tail_busy[1] = false;
		ctx->product_busy[0] = ctx->product_busy[1] = false;          // (every product waited for its queries' side: the prep stream is idle too)
	}
	for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
		float t = 0;
		if (e == hipSuccess && hipEventElapsedTime(&t, ctx->ev_pool[i], ctx->ev_pool[i + 1]) == hipSuccess) { ctx->defer_ms += t; ctx->have_timing = true; }
	}
	ctx->ev_used = 0;
	if (e != hipSuccess) return fail(ctx, MSC_ERR_HIP, "queued blocks failed: %s", hipGetErrorString(e));
	return read_error_word(ctx);
}

extern "C" int msc_score_multi(msc_ctx* ctx, const msc_model* model, const msc_hist_set* cands, const uint32_t* cand_slots, uint64_t m,
                               const msc_hist_set* qset, const uint32_t* q_slots, uint64_t n_q, int order, double* sum_out, double* csum_out,
                               uint8_t* close_out, uint64_t feat_mask, double* raw_out) {
	const int r = score_multi_impl(ctx, model, cands, cand_slots, m, qset, q_slots, n_q, order, sum_out, csum_out, close_out, feat_mask, raw_out);
	if (ctx && ctx->copy_pending) {          // the flag copies of the last blocks (issued beside the kernels that followed them)
		const hipError_t e = hipStreamSynchronize(ctx->copy_stream);
		ctx->copy_pending = false;
		ctx->close_pp_busy[0] = ctx->close_pp_busy[1] = false;
		if (e != hipSuccess && r == MSC_OK) return fail(ctx, MSC_ERR_HIP, "copy of the close flags failed: %s", hipGetErrorString(e));
	}
	return r;
}

static int score_multi_impl(msc_ctx* ctx, const msc_model* model, const msc_hist_set* cands, const uint32_t* cand_slots, uint64_t m,
                            const msc_hist_set* qset, const uint32_t* q_slots, uint64_t n_q, int order, double* sum_out, double* csum_out,
                            uint8_t* close_out, uint64_t feat_mask, double* raw_out) {
	if (!ctx || !cands || !qset || !q_slots) return MSC_ERR_INVALID_ARG;
	if (model && model->ctx != ctx) return MSC_ERR_INVALID_ARG;
	if (raw_out && (feat_mask == 0 || (feat_mask & ~kSupportedFeats))) return fail(ctx, MSC_ERR_UNSUPPORTED, "feat_mask holds statistics outside the GPU path");
	if (!raw_out) feat_mask = 0;
	if (n_q == 0 || m == 0) return MSC_OK;
	for (uint64_t i = 0; i < n_q; i++) if (q_slots[i] >= qset->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "query slot out of range");
	int r = validate_pair(ctx, cands, qset, q_slots[0], cand_slots, m);
	if (r) return r;
	const MscLayout& L = cands->L;
	const int nf = __builtin_popcountll(feat_mask);
	uint64_t want = feat_mask;
	if (model) for (int i = 0; i < model->h.n_singles; i++) want |= model->h.single_flag[i];
	const bool need_emd = (want & MSC_FEAT_EMD) != 0;           // Feature::compute evaluates only the model's singles too
	// The pass on the matrix cores (msc_pair_gemm.hip) serves blocks of up to 128 queries per pass over the candidates' bits; the older routes 64
	bool kb_fit = !ctx->no_kb_now && n_q >= 2 && kb_route_fits(cands, qset, need_emd);
	if (kb_fit) {
		if ((r = ensure_kb(ctx, cands)) || (r = ensure_kb(ctx, qset))) return r;
		kb_fit = cands->kb && qset->kb && !cands->kb_has_zero && !qset->kb_has_zero;
	}
	// close candidates per query, kept on the device for msc_last_close_counts (a caller that only needs the counts of a block of the
	// pairwise matrix does not have to add up n_q x m flags on the host)
	const bool top_level = !ctx->in_score_multi;
	if (top_level && close_out) {
		if ((r = ensure(ctx, ctx->close_counts, n_q * sizeof(uint64_t)))) return r;
		HIP_TRY(ctx, hipMemsetAsync(ctx->close_counts.p, 0, n_q * sizeof(uint64_t), ctx->stream));
		ctx->close_counts_n = n_q;
		ctx->close_counts_base = 0;
	} else if (top_level) { ctx->close_counts_n = 0; ctx->close_counts_base = 0; }
	const uint64_t blk = kb_fit ? 128 : 64;
	if (n_q > blk) {
		// blocks of queries: the unit of the pass on the matrix cores (a 128-row operand) and of the digest kernel (four groups of 16);
		// msc_last_kernel_ms / _launches then cover the whole call
		float ms = 0.f;
		int launches = 0;
		const bool was_in = ctx->in_score_multi;
		const uint64_t base0 = ctx->close_counts_base;
		ctx->in_score_multi = true;
		// the blocks of the matrix-core pass are queued back to back (score_multi_impl below, `deferred`): the whole call's query slots
		// go up once, here
		static const bool no_defer = getenv("MSC_GEMM_NO_QUEUE") != nullptr;
		const bool defer = kb_fit && top_level && !no_defer;
		if (defer) {
			if ((r = ensure(ctx, ctx->qslots_all, n_q * sizeof(uint32_t)))) { ctx->in_score_multi = was_in; return r; }
			HIP_TRY(ctx, hipMemcpyAsync(ctx->qslots_all.p, q_slots, n_q * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipEventRecord(ctx->ev_call, ctx->stream));
			ctx->defer = 1;
			ctx->defer_cands_up = false;
			ctx->defer_ms = 0.f;
			ctx->ev_used = 0;
		}
		for (uint64_t b = 0; b < n_q; b += blk) {
			const uint64_t nb = std::min<uint64_t>(blk, n_q - b);
			ctx->close_counts_base = base0 + b;
			ctx->defer_q_off = b;
			if ((r = score_multi_impl(ctx, model, cands, cand_slots, m, qset, q_slots + b, nb, order, sum_out ? sum_out + b * m : nullptr, csum_out ? csum_out + b * m : nullptr,
			                         close_out ? close_out + b * m : nullptr, feat_mask, raw_out ? raw_out + b * m * nf : nullptr))) {
				if (defer) { (void)flush_deferred(ctx); ctx->defer = 0; }          // (nothing of this call may still be running when it returns)
				ctx->in_score_multi = was_in;
				ctx->close_counts_base = base0;
				return r;
			}
			ms += ctx->tiles_ms_accum;
			launches += ctx->tiles_launches;
		}
		if (defer) {
			r = flush_deferred(ctx);
			ctx->defer = 0;
			ms += ctx->defer_ms;
		}
		ctx->in_score_multi = was_in;
		ctx->close_counts_base = base0;
		if (r) return r;
		ctx->tiles_ms_accum = ms;
		ctx->tiles_launches = launches;
		return MSC_OK;
	}
	// divergence statistics in the Q x M pass: the integer reductions come from the streaming kernel below, the two FP64 sums from
	// one merge pass per query over the sparse mirrors, queued behind it -- the same kernel, hence the same values,
	// as a 1 x M pass per query
	const bool want_div = (want & MSC_FEAT_DIV) != 0;
	const msc_hist_set *c_sp = nullptr, *q_sp = nullptr;
	// sim_mm / rre_k_r likewise: one group pass per query behind the streaming kernel, over the mirrors' lists or (histograms under
	// 64 KiB) the dense slots -- the kernels and records of the 1 x M pass
	const bool want_grp = (want & MSC_FEAT_GROUPS) != 0;
	if ((want_div || want_grp) && !cands->sparse && n_q > 1 && L.nbins == L.padded_bins && !needs_wide(cands, qset)) {
		if ((r = ensure_sparse_mirror(ctx, cands, &c_sp)) || (r = ensure_sparse_mirror(ctx, qset, &q_sp))) return r;
		if (!c_sp || !q_sp) c_sp = q_sp = nullptr;
	}
	const bool grp_dense = want_grp && !c_sp;
	const bool simple = (!grp_dense || std::max(cands->max_count, qset->max_count) <= 0xffffffffull) && (!want_div || c_sp) && L.nbins == L.padded_bins && n_q > 1 &&
	                    !needs_wide(cands, qset) && !cands->sparse;
	// Sparse sets: one merge-path pass per query, but queued back to back into one [n_q][m] record array with ONE epilogue and one
	// copy back -- no host round trip between the passes.
	static const bool no_sp_multi = getenv("MSC_SPARSE_NO_MULTI") != nullptr;
	const bool sparse_multi = cands->sparse && qset->sparse && !no_sp_multi && !(want & (MSC_FEAT_DIV | MSC_FEAT_GROUPS)) && n_q > 1 && !needs_wide(cands, qset) &&
	                          std::max(cands->max_count, qset->max_count) < 65536 && n_q * m <= 0x7fffffffull &&
	                          n_q * m * sizeof(MscPartial) <= (4096ull << 20) && !getenv("MSC_SPARSE_NO_MP") && !getenv("MSC_SPARSE_LDS");
	if (sparse_multi) {
		HIP_TRY(ctx, hipSetDevice(ctx->device));
		ctx->tiles_ms_accum = 0.f;
		ctx->tiles_launches = 0;
		ctx->have_timing = false;
		ctx->last_kernel = "k_pair_sparse_mp";
		ctx->last_query_tile = 1;
		ctx->last_partial_stride = 1;
		if ((r = ensure(ctx, ctx->err_word, sizeof(int32_t)))) return r;
		if ((r = ensure(ctx, ctx->qslots, n_q * sizeof(uint32_t)))) return r;
		if ((r = ensure(ctx, ctx->partials, n_q * m * sizeof(MscPartial)))) return r;
		if (sum_out && (r = ensure(ctx, ctx->soa_sum, n_q * m * sizeof(double)))) return r;
		if (csum_out && (r = ensure(ctx, ctx->soa_csum, n_q * m * sizeof(double)))) return r;
		if (close_out && (r = ensure(ctx, ctx->soa_close, n_q * m))) return r;
		if (raw_out && (r = ensure(ctx, ctx->raw, n_q * m * nf * sizeof(double)))) return r;
		HIP_TRY(ctx, hipMemcpyAsync(ctx->qslots.p, q_slots, n_q * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemsetAsync(ctx->err_word.p, 0, sizeof(int32_t), ctx->stream));
		if (cand_slots) {
			if ((r = ensure(ctx, ctx->slots, m * sizeof(uint32_t)))) return r;
			HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, cand_slots, m * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
		}
		const uint32_t* d_slots = cand_slots ? (const uint32_t*)ctx->slots.p : nullptr;
		// up to k = 9: the passes over the candidates' rank lists (msc_ranks_pass.hip), as in run_score
		const uint64_t q_kmers = qset->max_sum >= L.nbins ? qset->max_sum - L.nbins : ~0ull;
		bool rank_pass = false;
		if (getenv("MSC_NO_RANKS_1XM") == nullptr && q_kmers <= msc_ranks_pass_query_cap() && msc_ranks_pass_lds(L.nbins, q_kmers) != 0) {
			int e = MSC_OK;
			rank_pass = rank_lists_ready(ctx, cands, &e);
			if (e) return e;
			if (rank_pass && !ctx->rk_guard) {
				HIP_TRY(ctx, hipHostMalloc((void**)&ctx->rk_guard, 64, hipHostMallocDefault));
				*ctx->rk_guard = 0;
			}
			if (rank_pass && msc_ranks_pass_query_scratch(q_kmers) && (r = ensure(ctx, ctx->rk_q, msc_ranks_pass_query_scratch(q_kmers) * sizeof(uint32_t)))) return r;
			if (rank_pass) ctx->last_kernel = "k_pair_ranks_1xm";
		}
		if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_all0, ctx->stream));
		if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_tiles0, ctx->stream));
		for (uint64_t q = 0; q < n_q && rank_pass; q++)
			HIP_TRY(ctx, msc_launch_pair_ranks_1xm(ctx->stream, cands->rkl, cands->rkl_off, cands->rkl_n, cands->scalars, cands->scalar_stride, d_slots, 0, (uint32_t)m, qset->ent, qset->cum,
			                                       qset->hdr + q_slots[q], L.nbins, 0, 0, ~0ull, (MscPartial*)ctx->partials.p + q * m, ctx->num_cus, q_kmers, ctx->rk_guard, (uint32_t*)ctx->rk_q.p));
		for (uint64_t q = 0; q < n_q && !rank_pass; q++)
			HIP_TRY(ctx, msc_launch_pair_sparse_mp(ctx->stream, cands->ent, cands->cum, cands->hdr, cands->scalars, cands->scalar_stride, d_slots, (uint32_t)m, qset->ent,
			                                       qset->cum, qset->hdr + q_slots[q], qset->scalars + (uint64_t)q_slots[q] * qset->scalar_stride, L.nbins, 0, 0, ~0ull,
			                                       (MscPartial*)ctx->partials.p + q * m, nullptr, nullptr, order, ctx->num_cus,
			                                       (uint32_t)std::min<uint64_t>(0x7fffffffull, (uint64_t)qset->hdr_host[q_slots[q]].nnz + cands->max_nnz), 1,
			                                       qset->hdr_host[q_slots[q]].nnz, cands->max_nnz));
		if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_tiles1, ctx->stream));
		MscEpilogueArgs ea;
		memset(&ea, 0, sizeof ea);
		ea.partials = (const MscPartial*)ctx->partials.p;
		ea.S = 1;
		ea.m = (uint32_t)(n_q * m);
		ea.cand_scalars = cands->scalars;
		ea.cand_scalar_stride = cands->scalar_stride;
		ea.cand_slots = d_slots;
		ea.n_queries = (uint32_t)n_q;
		ea.m_per_query = (uint32_t)m;
		ea.q_slots = (const uint32_t*)ctx->qslots.p;
		ea.qset_scalars = qset->scalars;
		ea.q_scalar_stride = qset->scalar_stride;
		ea.q_scalars = qset->scalars + (uint64_t)q_slots[0] * qset->scalar_stride;
		ea.nbins = L.nbins;
		ea.dtype = cands->dtype;
		ea.order = order;
		ea.feat_mask = feat_mask;
		ea.sparse_base = L.nbins;
		ea.raw_out = raw_out ? (double*)ctx->raw.p : nullptr;
		ea.model = model ? model->d : nullptr;
		ea.sum_soa = sum_out ? (double*)ctx->soa_sum.p : nullptr;
		ea.csum_soa = csum_out ? (double*)ctx->soa_csum.p : nullptr;
		ea.close_soa = close_out ? (uint8_t*)ctx->soa_close.p : nullptr;
		ea.error_word = (int32_t*)ctx->err_word.p;
		HIP_TRY(ctx, msc_launch_epilogue(ctx->stream, ea));
		if (close_out && ctx->close_counts_n) HIP_TRY(ctx, msc_launch_close_counts(ctx->stream, (const uint8_t*)ctx->soa_close.p, (uint32_t)n_q, (uint32_t)m, (uint64_t*)ctx->close_counts.p + ctx->close_counts_base));
		if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_all1, ctx->stream));
		if (sum_out) HIP_TRY(ctx, hipMemcpyAsync(sum_out, ctx->soa_sum.p, n_q * m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
		if (csum_out) HIP_TRY(ctx, hipMemcpyAsync(csum_out, ctx->soa_csum.p, n_q * m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
		if (close_out) HIP_TRY(ctx, hipMemcpyAsync(close_out, ctx->soa_close.p, n_q * m, hipMemcpyDeviceToHost, ctx->stream));
		if (raw_out) HIP_TRY(ctx, hipMemcpyAsync(raw_out, ctx->raw.p, n_q * m * nf * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
		int32_t first_err = 0;
		HIP_TRY(ctx, hipMemcpyAsync(&first_err, ctx->err_word.p, sizeof first_err, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		float t = 0;
		if (ctx->timing && hipEventElapsedTime(&t, ctx->ev_tiles0, ctx->ev_tiles1) == hipSuccess) { ctx->tiles_ms_accum = t; ctx->tiles_launches = (int)n_q; ctx->have_timing = true; }
		if (rank_pass && *ctx->rk_guard) {
			*ctx->rk_guard = 0;
			return fail(ctx, MSC_ERR_HIP, "rank pass: a query's list is longer than its set's bound (max_sum not maintained by a writer of that set)");
		}
		if (first_err == MSC_ERR_ZERO_LENGTH) return fail(ctx, first_err, "length_difference: a point has length 0 (the reference throws 123, predict/Feature.cpp:878-886)");
		if (first_err == MSC_ERR_NAN) return fail(ctx, first_err, "normalisation produced NaN (the reference throws, predict/Feature.cpp:143-146)");
		if (first_err < 0) return fail(ctx, first_err, "feature evaluation failed with status %d", first_err);
		return MSC_OK;
	}
	if (!simple) {
		ctx->close_counts_n = 0;          // (no counts from this route: msc_last_close_counts says so)
		// divergence statistics / padded tiny histograms: one streaming pass per query through the single-query kernel
		float ms = 0.f;
		int launches = 0;
		for (uint64_t q = 0; q < n_q; q++) {
			ScoreRequest rq;
			rq.model = model; rq.cands = cands; rq.cand_slots = cand_slots; rq.m = m; rq.qset = qset; rq.q_slot = q_slots[q]; rq.order = order;
			rq.feat_mask = feat_mask; rq.raw_out = raw_out ? raw_out + q * m * nf : nullptr;
			rq.sum_out = sum_out ? sum_out + q * m : nullptr; rq.csum_out = csum_out ? csum_out + q * m : nullptr;
			rq.flags_out = close_out ? close_out + q * m : nullptr;
			if ((r = run_score(ctx, rq))) return r;
			ms += ctx->tiles_ms_accum;
			launches += ctx->tiles_launches;
		}
		ctx->tiles_ms_accum = ms;          // msc_last_kernel_ms / _launches cover the whole call
		ctx->tiles_launches = launches;
		return MSC_OK;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	ctx->tiles_ms_accum = 0.f;
	ctx->tiles_launches = 0;
	ctx->have_timing = false;
	int tq = n_q >= 4 ? 4 : 2;                     // TQ = 4 keeps the 32-bit register kernel HBM-bound
	if (const char* e = getenv("MSC_MULTI_TQ")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8) tq = v; }
	if (tq > (int)n_q && n_q >= 2) tq = n_q >= 4 ? 4 : 2;
	// wave totals of the per-lane 32-bit partial sums fit 32 bits when 64*R*max^2 and 64*R*max|prefix difference| do
	const uint64_t mc_ = std::max(cands->max_count, qset->max_count), ms_ = std::max(cands->max_sum, qset->max_sum);
	const bool compact = 64ull * L.R * mc_ * mc_ < (1ull << 32) && 64ull * L.R * ms_ < (1ull << 32);
	// every prefix of excess counts (count - 1) is at most the histogram's k-mer total = sum - 4^k: 16-bit prefix form when that fits
	const bool excess16 = ms_ >= L.nbins && ms_ - L.nbins < 65536;
	static const bool no_digest = getenv("MSC_MULTI_NO_DIGEST") != nullptr;
	static const bool no_ranks = getenv("MSC_MULTI_NO_RANKS") != nullptr;
	const bool tuned_by_hand = getenv("MSC_MULTI_TQ") || getenv("MSC_DIGEST_SLOTS");          // A/B switches of the older kernels: keep to them
	// The earth mover's distance from sorted k-mer ranks (msc_emd_ranks.hip) -- O(k-mers) per pair instead of O(bins): while the
	// longest list is a quarter of the bins or less, for up to 256 queries and 2^20 bins (32-bit wave sums)
	const bool ranks_fit = !no_ranks && !tuned_by_hand && n_q <= 256 && L.nbins <= (1ull << 20) && ms_ >= L.nbins && (ms_ - L.nbins) * 4 <= L.nbins && msc_digest_supported(L);
	// EVERYTHING on the matrix cores (msc_pair_gemm.hip): one int8 product per tile of bins over the presence-bit mirrors + corrections from the
	// lists of large bins -- exact for any counts of the narrow range; one read of a candidate byte per 256 queries, no partial records.
	// The queries' large bins become this block's hot list: its size is known here (the lists' lengths are mirrored on the host), and
	// a block whose list would average more than 64 entries per 128-bin step (long sequences in few bins: the walk over the list would
	// then take several times the step's product) is left to the older routes.
	bool manh_gemm = false, emd_ranks = false;
	uint64_t n_hot = 0;
	if (kb_fit && simple && !tuned_by_hand && !no_digest) {
		for (uint64_t q = 0; q < n_q; q++) n_hot += std::min(qset->mb_n_host[q_slots[q]], qset->mb_pitch);
		manh_gemm = n_hot <= 64 * (L.nbins / 128);
		if (manh_gemm && need_emd) {
			if ((r = ensure_ranks(ctx, cands)) || (r = ensure_ranks(ctx, qset))) return r;
			manh_gemm = emd_ranks = cands->ranks && qset->ranks;
		}
	}
	// (blocks of the matrix-core pass queued without a host wait between them: any other route first waits for them and reads their error word)
	if (!manh_gemm && ctx->defer == 2 && (r = flush_deferred(ctx))) return r;
	if (!manh_gemm && n_q > 64) {          // (a block of up to 256 was cut for the matrix cores: the older routes take it in blocks of 64)
		ctx->no_kb_now = true;
		r = score_multi_impl(ctx, model, cands, cand_slots, m, qset, q_slots, n_q, order, sum_out, csum_out, close_out, feat_mask, raw_out);
		ctx->no_kb_now = false;
		return r;
	}
	// A block of a larger call on the matrix cores is QUEUED: its query slots are part of the list the call sent up once, the error word is
	// cleared by the first block and read after the last, and nothing here waits for the stream -- the scratch buffers the next block
	// overwrites are ordered behind this block's kernels by the stream itself (a buffer that has to grow goes through hipFree, which waits).
	const bool deferred = manh_gemm && ctx->defer != 0;
	const uint32_t* dq_slots = nullptr;
	if ((r = ensure(ctx, ctx->err_word, sizeof(int32_t)))) return r;
	if (deferred) dq_slots = (const uint32_t*)ctx->qslots_all.p + ctx->defer_q_off;
	else {
		if ((r = ensure(ctx, ctx->qslots, n_q * sizeof(uint32_t)))) return r;
		HIP_TRY(ctx, hipMemcpyAsync(ctx->qslots.p, q_slots, n_q * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
		dq_slots = (const uint32_t*)ctx->qslots.p;
	}
	if (!deferred || ctx->defer == 1) HIP_TRY(ctx, hipMemsetAsync(ctx->err_word.p, 0, sizeof(int32_t), ctx->stream));
	if (cand_slots && !(deferred && ctx->defer_cands_up)) {
		if ((r = ensure(ctx, ctx->slots, m * sizeof(uint32_t)))) return r;
		HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, cand_slots, m * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
		if (deferred) ctx->defer_cands_up = true;
	}
	if (deferred) ctx->defer = 2;
	// Digest form (pair_digest.hip): sets whose counts and excess prefixes fit 16 bits, from four queries up. Sixteen (or 32)
	// queries share one HBM read of each candidate tile; the raw kernels below remain for everything else.
	// (one digest tile per lane-run of 16 bins: wave totals of 1024 * max^2 must fit 32 bits)
	// The mirror streams 4 bytes per bin: against 8/16-bit raw bins it pays once enough queries share each candidate read
	// (measured crossovers at k = 9: 7 queries for uint8_t, 5-6 for uint16_t, 4 for uint32_t)
	const uint64_t dg_min_q = cands->dtype == 8 ? 8 : cands->dtype == 16 ? 6 : 4;
	bool digest = !manh_gemm && !no_digest && excess16 && msc_digest_supported(L) && mc_ < 2048 && n_q >= dg_min_q && !getenv("MSC_MULTI_TQ");
	if (digest) {
		if ((r = ensure_digest(ctx, cands)) || (r = ensure_digest(ctx, qset))) return r;
		digest = cands->digest && qset->digest;
	}
	const bool gemm_dot = false;          // (r03's digest forms without their products took them from an int8 GEMM over a count mirror: the presence-bit route replaced both)
	// LDS-DMA ring form over the raw bins: 32/64-bit bins, compact totals, query groups of four or eight
	static const bool no_ring = getenv("MSC_MULTI_NO_RING") != nullptr;
	if (!digest && !manh_gemm && n_q >= 16 && !getenv("MSC_MULTI_TQ") && (cands->dtype == 32 || cands->dtype == 64)) tq = 8;      // measured best from 16 queries up
	const bool ring = !digest && !manh_gemm && !no_ring && compact && L.LPT == 4 && (cands->dtype == 32 || cands->dtype == 64) && (tq == 4 || tq == 8) && n_q >= 4;
	static const bool no_p16 = getenv("MSC_RING_NO_P16") != nullptr;
	const bool prefix16 = ring && !no_p16 && excess16;
	// partial records of one launch are capped at 4 GiB: equal candidate chunks
	const int tps = digest ? msc_digest_tiles_per_step(L, mc_) : 1;          // the digest kernel writes one record per step of tps tiles
	if (digest && need_emd && ranks_fit && tps == 2) {          // the digest kernel then runs its count-only form (two tiles per step)
		if ((r = ensure_ranks(ctx, cands)) || (r = ensure_ranks(ctx, qset))) return r;
		emd_ranks = cands->ranks && qset->ranks;
	}
	const bool digest_emd = need_emd && !emd_ranks;                          // the digest kernel streams and scores the prefix half
	const uint32_t n_rec = digest ? (uint32_t)(L.nbins / 1024) / tps : L.S;
	// manh is all that is left to the digest kernel: eight queries per wave (32 per candidate tile fetched), 4-byte records
	static const bool no_tq8 = getenv("MSC_DIGEST_NO_TQ8") != nullptr;
	const int dg_tq = digest && gemm_dot && !digest_emd && n_q > 16 && !no_tq8 ? 8 : 4;
	const uint64_t rec_bytes = digest && dg_tq == 8 ? 4 : digest && gemm_dot ? 8 : digest || ring ? 16 : sizeof(MscPartial);
	const uint64_t q_rows = digest ? (n_q + 4 * dg_tq - 1) / (4 * dg_tq) * (4 * dg_tq) : ring ? (n_q + tq - 1) / tq * tq : n_q;       // records cover the padded query count
	uint64_t chunk = (4096ull << 20) / ((uint64_t)n_rec * rec_bytes * q_rows);
	// (no records without the digest kernel: the product array of the GEMM, [slices][chunk][rows] int32, kept to 2 GiB)
	const uint32_t kb_qn = manh_gemm ? msc_pair_gemm_rows((uint32_t)n_q) : 0;
	if (manh_gemm) chunk = (2048ull << 20) / ((uint64_t)msc_pair_gemm_slices(L.nbins, (uint32_t)std::min<uint64_t>(m, 1u << 30), kb_qn, ctx->num_cus) * kb_qn * sizeof(int32_t));
	if (want_grp) chunk = std::min<uint64_t>(chunk, (1024ull << 20) / (n_q * 32 * sizeof(double)));      // [n_q][chunk][16][2] group records: 1 GiB
	chunk = std::min(std::max<uint64_t>(chunk, 256), m);
	chunk = (m + (m + chunk - 1) / chunk - 1) / ((m + chunk - 1) / chunk);
	if (!manh_gemm && (r = ensure(ctx, ctx->partials, q_rows * chunk * n_rec * rec_bytes))) return r;
	if (want_grp) {
		if ((r = ensure(ctx, ctx->grp_pairs, n_q * chunk * 32 * sizeof(double)))) return r;
		if ((r = ensure(ctx, ctx->grp_self, (chunk + n_q) * 16 * sizeof(double)))) return r;      // [candidates][16] then [queries][16]
	}
	SparseKernel spk = SPK_MP;
	uint32_t dvn = 1;        // {jd, js} records per pair (one stride for the whole block)
	if (want_div) {          // one kernel for the whole block: merge-path unless some query's lists are out of its range
		for (uint64_t q = 0; q < n_q; q++) if (pick_sparse_kernel(c_sp, q_sp, q_slots[q], mc_, false) != SPK_MP) spk = SPK_GENERIC;
		const uint32_t spn = sparse_records(spk);
		uint64_t q_nnz_max = 0;
		for (uint64_t q = 0; q < n_q; q++) q_nnz_max = std::max<uint64_t>(q_nnz_max, q_sp->hdr_host[q_slots[q]].nnz);
		dvn = div_records(spk, q_nnz_max + c_sp->max_nnz);
		if ((r = ensure(ctx, ctx->div_tables, chunk * 256 * 16))) return r;
		if ((r = ensure(ctx, ctx->div_partials, n_q * chunk * dvn * 16))) return r;
		if ((r = ensure(ctx, ctx->sp_partials, chunk * spn * sizeof(MscPartial)))) return r;
	}
	if (sum_out && (r = ensure(ctx, ctx->soa_sum, n_q * chunk * sizeof(double)))) return r;
	if (csum_out && (r = ensure(ctx, ctx->soa_csum, n_q * chunk * sizeof(double)))) return r;
	if (close_out && !manh_gemm && (r = ensure(ctx, ctx->soa_close, n_q * chunk))) return r;
	if (close_out && manh_gemm && ((r = ensure(ctx, ctx->close_pp[0], n_q * chunk)) || (r = ensure(ctx, ctx->close_pp[1], n_q * chunk)))) return r;
	if (raw_out && (r = ensure(ctx, ctx->raw, n_q * chunk * nf * sizeof(double)))) return r;
	const uint32_t gemm_slices = manh_gemm ? msc_pair_gemm_slices(L.nbins, (uint32_t)chunk, kb_qn, ctx->num_cus) : 0;
	uint32_t *hot_ptr = nullptr, *hot_cursor = nullptr, *hot_cnt = nullptr;
	// Queued blocks run in two stages on two streams (msc_objects.h): the product of block i on ctx->stream beside the rank walk of block i
	// and the epilogue of block i - 1 on tail_stream -- the product is bound by the matrix pipe, the other two by vector arithmetic and
	// latency. Blocks take turns on two copies of what both stages touch. (Single chunk, no divergence / group passes between the stages.)
	const bool piped = deferred && chunk == m && !want_div && !want_grp && ctx->block_pipe;
	const int pb = piped ? (int)(ctx->pipe_next++ & 1) : 0;
	hipStream_t tail = piped ? ctx->tail_stream : ctx->stream;
	DevBuf& b_qT = pb ? ctx->kb_qT2 : ctx->kb_qT;
	DevBuf& b_min = pb ? ctx->kb_min2 : ctx->kb_min;
	DevBuf& b_diff = pb ? ctx->kb_diff2 : ctx->kb_diff;
	DevBuf& b_anib = pb ? ctx->kb_anib2 : ctx->kb_anib;
	DevBuf& b_hot = pb ? ctx->kb_hot2 : ctx->kb_hot;
	DevBuf& b_hot_idx = pb ? ctx->kb_hot_idx2 : ctx->kb_hot_idx;
	// the queries' side of a piped block goes on the prep stream, under the product of the block before it (MSC_GEMM_NO_PREP: on the product's stream, as in r04)
	static const bool no_prep = getenv("MSC_GEMM_NO_PREP") != nullptr;
	hipStream_t prep = piped && !no_prep ? ctx->prep_stream : ctx->stream;
	if (ctx->tail_used)          // a block on ONE stream after piped ones waits for every epilogue in flight; a piped one for the epilogue that read its copy
		for (int i = 0; i < 2; i++)
			if (ctx->tail_busy[i] && (!piped || i == pb)) {
				HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_tail[i], 0));
				if (prep != ctx->stream) HIP_TRY(ctx, hipStreamWaitEvent(prep, ctx->ev_tail[i], 0));          // (it rewrites the transposed image that epilogue read)
				if (!piped) ctx->tail_busy[i] = false;
			}
	if (manh_gemm) {
		const uint64_t nsteps = L.nbins / 128;
		if ((r = ensure(ctx, b_anib, msc_pair_gemm_anib_bytes(L.nbins, kb_qn))) || (r = ensure(ctx, b_qT, msc_pair_gemm_qt_bytes(L.nbins, kb_qn))) || (r = ensure(ctx, b_min, (size_t)gemm_slices * chunk * kb_qn * sizeof(int32_t)))) return r;
		if (n_hot) {
			if ((r = ensure(ctx, b_hot, n_hot * 8)) || (r = ensure(ctx, b_hot_idx, 3 * (nsteps + 1) * sizeof(uint32_t))) ||
			    (r = ensure(ctx, b_diff, chunk * kb_qn * sizeof(int32_t)))) return r;
			hot_ptr = (uint32_t*)b_hot_idx.p;
			hot_cursor = hot_ptr + (nsteps + 1);
			hot_cnt = hot_cursor + (nsteps + 1);
		}
		if (prep != ctx->stream) {
			HIP_TRY(ctx, hipStreamWaitEvent(prep, ctx->ev_call, 0));          // the call's query slots are up
			if (ctx->product_busy[pb]) HIP_TRY(ctx, hipStreamWaitEvent(prep, ctx->ev_product[pb], 0));      // the product that read this copy is through
		}
		// the queries' side of the block, once for all chunks of candidates
		HIP_TRY(ctx, msc_launch_pair_gemm_queries(prep, L.nbins, qset->kb, qset->mb, qset->mb_n, qset->mb_pitch, dq_slots, (uint32_t)n_q, kb_qn,
		                                          (uint8_t*)b_qT.p, n_hot, b_hot.p, hot_ptr, hot_cursor, hot_cnt, (uint8_t*)b_anib.p));
		if (prep != ctx->stream) {
			HIP_TRY(ctx, hipEventRecord(ctx->ev_prep[pb], prep));
			HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prep[pb], 0));
		}
	}
	if (emd_ranks && (r = ensure(ctx, ctx->emd_out, chunk * (manh_gemm ? kb_qn : 64) * sizeof(uint64_t)))) return r;
	const bool count_only = digest && tps == 2 && !digest_emd;
	if (manh_gemm) {
		snprintf(ctx->last_kernel_buf, sizeof ctx->last_kernel_buf, "%s<%u query rows, one matrix product per tile of presence bits%s>", msc_pair_gemm_kernel_name(), kb_qn, emd_ranks ? ", emd by ranks" : ", no emd");
		ctx->last_kernel = ctx->last_kernel_buf;
	} else if (digest) {
		snprintf(ctx->last_kernel_buf, sizeof ctx->last_kernel_buf, "k_pair_digest_multi<%s counts%s%s>", mc_ < 256 ? "u8" : "u16",
		         emd_ranks ? ", emd by ranks" : count_only ? ", no emd" : "", gemm_dot ? ", dot by mfma" : "");
		ctx->last_kernel = ctx->last_kernel_buf;
	} else ctx->last_kernel = ring ? "k_pair_tiles_multi32_ring" : "k_pair_tiles_multi";
	// the digest kernel's workgroup scores up to 16 queries per candidate tile it fetches; the ring kernel's co-located query
	// blocks fetch the tile once per group of tq queries (the followers usually hit in L2, which is not counted on)
	ctx->last_query_tile = manh_gemm ? (int)n_q : digest ? (int)std::min<uint64_t>(n_q, 4 * dg_tq) : (int)std::min<uint64_t>(n_q, (uint64_t)tq);
	const bool whole = chunk == m;                 // one chunk: results land in the caller's arrays with plain copies
	if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_all0, ctx->stream));
	for (uint64_t off = 0; off < m; off += chunk) {
		const uint32_t mc = (uint32_t)std::min(chunk, m - off);
		const uint32_t* d_slots = cand_slots ? (const uint32_t*)ctx->slots.p + off : nullptr;
		const uint8_t* c_bins = cands->bins + (cand_slots ? 0 : off * L.slot_bytes);
		const uint8_t* c_scal = cands->scalars + (cand_slots ? 0 : off * cands->scalar_stride);
		hipEvent_t ev_t0 = ctx->ev_tiles0, ev_t1 = ctx->ev_tiles1;
		if (deferred && ctx->timing && ((r = pool_event(ctx, &ev_t0)) || (r = pool_event(ctx, &ev_t1)))) return r;      // (read when the call's last block is through)
		if (piped) {          // everything the tail needs from this stream so far (slot lists, the cleared error word) is behind this mark
			HIP_TRY(ctx, hipEventRecord(ctx->ev_head[pb], ctx->stream));
			HIP_TRY(ctx, hipStreamWaitEvent(tail, ctx->ev_head[pb], 0));
		}
		if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ev_t0, ctx->stream));
		if (manh_gemm)         // the whole pass over the candidates' bins: products and level products on the matrix cores (timed as the streaming kernel)
			HIP_TRY(ctx, msc_launch_pair_gemm(ctx->stream, L.nbins, cands->kb, d_slots, off, mc, kb_qn, gemm_slices, hot_ptr, b_hot.p,
			                                  (int32_t*)b_min.p, (int32_t*)b_diff.p, (const uint8_t*)b_anib.p));
		else if (digest)
			HIP_TRY(ctx, msc_launch_pair_digest_multi(ctx->stream, L, cands->digest + (cand_slots ? 0 : off * msc_digest_slot_bytes(L)), d_slots, mc, qset->digest,
			                                          dq_slots, (uint32_t)n_q, mc_ < 256, tps, digest_emd, ctx->partials.p, ctx->num_cus, !gemm_dot, dg_tq));
		else if (ring)
			HIP_TRY(ctx, msc_launch_pair_tiles_multi_ring(ctx->stream, L, cands->dtype, c_bins, c_scal, d_slots, mc, qset->bins, qset->L.slot_bytes, qset->scalars,
			                                              qset->scalar_stride, dq_slots, (uint32_t)n_q, tq, prefix16, ctx->partials.p, ctx->num_cus));
		else
			HIP_TRY(ctx, msc_launch_pair_tiles_multi(ctx->stream, L, cands->dtype, c_bins, c_scal, d_slots, mc, qset->bins, qset->L.slot_bytes, qset->scalars,
			                                         qset->scalar_stride, dq_slots, (uint32_t)n_q, tq, compact, (MscPartial*)ctx->partials.p, ctx->num_cus));
		if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ev_t1, ctx->stream));
		if (piped) { HIP_TRY(ctx, hipEventRecord(ctx->ev_product[pb], ctx->stream)); ctx->product_busy[pb] = true; }
		if (emd_ranks && cands->ranks16 && qset->ranks16 && cands->rk_pitch == qset->rk_pitch)          // every reduced rank of both sets fits 16 bits: two per v_sad_u16
			HIP_TRY(ctx, msc_launch_emd_ranks16(tail, L.nbins, cands->ranks16, cands->rk_pitch, cands->rk_n, d_slots, off, mc, qset->ranks16, qset->rk_n,
			                                    dq_slots, (uint32_t)n_q, (uint64_t*)ctx->emd_out.p, manh_gemm ? kb_qn : 64));
		else if (emd_ranks)
			HIP_TRY(ctx, msc_launch_emd_ranks(tail, L.nbins, cands->ranks, cands->rk_pitch, cands->rk_n, d_slots, off, mc, qset->ranks, qset->rk_pitch, qset->rk_n,
			                                  dq_slots, (uint32_t)n_q, (uint64_t*)ctx->emd_out.p, manh_gemm ? kb_qn : 64));
		if (want_div) {
			for (uint64_t q = 0; q < n_q; q++)
				HIP_TRY(ctx, launch_sparse_pass(ctx, spk, c_sp, cands->scalars, cands->scalar_stride, d_slots, off, mc, q_sp, q_slots[q],
				                                qset->scalars + (uint64_t)q_slots[q] * qset->scalar_stride, L.nbins, 0, 0, ~0ull, (MscPartial*)ctx->sp_partials.p,
				                                ctx->div_tables.p, (double*)ctx->div_partials.p + q * mc * dvn * 2, order, 1, dvn));
		}
		if (want_grp) {
			double* gp = (double*)ctx->grp_pairs.p;
			double* gs_c = (double*)ctx->grp_self.p;
			double* gs_q = gs_c + chunk * 16;
			const uint32_t* d_q = dq_slots;
			if (grp_dense) {
				HIP_TRY(ctx, msc_launch_self_markov_dense(ctx->stream, L, cands->dtype, cands->bins, d_slots, off, mc, gs_c));
				HIP_TRY(ctx, msc_launch_self_markov_dense(ctx->stream, qset->L, qset->dtype, qset->bins, d_q, 0, (uint32_t)n_q, gs_q));
				for (uint64_t q = 0; q < n_q; q++)
					HIP_TRY(ctx, msc_launch_pair_groups_dense(ctx->stream, L, cands->dtype, c_bins, c_scal, cands->scalar_stride, d_slots, mc,
					                                          qset->bins + (uint64_t)q_slots[q] * qset->L.slot_bytes, 0, 0, ~0ull, gp + q * mc * 32));
			} else {
				HIP_TRY(ctx, msc_launch_sparse_self_markov(ctx->stream, c_sp->ent, c_sp->hdr, d_slots, off, mc, gs_c));
				HIP_TRY(ctx, msc_launch_sparse_self_markov(ctx->stream, q_sp->ent, q_sp->hdr, d_q, 0, (uint32_t)n_q, gs_q));
				for (uint64_t q = 0; q < n_q; q++)
					HIP_TRY(ctx, msc_launch_pair_sparse_groups(ctx->stream, c_sp->ent, c_sp->hdr + (d_slots ? 0 : off), c_scal, cands->scalar_stride, d_slots, mc, q_sp->ent,
					                                           q_sp->hdr + q_slots[q], 0, 0, ~0ull, gp + q * mc * 32));
			}
		}
		MscEpilogueArgs ea;
		memset(&ea, 0, sizeof ea);
		ea.partials = (const MscPartial*)ctx->partials.p;
		if (want_div) { ea.div_direct = (const double*)ctx->div_partials.p; ea.div_direct_n = dvn; ea.div_base = L.nbins; }
		if (want_grp) { ea.grp_pairs = (const double*)ctx->grp_pairs.p; ea.grp_self_c = (const double*)ctx->grp_self.p; ea.grp_self_q = (const double*)ctx->grp_self.p + chunk * 16; }
		ea.partials16 = ring ? ctx->partials.p : nullptr;
		ea.partials_cq = digest ? ctx->partials.p : nullptr;
		if (manh_gemm) {
			ea.kb_min = (const int32_t*)b_min.p;
			ea.kb_diff = n_hot ? (const int32_t*)b_diff.p : nullptr;
			ea.kb_slices = gemm_slices;
			ea.kb_qn = kb_qn;
			ea.kb_first = cand_slots ? 0 : off;
			ea.kb_c_mb = cands->mb; ea.kb_c_mb_n = cands->mb_n; ea.kb_c_pitch = cands->mb_pitch;
			ea.kb_q_mb = qset->mb; ea.kb_q_mb_n = qset->mb_n; ea.kb_q_pitch = qset->mb_pitch;
			ea.kb_qT = (const uint8_t*)b_qT.p;
			ea.emd_stride = kb_qn;
		}
		ea.cq_group = 4 * dg_tq;
		if (emd_ranks) ea.emd_ranks = (const uint64_t*)ctx->emd_out.p;
		ea.S = n_rec;
		ea.m = (uint32_t)(n_q * mc);
		ea.cand_scalars = c_scal;
		ea.cand_scalar_stride = cands->scalar_stride;
		ea.cand_slots = d_slots;
		ea.n_queries = (uint32_t)n_q;
		ea.m_per_query = mc;
		ea.q_slots = dq_slots;
		ea.qset_scalars = qset->scalars;
		ea.q_scalar_stride = qset->scalar_stride;
		ea.q_scalars = qset->scalars + (uint64_t)q_slots[0] * qset->scalar_stride;
		ea.nbins = L.nbins;
		ea.dtype = cands->dtype;
		ea.order = order;
		ea.feat_mask = feat_mask;
		ea.raw_out = raw_out ? (double*)ctx->raw.p : nullptr;
		ea.model = model ? model->d : nullptr;
		ea.sum_soa = sum_out ? (double*)ctx->soa_sum.p : nullptr;
		ea.csum_soa = csum_out ? (double*)ctx->soa_csum.p : nullptr;
		// (matrix-core pass: the flags go into one of two buffers and back to the host on the copy stream, under the next block's kernels)
		const int pp = ctx->close_pp_next;
		uint8_t* d_close = !close_out ? nullptr : manh_gemm ? (uint8_t*)ctx->close_pp[pp].p : (uint8_t*)ctx->soa_close.p;
		if (close_out && manh_gemm) {
			ctx->close_pp_next ^= 1;
			if (ctx->close_pp_busy[pp]) HIP_TRY(ctx, hipStreamWaitEvent(tail, ctx->ev_copied[pp], 0));
		}
		ea.close_soa = d_close;
		// only the close flags are wanted: k_pair_epilogue_bits decides them in f32 with an error bound and evaluates in FP64 only the
		// pairs the bound leaves open -- the same flags (MSC_NO_SCREEN: FP64 for every pair)
		static const bool no_screen = getenv("MSC_NO_SCREEN") != nullptr;
		ea.screen = manh_gemm && model && model->h.screen_ok && d_close && !sum_out && !csum_out && !raw_out && !want_div && !want_grp && !no_screen;
		ea.error_word = (int32_t*)ctx->err_word.p;
		if (piped) HIP_TRY(ctx, hipStreamWaitEvent(tail, ctx->ev_product[pb], 0));
		HIP_TRY(ctx, msc_launch_epilogue(tail, ea));
		if (close_out && ctx->close_counts_n) HIP_TRY(ctx, msc_launch_close_counts(tail, d_close, (uint32_t)n_q, mc, (uint64_t*)ctx->close_counts.p + ctx->close_counts_base));
		if (piped) {
			HIP_TRY(ctx, hipEventRecord(ctx->ev_tail[pb], tail));
			ctx->tail_busy[pb] = true;
			ctx->tail_used = true;
		}
		if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev_all1, ctx->stream));
		// query-major [n_q][mc] on the device -> [n_q][m] at column `off` on the host
		const size_t rows = (size_t)n_q;
		// (one chunk: the rows are contiguous on both sides -- a plain copy. A 2-D copy whose width is not a multiple of four bytes goes row
		// by row inside the runtime: 1 024 rows of 6 250 flags took 9 ms of a 1.7 ms step)
		auto rows_home = [&](void* dst, size_t dpitch, const void* src, size_t width, hipStream_t st) -> hipError_t {
			if (dpitch == width) return hipMemcpyAsync(dst, src, width * rows, hipMemcpyDeviceToHost, st);
			return hipMemcpy2DAsync(dst, dpitch, src, width, width, rows, hipMemcpyDeviceToHost, st);
		};
		if (sum_out) HIP_TRY(ctx, rows_home(sum_out + off, m * sizeof(double), ctx->soa_sum.p, (size_t)mc * sizeof(double), tail));
		if (csum_out) HIP_TRY(ctx, rows_home(csum_out + off, m * sizeof(double), ctx->soa_csum.p, (size_t)mc * sizeof(double), tail));
		if (close_out && manh_gemm) {
			HIP_TRY(ctx, hipEventRecord(ctx->ev_scored[pp], tail));
			HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_scored[pp], 0));
			HIP_TRY(ctx, rows_home(close_out + off, m, d_close, (size_t)mc, ctx->copy_stream));
			HIP_TRY(ctx, hipEventRecord(ctx->ev_copied[pp], ctx->copy_stream));
			ctx->close_pp_busy[pp] = true;
			ctx->copy_pending = true;
		} else if (close_out) HIP_TRY(ctx, rows_home(close_out + off, m, d_close, (size_t)mc, tail));
		if (raw_out) HIP_TRY(ctx, rows_home(raw_out + off * nf, m * nf * sizeof(double), ctx->raw.p, (size_t)mc * nf * sizeof(double), tail));
		if (deferred) { ctx->tiles_launches++; continue; }
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		float t = 0;
		if (ctx->timing && hipEventElapsedTime(&t, ev_t0, ev_t1) == hipSuccess) { ctx->tiles_ms_accum += t; ctx->tiles_launches++; ctx->have_timing = true; }
		(void)whole;
	}
	if (deferred) return MSC_OK;
	return read_error_word(ctx);
}

