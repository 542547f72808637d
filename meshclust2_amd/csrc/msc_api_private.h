// msc_api_private.h -- what the translation units of the C ABI's host side (msc_api.hip: context, sets, builds, models; msc_api_score.hip: the
// scoring driver and the 1 x M calls; msc_api_multi.hip: msc_score_multi; msc_api_batch.hip: the batched update stage) share beyond
// msc_objects.h / msc_internal.h.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdlib>

#include "msc_objects.h"

// largest bin for which 32-bit per-lane partial sums of p*q cannot overflow: R * max^2 < 2^32 with R <= 64
static const uint64_t kNarrowMaxCount = 8191;
static const uint64_t kNarrowMaxSum = (1ull << 31) - 1;

// MSC_PROFILE_CALLS: the library's own timers inside a call (slot list / launches / stream wait), printed at msc_destroy
static const bool g_profile_calls = getenv("MSC_PROFILE_CALLS") != nullptr;
// from how many bins on the sparse mean sweeps only the 64-byte lines its members touched (MSC_SPARSE_MEAN_GROUPS_MIN_K for A/B runs; r05: from
// k = 9 on -- 200 000 x 1 kb: count + write sweeps 1.73 -> 0.38 ms per 4 150 centres, the update stage 1.27 -> 0.88 s; it was 11 only because the
// two places that cut a list into chunks disagreed below that)
static inline uint64_t msc_sparse_groups_min_bins() {
	static const uint64_t v = [] { const char* e = getenv("MSC_SPARSE_MEAN_GROUPS_MIN_K"); const int k = e ? atoi(e) : 9; return 1ull << (2 * std::max(5, std::min(16, k))); }();
	return v;
}
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int check_slot(msc_ctx* ctx, const msc_hist_set* s, uint64_t slot);          // msc_api.hip: MSC_ERR_INVALID_ARG (with a message) for a slot outside the set

// ---- msc_api_score.hip
const uint64_t kSupportedFeats = MSC_FEAT_SLOW | MSC_FEAT_GROUPS;          // the statistics the GPU path evaluates
static inline double trainer_get_id(double cutoff) { return cutoff > 1 ? cutoff / 100.0 : cutoff; }      // cluster/Trainer.h:35
// both sets and the slots named exist and belong to ctx; same k and dtype
int validate_pair(msc_ctx* ctx, const msc_hist_set* cands, const msc_hist_set* qset, uint64_t q_slot, const uint32_t* slots, uint64_t m);
// which of the merge kernels of sparse.hip takes a pass over the lists of c_sp against slot q_slot of q_sp
enum SparseKernel { SPK_LDS = 0, SPK_MP = 1, SPK_GENERIC = 2 };
SparseKernel pick_sparse_kernel(const msc_hist_set* c_sp, const msc_hist_set* q_sp, uint64_t q_slot, uint64_t max_count, bool wide);
// the rank lists of the sparse set (or sparse mirror) `s` (msc_ranks_pass.hip): true when they are current
bool rank_lists_ready(msc_ctx* ctx, const msc_hist_set* s, int* err, bool eager = false);
uint32_t sparse_records(SparseKernel k, uint32_t mp_parts = 1);          // records per candidate the merge kernel writes
uint32_t div_records(SparseKernel k, uint64_t entries);                  // ... and {jd, js} records per pair
// candidates [off, off + mc) (or the device slot list d_slots) of the sparse set / mirror c_sp against slot q_slot of q_sp
hipError_t launch_sparse_pass(msc_ctx* ctx, SparseKernel k, const msc_hist_set* c_sp, const uint8_t* c_scalars, uint64_t c_stride, const uint32_t* d_slots,
                              uint64_t off, uint32_t mc, const msc_hist_set* q_sp, uint64_t q_slot, const uint8_t* q_scal, uint64_t nbins, int use_window,
                              uint64_t min_len, uint64_t max_len, MscPartial* partials, void* div_tables, void* div_partials, int order, uint32_t parts = 1,
                              uint32_t div_stride = 1);
