// msc_objects.h -- the objects behind the opaque handles of include/meshclust2_hip.h and the host helpers the C-ABI translation
// units share (msc_api*.hip define them; msc_window.hip and msc_shard.hip use them). Private to the library.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "msc_internal.h"

struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
};

struct msc_ctx {
	int device = -1;
	int num_cus = 256;
	hipStream_t stream = nullptr;
	hipEvent_t ev_tiles0 = nullptr, ev_tiles1 = nullptr, ev_all0 = nullptr, ev_all1 = nullptr;
	bool have_timing = false;
	bool timing = true;                      // record the HIP events behind msc_last_kernel_ms (msc_set_kernel_timing)
	uint32_t last_partial_stride = 0;        // partial records per candidate written by the last run_score
	float tiles_ms_accum = 0.f;
	int tiles_launches = 0;
	const char* last_kernel = "";            // streaming kernel of the last scoring call
	char last_kernel_buf[96] = "";           // (where the Q x M pass composes the name of the form it ran)
	int last_query_tile = 1;                 // queries one HBM read of a candidate tile served in it
	std::string err;
	char dev_name[128] = {0};
	// growable device scratch
	DevBuf partials, pair_out, flags, reduce_out, slots, raw, singles, combos, packed, seg_seq, seg_start, kmer_off, nat, model_tmp,
	    floor_sum, mean, div_tables, div_partials, qslots, soa_sum, soa_csum, soa_close, err_word, seq_seg, seq_ids, seq_meta;
	DevBuf rk_items;                       // ... its list of (candidate, round) items
	DevBuf rk_acc;                         // ... of its long-list form: a record and a spot-term slot per item (k_pair_ranks_items)
	DevBuf rk_counters;                    // ... 2 x 16 words: the query's counts of counts and the number of items, two sets used in turn
	uint32_t rk_turn = 0;
	DevBuf rk_tables;                      // ... the query's tables as the pass's workgroups copy them into LDS, two sets used in turn
	uint32_t rk_table_words = 0;
	DevBuf rk_big;                         // ... the query's counts of 8 and more, for the divergence statistics of the long-list pass (MscRankDiv)
	DevBuf rk_q;                           // the rank list of a pass's query when it is too long for LDS (msc_ranks_pass.hip)
	uint32_t* rk_guard = nullptr;          // page-locked word the rank pass raises when a query's list is longer than its set's bound (msc_ranks_pass.hip)
	DevBuf pin_mean;                       // page-locked: the small pieces a sparse msc_mean_nearest call hands to and fro (msc_api_score.hip)
	DevBuf pin_parts;                      // page-locked: the parts of the fused epilogue + reduce kernel when the host folds them
	DevBuf pin_up, pin_down;               // page-locked HOST staging of the per-call slot list / reduce record + flags
	msc_hist_set* scratch_set = nullptr;   // one slot: the rounded mean of msc_mean_nearest
	msc_hist_set* sparse_scratch = nullptr; // dense slots the sparse builder compacts from
	DevBuf sp_counts, sp_cumbase, sp_acc, sp_chunk_off, sp_chunk_cum, sp_partials, grp_pairs, grp_self, sp_acc_batch, tile_scratch, reduce_parts, sp_touched;
	msc_hist_set* sparse_mean_set = nullptr;   // one sparse slot: the rounded mean of msc_mean_nearest on sparse members
	msc_hist_set* sparse_mean_batch = nullptr; // the rounded means of one chunk of centres (msc_update_centres on sparse sets)
	msc_hist_set* batch_scratch = nullptr;     // rounded means of one chunk of centres (msc_update_centres)
	DevBuf segs, pair_seg, dist;
	uint64_t sp_acc_bins = 0;
	// msc_shard.hip: the payload of msc_colsum_partial, the gathered column-sum lists of the other ranks, header staging
	DevBuf shard_payload, shard_hdrs;
	// msc_pair_gemm.hip: the queries' side of a block (bit image, transposed counts, hot list + its three step arrays), P1 per slice, P2
	DevBuf kb_qT, kb_hot, kb_hot_idx, kb_min, kb_diff, kb_anib;          // (kb_anib: the queries' tiles as nibbles, what the product copies into LDS)
	// the close flags of a block of the matrix-core pass go back to the host on a stream of their own, under the next block's kernels:
	// two device buffers take turns; msc_score_multi waits for the copies before it returns
	hipStream_t copy_stream = nullptr;
	hipEvent_t ev_scored[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
	DevBuf close_pp[2];
	bool close_pp_busy[2] = {false, false};
	int close_pp_next = 0;
	bool copy_pending = false;
	bool block_pipe = true;                // msc_set_block_pipe: the blocks of msc_score_multi on three streams
	bool mirror_pass = true;               // msc_set_mirror_pass: a dense set's 1 x M passes merge the lists of its sparse mirror
	bool packed_on_device = false;         // msc_hist_build_packed_dev: the 2-bit stream of the build in progress is device memory
	bool no_kb_now = false;                // msc_score_multi: this block is taken by the older routes (its hot list would be too long)
	DevBuf close_counts;                   // msc_score_multi: close candidates per query of the call in progress / the last call (msc_last_close_counts)
	uint64_t close_counts_n = 0, close_counts_base = 0;
	bool in_score_multi = false;
	// msc_score_multi queues the blocks of its matrix-core pass without waiting between them: 0 = off, 1 = the next queued block clears
	// the error word, 2 = blocks are in flight (flush_deferred)
	int defer = 0;
	bool defer_cands_up = false;           // the call's candidate slot list is on the device
	uint64_t defer_q_off = 0;              // first query of the block being queued, in qslots_all
	float defer_ms = 0.f;
	DevBuf qslots_all;                     // the query slots of the whole call
	// ... and in two stages on two streams: a block's product (stream) runs beside the rank walk of the same block and the epilogue of
	// the block before it (tail_stream). What both stages touch exists twice (second copies below; the first are kb_qT, kb_min, kb_diff):
	// ev_head[i] = the product of a block using copy i is about to start, ev_product[i] = it is through, ev_tail[i] = the epilogue that
	// read copy i is through
	hipStream_t tail_stream = nullptr;
	hipEvent_t ev_head[2] = {nullptr, nullptr}, ev_product[2] = {nullptr, nullptr}, ev_tail[2] = {nullptr, nullptr};
	DevBuf kb_qT2, kb_min2, kb_diff2;
	// r05: the queries' side of a block (k_kb_gather, k_hot_*) is prepared on a third stream while the product of the block before it
	// runs, so the product stream goes from product to product. Everything that side writes exists twice (the first copies are
	// kb_anib, kb_hot, kb_hot_idx, kb_qT). ev_call = this call's query slots are on the device; ev_prep[i] = the queries' side in copy i
	// is ready; product_busy[i] = a product that reads copy i has been queued and ev_product[i] says when it is through
	hipStream_t prep_stream = nullptr;
	hipEvent_t ev_call = nullptr, ev_prep[2] = {nullptr, nullptr};
	DevBuf kb_anib2, kb_hot2, kb_hot_idx2;
	bool product_busy[2] = {false, false};
	bool tail_busy[2] = {false, false};
	bool tail_used = false;
	uint32_t pipe_next = 0;
	std::vector<hipEvent_t> ev_pool;       // timing events of the queued blocks
	size_t ev_used = 0;
	DevBuf emd_out, rk_bad;                // msc_emd_ranks.hip: the distances of a chunk, the build's error word
	msc_hist_set* shard_gather = nullptr;
	// MSC_PROFILE_CALLS: host wall clock of the 1 x M scoring calls, split into preparing + queueing the slot list, issuing the
	// launches, and waiting for the stream (printed by msc_destroy)
	double prof_prep = 0, prof_issue = 0, prof_wait = 0;
	DevBuf prof_nnz;                       // {stored bins, candidates} the list passes scored inside their windows (device counters)
	double prof_tiles_ms = 0;              // streaming-kernel time of those passes (HIP events; needs kernel timing on)
	uint64_t prof_q_nnz = 0;               // sum over the passes of the query's stored bins
	uint64_t prof_calls = 0, prof_cands = 0;
};

struct msc_hist_set {
	msc_ctx* ctx = nullptr;
	int k = 0, dtype = 0;
	uint64_t capacity = 0;
	MscLayout L;
	uint64_t scalar_stride = 0;
	uint8_t* bins = nullptr;
	uint8_t* scalars = nullptr;
	// host-side bounds over every slot ever written (monotone; used to pick the kernels' integer range)
	uint64_t max_count = 0, max_sum = 0;
	// digest mirror (pair_digest.hip), allocated on the first Q x M pass that can use it; slots [dg_lo, dg_hi) are stale
	mutable uint8_t* digest = nullptr;    // a cache: maintained through const handles
	mutable uint64_t dg_lo = 0, dg_hi = 0;
	mutable bool digest_unavailable = false;      // allocation failed once: do not retry every pass
	// presence-bit mirror (msc_pair_gemm.hip, msc_kbits.h): one BIT per bin = [count >= 2], slots blocked by 32 -- the B operand of the
	// int8 product of the Q x M pass -- and beside it the lists of large bins (count - 1 >= 2) that make the pass exact for any counts:
	// mb[slot][mb_pitch] = (bin, count - 1), unordered; mb_n = entries per slot (also on the host: the size of a query block's hot list is
	// known without a read-back). slots [kb_lo, kb_hi) are stale (mark_stale)
	mutable uint8_t* kb = nullptr;
	mutable uint64_t kb_lo = 0, kb_hi = 0;
	mutable bool kb_unavailable = false;
	mutable bool kb_has_zero = false;         // a slot with a zero count went into the mirror (never the case for built histograms and their means)
	mutable void* mb = nullptr;
	mutable uint32_t* mb_n = nullptr;
	mutable uint32_t mb_pitch = 0;
	mutable std::vector<uint32_t> mb_n_host;
	// ranks mirror (msc_emd_ranks.hip): per slot the bins of its counted k-mers in bin order (rk_pitch entries, padded with 4^k) and
	// their number: the earth mover's distance of the Q x M pass in O(k-mers) instead of O(bins). Built from the digest mirror;
	// slots [rk_lo, rk_hi) are stale
	mutable uint32_t* ranks = nullptr;
	mutable uint32_t* rk_n = nullptr;
	mutable uint64_t rk_pitch = 0, rk_lo = 0, rk_hi = 0;
	mutable bool ranks_unavailable = false;
	// ... and their 16-bit form (rank - floor(t * 4^k / pitch) + 32768; msc_emd_ranks.hip, k_emd_ranks16), kept while every slot's reduced
	// ranks fit and the pitch is a multiple of 1 024; rk16_off: a slot did not fit (or the allocation failed): the 32-bit walk stays
	mutable uint16_t* ranks16 = nullptr;
	mutable bool rk16_off = false;
	// sparse mirror of a DENSE set: the sorted (bin, value) lists of its slots, kept so that the divergence
	// statistics of every route come from the one merge kernel; slots [sm_lo, sm_hi) are stale. Built on first use.
	mutable msc_hist_set* sp_mirror = nullptr;
	mutable uint64_t sm_lo = 0, sm_hi = 0;
	mutable bool sp_mirror_unavailable = false;
	std::vector<uint8_t> written;         // dense sets: slot holds a histogram (unwritten slots are never sparsified)
	// effective lengths as the host last learnt them (len_known[i] != 0): Trainer::get_close / filter / merge derive their length
	// window from the query's length, and reading it back from the device costs a stream round trip per call
	mutable std::vector<uint64_t> len_host;
	mutable std::vector<uint8_t> len_known;
	// sparse layout (sparse.hip): entry arena + per-slot headers instead of `bins`
	bool sparse = false;
	uint2* ent = nullptr;
	uint32_t* cum = nullptr;
	MscSparseHdr* hdr = nullptr;          // device, [capacity]
	std::vector<MscSparseHdr> hdr_host;   // mirror
	uint64_t ent_capacity = 0, ent_used = 0;
	uint32_t max_nnz = 0;                 // longest entry list ever stored (monotone)
	// rank lists of a sparse set (msc_ranks_pass.hip): per slot the bins of its counted k-mers, value - 1 copies of each, at rkl_off[slot],
	// rkl_n[slot] of them. A cache of the lists: every writer of a slot's list bumps list_epoch, the cache remembers the epoch it was built
	// at, and it is (re)built only once the same epoch has been asked for three times -- point sets build theirs once, centre stores (a
	// write every step) never do.
	uint64_t list_epoch = 0;
	mutable uint32_t* rkl = nullptr;
	mutable uint64_t* rkl_off = nullptr;
	mutable uint32_t* rkl_n = nullptr;
	mutable std::vector<uint64_t> rkl_off_host;          // (mirror: a pass whose query is a slot of this set reads its list in place)
	mutable uint64_t rkl_epoch = ~0ull, rkl_seen_epoch = ~0ull, rkl_entries = 0;
	mutable uint32_t rkl_seen = 0;
	mutable bool rkl_unavailable = false;
	// ... and beside them, for the long-list pass (k_pair_ranks_items), the slot's REPEATED bins: (bin, value - 1) for value >= 3, in bin
	// order, rkm_n[slot] of them at rkm_off[slot]; built at the first such pass of an epoch
	mutable uint2* rkm = nullptr;
	mutable uint64_t* rkm_off = nullptr;
	mutable uint32_t* rkm_n = nullptr;
	mutable uint64_t rkm_epoch = ~0ull, rkm_entries = 0;
};

struct msc_model {
	msc_ctx* ctx = nullptr;
	int k = 0;
	MscDevModel h;
	MscDevModel* d = nullptr;
};


int fail(msc_ctx* ctx, int code, const char* fmt, ...);
extern const bool g_trace_calls;
#define HIP_TRY(ctx, expr)                                                                                 \
	do {                                                                                                   \
		if (g_trace_calls) { fprintf(stderr, "[msc] %s:%d %.160s\n", __FILE__, __LINE__, #expr); fflush(stderr); } \
		hipError_t e_ = (expr);                                                                            \
		if (g_trace_calls && e_ == hipSuccess) e_ = hipDeviceSynchronize();                                \
		if (e_ != hipSuccess)                                                                              \
			return fail(ctx, e_ == hipErrorOutOfMemory ? MSC_ERR_OOM : MSC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
			            hipGetErrorString(e_), __FILE__, __LINE__);                                        \
	} while (0)

int ensure(msc_ctx* ctx, DevBuf& b, size_t bytes);             // growable device scratch
int ensure_pinned(msc_ctx* ctx, DevBuf& b, size_t bytes);      // growable page-locked host staging
void release(DevBuf& b);
void mark_written(msc_hist_set* s, uint64_t first, uint64_t n);
int refresh_bounds(msc_ctx* ctx, msc_hist_set* s, uint64_t first, uint64_t n);
void learn_length(const msc_hist_set* s, uint64_t slot, uint64_t len);
int slot_length(msc_ctx* ctx, const msc_hist_set* set, uint64_t slot, uint64_t* len);
int ensure_sparse_mirror(msc_ctx* ctx, const msc_hist_set* set, const msc_hist_set** out);

// the stages of the batched sparse mean (msc_api_batch.hip; msc_shard.hip puts an exchange between them)
int sparse_acc_prepare(msc_ctx* ctx, const MscLayout& L, uint32_t nc, uint32_t** touched_out);
int sparse_acc_scatter(msc_ctx* ctx, const msc_hist_set* src, const uint32_t* slots, const uint32_t* seg, uint64_t P, uint32_t* touched);
int sparse_acc_sweep(msc_ctx* ctx, const msc_hist_set* pts, uint32_t nc, const uint32_t* m_of, int value_bits, uint32_t* touched, uint64_t* floor_sum_out);
int sparse_distances_to_means(msc_ctx* ctx, const msc_hist_set* pts, const std::vector<MscBatchSeg>& segs, const std::vector<uint32_t>& pair_seg,
                              const std::vector<uint32_t>& members, uint32_t nc);
bool needs_wide(const msc_hist_set* a, const msc_hist_set* b);

// one 1 x M scoring pass (msc_api_score.hip): streaming kernel -> epilogue -> optional reduce
struct ScoreRequest {
	const msc_model* model = nullptr;
	const msc_hist_set* cands = nullptr;
	const uint32_t* cand_slots = nullptr;   // host
	uint64_t m = 0;
	const msc_hist_set* qset = nullptr;
	uint64_t q_slot = 0;
	int order = MSC_ORDER_CAND_FIRST;
	int use_window = 0;
	uint64_t min_len = 0, max_len = 0;
	uint64_t feat_mask = 0;
	// host outputs (nullable)
	double* raw_out = nullptr;
	double* singles_out = nullptr;
	double* combos_out = nullptr;
	double* sum_out = nullptr;
	double* csum_out = nullptr;
	double* combo0_out = nullptr;
	int32_t* status_out = nullptr;
	uint8_t* flags_out = nullptr;
	int reduce_mode = -1;                   // <0: no reduce kernel
	int64_t reduce_begin = 0;
	MscReduceOut* reduce_host = nullptr;
	bool only_tiles = false;                // msc_mean_nearest reuses the streaming kernel and folds partials itself
	bool slots_uploaded = false;            // ... and has put these very cand_slots into ctx->slots itself, on this stream (no second copy)
	// msc_get_close_window: the slot list is already on the device (cand_slots == nullptr), the close flags stay there, and
	// after_reduce queues its own kernel behind the reduce kernel, before the call's one stream sync (d_rec = the reduce record)
	const uint32_t* dev_slots = nullptr;
	uint8_t* dev_flags_out = nullptr;
	std::function<hipError_t(const MscReduceOut* d_rec)> after_reduce;
	// ... or, where the pass qualifies for the fused epilogue + reduce kernels, the window's close pass done inside them (pos != nullptr;
	// after_reduce is then not called)
	MscCloseList close_list{nullptr, nullptr, nullptr, nullptr};
};

int run_score(msc_ctx* ctx, ScoreRequest& rq);
