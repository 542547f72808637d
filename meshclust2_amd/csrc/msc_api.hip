// msc_api.hip -- the C ABI of libmeshclust2_hip.so (include/meshclust2_hip.h) and the host logic behind it: the context, histogram sets
// and their builds -- sequence encoding, 2-bit packing --, copies, weights-file parsing. The scoring calls live in
// msc_api_score.hip (1 x M), msc_api_multi.hip (Q x M), msc_api_batch.hip (the update stage); r05 split.
// All compute on histograms happens in the gfx950 kernels (hist_build.hip, pair_features.hip); there is no CPU
// fallback: without a usable HIP device msc_create() fails and nothing else can be called.
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

static thread_local std::string g_create_error;

int fail(msc_ctx* ctx, int code, const char* fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (ctx) ctx->err = buf; else g_create_error = buf;
	return code;
}

// the same for the other host translation units (msc_train.hip)
int msc_set_error(msc_ctx* ctx, int code, const char* msg) { return fail(ctx, code, "%s", msg); }
bool msc_ctx_owns(const msc_ctx* ctx, const msc_hist_set* set) { return ctx && set && set->ctx == ctx; }

// MSC_TRACE_CALLS (debugging a device fault): every runtime call / kernel launch is named on stderr before it is issued and the
// device is drained behind it, so the last line printed names the operation that faulted
const bool g_trace_calls = getenv("MSC_TRACE_CALLS") != nullptr;

int ensure(msc_ctx* ctx, DevBuf& b, size_t bytes) {
	if (bytes <= b.cap) return MSC_OK;
	// a buffer that has to grow grows by at least half: callers that come back with slightly larger batches (chunked builds)
	// would otherwise pay a hipFree + hipMalloc pair -- milliseconds each next to a resident 100 GB set -- on every call
	size_t cap = std::max<size_t>(bytes, 4096);
	if (b.p) { cap = std::max(cap, b.cap + b.cap / 2); HIP_TRY(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
	cap = (cap + 4095) / 4096 * 4096;
	if (hipMalloc(&b.p, cap) != hipSuccess) {            // no room for the slack: exactly what was asked for
		(void)hipGetLastError();
		b.p = nullptr;
		cap = (bytes + 4095) / 4096 * 4096;
		HIP_TRY(ctx, hipMalloc(&b.p, cap));
	}
	b.cap = cap;
	return MSC_OK;
}

// page-locked host staging: a pageable hipMemcpyAsync stalls the host on the runtime's own bounce buffer, which the accumulate
// loop pays once per step in each direction
int ensure_pinned(msc_ctx* ctx, DevBuf& b, size_t bytes) {
	if (bytes <= b.cap) return MSC_OK;
	if (b.p) { HIP_TRY(ctx, hipHostFree(b.p)); b.p = nullptr; b.cap = 0; }
	const size_t cap = (std::max<size_t>(bytes, 65536) + 4095) / 4096 * 4096;
	HIP_TRY(ctx, hipHostMalloc(&b.p, cap, hipHostMallocDefault));
	b.cap = cap;
	return MSC_OK;
}

void release(DevBuf& b) {
	if (b.p) (void)hipFree(b.p);
	b.p = nullptr;
	b.cap = 0;
}

// ================================================================================================ context
extern "C" int msc_abi_version(void) { return MSC_ABI_VERSION; }

extern "C" int msc_create(int device, msc_ctx** out) {
	if (!out) return fail(nullptr, MSC_ERR_INVALID_ARG, "msc_create: out is NULL");
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
		return fail(nullptr, MSC_ERR_NO_DEVICE, "msc_create: no HIP device available (%s); this library has no CPU fallback",
		            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
	if (device < 0 || device >= n) return fail(nullptr, MSC_ERR_NO_DEVICE, "msc_create: device %d out of range (0..%d)", device, n - 1);
	e = hipSetDevice(device);
	if (e != hipSuccess) return fail(nullptr, MSC_ERR_NO_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, device);
	if (e != hipSuccess) return fail(nullptr, MSC_ERR_NO_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(nullptr, MSC_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
	msc_ctx* ctx = new msc_ctx();
	ctx->device = device;
	ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	snprintf(ctx->dev_name, sizeof ctx->dev_name, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, ctx->num_cus);
	if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&ctx->ev_tiles0) != hipSuccess ||
	    hipEventCreate(&ctx->ev_tiles1) != hipSuccess || hipEventCreate(&ctx->ev_all0) != hipSuccess || hipEventCreate(&ctx->ev_all1) != hipSuccess ||
	    hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&ctx->tail_stream, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreateWithFlags(&ctx->ev_head[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_head[1], hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&ctx->ev_product[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_product[1], hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&ctx->ev_tail[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_tail[1], hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&ctx->ev_scored[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_scored[1], hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&ctx->ev_copied[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_copied[1], hipEventDisableTiming) != hipSuccess ||
	    hipStreamCreateWithFlags(&ctx->prep_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_call, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&ctx->ev_prep[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_prep[1], hipEventDisableTiming) != hipSuccess) {
		delete ctx;
		return fail(nullptr, MSC_ERR_HIP, "msc_create: stream/event creation failed");
	}
	ctx->mirror_pass = getenv("MSC_NO_MIRROR_1XM") == nullptr;
	ctx->block_pipe = getenv("MSC_GEMM_NO_PIPE") == nullptr;
	*out = ctx;
	return MSC_OK;
}

extern "C" void msc_destroy(msc_ctx* ctx) {
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	(void)hipStreamSynchronize(ctx->copy_stream);
	(void)hipStreamSynchronize(ctx->tail_stream);          // (nothing may still be running on any of the context's streams when its buffers go)
	(void)hipStreamSynchronize(ctx->prep_stream);
	if (g_profile_calls && ctx->prof_calls && ctx->prof_wait > 0) {
		fprintf(stderr, "[msc] 1 x M scoring calls: %llu (%llu candidates) | slot list %.3f s, launches %.3f s, stream wait %.3f s\n", (unsigned long long)ctx->prof_calls,
		        (unsigned long long)ctx->prof_cands, ctx->prof_prep, ctx->prof_issue, ctx->prof_wait);
		if (ctx->prof_nnz.p) {
			// [0], [1]: stored bins and candidates of the merge passes; [2], [3]: of the passes over rank lists (4 bytes per k-mer; a stored bin
			// is at least one k-mer, so 4 bytes per stored bin is a lower bound on what those read)
			uint64_t acc[4] = {0, 0, 0, 0};
			if (hipMemcpy(acc, ctx->prof_nnz.p, sizeof acc, hipMemcpyDeviceToHost) == hipSuccess && acc[1] + acc[3]) {
				const double bytes = 8.0 * acc[0] + 4.0 * acc[2];
				fprintf(stderr, "[msc] list passes: %llu pairs scored inside their length windows (%llu of them over rank lists), %.3f GB of candidate lists (8 bytes per stored bin"
				        " in a merge pass, 4 per k-mer in a rank pass) + %.3f GB of query lists"
				        " (once per pass); over the %.3f s of stream wait above: %.1f M pairs/s, %.1f GB/s of candidate lists = %.3f of the 8 TB/s HBM peak\n",
				        (unsigned long long)(acc[1] + acc[3]), (unsigned long long)acc[3], bytes / 1e9, 8.0 * ctx->prof_q_nnz / 1e9, ctx->prof_wait, (acc[1] + acc[3]) / ctx->prof_wait / 1e6,
				        bytes / ctx->prof_wait / 1e9, bytes / ctx->prof_wait / 8e12);
			}
		}
	}
	if (ctx->scratch_set) msc_hist_set_destroy(ctx->scratch_set);
	if (ctx->sparse_scratch) msc_hist_set_destroy(ctx->sparse_scratch);
	if (ctx->sparse_mean_set) msc_hist_set_destroy(ctx->sparse_mean_set);
	if (ctx->sparse_mean_batch) msc_hist_set_destroy(ctx->sparse_mean_batch);
	if (ctx->batch_scratch) msc_hist_set_destroy(ctx->batch_scratch);
	if (ctx->shard_gather) msc_hist_set_destroy(ctx->shard_gather);
	release(ctx->shard_payload);
	release(ctx->shard_hdrs);
	release(ctx->kb_anib);
	release(ctx->rk_q);
	release(ctx->rk_acc);
	release(ctx->rk_counters);
	release(ctx->rk_tables);
	release(ctx->rk_items);
	release(ctx->rk_big);
	release(ctx->kb_qT);
	release(ctx->kb_hot);
	release(ctx->kb_hot_idx);
	release(ctx->kb_min);
	release(ctx->kb_diff);
	release(ctx->emd_out);
	release(ctx->close_counts);
	release(ctx->rk_bad);
	release(ctx->prof_nnz);
	if (ctx->rk_guard) (void)hipHostFree(ctx->rk_guard);
	if (ctx->pin_up.p) (void)hipHostFree(ctx->pin_up.p);
	if (ctx->pin_down.p) (void)hipHostFree(ctx->pin_down.p);
	if (ctx->pin_parts.p) (void)hipHostFree(ctx->pin_parts.p);
	if (ctx->pin_mean.p) (void)hipHostFree(ctx->pin_mean.p);
	release(ctx->segs);
	release(ctx->pair_seg);
	release(ctx->dist);
	release(ctx->sp_counts);
	release(ctx->sp_cumbase);
	release(ctx->sp_acc);
	release(ctx->sp_chunk_off);
	release(ctx->sp_chunk_cum);
	release(ctx->sp_partials);
	release(ctx->grp_pairs);
	release(ctx->grp_self);
	release(ctx->tile_scratch);
	release(ctx->reduce_parts);
	release(ctx->sp_touched);
	release(ctx->sp_acc_batch);
	DevBuf* bufs[] = {&ctx->partials, &ctx->pair_out, &ctx->flags, &ctx->reduce_out, &ctx->slots, &ctx->raw, &ctx->singles, &ctx->combos,
	                  &ctx->packed, &ctx->seg_seq, &ctx->seg_start, &ctx->kmer_off, &ctx->nat, &ctx->model_tmp, &ctx->floor_sum, &ctx->mean,
	                  &ctx->div_tables, &ctx->div_partials, &ctx->qslots, &ctx->soa_sum, &ctx->soa_csum, &ctx->soa_close,
	                  &ctx->err_word, &ctx->seq_seg, &ctx->seq_ids, &ctx->seq_meta, &ctx->qslots_all};
	for (DevBuf* b : bufs) release(*b);
	for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
	(void)hipEventDestroy(ctx->ev_tiles0);
	(void)hipEventDestroy(ctx->ev_tiles1);
	(void)hipEventDestroy(ctx->ev_all0);
	(void)hipEventDestroy(ctx->ev_all1);
	for (int i = 0; i < 2; i++) { release(ctx->close_pp[i]); (void)hipEventDestroy(ctx->ev_scored[i]); (void)hipEventDestroy(ctx->ev_copied[i]); }
	(void)hipStreamDestroy(ctx->copy_stream);
	release(ctx->kb_qT2);
	release(ctx->kb_min2);
	release(ctx->kb_diff2);
	release(ctx->kb_anib2);
	release(ctx->kb_hot2);
	release(ctx->kb_hot_idx2);
	(void)hipEventDestroy(ctx->ev_call);
	for (int i = 0; i < 2; i++) (void)hipEventDestroy(ctx->ev_prep[i]);
	(void)hipStreamDestroy(ctx->prep_stream);
	for (int i = 0; i < 2; i++) { (void)hipEventDestroy(ctx->ev_head[i]); (void)hipEventDestroy(ctx->ev_product[i]); (void)hipEventDestroy(ctx->ev_tail[i]); }
	(void)hipStreamDestroy(ctx->tail_stream);
	(void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

extern "C" const char* msc_last_error(const msc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int msc_device_name(const msc_ctx* ctx, char* buf, size_t cap) {
	if (!ctx || !buf || cap == 0) return MSC_ERR_INVALID_ARG;
	snprintf(buf, cap, "%s", ctx->dev_name);
	return MSC_OK;
}

extern "C" int msc_synchronize(msc_ctx* ctx) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MSC_OK;
}

extern "C" int msc_set_block_pipe(msc_ctx* ctx, int on) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	ctx->block_pipe = on != 0;
	return MSC_OK;
}

extern "C" int msc_set_mirror_pass(msc_ctx* ctx, int on) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	ctx->mirror_pass = on != 0;
	return MSC_OK;
}

extern "C" int msc_set_kernel_timing(msc_ctx* ctx, int on) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	ctx->timing = on != 0;
	if (!ctx->timing) ctx->have_timing = false;
	return MSC_OK;
}

extern "C" int msc_last_kernel_launches(const msc_ctx* ctx) { return ctx && ctx->have_timing ? ctx->tiles_launches : 0; }

extern "C" int msc_last_kernel_info(const msc_ctx* ctx, char* buf, size_t cap, int* queries_per_candidate_read) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	if (buf && cap) snprintf(buf, cap, "%s", ctx->last_kernel);
	if (queries_per_candidate_read) *queries_per_candidate_read = ctx->last_query_tile;
	return MSC_OK;
}

extern "C" int msc_last_close_counts(msc_ctx* ctx, uint64_t* counts, uint64_t n_q) {
	if (!ctx || !counts) return MSC_ERR_INVALID_ARG;
	if (ctx->close_counts_n == 0 || n_q != ctx->close_counts_n) return fail(ctx, MSC_ERR_UNSUPPORTED, "no close counts of %llu queries on file (the last msc_score_multi call had %llu, or took a route that keeps none)",
	                                                                        (unsigned long long)n_q, (unsigned long long)ctx->close_counts_n);
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpyAsync(counts, ctx->close_counts.p, n_q * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MSC_OK;
}

extern "C" int msc_last_kernel_ms(const msc_ctx* ctx, float* tiles_ms, float* total_ms) {
	if (!ctx || !ctx->have_timing) return MSC_ERR_INVALID_ARG;
	if (tiles_ms) *tiles_ms = ctx->tiles_ms_accum;
	if (total_ms) {
		float t = 0;
		if (hipEventElapsedTime(&t, ctx->ev_all0, ctx->ev_all1) != hipSuccess) return MSC_ERR_HIP;
		*total_ms = t;
	}
	return MSC_OK;
}

// ================================================================================================ a1: encoding (host)
// Restates Chromosome::help (nonltr/Chromosome.cpp:130-154): toUpperCase, removeAmbiguous (:263-291),
// mergeSegments (:298-353, only when the sequence is longer than 20), makeSegmentList (:355-385, 1 Mb fragments),
// then ChromosomeOneDigit::encode with the DNA code table (nonltr/ChromosomeOneDigit.cpp:79-133,
// nonltr/ChromosomeOneDigitDna.cpp:48-68).
namespace {

struct Seg { int64_t s, e; };

struct CodeTable {
	int8_t t[256];
	CodeTable() {
		for (int i = 0; i < 256; i++) t[i] = -1;
		const char* zero = "AMV"; const char* one = "CYHN"; const char* two = "GRSX"; const char* three = "TKWBD";
		for (const char* p = zero; *p; p++) t[(unsigned char)*p] = 0;
		for (const char* p = one; *p; p++) t[(unsigned char)*p] = 1;
		for (const char* p = two; *p; p++) t[(unsigned char)*p] = 2;
		for (const char* p = three; *p; p++) t[(unsigned char)*p] = 3;
	}
};
const CodeTable kCodes;

// returns false on a character outside the table
bool encode_sequence(const char* seq, size_t len, std::vector<uint8_t>& codes, std::vector<Seg>& segs, uint64_t& eff_len) {
	const int64_t n = (int64_t)len;
	codes.resize(len);
	for (int64_t i = 0; i < n; i++) codes[i] = (uint8_t)toupper((unsigned char)seq[i]);

	// maximal runs of non-'N'. A run that begins on the very last character is dropped, exactly like the
	// reference's if / else-if chain (the "start" branch and the "close at end" branch are exclusive).
	std::vector<Seg> runs;
	int64_t start = -1;
	for (int64_t i = 0; i < n; i++) {
		const bool is_n = codes[i] == 'N';
		if (!is_n && start < 0) start = i;
		else if (is_n && start >= 0) { runs.push_back({start, i - 1}); start = -1; }
		else if (i == n - 1 && !is_n && start >= 0) { runs.push_back({start, i}); start = -1; }
	}

	// join runs separated by fewer than 10 positions, drop joined runs shorter than 20
	if (n > 20 && !runs.empty()) {
		std::vector<Seg> merged;
		Seg cur = runs[0];
		for (size_t i = 1; i < runs.size(); i++) {
			if (runs[i].s - cur.e < 10) cur.e = runs[i].e;
			else { if (cur.e - cur.s + 1 >= 20) merged.push_back(cur); cur = runs[i]; }
		}
		if (cur.e - cur.s + 1 >= 20) merged.push_back(cur);
		runs.swap(merged);
	}

	// cut runs longer than 1,000,000 into floor(len/1e6) fragments, the last one taking the remainder
	const int64_t frag = 1000000;
	segs.clear();
	for (const Seg& r : runs) {
		const int64_t l = r.e - r.s + 1;
		if (l > frag) {
			const int64_t nf = l / frag;
			for (int64_t h = 0; h < nf; h++) {
				const int64_t fs = r.s + h * frag;
				segs.push_back({fs, h == nf - 1 ? r.e : fs + frag - 1});
			}
		} else segs.push_back(r);
	}
	eff_len = 0;
	for (const Seg& s : segs) eff_len += (uint64_t)(s.e - s.s + 1);

	// digits inside segments (every char must map); outside segments everything but 'N' is mapped too and must be valid
	std::vector<uint8_t> in_seg;   // cheap marker only when needed
	size_t si = 0;
	for (int64_t i = 0; i < n; i++) {
		while (si < segs.size() && segs[si].e < i) si++;
		const bool inside = si < segs.size() && segs[si].s <= i;
		const uint8_t c = codes[i];
		if (inside) {
			const int8_t d = kCodes.t[c];
			if (d < 0) return false;
			codes[i] = (uint8_t)d;
		} else if (c != 'N') {
			if (segs.empty()) continue;           // the reference only walks the gaps when at least one segment exists
			const int8_t d = kCodes.t[c];
			if (d < 0) return false;
			codes[i] = (uint8_t)d;
		}
	}
	return true;
}

}  // namespace

extern "C" int msc_encode(const char* seq, size_t len, uint8_t* codes_out, int64_t* segs_out, size_t max_segs, size_t* n_segs,
                          uint64_t* eff_len) {
	if (!seq && len) return MSC_ERR_INVALID_ARG;
	std::vector<uint8_t> codes;
	std::vector<Seg> segs;
	uint64_t eff = 0;
	if (!encode_sequence(seq, len, codes, segs, eff)) return MSC_ERR_INVALID_INPUT;
	if (codes_out && len) memcpy(codes_out, codes.data(), len);
	for (size_t i = 0; i < segs.size() && i < max_segs && segs_out; i++) { segs_out[2 * i] = segs[i].s; segs_out[2 * i + 1] = segs[i].e; }
	if (n_segs) *n_segs = segs.size();
	if (eff_len) *eff_len = eff;
	return MSC_OK;
}

// ================================================================================================ histogram sets
static bool valid_dtype(int d) { return d == 8 || d == 16 || d == 32 || d == 64; }

extern "C" int msc_hist_set_create(msc_ctx* ctx, int k, int dtype, uint64_t capacity, msc_hist_set** out) {
	if (!ctx || !out) return MSC_ERR_INVALID_ARG;
	*out = nullptr;
	if (!valid_dtype(dtype)) return fail(ctx, MSC_ERR_INVALID_ARG, "dtype must be 8, 16, 32 or 64 (got %d)", dtype);
	if (k < 1 || k > 13) return fail(ctx, MSC_ERR_UNSUPPORTED, "dense histograms support 1 <= k <= 13 (got %d)", k);
	if (capacity == 0) return fail(ctx, MSC_ERR_INVALID_ARG, "capacity must be > 0");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	msc_hist_set* s = new msc_hist_set();
	s->ctx = ctx;
	s->k = k;
	s->dtype = dtype;
	s->capacity = capacity;
	s->L = msc_make_layout(k, dtype);
	s->scalar_stride = msc_scalar_stride(s->L.S);
	hipError_t e = hipMalloc((void**)&s->bins, s->L.slot_bytes * capacity);
	if (e == hipSuccess) e = hipMalloc((void**)&s->scalars, s->scalar_stride * capacity);
	if (e != hipSuccess) {
		if (s->bins) (void)hipFree(s->bins);
		delete s;
		return fail(ctx, MSC_ERR_OOM, "hipMalloc of %llu slots x %llu bytes failed: %s", (unsigned long long)capacity,
		            (unsigned long long)s->L.slot_bytes, hipGetErrorString(e));
	}
	e = hipMemsetAsync(s->scalars, 0, s->scalar_stride * capacity, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess) { (void)hipFree(s->bins); (void)hipFree(s->scalars); delete s; return fail(ctx, MSC_ERR_HIP, "memset: %s", hipGetErrorString(e)); }
	*out = s;
	return MSC_OK;
}

extern "C" int msc_hist_set_create_sparse(msc_ctx* ctx, int k, int dtype, uint64_t capacity, uint64_t max_entries, msc_hist_set** out) {
	if (!ctx || !out) return MSC_ERR_INVALID_ARG;
	*out = nullptr;
	if (!valid_dtype(dtype)) return fail(ctx, MSC_ERR_INVALID_ARG, "dtype must be 8, 16, 32 or 64 (got %d)", dtype);
	if (k < 1 || k > 15) return fail(ctx, MSC_ERR_UNSUPPORTED, "sparse histograms support k <= 15 (got %d)", k);
	if (capacity == 0 || max_entries == 0) return fail(ctx, MSC_ERR_INVALID_ARG, "capacity and max_entries must be > 0");
	const MscLayout L = msc_make_layout(k, dtype);
	if (L.S % MSC_SPARSE_SUB != 0) return fail(ctx, MSC_ERR_UNSUPPORTED, "sparse layout needs 4^k*sizeof(T) >= 64 KiB (k=%d, dtype=%d): use a dense set", k, dtype);
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	msc_hist_set* s = new msc_hist_set();
	s->ctx = ctx; s->k = k; s->dtype = dtype; s->capacity = capacity; s->L = L;
	s->sparse = true;
	s->scalar_stride = msc_scalar_stride(L.S);        // same record as a dense slot of this shape (tile prefixes unused)
	s->scalar_stride = sizeof(MscSlotScalars);
	s->ent_capacity = max_entries;
	hipError_t e = hipMalloc((void**)&s->scalars, s->scalar_stride * capacity);
	if (e == hipSuccess) e = hipMalloc((void**)&s->ent, (max_entries + 2) * sizeof(uint2));      // (+2: the merge kernel stages lists two entries at a time)
	if (e == hipSuccess) e = hipMalloc((void**)&s->cum, max_entries * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMalloc((void**)&s->hdr, capacity * sizeof(MscSparseHdr));
	if (e == hipSuccess) e = hipMemsetAsync(s->scalars, 0, s->scalar_stride * capacity, ctx->stream);
	if (e == hipSuccess) e = hipMemsetAsync(s->hdr, 0, capacity * sizeof(MscSparseHdr), ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess) {
		if (s->scalars) (void)hipFree(s->scalars);
		if (s->ent) (void)hipFree(s->ent);
		if (s->cum) (void)hipFree(s->cum);
		if (s->hdr) (void)hipFree(s->hdr);
		delete s;
		return fail(ctx, MSC_ERR_OOM, "sparse set allocation failed: %s", hipGetErrorString(e));
	}
	s->hdr_host.assign(capacity, MscSparseHdr{});
	*out = s;
	return MSC_OK;
}

static void forget_lengths(const msc_hist_set* s, uint64_t first, uint64_t n);
// a sparse set back to the state msc_hist_set_create_sparse left it in: no slot holds a list, the whole arena is free
extern "C" int msc_hist_set_clear(msc_ctx* ctx, msc_hist_set* s) {
	if (!ctx || !s || s->ctx != ctx) return MSC_ERR_INVALID_ARG;
	if (!s->sparse) return fail(ctx, MSC_ERR_UNSUPPORTED, "msc_hist_set_clear: sparse sets only (a dense slot is overwritten in place)");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemsetAsync(s->scalars, 0, s->scalar_stride * s->capacity, ctx->stream));
	HIP_TRY(ctx, hipMemsetAsync(s->hdr, 0, s->capacity * sizeof(MscSparseHdr), ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	s->hdr_host.assign(s->capacity, MscSparseHdr{});
	s->ent_used = 0;
	s->list_epoch++;
	s->max_nnz = 0;
	s->max_count = s->max_sum = 0;
	forget_lengths(s, 0, s->capacity);
	return MSC_OK;
}

extern "C" int msc_hist_set_is_sparse(const msc_hist_set* s) { return s && s->sparse ? 1 : 0; }
extern "C" uint64_t msc_hist_set_entries(const msc_hist_set* s, uint64_t slot) {
	if (!s || slot >= s->capacity) return 0;
	if (s->sparse) return s->hdr_host[slot].nnz;
	return s->sp_mirror && slot < s->sp_mirror->hdr_host.size() ? s->sp_mirror->hdr_host[slot].nnz : 0;
}

extern "C" void msc_hist_set_destroy(msc_hist_set* s) {
	if (!s) return;
	(void)hipSetDevice(s->ctx->device);
	(void)hipStreamSynchronize(s->ctx->stream);
	if (s->bins) (void)hipFree(s->bins);
	if (s->scalars) (void)hipFree(s->scalars);
	if (s->digest) (void)hipFree(s->digest);
	if (s->kb) (void)hipFree(s->kb);
	if (s->mb) (void)hipFree(s->mb);
	if (s->mb_n) (void)hipFree(s->mb_n);
	if (s->ranks) (void)hipFree(s->ranks);
	if (s->ranks16) (void)hipFree(s->ranks16);
	if (s->rk_n) (void)hipFree(s->rk_n);
	if (s->sp_mirror) msc_hist_set_destroy(s->sp_mirror);
	if (s->ent) (void)hipFree(s->ent);
	if (s->cum) (void)hipFree(s->cum);
	if (s->hdr) (void)hipFree(s->hdr);
	if (s->rkl) (void)hipFree(s->rkl);
	if (s->rkl_off) (void)hipFree(s->rkl_off);
	if (s->rkl_n) (void)hipFree(s->rkl_n);
	if (s->rkm) (void)hipFree(s->rkm);
	if (s->rkm_off) (void)hipFree(s->rkm_off);
	if (s->rkm_n) (void)hipFree(s->rkm_n);
	delete s;
}

extern "C" uint64_t msc_hist_set_capacity(const msc_hist_set* s) { return s ? s->capacity : 0; }
extern "C" int msc_hist_set_k(const msc_hist_set* s) { return s ? s->k : 0; }
extern "C" int msc_hist_set_dtype(const msc_hist_set* s) { return s ? s->dtype : 0; }
extern "C" uint64_t msc_hist_set_bytes(const msc_hist_set* s) {
	if (!s) return 0;
	if (s->sparse) return s->ent_capacity * 12 + (s->scalar_stride + sizeof(MscSparseHdr)) * s->capacity + (s->rkl ? s->rkl_entries * 4 + s->capacity * 12 : 0) + (s->rkm ? s->rkm_entries * 8 + s->capacity * 12 : 0);
	return (s->L.slot_bytes + (s->digest ? msc_digest_slot_bytes(s->L) : 0) + (s->kb ? s->L.padded_bins / 8 + 32 + (uint64_t)s->mb_pitch * 8 + 4 : 0) + (s->ranks ? (s->rk_pitch + 1) * 4 : 0) + s->scalar_stride) * s->capacity;
}

// every writer of slots ends here: both mirrors of a dense set (digest, sparse lists) are stale for [first, first + n)
static void mark_stale(msc_hist_set* s, uint64_t first, uint64_t n) {
	if (s->sparse || n == 0) return;
	if (s->digest) {
		if (s->dg_lo >= s->dg_hi) { s->dg_lo = first; s->dg_hi = first + n; }
		else { s->dg_lo = std::min(s->dg_lo, first); s->dg_hi = std::max(s->dg_hi, first + n); }
	}
	if (s->sp_mirror) {
		if (s->sm_lo >= s->sm_hi) { s->sm_lo = first; s->sm_hi = first + n; }
		else { s->sm_lo = std::min(s->sm_lo, first); s->sm_hi = std::max(s->sm_hi, first + n); }
	}
	if (s->kb) {
		if (s->kb_lo >= s->kb_hi) { s->kb_lo = first; s->kb_hi = first + n; }
		else { s->kb_lo = std::min(s->kb_lo, first); s->kb_hi = std::max(s->kb_hi, first + n); }
	}
	if (s->ranks) {
		if (s->rk_lo >= s->rk_hi) { s->rk_lo = first; s->rk_hi = first + n; }
		else { s->rk_lo = std::min(s->rk_lo, first); s->rk_hi = std::max(s->rk_hi, first + n); }
	}
}

static void forget_lengths(const msc_hist_set* s, uint64_t first, uint64_t n) {
	for (uint64_t i = first; i < first + n && i < s->len_known.size(); i++) s->len_known[i] = 0;
}
void learn_length(const msc_hist_set* s, uint64_t slot, uint64_t len) {
	if (s->len_known.size() < s->capacity) { s->len_known.resize(s->capacity, 0); s->len_host.resize(s->capacity, 0); }
	s->len_host[slot] = len;
	s->len_known[slot] = 1;
}

void mark_written(msc_hist_set* s, uint64_t first, uint64_t n) {
	forget_lengths(s, first, n);
	if (s->sparse || n == 0) return;
	if (s->written.size() < s->capacity) s->written.resize(s->capacity, 0);
	for (uint64_t i = first; i < first + n && i < s->capacity; i++) s->written[i] = 1;
	mark_stale(s, first, n);
}

// pull the scalar records of [first, first+n) and fold their maxima into the set's host-side bounds
int refresh_bounds(msc_ctx* ctx, msc_hist_set* s, uint64_t first, uint64_t n) {
	mark_written(s, first, n);
	std::vector<MscSlotScalars> h(n);
	HIP_TRY(ctx, hipMemcpy2DAsync(h.data(), sizeof(MscSlotScalars), s->scalars + first * s->scalar_stride, s->scalar_stride,
	                              sizeof(MscSlotScalars), n, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	for (uint64_t i = 0; i < n; i++) {
		s->max_count = std::max(s->max_count, h[i].max_count);
		s->max_sum = std::max(s->max_sum, h[i].sum);
		learn_length(s, first + i, h[i].length);
	}
	if (s->sparse) for (uint64_t i = first; i < first + n; i++) s->max_nnz = std::max(s->max_nnz, s->hdr_host[i].nnz);
	return MSC_OK;
}



// Compaction of dense slots into sparse slots (k_sparse_count + k_sparse_write): dense slots [d_first, d_first + n) of `dense`
// become slots [s_first, s_first + n) of the sparse set `sp`. need_only: just report how many entries they would take.
static int sparsify_slots(msc_ctx* ctx, const msc_hist_set* dense, uint64_t d_first, msc_hist_set* sp, uint64_t s_first, uint64_t n, uint64_t* need_only) {
	const MscLayout& L = dense->L;
	int r;
	const uint64_t B = std::min<uint64_t>(n, 16384);
	if ((r = ensure(ctx, ctx->sp_counts, B * MSC_SPARSE_SUB * 2 * sizeof(uint64_t)))) return r;
	if ((r = ensure(ctx, ctx->sp_cumbase, B * MSC_SPARSE_SUB * sizeof(uint64_t)))) return r;
	std::vector<uint64_t> counts, cumbase;
	if (need_only) *need_only = 0;
	for (uint64_t b0 = 0; b0 < n; b0 += B) {
		const uint64_t nb = std::min(B, n - b0);
		const uint8_t* src = dense->bins + (d_first + b0) * L.slot_bytes;
		HIP_TRY(ctx, msc_launch_sparse_count(ctx->stream, src, L, dense->dtype, (uint32_t)nb, (uint64_t*)ctx->sp_counts.p));
		counts.resize(nb * MSC_SPARSE_SUB * 2);
		HIP_TRY(ctx, hipMemcpyAsync(counts.data(), ctx->sp_counts.p, counts.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (need_only) {
			for (uint64_t i = 0; i < nb * MSC_SPARSE_SUB; i++) *need_only += counts[i * 2];
			continue;
		}
		cumbase.assign(nb * MSC_SPARSE_SUB, 0);
		for (uint64_t i = 0; i < nb; i++) {
			MscSparseHdr h{};
			uint64_t cnt = 0, ex = 0;
			for (int w = 0; w < MSC_SPARSE_SUB; w++) {
				h.split[w] = (uint32_t)cnt;
				cumbase[i * MSC_SPARSE_SUB + w] = ex;
				cnt += counts[(i * MSC_SPARSE_SUB + w) * 2];
				ex += counts[(i * MSC_SPARSE_SUB + w) * 2 + 1];
			}
			h.split[MSC_SPARSE_SUB] = (uint32_t)cnt;
			h.nnz = (uint32_t)cnt;
			if (sp->ent_used + cnt > sp->ent_capacity)
				return fail(ctx, MSC_ERR_OOM, "sparse set entry arena exhausted (%llu of %llu entries used, slot %llu needs %llu)",
				            (unsigned long long)sp->ent_used, (unsigned long long)sp->ent_capacity, (unsigned long long)(s_first + b0 + i), (unsigned long long)cnt);
			h.off = sp->ent_used;
			sp->ent_used += cnt;
			sp->hdr_host[s_first + b0 + i] = h;
			sp->list_epoch++;
			sp->max_nnz = std::max(sp->max_nnz, h.nnz);
		}
		HIP_TRY(ctx, hipMemcpyAsync(sp->hdr + s_first + b0, sp->hdr_host.data() + s_first + b0, nb * sizeof(MscSparseHdr), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_cumbase.p, cumbase.data(), cumbase.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, msc_launch_sparse_write(ctx->stream, src, L, dense->dtype, (uint32_t)nb, sp->hdr, s_first + b0, (const uint64_t*)ctx->sp_cumbase.p, sp->ent, sp->cum));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // cumbase / counts are reused by the next batch
	}
	return MSC_OK;
}

// The sparse mirror of a dense set: nullptr where the sparse layout does not exist (histograms under 64 KiB) or
// cannot be allocated. Refreshes the written slots of the stale hull; when the append-only arena runs out, the mirror is rebuilt
// compactly from every written slot.
int ensure_sparse_mirror(msc_ctx* ctx, const msc_hist_set* set, const msc_hist_set** out) {
	*out = nullptr;
	static const bool disabled = getenv("MSC_NO_SPARSE_MIRROR") != nullptr;
	if (set->sparse) { *out = set; return MSC_OK; }
	if (disabled || set->sp_mirror_unavailable || set->L.S % MSC_SPARSE_SUB != 0 || set->k > 15 || set->written.empty()) return MSC_OK;
	// runs of written slots inside [lo, hi)
	auto runs_of = [&](uint64_t lo, uint64_t hi) {
		std::vector<std::pair<uint64_t, uint64_t> > runs;
		for (uint64_t i = lo; i < hi;) {
			if (!set->written[i]) { i++; continue; }
			uint64_t j = i;
			while (j < hi && set->written[j]) j++;
			runs.emplace_back(i, j - i);
			i = j;
		}
		return runs;
	};
	int r;
	for (int attempt = 0; attempt < 2; attempt++) {
		uint64_t lo = set->sm_lo, hi = set->sm_hi;
		if (!set->sp_mirror) { lo = 0; hi = set->capacity; }
		if (lo >= hi) break;
		const auto runs = runs_of(lo, hi);
		uint64_t need = 0;
		for (const auto& run : runs) {
			uint64_t nr = 0;
			if ((r = sparsify_slots(ctx, set, run.first, nullptr, 0, run.second, &nr))) return r;
			need += nr;
		}
		if (set->sp_mirror && set->sp_mirror->ent_used + need > set->sp_mirror->ent_capacity) {
			// arena full (slots rewritten many times leave their old entries behind): start over, compactly
			msc_hist_set_destroy(set->sp_mirror);
			set->sp_mirror = nullptr;
			continue;
		}
		if (!set->sp_mirror) {
			msc_hist_set* m = nullptr;
			const uint64_t arena = need + need / 2 + (1u << 16);
			if (msc_hist_set_create_sparse(ctx, set->k, set->dtype, set->capacity, arena, &m) != MSC_OK) { set->sp_mirror_unavailable = true; return MSC_OK; }
			set->sp_mirror = m;
		}
		for (const auto& run : runs)
			if ((r = sparsify_slots(ctx, set, run.first, set->sp_mirror, run.first, run.second, nullptr))) return r;
		set->sm_lo = set->sm_hi = 0;
		break;
	}
	set->sp_mirror->max_count = set->max_count;
	set->sp_mirror->max_sum = set->max_sum;
	*out = set->sp_mirror;
	return MSC_OK;
}

// Direct sparse build (k_sparse_build_sort): returns 1 when the batch does not qualify (a sequence with > 32768 k-mers,
// segments not grouped by sequence, or MSC_NO_SORT_BUILD set) and the scratch + compaction path must be used instead.
static int build_sparse_sort(msc_ctx* ctx, msc_hist_set* set, uint64_t first_slot, uint64_t n_seqs, const uint8_t* packed, uint64_t n_bases,
                             const uint32_t* seg_seq, const uint64_t* seg_start, const uint64_t* seg_end, uint64_t n_segs, const uint64_t* eff_len,
                             const uint64_t* one_mers) {
	static const bool disabled = getenv("MSC_NO_SORT_BUILD") != nullptr;
	if (disabled) return 1;
	const int k = set->k;
	for (uint64_t j = 1; j < n_segs; j++) if (seg_seq[j] < seg_seq[j - 1]) return 1;
	std::vector<uint64_t> koff(n_segs + 1, 0), per_seq(n_seqs, 0), sbeg(n_seqs + 1, 0), aoff(n_seqs, 0);
	for (uint64_t j = 0; j < n_segs; j++) {
		if (seg_seq[j] >= n_seqs || seg_end[j] < seg_start[j] || seg_end[j] >= n_bases) return fail(ctx, MSC_ERR_INVALID_ARG, "segment %llu is malformed", (unsigned long long)j);
		const uint64_t len = seg_end[j] - seg_start[j] + 1;
		const uint64_t nk = len >= (uint64_t)k ? len - k + 1 : 0;
		koff[j + 1] = koff[j] + nk;
		per_seq[seg_seq[j]] += nk;
		sbeg[seg_seq[j] + 1]++;
	}
	for (uint64_t i = 0; i < n_seqs; i++) sbeg[i + 1] += sbeg[i];
	uint64_t longest = 0, need = 0;
	for (uint64_t i = 0; i < n_seqs; i++) { longest = std::max(longest, per_seq[i]); aoff[i] = set->ent_used + need; need += per_seq[i]; }
	if (longest > 32768) return 1;
	if (set->ent_used + need > set->ent_capacity)
		return fail(ctx, MSC_ERR_OOM, "sparse set entry arena exhausted (%llu of %llu entries used, this build may need %llu)", (unsigned long long)set->ent_used,
		            (unsigned long long)set->ent_capacity, (unsigned long long)need);
	uint32_t P = 64;
	while (P < longest) P <<= 1;
	int r;
	std::vector<MscSlotScalars> sc(n_seqs);
	memset(sc.data(), 0, sizeof(MscSlotScalars) * n_seqs);
	for (uint64_t i = 0; i < n_seqs; i++) {
		sc[i].length = eff_len[i];
		for (int b = 0; b < 4; b++) sc[i].one_mers[b] = one_mers ? one_mers[4 * i + b] : 0;
		sc[i].n_kmers = per_seq[i];
	}
	HIP_TRY(ctx, hipMemcpy2DAsync(set->scalars + first_slot * set->scalar_stride, set->scalar_stride, sc.data(), sizeof(MscSlotScalars), sizeof(MscSlotScalars), n_seqs,
	                              hipMemcpyHostToDevice, ctx->stream));
	// + 12 zero bytes behind the last whole word (the kernel's 64-bit window never reads past the end); the memset also clears the
	// last word itself before the copy fills it. A batch without a single base (n_bases == 0) is just the 12 zero bytes.
	const size_t packed_bytes = (size_t)((n_bases + 3) / 4), round4 = (packed_bytes + 3) / 4 * 4, padded_bytes = round4 + 12;
	const size_t tail_off = round4 >= 4 ? round4 - 4 : 0;
	if ((r = ensure(ctx, ctx->packed, padded_bytes))) return r;
	HIP_TRY(ctx, hipMemsetAsync((uint8_t*)ctx->packed.p + tail_off, 0, padded_bytes - tail_off, ctx->stream));
	if (packed_bytes) HIP_TRY(ctx, hipMemcpyAsync(ctx->packed.p, packed, packed_bytes, ctx->packed_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
	if ((r = ensure(ctx, ctx->seg_start, std::max<size_t>(n_segs, 1) * sizeof(uint64_t)))) return r;
	if ((r = ensure(ctx, ctx->kmer_off, (n_segs + 1) * sizeof(uint64_t)))) return r;
	if ((r = ensure(ctx, ctx->seq_seg, (n_seqs + 1) * sizeof(uint64_t)))) return r;
	if ((r = ensure(ctx, ctx->sp_cumbase, std::max<size_t>(n_seqs * sizeof(uint64_t), ctx->sp_cumbase.cap)))) return r;
	if (n_segs) HIP_TRY(ctx, hipMemcpyAsync(ctx->seg_start.p, seg_start, n_segs * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->kmer_off.p, koff.data(), (n_segs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->seq_seg.p, sbeg.data(), (n_seqs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_cumbase.p, aoff.data(), n_seqs * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, msc_launch_sparse_build_sort(ctx->stream, k, set->dtype, set->L.nbins, first_slot, (uint32_t)n_seqs, (const uint32_t*)ctx->packed.p,
	                                          (const uint64_t*)ctx->seg_start.p, (const uint64_t*)ctx->kmer_off.p, (const uint64_t*)ctx->seq_seg.p,
	                                          (const uint64_t*)ctx->sp_cumbase.p, P, set->scalars, set->scalar_stride, set->hdr, set->ent, set->cum));
	set->ent_used += need;
	set->list_epoch++;
	HIP_TRY(ctx, hipMemcpyAsync(set->hdr_host.data() + first_slot, set->hdr + first_slot, n_seqs * sizeof(MscSparseHdr), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return refresh_bounds(ctx, set, first_slot, n_seqs);
}

// Sparse build: dense-build a batch into scratch slots (the validated builder, bit-exact counts and saturation), then
// compact each scratch slot in index order into the set's entry arena (sparse.hip).
static int build_sparse(msc_ctx* ctx, msc_hist_set* set, uint64_t first_slot, uint64_t n_seqs, const uint8_t* packed, uint64_t n_bases,
                        const uint32_t* seg_seq, const uint64_t* seg_start, const uint64_t* seg_end, uint64_t n_segs, const uint64_t* eff_len,
                        const uint64_t* one_mers) {
	const MscLayout& L = set->L;
	int r0;
	if ((r0 = build_sparse_sort(ctx, set, first_slot, n_seqs, packed, n_bases, seg_seq, seg_start, seg_end, n_segs, eff_len, one_mers)) != 1) return r0;
	// scratch capacity: <= 8 GiB of dense slots
	uint64_t B = (8ull << 30) / L.slot_bytes;
	B = std::max<uint64_t>(1, std::min<uint64_t>(B, std::min<uint64_t>(n_seqs, 4096)));
	int r;
	if (!ctx->sparse_scratch || ctx->sparse_scratch->k != set->k || ctx->sparse_scratch->dtype != set->dtype || ctx->sparse_scratch->capacity < B) {
		if (ctx->sparse_scratch) { msc_hist_set_destroy(ctx->sparse_scratch); ctx->sparse_scratch = nullptr; }
		if ((r = msc_hist_set_create(ctx, set->k, set->dtype, B, &ctx->sparse_scratch))) return r;
	}
	msc_hist_set* sc = ctx->sparse_scratch;
	// segments grouped by sequence (msc_hist_build emits them so); slice them per batch
	for (uint64_t j = 1; j < n_segs; j++) if (seg_seq[j] < seg_seq[j - 1]) return fail(ctx, MSC_ERR_INVALID_ARG, "sparse build needs segments ordered by sequence");
	std::vector<uint64_t> sbeg(n_seqs + 1, 0);
	for (uint64_t j = 0; j < n_segs; j++) sbeg[seg_seq[j] + 1]++;
	for (uint64_t i = 0; i < n_seqs; i++) sbeg[i + 1] += sbeg[i];
	for (uint64_t b0 = 0; b0 < n_seqs; b0 += B) {
		const uint64_t nb = std::min(B, n_seqs - b0);
		const uint64_t s0 = sbeg[b0], s1 = sbeg[b0 + nb];
		std::vector<uint32_t> sseq(s1 - s0);
		for (uint64_t j = s0; j < s1; j++) sseq[j - s0] = seg_seq[j] - (uint32_t)b0;
		if ((r = msc_hist_build_packed(ctx, sc, 0, nb, packed, n_bases, sseq.data(), seg_start + s0, seg_end + s0, s1 - s0, eff_len + b0,
		                               one_mers ? one_mers + 4 * b0 : nullptr)))
			return r;
		if ((r = sparsify_slots(ctx, sc, 0, set, first_slot + b0, nb, nullptr))) return r;
		// the scalar record (mag, length, sums, max, 1-mers, stddev, overflow) is the dense slot's
		HIP_TRY(ctx, hipMemcpy2DAsync(set->scalars + (first_slot + b0) * set->scalar_stride, set->scalar_stride, sc->scalars, sc->scalar_stride,
		                              sizeof(MscSlotScalars), nb, hipMemcpyDeviceToDevice, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	return refresh_bounds(ctx, set, first_slot, n_seqs);
}

extern "C" int msc_hist_build_packed(msc_ctx* ctx, msc_hist_set* set, uint64_t first_slot, uint64_t n_seqs, const uint8_t* packed,
                                     uint64_t n_bases, const uint32_t* seg_seq, const uint64_t* seg_start, const uint64_t* seg_end,
                                     uint64_t n_segs, const uint64_t* eff_len, const uint64_t* one_mers) {
	if (!ctx || !set || set->ctx != ctx) return MSC_ERR_INVALID_ARG;
	if (first_slot + n_seqs > set->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "slots %llu..%llu exceed capacity %llu",
	                                                     (unsigned long long)first_slot, (unsigned long long)(first_slot + n_seqs), (unsigned long long)set->capacity);
	if (n_seqs == 0) return MSC_OK;
	if ((n_segs && (!seg_seq || !seg_start || !seg_end)) || !eff_len || (n_bases && !packed)) return fail(ctx, MSC_ERR_INVALID_ARG, "NULL input array");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (set->sparse) return build_sparse(ctx, set, first_slot, n_seqs, packed, n_bases, seg_seq, seg_start, seg_end, n_segs, eff_len, one_mers);
	const int k = set->k;
	const MscLayout& L = set->L;

	// k-mer ordinal of each segment's first k-mer (segments shorter than k contribute none, clutil/Loader.cpp:53-54)
	std::vector<uint64_t> koff(n_segs + 1, 0);
	std::vector<uint64_t> kmers_per_seq(n_seqs, 0);
	for (uint64_t j = 0; j < n_segs; j++) {
		if (seg_seq[j] >= n_seqs || seg_end[j] < seg_start[j] || seg_end[j] >= n_bases)
			return fail(ctx, MSC_ERR_INVALID_ARG, "segment %llu is malformed", (unsigned long long)j);
		const uint64_t len = seg_end[j] - seg_start[j] + 1;
		const uint64_t nk = len >= (uint64_t)k ? len - k + 1 : 0;
		koff[j + 1] = koff[j] + nk;
		kmers_per_seq[seg_seq[j]] += nk;
	}
	const uint64_t total_kmers = koff[n_segs];
	uint64_t tmax = set->dtype == 64 ? ~0ull : ((1ull << set->dtype) - 1);
	bool saturating = false;
	for (uint64_t i = 0; i < n_seqs; i++) if (kmers_per_seq[i] >= tmax) saturating = true;

	bool grouped = true;          // segments in ascending sequence order, as msc_hist_build emits them
	for (uint64_t j = 1; j < n_segs; j++) if (seg_seq[j] < seg_seq[j - 1]) grouped = false;
	// large k, every sequence's k-mers fit one workgroup's LDS: sort + single streaming write per slot (hist_build.hip)
	static const bool no_sort = getenv("MSC_NO_SORT_DENSE_BUILD") != nullptr;
	uint64_t longest = 0;
	for (uint64_t i = 0; i < n_seqs; i++) longest = std::max(longest, kmers_per_seq[i]);
	static const bool no_lds = getenv("MSC_NO_LDS_BUILD") != nullptr;
	const bool use_lds = msc_lds_build_supported(L) && grouped && !no_lds;      // small k: the whole histogram in LDS
	const bool use_sort = !use_lds && msc_sort_build_supported(L, k) && grouped && !no_sort && longest <= msc_sort_build_max_kmers() && n_seqs <= 0xffffffffull;

	std::vector<MscSlotScalars> sc;
	if (!use_sort) {
		// scalar records: length, k=1 table, overflow cleared (the sort builder writes whole records itself)
		sc.resize(n_seqs);
		memset(sc.data(), 0, sizeof(MscSlotScalars) * n_seqs);
		for (uint64_t i = 0; i < n_seqs; i++) {
			sc[i].length = eff_len[i];
			for (int b = 0; b < 4; b++) sc[i].one_mers[b] = one_mers ? one_mers[4 * i + b] : 0;
			sc[i].n_kmers = kmers_per_seq[i];
		}
		HIP_TRY(ctx, hipMemcpy2DAsync(set->scalars + first_slot * set->scalar_stride, set->scalar_stride, sc.data(), sizeof(MscSlotScalars),
		                              sizeof(MscSlotScalars), n_seqs, hipMemcpyHostToDevice, ctx->stream));
	}

	// packed stream + 12 zero bytes behind its last whole word (the kernel's 64-bit window never reads past the end); the memset
	// also clears that last word before the copy fills it. n_bases == 0 (every sequence of the batch empty, or nothing left of a
	// soft-masked chunk after the strip) leaves just the 12 zero bytes.
	const size_t packed_bytes = (size_t)((n_bases + 3) / 4);
	const size_t round4 = (packed_bytes + 3) / 4 * 4, padded_bytes = round4 + 12;
	const size_t tail_off = round4 >= 4 ? round4 - 4 : 0;
	int r;
	if ((r = ensure(ctx, ctx->packed, padded_bytes)) != MSC_OK) return r;
	HIP_TRY(ctx, hipMemsetAsync((uint8_t*)ctx->packed.p + tail_off, 0, padded_bytes - tail_off, ctx->stream));
	if (packed_bytes) HIP_TRY(ctx, hipMemcpyAsync(ctx->packed.p, packed, packed_bytes, ctx->packed_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
	if (n_segs) {
		if ((r = ensure(ctx, ctx->seg_seq, n_segs * sizeof(uint32_t))) != MSC_OK) return r;
		if ((r = ensure(ctx, ctx->seg_start, n_segs * sizeof(uint64_t))) != MSC_OK) return r;
		if ((r = ensure(ctx, ctx->kmer_off, (n_segs + 1) * sizeof(uint64_t))) != MSC_OK) return r;
		if (!use_sort) HIP_TRY(ctx, hipMemcpyAsync(ctx->seg_seq.p, seg_seq, n_segs * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->seg_start.p, seg_start, n_segs * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->kmer_off.p, koff.data(), (n_segs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	}
	// small k: one fused LDS pass per sequence (needs the segments grouped by ascending sequence)
	if (use_lds) {
		std::vector<uint64_t> sbeg(n_seqs + 1, 0);
		for (uint64_t j = 0; j < n_segs; j++) sbeg[seg_seq[j] + 1]++;
		for (uint64_t i = 0; i < n_seqs; i++) sbeg[i + 1] += sbeg[i];
		if ((r = ensure(ctx, ctx->seq_seg, (n_seqs + 1) * sizeof(uint64_t))) != MSC_OK) return r;
		if (n_segs == 0) {      // the kernel still reads kmer_off / seg_start pointers only inside empty loops
			if ((r = ensure(ctx, ctx->seg_start, 8)) != MSC_OK) return r;
			if ((r = ensure(ctx, ctx->kmer_off, 8)) != MSC_OK) return r;
		}
		HIP_TRY(ctx, hipMemcpyAsync(ctx->seq_seg.p, sbeg.data(), (n_seqs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, msc_launch_build_lds(ctx->stream, set->bins, set->scalars, L, k, set->dtype, first_slot, n_seqs, (const uint32_t*)ctx->packed.p,
		                                  (const uint64_t*)ctx->seg_start.p, (const uint64_t*)ctx->kmer_off.p, (const uint64_t*)ctx->seq_seg.p));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // sbeg lives on this stack frame
		return refresh_bounds(ctx, set, first_slot, n_seqs);
	}
	// Sequences are launched in classes of LDS footprint so that short ones keep their occupancy next to long ones.
	if (use_sort) {
		std::vector<uint64_t> sbeg(n_seqs + 1, 0), meta(6 * n_seqs);
		for (uint64_t j = 0; j < n_segs; j++) sbeg[seg_seq[j] + 1]++;
		for (uint64_t i = 0; i < n_seqs; i++) {
			sbeg[i + 1] += sbeg[i];
			meta[6 * i] = eff_len[i];
			for (int b = 0; b < 4; b++) meta[6 * i + 1 + b] = one_mers ? one_mers[4 * i + b] : 0;
			meta[6 * i + 5] = kmers_per_seq[i];
		}
		const uint32_t classes[] = {1024, 4096, 16384, 32768};
		std::vector<uint32_t> ids;
		ids.reserve(n_seqs);
		uint64_t class_begin[5] = {0, 0, 0, 0, 0};
		for (int c = 0; c < 4; c++) {
			const uint64_t lo_k = c ? classes[c - 1] : 0;
			for (uint64_t i = 0; i < n_seqs; i++)
				if ((c == 0 ? kmers_per_seq[i] <= classes[0] : (kmers_per_seq[i] > lo_k && kmers_per_seq[i] <= classes[c]))) ids.push_back((uint32_t)i);
			class_begin[c + 1] = ids.size();
		}
		if ((r = ensure(ctx, ctx->seq_seg, (n_seqs + 1) * sizeof(uint64_t))) != MSC_OK) return r;
		if ((r = ensure(ctx, ctx->seq_ids, n_seqs * sizeof(uint32_t))) != MSC_OK) return r;
		if ((r = ensure(ctx, ctx->seq_meta, meta.size() * sizeof(uint64_t) + 16)) != MSC_OK) return r;
		if (n_segs == 0) {
			if ((r = ensure(ctx, ctx->seg_start, 8)) != MSC_OK) return r;
			if ((r = ensure(ctx, ctx->kmer_off, 8)) != MSC_OK) return r;
		}
		uint64_t* d_bounds = (uint64_t*)ctx->seq_meta.p + meta.size();      // two words after the table
		HIP_TRY(ctx, hipMemsetAsync(d_bounds, 0, 16, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->seq_seg.p, sbeg.data(), (n_seqs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->seq_ids.p, ids.data(), n_seqs * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->seq_meta.p, meta.data(), meta.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
		for (int c = 0; c < 4; c++)
			HIP_TRY(ctx, msc_launch_build_sort(ctx->stream, set->bins, set->scalars, L, k, set->dtype, first_slot, (const uint32_t*)ctx->seq_ids.p + class_begin[c],
			                                   class_begin[c + 1] - class_begin[c], classes[c], (const uint32_t*)ctx->packed.p, (const uint64_t*)ctx->seg_start.p,
			                                   (const uint64_t*)ctx->kmer_off.p, (const uint64_t*)ctx->seq_seg.p, (const uint64_t*)ctx->seq_meta.p, d_bounds));
		uint64_t b[2] = {0, 0};
		HIP_TRY(ctx, hipMemcpyAsync(b, d_bounds, 16, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // sbeg, ids, meta and b live on this stack frame
		set->max_count = std::max(set->max_count, b[0]);
		set->max_sum = std::max(set->max_sum, b[1]);
		mark_written(set, first_slot, n_seqs);               // as refresh_bounds: these slots are stale in the mirrors
		for (uint64_t i = 0; i < n_seqs; i++) learn_length(set, first_slot + i, eff_len[i]);
		return MSC_OK;
	}
	HIP_TRY(ctx, msc_launch_fill(ctx->stream, set->bins, L, first_slot, n_seqs));
	HIP_TRY(ctx, msc_launch_count(ctx->stream, set->bins, set->scalars, L, k, set->dtype, first_slot, (const uint32_t*)ctx->packed.p,
	                              (const uint32_t*)ctx->seg_seq.p, (const uint64_t*)ctx->seg_start.p, (const uint64_t*)ctx->kmer_off.p,
	                              n_segs, total_kmers, saturating));
	HIP_TRY(ctx, msc_launch_finalize(ctx->stream, set->bins, set->scalars, L, set->dtype, first_slot, n_seqs, false));
	return refresh_bounds(ctx, set, first_slot, n_seqs);
}

// Host half of Loader::get_point for a batch: encode (a1) + 2-bit packing + the k=1 table, on several host threads. Every
// sequence starts on a byte boundary of the packed stream (up to three filler bases that no segment covers), so the threads'
// byte ranges are disjoint; segments carry global base offsets, which is all the device side looks at.
namespace {
struct EncodedRange {
	std::vector<uint8_t> packed;                     // this range's sequences, each padded to a multiple of four bases
	std::vector<uint32_t> seg_seq;
	std::vector<uint64_t> seg_start, seg_end;       // base offsets relative to the range's first base
	uint64_t n_bases = 0;                            // padded
	int64_t bad = -1;                                // first sequence of the range with a character outside the IUPAC map
};

void encode_range(const char* const* seqs, const uint64_t* lens, int strip, uint64_t i0, uint64_t i1, uint64_t* eff, uint64_t* ones, EncodedRange& out) {
	std::vector<uint8_t> codes;
	std::vector<Seg> segs;
	std::string stripped;
	uint64_t total = 0;
	for (uint64_t i = i0; i < i1; i++) total += (lens[i] + 3) / 4;
	out.packed.assign((size_t)total, 0);
	for (uint64_t i = i0; i < i1; i++) {
		const char* s = seqs[i];
		size_t len = (size_t)lens[i];
		if (strip) {       // Loader<T>::get_point(std::string...) keeps upper-case A/C/G/T only (clutil/Loader.cpp:115-121)
			stripped.clear();
			for (size_t j = 0; j < len; j++) { const char c = s[j]; if (c == 'A' || c == 'C' || c == 'G' || c == 'T') stripped.push_back(c); }
			s = stripped.data();
			len = stripped.size();
		}
		uint64_t e = 0;
		if (!encode_sequence(s, len, codes, segs, e)) { out.bad = (int64_t)i; return; }
		eff[i] = e;
		uint64_t om[4] = {1, 1, 1, 1};   // KmerHashTable<unsigned long,uint64_t> table_k1(1, 1), clutil/Loader.cpp:143
		for (const Seg& sg : segs) {
			for (int64_t p = sg.s; p <= sg.e; p++) om[codes[p] & 3]++;
			out.seg_seq.push_back((uint32_t)i);
			out.seg_start.push_back(out.n_bases + (uint64_t)sg.s);
			out.seg_end.push_back(out.n_bases + (uint64_t)sg.e);
		}
		for (int b = 0; b < 4; b++) ones[4 * i + b] = om[b];
		uint8_t* dst = out.packed.data() + (out.n_bases >> 2);
		size_t p = 0;
		for (; p + 4 <= len; p += 4) dst[p >> 2] = (uint8_t)((codes[p] & 3) | ((codes[p + 1] & 3) << 2) | ((codes[p + 2] & 3) << 4) | ((codes[p + 3] & 3) << 6));
		for (; p < len; p++) dst[p >> 2] |= (uint8_t)((codes[p] & 3) << (2 * (p & 3)));
		out.n_bases += (len + 3) / 4 * 4;
	}
}
}  // namespace

// the same with the 2-bit stream already on this device (a query block another rank sent over xGMI): no host hop for the bases
extern "C" int msc_hist_build_packed_dev(msc_ctx* ctx, msc_hist_set* set, uint64_t first_slot, uint64_t n_seqs, const void* packed_dev,
                                         uint64_t n_bases, const uint32_t* seg_seq, const uint64_t* seg_start, const uint64_t* seg_end,
                                         uint64_t n_segs, const uint64_t* eff_len, const uint64_t* one_mers) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	ctx->packed_on_device = true;
	const int r = msc_hist_build_packed(ctx, set, first_slot, n_seqs, (const uint8_t*)packed_dev, n_bases, seg_seq, seg_start, seg_end, n_segs, eff_len, one_mers);
	ctx->packed_on_device = false;
	return r;
}

extern "C" int msc_hist_build(msc_ctx* ctx, msc_hist_set* set, uint64_t first_slot, uint64_t n_seqs, const char* const* seqs,
                              const uint64_t* lens, int strip) {
	if (!ctx || !set || set->ctx != ctx) return MSC_ERR_INVALID_ARG;
	if (n_seqs && (!seqs || !lens)) return fail(ctx, MSC_ERR_INVALID_ARG, "NULL sequence array");
	if (n_seqs > 0xffffffffull) return fail(ctx, MSC_ERR_INVALID_ARG, "more than 2^32 sequences in one batch");
	std::vector<uint64_t> eff(n_seqs), ones(4 * n_seqs);
	uint64_t chars = 0;
	for (uint64_t i = 0; i < n_seqs; i++) chars += lens[i];
	// ranges of about equal character count, one host thread each (MSC_HOST_THREADS overrides; small batches stay on the caller's thread)
	static const unsigned env_threads = [] { const char* e = getenv("MSC_HOST_THREADS"); return e ? (unsigned)std::max(1, atoi(e)) : 0u; }();
	unsigned nt = env_threads ? env_threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
	if (!env_threads) nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(1, chars >> 18));          // >= 256 K characters per thread
	nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(1, n_seqs));
	std::vector<uint64_t> cut(nt + 1, n_seqs);
	cut[0] = 0;
	{
		uint64_t acc = 0;
		unsigned t = 1;
		for (uint64_t i = 0; i < n_seqs && t < nt; i++) {
			acc += lens[i];
			if (acc >= chars * t / nt) cut[t++] = i + 1;
		}
	}
	std::vector<EncodedRange> parts(nt);
	if (nt == 1) {
		encode_range(seqs, lens, strip, 0, n_seqs, eff.data(), ones.data(), parts[0]);
	} else {
		std::vector<std::thread> th;
		for (unsigned t = 0; t < nt; t++) th.emplace_back(encode_range, seqs, lens, strip, cut[t], cut[t + 1], eff.data(), ones.data(), std::ref(parts[t]));
		for (auto& x : th) x.join();
	}
	for (const EncodedRange& pr : parts)
		if (pr.bad >= 0) return fail(ctx, MSC_ERR_INVALID_INPUT, "sequence %llu holds a character outside the IUPAC nucleotide map", (unsigned long long)pr.bad);
	uint64_t n_bases = 0, n_segs = 0;
	for (const EncodedRange& pr : parts) { n_bases += pr.n_bases; n_segs += pr.seg_seq.size(); }
	std::vector<uint8_t> packed((size_t)(n_bases / 4));
	std::vector<uint32_t> seg_seq(n_segs);
	std::vector<uint64_t> seg_start(n_segs), seg_end(n_segs);
	uint64_t base = 0, so = 0;
	for (const EncodedRange& pr : parts) {
		if (pr.n_bases) memcpy(packed.data() + base / 4, pr.packed.data(), (size_t)(pr.n_bases / 4));      // with `strip` the range buffer is larger than what it holds
		for (size_t q = 0; q < pr.seg_seq.size(); q++) {
			seg_seq[so + q] = pr.seg_seq[q];
			seg_start[so + q] = base + pr.seg_start[q];
			seg_end[so + q] = base + pr.seg_end[q];
		}
		so += pr.seg_seq.size();
		base += pr.n_bases;
	}
	return msc_hist_build_packed(ctx, set, first_slot, n_seqs, packed.data(), n_bases, seg_seq.data(), seg_start.data(), seg_end.data(), n_segs, eff.data(),
	                             ones.data());
}

int check_slot(msc_ctx* ctx, const msc_hist_set* s, uint64_t slot) {
	if (!ctx || !s || s->ctx != ctx) return MSC_ERR_INVALID_ARG;
	if (slot >= s->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "slot %llu out of range (capacity %llu)", (unsigned long long)slot, (unsigned long long)s->capacity);
	return MSC_OK;
}

extern "C" int msc_hist_download(msc_ctx* ctx, const msc_hist_set* set, uint64_t slot, void* bins_out) {
	int r = check_slot(ctx, set, slot);
	if (r) return r;
	if (!bins_out) return MSC_ERR_INVALID_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const MscLayout& L = set->L;
	if (set->sparse) {          // expand on the host: every bin is 1 except the stored entries
		if (L.nbins * L.esz > (4ull << 30)) return fail(ctx, MSC_ERR_UNSUPPORTED, "dense download of a %llu-byte histogram refused", (unsigned long long)(L.nbins * L.esz));
		const MscSparseHdr& h = set->hdr_host[slot];
		std::vector<uint2> e(h.nnz);
		if (h.nnz) HIP_TRY(ctx, hipMemcpy(e.data(), set->ent + h.off, h.nnz * sizeof(uint2), hipMemcpyDeviceToHost));
		for (uint64_t i = 0; i < L.nbins; i++) {
			switch (set->dtype) { case 8: ((uint8_t*)bins_out)[i] = 1; break; case 16: ((uint16_t*)bins_out)[i] = 1; break; case 32: ((uint32_t*)bins_out)[i] = 1; break; default: ((uint64_t*)bins_out)[i] = 1; }
		}
		for (const uint2& x : e) {
			switch (set->dtype) { case 8: ((uint8_t*)bins_out)[x.x] = (uint8_t)x.y; break; case 16: ((uint16_t*)bins_out)[x.x] = (uint16_t)x.y; break; case 32: ((uint32_t*)bins_out)[x.x] = x.y; break; default: ((uint64_t*)bins_out)[x.x] = x.y; }
		}
		return MSC_OK;
	}
	if ((r = ensure(ctx, ctx->nat, L.slot_bytes)) != MSC_OK) return r;
	HIP_TRY(ctx, msc_launch_permute(ctx->stream, set->bins + slot * L.slot_bytes, ctx->nat.p, L, set->dtype, false));
	HIP_TRY(ctx, hipMemcpyAsync(bins_out, ctx->nat.p, L.nbins * L.esz, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MSC_OK;
}

extern "C" int msc_hist_upload(msc_ctx* ctx, msc_hist_set* set, uint64_t slot, const void* bins, uint64_t length, const uint64_t* one_mers) {
	int r = check_slot(ctx, set, slot);
	if (r) return r;
	if (!bins) return MSC_ERR_INVALID_ARG;
	if (set->sparse) return fail(ctx, MSC_ERR_UNSUPPORTED, "msc_hist_upload is not available for sparse sets");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const MscLayout& L = set->L;
	if ((r = ensure(ctx, ctx->nat, L.slot_bytes)) != MSC_OK) return r;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->nat.p, bins, L.nbins * L.esz, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, msc_launch_permute(ctx->stream, ctx->nat.p, set->bins + slot * L.slot_bytes, L, set->dtype, true));
	MscSlotScalars sc;
	memset(&sc, 0, sizeof sc);
	sc.length = length;
	if (one_mers) for (int b = 0; b < 4; b++) sc.one_mers[b] = one_mers[b];
	HIP_TRY(ctx, hipMemcpyAsync(set->scalars + slot * set->scalar_stride, &sc, sizeof sc, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // sc is on the stack
	HIP_TRY(ctx, msc_launch_finalize(ctx->stream, set->bins, set->scalars, L, set->dtype, slot, 1, false));
	return refresh_bounds(ctx, set, slot, 1);
}

extern "C" int msc_hist_info_get(msc_ctx* ctx, const msc_hist_set* set, uint64_t slot, msc_hist_info* out) {
	int r = check_slot(ctx, set, slot);
	if (r) return r;
	if (!out) return MSC_ERR_INVALID_ARG;
	MscSlotScalars sc;
	HIP_TRY(ctx, hipMemcpyAsync(&sc, set->scalars + slot * set->scalar_stride, sizeof sc, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	out->mag = sc.mag; out->length = sc.length; out->sum = sc.sum; out->sum_sq = sc.sum_sq; out->max_count = sc.max_count;
	for (int b = 0; b < 4; b++) out->one_mers[b] = sc.one_mers[b];
	out->stddev = sc.stddev; out->overflow = (int32_t)(sc.overflow != 0); out->pad_ = 0; out->id = sc.id;
	return MSC_OK;
}

// effective length of a slot: from the host-side cache when a writer left it there, else read back once
int slot_length(msc_ctx* ctx, const msc_hist_set* set, uint64_t slot, uint64_t* len) {
	int r = check_slot(ctx, set, slot);
	if (r) return r;
	if (slot < set->len_known.size() && set->len_known[slot]) { *len = set->len_host[slot]; return MSC_OK; }
	msc_hist_info hi;
	if ((r = msc_hist_info_get(ctx, set, slot, &hi))) return r;
	learn_length(set, slot, hi.length);
	*len = hi.length;
	return MSC_OK;
}

extern "C" int msc_hist_lengths(msc_ctx* ctx, const msc_hist_set* set, uint64_t first_slot, uint64_t n, uint64_t* lengths_out) {
	if (!ctx || !set || set->ctx != ctx || first_slot + n > set->capacity || (n && !lengths_out)) return MSC_ERR_INVALID_ARG;
	bool all_known = true;
	for (uint64_t i = first_slot; i < first_slot + n && all_known; i++) all_known = i < set->len_known.size() && set->len_known[i];
	if (!all_known && n) {          // one strided copy of the length words, remembered for the operators' window arithmetic
		HIP_TRY(ctx, hipSetDevice(ctx->device));
		HIP_TRY(ctx, hipMemcpy2DAsync(lengths_out, 8, set->scalars + first_slot * set->scalar_stride + offsetof(MscSlotScalars, length), set->scalar_stride, 8, n,
		                              hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		for (uint64_t i = 0; i < n; i++) learn_length(set, first_slot + i, lengths_out[i]);
		return MSC_OK;
	}
	for (uint64_t i = 0; i < n; i++) lengths_out[i] = set->len_host[first_slot + i];
	return MSC_OK;
}

extern "C" int msc_hist_set_id(msc_ctx* ctx, msc_hist_set* set, uint64_t slot, uint64_t id) {
	int r = check_slot(ctx, set, slot);
	if (r) return r;
	HIP_TRY(ctx, hipMemcpyAsync(set->scalars + slot * set->scalar_stride + offsetof(MscSlotScalars, id), &id, sizeof id, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MSC_OK;
}

static int copy_common(msc_ctx* ctx, msc_hist_set* dst, uint64_t ds, const msc_hist_set* src, uint64_t ss) {
	int r = check_slot(ctx, dst, ds);
	if (r) return r;
	if ((r = check_slot(ctx, src, ss))) return r;
	if (dst->k != src->k || dst->dtype != src->dtype || dst->sparse != src->sparse) return fail(ctx, MSC_ERR_INVALID_ARG, "sets differ in k, dtype or layout");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (src->sparse) {          // the bins of a sparse slot are its entry list: append a copy to dst's arena
		const MscSparseHdr sh = src->hdr_host[ss];
		if (dst->ent_used + sh.nnz > dst->ent_capacity) return fail(ctx, MSC_ERR_OOM, "sparse set entry arena exhausted");
		MscSparseHdr dh = sh;
		dh.off = dst->ent_used;
		dst->ent_used += sh.nnz;
		if (sh.nnz) {
			HIP_TRY(ctx, hipMemcpyAsync(dst->ent + dh.off, src->ent + sh.off, sh.nnz * sizeof(uint2), hipMemcpyDeviceToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(dst->cum + dh.off, src->cum + sh.off, sh.nnz * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
		}
		dst->hdr_host[ds] = dh;
		dst->list_epoch++;
		HIP_TRY(ctx, hipMemcpyAsync(dst->hdr + ds, &dst->hdr_host[ds], sizeof(MscSparseHdr), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		return MSC_OK;
	}
	HIP_TRY(ctx, hipMemcpyAsync(dst->bins + ds * dst->L.slot_bytes, src->bins + ss * src->L.slot_bytes, src->L.slot_bytes, hipMemcpyDeviceToDevice, ctx->stream));
	return MSC_OK;
}

// slot-to-slot copies inside the library: nothing copied can exceed the source set's own maxima, so the bounds merge on the
// host and the stream is not drained (the accumulate loop clones one centre per step)
static int inherit_bounds(msc_hist_set* dst, uint64_t ds, const msc_hist_set* src, uint64_t ss) {
	dst->max_count = std::max(dst->max_count, src->max_count);
	dst->max_sum = std::max(dst->max_sum, src->max_sum);
	if (dst->sparse) dst->max_nnz = std::max(dst->max_nnz, src->hdr_host[ss].nnz);
	mark_written(dst, ds, 1);
	if (ss < src->len_known.size() && src->len_known[ss]) learn_length(dst, ds, src->len_host[ss]);      // clone, set and copy all carry the length over
	return MSC_OK;
}

extern "C" int msc_hist_clone(msc_ctx* ctx, msc_hist_set* dst, uint64_t ds, const msc_hist_set* src, uint64_t ss) {
	int r = copy_common(ctx, dst, ds, src, ss);
	if (r) return r;
	uint8_t* d = dst->scalars + ds * dst->scalar_stride;
	const uint8_t* s = src->scalars + ss * src->scalar_stride;
	HIP_TRY(ctx, hipMemcpyAsync(d, s, dst->scalar_stride, hipMemcpyDeviceToDevice, ctx->stream));
	// the (pts, len) ctor re-sums mag from the bins (clutil/DivergencePoint.cpp:99-110)
	HIP_TRY(ctx, hipMemcpyAsync(d + offsetof(MscSlotScalars, mag), s + offsetof(MscSlotScalars, sum), 8, hipMemcpyDeviceToDevice, ctx->stream));
	return inherit_bounds(dst, ds, src, ss);
}

extern "C" int msc_hist_copy(msc_ctx* ctx, msc_hist_set* dst, uint64_t ds, const msc_hist_set* src, uint64_t ss) {
	int r = copy_common(ctx, dst, ds, src, ss);
	if (r) return r;
	HIP_TRY(ctx, hipMemcpyAsync(dst->scalars + ds * dst->scalar_stride, src->scalars + ss * src->scalar_stride, dst->scalar_stride, hipMemcpyDeviceToDevice, ctx->stream));
	return inherit_bounds(dst, ds, src, ss);
}

extern "C" int msc_hist_assign(msc_ctx* ctx, msc_hist_set* dst, uint64_t ds, const msc_hist_set* src, uint64_t ss) {
	int r = copy_common(ctx, dst, ds, src, ss);
	if (r) return r;
	uint8_t* d = dst->scalars + ds * dst->scalar_stride;
	const uint8_t* s = src->scalars + ss * src->scalar_stride;
	// points, length, id -- NOT mag, NOT stddev (clutil/DivergencePoint.cpp:182-190); the derived sums follow the bins
	const size_t a0 = offsetof(MscSlotScalars, length), a1 = offsetof(MscSlotScalars, one_mers);
	HIP_TRY(ctx, hipMemcpyAsync(d + a0, s + a0, a1 - a0, hipMemcpyDeviceToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(d + offsetof(MscSlotScalars, id), s + offsetof(MscSlotScalars, id), 8, hipMemcpyDeviceToDevice, ctx->stream));
	if (!dst->sparse) HIP_TRY(ctx, hipMemcpyAsync(d + sizeof(MscSlotScalars), s + sizeof(MscSlotScalars), 8ull * dst->L.S, hipMemcpyDeviceToDevice, ctx->stream));
	return inherit_bounds(dst, ds, src, ss);
}

// center->set(*next) for many centres at once (the tail of every mean_shift_update of a round): same field semantics as
// msc_hist_assign; destination slots must be distinct.
// center->set(*next) (exact = 0), an exact copy (1: every word of the record, the stale magnitude included) or a clone (2: an exact
// copy with the magnitude re-summed) of n slots in one launch per region
static int assign_or_copy_batch(msc_ctx* ctx, msc_hist_set* dst, const uint32_t* dst_slots, const msc_hist_set* src, const uint32_t* src_slots, uint64_t n, int exact) {
	if (!ctx || !dst || !src || dst->ctx != ctx || src->ctx != ctx) return MSC_ERR_INVALID_ARG;
	if (n == 0) return MSC_OK;
	if (!dst_slots || !src_slots) return MSC_ERR_INVALID_ARG;
	if (dst->k != src->k || dst->dtype != src->dtype || dst->sparse != src->sparse) return fail(ctx, MSC_ERR_INVALID_ARG, "sets differ in k, dtype or layout");
	uint32_t lo = ~0u, hi = 0;
	for (uint64_t i = 0; i < n; i++) {
		if (dst_slots[i] >= dst->capacity || src_slots[i] >= src->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "slot out of range");
		lo = std::min(lo, dst_slots[i]);
		hi = std::max(hi, dst_slots[i]);
	}
	if (n > 0x7fffffffull || (dst->sparse && src->scalar_stride != dst->scalar_stride)) {
		for (uint64_t i = 0; i < n; i++) {
			const int r = exact == 2 ? msc_hist_clone(ctx, dst, dst_slots[i], src, src_slots[i]) : exact ? msc_hist_copy(ctx, dst, dst_slots[i], src, src_slots[i])
			                                                                                             : msc_hist_assign(ctx, dst, dst_slots[i], src, src_slots[i]);
			if (r) return r;
		}
		return MSC_OK;
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int r;
	if (dst->sparse) {
		// every list is appended to the destination arena in one launch; all or nothing, so that a caller can compact and retry
		uint64_t need = 0;
		for (uint64_t i = 0; i < n; i++) need += src->hdr_host[src_slots[i]].nnz;
		if (dst->ent_used + need > dst->ent_capacity)
			return fail(ctx, MSC_ERR_OOM, "sparse set entry arena exhausted (%llu of %llu entries used, %llu needed)", (unsigned long long)dst->ent_used,
			            (unsigned long long)dst->ent_capacity, (unsigned long long)need);
		std::vector<uint64_t> off(n);
		uint64_t o = dst->ent_used;
		for (uint64_t i = 0; i < n; i++) { off[i] = o; o += src->hdr_host[src_slots[i]].nnz; }
		if ((r = ensure(ctx, ctx->slots, n * sizeof(uint32_t))) || (r = ensure(ctx, ctx->pair_seg, n * sizeof(uint32_t))) || (r = ensure(ctx, ctx->sp_chunk_off, n * sizeof(uint64_t)))) return r;
		HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, dst_slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->pair_seg.p, src_slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_chunk_off.p, off.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, msc_launch_sparse_assign_batch(ctx->stream, dst->ent, dst->cum, dst->hdr, src->ent, src->cum, src->hdr, (const uint32_t*)ctx->slots.p,
		                                            (const uint32_t*)ctx->pair_seg.p, (const uint64_t*)ctx->sp_chunk_off.p, (uint32_t)n));
		HIP_TRY(ctx, msc_launch_assign_scalars(ctx->stream, dst->scalars, src->scalars, dst->scalar_stride, (const uint32_t*)ctx->slots.p, (const uint32_t*)ctx->pair_seg.p, (uint32_t)n,
		                                       exact));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // off and the caller's slot arrays may go away
		for (uint64_t i = 0; i < n; i++) {
			MscSparseHdr h = src->hdr_host[src_slots[i]];
			h.off = off[i];
			dst->hdr_host[dst_slots[i]] = h;
			dst->max_nnz = std::max(dst->max_nnz, h.nnz);
			if (src_slots[i] < src->len_known.size() && src->len_known[src_slots[i]]) learn_length(dst, dst_slots[i], src->len_host[src_slots[i]]);
			else forget_lengths(dst, dst_slots[i], 1);
		}
		dst->ent_used = o;
		dst->list_epoch++;
		dst->max_count = std::max(dst->max_count, src->max_count);
		dst->max_sum = std::max(dst->max_sum, src->max_sum);
		return MSC_OK;
	}
	if ((r = ensure(ctx, ctx->slots, n * sizeof(uint32_t))) || (r = ensure(ctx, ctx->pair_seg, n * sizeof(uint32_t)))) return r;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, dst_slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->pair_seg.p, src_slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, msc_launch_assign_batch(ctx->stream, dst->L, dst->bins, dst->scalars, src->bins, src->scalars, (const uint32_t*)ctx->slots.p,
	                                     (const uint32_t*)ctx->pair_seg.p, (uint32_t)n, exact));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // the caller's slot arrays may go away
	// host-side bounds: nothing copied can exceed the source set's own maxima
	dst->max_count = std::max(dst->max_count, src->max_count);
	dst->max_sum = std::max(dst->max_sum, src->max_sum);
	if (dst->written.size() < dst->capacity) dst->written.resize(dst->capacity, 0);
	for (uint64_t i = 0; i < n; i++) {
		dst->written[dst_slots[i]] = 1;
		if (src_slots[i] < src->len_known.size() && src->len_known[src_slots[i]]) learn_length(dst, dst_slots[i], src->len_host[src_slots[i]]);
		else forget_lengths(dst, dst_slots[i], 1);
	}
	mark_stale(dst, lo, (uint64_t)hi + 1 - lo);
	return MSC_OK;
}

extern "C" int msc_hist_assign_batch(msc_ctx* ctx, msc_hist_set* dst, const uint32_t* dst_slots, const msc_hist_set* src, const uint32_t* src_slots, uint64_t n) {
	return assign_or_copy_batch(ctx, dst, dst_slots, src, src_slots, n, 0);
}
extern "C" int msc_hist_copy_batch(msc_ctx* ctx, msc_hist_set* dst, const uint32_t* dst_slots, const msc_hist_set* src, const uint32_t* src_slots, uint64_t n) {
	return assign_or_copy_batch(ctx, dst, dst_slots, src, src_slots, n, 1);
}
extern "C" int msc_hist_clone_batch(msc_ctx* ctx, msc_hist_set* dst, const uint32_t* dst_slots, const msc_hist_set* src, const uint32_t* src_slots, uint64_t n) {
	return assign_or_copy_batch(ctx, dst, dst_slots, src, src_slots, n, 2);
}

extern "C" int msc_hist_set_device_view(const msc_hist_set* set, void** bins, uint64_t* slot_bytes, void** scalars, uint64_t* scalar_bytes) {
	if (!set || set->sparse) return MSC_ERR_INVALID_ARG;
	if (bins) *bins = set->bins;
	if (slot_bytes) *slot_bytes = set->L.slot_bytes;
	if (scalars) *scalars = set->scalars;
	if (scalar_bytes) *scalar_bytes = set->scalar_stride;
	return MSC_OK;
}

extern "C" int msc_hist_import_done(msc_ctx* ctx, msc_hist_set* set, uint64_t first_slot, uint64_t n) {
	if (!ctx || !set || set->ctx != ctx || first_slot + n > set->capacity) return MSC_ERR_INVALID_ARG;
	if (n == 0) return MSC_OK;
	return refresh_bounds(ctx, set, first_slot, n);
}

// ================================================================================================ model
int msc_feat_is_sim(uint64_t f) {      // Feature<T>::feat_is_sim, predict/Feature.cpp:549-663
	switch (f) {
	case MSC_FEAT_NORMALIZED_VECTORS: case MSC_FEAT_PEARSON_COEFF: case MSC_FEAT_INTERSECTION: case MSC_FEAT_KULCZYNSKI2: case MSC_FEAT_SIMRATIO:
	case MSC_FEAT_SIM_MM:
		return 1;
	case MSC_FEAT_MANHATTAN: case MSC_FEAT_EUCLIDEAN: case MSC_FEAT_EMD: case MSC_FEAT_LENGTHD: case MSC_FEAT_JEFFEREY_DIV:
	case MSC_FEAT_JENSEN_SHANNON: case MSC_FEAT_RRE_K_R:
		return 0;
	default:
		return -1;     // the other 23 statistics of predict/Feature.h are `extraslow` only: out of scope
	}
}

static int model_index_of(const MscDevModel& m, uint64_t f) {
	for (int i = 0; i < m.n_singles; i++) if (m.single_flag[i] == f) return i;
	return -1;
}

// The f32 image behind the close-flag screen (pair_features.hip, screen_close): every single statistic must be one of the nine integer
// reductions' functions, every constant finite in f32, the bias 0 (the screen's threshold is s >= 0, GLM::logistic at 0.5).
static void model_screen_image(MscDevModel& h) {
	h.screen_ok = 0;
	if (h.bias != 0.0 || h.n_singles < 1 || h.n_combos < 1) return;
	for (int i = 0; i < h.n_singles; i++) {
		if (!(h.single_flag[i] & MSC_FEAT_FAST)) return;
		const double range = h.maxs[i] - h.mins[i];
		const float mn = (float)h.mins[i], inv = (float)(1.0 / range);
		if (!(range != 0.0) || !std::isfinite(mn) || !std::isfinite(inv) || inv == 0.f) return;
		h.s_min[i] = mn;
		h.s_inv[i] = inv;
	}
	for (int c = 0; c <= h.n_combos; c++) {
		h.s_w[c] = (float)h.weights[c];
		if (!std::isfinite(h.s_w[c])) return;
	}
	h.screen_ok = 1;
}

extern "C" int msc_model_create(msc_ctx* ctx, int k, int n_combos, const int* combo_kind, const uint64_t* combo_flags, const double* weights,
                                int n_singles, const uint64_t* single_flags, const double* mins, const double* maxs, double bias,
                                msc_model** out) {
	if (!ctx || !out) return MSC_ERR_INVALID_ARG;
	*out = nullptr;
	if (n_combos < 0 || n_combos > MSC_MAX_COMBOS) return fail(ctx, MSC_ERR_UNSUPPORTED, "n_combos %d exceeds %d", n_combos, MSC_MAX_COMBOS);
	if ((n_combos && (!combo_kind || !combo_flags)) || !weights || (n_singles && (!single_flags || !mins || !maxs))) return fail(ctx, MSC_ERR_INVALID_ARG, "NULL model array");
	MscDevModel h;
	memset(&h, 0, sizeof h);
	h.bias = bias;
	h.weights[0] = weights[0];
	// replay Feature::add_feature (predict/Feature.cpp:102-128): singles in ascending-bit order of first appearance
	for (int c = 0; c < n_combos; c++) {
		if (combo_kind[c] < 0 || combo_kind[c] > 3) return fail(ctx, MSC_ERR_INVALID_ARG, "combo %d has kind %d", c, combo_kind[c]);
		h.combo_kind[c] = combo_kind[c];
		h.weights[c + 1] = weights[c + 1];
		int n = 0;
		for (uint64_t f = 1; f != 0 && f <= combo_flags[c]; f <<= 1) {
			if (!(combo_flags[c] & f)) continue;
			if (model_index_of(h, f) < 0) {
				const int sim = msc_feat_is_sim(f);
				if (sim < 0) return fail(ctx, MSC_ERR_UNSUPPORTED, "single feature 2^%d is not supported by the GPU path", (int)log2((double)f));
				if (h.n_singles >= MSC_MAX_SINGLES) return fail(ctx, MSC_ERR_UNSUPPORTED, "too many single features");
				const int i = h.n_singles++;
				h.single_flag[i] = f;
				h.mins[i] = DBL_MAX;
				h.maxs[i] = DBL_MIN;
				h.is_sim[i] = sim;
			}
			if (n >= 2) return fail(ctx, MSC_ERR_UNSUPPORTED, "combo %d joins more than two single features", c);
			h.combo_idx[c][n++] = model_index_of(h, f);
		}
		if (n == 0) return fail(ctx, MSC_ERR_INVALID_ARG, "combo %d has no feature bits", c);
		if (n == 1 && (combo_kind[c] == MSC_COMBO_XY2 || combo_kind[c] == MSC_COMBO_X2Y))
			return fail(ctx, MSC_ERR_INVALID_ARG, "combo %d: xy2/x2y need two features (Feature.h:220-233 throws)", c);
		h.combo_n[c] = n;
	}
	h.n_combos = n_combos;
	for (int i = 0; i < n_singles; i++) {       // Feature::set_normal, predict/Feature.cpp:173-180
		const int idx = model_index_of(h, single_flags[i]);
		if (idx < 0) return fail(ctx, MSC_ERR_INVALID_ARG, "n_singles line for feature %llu that no combo uses", (unsigned long long)single_flags[i]);
		h.mins[idx] = mins[i];
		h.maxs[idx] = maxs[i];
	}
	model_screen_image(h);
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	msc_model* m = new msc_model();
	m->ctx = ctx;
	m->k = k;
	m->h = h;
	hipError_t e = hipMalloc((void**)&m->d, sizeof(MscDevModel));
	if (e == hipSuccess) e = hipMemcpy(m->d, &m->h, sizeof(MscDevModel), hipMemcpyHostToDevice);
	if (e != hipSuccess) { delete m; return fail(ctx, MSC_ERR_HIP, "model upload: %s", hipGetErrorString(e)); }
	*out = m;
	return MSC_OK;
}

// Predictor<T>::Predictor(filename) + read_from, predict/Predictor.cpp:47-79,125-185 (`in >> token` semantics)
extern "C" int msc_model_parse(msc_ctx* ctx, const char* text, int block, msc_model** out) {
	if (!ctx || !text || !out) return MSC_ERR_INVALID_ARG;
	std::istringstream in(text);
	std::string buf, datatype;
	int k = 0, max_feat = 0;
	unsigned mode = 0;
	double id = 0;
	uint64_t feats = 0;
	in >> buf >> k >> buf >> mode >> buf >> max_feat >> buf >> id >> buf >> datatype >> buf >> feats;
	if (!in) return fail(ctx, MSC_ERR_IO, "weights file: malformed header");
	const bool want_reg = block == 1;
	if (want_reg && !(mode & 2)) return fail(ctx, MSC_ERR_INVALID_ARG, "weights file has no regression block (mode %u)", mode);
	if (!want_reg && !(mode & 1)) return fail(ctx, MSC_ERR_INVALID_ARG, "weights file has no classification block (mode %u)", mode);
	const int skip = (want_reg && (mode & 1)) ? 1 : 0;
	for (int b = 0; b <= skip; b++) {
		int nc = 0, ns = 0;
		in >> buf >> nc;
		if (!in || nc < 0 || nc > 64) return fail(ctx, MSC_ERR_IO, "weights file: bad n_combos");
		std::vector<int> kinds(nc);
		std::vector<uint64_t> flags(nc);
		std::vector<double> w(nc + 1);
		in >> w[0];
		for (int i = 0; i < nc; i++) in >> kinds[i] >> flags[i] >> w[i + 1];
		in >> buf >> ns;
		if (!in || ns < 0 || ns > 64) return fail(ctx, MSC_ERR_IO, "weights file: bad n_singles");
		std::vector<uint64_t> sf(ns);
		std::vector<double> mn(ns), mx(ns);
		for (int i = 0; i < ns; i++) in >> sf[i] >> mn[i] >> mx[i];
		if (!in) return fail(ctx, MSC_ERR_IO, "weights file: truncated block");
		if (b == skip) return msc_model_create(ctx, k, nc, kinds.data(), flags.data(), w.data(), ns, sf.data(), mn.data(), mx.data(), 0.0, out);
	}
	return MSC_ERR_IO;
}

extern "C" int msc_model_load(msc_ctx* ctx, const char* path, int block, msc_model** out) {
	if (!ctx || !path || !out) return MSC_ERR_INVALID_ARG;
	std::ifstream f(path);
	if (!f) return fail(ctx, MSC_ERR_IO, "cannot open %s", path);
	std::stringstream ss;
	ss << f.rdbuf();
	return msc_model_parse(ctx, ss.str().c_str(), block, out);
}

extern "C" void msc_model_destroy(msc_model* m) {
	if (!m) return;
	if (m->d) (void)hipFree(m->d);
	delete m;
}
extern "C" int msc_model_k(const msc_model* m) { return m ? m->k : 0; }
extern "C" int msc_model_n_singles(const msc_model* m) { return m ? m->h.n_singles : 0; }
extern "C" int msc_model_n_combos(const msc_model* m) { return m ? m->h.n_combos : 0; }
extern "C" int msc_model_single_flags(const msc_model* m, uint64_t* out) {
	if (!m || !out) return MSC_ERR_INVALID_ARG;
	for (int i = 0; i < m->h.n_singles; i++) out[i] = m->h.single_flag[i];
	return MSC_OK;
}
extern "C" void msc_model_set_bias(msc_model* m, double bias) {
	if (!m) return;
	m->h.bias = bias;
	model_screen_image(m->h);
	(void)hipMemcpy(m->d, &m->h, sizeof(MscDevModel), hipMemcpyHostToDevice);
}

