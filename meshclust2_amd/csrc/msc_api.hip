// msc_api.hip -- the C ABI of libmeshclust2_hip.so (include/meshclust2_hip.h) and the host logic behind it:
// sequence encoding, 2-bit packing, weights-file parsing, launch sequencing, result read-back.
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

static thread_local std::string g_create_error;

// largest bin for which 32-bit per-lane partial sums of p*q cannot overflow: R * max^2 < 2^32 with R <= 64
static const uint64_t kNarrowMaxCount = 8191;
static const uint64_t kNarrowMaxSum = (1ull << 31) - 1;

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
static const bool g_profile_calls = getenv("MSC_PROFILE_CALLS") != nullptr;
// from how many bins on the sparse mean sweeps only the 64-byte lines its members touched (MSC_SPARSE_MEAN_GROUPS_MIN_K for A/B runs)
static uint64_t msc_sparse_groups_min_bins() {
	static const uint64_t v = [] { const char* e = getenv("MSC_SPARSE_MEAN_GROUPS_MIN_K"); const int k = e ? atoi(e) : 11; return 1ull << (2 * std::max(5, std::min(16, k))); }();
	return v;
}
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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

static int check_slot(msc_ctx* ctx, const msc_hist_set* s, uint64_t slot) {
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

// ================================================================================================ scoring driver
namespace {


// integer range of the fast streaming kernels (pair_features.hip header); outside it the 64-bit kernel runs
bool needs_wide_impl(const msc_hist_set* a, const msc_hist_set* b) {
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
enum SparseKernel { SPK_LDS = 0, SPK_MP = 1, SPK_GENERIC = 2 };
SparseKernel pick_sparse_kernel(const msc_hist_set* c_sp, const msc_hist_set* q_sp, uint64_t q_slot, uint64_t max_count, bool wide) {
	static const bool want_lds = getenv("MSC_SPARSE_LDS") != nullptr;
	static const bool no_mp = getenv("MSC_SPARSE_NO_MP") != nullptr;
	const uint64_t q_nnz = q_sp->hdr_host[q_slot].nnz;
	if (want_lds && !wide && max_count < 65536 && c_sp->L.nbins >= 64 && ((size_t)(q_nnz + 128) + 4ull * (c_sp->max_nnz + 128)) * 8 <= 96 * 1024) return SPK_LDS;
	if (!no_mp && !wide && max_count < 65536 && q_nnz + c_sp->max_nnz <= msc_sparse_mp_max_entries()) return SPK_MP;
	return SPK_GENERIC;
}
uint32_t sparse_records(SparseKernel k, uint32_t mp_parts = 1) { return k == SPK_GENERIC ? MSC_SPARSE_SUB : k == SPK_MP ? mp_parts : 1; }
// {jd, js} records per pair: the merge-path kernel leaves one per granule of the merged order (lists of up to `entries` together)
uint32_t div_records(SparseKernel k, uint64_t entries) { return k == SPK_GENERIC ? MSC_SPARSE_SUB : k == SPK_MP ? msc_sparse_mp_div_records(entries) : 1; }
const char* sparse_kernel_name(SparseKernel k) { return k == SPK_LDS ? "k_pair_sparse_lds" : k == SPK_MP ? "k_pair_sparse_mp" : "k_pair_sparse"; }

// candidates [off, off + mc) (or the device slot list d_slots) of the sparse set / mirror c_sp against slot q_slot of q_sp; the
// scalar records are those of the sets the lists belong to (a mirror has none of its own)
hipError_t launch_sparse_pass(msc_ctx* ctx, SparseKernel k, const msc_hist_set* c_sp, const uint8_t* c_scalars, uint64_t c_stride, const uint32_t* d_slots,
                              uint64_t off, uint32_t mc, const msc_hist_set* q_sp, uint64_t q_slot, const uint8_t* q_scal, uint64_t nbins, int use_window,
                              uint64_t min_len, uint64_t max_len, MscPartial* partials, void* div_tables, void* div_partials, int order, uint32_t parts = 1,
                              uint32_t div_stride = 1) {
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
bool rank_lists_ready(msc_ctx* ctx, const msc_hist_set* s, int* err, bool eager = false) {
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
	auto give_up = [&]() { (void)hipGetLastError(); s->rkl_unavailable = true; return false; };
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
}  // namespace
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
	if (rq.cand_slots) {
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
			                                         ctx->rk_items.p, (uint32_t*)ctx->rk_counters.p, (uint32_t*)ctx->rk_tables.p, ctx->rk_turn++,
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
				HIP_TRY(ctx, msc_launch_epilogue_reduce(ctx->stream, ea, rq.reduce_mode, rq.reduce_begin, d_flags, (MscReduceOut*)down, ctx->reduce_parts.p, rq.close_list));
				if (rq.after_reduce && !rq.close_list.pos) HIP_TRY(ctx, rq.after_reduce((const MscReduceOut*)down));
			} else {
				HIP_TRY(ctx, msc_launch_reduce(ctx->stream, (const MscPairOut*)ctx->pair_out.p, mc, rq.reduce_mode, rq.reduce_begin, d_flags, (MscReduceOut*)down, ctx->reduce_parts.p));
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
		if (rq.reduce_mode >= 0) {
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

namespace {
const uint64_t kSupportedFeats = MSC_FEAT_SLOW | MSC_FEAT_GROUPS;


double trainer_get_id(double cutoff) { return cutoff > 1 ? cutoff / 100.0 : cutoff; }      // cluster/Trainer.h:35

}  // namespace

static int run_score_fwd(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, uint64_t m, const msc_hist_set* rs) {
	ScoreRequest rq;
	rq.cands = set; rq.cand_slots = member_slots; rq.m = m; rq.qset = rs; rq.q_slot = 0; rq.only_tiles = true;
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
	if (member_slots) {
		if ((r = ensure(ctx, ctx->slots, m * sizeof(uint32_t)))) return r;
		HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, member_slots, m * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	}
	const uint32_t* d_slots = member_slots ? (const uint32_t*)ctx->slots.p : nullptr;
	const uint32_t n_chunks = (uint32_t)std::min<uint64_t>(1024, L.nbins / 256);      // multiple of 16 for every k >= 6
	const uint64_t chunk_bins = L.nbins / n_chunks;
	if ((r = ensure(ctx, ctx->sp_counts, std::max<size_t>(n_chunks * 3 * sizeof(uint64_t), ctx->sp_counts.cap)))) return r;
	if ((r = ensure(ctx, ctx->sp_chunk_off, n_chunks * sizeof(uint64_t)))) return r;
	if ((r = ensure(ctx, ctx->sp_chunk_cum, n_chunks * sizeof(uint64_t)))) return r;
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
		                                                (uint64_t*)ctx->sp_counts.p, (const uint32_t*)ctx->sp_touched.p));
	} else {
	HIP_TRY(ctx, msc_launch_sparse_scatter(ctx->stream, set->ent, set->hdr, d_slots, (uint32_t)m, (uint32_t*)ctx->sp_acc.p));
	HIP_TRY(ctx, msc_launch_sparse_mean_count(ctx->stream, set->dtype, (const uint32_t*)ctx->sp_acc.p, n_chunks, chunk_bins, (uint32_t)m, (uint64_t*)ctx->sp_counts.p));
	}
	std::vector<uint64_t> counts(n_chunks * 3), off(n_chunks), cb(n_chunks);
	HIP_TRY(ctx, hipMemcpyAsync(counts.data(), ctx->sp_counts.p, counts.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
	const uint64_t floor_sum = L.nbins + fl;
	if ((r = ensure(ctx, ctx->floor_sum, 8))) return r;
	HIP_TRY(ctx, hipMemcpyAsync(rs->hdr, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(rs->scalars, &sc, sizeof sc, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->floor_sum.p, &floor_sum, 8, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_chunk_off.p, off.data(), n_chunks * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_chunk_cum.p, cb.data(), n_chunks * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
	if (grouped)
		HIP_TRY(ctx, msc_launch_sparse_mean_write_batch(ctx->stream, set->dtype, (uint32_t*)ctx->sp_acc.p, L.nbins, n_chunks, chunk_bins, 1, (const uint32_t*)ctx->qslots.p,
		                                                (const uint64_t*)ctx->sp_chunk_off.p, (const uint64_t*)ctx->sp_chunk_cum.p, rs->ent, rs->cum, (uint32_t*)ctx->sp_touched.p));
	else
	HIP_TRY(ctx, msc_launch_sparse_mean_write(ctx->stream, set->dtype, (uint32_t*)ctx->sp_acc.p, n_chunks, chunk_bins, (uint32_t)m, (const uint64_t*)ctx->sp_chunk_off.p,
	                                          (const uint64_t*)ctx->sp_chunk_cum.p, rs->ent, rs->cum));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // h, sc, floor_sum, off, cb live on this frame
	// members vs the rounded mean: only the |p - r| reduction of the merge kernel is used
	if ((r = run_score_fwd(ctx, set, member_slots, m, rs))) return r;
	if ((r = ensure(ctx, ctx->reduce_out, sizeof(MscReduceOut)))) return r;
	if ((r = ensure(ctx, ctx->raw, m * sizeof(double)))) return r;
	HIP_TRY(ctx, msc_launch_distance_d(ctx->stream, (const MscPartial*)ctx->partials.p, ctx->last_partial_stride, (uint32_t)m, set->scalars, set->scalar_stride, d_slots,
	                                   rs->scalars, (const uint64_t*)ctx->floor_sum.p, (double*)ctx->raw.p, (MscReduceOut*)ctx->reduce_out.p));
	MscReduceOut ro;
	HIP_TRY(ctx, hipMemcpyAsync(&ro, ctx->reduce_out.p, sizeof ro, hipMemcpyDeviceToHost, ctx->stream));
	if (dist_out) HIP_TRY(ctx, hipMemcpyAsync(dist_out, ctx->raw.p, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	*nearest_pos = ro.best_pos;
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
int sparse_acc_prepare(msc_ctx* ctx, const MscLayout& L, uint32_t nc, uint32_t** touched_out) {
	int r;
	const uint32_t n_chunks = (uint32_t)std::min<uint64_t>(1024, L.nbins / 256);
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
	uint32_t n_chunks = (uint32_t)std::min<uint64_t>(1024, L.nbins / 256);      // a multiple of 16 for every sparse-capable k
	while (n_chunks > MSC_SPARSE_SUB && (uint64_t)n_chunks * nc > 65536) n_chunks /= 2;
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
	std::vector<uint8_t> keep;
	std::vector<double> dist;
	for (uint64_t c0 = 0; c0 < n_centres;) {
		uint64_t c1 = c0;
		while (c1 < n_centres && c1 - c0 < max_chunk_centres && (c1 == c0 || offsets[c1 + 1] - offsets[c0] <= max_chunk_pairs)) c1++;
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
		// ---- 2. survivors per centre
		members.clear();
		where.clear();
		pair_seg.clear();
		uint32_t max_m2 = 0;
		for (uint64_t c = c0; c < c1; c++) {
			MscBatchSeg& sg = segs[c - c0];
			const uint32_t first_old = sg.first, m_old = sg.m;
			sg.q_slot = (uint32_t)(c - c0);               // slot of this centre's rounded mean in the scratch set
			sg.first = (uint32_t)members.size();
			for (uint32_t i = 0; i < m_old; i++)
				if (keep[first_old + i]) { members.push_back(pt_slots[base + first_old + i]); where.push_back(i); pair_seg.push_back((uint32_t)(c - c0)); }
			sg.m = (uint32_t)members.size() - sg.first;
			sg.min_len = 0;
			sg.max_len = ~0ull;
			max_m2 = std::max(max_m2, sg.m);
			if (n_kept) n_kept[c] = sg.m;
			nearest_pos[c] = -1;
		}
		const uint64_t P2 = members.size();
		if (P2 == 0) { c0 = c1; continue; }
		// ---- 3. means of the survivors (exact integer column sums), rounded means as slots of a scratch set, distance_d of every survivor
		if (sp) {
			if ((r = sparse_means_and_distances(ctx, pts, segs, pair_seg, members, (uint32_t)nc))) return r;
		} else {
			if (!ctx->batch_scratch || ctx->batch_scratch->k != pts->k || ctx->batch_scratch->dtype != pts->dtype || ctx->batch_scratch->capacity < nc) {
				if (ctx->batch_scratch) { msc_hist_set_destroy(ctx->batch_scratch); ctx->batch_scratch = nullptr; }
				if ((r = msc_hist_set_create(ctx, pts->k, pts->dtype, std::min<uint64_t>(max_chunk_centres, std::max<uint64_t>(nc, 256)), &ctx->batch_scratch))) return r;
			}
			msc_hist_set* rs = ctx->batch_scratch;
			if ((r = ensure(ctx, ctx->floor_sum, nc * sizeof(uint64_t))) || (r = ensure(ctx, ctx->slots, P2 * sizeof(uint32_t))) ||
			    (r = ensure(ctx, ctx->pair_seg, P2 * sizeof(uint32_t))) || (r = ensure(ctx, ctx->partials, P2 * L.S * sizeof(MscPartial))) ||
			    (r = ensure(ctx, ctx->dist, P2 * sizeof(double))))
				return r;
			HIP_TRY(ctx, hipMemcpyAsync(ctx->segs.p, segs.data(), nc * sizeof(MscBatchSeg), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(ctx->pair_seg.p, pair_seg.data(), P2 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, members.data(), P2 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
			HIP_TRY(ctx, msc_launch_colsum_batch(ctx->stream, L, pts->dtype, pts->bins, (const uint32_t*)ctx->slots.p, (const MscBatchSeg*)ctx->segs.p, (uint32_t)nc,
			                                     rs->bins, (uint64_t*)ctx->floor_sum.p));
			HIP_TRY(ctx, hipMemsetAsync(rs->scalars, 0, rs->scalar_stride * nc, ctx->stream));
			HIP_TRY(ctx, msc_launch_finalize(ctx->stream, rs->bins, rs->scalars, L, pts->dtype, 0, nc, false));
			HIP_TRY(ctx, msc_launch_pair_tiles_batch(ctx->stream, L, pts->dtype, pts->bins, pts->scalars, (const uint32_t*)ctx->slots.p, (const MscBatchSeg*)ctx->segs.p,
			                                         (uint32_t)nc, max_m2, rs->bins, rs->L.slot_bytes, rs->scalars, rs->scalar_stride, 0, (MscPartial*)ctx->partials.p,
			                                         MSC_ORDER_CAND_FIRST));
			HIP_TRY(ctx, msc_launch_distance_batch(ctx->stream, (const MscPartial*)ctx->partials.p, L.S, (uint32_t)P2, pts->scalars, pts->scalar_stride,
			                                       (const uint32_t*)ctx->slots.p, (const uint32_t*)ctx->pair_seg.p, rs->scalars, rs->scalar_stride,
			                                       (const uint64_t*)ctx->floor_sum.p, (double*)ctx->dist.p));
		}
		dist.resize(P2);
		HIP_TRY(ctx, hipMemcpyAsync(dist.data(), ctx->dist.p, P2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		// first minimum wins (cluster/Trainer.cpp:150-153)
		for (uint64_t c = c0; c < c1; c++) {
			const MscBatchSeg& sg = segs[c - c0];
			if (sg.m == 0) continue;
			uint32_t best = 0;
			for (uint32_t i = 1; i < sg.m; i++) if (dist[sg.first + i] < dist[sg.first + best]) best = i;
			nearest_pos[c] = (int64_t)where[sg.first + best];
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
extern "C" int msc_merge_all(msc_ctx* ctx, const msc_model* model, double cutoff, const msc_hist_set* centres, const uint32_t* centre_slots, uint64_t n,
                             int delta, int64_t* best_out) {
	if (!ctx || !model || model->ctx != ctx || !centres || centres->ctx != ctx || (n && (!centre_slots || !best_out)) || delta < 0) return MSC_ERR_INVALID_ARG;
	if (n == 0) return MSC_OK;
	for (uint64_t i = 0; i < n; i++) if (centre_slots[i] >= centres->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "centre slot out of range");
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
		for (uint64_t i = 0; i < n; i++) {
			int r = msc_merge(ctx, model, cutoff, centres, centre_slots, n, (int64_t)i, (int64_t)i + 1, (int64_t)std::min<uint64_t>(n - 1, i + (uint64_t)delta), &best_out[i]);
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
	for (uint64_t c0 = 0; c0 < n;) {
		segs.clear(); pair_seg.clear(); cand.clear();
		uint64_t c1 = c0;
		uint32_t max_m = 0;
		while (c1 < n && (c1 == c0 || cand.size() + (uint64_t)delta <= max_chunk_pairs)) {
			MscBatchSeg sg;
			sg.q_slot = centre_slots[c1];
			sg.first = (uint32_t)cand.size();
			const uint64_t last = std::min<uint64_t>(n - 1, c1 + (uint64_t)delta);
			for (uint64_t j = c1 + 1; j <= last; j++) { cand.push_back(centre_slots[j]); pair_seg.push_back((uint32_t)(c1 - c0)); }
			sg.m = (uint32_t)cand.size() - sg.first;
			sg.pad_ = 0;
			const uint64_t len = clen[centre_slots[c1]];
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
					if (!(best_sim > p.combo0)) { best_sim = p.combo0; best = (int64_t)(c + 1 + i); }
				}
				best_out[c] = best;
			}
		}
		c0 = c1;
	}
	return MSC_OK;
}
