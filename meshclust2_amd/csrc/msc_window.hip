// msc_window.hip -- Trainer::get_close over a RANGE of a length-sorted order kept on the device (include/meshclust2_hip.h,
// "window" section).
//
// The accumulate loop (cluster/ClusterFactory.cpp:553-610) calls Trainer::get_close (cluster/Trainer.cpp:23-71) once per step on
// the iterator range bvec::get_range returned: a contiguous stretch of the length-sorted store minus the points that have already
// left it. The reference walks that range on the host (`for (i = istart; i < iend; ++i)`), and so did the r02 driver -- it rebuilt
// the slot list of the window element by element and copied it to the device every step, O(window) host work per step. Here the
// ORDER (position -> slot) and an ALIVE flag per position live in HBM: a step passes [first, end), a compaction kernel lists the
// alive slots of the range, the ordinary 1 x M pipeline scores them, and what comes back is the reduce record and the POSITIONS
// of the close candidates -- which the loop removes from its store next (ClusterFactory.cpp:598-601), so they die here at once.
// Host work per step is O(close + log n): a Fenwick tree over the alive flags answers "how many candidates" without a read-back.
#include <algorithm>
#include <cstring>

#include "msc_objects.h"

struct msc_window {
	msc_ctx* ctx = nullptr;
	const msc_hist_set* set = nullptr;
	uint64_t n = 0;
	uint32_t* d_order = nullptr;      // [n] slot of position i
	uint8_t* d_alive = nullptr;       // [n] 1 = still in the store
	uint32_t* d_slots = nullptr;      // [n] compacted slots of the current range
	uint32_t* d_pos = nullptr;        // [n] ... and their positions
	uint32_t* d_counts = nullptr;     // [0], [1]: the places claimed in the list of the current range (two counters used in turn) + [kMaxBlocks] the close counter
	uint32_t turn = 0;
	uint8_t* d_flags = nullptr;       // [n] close flags of the current range (candidate order)
	uint32_t* h_close = nullptr;      // page-locked, device-visible: [0] best position + 1, [1] n written, [2..] close positions
	uint64_t h_close_cap = 0;
	std::vector<uint32_t> fen;        // Fenwick tree over alive
	std::vector<uint32_t> sorted;     // close positions of the last call, ascending (what the caller reads)
	// positions msc_window_kill took out of the tree but not yet out of d_alive: most steps of an accumulate loop score nothing (empty
	// windows), so the device flags are brought up to date in ONE launch before the next range is compacted (a memset per kill was
	// 200 000 launches = 0.76 s of device time and most of "mark + take" in a 200 000-sequence run)
	std::vector<uint32_t> pending;
	uint32_t* d_kill = nullptr;       // page-locked host memory the kill kernel reads
	uint64_t d_kill_cap = 0;
	hipEvent_t ev_kill = nullptr;     // behind the last kill kernel: the list is rewritten only once that kernel is through
};

namespace {

constexpr uint32_t kWinBlock = 1024;
constexpr uint32_t kWinPer = 8;                        // positions per thread of the multi-block form
constexpr uint32_t kWinTile = kWinBlock * kWinPer;     // positions per block
constexpr uint32_t kMaxBlocks = 1u << 16;

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wave, uint32_t* total) {
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off, 64); if ((int)lane >= off) inc += t; }
	if (lane == 63) s_wave[wave] = inc;
	__syncthreads();
	uint32_t base = 0, tot = 0;
	for (uint32_t w = 0; w < (blockDim.x >> 6); w++) { const uint32_t c = s_wave[w]; if (w < wave) base += c; tot += c; }
	__syncthreads();
	*total = tot;
	return base + inc - v;
}

// one workgroup lists the alive positions of [first, first + range) in order: thread t takes a contiguous run
// (r05: the kills on file -- positions msc_window_kill took out of the tree since the last pass -- reach the alive flags HERE, ahead of
// the compaction that reads them: a launch of their own was one of a step's eight)
// the positions that left the window on the host's side since the last pass (taken as a seed, moved by the driver): up to four travel
// in the kernel's arguments -- the common case is one -- and a longer list in page-locked memory the kernel reads (r05: the list was always
// read from there, by every workgroup: a trip over PCIe in front of the count)
struct WinKills { uint32_t n; uint32_t p[4]; const uint32_t* list; };
__device__ __forceinline__ uint32_t kill_at(const WinKills& k, uint32_t i) { return k.n <= 4 ? (i == 0 ? k.p[0] : i == 1 ? k.p[1] : i == 2 ? k.p[2] : k.p[3]) : k.list[i]; }

__global__ void __launch_bounds__(kWinBlock) k_window_compact_one(uint8_t* alive, const uint32_t* __restrict__ order, uint32_t first, uint32_t range,
                                                                  uint32_t* __restrict__ slots, uint32_t* __restrict__ pos, uint32_t* __restrict__ close_counter,
                                                                  const WinKills kills) {
	const uint32_t n_kills = kills.n;
	__shared__ uint32_t s_wave[kWinBlock / 64];
	if (threadIdx.x == 0) *close_counter = 0;      // k_window_close counts from zero (a memset command of its own was 4 us of every step)
	if (n_kills) {
		for (uint32_t i = threadIdx.x; i < n_kills; i += kWinBlock) alive[kill_at(kills, i)] = 0;
		__threadfence_block();
		__syncthreads();
	}
	const uint32_t per = (range + kWinBlock - 1) / kWinBlock;
	const uint32_t lo = min(range, threadIdx.x * per), hi = min(range, lo + per);
	uint32_t c = 0;
	for (uint32_t i = lo; i < hi; i++) c += alive[first + i];
	uint32_t total;
	uint32_t o = block_excl_scan(c, s_wave, &total);
	for (uint32_t i = lo; i < hi; i++)
		if (alive[first + i]) { slots[o] = order[first + i]; pos[o] = first + i; o++; }
}

// the same over many workgroups in ONE launch: a workgroup counts its tile, claims its places in the list with one atomic and writes
// them -- the tiles' order in the list is whatever order the claims came in. Nothing downstream reads that order: a candidate's result
// depends on the candidate alone, the close positions are sorted by the host, and the best candidate is chosen by similarity and then by
// WINDOW POSITION (k_pair_epilogue_reduce_part), which is what the index in an ordered list stood for. (r05: count, then write with every
// block adding up the counts in front of it, were two of a pass's launches.) claim: two counters used in turn; a pass clears the other's.
// (the kills on file: every block applies those of its own tile before it counts; block 0 those outside the range as well -- nobody
// of this pass reads them)
__global__ void __launch_bounds__(kWinBlock) k_window_compact(uint8_t* alive, const uint32_t* __restrict__ order, uint32_t first, uint32_t range, uint32_t* __restrict__ slots,
                                                              uint32_t* __restrict__ pos, uint32_t* __restrict__ close_counter, uint32_t* __restrict__ claim,
                                                              uint32_t* __restrict__ claim_next, const WinKills kills) {
	const uint32_t n_kills = kills.n;
	__shared__ uint32_t s_wave[kWinBlock / 64];
	__shared__ uint32_t s_base;
	if (blockIdx.x == 0 && threadIdx.x == 0) { *close_counter = 0; *claim_next = 0; }
	if (n_kills) {
		const uint32_t t_lo = first + blockIdx.x * (blockDim.x * kWinPer), t_hi = t_lo + blockDim.x * kWinPer;
		for (uint32_t i = threadIdx.x; i < n_kills; i += blockDim.x) {
			const uint32_t p = kill_at(kills, i);
			if ((p >= t_lo && p < t_hi) || (blockIdx.x == 0 && (p < first || p >= first + range))) alive[p] = 0;
		}
		__threadfence_block();
		__syncthreads();
	}
	const uint32_t base = blockIdx.x * (blockDim.x * kWinPer) + threadIdx.x * kWinPer;
	uint32_t c = 0;
#pragma unroll
	for (uint32_t j = 0; j < kWinPer; j++) if (base + j < range) c += alive[first + base + j];
	uint32_t total;
	const uint32_t mine = block_excl_scan(c, s_wave, &total);
	if (threadIdx.x == 0) s_base = total ? atomicAdd(claim, total) : 0u;
	__syncthreads();
	uint32_t o = s_base + mine;
#pragma unroll
	for (uint32_t j = 0; j < kWinPer; j++)
		if (base + j < range && alive[first + base + j]) { slots[o] = order[first + base + j]; pos[o] = first + base + j; o++; }
}

// behind the reduce kernel: the positions of the close candidates (any order; the host sorts the few there are) go to host
// memory, their alive flags clear, and the best candidate's index becomes a position
__global__ void __launch_bounds__(256) k_window_close(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ pos, uint32_t m, uint8_t* __restrict__ alive,
                                                      uint32_t* __restrict__ counter, const MscReduceOut* __restrict__ rec, uint32_t* __restrict__ out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i == 0) out[0] = rec->best_pos >= 0 ? (uint32_t)rec->best_pos + 1u : 0u;          // (the reduce chose by window position: best_pos is one)
	if (i >= m || !flags[i]) return;
	const uint32_t p = pos[i];
	alive[p] = 0;
	out[2 + atomicAdd(counter, 1u)] = p;
}


void fen_add(std::vector<uint32_t>& f, uint64_t i, int d) { for (i++; i < f.size(); i += i & (~i + 1)) f[i] = (uint32_t)((int64_t)f[i] + d); }
uint64_t fen_prefix(const std::vector<uint32_t>& f, uint64_t i) { uint64_t s = 0; for (; i > 0; i -= i & (~i + 1)) s += f[i]; return s; }

double win_get_id(double cutoff) { return cutoff > 1 ? cutoff / 100.0 : cutoff; }

}  // namespace

extern "C" void msc_window_destroy(msc_window* w) {
	if (!w) return;
	if (w->ctx) (void)hipSetDevice(w->ctx->device);
	(void)hipFree(w->d_order); (void)hipFree(w->d_alive); (void)hipFree(w->d_slots); (void)hipFree(w->d_pos); (void)hipFree(w->d_counts); (void)hipFree(w->d_flags);
	if (w->h_close) (void)hipHostFree(w->h_close);
	if (w->d_kill) (void)hipHostFree(w->d_kill);
	if (w->ev_kill) (void)hipEventDestroy(w->ev_kill);
	delete w;
}

extern "C" int msc_window_create(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* slots, uint64_t n, msc_window** out) {
	if (!ctx || !set || set->ctx != ctx || !out || (!slots && n)) return MSC_ERR_INVALID_ARG;
	*out = nullptr;
	if (n > 0xfffffff0ull) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_window_create: too many positions");
	for (uint64_t i = 0; i < n; i++) if (slots[i] >= set->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_window_create: slot %u out of range", slots[i]);
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	msc_window* w = new msc_window();
	w->ctx = ctx; w->set = set; w->n = n;
	const size_t nn = std::max<uint64_t>(n, 1);
	if (hipMalloc(&w->d_order, nn * 4) != hipSuccess || hipMalloc(&w->d_alive, nn) != hipSuccess || hipMalloc(&w->d_slots, nn * 4) != hipSuccess ||
	    hipMalloc(&w->d_pos, nn * 4) != hipSuccess || hipMalloc(&w->d_counts, (kMaxBlocks + 1) * 4) != hipSuccess || hipMalloc(&w->d_flags, nn) != hipSuccess) {
		(void)hipGetLastError();
		msc_window_destroy(w);
		return fail(ctx, MSC_ERR_OOM, "msc_window_create: out of device memory");
	}
	if (n) {
		HIP_TRY(ctx, hipMemcpyAsync(w->d_order, slots, n * 4, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemsetAsync(w->d_alive, 1, n, ctx->stream));
		HIP_TRY(ctx, hipMemsetAsync(w->d_counts, 0, 8, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	w->fen.assign(n + 1, 0);
	for (uint64_t i = 1; i <= n; i++) {          // all ones, built in O(n)
		w->fen[i] += 1;
		const uint64_t j = i + (i & (~i + 1));
		if (j <= n) w->fen[j] += w->fen[i];
	}
	*out = w;
	return MSC_OK;
}

extern "C" uint64_t msc_window_alive(const msc_window* w, uint64_t first, uint64_t end) {
	if (!w || first >= end) return 0;
	end = std::min(end, w->n);
	if (first >= end) return 0;
	return fen_prefix(w->fen, end) - fen_prefix(w->fen, first);
}

extern "C" int msc_window_kill(msc_ctx* ctx, msc_window* w, const uint32_t* positions, uint64_t n) {
	if (!ctx || !w || w->ctx != ctx || (!positions && n)) return MSC_ERR_INVALID_ARG;
	if (n == 0) return MSC_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	// validate the whole call first, then apply it: a rejected call leaves the tree, the pending list and the device flags as they
	// were (a half-applied one would leave the host's count below the number of alive flags, and the next range scored short)
	std::vector<uint32_t> sorted(positions, positions + n);
	std::sort(sorted.begin(), sorted.end());
	for (uint64_t i = 0; i < n; i++) {
		if (sorted[i] >= w->n) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_window_kill: position out of range");
		if (i && sorted[i] == sorted[i - 1]) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_window_kill: position %u is listed twice", sorted[i]);
		// (a position may die once: the tree holds what the device flags hold)
		if (fen_prefix(w->fen, (uint64_t)sorted[i] + 1) - fen_prefix(w->fen, sorted[i]) == 0) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_window_kill: position %u is already dead", sorted[i]);
	}
	for (uint64_t i = 0; i < n; i++) fen_add(w->fen, positions[i], -1);
	w->pending.insert(w->pending.end(), positions, positions + n);
	return MSC_OK;
}

// the kills on file go to the page-locked list the next compaction kernel reads (it applies them ahead of its own reads of d_alive);
// *d_list / *n_out: what to hand that kernel
static int stage_kills(msc_ctx* ctx, msc_window* w, WinKills* out, bool* on_file) {
	*out = WinKills{0, {0, 0, 0, 0}, nullptr};
	*on_file = false;
	if (w->pending.empty()) return MSC_OK;
	const uint64_t n = w->pending.size();
	if (n <= 4) {
		out->n = (uint32_t)n;
		for (uint64_t i = 0; i < n; i++) out->p[i] = w->pending[i];
		w->pending.clear();
		return MSC_OK;
	}
	// the list sits in page-locked memory the kernel reads directly (a pageable hipMemcpyAsync of a handful of words was a copy command
	// and a staging stall per step); it stays untouched until the next pass, which comes after the stream was waited for by this one
	if (w->d_kill_cap < n) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (w->d_kill) (void)hipHostFree(w->d_kill);
		w->d_kill = nullptr;
		w->d_kill_cap = std::max<uint64_t>(2 * n, 4096);
		HIP_TRY(ctx, hipHostMalloc((void**)&w->d_kill, w->d_kill_cap * 4, hipHostMallocDefault));
	}
	if (w->ev_kill) HIP_TRY(ctx, hipEventSynchronize(w->ev_kill));          // (already complete: every pass ends with a wait for the stream)
	memcpy(w->d_kill, w->pending.data(), n * 4);
	w->pending.clear();
	uint32_t* d_list = nullptr;
	HIP_TRY(ctx, hipHostGetDevicePointer((void**)&d_list, w->d_kill, 0));
	out->n = (uint32_t)n;
	out->list = d_list;
	*on_file = true;
	return MSC_OK;
}

extern "C" int msc_get_close_window(msc_ctx* ctx, const msc_model* model, double cutoff, msc_window* w, uint64_t first, uint64_t end, const msc_hist_set* qset,
                                    uint64_t q_slot, const uint32_t** close_pos, uint64_t* n_close, int64_t* best_pos, double* best_sim, int* is_min) {
	if (!ctx || !model || model->ctx != ctx || !w || w->ctx != ctx || !qset || !n_close || !best_pos || !is_min) return MSC_ERR_INVALID_ARG;
	end = std::min(end, w->n);
	*n_close = 0; *best_pos = -1; *is_min = 1;
	if (best_sim) *best_sim = -1.0;
	if (close_pos) *close_pos = nullptr;
	const uint64_t m = msc_window_alive(w, first, end);
	if (m == 0) return MSC_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int r;
	uint64_t qlen = 0;
	if ((r = slot_length(ctx, qset, q_slot, &qlen))) return r;
	WinKills kills;
	bool kills_on_file = false;
	if ((r = stage_kills(ctx, w, &kills, &kills_on_file))) return r;
	const uint32_t range = (uint32_t)(end - first);
	// One workgroup walking the range alone takes 50 us for 40 000 positions (a third of a step on a window-bearing set: profiles/
	// r04_notes.md); from a few thousand positions on, many small workgroups count their tiles and claim their places (k_window_compact)
	if (range <= 4 * kWinBlock) {
		k_window_compact_one<<<dim3(1), dim3(kWinBlock), 0, ctx->stream>>>(w->d_alive, w->d_order, (uint32_t)first, range, w->d_slots, w->d_pos, w->d_counts + kMaxBlocks, kills);
	} else {
		const uint32_t tb = range <= 64 * kWinTile ? 256 : kWinBlock, tile = tb * kWinPer;
		const uint32_t blocks = (range + tile - 1) / tile;
		if (blocks > kMaxBlocks) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_get_close_window: range too long");
		uint32_t *claim = w->d_counts + (w->turn & 1u), *claim_next = w->d_counts + ((w->turn & 1u) ^ 1u);
		w->turn++;
		k_window_compact<<<dim3(blocks), dim3(tb), 0, ctx->stream>>>(w->d_alive, w->d_order, (uint32_t)first, range, w->d_slots, w->d_pos, w->d_counts + kMaxBlocks, claim, claim_next, kills);
	}
	HIP_TRY(ctx, hipGetLastError());
	if (kills_on_file) {          // (the list is rewritten only once this kernel is through)
		if (!w->ev_kill) HIP_TRY(ctx, hipEventCreateWithFlags(&w->ev_kill, hipEventDisableTiming));
		HIP_TRY(ctx, hipEventRecord(w->ev_kill, ctx->stream));
	}
	if (w->h_close_cap < m + 2) {
		if (w->h_close) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipHostFree(w->h_close)); w->h_close = nullptr; w->h_close_cap = 0; }
		const uint64_t cap = std::max<uint64_t>(m + 2, std::min<uint64_t>(w->n + 2, 2 * w->h_close_cap + 4096));
		HIP_TRY(ctx, hipHostMalloc((void**)&w->h_close, cap * 4, hipHostMallocDefault));
		w->h_close_cap = cap;
	}
	uint32_t* d_out = nullptr;
	HIP_TRY(ctx, hipHostGetDevicePointer((void**)&d_out, w->h_close, 0));
	uint32_t* counter = w->d_counts + kMaxBlocks;
	ScoreRequest rq;
	MscReduceOut ro;
	rq.model = model; rq.cands = w->set; rq.dev_slots = w->d_slots; rq.m = m; rq.qset = qset; rq.q_slot = q_slot;
	rq.order = MSC_ORDER_CAND_FIRST;
	rq.use_window = 1;
	rq.min_len = (uint64_t)((double)qlen * cutoff);          // cluster/Trainer.cpp:39-40: uint64 truncation
	rq.max_len = (uint64_t)((double)qlen / cutoff);
	rq.dev_flags_out = w->d_flags;
	rq.reduce_mode = MSC_REDUCE_GET_CLOSE;
	rq.reduce_host = &ro;
	rq.after_reduce = [&](const MscReduceOut* d_rec) -> hipError_t {
		k_window_close<<<dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream>>>(w->d_flags, w->d_pos, (uint32_t)m, w->d_alive, counter, d_rec, d_out);
		return hipGetLastError();
	};
	rq.close_list = MscCloseList{w->d_pos, w->d_alive, counter, d_out, w->h_close};
	(void)win_get_id;
	if ((r = run_score(ctx, rq))) return r;
	*n_close = ro.n_close;
	*is_min = ro.any_close ? 0 : 1;
	*best_pos = (int64_t)w->h_close[0] - 1;
	if (best_sim) *best_sim = ro.best_sim;
	w->sorted.assign(w->h_close + 2, w->h_close + 2 + ro.n_close);
	std::sort(w->sorted.begin(), w->sorted.end());
	for (uint32_t p : w->sorted) fen_add(w->fen, p, -1);
	if (close_pos) *close_pos = w->sorted.data();
	return MSC_OK;
}
