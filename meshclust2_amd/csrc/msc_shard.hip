// msc_shard.hip -- what a host driver that shards the points over several GPUs needs from each GPU's library instance
// (include/meshclust2_hip.h, "multi-GPU" section). The library stays per-device and owns no communicator: the driver
// (meshclust2_amd/host/msc_sharded.hpp, RCCL over xGMI) moves the byte ranges these entry points produce and consume.
//
//   pack / unpack   a histogram slot -- dense (scalar record + tile-permuted bins) or sparse (scalar record + sub-range table + the
//                   (bin, value) list + its cum array: ~12 bytes per k-mer of the sequence instead of 4^k bins) -- as one
//                   contiguous, self-describing byte range in device memory: the query of a get_close step is broadcast in that
//                   form, the new centres of an update round are all-gathered in it.
//   column sums     get_mean (cluster/ClusterFactory.cpp:338-380) and the mean of mean_shift_update (:297-326) over members
//                   that live on different GPUs: every rank sums ITS members (msc_colsum_partial), the sums are added across
//                   ranks -- dense sets: an all-reduce of uint64 columns in place; sparse sets: an all-gather of each rank's summed
//                   excesses as a sparse list, scatter-added on arrival -- and every rank rounds the same mean and measures only
//                   its own members (msc_colsum_nearest). Integer sums are exact, so the mean is the single-GPU mean bit for bit.
#include <algorithm>
#include <cstring>

#include "msc_objects.h"

namespace {

constexpr uint32_t kPackDense = 0x4d534431u, kPackSparse = 0x4d534332u;      // "MSD1" / "MSC2"
constexpr uint64_t kSplitBytes = 80;                                          // split[17] padded to a multiple of 16

struct PackHead { uint32_t kind, nnz; uint64_t bytes; };                      // first 16 bytes of a packed slot
static_assert(sizeof(PackHead) == 16, "packed slot header");

inline uint64_t up16(uint64_t v) { return (v + 15) & ~15ull; }

struct CopySeg { const uint8_t* src; uint8_t* dst; uint64_t bytes; };

// piece p of 16 KiB of segment seg_of[p]: one workgroup each (sizes and addresses are multiples of 4)
constexpr uint32_t kPiece = 16384;
__global__ void __launch_bounds__(256) k_copy_segments(const CopySeg* __restrict__ segs, const uint32_t* __restrict__ piece_seg, const uint32_t* __restrict__ piece_at) {
	const CopySeg s = segs[piece_seg[blockIdx.x]];
	const uint64_t at = (uint64_t)piece_at[blockIdx.x] * kPiece;
	const uint64_t n = min((uint64_t)kPiece, s.bytes - at);
	const uint8_t* src = s.src + at;
	uint8_t* dst = s.dst + at;
	if ((((uintptr_t)src | (uintptr_t)dst | n) & 15) == 0) {
		for (uint64_t i = threadIdx.x; i < n / 16; i += 256) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
	} else {
		for (uint64_t i = threadIdx.x; i < n / 4; i += 256) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
	}
}

__global__ void k_write_words(uint32_t* dst, uint4 w) { *reinterpret_cast<uint4*>(dst) = w; }

int run_copies(msc_ctx* ctx, const std::vector<CopySeg>& segs) {
	if (segs.empty()) return MSC_OK;
	std::vector<uint32_t> piece_seg, piece_at;
	for (size_t i = 0; i < segs.size(); i++)
		for (uint64_t at = 0; at * kPiece < segs[i].bytes; at++) { piece_seg.push_back((uint32_t)i); piece_at.push_back((uint32_t)at); }
	if (piece_seg.empty()) return MSC_OK;
	int r;
	const size_t a = segs.size() * sizeof(CopySeg), b = piece_seg.size() * 4;
	if ((r = ensure(ctx, ctx->shard_hdrs, a + 2 * b + 64))) return r;
	if ((r = ensure_pinned(ctx, ctx->pin_up, a + 2 * b + 64))) return r;
	uint8_t* h = (uint8_t*)ctx->pin_up.p;
	memcpy(h, segs.data(), a);
	memcpy(h + a, piece_seg.data(), b);
	memcpy(h + a + b, piece_at.data(), b);
	HIP_TRY(ctx, hipMemcpyAsync(ctx->shard_hdrs.p, h, a + 2 * b, hipMemcpyHostToDevice, ctx->stream));
	const uint8_t* d = (const uint8_t*)ctx->shard_hdrs.p;
	k_copy_segments<<<dim3((unsigned)piece_seg.size()), dim3(256), 0, ctx->stream>>>((const CopySeg*)d, (const uint32_t*)(d + a), (const uint32_t*)(d + a + b));
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // pin_up and shard_hdrs are free again; the copies have landed
	return MSC_OK;
}

uint64_t packed_bytes(const msc_hist_set* s, uint64_t slot) {
	const uint64_t sc = up16(s->scalar_stride);
	if (!s->sparse) return sizeof(PackHead) + sc + up16(s->L.slot_bytes);
	const uint64_t nnz = s->hdr_host[slot].nnz;
	return sizeof(PackHead) + sc + kSplitBytes + up16(nnz * 8) + up16(nnz * 4);
}

}  // namespace

// ================================================================================================ plumbing
extern "C" void* msc_stream_handle(msc_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int msc_device_malloc(msc_ctx* ctx, uint64_t bytes, void** out) {
	if (!ctx || !out) return MSC_ERR_INVALID_ARG;
	*out = nullptr;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMalloc(out, std::max<uint64_t>(bytes, 16)));
	return MSC_OK;
}
extern "C" int msc_device_free(msc_ctx* ctx, void* p) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	if (p) { HIP_TRY(ctx, hipSetDevice(ctx->device)); HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(p)); }
	return MSC_OK;
}
extern "C" int msc_host_alloc(msc_ctx* ctx, uint64_t bytes, void** out) {
	if (!ctx || !out) return MSC_ERR_INVALID_ARG;
	*out = nullptr;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipHostMalloc(out, std::max<uint64_t>(bytes, 16), hipHostMallocDefault));
	return MSC_OK;
}
extern "C" int msc_host_free(msc_ctx* ctx, void* p) {
	if (!ctx) return MSC_ERR_INVALID_ARG;
	if (p) { HIP_TRY(ctx, hipSetDevice(ctx->device)); HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipHostFree(p)); }
	return MSC_OK;
}
extern "C" int msc_memcpy_to_host(msc_ctx* ctx, void* dst, const void* src_dev, uint64_t bytes) {
	if (!ctx || (bytes && (!dst || !src_dev))) return MSC_ERR_INVALID_ARG;
	if (!bytes) return MSC_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MSC_OK;
}
extern "C" int msc_memcpy_to_device(msc_ctx* ctx, void* dst_dev, const void* src, uint64_t bytes) {
	if (!ctx || (bytes && (!dst_dev || !src))) return MSC_ERR_INVALID_ARG;
	if (!bytes) return MSC_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MSC_OK;
}

extern "C" int msc_memcpy_device(msc_ctx* ctx, void* dst_dev, const void* src_dev, uint64_t bytes) {
	if (!ctx || (bytes && (!dst_dev || !src_dev))) return MSC_ERR_INVALID_ARG;
	if (!bytes) return MSC_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));      // ordered with whatever follows on the ctx stream
	return MSC_OK;
}

// ================================================================================================ pack / unpack
extern "C" uint64_t msc_hist_packed_bytes(const msc_hist_set* set, uint64_t slot) {
	if (!set || slot >= set->capacity) return 0;
	return packed_bytes(set, slot);
}

extern "C" int msc_hist_pack(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* slots, uint64_t n, void* dev_dst, const uint64_t* offsets) {
	if (!ctx || !set || set->ctx != ctx || (n && (!slots || !dev_dst || !offsets))) return MSC_ERR_INVALID_ARG;
	if (n == 0) return MSC_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	std::vector<CopySeg> segs;
	std::vector<PackHead> heads(n);
	uint8_t* dst = (uint8_t*)dev_dst;
	const uint64_t sc = up16(set->scalar_stride);
	if (n == 1) {
		// one slot (the query of a get_close step): a head written by value and two to four device copies, all queued on the ctx
		// stream and not waited for -- a collective queued on the same stream (msc_stream_handle) follows them in order
		const uint32_t slot = slots[0];
		if (slot >= set->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_pack: slot out of range");
		if (offsets[0] & 15) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_pack: offsets must be multiples of 16");
		uint8_t* o = dst + offsets[0];
		PackHead ph;
		ph.bytes = packed_bytes(set, slot);
		ph.kind = set->sparse ? kPackSparse : kPackDense;
		ph.nnz = set->sparse ? set->hdr_host[slot].nnz : 0;
		uint4 w;
		memcpy(&w, &ph, sizeof w);
		k_write_words<<<dim3(1), dim3(1), 0, ctx->stream>>>((uint32_t*)o, w);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipMemcpyAsync(o + sizeof(PackHead), set->scalars + (uint64_t)slot * set->scalar_stride, set->scalar_stride, hipMemcpyDeviceToDevice, ctx->stream));
		if (!set->sparse) {
			HIP_TRY(ctx, hipMemcpyAsync(o + sizeof(PackHead) + sc, set->bins + (uint64_t)slot * set->L.slot_bytes, set->L.slot_bytes, hipMemcpyDeviceToDevice, ctx->stream));
		} else {
			const MscSparseHdr& h = set->hdr_host[slot];
			uint8_t* p = o + sizeof(PackHead) + sc;
			HIP_TRY(ctx, hipMemcpyAsync(p, (const uint8_t*)(set->hdr + slot) + offsetof(MscSparseHdr, split), sizeof(uint32_t) * (MSC_SPARSE_SUB + 1), hipMemcpyDeviceToDevice, ctx->stream));
			p += kSplitBytes;
			if (h.nnz) {
				HIP_TRY(ctx, hipMemcpyAsync(p, set->ent + h.off, (uint64_t)h.nnz * 8, hipMemcpyDeviceToDevice, ctx->stream));
				HIP_TRY(ctx, hipMemcpyAsync(p + up16((uint64_t)h.nnz * 8), set->cum + h.off, (uint64_t)h.nnz * 4, hipMemcpyDeviceToDevice, ctx->stream));
			}
		}
		return MSC_OK;
	}
	for (uint64_t i = 0; i < n; i++) {
		const uint32_t slot = slots[i];
		if (slot >= set->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_pack: slot out of range");
		if (offsets[i] & 15) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_pack: offsets must be multiples of 16");
		uint8_t* o = dst + offsets[i];
		heads[i].bytes = packed_bytes(set, slot);
		segs.push_back(CopySeg{set->scalars + (uint64_t)slot * set->scalar_stride, o + sizeof(PackHead), set->scalar_stride});
		if (!set->sparse) {
			heads[i].kind = kPackDense; heads[i].nnz = 0;
			segs.push_back(CopySeg{set->bins + (uint64_t)slot * set->L.slot_bytes, o + sizeof(PackHead) + sc, set->L.slot_bytes});
		} else {
			const MscSparseHdr& h = set->hdr_host[slot];
			heads[i].kind = kPackSparse; heads[i].nnz = h.nnz;
			uint8_t* p = o + sizeof(PackHead) + sc;
			segs.push_back(CopySeg{(const uint8_t*)(set->hdr + slot) + offsetof(MscSparseHdr, split), p, sizeof(uint32_t) * (MSC_SPARSE_SUB + 1)});
			p += kSplitBytes;
			if (h.nnz) {
				segs.push_back(CopySeg{(const uint8_t*)(set->ent + h.off), p, (uint64_t)h.nnz * 8});
				segs.push_back(CopySeg{(const uint8_t*)(set->cum + h.off), p + up16((uint64_t)h.nnz * 8), (uint64_t)h.nnz * 4});
			}
		}
	}
	// the 16-byte heads travel through the same copy kernel: staged behind the segment tables
	int r;
	if ((r = ensure(ctx, ctx->shard_payload, n * sizeof(PackHead)))) return r;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->shard_payload.p, heads.data(), n * sizeof(PackHead), hipMemcpyHostToDevice, ctx->stream));
	for (uint64_t i = 0; i < n; i++) segs.push_back(CopySeg{(const uint8_t*)ctx->shard_payload.p + i * sizeof(PackHead), dst + offsets[i], sizeof(PackHead)});
	return run_copies(ctx, segs);
}

extern "C" int msc_hist_set_reset(msc_ctx* ctx, msc_hist_set* set) {
	if (!ctx || !set || set->ctx != ctx) return MSC_ERR_INVALID_ARG;
	if (!set->sparse) return MSC_OK;
	// (whatever is queued on the ctx stream still reads the old lists; what overwrites them is queued behind it on the same stream)
	set->ent_used = 0;
	set->list_epoch++;
	for (MscSparseHdr& h : set->hdr_host) { h.nnz = 0; h.off = 0; }
	return MSC_OK;
}

extern "C" int msc_hist_unpack(msc_ctx* ctx, msc_hist_set* set, const uint32_t* slots, uint64_t n, const void* dev_src, const uint64_t* offsets) {
	if (!ctx || !set || set->ctx != ctx || (n && (!slots || !dev_src || !offsets))) return MSC_ERR_INVALID_ARG;
	if (n == 0) return MSC_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint8_t* src = (const uint8_t*)dev_src;
	const uint64_t sc = up16(set->scalar_stride);
	int r;
	if (n == 1) {
		// one slot: head, scalar record and sub-range table in ONE copy to the host (the only wait), the payload by device copies
		if (slots[0] >= set->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_unpack: slot out of range");
		if (offsets[0] & 15) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_unpack: offsets must be multiples of 16");
		const uint8_t* o = src + offsets[0];
		const uint64_t front = sizeof(PackHead) + sc + (set->sparse ? kSplitBytes : 0);
		std::vector<uint8_t> hh(front);
		HIP_TRY(ctx, hipMemcpyAsync(hh.data(), o, front, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		PackHead ph;
		memcpy(&ph, hh.data(), sizeof ph);
		if (ph.kind != (set->sparse ? kPackSparse : kPackDense)) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_unpack: the packed slot is not of this set's layout");
		const uint32_t slot = slots[0];
		HIP_TRY(ctx, hipMemcpyAsync(set->scalars + (uint64_t)slot * set->scalar_stride, o + sizeof(PackHead), set->scalar_stride, hipMemcpyDeviceToDevice, ctx->stream));
		if (!set->sparse) {
			if (ph.bytes != packed_bytes(set, slot)) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_unpack: packed slot of another k or bin type");
			HIP_TRY(ctx, hipMemcpyAsync(set->bins + (uint64_t)slot * set->L.slot_bytes, o + sizeof(PackHead) + sc, set->L.slot_bytes, hipMemcpyDeviceToDevice, ctx->stream));
		} else {
			if (set->ent_used + ph.nnz > set->ent_capacity) return fail(ctx, MSC_ERR_OOM, "sparse set entry arena exhausted (%llu of %llu entries used, %u more needed)",
			                                                            (unsigned long long)set->ent_used, (unsigned long long)set->ent_capacity, ph.nnz);
			MscSparseHdr h{};
			h.off = set->ent_used;
			h.nnz = ph.nnz;
			memcpy(h.split, hh.data() + sizeof(PackHead) + sc, sizeof h.split);
			if (h.split[MSC_SPARSE_SUB] != ph.nnz) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_unpack: corrupt list header");
			const uint8_t* p = o + sizeof(PackHead) + sc + kSplitBytes;
			if (ph.nnz) {
				HIP_TRY(ctx, hipMemcpyAsync(set->ent + h.off, p, (uint64_t)ph.nnz * 8, hipMemcpyDeviceToDevice, ctx->stream));
				HIP_TRY(ctx, hipMemcpyAsync(set->cum + h.off, p + up16((uint64_t)ph.nnz * 8), (uint64_t)ph.nnz * 4, hipMemcpyDeviceToDevice, ctx->stream));
			}
			set->ent_used += ph.nnz;
			set->list_epoch++;
			set->hdr_host[slot] = h;          // (the mirror outlives the copy below: it is the source)
			HIP_TRY(ctx, hipMemcpyAsync(set->hdr + slot, &set->hdr_host[slot], sizeof(MscSparseHdr), hipMemcpyHostToDevice, ctx->stream));
			set->max_nnz = std::max(set->max_nnz, h.nnz);
		}
		// bounds and length from the scalar record that came with the head: no second read-back
		MscSlotScalars rec;
		memcpy(&rec, hh.data() + sizeof(PackHead), sizeof rec);
		mark_written(set, slot, 1);
		set->max_count = std::max(set->max_count, rec.max_count);
		set->max_sum = std::max(set->max_sum, rec.sum);
		learn_length(set, slot, rec.length);
		return MSC_OK;
	}
	// 1. the heads (and, for lists, the sub-range tables) come to the host: list lengths decide where the entries go
	const uint64_t hb = sizeof(PackHead) + (set->sparse ? kSplitBytes : 0);
	std::vector<CopySeg> segs;
	if ((r = ensure(ctx, ctx->shard_payload, n * hb))) return r;
	for (uint64_t i = 0; i < n; i++) {
		if (slots[i] >= set->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_unpack: slot out of range");
		if (offsets[i] & 15) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_unpack: offsets must be multiples of 16");
		segs.push_back(CopySeg{src + offsets[i], (uint8_t*)ctx->shard_payload.p + i * hb, sizeof(PackHead)});
		if (set->sparse) segs.push_back(CopySeg{src + offsets[i] + sizeof(PackHead) + sc, (uint8_t*)ctx->shard_payload.p + i * hb + sizeof(PackHead), kSplitBytes});
	}
	if ((r = run_copies(ctx, segs))) return r;
	std::vector<uint8_t> hh(n * hb);
	HIP_TRY(ctx, hipMemcpyAsync(hh.data(), ctx->shard_payload.p, n * hb, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	// 2. the payloads
	segs.clear();
	std::vector<MscSparseHdr> new_hdr;
	uint64_t used = set->ent_used;
	for (uint64_t i = 0; i < n; i++) {
		PackHead ph;
		memcpy(&ph, hh.data() + i * hb, sizeof ph);
		if (ph.kind != (set->sparse ? kPackSparse : kPackDense)) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_unpack: packed slot %llu is not of this set's layout", (unsigned long long)i);
		const uint8_t* o = src + offsets[i];
		segs.push_back(CopySeg{o + sizeof(PackHead), set->scalars + (uint64_t)slots[i] * set->scalar_stride, set->scalar_stride});
		if (!set->sparse) {
			if (ph.bytes != packed_bytes(set, slots[i])) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_unpack: packed slot of another k or bin type");
			segs.push_back(CopySeg{o + sizeof(PackHead) + sc, set->bins + (uint64_t)slots[i] * set->L.slot_bytes, set->L.slot_bytes});
		} else {
			if (used + ph.nnz > set->ent_capacity) return fail(ctx, MSC_ERR_OOM, "sparse set entry arena exhausted (%llu of %llu entries used, %u more needed)",
			                                                    (unsigned long long)used, (unsigned long long)set->ent_capacity, ph.nnz);
			MscSparseHdr h{};
			h.off = used;
			h.nnz = ph.nnz;
			memcpy(h.split, hh.data() + i * hb + sizeof(PackHead), sizeof h.split);
			if (h.split[MSC_SPARSE_SUB] != ph.nnz) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_hist_unpack: corrupt list header");
			const uint8_t* p = o + sizeof(PackHead) + sc + kSplitBytes;
			if (ph.nnz) {
				segs.push_back(CopySeg{p, (uint8_t*)(set->ent + h.off), (uint64_t)ph.nnz * 8});
				segs.push_back(CopySeg{p + up16((uint64_t)ph.nnz * 8), (uint8_t*)(set->cum + h.off), (uint64_t)ph.nnz * 4});
			}
			used += ph.nnz;
			new_hdr.push_back(h);
		}
	}
	if (set->sparse) {
		if ((r = ensure(ctx, ctx->shard_payload, n * sizeof(MscSparseHdr)))) return r;
		HIP_TRY(ctx, hipMemcpyAsync(ctx->shard_payload.p, new_hdr.data(), n * sizeof(MscSparseHdr), hipMemcpyHostToDevice, ctx->stream));
		for (uint64_t i = 0; i < n; i++) segs.push_back(CopySeg{(const uint8_t*)ctx->shard_payload.p + i * sizeof(MscSparseHdr), (uint8_t*)(set->hdr + slots[i]), sizeof(MscSparseHdr)});
	}
	if ((r = run_copies(ctx, segs))) return r;
	if (set->sparse) {
		set->ent_used = used;
		set->list_epoch++;
		for (uint64_t i = 0; i < n; i++) set->hdr_host[slots[i]] = new_hdr[i];
	}
	// 3. the host-side bounds and lengths follow the records that arrived (consecutive slots in one strided copy)
	uint64_t i = 0;
	while (i < n) {
		uint64_t j = i + 1;
		while (j < n && slots[j] == slots[j - 1] + 1) j++;
		if ((r = refresh_bounds(ctx, set, slots[i], j - i))) return r;
		i = j;
	}
	return MSC_OK;
}

// ================================================================================================ column sums
namespace {

// lists: n member lists over member_slots; returns segs (q_slot = list index) and the flat pair_seg
int lists_of(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, const uint64_t* offsets, uint64_t n, std::vector<MscBatchSeg>& segs,
             std::vector<uint32_t>& pair_seg, uint32_t* max_m) {
	if (n > 0x7fffffffull) return fail(ctx, MSC_ERR_INVALID_ARG, "too many lists");
	const uint64_t total = offsets[n];
	if (total > 0xfffffff0ull) return fail(ctx, MSC_ERR_INVALID_ARG, "too many members");
	for (uint64_t i = 0; i < total; i++) if (member_slots[i] >= set->capacity) return fail(ctx, MSC_ERR_INVALID_ARG, "member slot out of range");
	segs.resize(n);
	pair_seg.resize(total);
	*max_m = 0;
	for (uint64_t c = 0; c < n; c++) {
		if (offsets[c + 1] < offsets[c]) return fail(ctx, MSC_ERR_INVALID_ARG, "offsets must be non-decreasing");
		MscBatchSeg& sg = segs[c];
		sg.q_slot = (uint32_t)c; sg.first = (uint32_t)offsets[c]; sg.m = (uint32_t)(offsets[c + 1] - offsets[c]); sg.pad_ = 0; sg.min_len = 0; sg.max_len = ~0ull;
		*max_m = std::max(*max_m, sg.m);
		for (uint32_t i = 0; i < sg.m; i++) pair_seg[sg.first + i] = (uint32_t)c;
	}
	return MSC_OK;
}

// sparse payload of one rank: u64 {n, bytes of the whole blob}, then per list {members, offset of its packed slot}, then the slots
uint64_t blob_head_bytes(uint64_t n) { return up16(16 + n * 16); }

}  // namespace

extern "C" uint64_t msc_colsum_list_bytes(const msc_hist_set* set) {
	if (!set) return 0;
	return set->sparse ? set->L.nbins * 4 : set->L.padded_bins * 8 + set->L.slot_bytes;
}

extern "C" int msc_colsum_partial(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, const uint64_t* offsets, uint64_t n, void** dev_payload,
                                  uint64_t* payload_bytes) {
	if (!ctx || !set || set->ctx != ctx || !offsets || !dev_payload || !payload_bytes || n == 0 || (offsets[n] && !member_slots)) return MSC_ERR_INVALID_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const MscLayout& L = set->L;
	std::vector<MscBatchSeg> segs;
	std::vector<uint32_t> pair_seg;
	uint32_t max_m = 0;
	int r;
	if ((r = lists_of(ctx, set, member_slots, offsets, n, segs, pair_seg, &max_m))) return r;
	const uint64_t total = offsets[n];
	if (!set->sparse) {
		// [n][padded_bins] uint64 column sums, then n uint64 member counts: one all-reduce (sum) adds both across ranks
		const uint64_t elems = n * L.padded_bins + n;
		if ((r = ensure(ctx, ctx->shard_payload, elems * 8)) || (r = ensure(ctx, ctx->segs, n * sizeof(MscBatchSeg))) || (r = ensure(ctx, ctx->slots, std::max<uint64_t>(total, 1) * 4))) return r;
		HIP_TRY(ctx, hipMemcpyAsync(ctx->segs.p, segs.data(), n * sizeof(MscBatchSeg), hipMemcpyHostToDevice, ctx->stream));
		if (total) HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, member_slots, total * 4, hipMemcpyHostToDevice, ctx->stream));
		std::vector<uint64_t> counts(n);
		for (uint64_t c = 0; c < n; c++) counts[c] = segs[c].m;
		HIP_TRY(ctx, hipMemcpyAsync((uint64_t*)ctx->shard_payload.p + n * L.padded_bins, counts.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, msc_launch_colsum_sums(ctx->stream, L, set->dtype, set->bins, (const uint32_t*)ctx->slots.p, (const MscBatchSeg*)ctx->segs.p, (uint32_t)n,
		                                    (uint64_t*)ctx->shard_payload.p));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		*dev_payload = ctx->shard_payload.p;
		*payload_bytes = elems * 8;
		return MSC_OK;
	}
	// sparse: the summed excesses of each list as a sparse slot whose values are 1 + sum (the sweep of the batched mean with one
	// "member" and 32-bit values), packed behind a table of {members, offset}
	uint32_t* touched = nullptr;
	if ((r = sparse_acc_prepare(ctx, L, (uint32_t)n, &touched))) return r;
	if ((r = sparse_acc_scatter(ctx, set, member_slots, pair_seg.data(), total, touched))) return r;
	std::vector<uint32_t> ones(n, 1);
	if ((r = sparse_acc_sweep(ctx, set, (uint32_t)n, ones.data(), 32, touched, nullptr))) return r;
	const msc_hist_set* ms = ctx->sparse_mean_batch;
	std::vector<uint64_t> table(2 + 2 * n), offs(n);
	uint64_t at = blob_head_bytes(n);
	for (uint64_t c = 0; c < n; c++) { offs[c] = at; table[2 + 2 * c] = segs[c].m; table[3 + 2 * c] = at; at += packed_bytes(ms, c); }
	table[0] = n; table[1] = at;
	void* blob = nullptr;
	{
		// (shard_payload doubles as msc_hist_pack's staging of the heads: the blob lives in its own buffer)
		if ((r = ensure(ctx, ctx->sp_partials, at))) return r;
		blob = ctx->sp_partials.p;
	}
	std::vector<uint32_t> ids(n);
	for (uint64_t c = 0; c < n; c++) ids[c] = (uint32_t)c;
	if ((r = msc_hist_pack(ctx, ms, ids.data(), n, blob, offs.data()))) return r;
	HIP_TRY(ctx, hipMemcpyAsync(blob, table.data(), table.size() * 8, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	*dev_payload = blob;
	*payload_bytes = at;
	return MSC_OK;
}

extern "C" int msc_colsum_nearest(msc_ctx* ctx, const msc_hist_set* set, const uint32_t* member_slots, const uint64_t* offsets, uint64_t n, const void* dev_global,
                                  uint64_t bytes_per_rank, int world, int64_t* nearest_pos, double* nearest_dist, uint64_t* m_total_out) {
	if (!ctx || !set || set->ctx != ctx || !offsets || !dev_global || n == 0 || !nearest_pos || !nearest_dist || world < 1 || (offsets[n] && !member_slots)) return MSC_ERR_INVALID_ARG;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const MscLayout& L = set->L;
	std::vector<MscBatchSeg> segs;
	std::vector<uint32_t> pair_seg;
	uint32_t max_m = 0;
	int r;
	if ((r = lists_of(ctx, set, member_slots, offsets, n, segs, pair_seg, &max_m))) return r;
	const uint64_t total = offsets[n];
	std::vector<uint32_t> members(member_slots, member_slots + total);
	std::vector<uint64_t> m_total(n, 0);
	for (uint64_t c = 0; c < n; c++) { nearest_pos[c] = -1; nearest_dist[c] = 0.0; }
	if (!set->sparse) {
		const uint64_t* g = (const uint64_t*)dev_global;
		HIP_TRY(ctx, hipMemcpyAsync(m_total.data(), g + n * L.padded_bins, n * 8, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		if (m_total_out) memcpy(m_total_out, m_total.data(), n * 8);
		if (total == 0) return MSC_OK;
		if (!ctx->batch_scratch || ctx->batch_scratch->k != set->k || ctx->batch_scratch->dtype != set->dtype || ctx->batch_scratch->capacity < n) {
			if (ctx->batch_scratch) { msc_hist_set_destroy(ctx->batch_scratch); ctx->batch_scratch = nullptr; }
			if ((r = msc_hist_set_create(ctx, set->k, set->dtype, std::max<uint64_t>(n, 16), &ctx->batch_scratch))) return r;
		}
		msc_hist_set* rs = ctx->batch_scratch;
		if ((r = ensure(ctx, ctx->floor_sum, n * 8)) || (r = ensure(ctx, ctx->slots, total * 4)) || (r = ensure(ctx, ctx->pair_seg, total * 4)) ||
		    (r = ensure(ctx, ctx->segs, n * sizeof(MscBatchSeg))) || (r = ensure(ctx, ctx->partials, total * L.S * sizeof(MscPartial))) || (r = ensure(ctx, ctx->dist, total * 8)))
			return r;
		HIP_TRY(ctx, hipMemcpyAsync(ctx->segs.p, segs.data(), n * sizeof(MscBatchSeg), hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->pair_seg.p, pair_seg.data(), total * 4, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->slots.p, members.data(), total * 4, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, msc_launch_mean_from_sums(ctx->stream, L, set->dtype, g, g + n * L.padded_bins, (uint32_t)n, rs->bins, (uint64_t*)ctx->floor_sum.p));
		HIP_TRY(ctx, hipMemsetAsync(rs->scalars, 0, rs->scalar_stride * n, ctx->stream));
		HIP_TRY(ctx, msc_launch_finalize(ctx->stream, rs->bins, rs->scalars, L, set->dtype, 0, n, false));
		HIP_TRY(ctx, msc_launch_pair_tiles_batch(ctx->stream, L, set->dtype, set->bins, set->scalars, (const uint32_t*)ctx->slots.p, (const MscBatchSeg*)ctx->segs.p, (uint32_t)n,
		                                         max_m, rs->bins, rs->L.slot_bytes, rs->scalars, rs->scalar_stride, 0, (MscPartial*)ctx->partials.p, MSC_ORDER_CAND_FIRST));
		HIP_TRY(ctx, msc_launch_distance_batch(ctx->stream, (const MscPartial*)ctx->partials.p, L.S, (uint32_t)total, set->scalars, set->scalar_stride, (const uint32_t*)ctx->slots.p,
		                                       (const uint32_t*)ctx->pair_seg.p, rs->scalars, rs->scalar_stride, (const uint64_t*)ctx->floor_sum.p, (double*)ctx->dist.p));
	} else {
		// the tables of every rank's blob, then all their lists into one scratch set, scatter-added into this rank's accumulators
		const uint64_t hb = blob_head_bytes(n);
		std::vector<uint64_t> tables((size_t)world * (hb / 8));
		for (int w = 0; w < world; w++)
			HIP_TRY(ctx, hipMemcpyAsync(tables.data() + (size_t)w * (hb / 8), (const uint8_t*)dev_global + (uint64_t)w * bytes_per_rank, hb, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		std::vector<uint64_t> offs;
		std::vector<uint32_t> gslots, gseg;
		uint64_t bound = 0;      // entries of all gathered lists: each is at most nbins long, and a rank's blob holds 12 bytes per entry
		for (int w = 0; w < world; w++) {
			const uint64_t* t = tables.data() + (size_t)w * (hb / 8);
			if (t[0] != n || t[1] > bytes_per_rank) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_colsum_nearest: rank %d sent a payload for another set of lists", w);
			bound += t[1] / 12 + 1;
			for (uint64_t c = 0; c < n; c++) {
				m_total[c] += t[2 + 2 * c];
				offs.push_back((uint64_t)w * bytes_per_rank + t[3 + 2 * c]);
				gslots.push_back((uint32_t)((uint64_t)w * n + c));
				gseg.push_back((uint32_t)c);
			}
		}
		if (m_total_out) memcpy(m_total_out, m_total.data(), n * 8);
		msc_hist_set*& gs = ctx->shard_gather;
		if (!gs || gs->k != set->k || gs->dtype != set->dtype || gs->capacity < (uint64_t)world * n || gs->ent_capacity < bound) {
			const uint64_t cap = std::max<uint64_t>((uint64_t)world * n, gs ? gs->capacity : 64), arena = std::max<uint64_t>(bound + bound / 2, gs ? gs->ent_capacity : (1u << 20));
			if (gs) { msc_hist_set_destroy(gs); gs = nullptr; }
			if ((r = msc_hist_set_create_sparse(ctx, set->k, set->dtype, cap, arena, &gs))) return r;
		}
		if ((r = msc_hist_set_reset(ctx, gs))) return r;
		if ((r = msc_hist_unpack(ctx, gs, gslots.data(), gslots.size(), dev_global, offs.data()))) return r;
		uint32_t* touched = nullptr;
		if ((r = sparse_acc_prepare(ctx, L, (uint32_t)n, &touched))) return r;
		if ((r = sparse_acc_scatter(ctx, gs, gslots.data(), gseg.data(), gslots.size(), touched))) return r;
		std::vector<uint32_t> m32(n);
		for (uint64_t c = 0; c < n; c++) {
			if (m_total[c] > 0xffffffffull) return fail(ctx, MSC_ERR_INVALID_ARG, "msc_colsum_nearest: too many members");
			m32[c] = (uint32_t)m_total[c];
		}
		// (every rank sweeps, members of its own or not: the accumulators must be zero again)
		if ((r = sparse_acc_sweep(ctx, set, (uint32_t)n, m32.data(), set->dtype, touched, nullptr))) return r;
		if (total == 0) return MSC_OK;
		if ((r = sparse_distances_to_means(ctx, set, segs, pair_seg, members, (uint32_t)n))) return r;
	}
	std::vector<double> dist(total);
	HIP_TRY(ctx, hipMemcpyAsync(dist.data(), ctx->dist.p, total * 8, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	for (uint64_t c = 0; c < n; c++) {          // first minimum wins (cluster/Trainer.cpp:150-153)
		const MscBatchSeg& sg = segs[c];
		if (sg.m == 0) continue;
		uint32_t best = 0;
		for (uint32_t i = 1; i < sg.m; i++) if (dist[sg.first + i] < dist[sg.first + best]) best = i;
		nearest_pos[c] = best;
		nearest_dist[c] = dist[sg.first + best];
	}
	return MSC_OK;
}
