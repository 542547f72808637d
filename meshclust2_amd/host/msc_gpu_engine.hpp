// msc_gpu_engine.hpp -- msc::ShardEngine over the C ABI: one rank's share of the points on one GPU (msc_sharded.hpp).
//   points   this rank's histograms, dense or sparse (built here from its sequences: Loader::get_point, clutil/Loader.cpp:112-179)
//   qset     one slot: the resident query of get_close, installed from the packed range the owner broadcast
//   centres  the replicated centre store (Center(c->clone()), center->set(*next): cluster/Center.h:13-40, ClusterFactory.cpp:328,331)
//   stage    where gathered packed centres land before msc_hist_assign_batch applies set() semantics (stale magnitude kept)
//   window   msc_window over the rank's own points in position order
// Every buffer a collective touches is device memory of this GPU (msc_device_malloc), so RCCL moves it without a host bounce.
#pragma once
#include <cstdlib>
#include <iostream>
#include <memory>

#include "meshclust2_host.hpp"
#include "msc_sharded.hpp"

namespace msc {

class GpuShardEngine : public ShardEngine {
public:
	// seqs = this rank's sequences in local order; longest / total_global = longest sequence and total bases of the WHOLE input
	GpuShardEngine(Context& ctx, Trainer& trn, int k, int dtype, bool sparse, double cutoff, const std::vector<std::string>& seqs, uint64_t longest, uint64_t total_global,
	               uint64_t n_total)
	    : ctx_(ctx), trn_(trn), k_(k), dtype_(dtype), sparse_(sparse), cutoff_(cutoff), n_(seqs.size()), longest_(longest) {
		uint64_t own_bases = 0;
		for (const std::string& s : seqs) own_bases += s.size();
		points_.reset(new PointSet(ctx, k, dtype, std::max<uint64_t>(n_, 1), sparse ? own_bases + 1024 : 0));
		for (size_t off = 0; off < seqs.size(); off += 8192) points_->get_points(off, seqs.data() + off, std::min(seqs.size(), off + 8192) - off);
		qset_.reset(new PointSet(ctx, k, dtype, 1, sparse ? longest + 1024 : 0));
		// sparse centre store: a slot is a header and a scalar record, so room for every point as its own centre costs little; its arena
		// takes every sequence's list twice (a round appends the new list of every moved centre before the old ones are compacted away)
		centre_arena_ = sparse ? 2 * total_global + 64 * longest + (1 << 20) : 0;
		if (const char* e = std::getenv("MSC_CLUSTER_CENTRE_ARENA")) {      // (tests: a snug arena, so that a small run compacts its store)
			if (sparse && std::atoll(e) > 0) centre_arena_ = (uint64_t)std::atoll(e);
		}
		centres_.reset(new PointSet(ctx, k, dtype, sparse ? std::max<uint64_t>(256, n_total) : 256, centre_arena_));
	}
	~GpuShardEngine() override {
		if (std::getenv("MSC_CLUSTER_PROFILE")) std::cout << "centre store: rebuilt " << n_rebuilds_ << " times" << std::endl;
		msc_window_destroy(win_);
		for (void* p : bufs_) if (p) (void)msc_device_free(ctx_.get(), p);
	}
	// how a Comm reaches device memory of this GPU
	void attach(Comm& c) {
		msc_ctx* h = ctx_.get();
		Context* cx = &ctx_;
		c.to_host = [h, cx](void* d, const void* s, size_t b) { cx->check(msc_memcpy_to_host(h, d, s, b)); };
		c.to_device = [h, cx](void* d, const void* s, size_t b) { cx->check(msc_memcpy_to_device(h, d, s, b)); };
		c.on_device = [h, cx](void* d, const void* s, size_t b) { cx->check(msc_memcpy_device(h, d, s, b)); };
	}

	uint64_t n_local() const override { return n_; }
	void lengths(std::vector<uint64_t>& out) override { out = n_ ? points_->get_lengths(0, n_) : std::vector<uint64_t>(); }
	void packed_sizes(std::vector<uint64_t>& out) override {
		out.resize((size_t)n_);
		for (uint64_t i = 0; i < n_; i++) out[(size_t)i] = msc_hist_packed_bytes(points_->get(), i);
	}
	bool device_buffers() const override { return true; }
	void* staging(int which, size_t bytes) override {
		if (bytes > caps_[which]) {
			if (bufs_[which]) ctx_.check(msc_device_free(ctx_.get(), bufs_[which]));
			bufs_[which] = nullptr;
			caps_[which] = bytes + bytes / 2 + 4096;
			ctx_.check(msc_device_malloc(ctx_.get(), caps_[which], &bufs_[which]));
		}
		return bufs_[which];
	}
	void pack_points(const uint32_t* local, size_t n, void* dst, const uint64_t* offsets) override { ctx_.check(msc_hist_pack(ctx_.get(), points_->get(), local, n, dst, offsets)); }
	void install_query(const void* packed) override {
		const uint32_t slot = 0;
		const uint64_t off = 0;
		ctx_.check(msc_hist_set_reset(ctx_.get(), qset_->get()));
		ctx_.check(msc_hist_unpack(ctx_.get(), qset_->get(), &slot, 1, packed, &off));
	}

	void set_order(const std::vector<uint32_t>& local_in_order) override {
		msc_window_destroy(win_);
		win_ = nullptr;
		ctx_.check(msc_window_create(ctx_.get(), points_->get(), local_in_order.data(), local_in_order.size(), &win_));
	}
	void kill(uint32_t index) override { ctx_.check(msc_window_kill(ctx_.get(), win_, &index, 1)); }
	void get_close(uint32_t lo, uint32_t hi, std::vector<uint32_t>& close_idx, int64_t& best_idx, double& best_sim) override {
		const uint32_t* list = nullptr;
		uint64_t n = 0;
		int im = 1;
		ctx_.check(msc_get_close_window(ctx_.get(), trn_.feature().get(), cutoff_, win_, lo, hi, qset_->get(), 0, &list, &n, &best_idx, &best_sim, &im));
		close_idx.assign(list, list + n);
	}

	uint32_t centre_from_query() override {
		if (n_centres_ == centres_->capacity()) rebuild_centres(centres_->capacity() * 2);
		const uint32_t slot = (uint32_t)n_centres_++;
		with_arena_retry([&] { centres_->clone(slot, *qset_, 0); });
		return slot;
	}
	void centres_assign(const uint32_t* centres, size_t n, const void* packed, const uint64_t* offsets) override {
		if (n == 0) return;
		uint64_t hi = 0;
		for (size_t i = 0; i < n; i++) hi = std::max(hi, offsets[i]);
		const uint64_t need_entries = sparse_ ? (hi + 16 + 12 * (longest_ + 64)) / 8 + n : 0;      // (a packed list is >= 12 bytes per entry)
		if (!stage_ || stage_->capacity() < n || stage_entries_ < need_entries) {
			stage_.reset();
			stage_entries_ = std::max(need_entries, stage_entries_);
			stage_.reset(new PointSet(ctx_, k_, dtype_, std::max<uint64_t>(n, 256), stage_entries_));
		}
		ids_.resize(n);
		for (size_t i = 0; i < n; i++) ids_[i] = (uint32_t)i;
		ctx_.check(msc_hist_set_reset(ctx_.get(), stage_->get()));
		ctx_.check(msc_hist_unpack(ctx_.get(), stage_->get(), ids_.data(), n, packed, offsets));
		with_arena_retry([&] { ctx_.check(msc_hist_assign_batch(ctx_.get(), centres_->get(), centres, stage_->get(), ids_.data(), n)); });
	}
	void filter_batch(const uint32_t* centres, size_t n, const uint32_t* local, const uint64_t* offsets, uint8_t* keep) override {
		ctx_.check(msc_filter_batch(ctx_.get(), trn_.feature().get(), cutoff_, centres_->get(), centres, n, points_->get(), local, offsets, keep));
	}
	long merge(const std::vector<uint32_t>& centres, long current, long begin, long last) override { return trn_.merge(*centres_, centres, current, begin, last); }
	void merge_all(const std::vector<uint32_t>& centres, int delta, std::vector<int64_t>& best) override {
		ctx_.check(msc_merge_all(ctx_.get(), trn_.feature().get(), cutoff_, centres_->get(), centres.data(), centres.size(), delta, best.data()));
	}
	bool merge_some(const std::vector<uint32_t>& centres, int delta, const std::vector<uint64_t>& which, std::vector<int64_t>& best) override {
		ctx_.check(msc_merge_some(ctx_.get(), trn_.feature().get(), cutoff_, centres_->get(), centres.data(), centres.size(), delta, which.data(), which.size(), best.data()));
		return true;
	}

	bool colsum_reduces() const override { return !sparse_; }
	size_t colsum_list_bytes() const override { return (size_t)msc_colsum_list_bytes(points_->get()); }
	void colsum_partial(const uint32_t* local, const uint64_t* offsets, size_t n, void*& payload, size_t& bytes) override {
		uint64_t b = 0;
		ctx_.check(msc_colsum_partial(ctx_.get(), points_->get(), local, offsets, n, &payload, &b));
		bytes = (size_t)b;
	}
	void colsum_nearest(const uint32_t* local, const uint64_t* offsets, size_t n, const void* global, size_t bytes_each, int world, int64_t* pos, double* dist) override {
		ctx_.check(msc_colsum_nearest(ctx_.get(), points_->get(), local, offsets, n, global, bytes_each, world, pos, dist, nullptr));
	}

private:
	// relocate every live centre into a fresh store: grows the slot count; for the sparse layout it compacts the append-only arena
	// (a compaction goes back and forth between two standing stores, as in msc_cluster.cpp: no arena is allocated or freed for it)
	void rebuild_centres(uint64_t capacity) {
		std::unique_ptr<PointSet> fresh;
		if (centre_arena_ && spare_ && spare_->capacity() == capacity) { fresh.swap(spare_); fresh->clear(); }
		else { spare_.reset(); fresh.reset(new PointSet(ctx_, k_, dtype_, capacity, centre_arena_)); }
		std::vector<uint32_t> all((size_t)n_centres_);
		for (size_t i = 0; i < all.size(); i++) all[i] = (uint32_t)i;
		if (!all.empty()) fresh->copy_batch(all, *centres_, all);
		centres_.swap(fresh);
		if (centre_arena_ && centres_->capacity() == fresh->capacity()) spare_.swap(fresh);
		n_rebuilds_++;
	}
	template <class F> void with_arena_retry(F&& f) {
		try { f(); }
		catch (const Error& e) {
			if (e.code != MSC_ERR_OOM || centre_arena_ == 0) throw;
			rebuild_centres(centres_->capacity());
			f();
		}
	}

	Context& ctx_;
	Trainer& trn_;
	int k_, dtype_;
	bool sparse_;
	double cutoff_;
	uint64_t n_, longest_;
	std::unique_ptr<PointSet> points_, qset_, centres_, stage_;
	uint64_t n_centres_ = 0, centre_arena_ = 0, stage_entries_ = 0, n_rebuilds_ = 0;
	std::unique_ptr<PointSet> spare_;
	msc_window* win_ = nullptr;
	void* bufs_[4] = {nullptr, nullptr, nullptr, nullptr};
	size_t caps_[4] = {0, 0, 0, 0};
	std::vector<uint32_t> ids_;
};

}  // namespace msc
