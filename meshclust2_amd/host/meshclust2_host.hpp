// meshclust2_host.hpp -- C++ host-side mirror of the reference's classes for the hot path, over the C ABI.
//
// The reference reaches the path through template member calls:
//   Loader<T>::get_point            clutil/Loader.cpp:112-179
//   Feature<T>::compute / operator() predict/Feature.h:197-239
//   Trainer<T>::get_close / filter / merge / closest   cluster/Trainer.cpp:23-157
//   Predictor<T>::close / similarity predict/Predictor.cpp:255-333
// This header gives a maintainer the same names and argument meaning with device-resident points: a Point is a
// (set, slot) handle instead of a heap DivergencePoint<T>*, errors are C++ exceptions again (msc::Error carries the
// msc_status the ABI returned), and T is the reference's run-time datatype (8/16/32/64) instead of a template
// parameter. Header-only; link against libmeshclust2_hip.so.
#pragma once
#include <cstdint>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/meshclust2_hip.h"

namespace msc {

struct Error : std::runtime_error {
	int code;
	Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

class Context {
public:
	explicit Context(int device = 0) {
		int rc = msc_create(device, &h_);
		if (rc != MSC_OK) throw Error(rc, msc_last_error(nullptr));
	}
	~Context() { msc_destroy(h_); }
	Context(const Context&) = delete;
	Context& operator=(const Context&) = delete;
	msc_ctx* get() const { return h_; }
	void set_kernel_timing(bool on) { check(msc_set_kernel_timing(h_, on ? 1 : 0)); }
	void check(int rc) const { if (rc != MSC_OK) throw Error(rc, msc_last_error(h_)); }
private:
	msc_ctx* h_ = nullptr;
};

// The `points` vector of the reference (cluster/CRunner.cpp:505-544), resident in HBM.
class PointSet {
public:
	// sparse_entries > 0: sparse layout (sorted (bin, value) lists) able to hold that many stored bins in total
	PointSet(Context& ctx, int k, int datatype_bits, uint64_t capacity, uint64_t sparse_entries = 0) : ctx_(ctx) {
		if (sparse_entries) ctx_.check(msc_hist_set_create_sparse(ctx_.get(), k, datatype_bits, capacity, sparse_entries, &h_));
		else ctx_.check(msc_hist_set_create(ctx_.get(), k, datatype_bits, capacity, &h_));
	}
	~PointSet() { msc_hist_set_destroy(h_); }
	PointSet(const PointSet&) = delete;
	PointSet& operator=(const PointSet&) = delete;
	msc_hist_set* get() const { return h_; }
	Context& ctx() const { return ctx_; }
	uint64_t capacity() const { return msc_hist_set_capacity(h_); }
	void clear() { ctx_.check(msc_hist_set_clear(ctx_.get(), h_)); }      // sparse sets: every slot empty, the whole arena free

	// Loader<T>::get_point for a batch. strip = the std::string overload (drops non-ACGT first).
	void get_points(uint64_t first_slot, const std::vector<std::string>& seqs, bool strip = false) { get_points(first_slot, seqs.data(), seqs.size(), strip); }
	// ... n sequences of a longer list, where they lie (no copies)
	void get_points(uint64_t first_slot, const std::string* seqs, size_t n, bool strip = false) {
		std::vector<const char*> p(n);
		std::vector<uint64_t> l(n);
		for (size_t i = 0; i < n; i++) { p[i] = seqs[i].data(); l[i] = seqs[i].size(); }
		ctx_.check(msc_hist_build(ctx_.get(), h_, first_slot, n, p.data(), l.data(), strip ? 1 : 0));
	}
	msc_hist_info info(uint64_t slot) const {
		msc_hist_info i;
		ctx_.check(msc_hist_info_get(ctx_.get(), h_, slot, &i));
		return i;
	}
	uint64_t get_length(uint64_t slot) const { return info(slot).length; }
	std::vector<uint64_t> get_lengths(uint64_t first, uint64_t n) const {
		std::vector<uint64_t> out(n);
		ctx_.check(msc_hist_lengths(ctx_.get(), h_, first, n, out.data()));
		return out;
	}
	template <class T> std::vector<T> get_data(uint64_t slot) const {     // DivergencePoint::points, natural order
		std::vector<T> v((size_t)1 << (2 * msc_hist_set_k(h_)));
		ctx_.check(msc_hist_download(ctx_.get(), h_, slot, v.data()));
		return v;
	}
	// Center(c) takes c->clone(); centre->set(*next) keeps the stale magnitude (clutil/DivergencePoint.cpp:182-190)
	void clone(uint64_t dst, const PointSet& src, uint64_t src_slot) { ctx_.check(msc_hist_clone(ctx_.get(), h_, dst, src.h_, src_slot)); }
	void set(uint64_t dst, const PointSet& src, uint64_t src_slot) { ctx_.check(msc_hist_assign(ctx_.get(), h_, dst, src.h_, src_slot)); }
	void copy(uint64_t dst, const PointSet& src, uint64_t src_slot) { ctx_.check(msc_hist_copy(ctx_.get(), h_, dst, src.h_, src_slot)); }
	void clone_batch(const std::vector<uint32_t>& dst, const PointSet& src, const std::vector<uint32_t>& src_slots) {
		ctx_.check(msc_hist_clone_batch(ctx_.get(), h_, dst.data(), src.h_, src_slots.data(), dst.size()));
	}
	void copy_batch(const std::vector<uint32_t>& dst, const PointSet& src, const std::vector<uint32_t>& src_slots) {
		ctx_.check(msc_hist_copy_batch(ctx_.get(), h_, dst.data(), src.h_, src_slots.data(), dst.size()));
	}
private:
	Context& ctx_;
	msc_hist_set* h_ = nullptr;
};

// Feature<T> (finalised) + the GLM weights of one block of a weights file.
class Feature {
public:
	Feature(Context& ctx, const std::string& weights_file, int block = 0) : ctx_(ctx) {
		ctx_.check(msc_model_load(ctx_.get(), weights_file.c_str(), block, &h_));
	}
	~Feature() { msc_model_destroy(h_); }
	Feature(const Feature&) = delete;
	Feature& operator=(const Feature&) = delete;
	msc_model* get() const { return h_; }
	size_t size() const { return (size_t)msc_model_n_combos(h_); }          // Feature::size()
	int n_singles() const { return msc_model_n_singles(h_); }

	// cache = feat->compute(*a_i, *b) for every candidate a_i: rows of normalised singles (the `cache` vectors)
	std::vector<double> compute(const PointSet& cands, const std::vector<uint32_t>& slots, const PointSet& q, uint64_t q_slot,
	                            int order = MSC_ORDER_CAND_FIRST, std::vector<double>* combos = nullptr,
	                            std::vector<double>* sums = nullptr) const {
		std::vector<double> singles(slots.size() * (size_t)n_singles());
		if (combos) combos->resize(slots.size() * size());
		if (sums) sums->resize(slots.size());
		ctx_.check(msc_score(ctx_.get(), h_, cands.get(), slots.data(), slots.size(), q.get(), q_slot, order, singles.data(),
		                     combos ? combos->data() : nullptr, sums ? sums->data() : nullptr, nullptr));
		return singles;
	}
private:
	Context& ctx_;
	msc_model* h_ = nullptr;
};

// cluster/Trainer.{h,cpp}, scoring half. `cutoff` is Trainer::cutoff as given on the command line (--id).
class Trainer {
public:
	Trainer(Context& ctx, const std::string& weights_file, double cutoff) : ctx_(ctx), feat_(ctx, weights_file, 0), cutoff_(cutoff) {}
	double get_id() const { return cutoff_ > 1 ? cutoff_ / 100.0 : cutoff_; }

	// std::tuple<Point<T>*,double,size_t,size_t> get_close(Point<T>*, bvec_iterator istart, iend, bool& is_min)
	// -> (position in `window` of the arg-max or -1, similarity, close flags); the caller owns the bvec bookkeeping.
	std::tuple<int64_t, double, std::vector<uint8_t>> get_close(const PointSet& points, const std::vector<uint32_t>& window,
	                                                            const PointSet& q, uint64_t q_slot, bool& is_min) const {
		std::vector<uint8_t> flags(window.size());
		int64_t pos = -1;
		double sim = -1;
		int im = 1;
		ctx_.check(msc_get_close(ctx_.get(), feat_.get(), cutoff_, points.get(), window.data(), window.size(), q.get(), q_slot,
		                         flags.data(), &pos, &sim, &im));
		is_min = im != 0;
		return std::make_tuple(pos, sim, std::move(flags));
	}
	// void filter(Point<T>*, vector<pair<Point<T>*,bool>>&): erases the rejected entries
	void filter(const PointSet& centre_set, uint64_t centre, const PointSet& points, std::vector<uint32_t>& vec) const {
		std::vector<uint8_t> keep(vec.size());
		uint64_t n = 0;
		ctx_.check(msc_filter(ctx_.get(), feat_.get(), cutoff_, centre_set.get(), centre, points.get(), vec.data(), vec.size(), keep.data(), &n));
		size_t w = 0;
		for (size_t i = 0; i < vec.size(); i++) if (keep[i]) vec[w++] = vec[i];
		vec.resize(w);
	}
	// long merge(vector<Center<T>>& centers, long current, long begin, long last)
	long merge(const PointSet& centres, const std::vector<uint32_t>& centre_slots, long current, long begin, long last) const {
		int64_t best = 0;
		ctx_.check(msc_merge(ctx_.get(), feat_.get(), cutoff_, centres.get(), centre_slots.data(), centre_slots.size(), current, begin, last, &best));
		return (long)best;
	}
	// Point<T>* closest(Point<double>* mean, vector<...>&) fused with the mean that precedes it (get_mean / mean_shift_update)
	int64_t closest(const PointSet& points, const std::vector<uint32_t>& members) const {
		int64_t pos = -1;
		ctx_.check(msc_mean_nearest(ctx_.get(), points.get(), members.data(), members.size(), &pos, nullptr, nullptr));
		return pos;
	}
	const Feature& feature() const { return feat_; }
private:
	Context& ctx_;
	Feature feat_;
	double cutoff_;
};

// Predictor<T>::close + similarity for one query against a database chunk (fastcar/FC_Runner.cpp:426-471). Like work(), it
// follows the file's mode (predict/Predictor.h:20-21): a `mode: 1` file (classification only: what meshclust2 --dump and
// msc_train_class write) gives similarity 1 for every close pair, a `mode: 2` file (regression only) calls every pair close.
class Predictor {
public:
	Predictor(Context& ctx, const std::string& weights_file) : ctx_(ctx) {
		std::ifstream in(weights_file.c_str());
		std::string tok;
		unsigned mode = 0;
		while (in >> tok) if (tok == "mode:") { in >> mode; break; }
		if (!(mode & 3)) throw Error(MSC_ERR_IO, "weights file " + weights_file + ": no `mode:` line with a classification or regression block");
		if (mode & 1) cls_.reset(new Feature(ctx, weights_file, 0));
		if (mode & 2) reg_.reset(new Feature(ctx, weights_file, 1));
	}
	uint8_t get_mode() const { return (uint8_t)((cls_ ? 1 : 0) | (reg_ ? 2 : 0)); }
	void search(const PointSet& db, const std::vector<uint32_t>& slots, const PointSet& q, uint64_t q_slot, std::vector<uint8_t>& close,
	            std::vector<double>& similarity) const {
		close.resize(slots.size());
		similarity.resize(slots.size());
		ctx_.check(msc_search(ctx_.get(), cls_ ? cls_->get() : nullptr, reg_ ? reg_->get() : nullptr, db.get(), slots.data(), slots.size(), q.get(), q_slot,
		                      close.data(), similarity.data()));
	}
	// the same for a block of queries in ONE pass over the database window (msc_score_multi: each candidate tile fetched from HBM
	// serves 16 queries): close[q * slots.size() + i], similarity likewise. Values are bit-identical to search() per query.
	void search_block(const PointSet& db, const std::vector<uint32_t>& slots, const PointSet& q, const std::vector<uint32_t>& q_slots,
	                  std::vector<uint8_t>& close, std::vector<double>& similarity) const {
		const size_t n = slots.size() * q_slots.size();
		close.assign(n, 1);
		similarity.assign(n, 1.0);
		if (n == 0) return;
		if (cls_)
			ctx_.check(msc_score_multi(ctx_.get(), cls_->get(), db.get(), slots.data(), slots.size(), q.get(), q_slots.data(), q_slots.size(), MSC_ORDER_CAND_FIRST,
			                           nullptr, nullptr, close.data(), 0, nullptr));
		if (reg_) {
			ctx_.check(msc_score_multi(ctx_.get(), reg_->get(), db.get(), slots.data(), slots.size(), q.get(), q_slots.data(), q_slots.size(), MSC_ORDER_CAND_FIRST,
			                           similarity.data(), nullptr, nullptr, 0, nullptr));
			for (double& v : similarity) v = v < 0 ? 0 : (v > 1 ? 1 : v);      // p_predict clamps to [0,1], predict/Predictor.cpp:293-298
		}
	}
private:
	Context& ctx_;
	std::unique_ptr<Feature> cls_, reg_;
};

}  // namespace msc
