// msc_driver_capi.cpp -- C entry points of libmsc_driver.so (include/meshclust2_driver.h): the clustering logic of
// msc_driver.hpp behind a table of callbacks, the length-binned store and the host matrix inverse on their own for the CPU fuzz.
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>

#include "../../include/meshclust2_driver.h"
#include "msc_driver.hpp"
#include "msc_hostmath.hpp"

namespace {

struct CallbackError { int code; };

struct CallbackBackend : msc::ClusterBackend {
	const msc_cluster_callbacks& cb;
	const msc_cluster_window_callbacks wcb;          // (a copy: all NULL when the caller has no window callbacks)
	CallbackBackend(const msc_cluster_callbacks& c, const msc_cluster_window_callbacks& w) : cb(c), wcb(w) {}
	static void check(int rc) { if (rc) throw CallbackError{rc}; }
	void get_close(uint32_t q, const std::vector<uint32_t>& window, std::vector<uint8_t>& flags, int64_t& pos, bool& is_min) override {
		flags.assign(window.size(), 0);
		int im = 1;
		pos = -1;
		check(cb.get_close(cb.user, q, window.data(), window.size(), flags.data(), &pos, &im));
		is_min = im != 0;
	}
	int64_t closest(const std::vector<uint32_t>& members) override {
		int64_t pos = -1;
		check(cb.closest(cb.user, members.data(), members.size(), &pos));
		if (pos < 0 || (size_t)pos >= members.size()) throw std::runtime_error("closest callback returned a position outside the member list");
		return pos;
	}
	uint32_t centre_new(uint32_t point) override {
		uint32_t c = 0;
		check(cb.centre_new(cb.user, point, &c));
		return c;
	}
	void centre_set(uint32_t centre, uint32_t point) override { check(cb.centre_set(cb.user, centre, point)); }
	void filter(uint32_t centre, const std::vector<uint32_t>& points, std::vector<uint8_t>& keep) override {
		keep.assign(points.size(), 0);
		check(cb.filter(cb.user, centre, points.data(), points.size(), keep.data()));
	}
	long merge(const std::vector<uint32_t>& centres, long current, long begin, long last) override {
		int64_t best = 0;
		check(cb.merge(cb.user, centres.data(), centres.size(), current, begin, last, &best));
		return (long)best;
	}
	bool update_centres(const std::vector<uint32_t>& centres, const std::vector<uint32_t>& points, const std::vector<uint64_t>& offsets,
	                    std::vector<int64_t>& nearest) override {
		if (!cb.update_centres) return false;
		check(cb.update_centres(cb.user, centres.data(), centres.size(), points.data(), offsets.data(), nearest.data()));
		return true;
	}
	bool centre_set_batch(const std::vector<uint32_t>& centres, const std::vector<uint32_t>& points) override {
		if (!cb.centre_set_batch) return false;
		check(cb.centre_set_batch(cb.user, centres.data(), points.data(), centres.size()));
		return true;
	}
	bool merge_all(const std::vector<uint32_t>& centres, int delta, std::vector<int64_t>& best) override {
		if (!cb.merge_all) return false;
		check(cb.merge_all(cb.user, centres.data(), centres.size(), delta, best.data()));
		return true;
	}
	bool set_order(const std::vector<uint32_t>& order) override {
		if (!wcb.set_order || !wcb.get_close_range || !wcb.kill) return false;
		check(wcb.set_order(cb.user, order.data(), order.size()));
		return true;
	}
	void get_close_range(uint32_t q, uint64_t first, uint64_t end, std::vector<uint32_t>& close, int64_t& best, bool& is_min) override {
		close.assign((size_t)(end - first), 0);
		uint64_t n = 0;
		int im = 1;
		best = -1;
		check(wcb.get_close_range(cb.user, q, first, end, close.data(), &n, &best, &im));
		if (n > close.size()) throw std::runtime_error("get_close_range callback returned more positions than the range holds");
		close.resize((size_t)n);
		is_min = im != 0;
	}
	void kill(uint64_t pos) override { check(wcb.kill(cb.user, pos)); }
};

void put_error(char* err, size_t cap, const std::string& msg) {
	if (!err || cap == 0) return;
	const size_t n = std::min(cap - 1, msg.size());
	memcpy(err, msg.data(), n);
	err[n] = 0;
}

struct Bins {
	std::vector<msc::SeqRecord> recs;
	std::unique_ptr<msc::LengthBins> store;
};

}  // namespace

extern "C" int msc_cluster_run(const msc_cluster_callbacks* cb, uint64_t n, const char* const* headers, const uint64_t* lengths, double similarity,
                               int delta, int iterations, const char* output, const char* log, int batch_update, char* err, size_t cap) {
	return msc_cluster_run_windows(cb, nullptr, n, headers, lengths, similarity, delta, iterations, output, log, batch_update, err, cap);
}

extern "C" int msc_cluster_run_windows(const msc_cluster_callbacks* cb, const msc_cluster_window_callbacks* wcb, uint64_t n, const char* const* headers,
                                       const uint64_t* lengths, double similarity, int delta, int iterations, const char* output, const char* log,
                                       int batch_update, char* err, size_t cap) {
	if (!cb || !cb->get_close || !cb->closest || !cb->centre_new || !cb->centre_set || !cb->filter || !cb->merge || (n && (!headers || !lengths))) {
		put_error(err, cap, "msc_cluster_run: NULL argument or missing callback");
		return -1;
	}
	try {
		std::vector<msc::SeqRecord> recs((size_t)n);
		for (uint64_t i = 0; i < n; i++) { recs[i].header = headers[i]; recs[i].length = lengths[i]; }
		std::ofstream logfile;
		if (log) logfile.open(log);
		CallbackBackend be(*cb, wcb ? *wcb : msc_cluster_window_callbacks{nullptr, nullptr, nullptr});
		msc::MeanShift ms(be, log ? (std::ostream&)logfile : (std::ostream&)std::cout);
		ms.batch_update = batch_update != 0;
		if (n) ms.run(recs, similarity, iterations, delta, output);
		return 0;
	} catch (const CallbackError& e) {
		put_error(err, cap, "a callback failed");
		return e.code;
	} catch (const std::exception& e) {
		put_error(err, cap, e.what());
		return -1;
	}
}

extern "C" void* msc_bins_create(const uint64_t* lengths, uint64_t n, uint64_t per_bin) {
	Bins* b = new Bins();
	b->recs.resize((size_t)n);
	std::vector<uint64_t> l(lengths, lengths + n);
	b->store.reset(new msc::LengthBins(l, per_bin));
	for (uint64_t i = 0; i < n; i++) { b->recs[i].length = lengths[i]; b->recs[i].id = i; b->recs[i].point = (uint32_t)i; b->store->add(&b->recs[i]); }
	b->store->seal();
	return b;
}
extern "C" void msc_bins_destroy(void* bins) { delete (Bins*)bins; }
extern "C" uint64_t msc_bins_count(const void* bins) { return ((const Bins*)bins)->store->bins.size(); }
extern "C" uint64_t msc_bins_layout(const void* bins, uint32_t* ids_out, uint64_t* sizes_out) {
	const msc::LengthBins& s = *((const Bins*)bins)->store;
	uint64_t n = 0;
	for (size_t i = 0; i < s.bins.size(); i++) {
		sizes_out[i] = s.bins[i].size();
		for (const auto& it : s.bins[i]) ids_out[n++] = it.rec->point;
	}
	return n;
}
extern "C" void msc_bins_range(const void* bins, uint64_t begin_len, uint64_t end_len, uint64_t out[5]) {
	const auto r = ((const Bins*)bins)->store->range(begin_len, end_len);
	out[0] = r.first.bin; out[1] = r.first.at; out[2] = r.second.bin; out[3] = r.second.at; out[4] = r.second.none ? 1 : 0;
}
extern "C" int64_t msc_bins_window(const void* bins, uint64_t begin_len, uint64_t end_len, uint32_t* ids_out, uint64_t cap) {
	const msc::LengthBins& s = *((const Bins*)bins)->store;
	const auto r = s.range(begin_len, end_len);
	msc::BinCursor it{r.first.bin, r.first.at, &s.bins};
	const msc::BinCursor end{r.second.bin, r.second.at, &s.bins};
	const int64_t trips = end.distance_from(it);
	for (int64_t n = 0; n < trips; n++) {
		if ((uint64_t)n < cap) ids_out[n] = s.bins.at(it.bin).at(it.at).rec->point;
		if (n + 1 < trips) it.step();
	}
	return trips;
}
extern "C" void msc_bins_mark(void* bins, uint64_t bin, uint64_t at) { ((Bins*)bins)->store->bins.at(bin).at(at).marked = true; }
extern "C" uint64_t msc_bins_take_marked(void* bins, uint64_t begin_len, uint64_t end_len, uint32_t* ids_out) {
	msc::LengthBins& s = *((Bins*)bins)->store;
	const auto r = s.range(begin_len, end_len);
	std::vector<msc::SeqRecord*> out;
	s.take_marked(r.first, r.second, out);
	for (size_t i = 0; i < out.size(); i++) ids_out[i] = out[i]->point;
	return out.size();
}
extern "C" int64_t msc_bins_take_first(void* bins) {
	msc::SeqRecord* r = ((Bins*)bins)->store->take_first();
	return r ? (int64_t)r->point : -1;
}
extern "C" void msc_bins_erase(void* bins, uint64_t bin, uint64_t at) { ((Bins*)bins)->store->erase(bin, at); }

extern "C" void msc_host_inverse(uint64_t n, const double* a, double* out) {
	msc::hostmath::Matrix m((size_t)n, (size_t)n);
	for (size_t i = 0; i < (size_t)(n * n); i++) m.v[i] = a[i];
	const msc::hostmath::Matrix r = msc::hostmath::inverse(m);
	for (size_t i = 0; i < (size_t)(n * n); i++) out[i] = r.v[i];
}
