// tests/sharded_oracle_main.cpp -- TEST INFRASTRUCTURE: the sharded mean-shift driver (meshclust2_amd/host/msc_sharded.hpp over
// msc_driver.hpp) with the CPU ORACLE as every rank's local scorer and plain sockets between the ranks, so that the N > 1 path --
// ownership, position windows, the get_close records, the column-sum reduction of get_mean, the chunked update round, the packed
// centre exchange -- runs and is checked WITHOUT a GPU. Started once per rank by tests/test_cluster_ranks.py with RANK / WORLD_SIZE /
// MASTER_ADDR / MASTER_PORT in the environment; rank 0 writes the .clstr, which must be the reference CLI's own file.
//   sharded_oracle <input.fa> <weights.txt> <k> <dtype> <similarity> <output.clstr> [block]
// Links oracle/libmsc_oracle.so (the checker); nothing of the product links this.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../meshclust2_amd/host/msc_fasta.hpp"
#include "../meshclust2_amd/host/msc_sharded.hpp"
#include "../oracle/msc_oracle.h"

namespace {

struct PackedHead { uint64_t mag, length, id; double stddev; };
inline uint64_t up16(uint64_t v) { return (v + 15) & ~15ull; }

class OracleShardEngine : public msc::ShardEngine {
public:
	OracleShardEngine(const orc_predictor& pred, double cutoff, int k, int dtype, const std::vector<std::string>& seqs) : pred_(pred), cutoff_(cutoff), k_(k), dtype_(dtype) {
		h_.resize(seqs.size());
		for (size_t i = 0; i < seqs.size(); i++)
			if (orc_hist_build(seqs[i].data(), seqs[i].size(), k, dtype, 0, &h_[i]) != 0) throw std::runtime_error("oracle: invalid sequence");
		nbins_ = 1ull << (2 * k);
		bin_bytes_ = nbins_ * (uint64_t)dtype / 8;
		memset(&q_, 0, sizeof q_);
	}
	uint64_t n_local() const override { return h_.size(); }
	void lengths(std::vector<uint64_t>& out) override { out.clear(); for (const orc_hist& h : h_) out.push_back(h.length); }
	void packed_sizes(std::vector<uint64_t>& out) override { out.assign(h_.size(), sizeof(PackedHead) + up16(bin_bytes_)); }
	bool device_buffers() const override { return false; }
	void* staging(int which, size_t bytes) override { if (bufs_[which].size() < bytes) bufs_[which].resize(bytes); return bufs_[which].data(); }
	void pack_points(const uint32_t* local, size_t n, void* dst, const uint64_t* offsets) override {
		for (size_t i = 0; i < n; i++) {
			const orc_hist& h = h_.at(local[i]);
			uint8_t* o = (uint8_t*)dst + offsets[i];
			const PackedHead ph{h.mag, h.length, h.id, h.stddev};
			memcpy(o, &ph, sizeof ph);
			memcpy(o + sizeof ph, h.bins, bin_bytes_);
		}
	}
	void unpack(const void* src, orc_hist& out, std::vector<uint8_t>& store) {
		PackedHead ph;
		memcpy(&ph, src, sizeof ph);
		store.assign((const uint8_t*)src + sizeof ph, (const uint8_t*)src + sizeof ph + bin_bytes_);
		memset(&out, 0, sizeof out);
		out.dtype = dtype_; out.k = k_; out.nbins = nbins_; out.bins = store.data();
		out.mag = ph.mag; out.length = ph.length; out.id = ph.id; out.stddev = ph.stddev;
	}
	void install_query(const void* packed) override { unpack(packed, q_, q_store_); }

	void set_order(const std::vector<uint32_t>& local_in_order) override { order_ = local_in_order; alive_.assign(order_.size(), 1); }
	void kill(uint32_t index) override { if (!alive_.at(index)) throw std::runtime_error("oracle engine: position killed twice"); alive_[index] = 0; }
	void get_close(uint32_t lo, uint32_t hi, std::vector<uint32_t>& close_idx, int64_t& best_idx, double& best_sim) override {
		std::vector<const orc_hist*> cands;
		std::vector<uint32_t> idx;
		for (uint32_t i = lo; i < hi; i++) if (alive_[i]) { cands.push_back(&h_[order_[i]]); idx.push_back(i); }
		std::vector<uint8_t> flags(cands.size() + 1, 0);
		int64_t bp = -1;
		int im = 1;
		best_sim = -1.0;
		if (orc_get_close(&pred_, cutoff_, &q_, cands.data(), cands.size(), flags.data(), &bp, &best_sim, &im) != 0) throw std::runtime_error("oracle: get_close threw");
		best_idx = bp >= 0 ? (int64_t)idx[(size_t)bp] : -1;
		close_idx.clear();
		for (size_t j = 0; j < cands.size(); j++) if (flags[j]) { close_idx.push_back(idx[j]); alive_[idx[j]] = 0; }
	}

	uint32_t centre_from_query() override {
		centres_.emplace_back();
		if (orc_hist_clone(&q_, &centres_.back()) != 0) throw std::runtime_error("oracle: clone failed");
		return (uint32_t)centres_.size() - 1;
	}
	void centres_assign(const uint32_t* centres, size_t n, const void* packed, const uint64_t* offsets) override {
		for (size_t i = 0; i < n; i++) {
			orc_hist t;
			std::vector<uint8_t> store;
			unpack((const uint8_t*)packed + offsets[i], t, store);
			orc_hist_set(&centres_.at(centres[i]), &t);
		}
	}
	void filter_batch(const uint32_t* centres, size_t n, const uint32_t* local, const uint64_t* offsets, uint8_t* keep) override {
		for (size_t c = 0; c < n; c++) {
			std::vector<const orc_hist*> pts;
			for (uint64_t j = offsets[c]; j < offsets[c + 1]; j++) pts.push_back(&h_.at(local[j]));
			if (!pts.empty()) orc_filter(&pred_, cutoff_, &centres_.at(centres[c]), pts.data(), pts.size(), keep + offsets[c]);
		}
	}
	long merge(const std::vector<uint32_t>& centres, long current, long begin, long last) override {
		std::vector<const orc_hist*> cs;
		for (uint32_t c : centres) cs.push_back(&centres_.at(c));
		return orc_merge(&pred_, cutoff_, cs.data(), cs.size(), current, begin, last);
	}
	void merge_all(const std::vector<uint32_t>& centres, int delta, std::vector<int64_t>& best) override {
		const long n = (long)centres.size();
		for (long i = 0; i < n; i++) best[(size_t)i] = merge(centres, i, i + 1, std::min(n - 1, i + delta));
	}
	bool merge_some(const std::vector<uint32_t>& centres, int delta, const std::vector<uint64_t>& which, std::vector<int64_t>& best) override {
		const long n = (long)centres.size();
		for (size_t w = 0; w < which.size(); w++) { const long i = (long)which[w]; best[w] = merge(centres, i, i + 1, std::min(n - 1, i + delta)); }
		some_calls_++; some_asked_ += which.size(); some_of_ += (uint64_t)n;
		return true;
	}
	uint64_t some_calls_ = 0, some_asked_ = 0, some_of_ = 0;          // (the test reads them: the subset form was taken, and for fewer centres than there are)

	bool colsum_reduces() const override { return true; }
	size_t colsum_list_bytes() const override { return (size_t)nbins_ * 8; }
	uint64_t bin(const orc_hist& h, uint64_t i) const {
		switch (dtype_) { case 8: return ((const uint8_t*)h.bins)[i]; case 16: return ((const uint16_t*)h.bins)[i]; case 32: return ((const uint32_t*)h.bins)[i]; default: return ((const uint64_t*)h.bins)[i]; }
	}
	void colsum_partial(const uint32_t* local, const uint64_t* offsets, size_t n, void*& payload, size_t& bytes) override {
		sums_.assign(n * nbins_ + n, 0);
		for (size_t c = 0; c < n; c++) {
			for (uint64_t j = offsets[c]; j < offsets[c + 1]; j++) {
				const orc_hist& h = h_.at(local[j]);
				for (uint64_t i = 0; i < nbins_; i++) sums_[c * nbins_ + i] += bin(h, i);
			}
			sums_[n * nbins_ + c] = offsets[c + 1] - offsets[c];
		}
		payload = sums_.data();
		bytes = sums_.size() * 8;
	}
	void colsum_nearest(const uint32_t* local, const uint64_t* offsets, size_t n, const void* global, size_t, int, int64_t* pos, double* dist) override {
		const uint64_t* g = (const uint64_t*)global;
		std::vector<double> mean(nbins_);
		for (size_t c = 0; c < n; c++) {
			pos[c] = -1; dist[c] = 0.0;
			const uint64_t m = g[n * nbins_ + c];
			if (m == 0 || offsets[c + 1] == offsets[c]) continue;
			// get_mean: the FP64 sum of the members' bins divided by N (cluster/ClusterFactory.cpp:349-357); integer sums are exact
			for (uint64_t i = 0; i < nbins_; i++) mean[i] = (double)g[c * nbins_ + i] / (double)m;
			for (uint64_t j = offsets[c]; j < offsets[c + 1]; j++) {
				const double d = orc_distance_d(&h_.at(local[j]), mean.data());
				if (pos[c] < 0 || d < dist[c]) { pos[c] = (int64_t)(j - offsets[c]); dist[c] = d; }
			}
		}
	}

private:
	const orc_predictor& pred_;
	double cutoff_;
	int k_, dtype_;
	uint64_t nbins_ = 0, bin_bytes_ = 0;
	std::vector<orc_hist> h_, centres_;
	orc_hist q_;
	std::vector<uint8_t> q_store_;
	std::vector<uint32_t> order_;
	std::vector<uint8_t> alive_;
	std::vector<uint8_t> bufs_[4];
	std::vector<uint64_t> sums_;
};

}  // namespace

int main(int argc, char** argv) {
	if (argc < 7) { std::fprintf(stderr, "usage: %s <input.fa> <weights.txt> <k> <dtype> <similarity> <output.clstr> [block]\n", argv[0]); return 2; }
	try {
		const msc::CommEnv env = msc::CommEnv::from_environment();
		std::vector<std::string> headers, seqs;
		msc::read_fasta(argv[1], headers, seqs);
		const int k = std::atoi(argv[3]), dtype = std::atoi(argv[4]);
		const double sim = std::atof(argv[5]);
		const uint64_t block = argc > 7 ? std::strtoull(argv[7], nullptr, 10) : 1000;
		orc_predictor pred;
		if (orc_predictor_load(argv[2], &pred) != 0) throw std::runtime_error("cannot read the weights file");
		orc_set_threads(std::max(1, orc_max_threads() / std::max(1, env.world)));
		msc::ShardPlan plan;
		plan.n = seqs.size(); plan.block = block; plan.world = env.world;
		std::vector<std::string> own((size_t)plan.count(env.rank));
		for (uint64_t l = 0; l < own.size(); l++) own[(size_t)l] = seqs[(size_t)plan.global(env.rank, l)];
		OracleShardEngine engine(pred, sim, k, dtype, own);
		std::unique_ptr<msc::Comm> comm;
		if (env.world > 1) comm.reset(new msc::TcpComm(env, 120)); else comm.reset(new msc::SelfComm());
		msc::ShardedBackend be(engine, *comm, seqs.size(), block);
		std::vector<msc::SeqRecord> records(seqs.size());
		for (size_t i = 0; i < seqs.size(); i++) { records[i].header = headers[i]; records[i].length = be.lengths()[i]; }
		std::ofstream quiet;
		msc::MeanShift ms(be, env.rank == 0 ? (std::ostream&)std::cout : (std::ostream&)quiet);
		if (std::getenv("MSC_SERIAL_UPDATE")) ms.batch_update = false;
		ms.run(records, sim, 15, 5, env.rank == 0 ? argv[6] : nullptr);
		if (env.rank == 0)
			std::cout << "collectives: broadcast " << comm->calls.broadcast << " all_gather " << comm->calls.all_gather << " all_reduce " << comm->calls.all_reduce << " bytes "
			          << comm->calls.bytes << " | get_close steps " << be.ops.get_close << " collectives " << be.ops.get_close_collectives << " overflow " << be.ops.get_close_overflow
			          << " | closest " << be.ops.closest << " update chunks " << be.ops.update_chunks << " set chunks " << be.ops.set_chunks << std::endl
			          << "merge rounds through merge_some: " << engine.some_calls_ << " asked " << engine.some_asked_ << " of " << engine.some_of_ << std::endl;
		return 0;
	} catch (const std::exception& e) {
		std::fprintf(stderr, "sharded_oracle: %s\n", e.what());
		std::_Exit(1);
	}
}
