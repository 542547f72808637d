// msc_sharded.hpp -- the mean-shift operators over points SHARDED across ranks (one GPU per rank): SURVEY 8(e), BASELINE cfg3 / cfg5.
//
// Every rank runs the clustering logic of msc_driver.hpp on replicated bookkeeping (length bins, cluster lists) -- given the same
// operator results it takes the same decisions, so no bookkeeping is ever exchanged. ShardedBackend is msc::ClusterBackend for that
// setting: the POINT histograms are dealt block-cyclically over the ranks in blocks of 1000 (one bvec bin, cluster/CRunner.cpp:585),
// each rank scores only its own candidates through a ShardEngine (the C ABI on a GPU; the CPU oracle in the tests), the CENTRE
// histograms are replicated, and the results meet in a handful of collectives (msc_comm.hpp: RCCL over xGMI):
//   get_close (one per accumulate step, cluster/ClusterFactory.cpp:566 -> cluster/Trainer.cpp:23-71)
//       1 broadcast  the query histogram as one packed byte range (dense slot, or a sparse list: ~12 KB at 1 kb) -- skipped when it
//                    is already resident on every rank
//       1 all-gather one fixed-size record per rank {n_close, best_sim, best position, the first kCloseInline close positions};
//                    a second all-gather only in a step that closes more than kCloseInline points on some rank
//   closest   (get_mean, :338-380)   the column-sum reduction: dense sets 1 all-reduce of uint64 column sums, sparse sets 2
//                    all-gathers (sizes, summed excess lists); then 1 all-gather of a (distance, member) record per rank
//   update_centres (one mean_shift_update per centre and round, :288-335,639) in CHUNKS of centres: local Trainer::filter of each
//                    rank's own members, one column-sum reduction per chunk, 1 all-gather of (distance, position) records per chunk
//   centre_set_batch (center->set(*next), :328,331) 1 all-gather of the packed new centres per chunk
//   merge / merge_all   none: the centres are replicated
// The windows of get_close are ranges of positions of the sealed length-sorted store (msc_driver.hpp set_order / get_close_range):
// a rank keeps the positions of ITS points and their alive flags on its device (msc_window), so a step costs the host O(close).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <vector>

#include "msc_comm.hpp"
#include "msc_driver.hpp"

namespace msc {

// global point index (position in the input) <-> (rank, local slot), block-cyclic
struct ShardPlan {
	uint64_t n = 0, block = 1000;
	int world = 1;
	int owner(uint64_t g) const { return (int)((g / block) % (uint64_t)world); }
	uint64_t local(uint64_t g) const { return (g / block / (uint64_t)world) * block + g % block; }
	uint64_t global(int rank, uint64_t local) const { return ((local / block) * (uint64_t)world + (uint64_t)rank) * block + local % block; }
	uint64_t count(int rank) const {
		const uint64_t full = n / block, rem = n % block;
		uint64_t c = (full / (uint64_t)world) * block;
		const uint64_t extra = full % (uint64_t)world;
		if ((uint64_t)rank < extra) c += block; else if ((uint64_t)rank == extra) c += rem;
		return c;
	}
};

// The rank-local work. Points are LOCAL slots 0 .. n_local-1; centres are replicated handles the engine hands out.
struct ShardEngine {
	virtual ~ShardEngine() {}
	virtual uint64_t n_local() const = 0;
	virtual void lengths(std::vector<uint64_t>& out) = 0;             // effective length of every own point
	virtual void packed_sizes(std::vector<uint64_t>& out) = 0;        // bytes of every own point as a packed range (multiples of 16)
	virtual bool device_buffers() const = 0;                          // staging() hands out device memory
	virtual void* staging(int which, size_t bytes) = 0;               // growable scratch 0 .. 3 the collectives read and write
	virtual void pack_points(const uint32_t* local, size_t n, void* dst, const uint64_t* offsets) = 0;
	virtual void install_query(const void* packed) = 0;               // the resident query <- a packed range
	// the window over this rank's points: index i = the i-th own point in position order
	virtual void set_order(const std::vector<uint32_t>& local_in_order) = 0;
	virtual void kill(uint32_t index) = 0;
	// Trainer::get_close of the resident query over the alive indices of [lo, hi): close ones (ascending) die
	virtual void get_close(uint32_t lo, uint32_t hi, std::vector<uint32_t>& close_idx, int64_t& best_idx, double& best_sim) = 0;
	// centres
	virtual uint32_t centre_from_query() = 0;                         // Center(c->clone()) of the resident query
	virtual void centres_assign(const uint32_t* centres, size_t n, const void* packed, const uint64_t* offsets) = 0;   // center->set(*next)
	virtual void filter_batch(const uint32_t* centres, size_t n, const uint32_t* local, const uint64_t* offsets, uint8_t* keep) = 0;
	virtual long merge(const std::vector<uint32_t>& centres, long current, long begin, long last) = 0;
	virtual void merge_all(const std::vector<uint32_t>& centres, int delta, std::vector<int64_t>& best) = 0;
	// ... for the centres which[w] only (best[w]); false: the engine has no such form and the driver asks about all of them again
	virtual bool merge_some(const std::vector<uint32_t>&, int, const std::vector<uint64_t>&, std::vector<int64_t>&) { return false; }
	// column sums of n lists of own members (msc_colsum_partial / msc_colsum_nearest)
	virtual bool colsum_reduces() const = 0;                          // true: the payload is all-reduced in place; false: all-gathered
	virtual size_t colsum_list_bytes() const = 0;                     // scratch one list costs (chunk sizing)
	virtual void colsum_partial(const uint32_t* local, const uint64_t* offsets, size_t n, void*& payload, size_t& bytes) = 0;
	virtual void colsum_nearest(const uint32_t* local, const uint64_t* offsets, size_t n, const void* global, size_t bytes_each, int world, int64_t* pos, double* dist) = 0;
};

class ShardedBackend : public ClusterBackend {
public:
	static constexpr uint32_t kCloseInline = 250;      // close positions that ride in the get_close record itself

	struct OpCounts { uint64_t get_close = 0, get_close_collectives = 0, get_close_overflow = 0, closest = 0, update_chunks = 0, set_chunks = 0; } ops;

	ShardedBackend(ShardEngine& e, Comm& c, uint64_t n_total, uint64_t block = 1000) : e_(e), x_(c) {
		plan_.n = n_total; plan_.block = block; plan_.world = c.world;
		if (e_.n_local() != plan_.count(c.rank)) throw std::runtime_error("sharded backend: the engine does not hold this rank's share of the points");
		// effective lengths and packed sizes of ALL points: every rank knows its own, one all-gather each spreads them
		uint64_t n_pad = 0;
		for (int r = 0; r < c.world; r++) n_pad = std::max(n_pad, plan_.count(r));
		std::vector<uint64_t> mine, all;
		for (int what = 0; what < 2; what++) {
			if (what == 0) e_.lengths(mine); else e_.packed_sizes(mine);
			mine.resize((size_t)n_pad, 0);
			all.assign((size_t)n_pad * (size_t)c.world, 0);
			x_.all_gather(mine.data(), n_pad * 8, all.data(), n_pad * 8, false);
			std::vector<uint64_t>& dst = what == 0 ? lengths_ : packed_;
			dst.assign((size_t)n_total, 0);
			for (int r = 0; r < c.world; r++)
				for (uint64_t l = 0; l < plan_.count(r); l++) dst[(size_t)plan_.global(r, l)] = all[(size_t)r * n_pad + l];
		}
		for (uint64_t b : packed_) max_packed_ = std::max(max_packed_, b);
	}
	const std::vector<uint64_t>& lengths() const { return lengths_; }

	// ---------------------------------------------------------------- the window (msc_driver.hpp)
	bool set_order(const std::vector<uint32_t>& order) override {
		order_ = order;
		own_pos_.clear();
		std::vector<uint32_t> local;
		for (size_t p = 0; p < order.size(); p++)
			if (plan_.owner(order[p]) == x_.rank) { own_pos_.push_back((uint32_t)p); local.push_back((uint32_t)plan_.local(order[p])); }
		e_.set_order(local);
		return true;
	}
	void kill(uint64_t pos) override {
		if (plan_.owner(order_.at((size_t)pos)) != x_.rank) return;
		e_.kill((uint32_t)(std::lower_bound(own_pos_.begin(), own_pos_.end(), (uint32_t)pos) - own_pos_.begin()));
	}
	void get_close_range(uint32_t q, uint64_t first, uint64_t end, std::vector<uint32_t>& close, int64_t& best, bool& is_min) override {
		ops.get_close++;
		const CommCalls before = x_.calls;
		make_resident(q);
		const uint32_t lo = (uint32_t)(std::lower_bound(own_pos_.begin(), own_pos_.end(), (uint32_t)first) - own_pos_.begin());
		const uint32_t hi = (uint32_t)(std::lower_bound(own_pos_.begin(), own_pos_.end(), (uint32_t)std::min<uint64_t>(end, 0xffffffffull)) - own_pos_.begin());
		int64_t bi = -1;
		double bs = -1.0;
		idx_.clear();
		if (hi > lo) e_.get_close(lo, hi, idx_, bi, bs);
		// one record per rank: {n_close, best_sim, best position, the first kCloseInline close positions}
		struct Rec { uint64_t n_close; double best_sim; int64_t best_pos; uint32_t close[kCloseInline]; };
		Rec mine;
		memset(&mine, 0, sizeof mine);
		mine.n_close = idx_.size();
		mine.best_sim = bs;
		mine.best_pos = bi >= 0 ? (int64_t)own_pos_[(size_t)bi] : -1;
		for (size_t i = 0; i < idx_.size() && i < kCloseInline; i++) mine.close[i] = own_pos_[idx_[i]];
		recs_.resize(sizeof(Rec) * (size_t)x_.world);
		x_.all_gather(&mine, sizeof mine, recs_.data(), sizeof mine, false);
		const Rec* rec = (const Rec*)recs_.data();
		// Trainer::get_close keeps the FIRST maximum in window order (strict '>' at one thread, cluster/Trainer.cpp:26-37,59)
		best = -1;
		double best_sim = -1.0;
		uint64_t total = 0, most = 0;
		for (int r = 0; r < x_.world; r++) {
			total += rec[r].n_close;
			most = std::max(most, rec[r].n_close);
			if (rec[r].best_pos >= 0 && (best < 0 || rec[r].best_sim > best_sim || (rec[r].best_sim == best_sim && rec[r].best_pos < best))) { best = rec[r].best_pos; best_sim = rec[r].best_sim; }
		}
		close.clear();
		if (most > kCloseInline) {          // a step that closes a great many points: the full lists in a second all-gather
			ops.get_close_overflow++;
			std::vector<uint32_t> mine_all((size_t)most, 0), all((size_t)most * (size_t)x_.world);
			for (size_t i = 0; i < idx_.size(); i++) mine_all[i] = own_pos_[idx_[i]];
			x_.all_gather(mine_all.data(), most * 4, all.data(), most * 4, false);
			for (int r = 0; r < x_.world; r++) close.insert(close.end(), all.begin() + (size_t)r * most, all.begin() + (size_t)r * most + (size_t)rec[r].n_close);
		} else {
			for (int r = 0; r < x_.world; r++) close.insert(close.end(), rec[r].close, rec[r].close + rec[r].n_close);
		}
		std::sort(close.begin(), close.end());
		is_min = total == 0;
		ops.get_close_collectives += (x_.calls.broadcast - before.broadcast) + (x_.calls.all_gather - before.all_gather) + (x_.calls.all_reduce - before.all_reduce);
	}
	// the slot-list form is never asked for once set_order returned true
	void get_close(uint32_t, const std::vector<uint32_t>&, std::vector<uint8_t>&, int64_t&, bool&) override {
		throw std::runtime_error("sharded backend: get_close is served through position ranges");
	}

	// ---------------------------------------------------------------- get_mean
	int64_t closest(const std::vector<uint32_t>& members) override {
		if (members.size() == 1) return 0;
		ops.closest++;
		std::vector<uint32_t> local, where;
		for (size_t i = 0; i < members.size(); i++) if (plan_.owner(members[i]) == x_.rank) { local.push_back((uint32_t)plan_.local(members[i])); where.push_back((uint32_t)i); }
		const uint64_t offsets[2] = {0, local.size()};
		int64_t pos = -1;
		double dist = 0.0;
		column_sums(local.data(), offsets, 1, &pos, &dist);
		struct Rec { double dist; int64_t member; };
		const Rec mine{dist, pos >= 0 ? (int64_t)where[(size_t)pos] : -1};
		const std::vector<Rec> all = x_.gather_values(mine);
		int64_t best = -1;
		double bd = 0.0;
		for (const Rec& r : all)          // first minimum in member order (cluster/Trainer.cpp:150-153)
			if (r.member >= 0 && (best < 0 || r.dist < bd || (r.dist == bd && r.member < best))) { best = r.member; bd = r.dist; }
		if (best < 0) throw std::runtime_error("sharded backend: no rank holds a member of this cluster");
		return best;
	}

	// ---------------------------------------------------------------- centres
	uint32_t centre_new(uint32_t point) override {
		make_resident(point);
		return e_.centre_from_query();
	}
	void centre_set(uint32_t centre, uint32_t point) override { centre_set_batch(std::vector<uint32_t>{centre}, std::vector<uint32_t>{point}); }
	bool centre_set_batch(const std::vector<uint32_t>& centres, const std::vector<uint32_t>& points) override {
		const size_t budget = (size_t)1 << 30;      // bytes of packed centres a rank contributes per all-gather
		size_t i0 = 0;
		while (i0 < centres.size()) {
			std::vector<uint64_t> per_rank((size_t)x_.world, 0);
			size_t i1 = i0;
			while (i1 < centres.size()) {
				const int ow = plan_.owner(points[i1]);
				if (i1 > i0 && per_rank[(size_t)ow] + packed_[points[i1]] > budget) break;
				per_rank[(size_t)ow] += packed_[points[i1]];
				i1++;
			}
			uint64_t each = 16;
			for (uint64_t b : per_rank) each = std::max(each, b);
			// where every point of the chunk lands in the gathered buffer; this rank packs its own into its range
			std::vector<uint64_t> at((size_t)x_.world, 0), offs(i1 - i0), my_offs;
			std::vector<uint32_t> my_local;
			for (size_t i = i0; i < i1; i++) {
				const int ow = plan_.owner(points[i]);
				offs[i - i0] = (uint64_t)ow * each + at[(size_t)ow];
				if (ow == x_.rank) { my_local.push_back((uint32_t)plan_.local(points[i])); my_offs.push_back(offs[i - i0]); }
				at[(size_t)ow] += packed_[points[i]];
			}
			uint8_t* buf = (uint8_t*)e_.staging(1, (size_t)each * (size_t)x_.world);
			if (!my_local.empty()) e_.pack_points(my_local.data(), my_local.size(), buf, my_offs.data());
			x_.all_gather(buf + (size_t)x_.rank * each, per_rank[(size_t)x_.rank], buf, each, e_.device_buffers());
			e_.centres_assign(centres.data() + i0, i1 - i0, buf, offs.data());
			ops.set_chunks++;
			i0 = i1;
		}
		return true;
	}
	void filter(uint32_t centre, const std::vector<uint32_t>& points, std::vector<uint8_t>& keep) override {
		std::vector<uint32_t> local, where;
		for (size_t i = 0; i < points.size(); i++) if (plan_.owner(points[i]) == x_.rank) { local.push_back((uint32_t)plan_.local(points[i])); where.push_back((uint32_t)i); }
		const uint64_t offsets[2] = {0, local.size()};
		std::vector<uint8_t> k(local.size() + 1, 0);
		if (!local.empty()) e_.filter_batch(&centre, 1, local.data(), offsets, k.data());
		std::vector<uint8_t> mine(points.size(), 0), all(points.size() * (size_t)x_.world);
		for (size_t j = 0; j < local.size(); j++) mine[where[j]] = k[j];
		x_.all_gather(mine.data(), mine.size(), all.data(), mine.size(), false);
		keep.assign(points.size(), 0);
		for (int r = 0; r < x_.world; r++) for (size_t i = 0; i < points.size(); i++) keep[i] |= all[(size_t)r * points.size() + i];
	}
	long merge(const std::vector<uint32_t>& centres, long current, long begin, long last) override { return e_.merge(centres, current, begin, last); }
	bool merge_all(const std::vector<uint32_t>& centres, int delta, std::vector<int64_t>& best) override { e_.merge_all(centres, delta, best); return true; }
	bool merge_some(const std::vector<uint32_t>& centres, int delta, const std::vector<uint64_t>& which, std::vector<int64_t>& best) override {
		return e_.merge_some(centres, delta, which, best);          // (every rank holds every centre and asks the same questions: no exchange)
	}

	// one update round: mean_shift_update of every centre (cluster/ClusterFactory.cpp:288-335,639), in chunks of centres
	bool update_centres(const std::vector<uint32_t>& centres, const std::vector<uint32_t>& flat, const std::vector<uint64_t>& offsets, std::vector<int64_t>& nearest) override {
		const size_t n = centres.size();
		const size_t per_list = std::max<size_t>(1, e_.colsum_list_bytes());
		const size_t max_lists = std::max<size_t>(1, std::min<size_t>(4096, ((size_t)2 << 30) / per_list));
		std::vector<uint32_t> local, where, kept_local, kept_where;
		std::vector<uint64_t> loffs, koffs;
		std::vector<uint8_t> keep;
		for (size_t c0 = 0; c0 < n;) {
			const size_t c1 = std::min(n, c0 + max_lists), nc = c1 - c0;
			// this rank's members of every list of the chunk, with their positions inside the list
			local.clear(); where.clear(); loffs.assign(1, 0);
			for (size_t c = c0; c < c1; c++) {
				for (uint64_t i = offsets[c]; i < offsets[c + 1]; i++)
					if (plan_.owner(flat[(size_t)i]) == x_.rank) { local.push_back((uint32_t)plan_.local(flat[(size_t)i])); where.push_back((uint32_t)(i - offsets[c])); }
				loffs.push_back(local.size());
			}
			keep.assign(local.size() + 1, 0);
			if (!local.empty()) e_.filter_batch(centres.data() + c0, nc, local.data(), loffs.data(), keep.data());
			kept_local.clear(); kept_where.clear(); koffs.assign(1, 0);
			for (size_t c = 0; c < nc; c++) {
				for (uint64_t j = loffs[c]; j < loffs[c + 1]; j++) if (keep[(size_t)j]) { kept_local.push_back(local[(size_t)j]); kept_where.push_back(where[(size_t)j]); }
				koffs.push_back(kept_local.size());
			}
			std::vector<int64_t> pos(nc, -1);
			std::vector<double> dist(nc, 0.0);
			column_sums(kept_local.data(), koffs.data(), nc, pos.data(), dist.data());
			struct Rec { double dist; int64_t at; };
			std::vector<Rec> mine(nc), all(nc * (size_t)x_.world);
			for (size_t c = 0; c < nc; c++) mine[c] = Rec{dist[c], pos[c] >= 0 ? (int64_t)kept_where[(size_t)(koffs[c] + (uint64_t)pos[c])] : -1};
			x_.all_gather(mine.data(), nc * sizeof(Rec), all.data(), nc * sizeof(Rec), false);
			for (size_t c = 0; c < nc; c++) {          // first minimum in list order
				int64_t best = -1;
				double bd = 0.0;
				for (int r = 0; r < x_.world; r++) {
					const Rec& rr = all[(size_t)r * nc + c];
					if (rr.at >= 0 && (best < 0 || rr.dist < bd || (rr.dist == bd && rr.at < best))) { best = rr.at; bd = rr.dist; }
				}
				nearest[c0 + c] = best;
			}
			ops.update_chunks++;
			c0 = c1;
		}
		return true;
	}

private:
	// the query of get_close / the source of a new centre on every rank: broadcast from its owner unless it is already there
	void make_resident(uint32_t point) {
		if (resident_ == (int64_t)point) return;
		const int ow = plan_.owner(point);
		void* buf = e_.staging(0, (size_t)max_packed_);
		if (ow == x_.rank) {
			const uint32_t l = (uint32_t)plan_.local(point);
			const uint64_t zero = 0;
			e_.pack_points(&l, 1, buf, &zero);
		}
		x_.broadcast(buf, (size_t)packed_[point], ow, e_.device_buffers());
		e_.install_query(buf);
		resident_ = point;
	}

	// the column-sum reduction of n lists of own members + each list's own nearest member
	void column_sums(const uint32_t* local, const uint64_t* offsets, size_t n, int64_t* pos, double* dist) {
		void* payload = nullptr;
		size_t bytes = 0;
		e_.colsum_partial(local, offsets, n, payload, bytes);
		if (e_.colsum_reduces()) {
			x_.all_reduce_sum_u64(payload, bytes / 8, e_.device_buffers());
			e_.colsum_nearest(local, offsets, n, payload, bytes, x_.world, pos, dist);
		} else {
			const uint64_t each = (x_.max_u64(bytes) + 15) & ~15ull;
			uint8_t* all = (uint8_t*)e_.staging(2, (size_t)each * (size_t)x_.world);
			x_.all_gather(payload, bytes, all, (size_t)each, e_.device_buffers());
			e_.colsum_nearest(local, offsets, n, all, (size_t)each, x_.world, pos, dist);
		}
	}

	ShardEngine& e_;
	Comm& x_;
	ShardPlan plan_;
	std::vector<uint64_t> lengths_, packed_;
	uint64_t max_packed_ = 16;
	std::vector<uint32_t> order_, own_pos_, idx_;
	std::vector<uint8_t> recs_;
	int64_t resident_ = -1;
};

}  // namespace msc
