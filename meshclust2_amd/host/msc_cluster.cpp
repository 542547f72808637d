// msc_cluster.cpp -- mean-shift driver + CD-HIT .clstr writer over the GPU hot path.
//
// The reference's clustering logic (cluster/ClusterFactory.cpp MS / accumulate / mean_shift_update / merge /
// print_output, cluster/bvec.cpp, cluster/CRunner.cpp do_run) is host control flow AROUND the hot path. It is written
// here from scratch against meshclust2_host.hpp so that "identical CLSTR output" can be produced and checked on the GPU
// box, where only this repository exists. Histograms never leave HBM: the driver handles slots, lengths, flags, scalars.
//
// Behaviour reproduced on purpose (SURVEY.md findings): the exclusive use of an inclusive end index in the scoring
// window (Q6), DivergencePoint::set keeping the stale magnitude of a moved centre (Q7), the OMP_NUM_THREADS=1 order of
// every reduction and of remove_available (Q10), the unstable std::sort orders (same libstdc++, same comparator, same
// input order => same permutation), bvec::insert's "middle of the least-filled bins" rule and the quirks of
// index_of / inner_index_of when a bin is empty.
//
// Usage (flag names are the reference's, cluster/CRunner.cpp:243-477; training is out of scope, so a model is required):
//   msc_cluster <input.fa> [--recover weights.txt] [--id 0.9] [--kmer K] [--datatype 8|16|32|64]   (no --recover: trains first)
//               [--output output.clstr] [--delta 5] [--iterations 15] [--single-file] [--sparse] [--serial-update] [--device 0]
#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>
#include <vector>

#include "meshclust2_host.hpp"

namespace {

struct Pt {                      // host shadow of one Point<T>: everything the clustering logic reads
	std::string header;          // full header line including '>'
	uint64_t length = 0;         // effective length
	uint64_t id = 0;
	uint32_t slot = 0;           // slot in the device point set
};

// ------------------------------------------------------------------ FASTA (nonltr/ChromListMaker.cpp:24-48,117-165)
bool safe_getline(std::istream& is, std::string& t) {
	t.clear();
	std::streambuf* sb = is.rdbuf();
	for (;;) {
		int c = sb->sbumpc();
		if (c == '\n') return true;
		if (c == '\r') { if (sb->sgetc() == '\n') sb->sbumpc(); return true; }
		if (c == std::streambuf::traits_type::eof()) { if (t.empty()) { is.setstate(std::ios::eofbit); return false; } return true; }
		t += (char)c;
	}
}

// single_file (--single-file, nonltr/ChromListMaker.cpp:123-147): the whole file is ONE sequence -- the first header, and
// the records joined by 50 'N'
void read_fasta(const std::string& path, std::vector<std::string>& headers, std::vector<std::string>& seqs, bool single_file) {
	std::ifstream in(path.c_str());
	if (!in) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(1); }
	std::string line;
	bool have = false;
	while (in.good()) {
		if (!safe_getline(in, line)) break;
		if (!line.empty() && line[0] == '>') {
			if (single_file && have) { seqs.back() += std::string(50, 'N'); continue; }
			headers.push_back(line);
			seqs.emplace_back();
			have = true;
		} else if (!line.empty() && (line[0] == ' ' || line[0] == '\t')) {
			continue;
		} else if (have) {
			seqs.back() += line;
		}
	}
}

// ------------------------------------------------------------------ bvec (cluster/bvec.{h,cpp})
struct BIdx { size_t first = 0, second = 0; bool is_empty = false; };
using Entry = std::pair<Pt*, bool>;

class BVec {
public:
	BVec(std::vector<uint64_t> lengths, uint64_t bin_size = 1000) {
		std::sort(lengths.begin(), lengths.end());
		for (uint64_t i = 0; i < lengths.size(); i += bin_size) begin_bounds.push_back(lengths[i]);
		data.resize(begin_bounds.size());
	}
	void insert(Pt* p) {                                   // bvec.cpp:150-184
		size_t front = 0, back = 0;
		index_of(p->length, &front, &back);
		std::vector<size_t> mins;
		size_t minimum = std::numeric_limits<size_t>::max();
		for (size_t i = front; i <= back; i++) {
			size_t sz = data[i].size();
			if (sz < minimum) { minimum = sz; mins.clear(); mins.push_back(i); }
			else if (sz == minimum) mins.push_back(i);
		}
		if (mins.empty()) { std::fprintf(stderr, "error: no bins to insert into\n"); std::exit(1); }
		data.at(mins[mins.size() / 2]).push_back(std::make_pair(p, false));
	}
	void insert_finalize() {                               // bvec.cpp:216-233
		for (auto& bin : data) std::sort(bin.begin(), bin.end(), [](const Entry a, const Entry b) { return a.first->length < b.first->length; });
	}
	Pt* pop() {                                            // bvec.cpp:27-37
		for (auto& bin : data) if (!bin.empty()) { Pt* p = bin[0].first; bin.erase(bin.begin()); return p; }
		return nullptr;
	}
	bool index_of(uint64_t point, size_t* pfront, size_t* pback) const {      // bvec.cpp:123-147
		size_t low = begin_bounds.size() - 1, high = 0;
		for (size_t i = 1; i < begin_bounds.size(); i++) {
			const uint64_t prev = begin_bounds[i - 1];
			if (point >= prev && point < begin_bounds[i]) { low = std::min(low, i - 1); high = std::max(high, i - 1); }
		}
		if (point >= begin_bounds[begin_bounds.size() - 1]) high = std::max(high, begin_bounds.size() - 1);
		if (pfront) *pfront = low;
		if (pback) *pback = high;
		return true;
	}
	bool inner_index_of(uint64_t length, size_t& idx, size_t* pfront, size_t* pback) const {   // bvec.cpp:52-120
		if (data.at(idx).empty()) {
			if (pfront) for (size_t i = 0; i < data.size(); i++) if (!data[i].empty()) { idx = i; *pfront = 0; break; }
			if (pback) for (long i = (long)data.size() - 1; i >= 0; i--) if (!data[i].empty()) { idx = (size_t)i; *pback = 0; break; }
			return true;
		}
		const auto& bin = data[idx];
		size_t front = 0, back = 0;
		size_t low = 0, high = bin.size() - 1;
		if (length < bin[low].first->length && pfront) *pfront = low;
		if (length > bin[high].first->length && pback) *pback = high;
		for (; low <= high;) {
			size_t mid = (low + high) / 2;
			uint64_t d = bin[mid].first->length;
			if (d == length) { front = mid; back = mid; break; }
			else if (length < d) high = mid;
			else low = mid + 1;
			if (low == high) { front = low; back = high; break; }
		}
		if (pfront) {
			for (long i = (long)front; i >= 0 && bin[i].first->length == length; i--) front = (size_t)i;
			*pfront = front;
		}
		if (pback) {
			for (size_t i = back; i < bin.size() && bin[i].first->length == length; i++) back = i;
			*pback = back;
		}
		return true;
	}
	std::pair<BIdx, BIdx> get_range(uint64_t begin_len, uint64_t end_len) const {             // bvec.cpp:261-330
		BIdx front, back;
		back.first = data.size() - 1;
		back.second = data[back.first].size() - 1;
		index_of(begin_len, &front.first, nullptr);
		index_of(end_len, nullptr, &back.first);
		inner_index_of(begin_len, front.first, &front.second, nullptr);
		inner_index_of(end_len, back.first, nullptr, &back.second);
		if (back.first == (size_t)-1 || back.second == (size_t)-1) back.is_empty = true;
		return std::make_pair(front, back);
	}
	void erase(size_t r, size_t c) { data.at(r).erase(data.at(r).begin() + (long)c); }
	void remove_available(BIdx begin, BIdx end, std::vector<Pt*>& available) {                // bvec.cpp:342-384, one thread
		if (begin.is_empty || end.is_empty) return;
		for (size_t i = begin.first; i <= end.first && i < data.size(); i++) {
			for (auto& kv : data[i]) if (kv.second) available.push_back(kv.first);
			data[i].erase(std::remove_if(data[i].begin(), data[i].end(), [](const Entry d) { return d.second; }), data[i].end());
		}
	}
	std::vector<std::vector<Entry>> data;
	std::vector<uint64_t> begin_bounds;
};

// bvec_iterator (cluster/bvec_iterator.{h,cpp})
struct BIter {
	size_t r, c;
	std::vector<std::vector<Entry>>* col;
	void next() {
		if (r != col->size()) {
			if (c + 1 < col->at(r).size()) c++;
			else { r++; c = 0; while (r < col->size() && col->at(r).empty()) r++; }
		} else { std::fprintf(stderr, "tried incrementing null iterator\n"); std::exit(1); }
	}
	bool less(const BIter& o) const { return r < o.r || (r == o.r && c < o.c); }
	// operator- (bvec_iterator.h:57-76): what the reference's `#pragma omp parallel for` uses as the trip count
	int64_t minus(const BIter& rhs) const {
		if (less(rhs)) return -1 * rhs.minus(*this);
		if (r == rhs.r) return (int64_t)(c - rhs.c);
		int64_t sum = (int64_t)c;
		sum += (int64_t)(col->at(rhs.r).size() - rhs.c);
		for (size_t i = rhs.r + 1; i < r; i++) sum += (int64_t)col->at(i).size();
		return sum;
	}
};

// ------------------------------------------------------------------ centres (cluster/Center.h)
struct Centre {
	uint32_t cslot = 0;          // slot in the device centre store (a clone of a point, possibly moved by set())
	std::string header;
	uint64_t id = 0, length = 0;
	std::vector<Pt*> points;
	bool to_delete = false;
};

struct Driver {
	msc::Context& ctx;
	msc::PointSet& points;
	msc::Trainer& trn;
	int k, dtype;
	std::unique_ptr<msc::PointSet> centres;
	uint64_t n_centres = 0;

	uint64_t centre_arena = 0;      // > 0: sparse centre store with that many entries

	Driver(msc::Context& c, msc::PointSet& p, msc::Trainer& t, int k_, int dt, uint64_t sparse_arena) : ctx(c), points(p), trn(t), k(k_), dtype(dt) {
		centre_arena = sparse_arena;
		centres.reset(new msc::PointSet(ctx, k, dtype, 256, centre_arena));
	}
	// relocate every live centre into a fresh store (exact copies: stale mags survive). Used to grow the slot count and,
	// for the sparse layout, to compact the append-only entry arena.
	void rebuild_centres(uint64_t capacity) {
		std::unique_ptr<msc::PointSet> fresh(new msc::PointSet(ctx, k, dtype, capacity, centre_arena));
		for (uint64_t i = 0; i < n_centres; i++) fresh->copy(i, *centres, i);
		centres.swap(fresh);
	}
	uint32_t new_centre_slot() {
		if (n_centres == centres->capacity()) rebuild_centres(centres->capacity() * 2);
		return (uint32_t)n_centres++;
	}
	template <class F> void with_arena_retry(F&& f) {      // sparse store: compact once when the arena runs out
		try { f(); }
		catch (const msc::Error& e) {
			if (e.code != MSC_ERR_OOM || centre_arena == 0) throw;
			rebuild_centres(centres->capacity());
			f();
		}
	}
	// Center(Point* c, pts): center(c->clone())
	Centre make_centre(Pt* c, const std::vector<Pt*>& pts) {
		Centre ce;
		ce.cslot = new_centre_slot();
		with_arena_retry([&] { centres->clone(ce.cslot, points, c->slot); });
		ce.header = c->header; ce.id = c->id; ce.length = c->length;
		ce.points = pts;
		return ce;
	}
	// center->set(*next): bins, length, header, id -- not mag
	void centre_set(Centre& ce, Pt* next) {
		with_arena_retry([&] { centres->set(ce.cslot, points, next->slot); });
		ce.header = next->header; ce.id = next->id; ce.length = next->length;
	}

	static std::vector<uint32_t> slots_of(const std::vector<Pt*>& v) {
		std::vector<uint32_t> s(v.size());
		for (size_t i = 0; i < v.size(); i++) s[i] = v[i]->slot;
		return s;
	}

	// get_mean (cluster/ClusterFactory.cpp:338-380): member nearest to the FP64 mean
	Pt* get_mean(const std::vector<Pt*>& available) {
		if (available.empty()) { std::fprintf(stderr, "N cannot be 0, bad\n"); std::exit(1); }
		int64_t pos = trn.closest(points, slots_of(available));
		return available[(size_t)pos];
	}

	// accumulate (cluster/ClusterFactory.cpp:553-610)
	size_t accumulate(Pt** last_ptr, BVec& bv, std::vector<Centre>& part, double sim) {
		Pt* last = *last_ptr;
		std::vector<Pt*> current = {last};
		bool is_min = false;
		while (!is_min) {
			const uint64_t len = last->length;
			auto bounds = bv.get_range((uint64_t)(len * sim), (uint64_t)(len / sim));
			// the window [iter(first), iter(second)) -- `i < iend` with an inclusive end index
			std::vector<uint32_t> window;
			std::vector<std::pair<size_t, size_t>> where;
			BIter it{bounds.first.first, bounds.first.second, &bv.data}, end{bounds.second.first, bounds.second.second, &bv.data};
			// OpenMP turns `for (i = istart; i < iend; ++i)` into (iend - istart) iterations of istart + n
			const int64_t trips = end.minus(it);
			for (int64_t n = 0; n < trips; n++) {
				window.push_back(bv.data.at(it.r).at(it.c).first->slot);
				where.emplace_back(it.r, it.c);
				if (n + 1 < trips) it.next();
			}
			auto res = trn.get_close(points, window, points, last->slot, is_min);
			const auto& flags = std::get<2>(res);
			for (size_t j = 0; j < flags.size(); j++) if (flags[j]) bv.data[where[j].first][where[j].second].second = true;
			if (is_min) {
				const int64_t pos = std::get<0>(res);
				if (pos < 0) {
					*last_ptr = bv.pop();
				} else {
					*last_ptr = bv.data[where[(size_t)pos].first][where[(size_t)pos].second].first;
					bv.erase(where[(size_t)pos].first, where[(size_t)pos].second);
				}
				std::vector<Pt*> none;
				bv.remove_available(bounds.first, bounds.second, none);
			} else {
				bv.remove_available(bounds.first, bounds.second, current);
				last = get_mean(current);
			}
		}
		part.push_back(make_centre(last, current));
		return current.size();
	}

	// mean_shift_update (cluster/ClusterFactory.cpp:288-335)
	void mean_shift_update(std::vector<Centre>& part, int j, int delta) {
		Centre& ce = part[(size_t)j];
		const int i_begin = std::max(0, j - delta);
		const int i_end = std::min(j + delta, (int)part.size() - 1);
		std::vector<Pt*> good;
		for (int i = i_begin; i <= i_end; i++) for (Pt* p : part[(size_t)i].points) good.push_back(p);
		std::vector<uint32_t> slots = slots_of(good);
		// trn.filter(center, good): keep what classifies close to the centre
		{
			std::vector<uint8_t> keep(slots.size());
			uint64_t n = 0;
			ctx.check(msc_filter(ctx.get(), trn.feature().get(), cutoff, centres->get(), ce.cslot, points.get(), slots.data(), slots.size(), keep.data(), &n));
			std::vector<Pt*> g2;
			for (size_t i = 0; i < good.size(); i++) if (keep[i]) g2.push_back(good[i]);
			good.swap(g2);
		}
		if (!good.empty()) {
			int64_t pos = trn.closest(points, slots_of(good));
			centre_set(ce, good[(size_t)pos]);
		} else if (delta == 0) {
			centre_set(ce, ce.points[0]);
		}
	}

	// the `omp parallel for` over mean_shift_update of one round (cluster/ClusterFactory.cpp:639,648) as ONE call: the centres of a
	// round are independent (each reads its own histogram and the member lists of its neighbours, none of which change during
	// the round), so filter + mean + closest of all of them are batched on the device (msc_update_centres)
	void mean_shift_update_all(std::vector<Centre>& part, int delta) {
		const size_t n = part.size();
		if (n == 0) return;
		std::vector<uint32_t> cslots(n), slots;
		std::vector<uint64_t> offsets(n + 1, 0);
		std::vector<Pt*> good;
		for (size_t j = 0; j < n; j++) {
			cslots[j] = part[j].cslot;
			const int i_begin = std::max(0, (int)j - delta);
			const int i_end = std::min((int)j + delta, (int)n - 1);
			for (int i = i_begin; i <= i_end; i++) for (Pt* p : part[(size_t)i].points) { good.push_back(p); slots.push_back(p->slot); }
			offsets[j + 1] = slots.size();
		}
		std::vector<int64_t> nearest(n, -1);
		ctx.check(msc_update_centres(ctx.get(), trn.feature().get(), cutoff, centres->get(), cslots.data(), n, points.get(), slots.data(), offsets.data(),
		                             nearest.data(), nullptr));
		// center->set(*next) of every centre that moves: one launch for the dense layout (the sparse arena is appended to one by one)
		std::vector<uint32_t> dst, src;
		for (size_t j = 0; j < n; j++) {
			Centre& ce = part[j];
			Pt* next = nearest[j] >= 0 ? good[(size_t)(offsets[j] + (uint64_t)nearest[j])] : (delta == 0 ? ce.points[0] : nullptr);
			if (!next) continue;
			if (centre_arena) { centre_set(ce, next); continue; }
			dst.push_back(ce.cslot);
			src.push_back(next->slot);
			ce.header = next->header; ce.id = next->id; ce.length = next->length;
		}
		if (!dst.empty()) ctx.check(msc_hist_assign_batch(ctx.get(), centres->get(), dst.data(), points.get(), src.data(), dst.size()));
	}

	// merge (cluster/ClusterFactory.cpp:383-401)
	bool merge(std::vector<Centre>& centers, int delta) {
		int num_merge = 0;
		std::vector<uint32_t> cs(centers.size());
		for (size_t c = 0; c < centers.size(); c++) cs[c] = centers[c].cslot;
		// every trn.merge(centers, i, i + 1, min(n - 1, i + delta)) of the loop at once: none of them changes a centre
		std::vector<int64_t> best(centers.size(), 0);
		if (batch_update) ctx.check(msc_merge_all(ctx.get(), trn.feature().get(), cutoff, centres->get(), cs.data(), cs.size(), delta, best.data()));
		for (int i = 0; i < (int)centers.size(); i++) {
			long ret = batch_update ? (long)best[(size_t)i] : trn.merge(*centres, cs, i, i + 1, std::min((int)centers.size() - 1, i + delta));
			if (ret > i) {
				num_merge++;
				auto& to_add = centers[(size_t)ret].points;
				auto& to_del = centers[(size_t)i].points;
				to_add.insert(to_add.end(), to_del.begin(), to_del.end());
				centers[(size_t)i].to_delete = true;
			}
		}
		centers.erase(std::remove_if(centers.begin(), centers.end(), [](const Centre& c) { return c.to_delete; }), centers.end());
		return num_merge > 0;
	}

	// print_output (cluster/ClusterFactory.cpp:404-435)
	static void print_output(const std::string& output, const std::vector<Centre>& partition) {
		std::ofstream ofs(output.c_str());
		int counter = 0;
		for (const auto& cen : partition) {
			if (cen.points.empty()) continue;
			ofs << ">Cluster " << counter << std::endl;
			int pt = 0;
			for (Pt* p : cen.points) {
				ofs << pt << "\t" << p->length << "nt, " << p->header << "... ";
				if (p->id == cen.id) ofs << "*";
				ofs << std::endl;
				pt++;
			}
			counter++;
		}
	}

	// Clock::stamp (clutil/Clock.cpp:12-19): same stage names as the reference's driver
	static void stamp(const char* desc) {
		static const auto t0 = std::chrono::steady_clock::now();
		std::cout << "timestamp " << desc << " " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << std::endl;
	}

	// ClusterFactory<T>::MS (cluster/ClusterFactory.cpp:621-656)
	void MS(BVec& bv, double sim, const std::string& output, int iter, int delta) {
		std::vector<Centre> part;
		Pt* last = bv.pop();
		while (last != nullptr) accumulate(&last, bv, part, sim);
		stamp("accumulate");
		std::cout << "Number of clusters before update: " << part.size() << std::endl;
		std::vector<size_t> num_clusters;
		for (int i = 0; i < iter; i++) {
			if (i >= 3 && part.size() == num_clusters[(size_t)i - 3]) break;
			if (batch_update) mean_shift_update_all(part, delta);
			else for (int j = 0; j < (int)part.size(); j++) mean_shift_update(part, j, delta);
			merge(part, delta);
			num_clusters.push_back(part.size());
		}
		if (batch_update) mean_shift_update_all(part, 0);
		else for (int j = 0; j < (int)part.size(); j++) mean_shift_update(part, j, 0);
		stamp("update");
		print_output(output, part);
		std::cout << "Number of clusters: " << part.size() << std::endl;
		stamp("done");
	}

	double cutoff = 0.9;
	bool batch_update = true;       // --serial-update: one centre at a time (the order the reference would take with one thread)
};

}  // namespace

// ------------------------------------------------------------------ running without --recover
// The reference then picks k (find_k, cluster/CRunner.cpp:479-502), the histogram type (:56-126) and trains a model on mutated
// templates (predict/Predictor.cpp:519-710). k and the type are chosen by the same rules; the training pairs come from this
// driver's OWN SplitMix64 mutator (substitutions + single-base indels at graded rates, labelled with the intended identity), not
// from the reference's generator, so the model is not the one `meshclust2` would train; selection + GLM are msc_train_class.
namespace {

struct SplitMix {
	uint64_t s;
	uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
	double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// Runner::find_k's per-record size (cluster/CRunner.cpp:479-502). It builds each record with ChromListMaker::makeChromList
// (nonltr/ChromListMaker.cpp:52-95), whose Chromosome(size) pre-fills the sequence with `size` blanks (nonltr/Chromosome.cpp:18-21) and
// then APPENDS the record's lines (appendToSequence, :88-97), so the effective size it reports is that of blanks + sequence: the
// leading blanks are one non-N run that continues into the first run of the record. Same run / merge (< 10 apart) / drop (< 20) rules as
// the encoder (nonltr/Chromosome.cpp:263-353), including the run that starts on the very last character being lost.
unsigned long long find_k_record_size(const std::string& seq, size_t n_blanks) {
	const size_t n = n_blanks + seq.size();
	auto is_n = [&](size_t i) { return i >= n_blanks && std::toupper((unsigned char)seq[i - n_blanks]) == 'N'; };
	std::vector<std::pair<long long, long long> > runs;
	long long start = -1;
	for (size_t i = 0; i < n; i++) {
		const bool nn = is_n(i);
		if (!nn && start == -1) start = (long long)i;
		else if (nn && start != -1) { runs.push_back({start, (long long)i - 1}); start = -1; }
		else if (i == n - 1 && !nn && start != -1) { runs.push_back({start, (long long)i}); start = -1; }
	}
	if (n > 20 && !runs.empty()) {
		std::vector<std::pair<long long, long long> > merged;
		long long s0 = runs[0].first, e0 = runs[0].second;
		for (size_t i = 1; i < runs.size(); i++) {
			if (runs[i].first - e0 < 10) e0 = runs[i].second;
			else { if (e0 - s0 + 1 >= 20) merged.push_back({s0, e0}); s0 = runs[i].first; e0 = runs[i].second; }
		}
		if (e0 - s0 + 1 >= 20) merged.push_back({s0, e0});
		runs.swap(merged);
	}
	unsigned long long eff = 0;
	for (const auto& r : runs) eff += (unsigned long long)(r.second - r.first + 1);
	return eff;
}

std::string acgt_only(const std::string& s) {
	std::string o;
	for (char c : s) { const char u = (char)std::toupper((unsigned char)c); if (u == 'A' || u == 'C' || u == 'G' || u == 'T') o.push_back(u); }
	return o;
}

std::string mutate(const std::string& tmpl, double rate, SplitMix& rng) {
	static const char B[4] = {'A', 'C', 'G', 'T'};
	const double sub = rate * 0.8, del = rate * 0.1, ins = rate * 0.1;
	std::string o;
	for (char c : tmpl) {
		const double u = rng.unit();
		if (u < sub) { char n; do { n = B[rng.next() >> 62]; } while (n == c); o.push_back(n); }
		else if (u < sub + del) { /* dropped */ }
		else if (u < sub + del + ins) { o.push_back(c); o.push_back(B[rng.next() >> 62]); }
		else o.push_back(c);
	}
	return o;
}

// cluster/CRunner.cpp:56-126: the narrowest type that holds the largest count (pseudocount included)
int choose_datatype(msc::Context& ctx, int k, const std::vector<std::string>& seqs) {
	auto largest = [&](int bits) {
		msc::PointSet probe(ctx, k, bits, 4096);
		uint64_t mx = 0;
		for (size_t off = 0; off < seqs.size(); off += 4096) {
			std::vector<std::string> part(seqs.begin() + (long)off, seqs.begin() + (long)std::min(seqs.size(), off + 4096));
			probe.get_points(0, part);
			for (size_t i = 0; i < part.size(); i++) mx = std::max<uint64_t>(mx, probe.info(i).max_count);
		}
		return mx;
	};
	const uint64_t m16 = largest(16);
	if (m16 <= 255) return 8;
	if (m16 < 65535) return 16;
	return largest(32) <= 65535 ? 16 : 32;
}

std::string train_model(msc::Context& ctx, int k, int dtype, double id, uint64_t feat_flags, int n_templates, int min_feat, int max_feat,
                        const std::vector<std::string>& seqs) {
	SplitMix rng{0xAAull};
	std::vector<std::string> pts;
	std::vector<uint32_t> first, second;
	std::vector<double> val;
	const double lo = std::max(0.35, id - 0.25);                 // negatives well below the cut-off (the reference's min_id, cluster/CRunner.h:41)
	const size_t nt = std::min<size_t>((size_t)n_templates, seqs.size());
	for (size_t t = 0; t < nt; t++) {
		const std::string tmpl = acgt_only(seqs[t * seqs.size() / nt]);
		if (tmpl.size() < 50) continue;
		pts.push_back(tmpl);
		const uint32_t ti = (uint32_t)pts.size() - 1;
		for (int j = 0; j < 8; j++) {
			const double target = j % 2 == 0 ? id + (1.0 - id) * rng.unit() : lo + (id - lo) * rng.unit();
			pts.push_back(mutate(tmpl, 1.0 - target, rng));
			first.push_back(ti); second.push_back((uint32_t)pts.size() - 1); val.push_back(target);
		}
	}
	if (first.size() < 16) throw msc::Error(MSC_ERR_INVALID_ARG, "too few usable sequences to train on");
	for (size_t i = first.size() - 1; i > 0; i--) {              // shuffle, then first half trains and second half tests
		const size_t j = (size_t)(rng.next() % (i + 1));
		std::swap(first[i], first[j]); std::swap(second[i], second[j]); std::swap(val[i], val[j]);
	}
	msc::PointSet set(ctx, k, dtype, pts.size());
	for (size_t off = 0; off < pts.size(); off += 4096) {
		std::vector<std::string> part(pts.begin() + (long)off, pts.begin() + (long)std::min(pts.size(), off + 4096));
		set.get_points(off, part);
	}
	const uint64_t n_train = first.size() / 2;
	std::vector<char> text(1 << 16);
	double atr = 0, ate = 0;
	ctx.check(msc_train_class(ctx.get(), set.get(), first.data(), second.data(), val.data(), n_train, first.size() - n_train, feat_flags, min_feat, max_feat, id,
	                          text.data(), text.size(), &atr, &ate));
	std::cout << "Training ACC: " << atr << std::endl << "Testing ACC: " << ate << std::endl;
	return std::string(text.data());
}

}  // namespace

int main(int argc, char** argv) {
	std::vector<std::string> files;
	std::string weights, output = "output.clstr";
	double similarity = 0.90;
	int k = -1, dtype = 0, delta = 5, iterations = 15, device = 0;
	bool single_file = false, sparse = false, serial_update = false;
	int n_templates = 300, min_feat = 4, max_feat = 4;      // cluster/CRunner.h:33-35
	uint64_t feat_flags = MSC_FEAT_FAST;                    // the CLI default (cluster/CRunner.h:51)
	std::string dump = "weights.txt";
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto need = [&](const char* what) { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", what); std::exit(1); } return std::string(argv[++i]); };
		if (a == "--id") {          // Runner::get_opts, cluster/CRunner.cpp:247-258: anything outside (0, 1) is refused
			similarity = std::atof(need("--id").c_str());
			if (!(similarity > 0 && similarity < 1)) { std::cerr << "Similarity must be between 0 and 1" << std::endl; return 1; }
		}
		else if (a == "--kmer" || a == "-k") k = std::atoi(need("--kmer").c_str());
		else if (a == "--datatype") { std::string v = need("--datatype"); dtype = v == "uint8_t" ? 8 : v == "uint16_t" ? 16 : v == "uint32_t" ? 32 : v == "uint64_t" ? 64 : std::atoi(v.c_str()); }
		else if (a == "--recover" || a == "-r") {
			// cluster/CRunner.cpp:291-297: the model's ID and k become the run's similarity and k on the spot (a later --id / --kmer overrides)
			weights = need("--recover");
			std::ifstream win(weights.c_str());
			std::string tok;
			while (win >> tok) {
				if (tok == "k:") win >> k;
				else if (tok == "ID:") win >> similarity;
				else if (tok == "n_combos:") break;
			}
		}
		else if (a == "--output" || a == "-o") output = need("--output");
		else if (a == "--delta" || a == "-d") delta = std::atoi(need("--delta").c_str());
		else if (a == "--iterations" || a == "-i" || a == "--iter") iterations = std::atoi(need("--iterations").c_str());
		else if (a == "--threads" || a == "-t") need("--threads");
		else if (a == "--device") device = std::atoi(need("--device").c_str());
		else if (a == "--feat" || a == "-f") { const std::string v = need("--feat"); feat_flags = v == "slow" ? MSC_FEAT_SLOW : MSC_FEAT_FAST; }
		else if (a == "--num-templates") n_templates = std::atoi(need("--num-templates").c_str());
		else if (a == "--min-feat" || a == "--min") min_feat = std::atoi(need("--min-feat").c_str());
		else if (a == "--max-feat" || a == "--max") max_feat = std::atoi(need("--max-feat").c_str());
		else if (a == "--dump") dump = need("--dump");
		else if (a == "--single-file") single_file = true;
		else if (a == "--serial-update") serial_update = true;
		else if (a == "--sparse") sparse = true;         // sparse histogram layout (required for k >= 13)
		else files.push_back(a);
	}
	if (files.empty()) {
		std::fprintf(stderr, "usage: %s <input.fa> [--recover weights.txt] [--id 0.9] [--kmer K] [--datatype 8|16|32|64] [--output out.clstr] [--delta 5] [--iterations 15]\n"
		                     "       without --recover a model is trained first (--feat fast|slow, --num-templates 300, --min-feat 4, --max-feat 4) and written to --dump (weights.txt)\n", argv[0]);
		return 1;
	}
	try {
		msc::Context ctx(device);
		std::vector<std::string> headers, seqs;
		std::vector<size_t> file_first;                // index of every file's first record (find_k averages per file, then over the files)
		for (const auto& f : files) { file_first.push_back(seqs.size()); read_fasta(f, headers, seqs, single_file); }
		file_first.push_back(seqs.size());
		const size_t n = seqs.size();
		if (n == 0) { std::fprintf(stderr, "no sequences\n"); return 1; }
		if (weights.empty()) {
			if (k < 0) {           // find_k, cluster/CRunner.cpp:479-502: ceil(log4(average record size)) - 1, integer averages
				unsigned long long length = 0;
				size_t n_files = 0;
				for (size_t f = 0; f + 1 < file_first.size(); f++) {
					if (file_first[f + 1] == file_first[f]) continue;
					unsigned long long l = 0;
					// --single-file: one record per file, pre-sized for every joined record + 50 (nonltr/ChromListMaker.cpp:58-65)
					for (size_t i = file_first[f]; i < file_first[f + 1]; i++) l += find_k_record_size(seqs[i], seqs[i].size() + (single_file ? 50 : 0));
					length += l / (file_first[f + 1] - file_first[f]);
					n_files++;
				}
				length /= std::max<size_t>(1, n_files);
				if (length == 0) { std::fprintf(stderr, "cannot choose k: the input holds no bases\n"); return 1; }
				k = (int)std::ceil(std::log((double)length) / std::log(4.0)) - 1;
				std::cout << "avg length: " << length << std::endl << "Recommended K: " << k << std::endl;
			}
			if (dtype == 0) { dtype = choose_datatype(ctx, k, seqs); std::cout << "Using " << dtype << " bit histograms" << std::endl; }
			const double id = similarity > 1 ? similarity / 100.0 : similarity;
			const std::string text = train_model(ctx, k, dtype, id, feat_flags, n_templates, min_feat, max_feat, seqs);
			std::ofstream(dump.c_str()) << text;       // the reference always leaves weights.txt behind (cluster/Trainer.cpp:188-190)
			weights = dump;
			Driver::stamp("GLM");
		}
		msc::Trainer trn(ctx, weights, similarity);
		if (k < 0) k = msc_model_k(trn.feature().get());
		if (dtype == 0) {      // "Datatype:" line of the weights file
			std::ifstream in(weights.c_str());
			std::string tok;
			while (in >> tok) if (tok == "Datatype:") { in >> tok; dtype = tok == "uint8_t" ? 8 : tok == "uint16_t" ? 16 : tok == "uint32_t" ? 32 : 64; break; }
		}
		uint64_t total_bases = 0, longest = 0;
		for (const auto& sq : seqs) { total_bases += sq.size(); longest = std::max<uint64_t>(longest, sq.size()); }
		msc::PointSet points(ctx, k, dtype, n, sparse ? total_bases + 1024 : 0);
		const size_t chunk = 8192;
		for (size_t off = 0; off < n; off += chunk) {
			std::vector<std::string> part(seqs.begin() + (long)off, seqs.begin() + (long)std::min(n, off + chunk));
			points.get_points(off, part);
		}
		std::vector<Pt> store(n);
		std::vector<Pt*> pts(n);
		for (size_t i = 0; i < n; i++) {
			store[i].header = headers[i];
			store[i].slot = (uint32_t)i;
			store[i].length = points.get_length(i);
			pts[i] = &store[i];
		}
		seqs.clear();
		// get_points: sort by header, then by length, both unstable std::sort (cluster/CRunner.cpp:538-539)
		std::sort(pts.begin(), pts.end(), [](Pt* a, Pt* b) { return a->header < b->header; });
		std::sort(pts.begin(), pts.end(), [](Pt* a, Pt* b) { return a->length < b->length; });
		std::vector<uint64_t> lengths;
		for (Pt* p : pts) lengths.push_back(p->length);
		BVec bv(lengths, 1000);
		uint64_t idx = 0;
		for (Pt* p : pts) { p->id = idx++; bv.insert(p); }
		bv.insert_finalize();
		Driver::stamp("read_in_points");
		Driver drv(ctx, points, trn, k, dtype, sparse ? total_bases + 64 * longest + (1 << 20) : 0);      // worst case every sequence stays its own centre; the slack absorbs set() appends between compactions
		drv.cutoff = similarity;
		drv.batch_update = !serial_update;
		drv.MS(bv, similarity, output, iterations, delta);
	} catch (const msc::Error& e) {
		std::fprintf(stderr, "msc error %d: %s\n", e.code, e.what());
		return 3;
	}
	return 0;
}
