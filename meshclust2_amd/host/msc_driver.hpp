// msc_driver.hpp -- the mean-shift clustering logic around the hot path, independent of where the
// histograms live.
//
// The reference's clustering loop (cluster/ClusterFactory.cpp MS / accumulate / mean_shift_update / merge / print_output,
// cluster/bvec.cpp, cluster/CRunner.cpp do_run) is host control flow AROUND the hot path. It is written here once, against
// an abstract ClusterBackend (seven operators: the Trainer<T> calls of cluster/ClusterFactory.cpp:312,327,387,566 and the
// Center constructor / center->set of :328,331,603), so that the same logic drives
//   - one GPU           (msc_cluster.cpp: the backend calls the C ABI directly),
//   - one GPU per rank  (meshclust2_amd/cluster.py through the C entry point of msc_driver_capi.cpp: every rank runs this
//                        logic on replicated flags / lists, the backend shards the scoring and does the exchanges),
//   - no GPU at all     (the world-size-2 gloo test, whose backend is the CPU oracle).
// Points and centres are small integer handles; histograms never appear here.
//
// Behaviour reproduced on purpose (SURVEY.md findings): the exclusive use of an inclusive end index in the scoring
// window (Q6), DivergencePoint::set keeping the stale magnitude of a moved centre (Q7: the backend's business), the
// OMP_NUM_THREADS=1 order of every reduction and of remove_available (Q10), the unstable std::sort orders (same
// libstdc++, same comparator, same input order => same permutation), and what bvec's range lookup returns when a bin
// is empty or the length lies outside every bin. LengthBins states those lookups with std::upper_bound /
// std::equal_range; tests/test_driver_cpu.py holds it to the reference's own bvec on random length multisets.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace msc {

struct SeqRecord {               // host shadow of one Point<T>: everything the clustering logic reads
	std::string header;          // full header line including '>'
	uint64_t length = 0;         // effective length
	uint64_t id = 0;             // position after the two sorts (cluster/CRunner.cpp:588-593)
	uint32_t point = 0;          // handle of the histogram (position in the input)
};

// The operators the loop calls. Handles: points 0..n-1 as given to mean_shift(); centres as centre_new() returns them.
struct ClusterBackend {
	virtual ~ClusterBackend() {}
	// Trainer::get_close(last, [istart, iend), is_min) (cluster/Trainer.cpp:23-71): query = point q, window = points in window
	// order. flags[j] = the candidate is marked; pos = position in `window` of the first arg-max of combo 0, -1 if nothing
	// passed the length filter; is_min = no candidate was close.
	virtual void get_close(uint32_t q, const std::vector<uint32_t>& window, std::vector<uint8_t>& flags, int64_t& pos, bool& is_min) = 0;
	// get_mean / Trainer::closest (cluster/ClusterFactory.cpp:338-380, cluster/Trainer.cpp:144-157): position of the member
	// nearest the FP64 mean of the members (first minimum)
	virtual int64_t closest(const std::vector<uint32_t>& members) = 0;
	virtual uint32_t centre_new(uint32_t point) = 0;                 // Center(c): center(c->clone())
	virtual void centre_set(uint32_t centre, uint32_t point) = 0;    // center->set(*next): bins, length, id -- not mag
	// Trainer::filter(centre, points) (cluster/Trainer.cpp:123-141): keep[i] = 1 iff point i survives
	virtual void filter(uint32_t centre, const std::vector<uint32_t>& points, std::vector<uint8_t>& keep) = 0;
	// Trainer::merge(centres, current, begin, last) (cluster/Trainer.cpp:74-109)
	virtual long merge(const std::vector<uint32_t>& centres, long current, long begin, long last) = 0;
	// Batched forms of one update round; return false where the backend has none (the loop then goes centre by centre).
	//   update_centres: for centre c and its list points[offsets[c] .. offsets[c+1]): filter, mean of the survivors, nearest
	//                   survivor -> nearest[c] = position inside the list or -1
	//   centre_set_batch: centre_set(centres[i], points[i]) for all i
	//   merge_all: best[i] = merge(centres, i, i + 1, min(n - 1, i + delta))
	virtual bool update_centres(const std::vector<uint32_t>&, const std::vector<uint32_t>&, const std::vector<uint64_t>&, std::vector<int64_t>&) { return false; }
	virtual bool centre_set_batch(const std::vector<uint32_t>&, const std::vector<uint32_t>&) { return false; }
	virtual bool merge_all(const std::vector<uint32_t>&, int, std::vector<int64_t>&) { return false; }
	//   merge_some: best[w] = merge(centres, which[w], which[w] + 1, min(n - 1, which[w] + delta))
	virtual bool merge_some(const std::vector<uint32_t>&, int, const std::vector<uint64_t>&, std::vector<int64_t>&) { return false; }
	// The window of get_close kept on the backend's side. set_order: order[pos] = point at position pos of the sealed length-binned
	// store (bins concatenated; every position alive); true = the backend takes ranges from now on. get_close_range: get_close
	// over the ALIVE positions of [first, end) in position order; `close` = the positions it marks, ascending -- they leave the store
	// next (remove_available, cluster/ClusterFactory.cpp:598), so the backend drops them at once; best = position of the arg-max
	// or -1. kill: a position that leaves the store otherwise (the next seed: bvec::erase / bvec::pop, :589,593).
	virtual bool set_order(const std::vector<uint32_t>&) { return false; }
	virtual void get_close_range(uint32_t, uint64_t, uint64_t, std::vector<uint32_t>&, int64_t&, bool&) {}
	virtual void kill(uint64_t) {}
};

// ------------------------------------------------------------------ the length-binned store (cluster/bvec.{h,cpp})
// Bins hold (record, marked) in ascending length; bin b starts at the length of every per_bin-th record of the sorted input.
class LengthBins {
public:
	struct Item { SeqRecord* rec; bool marked; uint32_t fixed = 0; };      // fixed = position in the sealed store (never changes)
	struct Pos { size_t bin = 0, at = 0; bool none = false; };          // bvec_idx_t
	typedef std::deque<Item> Bin;         // (bvec::pop takes the front record of a bin: equal-length inputs pop a million times from one bin)

	LengthBins(std::vector<uint64_t> lengths, uint64_t per_bin) {         // bvec.cpp:10-24
		std::sort(lengths.begin(), lengths.end());
		for (uint64_t i = 0; i < lengths.size(); i += per_bin) starts_.push_back(lengths[i]);
		bins.resize(starts_.size());
	}

	// bvec::index_of (bvec.cpp:123-147) scans every pair of neighbouring bin starts for start[i] <= len < start[i+1] and adds
	// the last bin when len >= its start. The starts are sorted, so at most one pair matches: with u = number of starts <= len,
	// the answer is bin u - 1 for both ends -- except u == 0 (shorter than everything), where the scan leaves its initial
	// values behind: front = last bin, back = first bin.
	std::pair<size_t, size_t> bins_of(uint64_t len) const {
		const size_t u = (size_t)(std::upper_bound(starts_.begin(), starts_.end(), len) - starts_.begin());
		if (u == 0) return std::make_pair(starts_.size() - 1, (size_t)0);
		return std::make_pair(u - 1, u - 1);
	}

	void add(SeqRecord* r) {                                              // bvec.cpp:150-184
		const std::pair<size_t, size_t> fb = bins_of(r->length);
		// the reference picks the middle one of the least-filled bins of [front, back]; with sorted starts that range is one bin,
		// or empty (front > back) for a length below every start -- which cannot happen for lengths the starts were drawn from
		if (fb.first > fb.second) throw std::runtime_error("length-binned store: no bin for this length");
		bins[fb.first].push_back(Item{r, false});
	}
	void seal() {                                                         // insert_finalize, bvec.cpp:216-233: unstable sort per bin
		for (Bin& b : bins) std::sort(b.begin(), b.end(), [](const Item x, const Item y) { return x.rec->length < y.rec->length; });
	}
	SeqRecord* take_first(uint32_t* fixed = nullptr) {                    // pop, bvec.cpp:27-37
		for (Bin& b : bins) if (!b.empty()) { SeqRecord* r = b.front().rec; if (fixed) *fixed = b.front().fixed; b.erase(b.begin()); return r; }
		return nullptr;
	}
	// positions of the sealed store: bins concatenated in order. order[pos] = point handle; bin_of_[pos] = its bin, for good
	void number(std::vector<uint32_t>& order) {
		order.clear();
		bin_of_.clear();
		for (size_t b = 0; b < bins.size(); b++)
			for (Item& it : bins[b]) { it.fixed = (uint32_t)order.size(); order.push_back(it.rec->point); bin_of_.push_back((uint32_t)b); }
	}
	// where the record of a position sits now (erasing keeps a bin in position order)
	std::pair<size_t, size_t> locate(uint32_t fixed) const {
		const size_t b = bin_of_.at(fixed);
		const Bin& bin = bins[b];
		const size_t at = (size_t)(std::lower_bound(bin.begin(), bin.end(), fixed, [](const Item& it, uint32_t f) { return it.fixed < f; }) - bin.begin());
		if (at >= bin.size() || bin[at].fixed != fixed) throw std::runtime_error("length-binned store: position is no longer there");
		return std::make_pair(b, at);
	}
	// the records at these positions (ascending) leave in bin order, as take_marked would take them
	void take_positions(const std::vector<uint32_t>& fixed, std::vector<SeqRecord*>& out) {
		size_t i = 0;
		while (i < fixed.size()) {
			const size_t b = bin_of_.at(fixed[i]);
			Bin& bin = bins[b];
			size_t j = i;
			for (; j < fixed.size() && bin_of_[fixed[j]] == b; j++) {
				const std::pair<size_t, size_t> w = locate(fixed[j]);
				bin[w.second].marked = true;
				out.push_back(bin[w.second].rec);
			}
			bin.erase(std::remove_if(bin.begin(), bin.end(), [](const Item d) { return d.marked; }), bin.end());
			i = j;
		}
	}

	// bvec::inner_index_of (bvec.cpp:52-120) inside a non-empty bin: its bisection ends on an element equal to `len` if there
	// is one and is then widened over the run of equal lengths (front = first of the run, back = last); otherwise it ends on the
	// first longer element, or on the bin's last element when none is longer, and front = back = that position.
	// For an EMPTY bin it moves to the first (front) / last (back) non-empty bin of the whole store, position 0; when every bin is
	// empty nothing is touched.
	void locate_front(uint64_t len, Pos& p) const {
		if (bins.at(p.bin).empty()) {
			for (size_t i = 0; i < bins.size(); i++) if (!bins[i].empty()) { p.bin = i; p.at = 0; break; }
			return;
		}
		p.at = run_of(bins[p.bin], len).first;
	}
	void locate_back(uint64_t len, Pos& p) const {
		if (bins.at(p.bin).empty()) {
			for (size_t i = bins.size(); i-- > 0;) if (!bins[i].empty()) { p.bin = i; p.at = 0; break; }
			return;
		}
		p.at = run_of(bins[p.bin], len).second;
	}

	// get_range (bvec.cpp:261-330): the (inclusive) positions of begin_len and end_len
	std::pair<Pos, Pos> range(uint64_t begin_len, uint64_t end_len) const {
		Pos front, back;
		back.bin = bins.size() - 1;
		back.at = bins[back.bin].size() - 1;                 // wraps to (size_t)-1 for an empty last bin: the `none` test below
		front.bin = bins_of(begin_len).first;
		back.bin = bins_of(end_len).second;
		locate_front(begin_len, front);
		locate_back(end_len, back);
		if (back.bin == (size_t)-1 || back.at == (size_t)-1) back.none = true;
		return std::make_pair(front, back);
	}
	void erase(size_t bin, size_t at) { bins.at(bin).erase(bins.at(bin).begin() + (long)at); }
	// remove_available (bvec.cpp:342-384), one thread: marked records of bins [begin.bin, end.bin] leave in bin order
	void take_marked(const Pos& begin, const Pos& end, std::vector<SeqRecord*>& out) {
		if (begin.none || end.none) return;
		for (size_t i = begin.bin; i <= end.bin && i < bins.size(); i++) {
			Bin& b = bins[i];
			for (const Item& it : b) if (it.marked) out.push_back(it.rec);
			b.erase(std::remove_if(b.begin(), b.end(), [](const Item d) { return d.marked; }), b.end());
		}
	}
	const std::vector<uint64_t>& starts() const { return starts_; }

	std::vector<Bin> bins;

private:
	std::vector<uint32_t> bin_of_;
	static std::pair<size_t, size_t> run_of(const Bin& b, uint64_t len) {
		const auto lt = [](const Item& it, uint64_t v) { return it.rec->length < v; };
		const auto gt = [](uint64_t v, const Item& it) { return v < it.rec->length; };
		const size_t lo = (size_t)(std::lower_bound(b.begin(), b.end(), len, lt) - b.begin());
		const size_t hi = (size_t)(std::upper_bound(b.begin(), b.end(), len, gt) - b.begin());
		if (lo < hi) return std::make_pair(lo, hi - 1);                   // the run of records of exactly this length
		const size_t t = std::min(hi, b.size() - 1);                     // first longer record, else the last one
		return std::make_pair(t, t);
	}
	std::vector<uint64_t> starts_;
};

// bvec_iterator (cluster/bvec_iterator.{h,cpp}): a position that steps over empty bins, and the reference's operator- -- the trip
// count OpenMP derives for `for (i = istart; i < iend; ++i)`
struct BinCursor {
	size_t bin, at;
	const std::vector<LengthBins::Bin>* bins;
	void step() {
		if (bin == bins->size()) throw std::runtime_error("tried incrementing null iterator");
		if (at + 1 < (*bins)[bin].size()) { at++; return; }
		bin++;
		at = 0;
		while (bin < bins->size() && (*bins)[bin].empty()) bin++;
	}
	bool before(const BinCursor& o) const { return bin < o.bin || (bin == o.bin && at < o.at); }
	int64_t distance_from(const BinCursor& from) const {                  // *this - from, bvec_iterator.h:57-76
		if (before(from)) return -from.distance_from(*this);
		if (bin == from.bin) return (int64_t)(at - from.at);
		int64_t d = (int64_t)at + (int64_t)((*bins)[from.bin].size() - from.at);
		for (size_t i = from.bin + 1; i < bin; i++) d += (int64_t)(*bins)[i].size();
		return d;
	}
};

// ------------------------------------------------------------------ centres (cluster/Center.h)
struct Cluster {
	uint32_t centre = 0;         // backend handle of the centre histogram (a clone of a point, possibly moved by set())
	std::string header;
	uint64_t id = 0, length = 0;
	uint32_t centre_point = 0;   // the point the centre histogram was cloned from or last set() to: a set() to the same point again changes nothing
	                             // (bins, length, id of that point; the magnitude stays whatever it was) and is skipped
	std::vector<SeqRecord*> members;
	bool merged_away = false;
	uint32_t serial = 0;         // the cluster's number in the order of making: names it for the whole run
	uint32_t version = 0;        // counts the changes of `members` since the cluster was made (a merge appends to them)
	// what the last update round asked about this centre -- the point it held and the clusters of its neighbourhood, in order, each as
	// (serial, version): equal pairs are equal member lists -- and what came back.
	// The same question has the same answer (the centre's histogram, the model and the list decide it): a round does not ask it again.
	// (r05: the lists themselves were kept and compared point by point, a pointer chase per member: 1.0 s of BASELINE cfg3's nine rounds.)
	std::vector<uint64_t> asked;          // [0] = the centre's point, then serial << 32 | version of clusters lo .. hi
	// ... and the same for the merge loop's question about this centre (cluster i against the centres of i + 1 .. i + delta): the centres'
	// histograms decide it -- (serial, number of set() calls) each; the answer is remembered as a distance down the list
	uint32_t centre_version = 0;
	std::vector<uint64_t> merge_asked;
	int32_t merge_partner = 0;          // ret - i where the call returned ret > i, else 0
	bool merge_answered = false;
	SeqRecord* answer = nullptr;
	bool answered = false;
};

class MeanShift {
public:
	MeanShift(ClusterBackend& b, std::ostream& log) : be_(b), log_(log) {}
	bool batch_update = true;    // false: one centre at a time (the order the reference takes with one thread)
	bool use_ranges = true;      // false: get_close always receives the window as a slot list rebuilt per step (the r02 form)
	// MSC_CLUSTER_PROFILE=1: where the step-serial accumulate stage spends its wall clock (seconds), printed behind its timestamp
	struct StepProfile { double window = 0, get_close = 0, mark = 0, closest = 0; uint64_t steps = 0, candidates = 0, closed = 0; } prof;
	bool profile = std::getenv("MSC_CLUSTER_PROFILE") != nullptr;
	// ... and the update stage: building the neighbourhood lists / update_centres / centre_set_batch / merge_all / the host's merge bookkeeping
	struct UpdateProfile { double lists = 0, update = 0, set = 0, merge = 0, book = 0; int rounds = 0; } uprof;
	static double seconds_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }

	// Clock::stamp (clutil/Clock.cpp:12-19): same stage names as the reference's driver
	void stamp(const char* desc) {
		log_ << "timestamp " << desc << " " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count() << std::endl;
	}

	// do_run's tail + ClusterFactory<T>::MS (cluster/CRunner.cpp:538-539,574-597; cluster/ClusterFactory.cpp:621-656).
	// records[i] = point i (header, effective length). output == nullptr: nothing is written (ranks other than 0).
	std::vector<Cluster> run(std::vector<SeqRecord>& records, double sim, int iterations, int delta, const char* output) {
		std::vector<SeqRecord*> pts(records.size());
		for (size_t i = 0; i < records.size(); i++) { records[i].point = (uint32_t)i; pts[i] = &records[i]; }
		// get_points: sort by header, then by length, both unstable std::sort (cluster/CRunner.cpp:538-539). Where equal lengths end up is
		// whatever std::sort does with them, so it is std::sort here too, with comparators that answer every question as those of the
		// reference do -- the sequence of moves depends on the answers alone -- but from a key that travels with the pointer (the first
		// eight bytes of the header, most significant first, zeros behind a shorter one: it orders as memcmp does wherever it differs;
		// then the length) instead of from two cache lines per question: 0.35 -> 0.15 s of BASELINE cfg3's 10^6 records.
		{
			struct Keyed { uint64_t key; SeqRecord* p; };
			std::vector<Keyed> v(records.size());
			for (size_t i = 0; i < v.size(); i++) {
				const std::string& h = records[i].header;
				uint64_t key = 0;
				for (size_t c = 0; c < 8; c++) key = key << 8 | (c < h.size() ? (unsigned char)h[c] : 0u);
				v[i] = Keyed{key, &records[i]};
			}
			std::sort(v.begin(), v.end(), [](const Keyed& a, const Keyed& b) { return a.key != b.key ? a.key < b.key : a.p->header < b.p->header; });
			for (Keyed& x : v) x.key = x.p->length;
			std::sort(v.begin(), v.end(), [](const Keyed& a, const Keyed& b) { return a.key < b.key; });
			for (size_t i = 0; i < v.size(); i++) pts[i] = v[i].p;
		}
		std::vector<uint64_t> lengths;
		for (SeqRecord* p : pts) lengths.push_back(p->length);
		LengthBins store(lengths, 1000);
		uint64_t idx = 0;
		for (SeqRecord* p : pts) { p->id = idx++; store.add(p); }
		store.seal();
		{
			std::vector<uint32_t> order;
			store.number(order);
			ranged_ = use_ranges && be_.set_order(order);
		}
		stamp("read_in_points");

		std::vector<Cluster> part;
		SeqRecord* last = take_first(store);
		while (last != nullptr) accumulate(&last, store, part, sim);
		stamp("accumulate");
		if (profile)
			log_ << "accumulate profile: steps " << prof.steps << " candidates " << prof.candidates << " closed " << prof.closed << " | window " << prof.window
			     << " s, get_close " << prof.get_close << " s, mark+take " << prof.mark << " s, closest " << prof.closest << " s" << std::endl;
		log_ << "Number of clusters before update: " << part.size() << std::endl;
		std::vector<size_t> history;
		bool quiet = false;
		for (int i = 0; i < iterations; i++) {
			if (i >= 3 && part.size() == history[(size_t)i - 3]) break;      // unchanged for three rounds (:636)
			// (a round that asked nothing new, moved no centre and merged no cluster leaves the state it found: the next one would repeat it
			// question for question -- the rounds the stopping rule waits through)
			if (!quiet) {
				const bool moved = update_round(part, delta);
				const bool merged = merge_round(part, delta);
				quiet = !moved && !merged;
			}
			history.push_back(part.size());
		}
		update_round(part, 0);
		stamp("update");
		if (profile)
			log_ << "update profile: rounds " << uprof.rounds << " | lists " << uprof.lists << " s, update_centres " << uprof.update << " s, centre_set " << uprof.set
			     << " s, merge_all " << uprof.merge << " s, merge bookkeeping " << uprof.book << " s" << std::endl;
		if (output) write_clstr(output, part);
		log_ << "Number of clusters: " << part.size() << std::endl;
		stamp("done");
		return part;
	}

	// print_output (cluster/ClusterFactory.cpp:404-435)
	static void write_clstr(const std::string& path, const std::vector<Cluster>& part) {
		std::ofstream ofs(path.c_str());
		int counter = 0;
		for (const Cluster& cl : part) {
			if (cl.members.empty()) continue;
			ofs << ">Cluster " << counter << '\n';          // (the same bytes as std::endl writes, without a flush -- a system call -- per line: 1.2 s of
			int pt = 0;                                          // BASELINE cfg3's 1.84 M lines)
			for (const SeqRecord* p : cl.members) {
				ofs << pt << "\t" << p->length << "nt, " << p->header << "... ";
				if (p->id == cl.id) ofs << "*";
				ofs << '\n';
				pt++;
			}
			counter++;
		}
	}

private:
	static std::vector<uint32_t> handles(const std::vector<SeqRecord*>& v) {
		std::vector<uint32_t> h(v.size());
		for (size_t i = 0; i < v.size(); i++) h[i] = v[i]->point;
		return h;
	}
	void move_centre(Cluster& cl, const SeqRecord* next) { cl.header = next->header; cl.id = next->id; cl.length = next->length; cl.centre_point = next->point; cl.centre_version++; }

	// accumulate (cluster/ClusterFactory.cpp:553-610): grow one cluster from *seed until a pass finds nothing close
	void accumulate(SeqRecord** seed, LengthBins& store, std::vector<Cluster>& part, double sim) {
		SeqRecord* last = *seed;
		std::vector<SeqRecord*> current = {last};
		bool is_min = false;
		std::vector<uint32_t> window;
		std::vector<std::pair<size_t, size_t> > where;
		std::vector<uint8_t> flags;
		while (!is_min) {
			const uint64_t len = last->length;
			const std::pair<LengthBins::Pos, LengthBins::Pos> bounds = store.range((uint64_t)(len * sim), (uint64_t)(len / sim));
			if (ranged_) {
				// the same window as below -- the (iend - istart) records from istart on, iend exclusive -- named by the
				// sealed-store positions of its two ends: the backend holds the order and knows which positions are still there
				const auto tq0 = std::chrono::steady_clock::now();
				const BinCursor it{bounds.first.bin, bounds.first.at, &store.bins}, end{bounds.second.bin, bounds.second.at, &store.bins};
				const int64_t trips = end.distance_from(it);
				int64_t best = -1;
				close_.clear();
				is_min = true;
				const auto tq1 = std::chrono::steady_clock::now();
				if (trips > 0)
					be_.get_close_range(last->point, store.bins.at(it.bin).at(it.at).fixed, store.bins.at(end.bin).at(end.at).fixed, close_, best, is_min);
				const auto tq2 = std::chrono::steady_clock::now();
				if (profile) {
					prof.steps++;
					prof.candidates += (uint64_t)std::max<int64_t>(trips, 0);
					prof.window += std::chrono::duration<double>(tq1 - tq0).count();
					prof.get_close += std::chrono::duration<double>(tq2 - tq1).count();
				}
				if (is_min) {
					if (best < 0) {
						*seed = take_first(store);
					} else {
						const std::pair<size_t, size_t> w = store.locate((uint32_t)best);
						*seed = store.bins[w.first][w.second].rec;
						store.erase(w.first, w.second);
						be_.kill((uint64_t)best);
					}
					if (profile) prof.mark += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq2).count();
				} else {
					const size_t before = current.size();
					store.take_positions(close_, current);
					if (current.empty()) throw std::runtime_error("N cannot be 0, bad");
					const auto tq3 = std::chrono::steady_clock::now();
					last = current[(size_t)be_.closest(handles(current))];       // get_mean (:338-380)
					if (profile) {
						prof.closest += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq3).count();
						prof.mark += std::chrono::duration<double>(tq3 - tq2).count();
						prof.closed += current.size() - before;
					}
				}
				continue;
			}
			// the window [iter(first), iter(second)): `i < iend` with an INCLUSIVE end position, walked the way OpenMP
			// walks it: (iend - istart) iterations of istart + n
			const auto tp0 = std::chrono::steady_clock::now();
			window.clear();
			where.clear();
			BinCursor it{bounds.first.bin, bounds.first.at, &store.bins};
			const BinCursor end{bounds.second.bin, bounds.second.at, &store.bins};
			const int64_t trips = end.distance_from(it);
			for (int64_t n = 0; n < trips; n++) {
				window.push_back(store.bins.at(it.bin).at(it.at).rec->point);
				where.emplace_back(it.bin, it.at);
				if (n + 1 < trips) it.step();
			}
			int64_t pos = -1;
			const auto tp1 = std::chrono::steady_clock::now();
			be_.get_close(last->point, window, flags, pos, is_min);
			const auto tp2 = std::chrono::steady_clock::now();
			for (size_t j = 0; j < flags.size(); j++) if (flags[j]) store.bins[where[j].first][where[j].second].marked = true;
			if (is_min) {
				if (pos < 0) {
					*seed = take_first(store);
				} else {
					*seed = store.bins[where[(size_t)pos].first][where[(size_t)pos].second].rec;
					store.erase(where[(size_t)pos].first, where[(size_t)pos].second);
				}
				std::vector<SeqRecord*> none;
				store.take_marked(bounds.first, bounds.second, none);
			} else {
				const size_t before = current.size();
				store.take_marked(bounds.first, bounds.second, current);
				if (current.empty()) throw std::runtime_error("N cannot be 0, bad");
				const auto tp3 = std::chrono::steady_clock::now();
				last = current[(size_t)be_.closest(handles(current))];       // get_mean (:338-380)
				if (profile) {
					prof.closest += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp3).count();
					prof.mark += std::chrono::duration<double>(tp3 - tp2).count();
					prof.closed += current.size() - before;
				}
			}
			if (profile) {
				prof.steps++;
				prof.candidates += window.size();
				prof.window += std::chrono::duration<double>(tp1 - tp0).count();
				prof.get_close += std::chrono::duration<double>(tp2 - tp1).count();
				if (is_min) prof.mark += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp2).count();
			}
		}
		Cluster cl;
		cl.centre = be_.centre_new(last->point);
		cl.serial = serial_++;
		move_centre(cl, last);
		cl.members = std::move(current);
		part.push_back(std::move(cl));
	}

	// the neighbourhood of centre j in a round: the members of clusters j - delta .. j + delta (:293-311)
	static void neighbourhood(const std::vector<Cluster>& part, size_t j, int delta, std::vector<SeqRecord*>& out) {
		const int lo = std::max(0, (int)j - delta), hi = std::min((int)j + delta, (int)part.size() - 1);
		for (int i = lo; i <= hi; i++) for (SeqRecord* p : part[(size_t)i].members) out.push_back(p);
	}

	// one `omp parallel for` over mean_shift_update (cluster/ClusterFactory.cpp:288-335,639,648). The centres of a round are
	// independent (each reads its own histogram and the member lists of its neighbours, none of which change during the round),
	// so a backend may take all of them in one call.
	// returns whether the round asked the backend anything or moved a centre
	bool update_round(std::vector<Cluster>& part, int delta) {
		const size_t n = part.size();
		if (n == 0) return false;
		if (batch_update) {
			auto t0 = std::chrono::steady_clock::now();
			// Only the centres whose question changed since the last round go to the backend (r05): once the clusters have settled -- BASELINE
			// cfg3: 10^6 centres, nine rounds -- a round repeats the round before it almost centre for centre.
			std::vector<uint32_t> centres, flat;
			std::vector<size_t> which;                       // the centres asked about this round
			std::vector<uint64_t> offsets(1, 0);
			std::vector<SeqRecord*> good, mine;
			for (size_t j = 0; j < n; j++) {
				const int lo = std::max(0, (int)j - delta), hi = std::min((int)j + delta, (int)n - 1);
				auto name = [&](int i) { return (uint64_t)part[(size_t)i].serial << 32 | part[(size_t)i].version; };
				if (part[j].answered) {          // the same question as last round? (compared in place: nothing is built for a centre that is not asked about)
					const std::vector<uint64_t>& q = part[j].asked;
					bool same = q.size() == (size_t)(hi - lo + 2) && q[0] == part[j].centre_point;
					for (int i = lo; same && i <= hi; i++) same = q[(size_t)(1 + i - lo)] == name(i);
					if (same) continue;
				}
				mine.clear();
				neighbourhood(part, j, delta, mine);
				std::vector<uint64_t>& q = part[j].asked;
				q.clear();
				q.push_back(part[j].centre_point);
				for (int i = lo; i <= hi; i++) q.push_back(name(i));
				part[j].answered = false;
				which.push_back(j);
				centres.push_back(part[j].centre);
				good.insert(good.end(), mine.begin(), mine.end());
				offsets.push_back(good.size());
			}
			flat = handles(good);
			std::vector<int64_t> nearest(which.size(), -1);
			uprof.rounds++;
			uprof.lists += seconds_since(t0);
			t0 = std::chrono::steady_clock::now();
			const bool took = which.empty() || be_.update_centres(centres, flat, offsets, nearest);
			uprof.update += seconds_since(t0);
			if (took) {
				t0 = std::chrono::steady_clock::now();
				for (size_t i = 0; i < which.size(); i++) {
					Cluster& cl = part[which[i]];
					cl.answer = nearest[i] >= 0 ? good[(size_t)(offsets[i] + (uint64_t)nearest[i])] : (delta == 0 ? cl.members[0] : nullptr);
					cl.answered = true;
				}
				std::vector<uint32_t> dst, src;
				for (size_t j = 0; j < n; j++) {
					SeqRecord* next = part[j].answer;
					if (!next || next->point == part[j].centre_point) continue;          // (BASELINE cfg3: five centres in six sit on their only member, round after round)
					dst.push_back(part[j].centre);
					src.push_back(next->point);
					move_centre(part[j], next);
				}
				if (!dst.empty() && !be_.centre_set_batch(dst, src))
					for (size_t i = 0; i < dst.size(); i++) be_.centre_set(dst[i], src[i]);
				uprof.set += seconds_since(t0);
				return !which.empty() || !dst.empty();
			}
			for (size_t j : which) part[j].answered = false;          // (the backend has no batched form: the loop below asks centre by centre)
		}
		for (size_t j = 0; j < n; j++) {
			Cluster& cl = part[j];
			std::vector<SeqRecord*> good;
			neighbourhood(part, j, delta, good);
			std::vector<uint8_t> keep;
			be_.filter(cl.centre, handles(good), keep);
			size_t w = 0;
			for (size_t i = 0; i < good.size(); i++) if (keep[i]) good[w++] = good[i];
			good.resize(w);
			SeqRecord* next = nullptr;
			if (!good.empty()) next = good[(size_t)be_.closest(handles(good))];
			else if (delta == 0) next = cl.members[0];
			if (next && next->point != cl.centre_point) { be_.centre_set(cl.centre, next->point); move_centre(cl, next); }
		}
		return true;          // (centre by centre: nothing is remembered)
	}

	// merge (cluster/ClusterFactory.cpp:383-401): centre i joins the best partner among the next delta centres
	// returns whether any cluster was merged away
	bool merge_round(std::vector<Cluster>& part, int delta) {
		const int n = (int)part.size();
		std::vector<uint32_t> centres((size_t)n);
		for (int c = 0; c < n; c++) centres[(size_t)c] = part[(size_t)c].centre;
		// no merge call changes a histogram, so all of them may be answered at once -- and only those whose question changed since the
		// last round need asking (r05: BASELINE cfg3 spends nine rounds over 10^6 centres of which few move after the second)
		std::vector<int64_t> best((size_t)n, 0);
		auto t0 = std::chrono::steady_clock::now();
		bool have = false;
		if (batch_update) {
			std::vector<uint64_t> which;
			auto name = [&](int i) { return (uint64_t)part[(size_t)i].serial << 32 | part[(size_t)i].centre_version; };
			for (int i = 0; i < n; i++) {
				Cluster& cl = part[(size_t)i];
				const int hi = std::min(n - 1, i + delta);
				std::vector<uint64_t>& q = cl.merge_asked;
				bool same = cl.merge_answered && q.size() == (size_t)(hi - i + 1);
				for (int j = i; same && j <= hi; j++) same = q[(size_t)(j - i)] == name(j);
				if (same) { best[(size_t)i] = cl.merge_partner ? i + cl.merge_partner : 0; continue; }
				q.clear();
				for (int j = i; j <= hi; j++) q.push_back(name(j));
				cl.merge_answered = false;
				which.push_back((uint64_t)i);
			}
			std::vector<int64_t> some(which.size(), 0);
			if (which.empty()) have = true;
			else if (be_.merge_some(centres, delta, which, some)) {
				for (size_t w = 0; w < which.size(); w++) best[(size_t)which[w]] = some[w];
				have = true;
			} else have = be_.merge_all(centres, delta, best);          // (a backend without the subset form answers everything again)
			if (have)
				for (uint64_t i : which) {
					Cluster& cl = part[(size_t)i];
					cl.merge_partner = best[(size_t)i] > (int64_t)i ? (int32_t)(best[(size_t)i] - (int64_t)i) : 0;
					cl.merge_answered = true;
				}
			else for (uint64_t i : which) part[(size_t)i].merge_answered = false;
		}
		uprof.merge += seconds_since(t0);
		t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < n; i++) {
			const long ret = have ? (long)best[(size_t)i] : be_.merge(centres, i, i + 1, std::min(n - 1, i + delta));
			if (ret > i) {
				std::vector<SeqRecord*>& to = part[(size_t)ret].members;
				const std::vector<SeqRecord*>& from = part[(size_t)i].members;
				to.insert(to.end(), from.begin(), from.end());
				part[(size_t)ret].version++;
				part[(size_t)i].merged_away = true;
			}
		}
		const size_t before = part.size();
		part.erase(std::remove_if(part.begin(), part.end(), [](const Cluster& c) { return c.merged_away; }), part.end());
		uprof.book += seconds_since(t0);
		return part.size() != before;
	}

	// bvec::pop; a backend that keeps the window learns that the position is gone
	SeqRecord* take_first(LengthBins& store) {
		uint32_t fixed = 0;
		SeqRecord* r = store.take_first(&fixed);
		if (r && ranged_) be_.kill(fixed);
		return r;
	}

	ClusterBackend& be_;
	std::ostream& log_;
	bool ranged_ = false;
	uint32_t serial_ = 0;
	std::vector<uint32_t> close_;
	std::chrono::steady_clock::time_point t0_ = std::chrono::steady_clock::now();
};

}  // namespace msc
