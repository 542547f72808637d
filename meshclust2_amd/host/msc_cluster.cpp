// msc_cluster.cpp -- the mean-shift driver on ONE GPU: FASTA in, CD-HIT .clstr out.
//
// The clustering logic itself (cluster/ClusterFactory.cpp MS / accumulate / mean_shift_update / merge / print_output,
// cluster/bvec.cpp, cluster/CRunner.cpp do_run) lives in msc_driver.hpp, written against an abstract backend of seven
// operators so that one GPU, one GPU per rank (meshclust2_amd/cluster.py) and the CPU oracle (the gloo test) share it.
// This file is what surrounds it on one GPU: reading FASTA, choosing k and the histogram type by the reference's rules,
// optional training, and GpuBackend -- the operators as direct calls into the C ABI (meshclust2_host.hpp). Histograms
// never leave HBM: the host handles slots, lengths, flags, scalars.
//
// Usage (flag names are the reference's, cluster/CRunner.cpp:243-477):
// Several GPUs: start one process per GPU with the environment torch.distributed.run sets (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR,
// MASTER_PORT; `python -m torch.distributed.run --no-python ... msc_cluster ...`, or meshclust2_amd/cluster.py). Every rank then holds
// 1 / WORLD_SIZE of the points and runs the same clustering logic over msc::ShardedBackend (msc_sharded.hpp): RCCL over xGMI between the
// ranks (MSC_COMM=tcp: plain sockets + host staging, what the tests use when several ranks share one GPU: MSC_ONE_GPU=1).
//   msc_cluster <input.fa> [--recover weights.txt] [--id 0.9] [--kmer K] [--datatype 8|16|32|64]   (no --recover: trains first)
//               [--output output.clstr] [--delta 5] [--iterations 15] [--single-file] [--sparse] [--serial-update] [--no-ranges] [--device 0]
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
#include "msc_driver.hpp"
#include "msc_fasta.hpp"
#include "msc_gpu_engine.hpp"
#ifdef MSC_WITH_RCCL
#include "msc_comm_rccl.hpp"
#endif

namespace {

using msc::read_fasta;

// ------------------------------------------------------------------ the hot path of ONE GPU behind msc::ClusterBackend
// Point handle = slot in the device point set (records are built in input order); centre handle = slot in the centre store.
struct GpuBackend : msc::ClusterBackend {
	msc::Context& ctx;
	msc::PointSet& points;
	msc::Trainer& trn;
	int k, dtype;
	double cutoff;
	std::unique_ptr<msc::PointSet> centres;
	uint64_t n_centres = 0;         // centre slots handed out
	uint64_t n_stored = 0;          // ... of which the first n_stored exist in the store (the rest are queued clones)
	uint64_t centre_arena = 0;      // > 0: sparse centre store with that many entries

	GpuBackend(msc::Context& c, msc::PointSet& p, msc::Trainer& t, int k_, int dt, double cut, uint64_t sparse_arena)
	    : ctx(c), points(p), trn(t), k(k_), dtype(dt), cutoff(cut), centre_arena(sparse_arena) {
		// a sparse slot is a header and a scalar record: room for every point as its own centre costs nothing; dense slots are whole
		// histograms, so that store starts small and doubles
		centres.reset(new msc::PointSet(ctx, k, dtype, centre_arena ? std::max<uint64_t>(256, points.capacity()) : 256, centre_arena));
	}
	// relocate every live centre into a fresh store (exact copies: stale mags survive). Used to grow the slot count and,
	// for the sparse layout, to compact the append-only entry arena.
	// A compaction of the sparse store goes back and forth between two stores of the same size: allocating and freeing a 26 GB
	// arena per compaction was 0.4 s of each at cfg5's full size.
	std::unique_ptr<msc::PointSet> spare;
	uint64_t n_rebuilds = 0;
	double rebuild_s = 0.0;
	void rebuild_centres(uint64_t capacity) {
		const auto t0 = std::chrono::steady_clock::now();
		std::unique_ptr<msc::PointSet> fresh;
		if (centre_arena && spare && spare->capacity() == capacity) { fresh.swap(spare); fresh->clear(); }
		else { spare.reset(); fresh.reset(new msc::PointSet(ctx, k, dtype, capacity, centre_arena)); }
		std::vector<uint32_t> all(n_stored);                          // (queued clones are the last slots and are not in the old store yet)
		for (uint64_t i = 0; i < all.size(); i++) all[i] = (uint32_t)i;
		fresh->copy_batch(all, *centres, all);          // one launch per region (slot by slot: 4 copy commands and a sync per centre)
		centres.swap(fresh);
		if (centre_arena && centres->capacity() == fresh->capacity()) spare.swap(fresh);
		n_rebuilds++;
		rebuild_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	}
	// Center(c->clone()) of the accumulate stage, queued: nothing reads a centre before the update stage, so the clones of many
	// clusters go out as one msc_hist_clone_batch (one by one each is four copy commands and a stream sync on a sparse store)
	std::vector<uint32_t> pend_slot, pend_point;
	void flush_clones() {
		if (pend_slot.empty()) return;
		std::vector<uint32_t> s_, p_;
		s_.swap(pend_slot);
		p_.swap(pend_point);
		with_arena_retry([&] { centres->clone_batch(s_, points, p_); });
		n_stored = n_centres;
	}
	template <class F> void with_arena_retry(F&& f) {      // sparse store: compact once when the arena runs out
		try { f(); }
		catch (const msc::Error& e) {
			if (e.code != MSC_ERR_OOM || centre_arena == 0) throw;
			rebuild_centres(centres->capacity());
			f();
		}
	}

	// the sealed store's order on the device (msc_window): a step passes two positions instead of a rebuilt slot list
	msc_window* window_ = nullptr;
	~GpuBackend() override {
		msc_window_destroy(window_);
		if (std::getenv("MSC_CLUSTER_PROFILE")) std::cout << "centre store: rebuilt " << n_rebuilds << " times, " << rebuild_s << " s" << std::endl;
	}
	bool set_order(const std::vector<uint32_t>& order) override {
		ctx.check(msc_window_create(ctx.get(), points.get(), order.data(), order.size(), &window_));
		return true;
	}
	void get_close_range(uint32_t q, uint64_t first, uint64_t end, std::vector<uint32_t>& close, int64_t& best, bool& is_min) override {
		const uint32_t* list = nullptr;
		uint64_t n = 0;
		int im = 1;
		ctx.check(msc_get_close_window(ctx.get(), trn.feature().get(), cutoff, window_, first, end, points.get(), q, &list, &n, &best, nullptr, &im));
		close.assign(list, list + n);
		is_min = im != 0;
	}
	void kill(uint64_t pos) override {
		const uint32_t p = (uint32_t)pos;
		ctx.check(msc_window_kill(ctx.get(), window_, &p, 1));
	}
	void get_close(uint32_t q, const std::vector<uint32_t>& window, std::vector<uint8_t>& flags, int64_t& pos, bool& is_min) override {
		auto res = trn.get_close(points, window, points, q, is_min);
		pos = std::get<0>(res);
		flags.swap(std::get<2>(res));
	}
	int64_t closest(const std::vector<uint32_t>& members) override { return trn.closest(points, members); }
	uint32_t centre_new(uint32_t point) override {
		if (n_centres == centres->capacity()) { flush_clones(); rebuild_centres(centres->capacity() * 2); }
		const uint32_t slot = (uint32_t)n_centres++;
		pend_slot.push_back(slot);
		pend_point.push_back(point);
		if (pend_slot.size() >= 8192) flush_clones();
		return slot;
	}
	void centre_set(uint32_t centre, uint32_t point) override { flush_clones(); with_arena_retry([&] { centres->set(centre, points, point); }); }
	void filter(uint32_t centre, const std::vector<uint32_t>& pts, std::vector<uint8_t>& keep) override {
		flush_clones();
		keep.assign(pts.size(), 0);
		uint64_t n = 0;
		ctx.check(msc_filter(ctx.get(), trn.feature().get(), cutoff, centres->get(), centre, points.get(), pts.data(), pts.size(), keep.data(), &n));
	}
	long merge(const std::vector<uint32_t>& cs, long current, long begin, long last) override { flush_clones(); return trn.merge(*centres, cs, current, begin, last); }
	bool update_centres(const std::vector<uint32_t>& cs, const std::vector<uint32_t>& pts, const std::vector<uint64_t>& offsets, std::vector<int64_t>& nearest) override {
		flush_clones();
		ctx.check(msc_update_centres(ctx.get(), trn.feature().get(), cutoff, centres->get(), cs.data(), cs.size(), points.get(), pts.data(), offsets.data(),
		                             nearest.data(), nullptr));
		return true;
	}
	bool centre_set_batch(const std::vector<uint32_t>& cs, const std::vector<uint32_t>& pts) override {
		// (a sparse store appends every moved centre's list to its arena in one launch, all or nothing: compact once when it runs
		// out; if even the compacted arena cannot take the whole round, go centre by centre -- each set() frees the list it replaces)
		flush_clones();
		try {
			with_arena_retry([&] { ctx.check(msc_hist_assign_batch(ctx.get(), centres->get(), cs.data(), points.get(), pts.data(), cs.size())); });
		} catch (const msc::Error& e) {
			if (e.code != MSC_ERR_OOM || centre_arena == 0) throw;
			return false;
		}
		return true;
	}
	bool merge_all(const std::vector<uint32_t>& cs, int delta, std::vector<int64_t>& best) override {
		flush_clones();
		ctx.check(msc_merge_all(ctx.get(), trn.feature().get(), cutoff, centres->get(), cs.data(), cs.size(), delta, best.data()));
		return true;
	}
	bool merge_some(const std::vector<uint32_t>& cs, int delta, const std::vector<uint64_t>& which, std::vector<int64_t>& best) override {
		flush_clones();
		ctx.check(msc_merge_some(ctx.get(), trn.feature().get(), cutoff, centres->get(), cs.data(), cs.size(), delta, which.data(), which.size(), best.data()));
		return true;
	}
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
int choose_datatype(msc::Context& ctx, int k, const std::vector<std::string>& seqs, bool sparse) {
	auto largest = [&](int bits) {
		uint64_t most = 0;          // (sparse layout: the probe's entry arena must hold the longest batch)
		for (size_t off = 0; off < seqs.size(); off += 4096) {
			uint64_t b = 0;
			for (size_t i = off; i < std::min(seqs.size(), off + 4096); i++) b += seqs[i].size();
			most = std::max(most, b);
		}
		uint64_t mx = 0;
		for (size_t off = 0; off < seqs.size(); off += 4096) {
			msc::PointSet probe(ctx, k, bits, 4096, sparse ? most + 1024 : 0);      // (a sparse arena is append-only: a fresh probe per batch)
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
                        const std::vector<std::string>& seqs, bool sparse) {
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
	uint64_t train_bases = 0;
	for (const std::string& p_ : pts) train_bases += p_.size();
	msc::PointSet set(ctx, k, dtype, pts.size(), sparse ? train_bases + 1024 : 0);      // (at k >= 13 a dense training set would not fit)
	for (size_t off = 0; off < pts.size(); off += 4096) {
		set.get_points(off, pts.data() + off, std::min(pts.size(), off + 4096) - off);
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
	bool single_file = false, sparse = false, serial_update = false, no_ranges = false;
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
		else if (a == "--no-ranges") no_ranges = true;   // get_close receives a slot list rebuilt on the host every step (the r02 form)
		else if (a == "--sparse") sparse = true;         // sparse histogram layout (required for k >= 13)
		else files.push_back(a);
	}
	if (files.empty()) {
		std::fprintf(stderr, "usage: %s <input.fa> [--recover weights.txt] [--id 0.9] [--kmer K] [--datatype 8|16|32|64] [--output out.clstr] [--delta 5] [--iterations 15]\n"
		                     "       without --recover a model is trained first (--feat fast|slow, --num-templates 300, --min-feat 4, --max-feat 4) and written to --dump (weights.txt)\n", argv[0]);
		return 1;
	}
	const msc::CommEnv env = msc::CommEnv::from_environment();
	// MSC_FORCE_SHARDED: the rank driver (ShardedBackend over RcclComm) also for WORLD_SIZE = 1 -- every collective of the RCCL transport
	// then runs on one GPU (the 1-GPU test of that code path)
	const bool sharded = env.world > 1 || std::getenv("MSC_FORCE_SHARDED") != nullptr;
	if (sharded && no_ranges) { std::fprintf(stderr, "--no-ranges is a switch of the one-process driver: the rank driver always scores position ranges\n"); return 1; }
	if (sharded && !std::getenv("MSC_ONE_GPU")) device = env.local_rank;
	try {
		const auto t_start = std::chrono::steady_clock::now();
		msc::Context ctx(device);
		ctx.set_kernel_timing(false);          // the accumulate loop is one get_close per step: no per-call event records
		if (std::getenv("MSC_CLUSTER_PROFILE"))
			std::cout << "input profile: context " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << " s" << std::endl;
		std::vector<std::string> headers, seqs;
		std::vector<size_t> file_first;                // index of every file's first record (find_k averages per file, then over the files)
		for (const auto& f : files) { file_first.push_back(seqs.size()); read_fasta(f, headers, seqs, single_file); }
		file_first.push_back(seqs.size());
		if (std::getenv("MSC_CLUSTER_PROFILE"))
			std::cout << "input profile: context + FASTA read " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << " s" << std::endl;
		const size_t n = seqs.size();
		if (n == 0) { std::fprintf(stderr, "no sequences\n"); return 1; }
		std::unique_ptr<msc::TcpComm> boot;          // rendezvous of the ranks (and the whole transport under MSC_COMM=tcp)
		if (sharded) boot.reset(new msc::TcpComm(env));
		bool rank_file = false;
		if (weights.empty() && sharded && env.rank != 0) {
			// rank 0 chooses k and the histogram type and trains; the others wait for (k, type, length of the weights text) and then the
			// text itself -- not for a file in a shared working directory -- with no deadline on this one wait (training has none)
			int64_t kd[3] = {0, 0, 0};
			boot->set_timeout(7 * 24 * 3600);
			boot->broadcast(kd, sizeof kd, 0, false);
			boot->set_timeout(300);
			k = (int)kd[0]; dtype = (int)kd[1];
			std::string text((size_t)kd[2], '\0');
			boot->broadcast(&text[0], text.size(), 0, false);
			weights = dump + ".rank" + std::to_string(env.rank);
			std::ofstream(weights.c_str()) << text;
			rank_file = true;          // (this rank's copy of the text rank 0 sent: removed once the model has been read from it)
		}
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
			if (dtype == 0) { dtype = choose_datatype(ctx, k, seqs, sparse); std::cout << "Using " << dtype << " bit histograms" << std::endl; }
			const double id = similarity > 1 ? similarity / 100.0 : similarity;
			const std::string text = train_model(ctx, k, dtype, id, feat_flags, n_templates, min_feat, max_feat, seqs, sparse);
			std::ofstream(dump.c_str()) << text;       // the reference always leaves weights.txt behind (cluster/Trainer.cpp:188-190)
			weights = dump;
			std::cout << "timestamp GLM " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << std::endl;
			if (sharded) {
				int64_t kd[3] = {k, dtype, (int64_t)text.size()};
				boot->broadcast(kd, sizeof kd, 0, false);
				std::string copy = text;
				boot->broadcast(&copy[0], copy.size(), 0, false);
			}
		}
		msc::Trainer trn(ctx, weights, similarity);
		if (k < 0) k = msc_model_k(trn.feature().get());
		if (dtype == 0) {      // "Datatype:" line of the weights file
			std::ifstream in(weights.c_str());
			std::string tok;
			while (in >> tok) if (tok == "Datatype:") { in >> tok; dtype = tok == "uint8_t" ? 8 : tok == "uint16_t" ? 16 : tok == "uint32_t" ? 32 : 64; break; }
		}
		if (rank_file) std::remove(weights.c_str());
		uint64_t total_bases = 0, longest = 0;
		for (const auto& sq : seqs) { total_bases += sq.size(); longest = std::max<uint64_t>(longest, sq.size()); }
		if (sharded) {
			// ---- one GPU per rank: this rank's share of the points, the operators over msc::ShardedBackend
			msc::ShardPlan plan;
			plan.n = n; plan.block = 1000; plan.world = env.world;
			if (const char* b = std::getenv("MSC_SHARD_BLOCK")) plan.block = std::max<uint64_t>(1, std::strtoull(b, nullptr, 10));      // (tests: small inputs over several ranks)
			std::vector<std::string> own((size_t)plan.count(env.rank));
			for (uint64_t l = 0; l < own.size(); l++) own[(size_t)l] = seqs[(size_t)plan.global(env.rank, l)];
			seqs.clear();
			seqs.shrink_to_fit();
			msc::GpuShardEngine engine(ctx, trn, k, dtype, sparse, similarity, own, longest, total_bases, n);
			own.clear();
			std::unique_ptr<msc::Comm> fabric;
			const char* want = std::getenv("MSC_COMM");
			msc::Comm* comm = boot.get();
			engine.attach(*boot);
			if (!want || std::string(want) != "tcp") {
#ifdef MSC_WITH_RCCL
				fabric.reset(new msc::RcclComm(ctx.get(), *boot));
				engine.attach(*fabric);
				comm = fabric.get();
#else
				throw std::runtime_error("this msc_cluster was built without RCCL: set MSC_COMM=tcp");
#endif
			}
			msc::ShardedBackend be(engine, *comm, n, plan.block);
			std::vector<msc::SeqRecord> records(n);
			for (size_t i = 0; i < n; i++) { records[i].header = headers[i]; records[i].length = be.lengths()[i]; }
			std::ofstream quiet;          // (an unopened stream swallows the log of the ranks that do not report)
			msc::MeanShift ms(be, env.rank == 0 ? (std::ostream&)std::cout : (std::ostream&)quiet);
			ms.batch_update = !serial_update;
			ms.run(records, similarity, iterations, delta, env.rank == 0 ? output.c_str() : nullptr);
			if (env.rank == 0)
				std::cout << "collectives: broadcast " << comm->calls.broadcast << " all_gather " << comm->calls.all_gather << " all_reduce " << comm->calls.all_reduce << " bytes "
				          << comm->calls.bytes << " | get_close steps " << be.ops.get_close << " collectives " << be.ops.get_close_collectives << " overflow " << be.ops.get_close_overflow
				          << " | closest " << be.ops.closest << " update chunks " << be.ops.update_chunks << " set chunks " << be.ops.set_chunks << std::endl;
			return 0;
		}
		msc::PointSet points(ctx, k, dtype, n, sparse ? total_bases + 1024 : 0);
		const size_t chunk = 8192;
		for (size_t off = 0; off < n; off += chunk) points.get_points(off, seqs.data() + off, std::min(n, off + chunk) - off);
		if (std::getenv("MSC_CLUSTER_PROFILE"))
			std::cout << "input profile: histograms built at " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << " s" << std::endl;
		std::vector<msc::SeqRecord> records(n);
		const std::vector<uint64_t> lens = points.get_lengths(0, n);      // (one call: a read-back per point costs 8 us each)
		for (size_t i = 0; i < n; i++) {
			records[i].header = headers[i];
			records[i].length = lens[i];                    // slot i == point handle i
		}
		seqs.clear();
		// (sparse centre store: room for every sequence as its own centre TWICE -- a round of the update stage appends the new list of
		// every moved centre before the old ones are compacted away)
		uint64_t centre_arena = sparse ? 2 * total_bases + 64 * longest + (1 << 20) : 0;
		if (const char* e = std::getenv("MSC_CLUSTER_CENTRE_ARENA")) {      // (tests: a snug arena, so that a small run compacts its store)
			if (sparse && std::atoll(e) > 0) centre_arena = (uint64_t)std::atoll(e);
		}
		GpuBackend gpu(ctx, points, trn, k, dtype, similarity, centre_arena);
		msc::MeanShift ms(gpu, std::cout);
		ms.batch_update = !serial_update;
		ms.use_ranges = !no_ranges;
		ms.run(records, similarity, iterations, delta, output.c_str());
	} catch (const msc::Error& e) {
		std::fprintf(stderr, "msc error %d: %s\n", e.code, e.what());
		if (sharded) std::_Exit(3);          // (no destructors: the other ranks must see this one go, not wait in a collective)
		return 3;
	} catch (const std::exception& e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		if (sharded) std::_Exit(1);
		return 1;
	}
	return 0;
}
