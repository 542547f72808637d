// msc_fastcar.cpp -- query x database identity search over the GPU hot path.
//
// The second caller of the path in the reference: fastcar (fastcar/FC_Runner.cpp). For every query, the database
// sequences inside its length window that the classification model calls "close" are reported with the regression
// model's identity estimate:  <query> \t <db> \t <100 * similarity>.  The loop structure, the per-chunk unstable
// std::sort by length, bin_search, format_header and the output formatting follow fastcar/FC_Runner.cpp:389-471,
// 560-611 at one thread (one output file "<prefix>0"); training is out of scope, a two-block weights file is required.
//
//   msc_fastcar <db.fa> --query <q.fa> --recover weights.txt [--output output] [--chunk 10000] [--no-format] [--device 0]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <memory>
#include <string>
#include <vector>

#include "meshclust2_host.hpp"

namespace {

struct Rec { std::string header, seq; };

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

// SingleFileLoader::next (clutil/SingleFileLoader.cpp:45-83): continuation lines that start with blank extend the header
std::vector<Rec> read_fasta(const std::string& path) {
	std::ifstream in(path.c_str());
	if (!in) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(1); }
	std::vector<Rec> out;
	std::string line;
	while (in.good()) {
		if (!safe_getline(in, line)) break;
		if (line.empty()) continue;
		if (line[0] == '>') out.push_back({line, ""});
		else if (out.empty()) continue;
		else if (line[0] == ' ' || line[0] == '\t') {
			bool all = true;
			for (char c : line) if (c != ' ' && c != '\t') all = false;
			if (!all) out.back().header += line;
		} else out.back().seq += line;
	}
	return out;
}

struct Pt { std::string header; uint64_t length; uint32_t slot; };

long bin_search(const std::vector<Pt*>& points, size_t begin, size_t last, size_t length) {      // FC_Runner.cpp:389-407
	if (last < begin) return 0;
	size_t idx = begin + (last - begin) / 2;
	if (points.at(idx)->length == length) {
		while (idx > 0 && points[idx - 1]->length == length) idx--;
		return (long)idx;
	} else if (points.at(idx)->length > length) {
		if (begin == idx) return (long)idx;
		return bin_search(points, begin, idx - 1, length);
	}
	return bin_search(points, idx + 1, last, length);
}

std::string format_header(const std::string& hdr) {                                              // FC_Runner.cpp:409-424
	long len = (long)hdr.length(), b = 0;
	if (!hdr.empty() && hdr[0] == '>') b++;
	for (long i = b; i < len; i++) if (hdr[(size_t)i] == ' ' || hdr[(size_t)i] == '\t') { len = i + 1; break; }
	return hdr.substr((size_t)b, (size_t)(len - b));
}

void load_chunk(msc::PointSet& set, const std::vector<Rec>& recs, size_t off, size_t n, std::vector<Pt>& pts) {
	std::vector<std::string> seqs;
	for (size_t i = 0; i < n; i++) seqs.push_back(recs[off + i].seq);
	set.get_points(0, seqs, /*strip=*/true);               // Loader<T>::get_point(header, base, ...): the string overload
	pts.resize(n);
	for (size_t i = 0; i < n; i++) pts[i] = Pt{recs[off + i].header, set.get_length(i), (uint32_t)i};
}

}  // namespace

int main(int argc, char** argv) {
	std::vector<std::string> files, qfiles;
	std::string weights, output = "output";
	size_t chunk = 10000, qblock = 16;
	bool format = true, sparse = false, report_kernels = false;
	int device = 0;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto need = [&](const char* w) { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", w); std::exit(1); } return std::string(argv[++i]); };
		if (a == "--query" || a == "-q") qfiles.push_back(need("--query"));
		else if (a == "--recover" || a == "-r") weights = need("--recover");
		else if (a == "--output" || a == "-o") output = need("--output");
		else if (a == "--chunk" || a == "-c") chunk = (size_t)std::atol(need("--chunk").c_str());
		else if (a == "--query-block") qblock = std::max<size_t>(1, (size_t)std::atol(need("--query-block").c_str()));
		else if (a == "--no-format" || a == "--noformat") format = false;
		else if (a == "--threads" || a == "-t") need("--threads");
		else if (a == "--kernels") report_kernels = true;   // after the run: the streaming kernels the library picked for the scoring passes, on stderr
		else if (a == "--sparse") sparse = true;        // sparse histogram layout (required for k >= 13; also the faster one for --feat slow models)
		else if (a == "--device") device = std::atoi(need("--device").c_str());
		else files.push_back(a);
	}
	if (files.empty() || qfiles.empty() || weights.empty()) {
		std::fprintf(stderr, "usage: %s <db.fa> --query <q.fa> --recover weights.txt [--output prefix] [--chunk 10000] [--no-format] [--sparse]\n", argv[0]);
		return 1;
	}
	try {
		int k = 0, dtype = 16;
		double similarity = 0.9;
		{
			std::ifstream in(weights.c_str());
			std::string tok;
			while (in >> tok) {
				if (tok == "k:") in >> k;
				else if (tok == "ID:") in >> similarity;
				else if (tok == "Datatype:") { in >> tok; dtype = tok == "uint8_t" ? 8 : tok == "uint16_t" ? 16 : tok == "uint32_t" ? 32 : 64; }
				else if (tok == "n_combos:") break;
			}
		}
		msc::Context ctx(device);
		msc::Predictor pred(ctx, weights);
		std::vector<Rec> db, queries;
		for (const auto& f : files) { auto r = read_fasta(f); db.insert(db.end(), r.begin(), r.end()); }
		for (const auto& f : qfiles) { auto r = read_fasta(f); queries.insert(queries.end(), r.begin(), r.end()); }
		// dense sets are refilled chunk after chunk; a sparse set's entry arena is append-only, so each chunk gets a fresh one
		auto bases_of = [](const std::vector<Rec>& recs, size_t off, size_t n) { uint64_t t = 0; for (size_t i = 0; i < n; i++) t += recs[off + i].seq.size(); return t; };
		auto make_set = [&](const std::vector<Rec>& recs, size_t off, size_t n) {
			return std::unique_ptr<msc::PointSet>(new msc::PointSet(ctx, k, dtype, std::max<size_t>(1, n), sparse ? bases_of(recs, off, n) + 1024 : 0));
		};
		std::unique_ptr<msc::PointSet> qset_p = make_set(queries, 0, std::min(chunk, queries.size()));
		std::unique_ptr<msc::PointSet> dset_p = make_set(db, 0, std::min(chunk, db.size()));
		const std::string delim = format ? "\t" : "!";
		std::ofstream out((output + "0").c_str());
		uint64_t num_pred_pos = 0;
		std::vector<std::string> kernels;
		auto note_kernel = [&] {
			char name[160];
			int q_per_read = 0;
			if (!report_kernels || msc_last_kernel_info(ctx.get(), name, sizeof name, &q_per_read) != MSC_OK) return;
			if (std::find(kernels.begin(), kernels.end(), name) == kernels.end()) kernels.push_back(name);
		};
		for (size_t qo = 0; qo < queries.size(); qo += chunk) {
			std::vector<Pt> qp;
			if (sparse && qo) qset_p = make_set(queries, qo, std::min(chunk, queries.size() - qo));
			msc::PointSet& qset = *qset_p;
			load_chunk(qset, queries, qo, std::min(chunk, queries.size() - qo), qp);
			for (size_t d0 = 0; d0 < db.size(); d0 += chunk) {
				std::vector<Pt> dp;
				if (sparse && (d0 || qo)) dset_p = make_set(db, d0, std::min(chunk, db.size() - d0));
				msc::PointSet& dset = *dset_p;
				load_chunk(dset, db, d0, std::min(chunk, db.size() - d0), dp);
				std::vector<Pt*> pts;
				for (auto& p : dp) pts.push_back(&p);
				std::sort(pts.begin(), pts.end(), [](Pt* a, Pt* b) { return a->length < b->length; });      // FC_Runner.cpp:590-592
				if (pts.empty()) continue;
				// work() (:426-471) for every query of the chunk. Queries are taken in blocks of `qblock` neighbours in LENGTH order, so
				// their length windows nearly coincide and one Q x M pass over the union window serves the block; each query then
				// keeps only its own window, and the lines are written in the reference's order (query order, then window order).
				struct Hit { size_t cand; double sim; };
				std::vector<std::vector<Hit> > hits(qp.size());
				std::vector<size_t> win_start(qp.size()), win_end(qp.size());
				for (size_t qi = 0; qi < qp.size(); qi++) {
					const size_t q_len = qp[qi].length;
					const size_t begin_length = (size_t)(q_len * similarity);
					const size_t end_length = (size_t)(q_len / similarity);
					size_t s0 = (size_t)bin_search(pts, 0, pts.size() - 1, begin_length), e0 = s0;
					while (e0 < pts.size() && pts[e0]->length <= end_length) e0++;
					win_start[qi] = s0;
					win_end[qi] = e0;
				}
				std::vector<size_t> by_len(qp.size());
				for (size_t i = 0; i < by_len.size(); i++) by_len[i] = i;
				std::stable_sort(by_len.begin(), by_len.end(), [&](size_t a, size_t b) { return qp[a].length < qp[b].length; });
				for (size_t b0 = 0; b0 < by_len.size(); b0 += qblock) {
					const size_t nb = std::min(qblock, by_len.size() - b0);
					size_t lo = pts.size(), hi = 0;
					std::vector<uint32_t> q_slots;
					std::vector<size_t> members;
					for (size_t j = 0; j < nb; j++) {
						const size_t qi = by_len[b0 + j];
						if (win_start[qi] >= win_end[qi]) continue;
						lo = std::min(lo, win_start[qi]);
						hi = std::max(hi, win_end[qi]);
						q_slots.push_back(qp[qi].slot);
						members.push_back(qi);
					}
					if (members.empty()) continue;
					std::vector<uint32_t> window;
					for (size_t i = lo; i < hi; i++) window.push_back(pts[i]->slot);
					std::vector<uint8_t> close;
					std::vector<double> sim;
					if (members.size() == 1) pred.search(dset, window, qset, q_slots[0], close, sim);      // pred->close / similarity, one query
					else pred.search_block(dset, window, qset, q_slots, close, sim);
					note_kernel();
					for (size_t j = 0; j < members.size(); j++) {
						const size_t qi = members[j];
						for (size_t i = win_start[qi]; i < win_end[qi]; i++) {
							const size_t at = j * window.size() + (i - lo);
							if (!close[at]) continue;
							hits[qi].push_back(Hit{i, sim[at]});
						}
					}
				}
				for (size_t qi = 0; qi < qp.size(); qi++) {
					const Pt& query = qp[qi];
					for (const Hit& h : hits[qi]) {
						num_pred_pos++;
						if (h.sim > 0) {
							if (format) out << format_header(query.header) << delim << format_header(pts[h.cand]->header) << delim << 100 * h.sim << std::endl;
							else out << query.header << delim << pts[h.cand]->header << delim << 100 * h.sim << std::endl;
						}
					}
				}
			}
		}
		std::cout << "# of predicted positive: " << num_pred_pos << std::endl;
		for (const auto& kn : kernels) std::fprintf(stderr, "kernel: %s\n", kn.c_str());
	} catch (const msc::Error& e) {
		std::fprintf(stderr, "msc error %d: %s\n", e.code, e.what());
		return 3;
	}
	return 0;
}
