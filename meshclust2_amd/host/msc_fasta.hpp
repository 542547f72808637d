// msc_fasta.hpp -- FASTA reading as the reference does it (nonltr/ChromListMaker.cpp:24-48,117-165): CR / LF / CRLF line ends, lines
// that start with a blank are skipped, text in front of the first header is dropped; --single-file joins the records of a file by 50 'N'
// under its first header (:123-147). Shared by msc_cluster, msc_fastcar and the sharded-driver test.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <istream>
#include <string>
#include <vector>

namespace msc {

inline bool safe_getline(std::istream& is, std::string& t) {
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

inline void read_fasta(const std::string& path, std::vector<std::string>& headers, std::vector<std::string>& seqs, bool single_file = false) {
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

}  // namespace msc
