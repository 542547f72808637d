// msc_fasta.hpp -- FASTA reading as the reference does it (nonltr/ChromListMaker.cpp:24-48,117-165): CR / LF / CRLF line ends, lines
// that start with a blank are skipped, text in front of the first header is dropped; --single-file joins the records of a file by 50 'N'
// under its first header (:123-147). Shared by msc_cluster, msc_fastcar and the sharded-driver test.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// (r05: the file is read in one piece and cut at its line ends in memory -- a character at a time through the stream buffer was 1.5 s of
// BASELINE cfg3's 1 GB of FASTA; safe_getline stays for the callers that read streams)
inline void read_fasta(const std::string& path, std::vector<std::string>& headers, std::vector<std::string>& seqs, bool single_file = false) {
	std::FILE* f = std::fopen(path.c_str(), "rb");
	if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(1); }
	std::string buf;
	{
		std::fseek(f, 0, SEEK_END);
		const long size = std::ftell(f);
		std::fseek(f, 0, SEEK_SET);
		if (size > 0) buf.resize((size_t)size);
		size_t got = 0;
		while (got < buf.size()) {
			const size_t r = std::fread(&buf[got], 1, buf.size() - got, f);
			if (r == 0) break;
			got += r;
		}
		buf.resize(got);
		std::fclose(f);
	}
	bool have = false;
	const char* p = buf.data();
	const char* const end = p + buf.size();
	while (p < end) {
		// a line ends at LF, at CR, or at CR LF (one end); the last line may have none
		const char* e = p;
		while (e < end && *e != '\n' && *e != '\r') e++;
		const char* next = e;
		if (next < end) next += (*next == '\r' && next + 1 < end && next[1] == '\n') ? 2 : 1;
		const size_t len = (size_t)(e - p);
		if (len && p[0] == '>') {
			if (single_file && have) seqs.back().append(50, 'N');
			else {
				headers.emplace_back(p, len);
				seqs.emplace_back();
				have = true;
			}
		} else if (len && (p[0] == ' ' || p[0] == '\t')) {
			// skipped
		} else if (have && len) {
			seqs.back().append(p, len);
		}
		p = next;
	}
}

}  // namespace msc
