// msc_fasta.hpp -- FASTA reading as the reference does it (nonltr/ChromListMaker.cpp:24-48,117-165): CR / LF / CRLF line ends, lines
// that start with a blank are skipped, text in front of the first header is dropped; --single-file joins the records of a file by 50 'N'
// under its first header (:123-147). Shared by msc_cluster, msc_fastcar and the sharded-driver test.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <istream>
#include <memory>
#include <stdexcept>
#include <thread>
#include <unistd.h>
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

// Cuts [p, end) at its line ends into records. A line ends at LF, at CR, or at CR LF (one end); the last line may have none; a line that
// starts with a blank is skipped; text in front of the first header is dropped (nonltr/ChromListMaker.cpp:24-48,117-165).
inline void cut_fasta(const char* p, const char* const end, std::vector<std::string>& headers, std::vector<std::string>& seqs, bool single_file, bool have) {
	while (p < end) {
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

// (r05: the file is read in one piece and cut at its line ends in memory -- a character at a time through the stream buffer was 1.5 s of
// BASELINE cfg3's 1 GB of FASTA; then by several threads, each reading its share of the bytes and cutting the records that START in its
// share -- a share begins at the first line that opens with '>' at or behind its first byte: 1.1 -> 0.3 s; safe_getline stays for the
// callers that read streams. --single-file joins every record into one and stays with one thread.)
inline void read_fasta(const std::string& path, std::vector<std::string>& headers, std::vector<std::string>& seqs, bool single_file = false) {
	std::FILE* f = std::fopen(path.c_str(), "rb");
	if (!f) throw std::runtime_error("cannot open " + path);
	std::fseek(f, 0, SEEK_END);
	const long fsize = std::ftell(f);
	std::fseek(f, 0, SEEK_SET);
	const size_t size = fsize > 0 ? (size_t)fsize : 0;
	std::unique_ptr<char[]> buf(new char[size + 1]);          // (not value-initialised: a gigabyte of zeros first is 0.2 s)
	size_t share_min = (size_t)32 << 20;          // (MSC_FASTA_SHARE=bytes: tests cut small files into many shares)
	if (const char* e = std::getenv("MSC_FASTA_SHARE")) { const long v = std::atol(e); if (v > 0) share_min = (size_t)v; }
	size_t n_threads = 1;
	if (!single_file && size >= 2 * share_min) {
		n_threads = std::min<size_t>({(size_t)16, (size_t)std::max(1u, std::thread::hardware_concurrency()), size / share_min});
		if (const char* e = std::getenv("MSC_HOST_THREADS")) { const long v = std::atol(e); if (v > 0) n_threads = std::min(n_threads, (size_t)v); }
	}
	size_t got = 0;
	if (n_threads == 1) {
		while (got < size) {
			const size_t r = std::fread(buf.get() + got, 1, size - got, f);
			if (r == 0) break;
			got += r;
		}
		std::fclose(f);
		cut_fasta(buf.get(), buf.get() + got, headers, seqs, single_file, false);
		return;
	}
	// every thread reads its share of the bytes (pread: no shared file position) ...
	const int fd = fileno(f);
	std::vector<size_t> read_n(n_threads, 0);
	{
		std::vector<std::thread> th;
		for (size_t t = 0; t < n_threads; t++)
			th.emplace_back([&, t] {
				const size_t a = size * t / n_threads, b = size * (t + 1) / n_threads;
				size_t at = a;
				while (at < b) {
					const ssize_t r = ::pread(fd, buf.get() + at, b - at, (off_t)at);
					if (r <= 0) break;
					at += (size_t)r;
				}
				read_n[t] = at - a;
			});
		for (auto& x : th) x.join();
	}
	std::fclose(f);
	for (size_t t = 0; t < n_threads; t++) {          // (a file that shrank under the read: what is whole up to the first short share)
		got += read_n[t];
		if (read_n[t] != size * (t + 1) / n_threads - size * t / n_threads) break;
	}
	// ... and cuts the records that start in it: share t begins at the first header line at or behind byte size * t / n
	const char* const base = buf.get();
	std::vector<size_t> start(n_threads + 1, got);
	start[0] = 0;
	for (size_t t = 1; t < n_threads; t++) {
		size_t i = std::max(start[t - 1], std::min(got, size * t / n_threads));
		while (i < got && !(base[i] == '>' && (i == 0 || base[i - 1] == '\n' || base[i - 1] == '\r'))) i++;
		start[t] = i;
	}
	std::vector<std::vector<std::string> > hs(n_threads), ss(n_threads);
	{
		std::vector<std::thread> th;
		for (size_t t = 0; t < n_threads; t++)
			th.emplace_back([&, t] { cut_fasta(base + start[t], base + start[t + 1], hs[t], ss[t], false, false); });
		for (auto& x : th) x.join();
	}
	size_t total = headers.size();
	for (size_t t = 0; t < n_threads; t++) total += hs[t].size();
	headers.reserve(total);
	seqs.reserve(total);
	for (size_t t = 0; t < n_threads; t++) {
		for (auto& h : hs[t]) headers.push_back(std::move(h));
		for (auto& q : ss[t]) seqs.push_back(std::move(q));
	}
}

}  // namespace msc
