// fasta_time.cpp -- msc::read_fasta alone on a file, three times: g++ -O2 -std=c++17 -pthread -I meshclust2_amd/host tools/probes/fasta_time.cpp -o /tmp/fasta_time
//   /tmp/fasta_time file.fa      (MSC_HOST_THREADS=n: at most n reading threads; MSC_FASTA_SHARE=bytes: a thread's share at least)
#include "msc_fasta.hpp"
#include <chrono>
#include <cstdio>
int main(int argc, char** argv) {
	if (argc < 2) return 2;
	for (int rep = 0; rep < 3; rep++) {
		std::vector<std::string> h, s;
		const auto t0 = std::chrono::steady_clock::now();
		msc::read_fasta(argv[1], h, s);
		size_t bases = 0;
		for (const auto& x : s) bases += x.size();
		std::printf("%zu records, %zu bases in %.3f s\n", h.size(), bases, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
	}
	return 0;
}
