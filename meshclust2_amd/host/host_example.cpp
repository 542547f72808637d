// host_example.cpp -- the reference's get_close inner loop written against meshclust2_host.hpp.
//   g++ -std=c++17 -O2 host_example.cpp -L.. -lmeshclust2_hip -Wl,-rpath,'$ORIGIN/..' -o host_example
//   ./host_example weights.txt [cutoff]          (needs an MI355X; exits 3 with the library's message otherwise)
// Prints one line per query: arg-max position, similarity, number of close points, is_min.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "meshclust2_host.hpp"

int main(int argc, char** argv) {
	if (argc < 2) { std::fprintf(stderr, "usage: %s weights.txt [cutoff]\n", argv[0]); return 2; }
	const double cutoff = argc > 2 ? std::atof(argv[2]) : 0.9;
	try {
		msc::Context ctx(0);
		msc::Trainer trn(ctx, argv[1], cutoff);
		const int k = msc_model_k(trn.feature().get());
		std::mt19937 rng(7);
		std::vector<std::string> seqs;
		std::string tmpl;
		for (int i = 0; i < 64; i++) {
			if (i % 8 == 0) { tmpl.clear(); for (int b = 0; b < 1000; b++) tmpl.push_back("ACGT"[rng() & 3]); }
			std::string s = tmpl;
			for (auto& c : s) if (rng() % 100 < 3) c = "ACGT"[rng() & 3];
			seqs.push_back(s);
		}
		msc::PointSet points(ctx, k, 16, seqs.size());
		points.get_points(0, seqs);
		for (uint32_t q = 0; q < 3; q++) {
			std::vector<uint32_t> window;
			for (uint32_t i = 0; i < seqs.size(); i++) if (i != q) window.push_back(i);
			bool is_min = false;
			auto res = trn.get_close(points, window, points, q, is_min);
			size_t n_close = 0;
			for (uint8_t f : std::get<2>(res)) n_close += f;
			std::printf("query %u: best %lld sim %.6f close %zu is_min %d\n", q, (long long)std::get<0>(res), std::get<1>(res), n_close, (int)is_min);
		}
	} catch (const msc::Error& e) {
		std::fprintf(stderr, "msc error %d: %s\n", e.code, e.what());
		return 3;
	}
	return 0;
}
