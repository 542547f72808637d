// gemm_bench.hip -- the product kernel of the Q x M pass (msc_pair_gemm.hip) alone on synthetic presence bits of cfg2's density, for
// A/B runs of kernel variants on the GPU box: every variant's P1 is compared with the shipped kernel's, then timed in interleaved rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../meshclust2_amd/csrc gemm_bench.hip -o gemm_bench && ./gemm_bench [candidates] [rounds]
#include "../../meshclust2_amd/csrc/msc_pair_gemm.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }
// `per` random bins set per slot
__global__ void k_fill_kb(uint8_t* kb, uint64_t nbins, uint32_t m, uint32_t per) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (uint64_t)m * per) return;
	const uint64_t slot = i / per, bin = mix(i) % nbins;
	const uint64_t off = msc_kb_offset(slot, bin, nbins);
	atomicOr(reinterpret_cast<uint32_t*>(kb + (off & ~3ull)), (1u << (bin & 15)) << (8 * (off & 2)));
}
__global__ void k_iota(uint32_t* p, uint32_t n, uint32_t mul) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = i * mul; }
__global__ void k_diff(const int32_t* a, const int32_t* b, uint64_t n, unsigned long long* bad, unsigned long long* sum) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (a[i] != b[i]) atomicAdd(bad, 1ull);
	if (a[i]) atomicAdd(sum, (unsigned long long)a[i]);
}

#include "gemm_variants.inc"

int main(int argc, char** argv) {
	const uint32_t m = argc > 1 ? (uint32_t)atoi(argv[1]) : 98304;
	const int rounds = argc > 2 ? atoi(argv[2]) : 7;
	const uint64_t nbins = 262144;
	const uint32_t qn = 128;
	const uint64_t kb_bytes = (uint64_t)(m + 31) / 32 * msc_kb_block_bytes(nbins);
	uint8_t *kb, *qT, *anib;
	uint32_t* qs;
	int32_t *ref, *out, *diff;
	unsigned long long* cnt;
	CK(hipMalloc(&kb, kb_bytes)); CK(hipMemset(kb, 0, kb_bytes));
	CK(hipMalloc(&qT, msc_pair_gemm_qt_bytes(nbins, qn))); CK(hipMalloc(&anib, nbins / 2 * qn));
	CK(hipMalloc(&qs, qn * 4)); CK(hipMalloc(&ref, (size_t)m * qn * 4)); CK(hipMalloc(&out, (size_t)m * qn * 4)); CK(hipMalloc(&diff, (size_t)m * qn * 4)); CK(hipMalloc(&cnt, 16));
	const uint32_t per = 990;
	k_fill_kb<<<(unsigned)(((uint64_t)m * per + 255) / 256), 256>>>(kb, nbins, m, per);
	k_iota<<<1, 128>>>(qs, qn, 701);          // queries = candidates 0, 701, 1402 ..
	CK(hipDeviceSynchronize());
	CK(msc_launch_pair_gemm_queries(0, nbins, kb, nullptr, nullptr, 0, qs, qn, qn, qT, 0, nullptr, nullptr, nullptr, nullptr, anib));
	CK(hipDeviceSynchronize());
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto timed = [&](auto&& launch) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; };
	auto base = [&] { CK(msc_launch_pair_gemm(0, nbins, kb, nullptr, 0, m, qn, 1, nullptr, nullptr, ref, diff, anib)); };
	base();
	CK(hipDeviceSynchronize());
	std::vector<Variant> vs = variants();
	for (auto& v : vs) {
		CK(hipMemset(out, 0xff, (size_t)m * qn * 4));
		v.launch(kb, m, anib, nbins, out);
		CK(hipDeviceSynchronize());
		CK(hipMemset(cnt, 0, 16));
		k_diff<<<(unsigned)(((uint64_t)m * qn + 255) / 256), 256>>>(ref, out, (uint64_t)m * qn, cnt, cnt + 1);
		unsigned long long h[2]; CK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
		printf("%-40s differs from the shipped kernel in %llu of %llu sums (sum of all %llu)\n", v.name, h[0], (unsigned long long)m * qn, h[1]);
	}
	std::vector<std::vector<float>> t(vs.size() + 1);
	for (int r = 0; r < rounds; r++) {
		t[0].push_back(timed(base));
		for (size_t i = 0; i < vs.size(); i++) t[i + 1].push_back(timed([&] { vs[i].launch(kb, m, anib, nbins, out); }));
	}
	const double ops = 2.0 * m * qn * nbins;
	for (size_t i = 0; i <= vs.size(); i++) {
		std::sort(t[i].begin(), t[i].end());
		printf("%-40s median %.3f ms  min %.3f  -> %.2f POPS (of ~10 FP4 dense)\n", i ? vs[i - 1].name : "k_pair_gemm_fp4_dma (shipped)", t[i][t[i].size() / 2], t[i][0], ops / (t[i][0] * 1e-3) / 1e15);
	}
	{          // the phase stamps of the last variant (it ran last): cycles per super-step in each phase, wave 0 of the first 400 workgroups
		std::vector<unsigned long long> stp(2000);
		CK(hipMemcpyFromSymbol(stp.data(), HIP_SYMBOL(g_stamps), stp.size() * 8));
		double tot[5] = {0, 0, 0, 0, 0};
		for (int b = 0; b < 400; b++) for (int i = 0; i < 5; i++) tot[i] += (double)stp[5 * b + i] / 400.0 / (nbins / 256.0);
		printf("phases, shader clocks per super-step (wave 0): issue of the 5 LDS-DMA copies %.0f | read of the candidates' bits %.0f | 16 operand reads + products issued %.0f | vmcnt(1) %.0f | barrier %.0f\n",
		       tot[0], tot[1], tot[2], tot[3], tot[4]);
	}
	return 0;
	// the clock the chip held inside the loop of the probe variant (run last in every round): shader clocks per 100 MHz tick
	std::vector<unsigned long long> st(4096 * 2);
	CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
	std::vector<double> ghz, us;
	for (uint32_t b = 0; b < std::min<uint32_t>(4096, (m + 127) / 128); b++) if (st[2 * b + 1]) { ghz.push_back((double)st[2 * b] / st[2 * b + 1] * 0.1); us.push_back(st[2 * b + 1] * 0.01); }
	if (!ghz.empty()) {
		std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
		printf("in-kernel clock of the probe: median %.3f GHz (min %.3f, max %.3f); a workgroup's loop took median %.1f us (min %.1f, max %.1f)\n", ghz[ghz.size() / 2], ghz[0], ghz.back(), us[us.size() / 2], us[0], us.back());
	}
	return 0;
}
