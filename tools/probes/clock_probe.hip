// clock_probe.hip -- which clock do short kernels between host synchronisations run at? (r05: cfg5's window passes are 40-170 us kernels,
// one stream wait per step.) A kernel of dependent VALU work reads the shader clock counter (clock64: s_memtime) and the constant 100 MHz
// counter (wall_clock64: s_memrealtime) on entry and exit; MHz = shader ticks / (real ticks / 100).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/clock_probe tools/probes/clock_probe.hip && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_spin(uint32_t n, unsigned long long* out, uint32_t* sink) {
	const unsigned long long c0 = clock64(), w0 = wall_clock64();
	uint32_t x = threadIdx.x + blockIdx.x;
	for (uint32_t i = 0; i < n; i++) x = x * 1664525u + 1013904223u;
	const unsigned long long c1 = clock64(), w1 = wall_clock64();
	if (x == 0xdeadbeefu) *sink = x;
	if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
int main() {
	unsigned long long* out;
	uint32_t* sink;
	hipHostMalloc((void**)&out, 16 * 4096);
	hipMalloc((void**)&sink, 4);
	hipStream_t st;
	hipStreamCreate(&st);
	auto run = [&](const char* name, uint32_t n, int launches, int gap_us, bool sync_each) {
		double mhz_sum = 0, us_sum = 0;
		int cnt = 0;
		const auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < launches; i++) {
			k_spin<<<1024, 256, 0, st>>>(n, out + 2 * (i % 2048), sink);
			if (sync_each) {
				hipStreamSynchronize(st);
				if (gap_us) { const auto t = std::chrono::steady_clock::now(); while (std::chrono::steady_clock::now() - t < std::chrono::microseconds(gap_us)) {} }
				if (i >= launches / 2) { mhz_sum += (double)out[2 * (i % 2048)] / ((double)out[2 * (i % 2048) + 1] / 100.0); us_sum += (double)out[2 * (i % 2048) + 1] / 100.0; cnt++; }
			}
		}
		hipStreamSynchronize(st);
		if (!sync_each) for (int i = launches / 2; i < launches && i < 2048; i++) { mhz_sum += (double)out[2 * i] / ((double)out[2 * i + 1] / 100.0); us_sum += (double)out[2 * i + 1] / 100.0; cnt++; }
		const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		printf("%-44s kernel %.1f us at %.0f MHz (second half of %d launches; %.3f s wall)\n", name, us_sum / cnt, mhz_sum / cnt, launches, wall);
	};
	for (int rep = 0; rep < 2; rep++) {
		run("back to back, ~40 us kernels", 4000, 2000, 0, false);
		run("sync after each, ~40 us kernels", 4000, 5000, 0, true);
		run("sync + 20 us of host work, ~40 us kernels", 4000, 5000, 20, true);
		run("sync + 100 us of host work, ~40 us kernels", 4000, 5000, 100, true);
		run("sync after each, ~400 us kernels", 40000, 1000, 0, true);
		run("back to back, ~400 us kernels", 40000, 1000, 0, false);
	}
	return 0;
}
