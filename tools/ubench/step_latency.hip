// step_latency.hip -- what one accumulate step costs in launch + sync overhead (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 step_latency.hip -o step_latency && ./step_latency
// (a) the sequence msc_get_close issues: H2D of a slot list from pinned memory, three small kernels, two D2H into pinned memory, one
//     stream sync;  (b) the same nodes captured once into a hipGraph and relaunched;  (c) one kernel + sync, the floor.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_small(const uint32_t* in, uint32_t* out, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = in[i] * 2654435761u + 1u;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
	const uint32_t n = 1024, iters = 5000;
	hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	uint32_t *h_in, *h_out, *h_out2, *d_a, *d_b, *d_c, *d_d;
	CK(hipHostMalloc(&h_in, n * 4)); CK(hipHostMalloc(&h_out, n * 4)); CK(hipHostMalloc(&h_out2, 64));
	CK(hipMalloc(&d_a, n * 4)); CK(hipMalloc(&d_b, n * 4)); CK(hipMalloc(&d_c, n * 4)); CK(hipMalloc(&d_d, n * 4));
	for (uint32_t i = 0; i < n; i++) h_in[i] = i;
	auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	auto seq = [&](hipStream_t s) {
		CK(hipMemcpyAsync(d_a, h_in, n * 4, hipMemcpyHostToDevice, s));
		k_small<<<n / 256, 256, 0, s>>>(d_a, d_b, n);
		k_small<<<n / 256, 256, 0, s>>>(d_b, d_c, n);
		k_small<<<1, 256, 0, s>>>(d_c, d_d, 256);
		CK(hipMemcpyAsync(h_out2, d_d, 64, hipMemcpyDeviceToHost, s));
		CK(hipMemcpyAsync(h_out, d_c, n, hipMemcpyDeviceToHost, s));
	};
	for (int w = 0; w < 100; w++) { seq(st); CK(hipStreamSynchronize(st)); }
	double t0 = now();
	for (uint32_t i = 0; i < iters; i++) { seq(st); CK(hipStreamSynchronize(st)); }
	printf("(a) copy + 3 kernels + 2 copies + sync : %6.2f us per step\n", (now() - t0) / iters);

	hipGraph_t g; hipGraphExec_t ge;
	CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
	seq(st);
	CK(hipStreamEndCapture(st, &g));
	CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
	for (int w = 0; w < 100; w++) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
	t0 = now();
	for (uint32_t i = 0; i < iters; i++) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
	printf("(b) the same as one hipGraph launch   : %6.2f us per step\n", (now() - t0) / iters);

	t0 = now();
	for (uint32_t i = 0; i < iters; i++) { k_small<<<n / 256, 256, 0, st>>>(d_a, d_b, n); CK(hipStreamSynchronize(st)); }
	printf("(c) one kernel + sync                  : %6.2f us per step\n", (now() - t0) / iters);
	t0 = now();
	for (uint32_t i = 0; i < iters; i++) {
		k_small<<<n / 256, 256, 0, st>>>(d_a, d_b, n); k_small<<<n / 256, 256, 0, st>>>(d_b, d_c, n); k_small<<<1, 256, 0, st>>>(d_c, d_d, 256);
		CK(hipStreamSynchronize(st));
	}
	printf("(d) three kernels + sync               : %6.2f us per step\n", (now() - t0) / iters);
	t0 = now();
	for (uint32_t i = 0; i < iters; i++) {
		CK(hipMemcpyAsync(d_a, h_in, n * 4, hipMemcpyHostToDevice, st));
		k_small<<<n / 256, 256, 0, st>>>(d_a, d_b, n);
		CK(hipMemcpyAsync(h_out, d_b, n, hipMemcpyDeviceToHost, st));
		CK(hipStreamSynchronize(st));
	}
	printf("(e) copy + kernel + copy + sync        : %6.2f us per step\n", (now() - t0) / iters);
	// (f) kernel reads the list from pinned host memory and writes its result into pinned host memory: no copy commands at all
	uint32_t *z_in, *z_out; CK(hipHostMalloc(&z_in, n * 4, hipHostMallocMapped)); CK(hipHostMalloc(&z_out, n * 4, hipHostMallocMapped));
	t0 = now();
	for (uint32_t i = 0; i < iters; i++) { k_small<<<n / 256, 256, 0, st>>>(z_in, z_out, n); CK(hipStreamSynchronize(st)); }
	printf("(f) one kernel on mapped host memory   : %6.2f us per step\n", (now() - t0) / iters);
	return 0;
}
