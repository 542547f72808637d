// dma_mfma_probe.hip -- what two gfx950 instructions do, before the digest kernel is built on them (run on the GPU box):
//   1. global_load_lds_dwordx3: 12 bytes per lane straight into LDS -- where does lane l's data land? (M0 base + 12 * l is assumed)
//   2. v_mfma_i32_16x16x64_i8 fed the SAME per-lane registers the VALU path holds (lane l = 16 consecutive u8 bins of a histogram, for A the
//      query, for B the candidate): D[r][c] = sum over kb, t of q[lane r + 16 kb][t] * cand[lane c + 16 kb][t], so the DIAGONAL of D summed over
//      r is the dot product of the two 1024-bin tiles. Checked against a scalar loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 dma_mfma_probe.hip -o dma_mfma_probe && ./dma_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k_dma3(const uint8_t* __restrict__ src, uint32_t* __restrict__ out) {
	__shared__ __attribute__((aligned(16))) uint8_t lds[2048];
	const uint32_t lane = threadIdx.x;
	for (uint32_t i = lane; i < 512; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = 0xdeadbeefu;
	__syncthreads();
	const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
	const uint32_t off = lane * 12;
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
	             : "=&s"(keep) : "v"(off), "s"((uint64_t)src), "s"(base) : "memory");
	__syncthreads();
	for (uint32_t i = lane; i < 512; i += 64) out[i] = reinterpret_cast<uint32_t*>(lds)[i];
	// ds_read_b96 of the lane's 12 bytes
	uint32_t a, b, c;
	const uint32_t addr = base + lane * 12;
	asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(3))) uint32_t*>(&out[512 + lane * 4]))) : "v"(addr) : "memory");
	(void)a; (void)b; (void)c;
}

__global__ void k_mfma(const uint32_t* __restrict__ q, const uint32_t* __restrict__ cand, int* __restrict__ out, int reps) {
	const uint32_t lane = threadIdx.x;
	v4i A, B;
	for (int i = 0; i < 4; i++) { A[i] = (int)q[lane * 4 + i]; B[i] = (int)cand[lane * 4 + i]; }
	v4i D = {0, 0, 0, 0};
	for (int r = 0; r < reps; r++) D = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, D, 0, 0, 0);
	// diagonal: lane l holds column l % 16, rows 4 * (l / 16) + v
	const int col = lane & 15, row0 = 4 * (lane >> 4);
	int diag = 0;
	for (int v = 0; v < 4; v++) if (row0 + v == col) diag = D[v];
	for (int off = 32; off >= 1; off >>= 1) diag += __shfl_xor(diag, off, 64);
	if (lane == 0) out[0] = diag;
	out[1 + lane * 4 + 0] = D[0]; out[1 + lane * 4 + 1] = D[1]; out[1 + lane * 4 + 2] = D[2]; out[1 + lane * 4 + 3] = D[3];
}

// throughput: 8 MFMA vs 8 x 4 dot4 per "step" in a loop, per wave
__global__ void k_rate(const uint32_t* __restrict__ q, const uint32_t* __restrict__ cand, int* __restrict__ out, int iters, int use_mfma) {
	const uint32_t lane = threadIdx.x & 63;
	v4i A, B;
	for (int i = 0; i < 4; i++) { A[i] = (int)q[lane * 4 + i]; B[i] = (int)cand[lane * 4 + i]; }
	v4i D0 = {0, 0, 0, 0}, D1 = D0, D2 = D0, D3 = D0;
	uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
	for (int it = 0; it < iters; it++) {
		if (use_mfma) {
			D0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, D0, 0, 0, 0);
			D1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, D1, 0, 0, 0);
			D2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, D2, 0, 0, 0);
			D3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, D3, 0, 0, 0);
		} else {
			for (int i = 0; i < 4; i++) {
				d0 = __builtin_amdgcn_udot4((uint32_t)A[i], (uint32_t)B[i], d0, false);
				d1 = __builtin_amdgcn_udot4((uint32_t)A[i], (uint32_t)B[i], d1, false);
				d2 = __builtin_amdgcn_udot4((uint32_t)A[i], (uint32_t)B[i], d2, false);
				d3 = __builtin_amdgcn_udot4((uint32_t)A[i], (uint32_t)B[i], d3, false);
			}
		}
		asm volatile("" : "+v"(A[0]));
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = D0[0] + D1[1] + D2[2] + D3[3] + (int)(d0 + d1 + d2 + d3);
}

int main() {
	std::vector<uint8_t> h(64 * 12 + 64);
	for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)i;
	uint8_t* dsrc; uint32_t* dout;
	hipMalloc(&dsrc, h.size()); hipMalloc(&dout, 4096 * 4);
	hipMemcpy(dsrc, h.data(), h.size(), hipMemcpyHostToDevice);
	k_dma3<<<1, 64>>>(dsrc, dout);
	std::vector<uint32_t> o(4096);
	hipMemcpy(o.data(), dout, 4096 * 4, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int l = 0; l < 64; l++) for (int b = 0; b < 12; b++) { const uint8_t got = reinterpret_cast<uint8_t*>(o.data())[l * 12 + b]; if (got != (uint8_t)(l * 12 + b)) bad++; }
	printf("global_load_lds_dwordx3: lane l -> LDS base + 12 l : %s (%d mismatching bytes); first words %08x %08x %08x %08x\n", bad ? "NO" : "yes", bad, o[0], o[1], o[2], o[3]);
	bad = 0;
	for (int l = 0; l < 64; l++) for (int w = 0; w < 3; w++) if (o[512 + l * 4 + w] != o[l * 3 + w]) bad++;
	printf("ds_read_b96 returns the lane's 12 bytes: %s\n", bad ? "NO" : "yes");

	std::vector<uint32_t> q(256), c(256);
	srand(7);
	for (int i = 0; i < 256; i++) { uint32_t a = 0, b = 0; for (int j = 0; j < 4; j++) { a |= (uint32_t)(1 + rand() % 100) << (8 * j); b |= (uint32_t)(1 + rand() % 100) << (8 * j); } q[i] = a; c[i] = b; }
	long ref = 0;
	for (int i = 0; i < 256; i++) for (int j = 0; j < 4; j++) ref += (long)((q[i] >> (8 * j)) & 255) * ((c[i] >> (8 * j)) & 255);
	uint32_t *dq, *dc; int* dres;
	hipMalloc(&dq, 1024); hipMalloc(&dc, 1024); hipMalloc(&dres, 4 * (1 + 256 + 16));
	hipMemcpy(dq, q.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
	k_mfma<<<1, 64>>>(dq, dc, dres, 1);
	int got = 0;
	hipMemcpy(&got, dres, 4, hipMemcpyDeviceToHost);
	printf("v_mfma_i32_16x16x64_i8 diagonal sum = %d, scalar dot = %ld : %s\n", got, ref, got == ref ? "equal" : "DIFFERENT");
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	int* big; hipMalloc(&big, 4 * 256 * 4096);
	for (int use = 0; use < 2; use++) {
		k_rate<<<256 * 4, 256>>>(dq, dc, big, 1000, use);
		hipEventRecord(e0);
		k_rate<<<256 * 4, 256>>>(dq, dc, big, 20000, use);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		// per wave per iteration: 4 x (64-bin x 16 lanes-of-16) products = 4 tiles-worth of one query: compare ms directly
		printf("%s: %.3f ms for 20000 iterations of 4 tile-dots per wave, 16 waves per CU\n", use ? "mfma 16x16x64 i8 (4 per iteration)" : "v_dot4_u32_u8 (16 per iteration)", ms);
	}
	return 0;
}
