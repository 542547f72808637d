// lanes_exp.hip -- prototype of the "candidate per lane" Q x M kernel (run on the GPU box; random data, results unchecked).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../meshclust2_amd/csrc lanes_exp.hip -o lanes_exp && ./lanes_exp [M]
// Lane l of every wave owns candidate l of the workgroup's 64; the query operands are wave-uniform, so they come from
// SGPRs (scalar loads of a per-launch query stream); no cross-lane reduction and no per-tile partial records exist.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "msc_wave.h"

namespace {
typedef const __attribute__((address_space(4))) uint32_t* cptr_t;      // constant address space: uniform loads become s_load

constexpr uint32_t kChunk = 1024;                 // bytes of one candidate per stage (16 groups of 16 bins, 64 B each)
constexpr uint32_t kCandStride = kChunk + 16;     // LDS stride between candidates: lane l reads bank offset 4*l -> conflict-free b128
constexpr uint32_t kStageBytes = 64 * kCandStride;

__device__ __forceinline__ uint32_t pack_u8(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }

// 1 KiB of one candidate: 64 lanes x 16 B from sbase + voff into LDS at m0
__device__ __forceinline__ void dma_cand(uint64_t sbase, uint32_t voff, uint32_t lds_dst) {
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int W, int TQ, int QW, int QMODE>
__global__ void __launch_bounds__(W * 64) k_lanes(const uint8_t* __restrict__ cand_dg, uint64_t slot_bytes, uint32_t m, const uint32_t* __restrict__ qsd_g,
                                                    uint32_t qpad, uint32_t n_chunks, uint32_t P, u32x4* __restrict__ records) {
	extern __shared__ __attribute__((aligned(16))) uint8_t s_stage[];      // [2][64][kCandStride]
	constexpr int CPW = 64 / W;                  // candidates each wave moves per stage
	constexpr int GW = TQ * QW;                  // dwords of this wave's query stream per group
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t part = blockIdx.x % P, cg = blockIdx.x / P;
	const uint32_t nk = n_chunks / P, k0 = part * nk;
	const uint32_t q0 = wib * TQ;
	(void)qpad;
	// query stream of THIS wave: [wave][group][TQ][QW] dwords, contiguous per wave
	cptr_t qw = (cptr_t)qsd_g + ((uint64_t)wib * (n_chunks * 16) + (QMODE == 1 ? 0u : k0 * 16u)) * GW;
	const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)s_stage);
	const uint32_t voff = lane * 16;
	auto fetch = [&](uint32_t buf, uint32_t kk) {
#pragma unroll
		for (int i = 0; i < CPW; i++) {
			uint32_t c = cg * 64 + wib * CPW + i;
			c = c < m ? c : m - 1;
			asm volatile("" : "+s"(c));      // keep the 64-bit bases out of long-lived SGPRs: recomputed per stage (scalar ops are free here)
			const uint64_t base = (uint64_t)cand_dg + (uint64_t)c * slot_bytes + (uint64_t)(k0 + kk) * kChunk;
			dma_cand(base, voff, lds0 + buf * kStageBytes + (wib * CPW + i) * kCandStride);
		}
	};
	uint32_t manh[TQ], dot[TQ], emd[TQ];
#pragma unroll
	for (int j = 0; j < TQ; j++) { manh[j] = 0; dot[j] = 0; emd[j] = 0; }
	fetch(0, 0);
	for (uint32_t kk = 0; kk < nk; kk++) {
		wait_vm<0>();
		__builtin_amdgcn_s_barrier();
		if (kk + 1 < nk) fetch((kk + 1) & 1, kk + 1);
		const u32x4* lp = reinterpret_cast<const u32x4*>(s_stage + (kk & 1) * kStageBytes + lane * kCandStride);
#pragma unroll
		for (int h = 0; h < 2; h++) {
			uint32_t sink = 0;
			if constexpr (QMODE == 2) {
				// touch every 64-byte line of the eight groups AFTER the ones scored now: they sit in the scalar cache when their turn comes.
				// All loads land in ONE scratch SGPR that stays reserved (it is an operand of the wait below) until they have returned.
				cptr_t nx = qw + 8 * GW;
#pragma unroll
				for (int l = 0; l < (8 * GW * 4 + 63) / 64; l++) asm volatile("s_load_dword %0, %1, %2" : "+s"(sink) : "s"(nx), "n"(l * 64));
			}
#pragma unroll 2
			for (int t = 0; t < 8; t++) {
				const int tt = h * 8 + t;
				const u32x4 v0 = lp[4 * tt], v1 = lp[4 * tt + 1], v2 = lp[4 * tt + 2], v3 = lp[4 * tt + 3];
				const uint32_t c8[4] = {pack_u8(v0.x, v0.y), pack_u8(v0.z, v0.w), pack_u8(v1.x, v1.y), pack_u8(v1.z, v1.w)};
				const uint32_t cp[8] = {v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
				cptr_t qs = qw + t * GW;
#pragma unroll
				for (int j = 0; j < TQ; j++) {
#pragma unroll
					for (int i = 0; i < 4; i++) {
						manh[j] = __builtin_amdgcn_sad_u8(c8[i], qs[j * QW + i], manh[j]);
						dot[j] = __builtin_amdgcn_udot4(c8[i], qs[j * QW + i], dot[j], false);
					}
#pragma unroll
					for (int i = 0; i < 8; i++) emd[j] = __builtin_amdgcn_sad_u16(cp[i], qs[j * QW + 4 + i], emd[j]);
				}
			}
			if constexpr (QMODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(sink));
			if constexpr (QMODE != 1) qw += 8 * GW;
		}
	}
	const uint32_t c = cg * 64 + lane;
	if (c < m) {
#pragma unroll
		for (int j = 0; j < TQ; j++) {
			u32x4 rec; rec.x = manh[j]; rec.y = dot[j]; rec.z = emd[j]; rec.w = 0;
			records[((uint64_t)c * P + part) * 16 + q0 + j] = rec;
		}
	}
}

__global__ void k_fill(uint32_t* p, size_t n, uint32_t seed) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) { uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; p[i] = x & 0x03030303u; }
}

template <int W, int TQ, int QW, int QMODE>
void run(const char* name, const uint8_t* cand, const uint32_t* qsd, uint32_t m, uint32_t P, void* records) {
	const uint32_t n_chunks = 1024, n_q = W * TQ;
	const size_t lds = 2 * kStageBytes;
	(void)hipFuncSetAttribute((const void*)k_lanes<W, TQ, QW, QMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	float best = 1e9f;
	const uint32_t blocks = (m + 63) / 64 * P;
	for (int rep = 0; rep < 4; rep++) {
		(void)hipEventRecord(e0);
		k_lanes<W, TQ, QW, QMODE><<<blocks, W * 64, lds>>>(cand, 1u << 20, m, qsd, n_q, n_chunks, P, (u32x4*)records);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		if (rep && ms < best) best = ms;
	}
	hipError_t e = hipGetLastError();
	printf("%-32s W=%2d TQ=%d QW=%2d P=%2u  %7.3f ms  %6.1f M pairs/s  %5.2f TB/s  %s\n", name, W, TQ, QW, P, best, n_q * (double)m / best / 1e3, (double)m * (1 << 20) / best / 1e9,
	       e == hipSuccess ? "" : hipGetErrorString(e));
	fflush(stdout);
}
}  // namespace

int main(int argc, char** argv) {
	const uint32_t m = argc > 1 ? atoi(argv[1]) : 16384;
	uint8_t* cand; uint32_t* qsd; void* records;
	(void)hipMalloc(&cand, (size_t)m << 20); (void)hipMalloc(&qsd, (size_t)(16384 + 64) * 32 * 16 * 4); (void)hipMalloc(&records, (size_t)m * 16 * 16 * 16);
	k_fill<<<4096, 256>>>((uint32_t*)cand, ((size_t)m << 20) / 4, 1u);
	k_fill<<<1024, 256>>>(qsd, (size_t)16384 * 32 * 16, 7u);
	(void)hipDeviceSynchronize();
	run<8, 2, 12, 0>("lanes", cand, qsd, m, 4, records);
	run<8, 2, 12, 1>("lanes, query stream cached", cand, qsd, m, 4, records);
	run<8, 2, 12, 2>("lanes, scalar-cache prefetch", cand, qsd, m, 4, records);
	run<16, 1, 12, 2>("lanes, scalar-cache prefetch", cand, qsd, m, 4, records);
	run<16, 2, 12, 2>("lanes (32 q), sc prefetch", cand, qsd, m, 4, records);
	run<8, 4, 12, 2>("lanes (32 q), sc prefetch", cand, qsd, m, 4, records);
	run<8, 1, 12, 2>("lanes (8 q), sc prefetch", cand, qsd, m, 4, records);
	return 0;
}
