// digest_exp2.hip -- producer/consumer split of k_pair_digest_multi's loop (run on the GPU box; random data, results unchecked).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../meshclust2_amd/csrc digest_exp2.hip -o digest_exp2 && ./digest_exp2 [M]
// PROD = 0: the product's structure (every wave moves its pieces, waits for them, then scores).
// PROD = 1: a fifth wave moves all eight pieces of a step and waits for them; the four scoring waves only meet it at the barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "msc_wave.h"

namespace {
constexpr uint32_t kTileBytes = 4096, kPieceBytes = 1024;

__device__ __forceinline__ void dma_piece(uint64_t sbase, uint32_t lane_off, uint32_t lds_dst) {
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t pack_u8(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }

__device__ __forceinline__ uint32_t fold12(const uint32_t (&manh)[4], const uint32_t (&dot)[4], const uint32_t (&emd)[4]) {
	auto fold32 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false); return r.x + r.y; };
	auto fold16 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false); return r.x + r.y; };
	auto dpp = [](uint32_t old, uint32_t v, auto ctrl, auto bank) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, decltype(ctrl)::value, 0xf, decltype(bank)::value, false); };
	using std::integral_constant;
	const uint32_t A = fold16(fold32(manh[0], manh[2]), fold32(manh[1], manh[3]));
	const uint32_t B = fold16(fold32(emd[0], emd[2]), fold32(emd[1], emd[3]));
	const uint32_t C = fold16(fold32(dot[0], dot[2]), fold32(dot[1], dot[3]));
	const integral_constant<int, 0x128> ror8; const integral_constant<int, 0x141> half_mirror; const integral_constant<int, 0xe4> ident;
	const integral_constant<int, 0xb1> swap1; const integral_constant<int, 0x4e> swap2; const integral_constant<int, 0xf> all;
	const uint32_t Xa = A + dpp(0, A, ror8, all), Yb = B + dpp(0, B, ror8, all), Zc = C + dpp(0, C, ror8, all);
	uint32_t P = dpp(Xa, Yb, ident, integral_constant<int, 0xc>());
	P += dpp(0, P, half_mirror, all);
	const uint32_t Z2 = Zc + dpp(0, Zc, half_mirror, all);
	P = dpp(P, Z2, ident, integral_constant<int, 0x2>());
	P += dpp(0, P, swap1, all);
	P += dpp(0, P, swap2, all);
	return P;
}

template <int NB, int PROD, int PRIO = 0>
__global__ void __launch_bounds__(PROD ? 320 : 256) k_exp(const uint8_t* __restrict__ cand_dg, uint64_t slot_bytes, uint32_t m, const uint8_t* __restrict__ q_dg,
                                                         uint32_t ST, uint32_t G, u32x4* __restrict__ partials16) {
	constexpr int TQ = 4, D = NB - 1, TPI = 2, NC = 4;
	constexpr uint32_t kStepBytes = TPI * kTileBytes;
	extern __shared__ __attribute__((aligned(16))) uint8_t s_ring[];
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t s = blockIdx.x % ST, g = blockIdx.x / ST;
	if (g >= G) return;
	const uint32_t n_iter = (m - g + G - 1) / G;
	const uint32_t lane16 = lane * 16u;
	const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)s_ring);
	if (PROD && wib == 4) {
		if (PRIO) __builtin_amdgcn_s_setprio(3);
		// ---- producer wave: all eight pieces of every step
		auto fetch = [&](uint32_t it, uint32_t slot_idx) {
			const uint32_t cand = g + (it < n_iter ? it : n_iter - 1) * G;
			const uint64_t base = (uint64_t)cand_dg + (uint64_t)cand * slot_bytes + (uint64_t)s * kStepBytes;
#pragma unroll
			for (int p = 0; p < 8; p++) dma_piece(base + p * kPieceBytes, lane16, lds0 + slot_idx * kStepBytes + p * kPieceBytes);
		};
#pragma unroll
		for (int d = 0; d < D; d++) fetch((uint32_t)d, (uint32_t)d);
		uint32_t wr = D % NB;
		for (uint32_t it = 0; it < n_iter; it++) {
			wait_vm<8 * (D - 1)>();
			__builtin_amdgcn_s_barrier();
			fetch(it + D, wr);
			wr = wr + 1 == NB ? 0 : wr + 1;
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		return;
	}
	const uint32_t q0 = wib * TQ;
	uint32_t qc[TQ][TPI][NC], qp[TQ][TPI][8];
#pragma unroll
	for (int j = 0; j < TQ; j++)
#pragma unroll
		for (int u = 0; u < TPI; u++) {
			const u32x4* p = reinterpret_cast<const u32x4*>(q_dg + (uint64_t)(q0 + j) * slot_bytes + (uint64_t)s * kStepBytes + u * kTileBytes) + lane;
			const u32x4 v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192];
			qc[j][u][0] = pack_u8(v0.x, v0.y); qc[j][u][1] = pack_u8(v0.z, v0.w); qc[j][u][2] = pack_u8(v1.x, v1.y); qc[j][u][3] = pack_u8(v1.z, v1.w);
			qp[j][u][0] = v2.x; qp[j][u][1] = v2.y; qp[j][u][2] = v2.z; qp[j][u][3] = v2.w; qp[j][u][4] = v3.x; qp[j][u][5] = v3.y; qp[j][u][6] = v3.z; qp[j][u][7] = v3.w;
		}
#pragma unroll
	for (int j = 0; j < TQ; j++)
#pragma unroll
		for (int u = 0; u < TPI; u++) {
#pragma unroll
			for (int i = 0; i < NC; i++) asm volatile("" : "+v"(qc[j][u][i]));
#pragma unroll
			for (int i = 0; i < 8; i++) asm volatile("" : "+v"(qp[j][u][i]));
		}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	const uint32_t ring_lds = lds0 + wib * kPieceBytes;
	const uint64_t src_off = (uint64_t)s * kStepBytes + wib * kPieceBytes;
	auto fetch = [&](uint32_t it, uint32_t slot_idx) {
		const uint32_t cand = g + (it < n_iter ? it : n_iter - 1) * G;
		const uint64_t base = (uint64_t)cand_dg + (uint64_t)cand * slot_bytes + src_off;
#pragma unroll
		for (int u = 0; u < TPI; u++) dma_piece(base + u * kTileBytes, lane16, ring_lds + slot_idx * kStepBytes + u * kTileBytes);
	};
	if (!PROD) {
#pragma unroll
		for (int d = 0; d < D; d++) fetch((uint32_t)d, (uint32_t)d);
	}
	const bool owner = (lane & 3) == 0;
	uint64_t out_base = (uint64_t)partials16 + (((uint64_t)g * ST + s) * 16 + wib * TQ) * sizeof(u32x4);
	const uint64_t out_step = (uint64_t)G * ST * 16 * sizeof(u32x4);
	uint32_t rd = 0, wr = D % NB;
	for (uint32_t it = 0; it < n_iter; it++) {
		if (!PROD) { if (it >= (uint32_t)D) wait_vm<(TPI + 1) * (D - 1) + 1>(); else wait_vm<TPI * (D - 1)>(); }
		__builtin_amdgcn_s_barrier();
		if (!PROD) { fetch(it + D, wr); wr = wr + 1 == NB ? 0 : wr + 1; }
		uint32_t manh[TQ], dot[TQ], emd[TQ];
#pragma unroll
		for (int j = 0; j < TQ; j++) { manh[j] = 0; dot[j] = 0; emd[j] = 0; }
#pragma unroll
		for (int u = 0; u < TPI; u++) {
			const u32x4* sl = reinterpret_cast<const u32x4*>(s_ring + rd * kStepBytes + u * kTileBytes) + lane;
			const u32x4 v0 = sl[0], v1 = sl[64], v2 = sl[128], v3 = sl[192];
			const uint32_t cc[4] = {pack_u8(v0.x, v0.y), pack_u8(v0.z, v0.w), pack_u8(v1.x, v1.y), pack_u8(v1.z, v1.w)};
			const uint32_t cp[8] = {v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
			for (int j = 0; j < TQ; j++) {
#pragma unroll
				for (int i = 0; i < NC; i++) { manh[j] = __builtin_amdgcn_sad_u8(cc[i], qc[j][u][i], manh[j]); dot[j] = __builtin_amdgcn_udot4(cc[i], qc[j][u][i], dot[j], false); }
#pragma unroll
				for (int i = 0; i < 8; i++) emd[j] = __builtin_amdgcn_sad_u16(cp[i], qp[j][u][i], emd[j]);
			}
		}
		const uint32_t word = fold12(manh, dot, emd);
		if (owner) {
			uint32_t off;
			asm volatile("v_lshrrev_b32 %0, 4, %1\n\tv_and_b32 %0, 0x3c, %0\n\tglobal_store_dword %0, %2, %3" : "=&v"(off) : "v"(lane16), "v"(word), "s"(out_base) : "memory");
		}
		out_base += out_step;
		rd = rd + 1 == NB ? 0 : rd + 1;
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void k_fill(uint32_t* p, size_t n, uint32_t seed) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	// shaped like a real digest: words 0-7 of a lane's 16 (pieces 0,1 of each 4 KiB tile) hold counts (mostly 1, a few 2), words 8-15 prefixes
	for (; i < n; i += stride) {
		uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15;
		const bool counts = ((i >> 8) & 3) < 2;          // 256 words = one 1 KiB piece
		p[i] = counts ? 0x00010001u + ((x & 0xff) == 0 ? 1u : 0u) : ((x % 1000u) | ((x >> 10) % 1000u) << 16);
	}
}

template <int NB, int PROD, int PRIO = 0>
void run(const char* name, const uint8_t* cand, const uint8_t* q, uint32_t m, void* partials) {
	const uint32_t ST = 128, n_q = 16, threads = PROD ? 320 : 256;
	const size_t lds = (size_t)NB * 2 * kTileBytes;
	int bpc = 0;
	(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)k_exp<NB, PROD, PRIO>, threads, lds);
	const uint32_t G = 256u * bpc / ST;
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 5; rep++) {
		(void)hipEventRecord(e0);
		k_exp<NB, PROD, PRIO><<<ST * G, threads, lds>>>(cand, 1u << 20, m, q, ST, G, (u32x4*)partials);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		if (rep && ms < best) best = ms;
	}
	printf("%-34s NB=%d blocks/CU=%d  %7.3f ms  %6.1f M pairs/s  %5.2f TB/s  %s\n", name, NB, bpc, best, n_q * (double)m / best / 1e3, (double)m * (1 << 20) / best / 1e9,
	       hipGetErrorString(hipGetLastError()));
	fflush(stdout);
}
}  // namespace

int main(int argc, char** argv) {
	const uint32_t m = argc > 1 ? atoi(argv[1]) : 32768;
	uint8_t *cand, *q; void* partials;
	(void)hipMalloc(&cand, (size_t)m << 20); (void)hipMalloc(&q, 16u << 20); (void)hipMalloc(&partials, (size_t)16 * m * 128 * 16);
	k_fill<<<4096, 256>>>((uint32_t*)cand, ((size_t)m << 20) / 4, 1u);
	k_fill<<<1024, 256>>>((uint32_t*)q, (16u << 20) / 4, 7u);
	(void)hipDeviceSynchronize();
	for (int rep = 0; rep < 2; rep++) {
		run<4, 0>("every wave moves its pieces", cand, q, m, partials);
		run<6, 0>("every wave moves its pieces", cand, q, m, partials);
		run<5, 1>("producer wave + 4 scoring waves", cand, q, m, partials);
		run<6, 1>("producer wave + 4 scoring waves", cand, q, m, partials);
		run<7, 1>("producer wave + 4 scoring waves", cand, q, m, partials);
		run<6, 1, 1>("producer (prio 3) + 4 scoring", cand, q, m, partials);
		run<5, 1, 1>("producer (prio 3) + 4 scoring", cand, q, m, partials);
	}
	return 0;
}
