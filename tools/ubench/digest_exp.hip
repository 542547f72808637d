// digest_exp.hip -- timing experiments on the structure of k_pair_digest_multi (run on the GPU box; random data, results unchecked).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../meshclust2_amd/csrc digest_exp.hip -o digest_exp && ./digest_exp [M]
// Each variant removes or changes ONE element of the loop so its cost can be read off the difference:
//   0 baseline   1 no wave reduction (per-lane sums stored)   2 no barrier   3 no record store   4 no arithmetic (DMA + barrier + LDS reads only)
//   5 reduction folded with bank-masked DPP (fewer ops; scattered dword stores)   6 records laid out so a workgroup step writes one 256-byte run
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "msc_wave.h"

namespace {
constexpr int kBlock = 256, kWaves = 4;
constexpr uint32_t kTileBytes = 4096, kPieceBytes = 1024;

template <bool NT>
__device__ __forceinline__ void dma_piece(const uint8_t* lane_src, uint32_t lds_dst) {
	uint32_t keep;
	if constexpr (NT)
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_src), "s"(lds_dst) : "memory");
	else
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_src), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t pack_u8(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }

template <int NB, int EXP, int WPS>
__global__ void __launch_bounds__(kBlock, WPS) k_exp(const uint8_t* __restrict__ cand_dg, uint64_t slot_bytes, uint32_t m, const uint8_t* __restrict__ q_dg,
                                                     uint32_t n_q, uint32_t S, uint32_t G, u32x4* __restrict__ partials16) {
	constexpr int TQ = 4, D = NB - 1, NC = 4;
	extern __shared__ __attribute__((aligned(16))) uint8_t s_ring[];
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t rest = blockIdx.x;
	const uint32_t s = rest % S, g = rest / S;
	if (g >= G) return;
	const uint32_t q0 = wib * TQ;
	uint32_t qc[TQ][NC], qp[TQ][8];
#pragma unroll
	for (int j = 0; j < TQ; j++) {
		const u32x4* p = reinterpret_cast<const u32x4*>(q_dg + (uint64_t)(q0 + j) * slot_bytes + (uint64_t)s * kTileBytes) + lane;
		const u32x4 v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192];
		qc[j][0] = pack_u8(v0.x, v0.y); qc[j][1] = pack_u8(v0.z, v0.w); qc[j][2] = pack_u8(v1.x, v1.y); qc[j][3] = pack_u8(v1.z, v1.w);
		qp[j][0] = v2.x; qp[j][1] = v2.y; qp[j][2] = v2.z; qp[j][3] = v2.w; qp[j][4] = v3.x; qp[j][5] = v3.y; qp[j][6] = v3.z; qp[j][7] = v3.w;
	}
#pragma unroll
	for (int j = 0; j < TQ; j++) {
#pragma unroll
		for (int i = 0; i < NC; i++) asm volatile("" : "+v"(qc[j][i]));
#pragma unroll
		for (int i = 0; i < 8; i++) asm volatile("" : "+v"(qp[j][i]));
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)s_ring) + wib * kPieceBytes;
	const uint32_t n_iter = (m - g + G - 1) / G;
	const uint64_t src_off = (uint64_t)s * kTileBytes + wib * kPieceBytes + lane * 16u;
	auto fetch = [&](uint32_t it, uint32_t slot_idx) {
		const uint32_t cand = g + (it < n_iter ? it : n_iter - 1) * G;
		dma_piece<(EXP >= 10)>(cand_dg + (uint64_t)cand * slot_bytes + src_off, ring_lds + slot_idx * kTileBytes);
	};
#pragma unroll
	for (int d = 0; d < D; d++) fetch((uint32_t)d, (uint32_t)d);
	const uint32_t row = lane >> 4;
	const uint32_t jrow = ((row & 1) << 1) | (row >> 1);
	const bool owner = (lane & 15) == 15;
	u32x4* out_ptr = (EXP == 6 || EXP == 10) ? partials16 + ((uint64_t)g * S + s) * 16 + q0 + jrow : partials16 + ((uint64_t)(q0 + jrow) * m + g) * S + s;
	const uint64_t out_step = (EXP == 6 || EXP == 10) ? (uint64_t)G * S * 16 : (uint64_t)G * S;
	// EXP 5: after the folded reduction lane (16*r + 4*t + 3) of row r holds value t (0 manh, 1 dot, 2 emd) of query jrow(r)... see below
	const uint32_t bank = (lane >> 2) & 3;                 // record word: bank 0 manh -> 0, bank 1 emd -> 2, bank 2 dot -> 1
	uint32_t* out32 = reinterpret_cast<uint32_t*>(partials16 + ((uint64_t)(q0 + jrow) * m + g) * S + s) + (bank == 0 ? 0 : bank == 1 ? 2 : 1);
	const bool owner5 = (lane & 3) == 0 && bank < 3;
	uint32_t rd = 0, wr = D % NB;
	for (uint32_t it = 0; it < n_iter; it++) {
		constexpr int STORES = (EXP == 3 || EXP == 4 || EXP == 14) ? 0 : 1;
		if (it >= (uint32_t)D) wait_vm<D - 1 + D * STORES>(); else wait_vm<D - 1>();
		if constexpr (EXP != 2) __builtin_amdgcn_s_barrier();
		fetch(it + D, wr);
		wr = wr + 1 == NB ? 0 : wr + 1;
		const u32x4* sl = reinterpret_cast<const u32x4*>(s_ring + rd * kTileBytes) + lane;
		const u32x4 v0 = sl[0], v1 = sl[64], v2 = sl[128], v3 = sl[192];
		if constexpr (EXP == 4 || EXP == 14) {
			u32x4 t = v0 + v1 + v2 + v3;
			if (t.x == 0x12345678u && t.y == 77u) out_ptr[0] = t;      // practically never
		} else {
			uint32_t cc[NC];
			cc[0] = pack_u8(v0.x, v0.y); cc[1] = pack_u8(v0.z, v0.w); cc[2] = pack_u8(v1.x, v1.y); cc[3] = pack_u8(v1.z, v1.w);
			const uint32_t cp[8] = {v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
			uint32_t manh[TQ], dot[TQ], emd[TQ];
#pragma unroll
			for (int j = 0; j < TQ; j++) {
				manh[j] = 0; dot[j] = 0; emd[j] = 0;
#pragma unroll
				for (int i = 0; i < NC; i++) {
					manh[j] = __builtin_amdgcn_sad_u8(cc[i], qc[j][i], manh[j]);
					dot[j] = __builtin_amdgcn_udot4(cc[i], qc[j][i], dot[j], false);
				}
#pragma unroll
				for (int i = 0; i < 8; i++) emd[j] = __builtin_amdgcn_sad_u16(cp[i], qp[j][i], emd[j]);
			}
			if constexpr (EXP == 5) {
				// 12 values -> 3 registers (two swap stages), then the three registers are folded into ONE with bank-masked DPP:
				// within each 16-lane row, lanes 0-3 end up with manh, 4-7 dot, 8-11 emd (quad totals), then two more steps
				auto fold32 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false); return r.x + r.y; };
				auto fold16 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false); return r.x + r.y; };
				uint32_t A = fold16(fold32(manh[0], manh[1]), fold32(manh[2], manh[3]));
				uint32_t B = fold16(fold32(dot[0], dot[1]), fold32(dot[2], dot[3]));
				uint32_t C = fold16(fold32(emd[0], emd[1]), fold32(emd[2], emd[3]));
				// 16-lane rows: any pairing works as long as all 16 lanes end up summed. ror:8 pairs (l, l^8); half_mirror pairs (l, 7-l)
				// inside each 8-lane half; two quad_perms finish the quads. Bank masks merge the three values into ONE register:
				// bank 0 (lanes 0-3) manh, bank 1 emd, bank 2 dot.
				const uint32_t Xa = A + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)A, 0x128, 0xf, 0xf, false);
				const uint32_t Yb = B + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)B, 0x128, 0xf, 0xf, false);
				const uint32_t Zc = C + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)C, 0x128, 0xf, 0xf, false);
				uint32_t P = (uint32_t)__builtin_amdgcn_update_dpp((int)Xa, (int)Yb, 0xe4, 0xf, 0xc, false);      // lanes 8-15 <- Yb
				P += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x141, 0xf, 0xf, false);                      // half_mirror
				const uint32_t Z2 = Zc + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Zc, 0x141, 0xf, 0xf, false);
				P = (uint32_t)__builtin_amdgcn_update_dpp((int)P, (int)Z2, 0xe4, 0xf, 0x2, false);                   // lanes 4-7 <- Z2
				P += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0xb1, 0xf, 0xf, false);                       // quad_perm [1,0,3,2]
				P += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x4e, 0xf, 0xf, false);                       // quad_perm [2,3,0,1]
				if (owner5) asm volatile("global_store_dword %0, %1, off" ::"v"(out32), "v"(P) : "memory");
				out32 += out_step * 4;
			} else {
				u32x4 rec;
				if constexpr (EXP == 1) {
					rec.x = manh[0] + manh[1] + manh[2] + manh[3]; rec.y = dot[0] + dot[1] + dot[2] + dot[3]; rec.z = emd[0] + emd[1] + emd[2] + emd[3];
				} else {
					rec.x = wave_sum4_rows(manh[0], manh[1], manh[2], manh[3]);
					rec.y = wave_sum4_rows(dot[0], dot[1], dot[2], dot[3]);
					rec.z = wave_sum4_rows(emd[0], emd[1], emd[2], emd[3]);
				}
				rec.w = 0;
				if constexpr (EXP == 3) {
					if (rec.x == 0x12345678u && rec.y == 77u) out_ptr[0] = rec;
				} else {
					if (owner) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(out_ptr), "v"(rec) : "memory");
				}
				out_ptr += out_step;
			}
		}
		rd = rd + 1 == NB ? 0 : rd + 1;
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void k_fill(uint32_t* p, size_t n, uint32_t seed) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) { uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; p[i] = x & 0x03030303u; }
}

template <int NB, int EXP, int WPS>
void run(const char* name, const uint8_t* cand, const uint8_t* q, uint32_t m, void* partials) {
	const uint32_t S = 256, n_q = 16;
	const size_t lds = (size_t)NB * kTileBytes;
	int bpc = 0;
	hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)k_exp<NB, EXP, WPS>, kBlock, lds);
	const uint32_t G = 256u * bpc / S;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 4; rep++) {
		hipEventRecord(e0);
		k_exp<NB, EXP, WPS><<<S * G, kBlock, lds>>>(cand, 1u << 20, m, q, n_q, S, G, (u32x4*)partials);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (rep && ms < best) best = ms;
	}
	printf("%-44s NB=%d blocks/CU=%d  %7.3f ms  %6.1f M pairs/s  %5.2f TB/s\n", name, NB, bpc, best, n_q * (double)m / best / 1e3, (double)m * (1 << 20) / best / 1e9);
	fflush(stdout);
}
}  // namespace

int main(int argc, char** argv) {
	const uint32_t m = argc > 1 ? atoi(argv[1]) : 16384;
	uint8_t *cand, *q; void* partials;
	hipMalloc(&cand, (size_t)m << 20); hipMalloc(&q, 16u << 20); hipMalloc(&partials, (size_t)16 * m * 256 * 16);
	k_fill<<<4096, 256>>>((uint32_t*)cand, ((size_t)m << 20) / 4, 1u);
	k_fill<<<1024, 256>>>((uint32_t*)q, (16u << 20) / 4, 7u);
	hipDeviceSynchronize();
	run<4, 0, 1>("baseline", cand, q, m, partials);
	run<2, 0, 1>("baseline", cand, q, m, partials);
	run<8, 0, 1>("baseline", cand, q, m, partials);
	run<4, 0, 4>("baseline, 4 waves/SIMD", cand, q, m, partials);
	run<4, 1, 1>("no wave reduction", cand, q, m, partials);
	run<4, 2, 1>("no barrier", cand, q, m, partials);
	run<4, 3, 1>("no record store", cand, q, m, partials);
	run<4, 4, 1>("no arithmetic (DMA+barrier+LDS reads)", cand, q, m, partials);
	run<8, 4, 1>("no arithmetic (DMA+barrier+LDS reads)", cand, q, m, partials);
	run<4, 6, 1>("records [cand][tile][query]: 256 B runs", cand, q, m, partials);
	run<8, 6, 1>("records [cand][tile][query]: 256 B runs", cand, q, m, partials);
	run<4, 10, 1>("256 B runs + nt loads", cand, q, m, partials);
	run<8, 10, 1>("256 B runs + nt loads", cand, q, m, partials);
	run<4, 14, 1>("no arithmetic + nt loads", cand, q, m, partials);
	run<8, 14, 1>("no arithmetic + nt loads", cand, q, m, partials);
	run<4, 5, 1>("folded DPP reduction, dword stores", cand, q, m, partials);
	run<8, 5, 1>("folded DPP reduction, dword stores", cand, q, m, partials);
	return 0;
}
