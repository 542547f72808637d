// pair_digest.hip -- the "digest" mirror of an 8/16/32-bit histogram set and the Q x M kernel that streams it (gfx950).
//
// Same arithmetic as k_pair_tiles_multi32_ring (pair_features.hip): per (candidate, tile, query) three integer reductions
//     manh = sum |p-q|      dot = sum p*q      emd = sum |prefix(p) - prefix(q)|
// from which the epilogue derives every statistic of predict/Feature.cpp that is in scope (see pair_features.hip).
// Different data: the ring kernel spends 2/3 of its VALU issue slots turning each raw 32-bit tile into the form the
// reductions want (run sums, a wave scan, 16 dependent prefix adds, re-packing two bins per word) -- once per candidate
// tile PER QUERY GROUP -- and measured 69 % VALU-busy at 2.6 TB/s of HBM traffic: VALU-bound, not memory-bound. The
// digest stores that form in HBM instead, built once per histogram (k_digest_build, one pass) and reused by every pass
// of the all-pairs matrix. It is a lossless re-encoding with 4 bytes per bin (the size of a 32-bit set; an 8/16-bit set pays
// 4x / 2x its own bytes for it and uses it from 8 / 6 queries per pass up):
//
//   digest tile = 1024 bins = 64 lanes x 16 words (4 KiB), word w of lane l at byte (w/4)*1024 + l*16 + (w%4)*4
//     words 0..7  : the lane's 16 consecutive bins, two per word        (bin 2i | bin 2i+1 << 16)
//     words 8..15 : inclusive prefix of the EXCESS counts (count - 1) over the whole histogram up to and including
//                   bin 2i | bin 2i+1 << 16. Both histograms carry the same pseudocount baseline (bin index + 1), so
//                   prefix differences are unchanged; an excess prefix is at most the sequence's k-mer count.
//   valid while every count and every excess prefix fits 16 bits (host-checked per set; otherwise the raw kernels run).
//
// k_pair_digest_multi: one workgroup = one tile index s, one candidate group g, SIXTEEN queries (4 waves x TQ = 4 query
// digests held in registers for the whole launch). The four waves share ONE copy of each candidate tile: every wave
// moves a quarter of it (1 KiB) by LDS-DMA into a ring of NB slots, NB-1 tiles ahead; one s_barrier per iteration
// publishes the tile to all four. So a candidate byte is fetched from HBM once per 16 queries, the loop body is
// 4 ds_read_b128 + 16 (8-bit counts) or 24 (16-bit counts) VALU ops per query + one transposed wave reduction per
// four queries, and nothing else.
#include "msc_internal.h"
#include "msc_wave.h"
#include <type_traits>

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr uint32_t kTileBytes = 4096;
constexpr uint32_t kPieceBytes = kTileBytes / kWaves;

// ---------------------------------------------------------------------------------------- digest build
// One wave per RAW tile (64 lanes x R = 64 / 32 / 16 bins of uint8 / uint16 / uint32 in the layout of msc_layout.h): each
// lane scans its run, one wave scan stitches the runs, and the lane emits R / 16 digest runs of 16 bins. A digest tile
// always covers 1024 bins, whatever the bin type of the set.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_digest_build(const uint8_t* __restrict__ bins, uint64_t slot_bytes, const uint8_t* __restrict__ scalars,
                                                         uint64_t scalar_stride, uint8_t* __restrict__ digest, uint64_t dg_slot_bytes, uint64_t first_slot,
                                                         uint64_t n_slots, uint32_t S) {
	constexpr int R = 64 / sizeof(T);         // bins per lane per raw tile (LPT = 4)
	constexpr int NR = R / 16;                // digest runs per lane
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t W = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
	if (W >= n_slots * S) return;
	const uint64_t slot = first_slot + W / S;
	const uint32_t s = (uint32_t)(W % S);
	const u32x4* src = reinterpret_cast<const u32x4*>(bins + slot * slot_bytes + (uint64_t)s * kTileBytes) + lane;
	u32x4 v[4];
#pragma unroll
	for (int l = 0; l < 4; l++) v[l] = __builtin_nontemporal_load(src + 64 * l);
	const T* w = reinterpret_cast<const T*>(v);      // the lane's R logically consecutive bins
	uint32_t t = 0;
#pragma unroll
	for (int r = 0; r < R; r++) t += (uint32_t)w[r];
	const uint64_t* prefix = reinterpret_cast<const uint64_t*>(scalars + slot * scalar_stride + sizeof(MscSlotScalars));
	uint32_t run = (uint32_t)prefix[s] + wave_incl_scan(t) - t;     // sum of every bin before this lane's run
	const uint32_t first_bin = s * (64u * R) + lane * R;
	run -= first_bin;                                                // minus the pseudocount baseline -> excess
#pragma unroll
	for (int n = 0; n < NR; n++) {
		uint32_t ev[16];
#pragma unroll
		for (int r = 0; r < 16; r++) { run += (uint32_t)w[16 * n + r] - 1u; ev[r] = run; }
		u32x4 o[4];
		uint32_t* ow = reinterpret_cast<uint32_t*>(o);
#pragma unroll
		for (int r = 0; r < 8; r++) ow[r] = ((uint32_t)w[16 * n + 2 * r] & 0xffffu) | ((uint32_t)w[16 * n + 2 * r + 1] << 16);
#pragma unroll
		for (int r = 0; r < 8; r++) ow[8 + r] = (ev[2 * r] & 0xffffu) | (ev[2 * r + 1] << 16);
		const uint32_t bin = first_bin + 16 * n;                     // first bin of this digest run
		u32x4* dst = reinterpret_cast<u32x4*>(digest + slot * dg_slot_bytes + (uint64_t)(bin >> 10) * kTileBytes) + ((bin >> 4) & 63);
#pragma unroll
		for (int l = 0; l < 4; l++) dst[64 * l] = o[l];
	}
}

// ---------------------------------------------------------------------------------------- Q x M kernel
// one wave's quarter of a tile: 64 lanes x 16 bytes from global memory (wave-uniform base in SGPRs + per-lane byte offset)
// straight into LDS (no destination VGPRs, no 64-bit vector address arithmetic)
__device__ __forceinline__ void dma_piece(uint64_t sbase, uint32_t lane_off, uint32_t lds_dst, bool stream_once) {
	uint32_t keep;      // m0 is the compiler's: saved and restored
	if (stream_once)    // nontemporal: the bare DMA loop streams 6.3-6.7 TB/s with it against 5.7-6.1 without (tools/ubench/digest_exp.hip)
		asm volatile(
		    "s_mov_b32 %0, m0\n\t"
		    "s_mov_b32 m0, %3\n\t"
		    "s_nop 0\n\t"
		    "global_load_lds_dwordx4 %1, %2 nt\n\t"
		    "s_mov_b32 m0, %0"
		    : "=&s"(keep)
		    : "v"(lane_off), "s"(sbase), "s"(lds_dst)
		    : "memory");
	else
		asm volatile(
		    "s_mov_b32 %0, m0\n\t"
		    "s_mov_b32 m0, %3\n\t"
		    "s_nop 0\n\t"
		    "global_load_lds_dwordx4 %1, %2\n\t"
		    "s_mov_b32 m0, %0"
		    : "=&s"(keep)
		    : "v"(lane_off), "s"(sbase), "s"(lds_dst)
		    : "memory");
}

// Transposed reduction of 4 queries x 3 sums over the 64 lanes into ONE register.
//   v_permlane32_swap a, b : a[32..63] <-> b[0..31]     -> a + b = [a_lo + a_hi | b_lo + b_hi]            (12 -> 6 registers)
//   v_permlane16_swap x, y : x rows 1,3 <-> y rows 0,2  -> x + y = rows [x0+x1 | y0+y1 | x2+x3 | y2+y3]   ( 6 -> 3 registers)
// fed (q0, q2) and (q1, q3), row r then holds 16 partial sums of query r; the three registers (manh, dot, emd) are folded
// inside the 16-lane rows with DPP, bank masks merging them into one register on the way: row_ror:8 pairs lanes (l, l^8),
// row_half_mirror pairs (l, 7-l), two quad_perms finish the quads. Result: every lane of bank 0 holds the row's manh total,
// bank 1 dot, bank 2 emd (bank 3: emd again). 9 swaps + 9 adds + 9 DPP operations for the 12 sums.
// EMD = false: the emd sums are not reduced (banks 2 and 3 then hold the manh total too).
// DOT = false: the dot sums are not reduced either (the products come from the int8 GEMM of msc_dot_gemm.hip): banks 0 and 1 both
// hold the manh total, banks 2 and 3 the emd total -- the kernel stores 8-byte records from banks 0 and 2.
template <bool EMD, bool DOT = true>
__device__ __forceinline__ uint32_t fold12(const uint32_t (&manh)[4], const uint32_t (&dot)[4], const uint32_t (&emd)[4]) {
	auto fold32 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false); return r.x + r.y; };
	auto fold16 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false); return r.x + r.y; };
	auto dpp = [](uint32_t old, uint32_t v, auto ctrl, auto bank) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, decltype(ctrl)::value, 0xf, decltype(bank)::value, false); };
	using std::integral_constant;
	const uint32_t A = fold16(fold32(manh[0], manh[2]), fold32(manh[1], manh[3]));
	uint32_t C = 0;
	if constexpr (DOT) C = fold16(fold32(dot[0], dot[2]), fold32(dot[1], dot[3]));
	const integral_constant<int, 0x128> ror8;
	const integral_constant<int, 0x141> half_mirror;
	const integral_constant<int, 0xe4> ident;
	const integral_constant<int, 0xb1> swap1;
	const integral_constant<int, 0x4e> swap2;
	const integral_constant<int, 0xf> all;
	const uint32_t Xa = A + dpp(0, A, ror8, all);
	uint32_t Zc = 0;
	if constexpr (DOT) Zc = C + dpp(0, C, ror8, all);
	uint32_t P = Xa;
	if constexpr (EMD) {
		const uint32_t B = fold16(fold32(emd[0], emd[2]), fold32(emd[1], emd[3]));
		const uint32_t Yb = B + dpp(0, B, ror8, all);
		P = dpp(Xa, Yb, ident, integral_constant<int, 0xc>());         // lanes 8-15 of every row <- emd
	}
	P += dpp(0, P, half_mirror, all);
	if constexpr (DOT) {
		const uint32_t Z2 = Zc + dpp(0, Zc, half_mirror, all);
		P = dpp(P, Z2, ident, integral_constant<int, 0x2>());                 // lanes 4-7 <- dot
	}
	P += dpp(0, P, swap1, all);
	P += dpp(0, P, swap2, all);
	return P;
}

// Eight queries, manh only (the form that leaves the products to the matrix cores and the earth mover's distance to the ranks): the
// same swaps take 8 -> 4 -> 2 registers whose row r holds 16 partial sums of query r / query 4 + r; after row_ror:8 the second is
// merged into lanes 8-15 of the first, and three more DPP adds leave query r's total in lanes 0-7 of row r, query 4 + r's in lanes 8-15.
__device__ __forceinline__ uint32_t fold8(const uint32_t (&manh)[8]) {
	auto fold32 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false); return r.x + r.y; };
	auto fold16 = [](uint32_t a, uint32_t b) { const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false); return r.x + r.y; };
	const uint32_t A = fold16(fold32(manh[0], manh[2]), fold32(manh[1], manh[3]));
	const uint32_t B = fold16(fold32(manh[4], manh[6]), fold32(manh[5], manh[7]));
	const uint32_t Xa = A + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)A, 0x128, 0xf, 0xf, false);
	const uint32_t Yb = B + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)B, 0x128, 0xf, 0xf, false);
	uint32_t P = (uint32_t)__builtin_amdgcn_update_dpp((int)Xa, (int)Yb, 0xe4, 0xf, 0xc, false);      // lanes 8-15 of every row <- queries 4..7
	P += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x141, 0xf, 0xf, false);
	P += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0xb1, 0xf, 0xf, false);
	P += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x4e, 0xf, 0xf, false);
	return P;
}

// bytes 0 and 2 of `lo` and of `hi`: four 16-bit counts (< 256) -> four bytes
__device__ __forceinline__ uint32_t pack_u8(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }

// EMD = false (the model and the requested statistics do not include the earth mover's distance -- Feature::compute evaluates
// only the model's own single features too, predict/Feature.cpp:156-171): the prefix half of every tile is neither fetched nor
// scored; a step is then two 1 KiB count pieces per tile, one DMA piece per wave, and 16 instead of 32 operations per query.
// DOT = false: the products are left to the matrix cores (msc_dot_gemm.hip): a quarter of this kernel's arithmetic gone.
// TQ = 8 (manh only: EMD = DOT = false, 8-bit counts): eight queries per wave, 32 per workgroup -- the count-only form at four was bound
// by the 7.6 TB/s its workgroups pulled from L2 into LDS (each candidate tile once per 16 queries), not by arithmetic; 4-byte records.
template <int NB, bool U8, int TPI, bool EMD, bool DOT = true, int TQ = 4>
__global__ void __launch_bounds__(kBlock, (U8 && TPI == 2) ? 4 : 1) k_pair_digest_multi(
    const uint8_t* __restrict__ cand_dg, uint64_t slot_bytes, const uint32_t* __restrict__ cand_slots, uint32_t m,
    const uint8_t* __restrict__ q_dg, uint64_t q_slot_bytes, const uint32_t* __restrict__ q_slots, uint32_t n_q, uint32_t ST, uint32_t G,
    uint32_t nqg, bool stream_once, u32x4* __restrict__ partials16) {
	static_assert(NB >= 2 && NB <= 8 && (TPI == 1 || TPI == 2) && (EMD || TPI == 2), "ring depth, tiles per step; the count-only form moves 4 pieces per step");
	static_assert(TQ == 4 || (TQ == 8 && U8 && !EMD && !DOT), "eight queries per wave: manh only");
	constexpr int PF = EMD ? TPI : 1;         // DMA pieces each wave issues per step
	constexpr int D = NB - 1;                 // steps in flight ahead of the one being scored
	constexpr int NC = U8 ? 4 : 8;            // count words per lane per tile
	constexpr uint32_t kStepBytes = TPI * kTileBytes;      // a step = TPI consecutive tiles of one candidate (ST = S / TPI steps per histogram)
	extern __shared__ __attribute__((aligned(16))) uint8_t s_ring[];      // [NB][kStepBytes], shared by the four waves
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// blocks b and b + 8 share an XCD (round-robin dispatch): the query groups that stream the same candidate tiles are
	// placed 8 apart so the second one finds the tile in that XCD's L2. Speed only; any placement is correct.
	const uint32_t lo = blockIdx.x & 7, hi = blockIdx.x >> 3;
	const uint32_t qg = hi % nqg;
	const uint32_t rest = (hi / nqg) * 8 + lo;
	const uint32_t s = rest % ST, g = rest / ST;
	if (g >= G) return;                       // whole workgroup
	const uint32_t q0 = (qg * kWaves + wib) * TQ;
	const bool active = q0 < n_q;             // a wave without queries still moves its quarter of every tile

	uint32_t qc[TQ][TPI][NC], qp[TQ][TPI][EMD ? 8 : 1];
#pragma unroll
	for (int j = 0; j < TQ; j++) {
		const uint32_t qi = q0 + j < n_q ? q0 + j : n_q - 1;      // padded queries score a valid slot; their records are ignored
#pragma unroll
		for (int u = 0; u < TPI; u++) {
			const u32x4* p = reinterpret_cast<const u32x4*>(q_dg + (uint64_t)q_slots[qi] * q_slot_bytes + (uint64_t)s * kStepBytes + u * kTileBytes) + lane;
			const u32x4 v0 = p[0], v1 = p[64];
			if constexpr (U8) {
				qc[j][u][0] = pack_u8(v0.x, v0.y); qc[j][u][1] = pack_u8(v0.z, v0.w); qc[j][u][2] = pack_u8(v1.x, v1.y); qc[j][u][3] = pack_u8(v1.z, v1.w);
			} else {
				qc[j][u][0] = v0.x; qc[j][u][1] = v0.y; qc[j][u][2] = v0.z; qc[j][u][3] = v0.w;
				qc[j][u][4] = v1.x; qc[j][u][5] = v1.y; qc[j][u][6] = v1.z; qc[j][u][7] = v1.w;
			}
			if constexpr (EMD) {
				const u32x4 v2 = p[128], v3 = p[192];
				qp[j][u][0] = v2.x; qp[j][u][1] = v2.y; qp[j][u][2] = v2.z; qp[j][u][3] = v2.w;
				qp[j][u][4] = v3.x; qp[j][u][5] = v3.y; qp[j][u][6] = v3.z; qp[j][u][7] = v3.w;
			} else {
				qp[j][u][0] = 0;
			}
		}
	}
	// The compiler must see the query loads complete HERE: a wait of its own inside the loop (its scoreboard knows nothing
	// of the DMA pieces issued from inline asm) would drain the whole ring every iteration.
#pragma unroll
	for (int j = 0; j < TQ; j++) {
#pragma unroll
		for (int u = 0; u < TPI; u++) {
#pragma unroll
			for (int i = 0; i < NC; i++) asm volatile("" : "+v"(qc[j][u][i]));
#pragma unroll
			for (int i = 0; i < (EMD ? 8 : 1); i++) asm volatile("" : "+v"(qp[j][u][i]));
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the counted waits below start from an empty queue

	// which piece(s) of a step this wave moves: piece `wib` of each tile, or (count-only form) piece wib % 2 of tile wib / 2
	const uint32_t my_piece = EMD ? wib * kPieceBytes : (wib >> 1) * kTileBytes + (wib & 1) * kPieceBytes;
	const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)s_ring) + my_piece;
	const uint32_t n_iter = (m - g + G - 1) / G;       // g < G <= m
	const uint64_t src_off = (uint64_t)s * kStepBytes + my_piece;
	const uint32_t lane16 = lane * 16u;
	// stream_once (host: a single query group, nobody else will want these candidate bytes from L2): nontemporal loads
	auto fetch = [&](uint32_t it, uint32_t slot_idx) {
		const uint32_t cand = g + (it < n_iter ? it : n_iter - 1) * G;      // past the end: re-fetch the last step (keeps the count fixed)
		const uint32_t slot = cand_slots ? cand_slots[cand] : cand;
		const uint64_t base = (uint64_t)cand_dg + (uint64_t)slot * slot_bytes + src_off;      // wave-uniform: scalar arithmetic
#pragma unroll
		for (int u = 0; u < PF; u++) dma_piece(base + u * kTileBytes, lane16, ring_lds + slot_idx * kStepBytes + u * kTileBytes, stream_once);
	};
#pragma unroll
	for (int d = 0; d < D; d++) fetch((uint32_t)d, (uint32_t)d);

	// records: [candidate][query group][step][query in group] -- the four waves of a workgroup write one run per step; wave-uniform
	// base in SGPRs, advanced per step, + the owner lanes' constant offset. With the products: 16 bytes (manh, dot, emd, -), the first
	// lane of every quad stores one word (see fold12 below). Without (DOT = false): 8 bytes (manh, emd), the first lane of every
	// second quad stores -- half the record traffic of this kernel and of the epilogue that sums them.
	// Eight queries per wave: 4 bytes (manh), 32 queries per run, lanes 0 and 8 of every row store.
	constexpr uint32_t kRec = TQ == 8 ? 4 : DOT ? 16 : 8;
	const bool owner = DOT ? (lane & 3) == 0 : (lane & 7) == 0;
	uint64_t out_base = (uint64_t)partials16 + ((((uint64_t)g * nqg + qg) * ST + s) * (kWaves * TQ) + wib * TQ) * kRec;
	const uint64_t out_step = (uint64_t)G * nqg * ST * (kWaves * TQ) * kRec;

	uint32_t rd = 0, wr = D % NB;
	for (uint32_t it = 0; it < n_iter; it++) {
		// Vector-memory operations retire in issue order. Younger than this wave's last piece of step `it`: the TPI pieces of
		// each of the steps it+1 .. it+D-1 and, for a scoring wave past ramp-up, D record stores; fewer stores during ramp-up.
		if (active && it >= (uint32_t)D) wait_vm<(PF + 1) * (D - 1) + 1>(); else wait_vm<PF * (D - 1)>();
		// all four quarters of every tile of step `it` have landed, and every wave has consumed step it-1 (its slot is refilled next)
		__builtin_amdgcn_s_barrier();
		fetch(it + D, wr);
		wr = wr + 1 == NB ? 0 : wr + 1;
		if (active) {
			uint32_t manh[TQ], dot[TQ], emd[TQ];
#pragma unroll
			for (int j = 0; j < TQ; j++) { manh[j] = 0; dot[j] = 0; emd[j] = 0; }
#pragma unroll
			for (int u = 0; u < TPI; u++) {
				const u32x4* sl = reinterpret_cast<const u32x4*>(s_ring + rd * kStepBytes + u * kTileBytes) + lane;
				const u32x4 v0 = sl[0], v1 = sl[64];
				uint32_t cc[NC];
				if constexpr (U8) {
					cc[0] = pack_u8(v0.x, v0.y); cc[1] = pack_u8(v0.z, v0.w); cc[2] = pack_u8(v1.x, v1.y); cc[3] = pack_u8(v1.z, v1.w);
				} else {
					cc[0] = v0.x; cc[1] = v0.y; cc[2] = v0.z; cc[3] = v0.w; cc[4] = v1.x; cc[5] = v1.y; cc[6] = v1.z; cc[7] = v1.w;
				}
				uint32_t cp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
				if constexpr (EMD) {
					const u32x4 v2 = sl[128], v3 = sl[192];
					cp[0] = v2.x; cp[1] = v2.y; cp[2] = v2.z; cp[3] = v2.w; cp[4] = v3.x; cp[5] = v3.y; cp[6] = v3.z; cp[7] = v3.w;
				}
#pragma unroll
				for (int j = 0; j < TQ; j++) {
#pragma unroll
					for (int i = 0; i < NC; i++) {
						if constexpr (U8) {
							manh[j] = __builtin_amdgcn_sad_u8(cc[i], qc[j][u][i], manh[j]);
							if constexpr (DOT) dot[j] = __builtin_amdgcn_udot4(cc[i], qc[j][u][i], dot[j], false);
						} else {
							manh[j] = __builtin_amdgcn_sad_u16(cc[i], qc[j][u][i], manh[j]);
							if constexpr (DOT) dot[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, cc[i]), __builtin_bit_cast(u16x2, qc[j][u][i]), dot[j], false);
						}
					}
					if constexpr (EMD) {
#pragma unroll
						for (int i = 0; i < 8; i++) emd[j] = __builtin_amdgcn_sad_u16(cp[i], qp[j][u][i], emd[j]);
					}
				}
			}
			// 12 per-lane sums -> 4 records of (manh, dot, emd, -): row r of the wave ends up holding query r, its bank b
			// (lanes 4b .. 4b+3 of the row) word b of that query's record; the first lane of each quad stores its word, so the
			// wave writes its 64 bytes of the workgroup's 256-byte run with one dword store
			uint32_t word;
			if constexpr (TQ == 8) word = fold8(manh); else word = fold12<EMD, DOT>(manh, dot, emd);
			if (owner) {
				// byte offset = 16 * row + 4 * bank = lane & 0x3c, recomputed from lane * 16 (live anyway) instead of kept in a
				// register across the loop: the kernel stays within 128 VGPRs
				// (8-byte records: 8 * row + 4 * (bank / 2) = (lane >> 1) & 0x1c)
				uint32_t off;
				// (eight queries: query r of row r at 4 r, query 4 + r (lanes 8-15) at 16 + 4 r = ((lane >> 2) & 0xc) + ((lane << 1) & 0x10))
				if constexpr (TQ == 8) {
					uint32_t half;
					asm volatile("v_lshrrev_b32 %0, 6, %2\n\tv_and_b32 %0, 0xc, %0\n\tv_bfe_u32 %1, %2, 7, 1\n\tv_lshl_or_b32 %0, %1, 4, %0\n\tglobal_store_dword %0, %3, %4"
					             : "=&v"(off), "=&v"(half) : "v"(lane16), "v"(word), "s"(out_base) : "memory");
				} else if constexpr (DOT)
					asm volatile("v_lshrrev_b32 %0, 4, %1\n\tv_and_b32 %0, 0x3c, %0\n\tglobal_store_dword %0, %2, %3"
					             : "=&v"(off) : "v"(lane16), "v"(word), "s"(out_base) : "memory");
				else
					asm volatile("v_lshrrev_b32 %0, 5, %1\n\tv_and_b32 %0, 0x1c, %0\n\tglobal_store_dword %0, %2, %3"
					             : "=&v"(off) : "v"(lane16), "v"(word), "s"(out_base) : "memory");
			}
			out_base += out_step;
		}
		rd = rd + 1 == NB ? 0 : rd + 1;
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring must not be released with fetches in flight
}

template <int NB, bool U8, int TPI, bool EMD = true, bool DOT = true, int TQ = 4>
hipError_t launch_digest_multi(hipStream_t st, uint32_t S, const uint8_t* cand_dg, uint64_t slot_bytes, const uint32_t* cand_slots, uint32_t m,
                               const uint8_t* q_dg, uint64_t q_slot_bytes, const uint32_t* q_slots, uint32_t n_q, void* partials16, int num_cus) {
	const size_t lds = (size_t)NB * TPI * kTileBytes;
	const void* fn = (const void*)k_pair_digest_multi<NB, U8, TPI, EMD, DOT, TQ>;
	int blocks_per_cu = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, fn, kBlock, lds) != hipSuccess || blocks_per_cu < 1) blocks_per_cu = 1;
	const uint32_t nqg = (n_q + TQ * kWaves - 1) / (TQ * kWaves);
	const uint32_t ST = S / TPI;
	// one resident round of equal-length workgroups
	uint64_t G = (uint64_t)num_cus * blocks_per_cu / ((uint64_t)ST * nqg);
	if (G < 1) G = 1;
	if (G > m) G = m;
	const uint64_t rest_pad = ((uint64_t)ST * G + 7) / 8 * 8;
	const unsigned blocks = (unsigned)(rest_pad * nqg);
	k_pair_digest_multi<NB, U8, TPI, EMD, DOT, TQ><<<dim3(blocks), dim3(kBlock), lds, st>>>(cand_dg, slot_bytes, cand_slots, m, q_dg, q_slot_bytes, q_slots, n_q, ST, (uint32_t)G, nqg,
	                                                                            nqg == 1, (u32x4*)partials16);
	return hipGetLastError();
}

}  // namespace

uint64_t msc_digest_slot_bytes(const MscLayout& L) { return L.nbins * 4; }
bool msc_digest_supported(const MscLayout& L) { return L.LPT == 4 && L.esz <= 4 && L.nbins % 1024 == 0; }

hipError_t msc_launch_digest_build(hipStream_t st, const MscLayout& L, const uint8_t* bins, const uint8_t* scalars, uint8_t* digest, uint64_t first_slot,
                                   uint64_t n_slots) {
	if (n_slots == 0) return hipSuccess;
	if (!msc_digest_supported(L)) return hipErrorInvalidValue;
	const uint64_t waves = n_slots * L.S;
	const unsigned blocks = (unsigned)((waves + kWaves - 1) / kWaves);
	const uint64_t ss = msc_scalar_stride(L.S), db = msc_digest_slot_bytes(L);
	if (L.esz == 1) k_digest_build<uint8_t><<<dim3(blocks), dim3(kBlock), 0, st>>>(bins, L.slot_bytes, scalars, ss, digest, db, first_slot, n_slots, L.S);
	else if (L.esz == 2) k_digest_build<uint16_t><<<dim3(blocks), dim3(kBlock), 0, st>>>(bins, L.slot_bytes, scalars, ss, digest, db, first_slot, n_slots, L.S);
	else k_digest_build<uint32_t><<<dim3(blocks), dim3(kBlock), 0, st>>>(bins, L.slot_bytes, scalars, ss, digest, db, first_slot, n_slots, L.S);
	return hipGetLastError();
}

int msc_digest_tiles_per_step(const MscLayout& L, uint64_t max_count) {
	// two tiles per step halve the wave reductions and the partial records per byte streamed; per-lane sums then cover 32 bins
	const bool fits = (L.nbins / 1024) % 2 == 0 && 64ull * 32 * max_count * max_count < (1ull << 32);
	if (!fits) return 1;
	return 2;
}

hipError_t msc_launch_pair_digest_multi(hipStream_t st, const MscLayout& L, const uint8_t* cand_digest, const uint32_t* cand_slots, uint32_t m,
                                        const uint8_t* q_digest, const uint32_t* q_slots, uint32_t n_q, bool counts_fit_u8,
                                        int tiles_per_step, bool need_emd, void* partials16, int num_cus, bool need_dot, int queries_per_wave) {
	if (m == 0 || n_q == 0) return hipSuccess;
	const uint32_t n_tiles = (uint32_t)(L.nbins / 1024);
	if (!msc_digest_supported(L) || (tiles_per_step != 1 && tiles_per_step != 2) || n_tiles % tiles_per_step) return hipErrorInvalidValue;
	const uint64_t db = msc_digest_slot_bytes(L);
	static const int nb_env = [] { const char* e = getenv("MSC_DIGEST_SLOTS"); return e ? atoi(e) : 0; }();
#define MSC_DG_ARGS st, n_tiles, cand_digest, db, cand_slots, m, q_digest, db, q_slots, n_q, partials16, num_cus
	if (queries_per_wave != 4 && (queries_per_wave != 8 || need_dot || need_emd || tiles_per_step != 2 || !counts_fit_u8 || (nb_env != 0 && nb_env != 4))) return hipErrorInvalidValue;
	// the products come from the GEMM (the caller checked the ranges): the forms the bench and the drivers reach, 8-bit counts, two tiles per step
	if (!need_dot && tiles_per_step == 2 && counts_fit_u8 && (nb_env == 0 || nb_env == 4))
		return need_emd ? launch_digest_multi<4, true, 2, true, false>(MSC_DG_ARGS)
		       : queries_per_wave == 8 ? launch_digest_multi<4, true, 2, false, false, 8>(MSC_DG_ARGS) : launch_digest_multi<4, true, 2, false, false>(MSC_DG_ARGS);
#define MSC_DG_NB(U8, TPI)                                                                                                \
	(nb_env == 2 ? launch_digest_multi<2, U8, TPI>(MSC_DG_ARGS) : nb_env == 3 ? launch_digest_multi<3, U8, TPI>(MSC_DG_ARGS) \
	 : nb_env == 6 ? launch_digest_multi<6, U8, TPI>(MSC_DG_ARGS) : nb_env == 8 ? launch_digest_multi<8, U8, TPI>(MSC_DG_ARGS) \
	                                                                            : launch_digest_multi<4, U8, TPI>(MSC_DG_ARGS))
	if (tiles_per_step == 2 && !need_emd && (nb_env == 0 || nb_env == 4))      // count-only form (ring of 4 only)
		return counts_fit_u8 ? launch_digest_multi<4, true, 2, false>(MSC_DG_ARGS) : launch_digest_multi<4, false, 2, false>(MSC_DG_ARGS);
	if (tiles_per_step == 2) return counts_fit_u8 ? MSC_DG_NB(true, 2) : MSC_DG_NB(false, 2);
	return counts_fit_u8 ? MSC_DG_NB(true, 1) : MSC_DG_NB(false, 1);
#undef MSC_DG_NB
#undef MSC_DG_ARGS
}
